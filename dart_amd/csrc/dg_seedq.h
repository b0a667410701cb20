// dart_amd/csrc/dg_seedq.h -- k_seed_q: the seeding stage as a workgroup of waves around LDS work queues.
//
// Same work as k_seed (dg_fm.h: IdentifySeedPairs, AlignmentCandidates.cpp:181-215, around BWT_Search, bwt_search.cpp:139-182)
// and the same per-trip functions (d_begin_issue/finish, d_trip_issue/load/finish), so hits, their order and the
// reference-equivalent counters are the same by construction.  What differs is who runs which trip:
//
//   k_seed     lane = read.  The 64 reads of a wave are wherever their greedy walks happen to be, so every trip runs the code
//              of every mode with 10-25 of 64 lanes active: 785 wave-instructions per trip, 522 M per 2 M reads.
//   k_seed_q   a workgroup keeps NSLOT reads (slots) in LDS: their words and the 48-byte state of their current search.  A slot
//              sits in exactly one of five queues -- begin a search, Occ step, text comparison, locate, free -- and every
//              wave-trip takes up to 64 slots from ONE queue: all its lanes run the same code with the same kind of load.
//              A phase = [barrier] every wave reads the same queue counts and derives the same assignment (fullest queue
//              first, 64 slots per wave) [barrier] pop, load the slot's state, issue, wait, finish, store, push to the queue
//              of the slot's new mode.  A free slot is refilled with the next read of the batch (one wave per phase).
//
// LDS per workgroup: NSLOT x (48 B state + 4 W B read words + 20 B queue entries); 512 slots of 101-base reads = 63 KB.
// Bound: HBM / Infinity-Cache random 64-byte reads (one Occ block, one SA entry, one table entry or 20 B of text per lane-trip).
#pragma once
#include "dg_fm.h"

#define SQ_WAVES   4
#define SQ_THREADS (SQ_WAVES * 64)
enum { SQ_BEGIN = 0, SQ_STEP = 1, SQ_CMP = 2, SQ_LOC = 3, SQ_FREE = 4, SQ_NQ = 5 };     // 0..3 = Search::mode
#define SQ_MAX_PHASES (1u << 20)     // no batch comes near (a workgroup of a 2 M-read batch runs ~300 phases): a safety net -- the host runs the batch again

__device__ __forceinline__ uint32_t sq_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// bytes of dynamic LDS for 2^nslot_lg slots of W-word reads
__host__ __device__ inline size_t sq_lds_bytes(int nslot_lg, int W)
{
    const size_t n = (size_t)1 << nslot_lg;
    return n * 48 + n * 4 * (size_t)W + 2 * n * 2 * SQ_NQ + SQ_WAVES * 64 * 2 + 16 * 4;
}

// what a trip needs besides its slot (kernel-wide, uniform)
struct SqEnv {
    const DIndex &ix; const DParams &pr;
    uint4 *st; uint32_t *rd;
    int NSLOT, W2, K, H, bail_trips; bool direct;
    int multi;                       // k_seed_qf with the full suffix array: intervals of up to `multi` rows are located and compared with the text (0: only single rows)
    DHit *hits; uint32_t *nhits, *nseeds; DHeavy *heavy; unsigned int *n_heavy;
};

// One trip of up to 64 searches that are all in mode MODE (a compile-time constant: every instance holds only its mode's code, and the
// four instances share no registers across a merge -- as ONE body switched by a uniform mode the compiler kept all modes' state alive
// and moved it around: 448 v_mov and 126 SGPR reloads in a 2500-instruction loop).  Returns the queue the slot goes to.
template <int MODE>
__device__ __forceinline__ int sq_trip(const SqEnv &e, const bool act, const uint32_t slot, SeedCtr &c, uint32_t &max_trips)
{
    const DIndex &ix = e.ix;
    const int NSLOT = e.NSLOT, W2 = e.W2;
    uint32_t *rd = e.rd; uint4 *st = e.st;
    int nq = SQ_FREE;
    uint4 A = make_uint4(0, 0, 0, 0), B = A, C = A;
    if (act) { A = st[slot * 3]; if (MODE != SQ_BEGIN) { B = st[slot * 3 + 1]; C = st[slot * 3 + 2]; } }
    const int r = (int)A.x, len = (int)(A.y & 0xFFFFu), end_pos = len - 13;
    int pos = (int)(A.y >> 16), nh = (int)(A.z & 0xFFFu);
    uint32_t nsearch = (A.z >> 12) & 0xFFu, trips = A.z >> 20, ns = A.w;
    Search s;
    s.mode = MODE; s.hit_len = 0; s.located = false;
    s.start = (int)(B.x & 0xFFFFu); s.p = (int)(B.x >> 16); s.ref_steps = B.y & 0xFFFFu; s.ref_blocks = B.y >> 16;
    s.x0 = s.x1 = s.lk = 0; s.x2 = 1; s.tpos = 0; s.lsteps = 0;
    if (MODE == SQ_STEP) { s.x0 = d_u64(B.z, B.w); s.x1 = d_u64(C.x, C.y); s.x2 = d_u64(C.z, C.w); }
    else if (MODE == SQ_LOC) { s.lk = d_u64(B.z, B.w); s.lsteps = C.x; }
    else if (MODE == SQ_CMP) { s.tpos = (int64_t)d_u64(B.z, B.w); s.lk = d_u64(C.x, C.y); s.lsteps = C.z; }
    auto rb = [&](int w) -> uint32_t { const int wc = w < W2 ? w : W2 - 1; const uint32_t v = rd[(size_t)wc * NSLOT + slot]; return w < W2 ? v : 0u; };
    auto rm = [&](int w) -> uint32_t { const int wc = w < W2 ? w : W2 - 1; const uint32_t v = rd[(size_t)(W2 + wc) * NSLOT + slot]; return w < W2 ? v : 0xFFFFFFFFu; };
    bool live = act, finished = false, beginning = false;
    TripData t; t.aux = T_NONE;
    TripAddr ta = {nullptr, nullptr, nullptr, nullptr};
    if (act) {
        trips = trips < 4095u ? trips + 1u : trips;
        if (MODE == SQ_BEGIN) {                        // IdentifySeedPairs :191-211: next start
            while (pos < end_pos && d_at(rm, pos)) pos++;
            if (pos >= end_pos) finished = true;
            else if (nsearch >= SEED_BAIL || trips >= (uint32_t)e.bail_trips) {     // a long walk: let a whole wave finish this read
                DHeavy hv; hv.read = (uint32_t)r; hv.pos = pos; hv.nh = nh; hv.ns = ns;
                e.heavy[atomicAdd(e.n_heavy, 1u)] = hv;
                max_trips = trips > max_trips ? trips : max_trips;
                live = false;
            } else { nsearch++; beginning = true; d_begin_issue(ix, e.K, rb, rm, pos, s, c, ta, t); }
        } else d_trip_issue(ix, rm, len, e.direct, s, c, ta, t);
    }
    d_trip_load(ta, t);
    if (live) {
        if (MODE == SQ_BEGIN) { if (beginning) d_begin_finish(ix, e.K, rb, s, c, t); }
        else if (t.aux != T_NONE) d_trip_finish(ix, e.pr, rb, rm, len, s, c, t);
        if (e.direct && s.mode == 1 && s.x2 == 1) { s.mode = 3; s.lk = s.x0; s.lsteps = 0; }   // unique: locate, then compare with the text
        if (s.mode == 1 && (s.p >= len || d_at(rm, s.p))) d_search_end(e.pr, s);              // what its next trip would find (T_STOP), without the trip
        if (!finished && s.mode == 0) {          // a search just ended (or the table said "absent")
            c.steps += s.ref_steps; c.blocks += s.ref_blocks;
            if (s.hit_len) {
                if (nh < e.H) {
                    DHit h; h.rPos = (uint16_t)s.start; h.len = (uint16_t)s.hit_len;
                    if (s.located) { h.x0 = (uint64_t)s.tpos; h.freq = 1u | 0x80000000u; c.lf_ref += s.lsteps + (uint32_t)(s.lk >> 40); }
                    else { h.x0 = s.x0; h.freq = (uint32_t)s.x2; }
                    e.hits[(size_t)r * e.H + nh] = h;
                }
                nh++; ns += (uint32_t)s.x2;
                pos = s.start + s.hit_len;
            } else pos = s.start + 1;
            while (pos < end_pos && d_at(rm, pos)) pos++;                                   // the next start, or the end of the read:
            if (pos >= end_pos) finished = true;                                            // no begin-trip just to find out
        }
        if (finished) { e.nhits[r] = (uint32_t)nh; e.nseeds[r] = ns; max_trips = trips > max_trips ? trips : max_trips; }
        else {
            nq = s.mode;
            A.y = (uint32_t)len | ((uint32_t)pos << 16); A.z = (uint32_t)nh | (nsearch << 12) | (trips << 20); A.w = ns;
            B.x = (uint32_t)s.start | ((uint32_t)s.p << 16); B.y = (s.ref_steps & 0xFFFFu) | (s.ref_blocks << 16);
            if (s.mode == 1) { B.z = (uint32_t)s.x0; B.w = (uint32_t)(s.x0 >> 32); C = make_uint4((uint32_t)s.x1, (uint32_t)(s.x1 >> 32), (uint32_t)s.x2, (uint32_t)(s.x2 >> 32)); }
            else if (s.mode == 3) { B.z = (uint32_t)s.lk; B.w = (uint32_t)(s.lk >> 32); C.x = s.lsteps; }
            else if (s.mode == 2) { B.z = (uint32_t)s.tpos; B.w = (uint32_t)((uint64_t)s.tpos >> 32); C.x = (uint32_t)s.lk; C.y = (uint32_t)(s.lk >> 32); C.z = s.lsteps; }
            st[slot * 3] = A;
            if (s.mode != 0) { st[slot * 3 + 1] = B; st[slot * 3 + 2] = C; }
        }
    }
    return nq;
}

__global__ void __launch_bounds__(SQ_THREADS)
k_seed_q(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int n_reads, int W, int H, int nslot_lg,
         DHit *__restrict__ hits, uint32_t *__restrict__ nhits, uint32_t *__restrict__ nseeds, unsigned int *next_read,
         DHeavy *__restrict__ heavy, unsigned int *n_heavy, unsigned long long *ctr, int bail_trips, int *err)
{
    extern __shared__ uint4 sq_sh[];
    const int NSLOT = 1 << nslot_lg;
    const int QCAP = 2 * NSLOT;                                  // ring entries per queue: the slots a phase puts back must not land on entries
    const uint32_t SM = (uint32_t)QCAP - 1u;                      // another wave of the same phase is still taking (a queue can hold almost all slots)
    uint4 *st = sq_sh;                                           // [slot][3]
    uint32_t *rd = (uint32_t *)(st + 3 * (size_t)NSLOT);         // [word][slot]
    uint16_t *q = (uint16_t *)(rd + (size_t)W * NSLOT);          // [queue][ring of QCAP slot numbers]
    uint16_t *tab = q + (size_t)SQ_NQ * QCAP;                   // [wave][64]: slots of the reads a refill is fetching
    uint32_t *ctl = (uint32_t *)(tab + SQ_WAVES * 64);           // head[5] (entries taken), tail[5] (entries put), [10] = the batch has no more reads
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)sq_rfl((uint32_t)(tid >> 6));
    const int W2 = W >> 1;
    const int K = ix.ktab ? ix.ktab_k : 0;
    const bool direct = ix.sa_dense != nullptr;
    const uint32_t w_magic = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;       // i / W == (i * w_magic) >> 20 for i < 64 W <= 2^12
    SeedCtr c = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t max_trips = 0, wtrips = 0;
    const SqEnv env = { ix, pr, st, rd, NSLOT, W2, K, H, bail_trips, direct, 0, hits, nhits, nseeds, heavy, n_heavy };
    uint32_t q_trips[SQ_NQ] = {0, 0, 0, 0, 0}, q_lanes[SQ_NQ] = {0, 0, 0, 0, 0};

    for (int i = tid; i < NSLOT; i += SQ_THREADS) q[(size_t)SQ_FREE * QCAP + i] = (uint16_t)i;
    if (tid < 16) ctl[tid] = tid == 5 + SQ_FREE ? (uint32_t)NSLOT : 0u;

    uint32_t n_phases = 0;
    for (uint32_t phase = 0; ; phase++) {
        n_phases = phase;
        __syncthreads();                                          // every push of the previous phase has landed
        uint32_t hd[SQ_NQ], cn[SQ_NQ];
#pragma unroll
        for (int k = 0; k < SQ_NQ; k++) { hd[k] = sq_rfl(ctl[k]); cn[k] = sq_rfl(ctl[5 + k]) - hd[k]; }
        const bool ex = sq_rfl(ctl[10]) != 0u;
        __syncthreads();                                          // every wave holds the same snapshot before anybody changes it
        if (cn[SQ_FREE] == (uint32_t)NSLOT && ex) break;          // all slots free, nothing left to claim
        if (phase >= SQ_MAX_PHASES) { if (tid == 0) atomicMax(err, DG_E_SEEDQ); break; }
        // Which 64 slots this wave takes -- every wave derives the same plan from the same snapshot.  Full chunks of 64 first:
        // a refill whenever 64 slots are free (keeps the slots busy), then the queues deepest stage first (text comparison,
        // locate, Occ step, begin), chunk number = wave number.  Only when fewer than SQ_WAVES full chunks exist are the
        // remainders (< 64 slots of a queue) handed out, in the same order.
        const uint32_t f_free = (ex || cn[SQ_FREE] < 64u) ? 0u : 1u;                      // one refill per phase
        const uint32_t r_free = (ex || cn[SQ_FREE] >= 64u) ? 0u : cn[SQ_FREE];
        const uint32_t p0 = f_free, p1 = p0 + (cn[SQ_CMP] >> 6), p2 = p1 + (cn[SQ_LOC] >> 6), p3 = p2 + (cn[SQ_STEP] >> 6), p4 = p3 + (cn[SQ_BEGIN] >> 6);
        int my_q = SQ_BEGIN;
        uint32_t my_n = 0, my_first = 0;
        const uint32_t wv = (uint32_t)wave;
        if (wv < p4) {
            my_n = 64u;
            if (wv < p0) { my_q = SQ_FREE; my_first = hd[SQ_FREE]; }
            else if (wv < p1) { my_q = SQ_CMP; my_first = hd[SQ_CMP] + ((wv - p0) << 6); }
            else if (wv < p2) { my_q = SQ_LOC; my_first = hd[SQ_LOC] + ((wv - p1) << 6); }
            else if (wv < p3) { my_q = SQ_STEP; my_first = hd[SQ_STEP] + ((wv - p2) << 6); }
            else { my_q = SQ_BEGIN; my_first = hd[SQ_BEGIN] + ((wv - p3) << 6); }
        } else {
            uint32_t j = wv - p4;                                                       // the j-th non-empty remainder
            const uint32_t r_cmp = cn[SQ_CMP] & 63u, r_loc = cn[SQ_LOC] & 63u, r_step = cn[SQ_STEP] & 63u, r_beg = cn[SQ_BEGIN] & 63u;
            bool got = false;
            if (r_free) { if (j == 0) { my_q = SQ_FREE; my_n = r_free; my_first = hd[SQ_FREE]; got = true; } else j--; }
            if (!got && r_cmp) { if (j == 0) { my_q = SQ_CMP; my_n = r_cmp; my_first = hd[SQ_CMP] + (cn[SQ_CMP] & ~63u); got = true; } else j--; }
            if (!got && r_loc) { if (j == 0) { my_q = SQ_LOC; my_n = r_loc; my_first = hd[SQ_LOC] + (cn[SQ_LOC] & ~63u); got = true; } else j--; }
            if (!got && r_step) { if (j == 0) { my_q = SQ_STEP; my_n = r_step; my_first = hd[SQ_STEP] + (cn[SQ_STEP] & ~63u); got = true; } else j--; }
            if (!got && r_beg) { if (j == 0) { my_q = SQ_BEGIN; my_n = r_beg; my_first = hd[SQ_BEGIN] + (cn[SQ_BEGIN] & ~63u); got = true; } else j--; }
        }
        if (my_n == 0) continue;                                  // (uniform per wave; the barriers are at the top)
        wtrips++;
#pragma unroll
        for (int k = 0; k < SQ_NQ; k++) if (k == my_q) { q_trips[k]++; q_lanes[k] += my_n; }
        const bool act = (uint32_t)lane < my_n;
        uint32_t slot = 0;
        if (act) slot = q[(size_t)my_q * QCAP + ((my_first + (uint32_t)lane) & SM)];
        if (lane == 0) atomicAdd(&ctl[my_q], my_n);
        int nq = SQ_FREE;                                         // the queue this lane's slot goes to

        if (my_q == SQ_FREE) {
            // ---- refill: the next my_n reads of the batch move into the free slots ----
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(next_read, my_n);
            base = sq_rfl(base);
            const uint32_t avail = base < (unsigned int)n_reads ? (unsigned int)n_reads - base : 0u;
            const uint32_t take = avail < my_n ? avail : my_n;
            if (take < my_n && lane == 0) ctl[10] = 1u;
            if (act && (uint32_t)lane < take) {
                const uint32_t r = base + (uint32_t)lane;
                tab[wave * 64 + lane] = (uint16_t)slot;
                st[slot * 3] = make_uint4(r, (uint32_t)rlen[r], 0u, 0u);       // r | len, pos = 0 | nh, searches, trips = 0 | ns = 0
                nq = SQ_BEGIN;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tab is read by the other lanes of this wave
            const uint32_t total = take * (uint32_t)W;            // the reads are consecutive: one contiguous run of enc
            const uint32_t *src = enc + (size_t)base * W;
            for (uint32_t i0 = 0; i0 < total; i0 += 256u) {
                uint32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint32_t i = i0 + (uint32_t)(k * 64 + lane); v[k] = i < total ? src[i] : 0u; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = i0 + (uint32_t)(k * 64 + lane);
                    if (i < total) { const uint32_t rk = (i * w_magic) >> 20; rd[(size_t)(i - rk * (uint32_t)W) * NSLOT + tab[wave * 64 + rk]] = v[k]; }
                }
            }
        } else if (my_q == SQ_BEGIN) nq = sq_trip<SQ_BEGIN>(env, act, slot, c, max_trips);
        else if (my_q == SQ_STEP) nq = sq_trip<SQ_STEP>(env, act, slot, c, max_trips);
        else if (my_q == SQ_CMP) nq = sq_trip<SQ_CMP>(env, act, slot, c, max_trips);
        else nq = sq_trip<SQ_LOC>(env, act, slot, c, max_trips);
        // ---- every slot of this trip goes to the queue of its new state: lane k reserves queue k's entries, one round trip for all five ----
        {
            unsigned long long m[SQ_NQ];
#pragma unroll
            for (int k = 0; k < SQ_NQ; k++) m[k] = __ballot(act && nq == k);
            uint32_t mine = 0;
            unsigned long long mq = 0;
#pragma unroll
            for (int k = 0; k < SQ_NQ; k++) { if (lane == k) mine = (uint32_t)__popcll(m[k]); if (nq == k) mq = m[k]; }
            uint32_t base = 0;
            if (lane < SQ_NQ && mine) base = atomicAdd(&ctl[5 + lane], mine);
            base = (uint32_t)__shfl((int)base, nq, 64);
            if (act) q[(size_t)nq * QCAP + ((base + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull))) & SM)] = (uint16_t)slot;
        }
    }
    atomicMax(d_ctr_stripe(ctr) + CTR_MAXTRIPS, (unsigned long long)max_trips);
    if (tid == 0) atomicAdd(d_ctr_stripe(ctr) + CTR_SQ_PHASES, (unsigned long long)n_phases);
    if (lane == 0) {
        atomicMax(d_ctr_stripe(ctr) + CTR_WTRIPS_MAX, (unsigned long long)wtrips); atomicAdd(d_ctr_stripe(ctr) + CTR_WTRIPS_SUM, (unsigned long long)wtrips);
#pragma unroll
        for (int k = 0; k < SQ_NQ; k++) if (q_trips[k]) { atomicAdd(d_ctr_stripe(ctr) + CTR_SQ_TRIPS + k, (unsigned long long)q_trips[k]); atomicAdd(d_ctr_stripe(ctr) + CTR_SQ_LANES + k, (unsigned long long)q_lanes[k]); }
    }
    d_wave_add(ctr + CTR_STEPS, c.steps);
    d_wave_add(ctr + CTR_BLOCKS, c.blocks);
    d_wave_add(ctr + CTR_STEPS_ACT, c.steps_act);
    d_wave_add(ctr + CTR_BLOCKS_ACT, c.blocks_act);
    d_wave_add(ctr + CTR_KTAB, c.ktab);
    d_wave_add(ctr + CTR_LF, c.lf_ref);
    d_wave_add(ctr + CTR_LF_ACT, c.lf_act);
    d_wave_add(ctr + CTR_DIRECT, c.n_direct);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// k_seed_qf: the same slots, queues and trip functions WITHOUT phases.  k_seed_q's waves meet at two barriers per phase and replay one
// plan from one snapshot; here every wave runs on its own: look at the queue counters (one LDS instruction: lane k reads word k), pick
// the queue to serve, reserve entries with ONE compare-and-swap on that queue's head, pop, trip, push -- no barrier after the
// initial one, no wave ever waits for another wave's trip.
//
// Queue = ring of NSLOT 16-bit entries + head (entries reserved by poppers) + tail (entries reserved by pushers), all in LDS.
//   push   a wave adds its count to `tail` (one atomic for all five queues, lane k serves queue k) and writes its entries
//   pop    a wave takes n <= min(64, tail - head) and compare-and-swaps head -> head + n; a failed swap (another wave was faster)
//          means: look again
// An entry reserved through `tail` may not be written yet when a popper reserves it through `head`, and an entry reserved through
// `head` may not be read yet when the ring comes round to it: every entry therefore carries FULL (bit 15) and the parity of its lap
// (bit 14).  The popper of lap L waits for FULL|parity(L), takes the slot, and leaves EMPTY|parity(L); the pusher of lap L waits
// for EMPTY|parity(L-1).  Both waits are almost never taken (the other side is a few instructions away) and neither can deadlock:
// each waits for a wave that is past its reservation and busy storing.  Because of this handshake the ring needs no slack: NSLOT
// entries per queue (the phased kernel needs 2 NSLOT), and a slot's state is 32 bytes (below), so a workgroup of 512 slots of
// 101-base reads takes 50 KB and three of them share a CU: 12 waves instead of 8 to cover the memory latency of a trip.
// LDS instructions of one wave execute in order, so "state stored, then entry stored" needs no wait in between -- only the
// compiler must keep the order (workgroup-scope fences on the local address space: no vmcnt wait for the hit stores in flight).
//
// Slot state, 2 x uint4 (reads up to 496 bases: 10-bit positions):
//   A.x = read   A.y = len | pos << 10 | p << 20   (pos = start of the current search, or where the next one begins)
//   A.z = hits (5) | searches (5) << 5 | trips (8, saturating) << 10 | ref_steps (10) << 18     A.w = occurrences (20) | ref_blocks (12) << 20
//   B   = step: x0, x1, x2 low words, then their bits 32..39 byte-wise;  locate: row low word, row bits 32..39 | LF steps << 8;
//         compare: text position low word, its bits 32..39 | LF steps << 8, the SA entry's memoised LF count
// ---------------------------------------------------------------------------------------------------------------------------------
#define SQF_FULL 0x8000u
#define SQF_LAP  0x4000u
#define SQF_MAX_LOOKS (1u << 24)     // looks at the counters per wave (a 2 M-read batch needs ~10^3): the safety net, as SQ_MAX_PHASES

__host__ __device__ inline size_t sqf_lds_bytes(int nslot_lg, int W, int n_waves)
{
    const size_t n = (size_t)1 << nslot_lg;
    return n * 32 + n * 4 * (size_t)W + n * 2 * SQ_NQ + (size_t)n_waves * 64 * 2 + 32 * 4;
}

__device__ __forceinline__ uint32_t sqf_ld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t sqf_ld16(const uint16_t *p) { return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sqf_st16(uint16_t *p, uint32_t v) { __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// what a trip adds to the wave's work counters (added to the accumulators at ONE place in the loop: accumulators that are
// updated inside the four trip instances get a different register in each and a block of moves at every loop edge)
struct SqfDelta { uint32_t steps, blocks, lf_ref, max_trips;
#ifdef DG_SQF_PROF
    unsigned long long t, prof[12];
#endif
};
#ifdef DG_SQF_PROF          // probe build only (profiles/probes/seed_phases.sh): where a wave's cycles go; the waits are forced, so the sum is a little above the product build's
#define SQF_T(dl, i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (dl).prof[i] += n_ - (dl).t; (dl).t = n_; } while (0)
#define SQF_TW(dl, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); SQF_T(dl, i); } while (0)
#else
#define SQF_T(dl, i)
#define SQF_TW(dl, i)
#endif

// sq_trip with the 32-byte state.  A trip whose search reaches the text comparison (a located prefix-table entry in a begin trip,
// the SA entry in a locate trip) makes its first comparison at once, in the same trip: a second dependent load for those lanes,
// but no push / pop / state round trip in between (a fifth of all slot-trips were first comparisons).
#define SQF_MULTI_MAX 4
template <int MODE, bool MULTI = false>
__device__ __forceinline__ int sqf_trip(const SqEnv &e, const bool act, const uint32_t slot, SqfDelta &dl)
{
    const DIndex &ix = e.ix;
    const int NSLOT = e.NSLOT, W2 = e.W2;
    uint32_t *rd = e.rd; uint4 *st = e.st;
    int nq = SQ_FREE;
    SeedCtr c = {0, 0, 0, 0, 0, 0, 0, 0};                // (only steps, blocks and lf_ref leave this function; the rest is dead code here)
    uint4 A = make_uint4(0, 0, 0, 0), B = A;
    if (act) { A = st[slot * 2]; if (MODE != SQ_BEGIN) B = st[slot * 2 + 1]; }
    const int r = (int)A.x, len = (int)(A.y & 0x3FFu), end_pos = len - 13;
    int pos = (int)((A.y >> 10) & 0x3FFu), nh = (int)(A.z & 31u);
    uint32_t nsearch = (A.z >> 5) & 31u, trips = (A.z >> 10) & 0xFFu, ns = A.w & 0xFFFFFu;
    Search s;
    s.mode = MODE; s.hit_len = 0; s.located = false;
    s.start = pos; s.p = (int)(A.y >> 20); s.ref_steps = A.z >> 18; s.ref_blocks = A.w >> 20;
    s.x0 = s.x1 = s.lk = 0; s.x2 = 1; s.tpos = 0; s.lsteps = 0;
    if (MODE == SQ_STEP) { s.x0 = d_u64(B.x, B.w & 0xFFu); s.x1 = d_u64(B.y, (B.w >> 8) & 0xFFu); s.x2 = d_u64(B.z, (B.w >> 16) & 0xFFu); }
    else if (MODE == SQ_LOC) { s.lk = d_u64(B.x, B.y & 0xFFu); s.lsteps = B.y >> 8; }
    else if (MODE == SQ_CMP) { s.tpos = (int64_t)d_u64(B.x, B.y & 0xFFu); s.lsteps = B.y >> 8; s.lk = (uint64_t)B.z << 40; }
    uint32_t nx = MODE == SQ_LOC ? B.z : 1u;              // rows of the interval a locate trip holds (1 unless MULTI)
    auto rb = [&](int w) -> uint32_t { const int wc = w < W2 ? w : W2 - 1; const uint32_t v = rd[(size_t)wc * NSLOT + slot]; return w < W2 ? v : 0u; };
    auto rm = [&](int w) -> uint32_t { const int wc = w < W2 ? w : W2 - 1; const uint32_t v = rd[(size_t)(W2 + wc) * NSLOT + slot]; return w < W2 ? v : 0xFFFFFFFFu; };
    bool live = act, finished = false, beginning = false;
    TripData t; t.aux = T_NONE;
    TripAddr ta = {nullptr, nullptr, nullptr, nullptr};
    uint32_t mt = 0;
    if (act) {
        trips = trips < 255u ? trips + 1u : trips;
        if (MODE == SQ_BEGIN) {                        // IdentifySeedPairs :191-211: next start
            while (pos < end_pos && d_at(rm, pos)) pos++;
            if (pos >= end_pos) finished = true;
            else if (nsearch >= SEED_BAIL || trips >= (uint32_t)e.bail_trips) {     // a long walk: let a whole wave finish this read
                DHeavy hv; hv.read = (uint32_t)r; hv.pos = pos; hv.nh = nh; hv.ns = ns;
                e.heavy[atomicAdd(e.n_heavy, 1u)] = hv;
                mt = trips;
                live = false;
            } else { nsearch++; beginning = true; d_begin_issue(ix, e.K, rb, rm, pos, s, c, ta, t); }
        } else if (!(MODE == SQ_LOC && MULTI)) d_trip_issue<MODE == SQ_STEP ? TM_STEP : (MODE == SQ_CMP ? TM_CMP : TM_LOC)>(ix, rm, len, e.direct, s, c, ta, t);
    }
    SQF_TW(dl, 2);
    if (!(MODE == SQ_LOC && MULTI)) d_trip_load(ta, t);
    SQF_TW(dl, 3);
    // what a search does between two memory accesses: a one-row interval goes on to be located and compared with the text; a step whose next
    // base is an N or past the read's end is the end of the search (what its next trip would find, without the trip)
    auto between = [&]() {
        if (e.direct && s.mode == 1 && s.x2 == 1) { s.mode = 3; s.lk = s.x0; s.lsteps = 0; nx = 1u; }
        if (s.mode == 1 && (s.p >= len || d_at(rm, s.p))) d_search_end(e.pr, s);
        // a few rows (full suffix array): their text positions are one load away and comparing the texts with the read finishes the search --
        // no more Occ steps for this interval (a 16-mer of a human-sized text has 2-3 occurrences by chance alone: these steps were 45 % of all trips)
        if (s.mode == 1 && s.x2 <= (uint64_t)e.multi) { s.mode = 3; s.lk = s.x0; s.lsteps = 0; nx = (uint32_t)s.x2; }
    };
    auto multi = [&](const bool go) {
        // ---- locate the nx <= 4 rows of the interval and compare all their texts with the read, in this trip ----
        // (a lane in mode 2 comes from a prefix-table entry that already holds its one text position: no suffix-array load for it)
        // The interval's rows are suffixes in order, so the rows that match longest are neighbours: the search ends (bwt_search.cpp:152-170 run to
        // its empty interval) with x0 = the first of them, x2 = how many, len = that match length.  Steps are counted as the reference's loop makes
        // them (one per base + the failing one); Occ blocks as one per step, as in the single-row comparison (d_trip_finish).
        uint2 sa[SQF_MULTI_MAX];
#pragma unroll
        for (int i = 0; i < SQF_MULTI_MAX; i++) sa[i] = make_uint2(0u, 0u);
        const bool have_pos = s.mode == 2;
        if (have_pos) { nx = 1u; sa[0] = make_uint2((uint32_t)s.lk, (uint32_t)(s.lk >> 32)); }
        const uint2 *sap = (const uint2 *)ix.sa_dense + (have_pos ? 0ull : s.lk);
#pragma unroll
        for (int i = 0; i < SQF_MULTI_MAX; i++) if (go && !have_pos && (uint32_t)i < nx) sa[i] = sap[i];
        SQF_TW(dl, 3);
        const int64_t L = ix.l_pac;
        const uint32_t *pw = (const uint32_t *)ix.pac;
        uint4 s16[SQF_MULTI_MAX]; uint2 s8[SQF_MULTI_MAX]; uint32_t sh[SQF_MULTI_MAX]; int kind[SQF_MULTI_MAX];      // kind: 0 none, 1 forward, 2 reverse, 3 slow
#pragma unroll
        for (int i = 0; i < SQF_MULTI_MAX; i++) {
            s16[i] = make_uint4(0u, 0u, 0u, 0u); s8[i] = make_uint2(0u, 0u); sh[i] = 0u; kind[i] = 0;
            if (go && (uint32_t)i < nx) {
                const int64_t tt = (int64_t)(d_u64(sa[i].x, sa[i].y & 0xFFu) - 1ull) + (s.p - s.start);
                int64_t f0 = -1;
                if (tt >= 0 && tt + 64 <= L) { f0 = tt; kind[i] = 1; }
                else if (tt >= L && tt + 64 <= 2 * L) { f0 = 2 * L - 1 - tt - 63; kind[i] = 2; }
                else kind[i] = 3;
                if (f0 >= 0) { s16[i] = *(const uint4_a4 *)(pw + (f0 >> 4)); s8[i] = *(const uint2_a4 *)(pw + (f0 >> 4) + 4); sh[i] = (uint32_t)((f0 & 15) << 1); }
            }
        }
        SQF_TW(dl, 5);
        if (go) {
            const int w = s.p >> 4;
            const uint32_t o = (uint32_t)((s.p & 15) << 1);
            const uint32_t b0 = rb(w), b1 = rb(w + 1), b2 = rb(w + 2), b3 = rb(w + 3), b4 = rb(w + 4);
            const uint32_t m0 = rm(w), m1 = rm(w + 1), m2 = rm(w + 2), m3 = rm(w + 3), m4 = rm(w + 4);
            const uint32_t R0 = __funnelshift_l(b1, b0, o), R1 = __funnelshift_l(b2, b1, o), R2 = __funnelshift_l(b3, b2, o), R3 = __funnelshift_l(b4, b3, o);
            const uint32_t N0 = __funnelshift_l(m1, m0, o), N1 = __funnelshift_l(m2, m1, o), N2 = __funnelshift_l(m3, m2, o), N3 = __funnelshift_l(m4, m3, o);
            bool slow = false;
#pragma unroll
            for (int i = 0; i < SQF_MULTI_MAX; i++) slow = slow || kind[i] == 3;
            const int chunk = slow ? 16 : 64;                                          // a row near a strand boundary or the end of the text: everybody compares 16 symbols
            const int in_read = len - s.p < chunk ? len - s.p : chunk;                 // bases left in the read
            int jj[SQF_MULTI_MAX];
#pragma unroll
            for (int i = 0; i < SQF_MULTI_MAX; i++) {
                jj[i] = -1;
                if ((uint32_t)i < nx) {
                    int nv = 64;
                    uint32_t T0, T1 = 0, T2 = 0, T3 = 0;
                    if (kind[i] != 3) {
                        const uint32_t d0 = __builtin_bswap32(s16[i].x), d1 = __builtin_bswap32(s16[i].y), d2 = __builtin_bswap32(s16[i].z),
                                       d3 = __builtin_bswap32(s16[i].w), d4 = __builtin_bswap32(s8[i].x);
                        const uint32_t t0 = __funnelshift_l(d1, d0, sh[i]), t1 = __funnelshift_l(d2, d1, sh[i]), t2 = __funnelshift_l(d3, d2, sh[i]), t3 = __funnelshift_l(d4, d3, sh[i]);
                        if (kind[i] == 2) { T0 = ~d_rev2(t3); T1 = ~d_rev2(t2); T2 = ~d_rev2(t1); T3 = ~d_rev2(t0); }   // T[t+j] = 3 - fwd[2L-1-t-j]
                        else { T0 = t0; T1 = t1; T2 = t2; T3 = t3; }
                    } else T0 = d_text16_slow(ix, (int64_t)(d_u64(sa[i].x, sa[i].y & 0xFFu) - 1ull) + (s.p - s.start), nv);
                    const int lim = in_read < nv ? in_read : nv;
                    auto tail = [&](int k) -> uint32_t { const int rem = lim - 16 * k; return rem >= 16 ? 0u : (rem <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu >> (2 * rem)); };
                    const uint32_t e0 = (R0 ^ T0) | N0 | tail(0), e1 = (R1 ^ T1) | N1 | tail(1), e2 = (R2 ^ T2) | N2 | tail(2), e3 = (R3 ^ T3) | N3 | tail(3);
                    jj[i] = e0 ? __clz((int)e0) >> 1 : e1 ? 16 + (__clz((int)e1) >> 1) : e2 ? 32 + (__clz((int)e2) >> 1) : e3 ? 48 + (__clz((int)e3) >> 1) : 64;
                }
            }
            int j = -1, first = 0, cnt = 0;
#pragma unroll
            for (int i = 0; i < SQF_MULTI_MAX; i++) if (jj[i] > j) j = jj[i];
#pragma unroll
            for (int i = SQF_MULTI_MAX - 1; i >= 0; i--) if (jj[i] == j) { first = i; cnt++; }
            uint2 se = sa[0];
#pragma unroll
            for (int i = 1; i < SQF_MULTI_MAX; i++) if (first == i) se = sa[i];
            s.ref_steps += (uint32_t)j; s.ref_blocks += (uint32_t)j;
            s.p += j;
            c.n_direct++;
            if (cnt == 1) {                                     // one row left: what a locate trip of that row would have found (T_SA)
                const uint64_t en = d_u64(se.x, se.y);
                s.tpos = (int64_t)((en & 0xFFFFFFFFFFull) - 1ull); s.lk = en; s.lsteps = 0; s.x2 = 1; s.mode = 2;
            } else { s.x0 = s.lk + (uint64_t)first; s.x2 = (uint64_t)cnt; s.lk = s.x0; nx = (uint32_t)cnt; s.mode = j < chunk ? 1 : 3; }
            if (j < chunk) {
                // a mismatch or the end of the text costs the reference one more (failing) step; N / end of read do not
                const uint32_t nsel = j < 16 ? N0 : j < 32 ? N1 : j < 48 ? N2 : N3;
                if (j < in_read && !((nsel >> (30 - ((j & 15) << 1))) & 1u)) { s.ref_steps++; s.ref_blocks++; }
                d_search_end(e.pr, s);
            }
        }
    };
    if (MODE == SQ_LOC && MULTI) multi(live);
    else if (live) {
        if (MODE == SQ_BEGIN) { if (beginning) d_begin_finish(ix, e.K, rb, s, c, t); }
        else if (t.aux != T_NONE) d_trip_finish<MODE == SQ_STEP ? TM_STEP : (MODE == SQ_CMP ? TM_CMP : TM_LOC)>(ix, e.pr, rb, rm, len, s, c, t);
        between();
    }
    SQF_TW(dl, 4);
    // A second access in the same trip where the first one has just produced its address: the first comparison of a search that has found its
    // text position (begin: a located table entry; locate: the SA entry).  One more dependent load for those lanes, but no push / pop / state
    // round trip in between: a fifth of all slot-trips were first comparisons.  (Chaining Occ steps the same way -- the first step behind a table
    // entry, two steps per step trip -- was measured too: 6 % fewer instructions, the same time with twelve batches in flight, 10 % longer alone.)
    if (MODE == SQ_BEGIN && MULTI) multi(live && (s.mode == 2 || s.mode == 3));      // the table entry's position, or its few rows: located and compared at once
    else if (MODE == SQ_BEGIN || MODE == SQ_LOC) {
        constexpr int M2 = TM_CMP;
        const bool go = live && ((M2 & TM_STEP) && s.mode == 1 || (M2 & TM_CMP) && s.mode == 2);
        TripData t2; t2.aux = T_NONE;
        TripAddr ta2 = {nullptr, nullptr, nullptr, nullptr};
        if (go) d_trip_issue<M2>(ix, rm, len, e.direct, s, c, ta2, t2);
        d_trip_load(ta2, t2);
        SQF_TW(dl, 5);
        if (go && t2.aux != T_NONE) { d_trip_finish<M2>(ix, e.pr, rb, rm, len, s, c, t2); between(); }
    }
    if (live) {
        if (!finished && s.mode == 0) {          // a search just ended (or the table said "absent")
            if (s.hit_len) {
                if (nh < e.H) {
                    DHit h; h.rPos = (uint16_t)s.start; h.len = (uint16_t)s.hit_len;
                    if (s.located) { h.x0 = (uint64_t)s.tpos; h.freq = 1u | 0x80000000u; c.lf_ref += s.lsteps + (uint32_t)(s.lk >> 40); }
                    else { h.x0 = s.x0; h.freq = (uint32_t)s.x2; }
                    e.hits[(size_t)r * e.H + nh] = h;
                }
                nh++; ns += (uint32_t)s.x2;
                pos = s.start + s.hit_len;
            } else pos = s.start + 1;
            dl.steps += s.ref_steps; dl.blocks += s.ref_blocks;
            while (pos < end_pos && d_at(rm, pos)) pos++;                                   // the next start, or the end of the read:
            if (pos >= end_pos) finished = true;                                            // no begin-trip just to find out
            s.ref_steps = s.ref_blocks = 0;
        }
        if (finished) { e.nhits[r] = (uint32_t)nh; e.nseeds[r] = ns; mt = trips; }
        else {
            nq = s.mode;
            A.y = (uint32_t)len | ((uint32_t)pos << 10) | ((uint32_t)(s.mode ? s.p : 0) << 20);
            A.z = (uint32_t)nh | (nsearch << 5) | (trips << 10) | (s.ref_steps << 18); A.w = ns | (s.ref_blocks << 20);
            st[slot * 2] = A;
            if (s.mode == 1) st[slot * 2 + 1] = make_uint4((uint32_t)s.x0, (uint32_t)s.x1, (uint32_t)s.x2, (uint32_t)(s.x0 >> 32) | ((uint32_t)(s.x1 >> 32) << 8) | ((uint32_t)(s.x2 >> 32) << 16));
            else if (s.mode == 3) st[slot * 2 + 1] = make_uint4((uint32_t)s.lk, (uint32_t)((s.lk >> 32) & 0xFFu) | (s.lsteps << 8), nx, 0u);
            else if (s.mode == 2) st[slot * 2 + 1] = make_uint4((uint32_t)s.tpos, (uint32_t)(((uint64_t)s.tpos >> 32) & 0xFFu) | (s.lsteps << 8), (uint32_t)(s.lk >> 40), 0u);
        }
    }
    dl.lf_ref += (uint32_t)c.lf_ref; dl.max_trips = mt > dl.max_trips ? mt : dl.max_trips;
    SQF_TW(dl, 6);
    return nq;
}

__global__ void __launch_bounds__(512)
k_seed_qf(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int n_reads, int W, int H, int nslot_lg,
          DHit *__restrict__ hits, uint32_t *__restrict__ nhits, uint32_t *__restrict__ nseeds, unsigned int *next_read,
          DHeavy *__restrict__ heavy, unsigned int *n_heavy, unsigned long long *ctr, int bail_trips, int partial_min, int multi, int *err, int drain_bail)
{
    const unsigned long long t_wave0 = wall_clock64();
    extern __shared__ uint4 sq_sh[];
    const int NSLOT = 1 << nslot_lg;
    const int QCAP = NSLOT, QLG = nslot_lg;
    const uint32_t SM = (uint32_t)QCAP - 1u;
    const int n_waves = (int)(blockDim.x >> 6);
    uint4 *st = sq_sh;                                           // [slot][2]
    uint32_t *rd = (uint32_t *)(st + 2 * (size_t)NSLOT);         // [word][slot]
    uint16_t *q = (uint16_t *)(rd + (size_t)W * NSLOT);          // [queue][ring of QCAP entries]
    uint16_t *tab = q + (size_t)SQ_NQ * QCAP;                   // [wave][64]: slots of the reads a refill is fetching
    uint32_t *ctl = (uint32_t *)(tab + (size_t)n_waves * 64);    // head[0..4], [5] = the batch has no more reads, tail[8..12]; statistics [16..31]: trips [16+q], slots [24+q], [21] idle looks
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)sq_rfl((uint32_t)(tid >> 6));
    const int W2 = W >> 1;
    const int K = ix.ktab ? ix.ktab_k : 0;
    const bool direct = ix.sa_dense != nullptr;
    const uint32_t w_magic = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;       // i / W == (i * w_magic) >> 20 for i < 64 W <= 2^12
    const SqEnv env = { ix, pr, st, rd, NSLOT, W2, K, H, bail_trips, direct, (direct && ix.sa_dense_intv == 1) ? multi : 0, hits, nhits, nseeds, heavy, n_heavy };
    // once the batch has no more reads to hand out, a workgroup only drains its slots: the launch then lasts as long as the longest chain of trips still
    // ahead of one read, so reads give up for k_seed_heavy (a wave each) after fewer trips (DG_SEED_DRAIN_BAIL; = bail_trips: no difference)
    SqEnv env_drain = env; env_drain.bail_trips = drain_bail;
    unsigned long long acc_steps = 0, acc_blocks = 0, acc_lf = 0;
    uint32_t max_trips = 0, wtrips = 0;

    // every ring starts EMPTY with the parity of "lap -1"; the free queue's first lap holds all slots
    for (int i = tid; i < SQ_NQ * QCAP; i += (int)blockDim.x) q[i] = (uint16_t)(i >= SQ_FREE * QCAP ? (SQF_FULL | (uint32_t)(i - SQ_FREE * QCAP)) : SQF_LAP);
    if (tid < 32) ctl[tid] = tid == 8 + SQ_FREE ? (uint32_t)NSLOT : 0u;
    __syncthreads();                                              // the only barrier of the kernel

    uint32_t looks = 0, lazy = 0;
    bool running = true;
#ifdef DG_SQF_PROF
    SqfDelta pf; for (int i = 0; i < 12; i++) pf.prof[i] = 0; pf.t = __builtin_amdgcn_s_memtime();
    const unsigned long long pf_t0 = pf.t;
#endif
    while (running) {
        const uint32_t cw = sqf_ld32(&ctl[lane & 15]);            // all sixteen control words in one LDS instruction: a snapshot
        uint32_t hd[SQ_NQ], cn[SQ_NQ];
#pragma unroll
        for (int k = 0; k < SQ_NQ; k++) {
            hd[k] = (uint32_t)__builtin_amdgcn_readlane((int)cw, k);
            cn[k] = (uint32_t)__builtin_amdgcn_readlane((int)cw, 8 + k) - hd[k];
            if (cn[k] > (uint32_t)NSLOT) cn[k] = 0u;              // (cannot happen while one LDS instruction is one snapshot; the swap below would catch it anyway)
        }
        const bool ex = __builtin_amdgcn_readlane((int)cw, 5) != 0;
        // which queue: a refill whenever 64 slots are free (keeps the slots busy), then the deepest stage that has a full chunk (text
        // comparison, locate, Occ step, begin); without a full chunk anywhere the fullest queue -- but a small remainder only after
        // a short nap or two (the trips in flight are about to deliver their slots)
        int my_q = -1;
        uint32_t my_n = 0;
        if (!ex && cn[SQ_FREE] >= 64u) { my_q = SQ_FREE; my_n = 64u; }
        else if (cn[SQ_CMP] >= 64u) { my_q = SQ_CMP; my_n = 64u; }
        else if (cn[SQ_LOC] >= 64u) { my_q = SQ_LOC; my_n = 64u; }
        else if (cn[SQ_STEP] >= 64u) { my_q = SQ_STEP; my_n = 64u; }
        else if (cn[SQ_BEGIN] >= 64u) { my_q = SQ_BEGIN; my_n = 64u; }
        else {
            uint32_t best = ex ? 0u : cn[SQ_FREE];
            my_q = best ? SQ_FREE : -1;
            if (cn[SQ_CMP] > best) { best = cn[SQ_CMP]; my_q = SQ_CMP; }
            if (cn[SQ_LOC] > best) { best = cn[SQ_LOC]; my_q = SQ_LOC; }
            if (cn[SQ_STEP] > best) { best = cn[SQ_STEP]; my_q = SQ_STEP; }
            if (cn[SQ_BEGIN] > best) { best = cn[SQ_BEGIN]; my_q = SQ_BEGIN; }
            my_n = best;
            if (best == 0u || (best < (uint32_t)partial_min && lazy < 2u)) { my_q = -1; if (best) lazy++; }   // nothing, or little and worth a moment's wait
        }
        uint32_t got = 0;
        if (my_q >= 0 && lane == 0) got = atomicCAS(&ctl[my_q], hd[my_q], hd[my_q] + my_n) == hd[my_q] ? 1u : 0u;   // a failed swap: another wave took from this queue meanwhile, look again
        const bool go = my_q >= 0 && sq_rfl(got) != 0u;
        SQF_TW(pf, 0);
        if (go) {
            lazy = 0;
            const uint32_t my_first = hd[my_q];
            wtrips++;
            if (lane == 0) { atomicAdd(&ctl[16 + my_q], 1u); atomicAdd(&ctl[24 + my_q], my_n); }
            const bool act = (uint32_t)lane < my_n;
            uint32_t slot = 0;
            if (act) {
                const uint32_t a = my_first + (uint32_t)lane, want = SQF_FULL | (((a >> QLG) & 1u) ? SQF_LAP : 0u);
                uint16_t *ep = q + (size_t)my_q * QCAP + (a & SM);
                uint32_t e = sqf_ld16(ep);
                while ((e & (SQF_FULL | SQF_LAP)) != want) { __builtin_amdgcn_s_sleep(1); e = sqf_ld16(ep); }
                slot = e & 0x3FFFu;
                sqf_st16(ep, want & SQF_LAP);                     // EMPTY, this lap
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");      // the slot's state and words were stored before its entry
            int nq = SQ_FREE;                                     // the queue this lane's slot goes to
            SqfDelta dl = {0u, 0u, 0u, 0u};
#ifdef DG_SQF_PROF
            SQF_TW(pf, 1);
            for (int i = 0; i < 12; i++) dl.prof[i] = 0;
            dl.t = pf.t;
#endif

            if (my_q == SQ_FREE) {
                // ---- refill: the next my_n reads of the batch move into the free slots ----
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(next_read, my_n);
                base = sq_rfl(base);
                SQF_TW(dl, 10);
                const uint32_t avail = base < (unsigned int)n_reads ? (unsigned int)n_reads - base : 0u;
                const uint32_t take = avail < my_n ? avail : my_n;
                if (take < my_n && lane == 0) __hip_atomic_store(&ctl[5], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // One memory latency for the whole refill: the lengths and up to 16 x 64 words of the reads (all of them up to 128 bases) are
                // requested before anything is stored.  (Four words per lane and pass cost a refill trip four latencies in a row:
                // 14 % of a wave's cycles, profiles/r03/seed_phases.txt.)
                const bool mine_r = act && (uint32_t)lane < take;
                const uint32_t total = take * (uint32_t)W;            // the reads are consecutive: one contiguous run of enc
                const uint32_t *src = enc + (size_t)base * W;
                uint32_t my_len = 0;
                if (mine_r) { my_len = (uint32_t)rlen[base + (uint32_t)lane]; tab[wave * 64 + lane] = (uint16_t)slot; nq = SQ_BEGIN; }
                for (uint32_t i0 = 0; i0 < total; i0 += 1024u) {
                    uint32_t v[16];
#ifdef DG_SQF_PROF
                    { const uint32_t i = i0 + (uint32_t)lane; v[0] = i < total ? src[i] : 0u; }
                    SQF_TW(dl, 11);
#pragma unroll
                    for (int k = 1; k < 16; k++) { const uint32_t i = i0 + (uint32_t)(k * 64 + lane); v[k] = i < total ? src[i] : 0u; }
                    SQF_TW(dl, 9);
#else
#pragma unroll
                    for (int k = 0; k < 16; k++) { const uint32_t i = i0 + (uint32_t)(k * 64 + lane); v[k] = i < total ? src[i] : 0u; }
#endif
                    if (i0 == 0u) {
                        if (mine_r) st[slot * 2] = make_uint4(base + (uint32_t)lane, my_len, 0u, 0u);   // r | len, pos = 0, p = 0 | nothing yet | no occurrences
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tab is read by the other lanes of this wave
                    }
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const uint32_t i = i0 + (uint32_t)(k * 64 + lane);
                        if (i < total) { const uint32_t rk = (i * w_magic) >> 20; rd[(size_t)(i - rk * (uint32_t)W) * NSLOT + tab[wave * 64 + rk]] = v[k]; }
                    }
                }
            } else if (my_q == SQ_BEGIN) { const SqEnv &eb = ex ? env_drain : env; if (env.multi) nq = sqf_trip<SQ_BEGIN, true>(eb, act, slot, dl); else nq = sqf_trip<SQ_BEGIN>(eb, act, slot, dl); }
            else if (my_q == SQ_STEP) nq = sqf_trip<SQ_STEP>(env, act, slot, dl);
            else if (my_q == SQ_CMP) nq = sqf_trip<SQ_CMP>(env, act, slot, dl);
            else if (env.multi) nq = sqf_trip<SQ_LOC, true>(env, act, slot, dl);
            else nq = sqf_trip<SQ_LOC>(env, act, slot, dl);
#ifdef DG_SQF_PROF
            if (my_q == SQ_FREE) SQF_TW(dl, 9);
            for (int i = 0; i < 12; i++) pf.prof[i] += dl.prof[i];
            pf.t = dl.t;
#endif
            acc_steps += dl.steps; acc_blocks += dl.blocks; acc_lf += dl.lf_ref; max_trips = dl.max_trips > max_trips ? dl.max_trips : max_trips;
            // ---- every slot of this trip goes to the queue of its new state: lane k reserves queue k's entries, one round trip for all five ----
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // state and words first, entries after
            unsigned long long m[SQ_NQ];
#pragma unroll
            for (int k = 0; k < SQ_NQ; k++) m[k] = __ballot(act && nq == k);
            uint32_t mine = 0;
            unsigned long long mq = 0;
#pragma unroll
            for (int k = 0; k < SQ_NQ; k++) { if (lane == k) mine = (uint32_t)__popcll(m[k]); if (nq == k) mq = m[k]; }
            uint32_t base = 0;
            if (lane < SQ_NQ && mine) base = atomicAdd(&ctl[8 + lane], mine);
            base = (uint32_t)__shfl((int)base, nq, 64);
            if (act) {
                const uint32_t a = base + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull));
                const uint32_t lap = ((a >> QLG) & 1u) ? SQF_LAP : 0u;
                uint16_t *ep = q + (size_t)nq * QCAP + (a & SM);
                while (sqf_ld16(ep) != (lap ^ SQF_LAP)) __builtin_amdgcn_s_sleep(1);      // EMPTY of the previous lap (almost never waits)
                sqf_st16(ep, SQF_FULL | lap | slot);
            }
            SQF_TW(pf, 7);
        } else if (my_q < 0) {
            if (ex && cn[SQ_FREE] == (uint32_t)NSLOT) running = false;          // all slots free, nothing left to claim
            else {
                if (lane == 0) atomicAdd(&ctl[21], 1u);
                __builtin_amdgcn_s_sleep(32);                                   // ~1 us: a trip in flight lasts several
            }
            SQF_T(pf, 8);
        }
        if (++looks > SQF_MAX_LOOKS) { if (lane == 0) atomicMax(err, DG_E_SEEDQ); running = false; }
    }
#ifdef DG_SQF_PROF
    if (lane == 0 && (blockIdx.x % 97u) == 0u && (wave & 3) == 0)
        printf("sqf wg %u wave %d: total %llu trips %u looks %u | look %llu pop %llu issue %llu mem1 %llu fin1 %llu mem2 %llu fin2+store %llu push %llu idle %llu refill %llu refill_atomic %llu refill_loads %llu\n", blockIdx.x, wave,
               __builtin_amdgcn_s_memtime() - pf_t0, wtrips, looks, pf.prof[0], pf.prof[1], pf.prof[2], pf.prof[3], pf.prof[4], pf.prof[5], pf.prof[6], pf.prof[7], pf.prof[8], pf.prof[9], pf.prof[10], pf.prof[11]);
#endif
    atomicMax(d_ctr_stripe(ctr) + CTR_MAXTRIPS, (unsigned long long)max_trips);
    if (lane == 0) { atomicMax(d_ctr_stripe(ctr) + CTR_WTRIPS_MAX, (unsigned long long)wtrips); atomicAdd(d_ctr_stripe(ctr) + CTR_WTRIPS_SUM, (unsigned long long)wtrips); }
    d_wave_add(ctr + CTR_STEPS, acc_steps);
    d_wave_add(ctr + CTR_BLOCKS, acc_blocks);
    d_wave_add(ctr + CTR_LF, acc_lf);
    d_wave_resident(ctr, CTR_WT_SEEDQF, t_wave0);
    __syncthreads();                                              // every wave has left the loop: the workgroup's statistics are final
    if (tid < SQ_NQ) {
        atomicAdd(d_ctr_stripe(ctr) + CTR_SQ_TRIPS + tid, (unsigned long long)ctl[16 + tid]); atomicAdd(d_ctr_stripe(ctr) + CTR_SQ_LANES + tid, (unsigned long long)ctl[24 + tid]);
        if (tid == 0) atomicAdd(d_ctr_stripe(ctr) + CTR_SQ_PHASES, (unsigned long long)ctl[21]);          // (here: looks that found nothing to do)
    }
}
