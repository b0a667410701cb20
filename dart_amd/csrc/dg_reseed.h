// dart_amd/csrc/dg_reseed.h -- candidate clean-up (k_prep) and wave-cooperative 8-mer re-seeding
// (k_reseed).
//
// k_prep replaces RemoveTandemRepeatSeeds / RemoveTranslocatedSeeds (AlignmentCandidates.cpp:
// 817-902) and the *enumeration* half of IdentifyMissingSeeds (:685-700): every
// ReseedingWithSpecificRegion call the reference would make becomes one DJob.
// k_reseed replaces ReseedingWithSpecificRegion (:596-624) + KmerAnalysis.cpp:25-166.
//
// Why a separate kernel: a re-seeding window is up to MaxIntronSize (5e5) bases but only
// 0.03-0.8 % of reads need one.  Inside the lane-per-read report kernel one such read
// stalls its whole wave for tens of milliseconds; as a job queue the windows are spread over the
// chip, one wave per window, 64 window positions per step.
//
// Algorithm (order-equivalent to the reference's sort/join/sort, see dg_report.h d_reseed for the
// serial form): the read-gap 8-mers sit sorted in LDS behind a 64 Kbit presence filter; every lane
// extracts the 8-mer of its window position straight from the 2-bit pac (one unaligned 24-bit
// fetch; reverse-strand windows are read from the forward pac and reverse-complemented in
// registers), and hits set bit rPos of diagonal (g - rPos) in an LDS ring of diagonals.  A diagonal
// is complete once the window front has moved past it by the gap length; complete diagonals are
// summarised 64 at a time (count, first, last rPos) and folded in increasing order into the
// running (s, max_len) state of GenerateLongestSimplePairsFromFragmentPair -- bit-identical,
// including the carry-over quirk of `s` (KmerAnalysis.cpp:147-161).
// Bound: streaming pac reads (2 bit/base) + LDS; no MFMA.
#pragma once
#include "dg_common.h"
#include "dg_report.h"

#define RS_MAX_RL   263          // longest read gap handled cooperatively (span <= 255 -> 4 bitmap words)
#define RS_RING     1024         // diagonals in the ring (>= span + 63 + RS_CHUNK live at any time): 32 KB of LDS, 4 waves per CU
#define RS_WORDS    4

// One thread per read of the general path's units (slow_units[], two reads per unit when paired).  Each live candidate gets its
// private working region (bump-allocated per wave: the regions are internal, only the records' layout is deterministic), its
// seeds unpacked into it, the clean-up passes, and its re-seeding jobs queued.
__global__ void __launch_bounds__(256)
k_prep(const DParams pr, int paired, const uint32_t *__restrict__ slow_units, const DSizes *__restrict__ sizes, const uint32_t *__restrict__ seed_off,
       const SKey *__restrict__ seeds, DCand *__restrict__ cands, const uint32_t *__restrict__ ncand, DSeed *__restrict__ work, unsigned int *worktop, uint32_t workcap,
       DJob *__restrict__ jobs, unsigned int *jobtop, uint32_t jobcap, int *err, const uint16_t *__restrict__ rlen, uint8_t *__restrict__ key, unsigned int *class_hist)
{
    if (*err >= DG_ABORT) return;
    const unsigned int n_items = sizes->n_slow_units * (paired ? 2u : 1u);
    const int lane = threadIdx.x & 63;
    for (unsigned int base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {   // uniform per workgroup
        const unsigned int it = base + threadIdx.x;
        const bool on = it < n_items;
        const int r = on ? (paired ? (int)(2u * slow_units[it >> 1] + (it & 1u)) : (int)slow_units[it]) : 0;
        DCand *cd = cands + seed_off[r];
        const int nc = on ? (int)ncand[r] : 0;
        uint32_t need = 0;
        for (int i = 0; i < nc; i++) if (cd[i].Score != 0) need += d_work_need(cd[i].count);
        // the wave's regions in one atomic
        uint32_t incl = need;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        const uint32_t wave_total = __shfl(incl, 63, 64);
        uint32_t wave_base = 0;
        if (lane == 0 && wave_total) wave_base = atomicAdd(worktop, wave_total);
        wave_base = __shfl(wave_base, 0, 64);
        if ((uint64_t)wave_base + wave_total > workcap) { if (lane == 0) atomicMax(err, DG_E_WORK); continue; }   // (worktop keeps the need)
        uint32_t wo = wave_base + incl - need;
        uint32_t k_tot = 0, k_live = 0; bool k_jobs = false, k_big = false, k_nw = false;     // the read's shape, for k_report's work order (below)
        for (int i = 0; i < nc; i++) {
            DCand &c = cd[i];
            c.final_n = 0; c.n_a = 0; c.job_count = 0; c.job_first = 0;
            if (c.Score == 0) continue;
            c.work_off = wo; wo += d_work_need(c.count);
            DSeed *s = work + c.work_off;
            int n = c.count;
            for (int q = 0; q < n; q++) {
                const SKey k = seeds[c.first + q];
                DSeed x; x.gPos = sk_gpos(k); x.rPos = sk_rpos(k); x.rLen = x.gLen = sk_rlen(k); x.flags = SEED_SIMPLE;
                s[q] = x;
            }
            n = d_untangle_seeds(s, n, (int)rlen[r], (uint32_t *)(s + 6 * c.count + 4));       // tail of the working region as scratch
            c.n_a = n;
            int cnt = 0;                                     // IdentifyMissingSeeds :691-697, enumeration only
            int big = n > 0 && (s[0].rPos > PM_MAX || (int)rlen[r] - (s[n - 1].rPos + s[n - 1].rLen) > PM_MAX) ? 1 : 0, indel = 0;
            for (int k = 1; k < n; k++) {
                const int pd = (int)((s[k].gPos - s[k].rPos) - (s[k - 1].gPos - s[k - 1].rPos));
                const int rGaps = s[k].rPos - s[k - 1].rPos - s[k - 1].rLen;
                if (pd > pr.max_gaps && rGaps > 20) cnt++;
                if (rGaps > PM_MAX || rGaps + pd > PM_MAX) big = 1;
                if (pd != 0) indel = 2;
            }
            c.final_n = big | indel;                         // scheduling hints only (k_report sets the real value): bit 0 = some
                                                             // segment pair is longer than PM_MAX (string path, maybe a wave-wide alignment),
                                                             // bit 1 = two seeds on different diagonals (an nw_alignment is certain)
            k_live++; k_tot += (uint32_t)n; k_jobs = k_jobs || cnt > 0; k_big = k_big || big != 0; k_nw = k_nw || indel != 0;
            if (cnt == 0) continue;
            const unsigned int first = atomicAdd(jobtop, (unsigned int)cnt);
            if (first + (unsigned int)cnt > jobcap) { atomicMax(err, DG_E_JOBS); continue; }
            c.job_first = first; c.job_count = cnt;
            int w = 0;
            for (int k = 1; k < n; k++) {
                const int pd = (int)((s[k].gPos - s[k].rPos) - (s[k - 1].gPos - s[k - 1].rPos));
                const int rGaps = s[k].rPos - s[k - 1].rPos - s[k - 1].rLen;
                if (pd > pr.max_gaps && rGaps > 20) {
                    DJob j;
                    j.Lb = s[k - 1].gPos + s[k - 1].gLen;
                    j.glen = (int32_t)(s[k].gPos - j.Lb);
                    j.rBegin = s[k - 1].rPos + s[k - 1].rLen; j.rl = rGaps;
                    j.read = (uint32_t)r; j.found = rGaps > RS_MAX_RL ? -1 : 0;   // -1: left to the serial path in k_report
                    j.gPos = 0; j.rPos = 0; j.len = 0; j.pad = 0;
                    jobs[first + w++] = j;
                }
            }
        }
        // ---- work order of k_report: the read's cost class (reads of one class do the same things in the same order, so a wave's lanes stay
        // converged): 0 = waits for k_reseed; 1-4 = some pair is too big for the register-only path (by seed count); 5-14 = small pairs only, by
        // (live candidates, seeds); 15 = nothing to report.  Every class is split in two: first the reads with two seeds on different
        // diagonals (an nw_alignment is certain), then the rest.  Round 2 classified all 2 M reads of the batch in two more launches and a
        // three-launch scan (k_cost / k_cost_scatter); only the ~5 % on this list need an order.
        uint32_t kc = DG_COST_CLASSES;
        if (on) {
            if (k_jobs) kc = 0;
            else if (k_live == 0) kc = 15;
            else if (k_big) kc = k_tot > 12 ? 1u : k_tot > 6 ? 2u : k_tot > 3 ? 3u : 4u;
            else if (k_live >= 3) kc = k_tot > 8 ? 5u : 6u;
            else if (k_live == 2) kc = k_tot > 4 ? 7u : k_tot > 2 ? 8u : 9u;
            else kc = k_tot >= 5 ? 10u : k_tot == 4 ? 11u : k_tot == 3 ? 12u : k_tot == 2 ? 13u : 14u;
            kc = 2 * kc + ((k_nw || kc == 15) ? 0u : 1u);
            key[r] = (uint8_t)kc;
        }
        for (unsigned long long rem = __ballot(kc < DG_COST_CLASSES); rem; ) {          // only the classes that occur in the wave
            const uint32_t cls = (uint32_t)__builtin_amdgcn_readlane((int)kc, __ffsll((long long)rem) - 1);
            const unsigned long long m = __ballot(kc == cls);
            if (lane == 0) atomicAdd(class_hist + cls, (unsigned int)__popcll(m));
            rem &= ~m;
        }
    }
}

// k_report's work list from the keys and the class histogram of k_prep: perm = the listed reads grouped by class, class 0 first (any order
// inside a class: the order decides who computes what when, never a result); info[0] = reads of class 0 (they wait for k_reseed), info[1] =
// end of the heavy classes 1-3 (keys 2..7), info[2] = all listed reads.  One launch over the list.
__global__ void __launch_bounds__(256)
k_order_reads(int paired, const uint32_t *__restrict__ slow_units, const DSizes *__restrict__ sizes, const uint8_t *__restrict__ key, const unsigned int *__restrict__ class_hist,
              unsigned int *class_fill, uint32_t *__restrict__ perm, uint32_t *__restrict__ info, const int *__restrict__ abort_p)
{
    __shared__ uint32_t s_start[DG_COST_CLASSES];
    if (*abort_p >= DG_ABORT) return;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {                                        // exclusive scan of the 32 class totals by the first wave
        uint32_t v = lane < DG_COST_CLASSES ? class_hist[lane] : 0u, incl = v;
        for (int o = 1; o < DG_COST_CLASSES; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane < DG_COST_CLASSES) s_start[lane] = incl - v;
        if (blockIdx.x == 0) {
            if (lane == 1) info[0] = incl;                         // keys 0 and 1 = class 0
            if (lane == 7) info[1] = incl;                         // keys 0..7 = classes 0..3
            if (lane == DG_COST_CLASSES - 1) info[2] = incl;
        }
    }
    __syncthreads();
    const unsigned int n_items = sizes->n_slow_units * (paired ? 2u : 1u);
    for (unsigned int base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {
        const unsigned int it = base + threadIdx.x;
        const bool on = it < n_items;
        const uint32_t r = on ? (paired ? 2u * slow_units[it >> 1] + (it & 1u) : slow_units[it]) : 0u;
        const uint32_t kc = on ? key[r] : DG_COST_CLASSES;
        for (unsigned long long rem = __ballot(kc < DG_COST_CLASSES); rem; ) {
            const int leader = __ffsll((long long)rem) - 1;
            const uint32_t cls = (uint32_t)__builtin_amdgcn_readlane((int)kc, leader);
            const unsigned long long m = __ballot(kc == cls);
            uint32_t b = 0;
            if (lane == leader) b = atomicAdd(class_fill + cls, (unsigned int)__popcll(m));
            b = (uint32_t)__shfl((int)b, leader, 64);
            if (kc == cls) perm[s_start[cls] + b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
            rem &= ~m;
        }
    }
}

// 8-mer id (CreateKmerID, KmerAnalysis.cpp:25-32) of text positions t..t+7, t in [0, 2L)
__device__ __forceinline__ uint32_t d_window_kmer(const DIndex &ix, int64_t t)
{
    const int64_t L = ix.l_pac;
    if (t >= 0 && t + 7 < L) {
        const uint8_t *p = ix.pac + (t >> 2);
        const uint32_t v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[2];
        return (v >> (8 - 2 * (int)(t & 3))) & 0xFFFFu;
    }
    if (t >= L && t + 7 < 2 * L) {                      // reverse half: complement of the mirrored forward 8-mer
        const int64_t u = 2 * L - 1 - t - 7;
        const uint8_t *p = ix.pac + (u >> 2);
        const uint32_t v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[2];
        uint32_t x = (v >> (8 - 2 * (int)(u & 3))) & 0xFFFFu;
        x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
        x = ((x & 0x0F0Fu) << 4) | ((x >> 4) & 0x0F0Fu);
        x = ((x << 8) | (x >> 8)) & 0xFFFFu;
        return x ^ 0xFFFFu;
    }
    uint32_t wid = 0;                                    // straddles the strand boundary or the end
    for (int i = 0; i < 8; i++) wid = (wid << 2) + d_nt4((unsigned char)d_refchar(ix, t + i));
    return wid;
}

// running state of GenerateLongestSimplePairsFromFragmentPair's scan (KmerAnalysis.cpp:146-163)
struct RsFold { int s, max_len, best_r; int64_t best_g, next_fin; };

// LDS ring of diagonals, word-major (ring[w * RS_RING + slot]) so that 64 consecutive diagonals are
// read conflict-free; `dirty` has one bit per group of 64 diagonals that received a hit.
// Folds the complete diagonal groups below `lim` (all remaining ones when `final`) into st, in
// increasing diagonal order; every lane of the wave calls it with the same arguments.  Only the groups
// whose dirty bit is set cost anything: the mask is read once and walked in a scalar register.
template <int WORDS>
__device__ __forceinline__ void d_rs_finalize(unsigned long long *ring, uint32_t *dirty, RsFold &st, const int64_t lim, const bool final, const int lane)
{
    static_assert(RS_RING / 64 == 16, "the dirty mask is walked as a 16-bit rotation");
    const int64_t room = lim - st.next_fin;
    const int ng = final ? (room > 0 ? (int)((room + 63) >> 6) : 0) : (room >= 64 ? (int)(room >> 6) : 0);   // (at most 16: the ring never holds more)
    if (ng == 0) return;
    const uint32_t dm = __builtin_amdgcn_readfirstlane(*dirty);
    if (dm) {
        const uint32_t gi = (uint32_t)((uint64_t)st.next_fin >> 6) & 15u;
        uint32_t m = ((dm | (dm << 16)) >> gi) & (ng >= 16 ? 0xFFFFu : ((1u << ng) - 1u));        // bit k: group next_fin + 64 k has hits
        uint32_t cleared = 0;
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1;
            const int64_t b0 = st.next_fin + 64 * k;
            cleared |= 1u << ((gi + (uint32_t)k) & 15u);
            const int slot = (int)((uint64_t)(b0 + lane) & (RS_RING - 1));
            int cnt = 0, first = 0, last = 0;
            bool any = false;
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                const unsigned long long v = ring[w * RS_RING + slot];
                if (v) {
                    cnt += __popcll(v);
                    if (!any) first = w * 64 + (__ffsll((long long)v) - 1);
                    any = true;
                    last = w * 64 + 63 - __clzll((long long)v);
                    ring[w * RS_RING + slot] = 0;
                }
            }
            const uint32_t pk = (uint32_t)cnt | ((uint32_t)first << 9) | ((uint32_t)last << 17);      // cnt <= 256, first / last < 256
            unsigned long long mask = __ballot(any);
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)pk, l);
                const int c = (int)(q & 511u), f = (int)((q >> 9) & 255u), la = (int)(q >> 17);
                st.s += c - 1;
                const int len = 8 + (la - f);
                if (len > st.max_len && st.s > (len - 8) / 2) { st.best_r = f; st.best_g = b0 + l + f; st.max_len = len; st.s = 1; }
                mask &= mask - 1;
            }
        }
        if (lane == 0) *dirty = dm & ~cleared;
    }
    st.next_fin += 64 * (int64_t)ng;
}

// the block is ONE wave: LDS operations of a wave complete in order, so ordering needs only a
// compiler barrier + an LDS drain -- unlike __syncthreads() this leaves the prefetched global
// load in flight (hipcc drains vmcnt at every __syncthreads fence)
#define RS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define RS_CHUNK 512             // window positions per trip: one coalesced pac load, RS_PPL consecutive positions per lane;
                                 // live diagonals: span + 63 + RS_CHUNK <= RS_RING
#define RS_PPL (RS_CHUNK / 64)
#define RS_SUPER (8 * RS_CHUNK)  // window positions per pac fetch: one uint4 per lane covers eight chunks, so a wave waits for memory once
                                 // per 4096 positions instead of once per 512 (the fetch of chunk g+1 issued during chunk g did not
                                 // hide a miss: 4.6 us per chunk was the latency, not the work)

// Work order of the re-seeding jobs: per ring size (1, 2, 4 bitmap words per diagonal) the jobs of that size, longest window first
// (32 length classes by the position of the top bit): windows are log-uniform up to MaxIntronSize, a wave takes one window at a
// time, so with the static job -> wave assignment of round 1 the kernel's time was the unluckiest wave's (one 500 kb window = 1000
// chunks after the others had finished).  lists[w] (w = 0, 1, 2) holds the job indices, counts[w] their number.  One workgroup.
__global__ void __launch_bounds__(1024)
k_order_jobs(const DJob *__restrict__ jobs, const unsigned int *__restrict__ jobtop, uint32_t jobcap, uint32_t *__restrict__ lists, unsigned int *__restrict__ counts,
             const int *__restrict__ abort_p)
{
    __shared__ unsigned int hist[3 * 32], start[3 * 32];
    if (*abort_p >= DG_ABORT) return;
    const unsigned int njobs = *jobtop < jobcap ? *jobtop : jobcap;
    if (threadIdx.x < 96) hist[threadIdx.x] = 0;
    __syncthreads();
    auto cls = [](const DJob &j, int &w, int &b) -> bool {
        if (j.found < 0) return false;                                    // too long for the LDS ring: serial path in k_report
        const int need = j.rl >= 8 ? (j.rl - 8) / 64 + 1 : 1;
        w = need <= 1 ? 0 : (need <= 2 ? 1 : 2);
        b = 31 - (j.glen > 0 ? 31 - __clz(j.glen) : 0);                    // longest windows -> class 0
        return true;
    };
    for (unsigned int i = threadIdx.x; i < njobs; i += blockDim.x) { int w, b; if (cls(jobs[i], w, b)) atomicAdd(&hist[w * 32 + b], 1u); }
    __syncthreads();
    if (threadIdx.x < 3) { unsigned int run = 0; for (int b = 0; b < 32; b++) { start[threadIdx.x * 32 + b] = run; run += hist[threadIdx.x * 32 + b]; } counts[threadIdx.x] = run; }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < njobs; i += blockDim.x) { int w, b; if (cls(jobs[i], w, b)) lists[(size_t)w * jobcap + atomicAdd(&start[w * 32 + b], 1u)] = i; }
}

#define RS_TAB      512          // open-addressing table: 8-mer id -> first entry of km[] with that id (at most 256 ids: half full)
#define RS_FLT_BITS 14           // the bitmap filter looks at the low 14 bits of an id (2 KB); the table decides
#define RS_EMPTY    0xFFFFFFFFu
__device__ __forceinline__ uint32_t d_rs_hash(uint32_t w) { return (w ^ (w >> 7)) & (RS_TAB - 1); }

// WORDS = 64-bit bitmap words per diagonal: a job with read gap rl needs (rl - 8) / 64 + 1 of them; the kernel is
// instantiated for 1, 2 and 4 (8 / 16 / 32 KB of ring + 6.5 KB of tables: 10 / 6 / 4 waves per CU) and each instance takes the
// jobs of its size.  Keys of the read gap's 8-mers are (id << 9 | position): ids stay below 2^17 (CreateKmerID adds
// nst_nt4_table values up to 5 without masking), positions below 512.
// Per trip a wave looks at RS_CHUNK window positions, RS_PPL consecutive ones per lane: the lane's 15 bases come from two
// dwords of the staged pac bytes (reverse half: bit-reversed and complemented once, not per 8-mer), every 8-mer is one
// bit-field extract, the bitmap says which of them can be in the read gap at all (0.3 % are), and those go through the
// table -- one or two LDS reads -- to their entries of km[].
template <int WORDS>
__global__ void __launch_bounds__(64)
k_reseed(const DIndex ix, const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off,
         DJob *__restrict__ jobs, const uint32_t *__restrict__ list, const unsigned int *__restrict__ count_p, unsigned int *ticket, unsigned long long *ctr, const int *__restrict__ abort_p)
{
    __shared__ unsigned char rs[RS_MAX_RL + 9];
    __shared__ uint32_t km[RS_MAX_RL + 1];
    __shared__ uint32_t flt[(1 << RS_FLT_BITS) / 32];
    __shared__ uint32_t tab[RS_TAB];
    __shared__ unsigned long long ring[RS_RING * WORDS];
    __shared__ uint32_t pacbuf[RS_SUPER / 16 + 16];         // one super-chunk of pac: RS_SUPER / 4 bytes + alignment slack
    __shared__ uint32_t s_dirty;
    __shared__ int s_nk;
    __shared__ unsigned int s_next;
    uint32_t *tmpk = (uint32_t *)ring;                       // the unsorted keys live in the ring's space until the ring is cleared
    const int lane = threadIdx.x;
    if (*abort_p >= DG_ABORT) return;
    const unsigned int njobs = *count_p;               // this ring size's jobs, longest window first (k_order_jobs)
    const int64_t L = ix.l_pac;
    const uint64_t t_start = wall_clock64();
    unsigned long long n_done = 0, w_done = 0, n_trips = 0;
    while (true) {
        __syncthreads();
        if (lane == 0) s_next = atomicAdd(ticket, 1u);
        __syncthreads();
        if (s_next >= njobs) break;
        const unsigned int jb = list[s_next];
        const DJob job = jobs[jb];
        const int rl = job.rl, glen = job.glen;
        int thr = (int)(rl * 0.85); if (thr < 8) thr = 8;
        n_done += 1; w_done += (unsigned long long)(glen > 0 ? glen : 0);
        const unsigned char *rd = seq + seq_off[job.read] + job.rBegin;
        bool plain = true;                                   // only A, C, G, T in the gap (either case): the ids need no carries, and the reference's restart
                                                             // after an 'N' (which leaves its window one base behind its label from there on) does not happen
        for (int i = lane; i < rl + 8; i += 64) {
            const unsigned char ch = i < rl ? rd[i] : (unsigned char)'N';
            rs[i] = ch;
            if (i < rl && d_nt4(ch) > 3) plain = false;
        }
        for (int i = lane; i < (1 << RS_FLT_BITS) / 32; i += 64) flt[i] = 0;
        for (int i = lane; i < RS_TAB; i += 64) tab[i] = RS_EMPTY;
        if (lane == 0) s_dirty = 0;
        plain = __ballot(!plain) == 0;
        __syncthreads();
        // CreateKmerVecFromReadSeq :34-80 on the read gap: one key per position
        if (plain) {
            int nk = 0;
            for (int base = 0; base + 8 <= rl; base += 64) {
                const int head = base + lane;
                const bool ok = head + 8 <= rl;
                uint32_t wid = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) wid = (wid << 2) + (d_nt4(rs[head + i]) & 3u);
                const unsigned long long mk = __ballot(ok);
                if (ok) tmpk[nk + __popcll(mk & ((1ull << lane) - 1ull))] = (wid << 9) | (uint32_t)head;
                nk += __popcll(mk);
            }
            if (lane == 0) s_nk = nk;
        } else if (lane == 0) {   // the reference's rolling form, literally
            int nk = 0, count = 0, head, tail = 0;
            uint32_t wid = 0;
            while (count < 8 && tail < rl) { if (rs[tail++] != 'N') count++; else count = 0; }
            if (count == 8) {
                head = tail - 8; wid = 0;
                for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(rs[i]);
                tmpk[nk++] = (wid << 9) | (uint32_t)head;
                for (head += 1; tail < rl; head++, tail++) {
                    if (rs[tail] != 'N') {
                        wid = ((wid & 0x3FFF) << 2) + d_nt4(rs[tail]);
                        tmpk[nk++] = (wid << 9) | (uint32_t)head;
                    } else {
                        count = 0; tail++;
                        while (count < 8 && tail < rl) { if (rs[tail++] != 'N') count++; else count = 0; }
                        if (count == 8) {
                            head = tail - 8; wid = 0;
                            for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(rs[i]);
                            tmpk[nk++] = (wid << 9) | (uint32_t)head;
                        } else break;
                    }
                }
            }
            s_nk = nk;
        }
        __syncthreads();
        const int nk = s_nk;
        int found = 0, best_r = 0, max_len = 0;
        int64_t best_g = 0;
        if (nk > 0 && glen >= 8) {
            for (int e = lane; e < nk; e += 64) {           // rank sort by (id, position); keys are distinct
                const uint32_t key = tmpk[e];
                int rank = 0;
                for (int j = 0; j < nk; j++) rank += tmpk[j] < key ? 1 : 0;
                km[rank] = key;
                const uint32_t fw = (key >> 9) & ((1u << RS_FLT_BITS) - 1u);
                atomicOr(&flt[fw >> 5], 1u << (fw & 31));
            }
            __syncthreads();
            for (int i = lane; i < RS_RING * WORDS; i += 64) ring[i] = 0;       // (tmpk is dead from here)
            for (int e = lane; e < nk; e += 64) {            // the first entry of every id goes into the table
                const uint32_t id = km[e] >> 9;
                if (e == 0 || (km[e - 1] >> 9) != id) {
                    uint32_t slot = d_rs_hash(id);
                    while (atomicCAS(&tab[slot], RS_EMPTY, (id << 9) | (uint32_t)e) != RS_EMPTY) slot = (slot + 1) & (RS_TAB - 1);
                }
            }
            __syncthreads();
            const int span = rl - 8;
            RsFold st; st.s = 1; st.max_len = 0; st.best_r = 0; st.best_g = 0;
            st.next_fin = -(int64_t)(((span + 63) >> 6) << 6);          // aligned to the 64-diagonal groups
            // window entirely inside one strand half -> k-mers come from coalesced pac dwords staged in LDS
            const bool fwd = job.Lb >= 0 && job.Lb + glen <= L;
            const bool rev = job.Lb >= L && job.Lb + glen <= 2 * L;
            auto super_base = [&](int gs) -> int64_t {                  // first pac byte (16-aligned) of the super-chunk that starts at window position gs
                if (fwd) return (int64_t)(((job.Lb + gs) >> 2) & ~(int64_t)15);
                int64_t ulo = 2 * L - 1 - (job.Lb + gs + RS_SUPER - 1) - 7;
                if (ulo < 0) ulo = 0;
                return (int64_t)((ulo >> 2) & ~(int64_t)15);
            };
            uint4 pre = make_uint4(0, 0, 0, 0), pre_t = make_uint4(0, 0, 0, 0);
            if (fwd || rev) {
                const uint4 *src = (const uint4 *)(ix.pac + super_base(0));
                pre = src[lane]; if (lane < 4) pre_t = src[64 + lane];
            }
            int64_t B0 = 0;
            for (int g0 = 0; g0 + 8 <= glen; g0 += RS_CHUNK) {
                n_trips++;
                d_rs_finalize<WORDS>(ring, &s_dirty, st, (int64_t)g0 - span, false, lane);
                const int p0 = g0 + RS_PPL * lane;
                uint32_t y = 0;                 // the lane's RS_PPL + 7 window bases (and one more), first base in the top bits
                if (fwd || rev) {
                    if (g0 % RS_SUPER == 0) {                              // (uniform) a new super-chunk: the fetch issued one super-chunk ago has arrived
                        B0 = super_base(g0);
                        RS_WAVE_SYNC();
                        ((uint4 *)pacbuf)[lane] = pre; if (lane < 4) ((uint4 *)pacbuf)[64 + lane] = pre_t;
                        if (g0 + RS_SUPER + 8 <= glen) {
                            const uint4 *src = (const uint4 *)(ix.pac + super_base(g0 + RS_SUPER));
                            pre = src[lane]; if (lane < 4) pre_t = src[64 + lane];
                        }
                        RS_WAVE_SYNC();
                    }
                    // first forward base this lane needs, relative to base 4*B0 of the staged bytes
                    int fb, o_rev = 0;
                    if (fwd) fb = (int)(job.Lb + p0 - 4 * B0);
                    else { const int64_t f64 = 2 * L - 1 - (job.Lb + p0 + RS_PPL - 1) - 7 - 4 * B0; fb = (int)f64; if (f64 < 0) { o_rev = f64 < -64 ? 64 : (int)-f64; fb = 0; } }
                    const int m = fb >> 4, o = fb & 15;                                   // dword index, base offset inside it
                    const uint32_t w0 = __builtin_bswap32(pacbuf[m]), w1 = __builtin_bswap32(pacbuf[m + 1]);
                    uint32_t f = o ? ((w0 << (2 * o)) | (w1 >> (32 - 2 * o))) : w0;       // 16 forward bases from fb on
                    if (fwd) y = f;
                    else {
                        // window base k is the complement of forward base 14 - k: the 2-bit groups reversed, one group up, complemented
                        if (o_rev) f = o_rev > 15 ? 0u : f >> (2 * o_rev);   // window start clipped at forward base 0 (never a valid position)
                        uint32_t r = __builtin_bitreverse32(f);
                        r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
                        y = ~r << 2;
                    }
                } else {                                                    // straddles the strand boundary or the end of the text
                    for (int i = 0; i < RS_PPL + 7; i++)
                        if (p0 + i < glen) y |= (uint32_t)(d_nt4((unsigned char)d_refchar(ix, job.Lb + p0 + i)) & 3u) << (30 - 2 * i);
                }
                const int n_valid = glen - 7 - p0;                          // positions p0 + j with j < n_valid have their 8 bases inside the window
                uint32_t pass = 0;
#pragma unroll
                for (int j = 0; j < RS_PPL; j++) {
                    const uint32_t fw = (y >> (16 - 2 * j)) & ((1u << RS_FLT_BITS) - 1u);
                    pass |= ((flt[fw >> 5] >> (fw & 31)) & 1u) << j;
                }
                pass &= n_valid >= RS_PPL ? (1u << RS_PPL) - 1u : (n_valid > 0 ? (1u << n_valid) - 1u : 0u);
                while (pass) {
                    const int j = __ffs((int)pass) - 1;
                    pass &= pass - 1;
                    const uint32_t w = (y >> (16 - 2 * j)) & 0xFFFFu;
                    uint32_t slot = d_rs_hash(w), e = tab[slot];
                    for (int t = 0; e != RS_EMPTY && (e >> 9) != w && t < RS_TAB; t++) { slot = (slot + 1) & (RS_TAB - 1); e = tab[slot]; }
                    if (e == RS_EMPTY || (e >> 9) != w) continue;                          // only the low bits of the id were in the read gap
                    const int p = p0 + j;
                    for (int lo = (int)(e & 511u); lo < nk; lo++) {
                        const uint32_t key = km[lo];
                        if ((key >> 9) != w) break;
                        const int rp = (int)(key & 511u);
                        const int64_t d = (int64_t)p - rp;
                        atomicOr(&ring[(rp >> 6) * RS_RING + (int)((uint64_t)d & (RS_RING - 1))], 1ull << (rp & 63));
                        atomicOr(&s_dirty, 1u << ((uint32_t)((uint64_t)d >> 6) & (RS_RING / 64 - 1)));
                    }
                }
                RS_WAVE_SYNC();
            }
            d_rs_finalize<WORDS>(ring, &s_dirty, st, (int64_t)(glen - 8) + 1, true, lane);
            max_len = st.max_len; best_r = st.best_r; best_g = st.best_g;
            found = (max_len >= thr && max_len > 0) ? 1 : 0;
        }
        if (lane == 0) {
            DJob &o = jobs[jb];
            o.found = found; o.len = max_len; o.rPos = best_r + job.rBegin; o.gPos = best_g + job.Lb;
        }
    }
    if (lane == 0) {
        if (n_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED, n_done);
        if (w_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEEDW, w_done);
        if (n_trips) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_TRIPS, n_trips);
        atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_TICKS, (unsigned long long)(wall_clock64() - t_start));
    }
}
