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

// The report a candidate that is not aligned gets (GenMappingReport :1079-1207 for a candidate whose Score the candidate rules set to 0, and the single
// empty report of a read without candidates): k_prep writes these, k_report only sees candidates that are alive.
template <typename ReportT>
__device__ __forceinline__ ReportT d_blank_report(int paired_idx)
{
    ReportT rp;
    rp.aln_score = 0; rp.sj_type = -1; rp.flag = 0; rp.paired_idx = paired_idx; rp.chr = -1; rp.bdir = 0; rp.pos = 0; rp.cigar_off = 0; rp.n_cigar = 0;
    return rp;
}

// k_report's work order is over CANDIDATES (round 5; rounds 1-4: over reads, a lane walking its read's candidates one after the other -- a wave of 64 reads
// lasted as long as its read with the most candidates times the slowest candidate of each round, and on a genome with human-like repeat content, where a
// read of the general path has several live candidates, the kernel was one long tail).  Candidates are independent until the best / second-best bookkeeping
// of :1161-1172, which k_finalize replays in candidate order.  The cost class of a candidate (candidates of one class do the same things in the same order,
// so a wave's lanes stay converged): 0 = waits for k_reseed; 1-4 = some segment pair is too big for the register-only path (by seed count); 5-11 = small
// pairs only, by seed count.  Every class is split in two: first the candidates with two seeds on different diagonals (an nw_alignment is certain), then the rest.
__host__ __device__ __forceinline__ uint32_t d_cand_class(bool jobs, bool big, bool indel, int n)
{
    const uint32_t kc = jobs ? 0u : big ? (n > 12 ? 1u : n > 6 ? 2u : n > 3 ? 3u : 4u) : (n >= 8 ? 5u : n >= 6 ? 6u : n == 5 ? 7u : n == 4 ? 8u : n == 3 ? 9u : n == 2 ? 10u : 11u);
    return 2u * kc + (indel ? 0u : 1u);
}

// One thread per read of the general path's units (slow_units[], two reads per unit when paired).  Each live candidate gets its
// private working region (bump-allocated per wave: the regions are internal, only the records' layout is deterministic), its
// seeds unpacked into it, the clean-up passes, its re-seeding jobs queued, and its cost class (kept in final_n until k_report sets the real value;
// the class histogram is summed in LDS and added to class_hist once per workgroup).  Candidates that are not alive get their blank report here.
template <typename ReportT>
__global__ void __launch_bounds__(256)
k_prep(const DParams pr, int paired, const uint32_t *__restrict__ slow_units, const DSizes *__restrict__ sizes, const uint32_t *__restrict__ seed_off,
       const SKey *__restrict__ seeds, DCand *__restrict__ cands, const uint32_t *__restrict__ ncand, DSeed *__restrict__ work, unsigned int *worktop, uint32_t workcap,
       DJob *__restrict__ jobs, unsigned int *jobtop, uint32_t jobcap, int *err, const uint16_t *__restrict__ rlen, const uint32_t *__restrict__ rep_off,
       ReportT *__restrict__ reports, unsigned int *class_hist, const uint32_t *__restrict__ enc, int W2, unsigned char *__restrict__ seq)
{
    __shared__ unsigned int s_hist[DG_COST_CLASSES];
    if (*err >= DG_ABORT) return;
    if (threadIdx.x < DG_COST_CLASSES) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned int n_items = sizes->n_slow_units * (paired ? 2u : 1u);
    const int lane = threadIdx.x & 63;
    for (unsigned int base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {   // uniform per workgroup
        const unsigned int it = base + threadIdx.x;
        const bool on = it < n_items;
        const int r = on ? (paired ? (int)(2u * slow_units[it >> 1] + (it & 1u)) : (int)slow_units[it]) : 0;
        DCand *cd = cands + seed_off[r];
        const int nc = on ? (int)ncand[r] : 0;
        if (on && enc) {
            // A packed batch has no ASCII copy (k_pair reads the 2-bit + mask words); the reads of the general path get theirs here: A/C/G/T by code, 'N' where
            // the mask says so, 16 bases per store (round 4: a launch of its own, k_unpack_listed).  Four codes (one byte of w, first base on top) -> one
            // selector byte each (copies of the byte at shifts 0, 10, 20, 30 put the k-th pair at bit 8 k + 6), then "ACGT"[code] for all four with one byte
            // permute; the mask's pairs the same way -> 'N'
            const uint32_t *wsrc = enc + (size_t)r * 2 * W2;
            unsigned char *dst = seq + (size_t)r * 16 * W2;
            for (int ww = 0; ww < W2; ww++) {
                const uint32_t w = wsrc[ww], m = wsrc[W2 + ww];
                uint32_t out[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t a = (w >> (24 - 8 * q)) & 0xFFu, b = (m >> (24 - 8 * q)) & 0xFFu;
                    const uint32_t sel = ((a * 0x40100401u) >> 6) & 0x03030303u, nb = ((b * 0x40100401u) >> 6) & 0x03030303u;
                    const uint32_t isn = ((nb | (nb >> 1)) & 0x01010101u) * 0xFFu;
                    out[q] = (__builtin_amdgcn_perm(0u, 0x54474341u, sel) & ~isn) | (0x4E4E4E4Eu & isn);
                }
                *(uint4 *)(dst + 16 * ww) = make_uint4(out[0], out[1], out[2], out[3]);       // (bases past the end: 'N', never read)
            }
        }
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
        ReportT *rp = reports + (on ? rep_off[r] : 0u);
        if (on && nc == 0) rp[0] = d_blank_report<ReportT>(-1);
        for (int i = 0; i < nc; i++) {
            DCand &c = cd[i];
            c.final_n = 0; c.n_a = 0; c.job_count = 0; c.job_first = 0;
            if (c.Score == 0) { rp[i] = d_blank_report<ReportT>(c.PairedIdx); continue; }
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
            // scheduling hints only: big = some segment pair is longer than PM_MAX (string path, maybe a wave-wide alignment), indel = two seeds on
            // different diagonals (an nw_alignment is certain)
            bool big = n > 0 && (s[0].rPos > PM_MAX || (int)rlen[r] - (s[n - 1].rPos + s[n - 1].rLen) > PM_MAX), indel = false;
            for (int k = 1; k < n; k++) {
                const int pd = (int)((s[k].gPos - s[k].rPos) - (s[k - 1].gPos - s[k - 1].rPos));
                const int rGaps = s[k].rPos - s[k - 1].rPos - s[k - 1].rLen;
                if (pd > pr.max_gaps && rGaps > 20) cnt++;
                if (rGaps > PM_MAX || rGaps + pd > PM_MAX) big = true;
                if (pd != 0) indel = true;
            }
            const uint32_t cls = d_cand_class(cnt > 0, big, indel, n);
            c.final_n = (int32_t)cls;                        // (k_order_items reads it; k_report sets the real value)
            atomicAdd(&s_hist[cls], 1u);
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
                    j.gPos = 0; j.rPos = 0; j.len = 0; j.n_chunks = 1; j.out_first = 0; j.done = 0; j.pad[0] = j.pad[1] = 0;
                    jobs[first + w++] = j;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < DG_COST_CLASSES && s_hist[threadIdx.x]) atomicAdd(class_hist + threadIdx.x, s_hist[threadIdx.x]);
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
// one diagonal with hits enters the scan: cnt hits, the first at read-gap position f, the last at la; g0 = the diagonal (window position minus read-gap position)
__device__ __forceinline__ void d_rs_fold_step(RsFold &st, int c, int f, int la, int64_t diag)
{
    st.s += c - 1;
    const int len = 8 + (la - f);
    if (len > st.max_len && st.s > (len - 8) / 2) { st.best_r = f; st.best_g = diag + f; st.max_len = len; st.s = 1; }
}

// A window of more than `chunk` diagonals is shared by several waves (one wave per 500 kb window was ~1000 serial trips, the kernel's
// duration).  The scan above is sequential over the diagonals that have hits, but almost none of them can change it: a diagonal with ONE hit adds 0 to s and
// offers len = 8, which only counts while max_len is still 0 -- so of a chunk's single-hit diagonals only the first can matter; diagonals with two or more
// hits all matter (s grows).  A chunk's wave therefore reports, in diagonal order, its first single-hit diagonal and every multi-hit one (a random 8-mer meets
// one of <= 256 read k-mers with p = 0.4 %: two on one diagonal ~ 7e-6 per diagonal, i.e. a handful per chunk plus the true match) as 64-bit entries
//   diagonal - D0 (32 bits) | last (9) | first (9) | cnt (9)        D0 = the window's lowest diagonal group
// and the wave that finishes a window's LAST chunk replays all entries through d_rs_fold_step -- the same sequence of state changes as one wave over the
// whole window.  Up to RS_ENT_INLINE entries live in the chunk's RsChunkOut record; a chunk with more (a read gap inside a satellite or an interspersed
// repeat against a window full of its copies: a third of the windows of a genome with human-like repeat content) writes ALL its entries as blocks of 64 into
// a pool, chained through blk_next, 64 at a time as its LDS buffer fills.  Only when the pool is exhausted does the last wave scan the whole window by
// itself, as round 4 did for every window.
#define RS_ENT_INLINE 15
#define RS_ENT_BUF 128          // entries the wave buffers in LDS: a group of 64 diagonals adds at most 64, a block of 64 leaves when 64 are there
struct __attribute__((aligned(16))) RsChunkOut { unsigned long long w[RS_ENT_INLINE + 1]; };      // w[0] = entries (32 bits) | first pool block << 32 (RS_NO_BLOCK: inline in w[1..]), or RS_OVERFLOW
#define RS_OVERFLOW 0xFFFFFFFFFFFFFFFFull
#define RS_NO_BLOCK 0xFFFFFFFFu
struct RsPool { unsigned long long *ent; uint32_t *next; unsigned int *top; uint32_t cap; };      // blocks of 64 entries; next[b] = the chunk's following block
struct RsEmit { unsigned long long *buf; int n_buf; uint32_t n_total; bool have_single, lost; uint32_t first_blk, last_blk; int64_t D0; int inline_max; };

// device-coherent accesses to what one wave of the launch writes and another reads, maybe on another XCD (`sc1`, as the scans' state words)
__device__ __forceinline__ void rs_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long rs_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rs_store32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t rs_load32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the first 64 buffered entries leave for the pool as one block (every lane calls it; n_buf >= 64, or `rest`: the chunk's last, partial block)
__device__ __forceinline__ void d_rs_flush_block(RsEmit &em, const RsPool &pool, const int lane, const bool rest)
{
    const int n = rest ? em.n_buf : 64;
    if (n <= 0) return;
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(pool.top, 1u);
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    if (b >= pool.cap) em.lost = true;
    else {
        if (lane < n) rs_store(pool.ent + (size_t)b * 64 + lane, em.buf[lane]);
        if (lane == 0) { rs_store32(pool.next + b, RS_NO_BLOCK); if (em.first_blk != RS_NO_BLOCK) rs_store32(pool.next + em.last_blk, b); }
        if (em.first_blk == RS_NO_BLOCK) em.first_blk = b;
        em.last_blk = b;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long mv = lane + 64 < em.n_buf ? em.buf[lane + 64] : 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    em.buf[lane] = mv;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    em.n_buf = em.n_buf > 64 ? em.n_buf - 64 : 0;
}

// LDS ring of diagonals, word-major (ring[w * RS_RING + slot]) so that 64 consecutive diagonals are
// read conflict-free; `dirty` has one bit per group of 64 diagonals that received a hit.
// Folds the complete diagonal groups below `lim` (all remaining ones when `final`) into st, in
// increasing diagonal order; every lane of the wave calls it with the same arguments.  Only the groups
// whose dirty bit is set cost anything: the mask is read once and walked in a scalar register.
// EMIT: instead of folding, the diagonals that can matter are appended to em's LDS buffer, full blocks of 64 leaving for the pool (see above); st only carries next_fin.
template <int WORDS, bool EMIT>
__device__ __forceinline__ void d_rs_finalize(unsigned long long *ring, uint32_t *dirty, RsFold &st, RsEmit &em, const RsPool &pool, const int64_t lim, const bool final, const int lane)
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
            const uint32_t pk = (uint32_t)cnt | ((uint32_t)first << 9) | ((uint32_t)last << 18);      // cnt <= 256, first / last < 256
            if (EMIT) {
                const unsigned long long singles = __ballot(any && cnt == 1);
                const bool take = any && (cnt > 1 || (!em.have_single && lane == __ffsll((long long)singles) - 1));
                const unsigned long long tm = __ballot(take);
                if (take) em.buf[em.n_buf + __popcll(tm & ((1ull << lane) - 1ull))] = ((unsigned long long)(uint32_t)(b0 + lane - em.D0) << 27) | pk;     // (n_buf < 64 here: room for 64 more)
                em.n_buf += __popcll(tm); em.n_total += (uint32_t)__popcll(tm);
                em.have_single = em.have_single || singles != 0ull;
                if (em.n_buf >= 64 && !em.lost) d_rs_flush_block(em, pool, lane, false);
                else if (em.n_buf >= 64) em.n_buf = 0;                                                                                               // (pool exhausted: the window will be scanned again whole)
            } else {
                unsigned long long mask = __ballot(any);
                while (mask) {
                    const int l = __ffsll((long long)mask) - 1;
                    const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)pk, l);
                    d_rs_fold_step(st, (int)(q & 511u), (int)((q >> 9) & 511u), (int)(q >> 18), b0 + l);
                    mask &= mask - 1;
                }
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

// Work list of the re-seeding kernel: per ring size (1, 2, 4 bitmap words per diagonal) the CHUNKS of that size's jobs.  A window of glen bases has the
// diagonals [D0, glen - 8] (D0 = -(read-gap span) rounded down to a group of 64); chunk 0 takes [D0, C), chunk k >= 1 takes [k C, (k + 1) C): n_chunks =
// ceil((glen - 7) / C), at least 1, C = RS_CHUNK_DIAGS (a multiple of the 4096-position pac super-chunk).  A window whose RsChunkOut records do not fit the
// buffer any more stays whole (one wave, as round 4 did for every window): no new capacity to overflow.
// Round 4: one wave per window, longest first -- a 500 kb window was ~1000 trips of one wave, and the kernel's duration.  Items are (job << 32 | chunk).
// info[0..2] = items per ring size (atomics), info[3] = C; out_top = RsChunkOut records handed out.  Every thread of k_order's grid takes jobs (a first version gave
// the whole list to the launch's last workgroup: 0.5 ms alone for the 31 k jobs of a spliced batch, in front of k_reseed).
#define RS_CHUNK_DIAGS 32768
__device__ __forceinline__ uint32_t d_rs_nchunks(int glen, uint32_t C) { return glen > 7 ? (uint32_t)(((int64_t)(glen - 7) + (int64_t)C - 1) / (int64_t)C) : 1u; }
struct RsOrder { DJob *jobs; const unsigned int *jobtop; uint32_t jobcap; unsigned long long *lists; uint32_t list_cap, out_cap; int max_words /* widest ring the host launches */;
                 uint32_t chunk0 /* diagonals per chunk: RS_CHUNK_DIAGS (a test hook lowers it) */; unsigned int *info; unsigned int *out_top; };
__device__ inline void d_order_jobs(const RsOrder &o)               // called by every thread of the grid
{
    DJob *__restrict__ jobs = o.jobs; unsigned long long *__restrict__ lists = o.lists;
    const unsigned int njobs = *o.jobtop < o.jobcap ? *o.jobtop : o.jobcap;
    const uint32_t C = o.chunk0;
    if (blockIdx.x == 0 && threadIdx.x == 0) o.info[3] = C;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < njobs; i += gridDim.x * blockDim.x) {
        DJob &j = jobs[i];
        if (j.found < 0) continue;                                        // too long for the LDS ring: serial path in k_report
        const int need_w = j.rl >= 8 ? (j.rl - 8) / 64 + 1 : 1;
        const int w = need_w <= 1 ? 0 : (need_w <= 2 ? 1 : 2);
        if ((1 << w) > o.max_words) { j.found = -1; continue; }                // (cannot happen: the host derives max_words from the longest read; the serial path in k_report would take it)
        uint32_t n = d_rs_nchunks(j.glen, C), first = 0;
        if (n > 1) {
            first = atomicAdd(o.out_top, n);
            if ((uint64_t)first + n > o.out_cap) { n = 1; first = 0; }          // (the records are full: this window by one wave)
        }
        const unsigned int at = atomicAdd(o.info + w, n);
        j.n_chunks = n; j.done = 0; j.out_first = first;
        for (uint32_t k = 0; k < n; k++) if (at + k < o.list_cap) lists[(size_t)w * o.list_cap + at + k] = ((unsigned long long)i << 32) | k;     // (list_cap = jobcap + out_cap: always enough)
    }
}

// k_order: k_report's work list from the candidates' classes and the class histogram of k_prep (and k_reseed's: d_order_jobs): items = (read << 32 | index of the candidate in cands[]) grouped by
// class, class 0 first (any order inside a class: the order decides who computes what when, never a result); info[0] = items of class 0 (they wait for
// k_reseed), info[1] = all items.  One launch over the listed reads; a workgroup reserves its share of every class with one atomic per class.
__global__ void __launch_bounds__(256)
k_order(int paired, const uint32_t *__restrict__ slow_units, const DSizes *__restrict__ sizes, const uint32_t *__restrict__ seed_off, const DCand *__restrict__ cands,
        const uint32_t *__restrict__ ncand, const unsigned int *__restrict__ class_hist, unsigned int *class_fill, unsigned long long *__restrict__ items, uint32_t *__restrict__ info,
        const RsOrder jobs_order, const int *__restrict__ abort_p)
{
    __shared__ uint32_t s_start[DG_COST_CLASSES], s_cnt[DG_COST_CLASSES], s_base[DG_COST_CLASSES];
    if (*abort_p >= DG_ABORT) return;
    d_order_jobs(jobs_order);                                                   // k_reseed's work list: every thread takes re-seeding jobs (above); then k_report's
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {                                        // exclusive scan of the 32 class totals by the first wave
        uint32_t v = lane < DG_COST_CLASSES ? class_hist[lane] : 0u, incl = v;
        for (int o = 1; o < DG_COST_CLASSES; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane < DG_COST_CLASSES) { s_start[lane] = incl - v; s_cnt[lane] = 0; }
        if (blockIdx.x == 0) {
            if (lane == 1) info[0] = incl;                         // keys 0 and 1 = class 0
            if (lane == DG_COST_CLASSES - 1) info[1] = incl;
        }
    }
    __syncthreads();
    const unsigned int n_items = sizes->n_slow_units * (paired ? 2u : 1u);
    for (unsigned int base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {      // uniform per workgroup
        const unsigned int it = base + threadIdx.x;
        const bool on = it < n_items;
        const uint32_t r = on ? (paired ? 2u * slow_units[it >> 1] + (it & 1u) : slow_units[it]) : 0u;
        const uint32_t c0 = seed_off[r];
        const int nc = on ? (int)ncand[r] : 0;
        for (int i = 0; i < nc; i++) if (cands[c0 + i].Score != 0) atomicAdd(&s_cnt[cands[c0 + i].final_n & (DG_COST_CLASSES - 1)], 1u);
        __syncthreads();
        if (threadIdx.x < DG_COST_CLASSES) { const uint32_t k = s_cnt[threadIdx.x]; s_base[threadIdx.x] = s_start[threadIdx.x] + (k ? atomicAdd(class_fill + threadIdx.x, k) : 0u); s_cnt[threadIdx.x] = 0; }
        __syncthreads();
        for (int i = 0; i < nc; i++) if (cands[c0 + i].Score != 0) {
            const uint32_t cls = (uint32_t)cands[c0 + i].final_n & (DG_COST_CLASSES - 1);
            items[s_base[cls] + atomicAdd(&s_cnt[cls], 1u)] = ((unsigned long long)r << 32) | (unsigned long long)(c0 + (uint32_t)i);
        }
        __syncthreads();
        if (threadIdx.x < DG_COST_CLASSES) s_cnt[threadIdx.x] = 0;
        __syncthreads();
    }
}

#define RS_TAB      512          // open-addressing table: 8-mer id -> first entry of km[] with that id (at most 256 ids: half full)
#define RS_FLT_BITS 14           // the bitmap filter looks at the low 14 bits of an id (2 KB); the table decides
#define RS_EMPTY    0xFFFFFFFFu
__device__ __forceinline__ uint32_t d_rs_hash(uint32_t w) { return (w ^ (w >> 7)) & (RS_TAB - 1); }

// what k_reseed keeps in LDS for the job at hand (one wave = one workgroup)
template <int WORDS>
struct RsLds {
    unsigned char rs[RS_MAX_RL + 9];
    uint32_t km[RS_MAX_RL + 1];
    uint32_t flt[(1 << RS_FLT_BITS) / 32];
    uint32_t tab[RS_TAB];
    unsigned long long ring[RS_RING * WORDS];
    uint32_t pacbuf[RS_SUPER / 16 + 16];         // one super-chunk of pac: RS_SUPER / 4 bytes + alignment slack
    unsigned long long ent[RS_ENT_BUF];
    uint32_t dirty;
    int nk;
    unsigned int next;
};

// The window positions whose hits fall on the diagonals [d_lo, d_hi) are streamed once: positions [g_lo, g_hi], g_lo a multiple of RS_SUPER.  EMIT = false: the
// whole window by this wave, folded as it goes (st); EMIT = true: one chunk, entries to em.  Per trip a wave looks at RS_CHUNK window positions, RS_PPL
// consecutive ones per lane: the lane's 15 bases come from two dwords of the staged pac bytes (reverse half: bit-reversed and complemented once, not per
// 8-mer), every 8-mer is one bit-field extract, the bitmap says which of them can be in the read gap at all (0.3 % are), and those go through the table --
// one or two LDS reads -- to their entries of km[].  Returns the trips made.
template <int WORDS, bool EMIT>
__device__ __forceinline__ unsigned int d_rs_scan(const DIndex &ix, RsLds<WORDS> &S, const DJob &job, const int nk, const int span, const int g_lo, const int g_hi,
                                                  const int64_t d_lo, const int64_t d_hi, RsFold &st, RsEmit &em, const RsPool &pool, const int lane)
{
    const int64_t L = ix.l_pac;
    const int glen = job.glen;
    unsigned int n_trips = 0;
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
        const uint4 *src = (const uint4 *)(ix.pac + super_base(g_lo));
        pre = src[lane]; if (lane < 4) pre_t = src[64 + lane];
    }
    int64_t B0 = 0;
    for (int g0 = g_lo; g0 <= g_hi; g0 += RS_CHUNK) {
        n_trips++;
        d_rs_finalize<WORDS, EMIT>(S.ring, &S.dirty, st, em, pool, (int64_t)g0 - span, false, lane);
        const int p0 = g0 + RS_PPL * lane;
        uint32_t y = 0;                 // the lane's RS_PPL + 7 window bases (and one more), first base in the top bits
        if (fwd || rev) {
            if (g0 % RS_SUPER == 0) {                              // (uniform) a new super-chunk: the fetch issued one super-chunk ago has arrived
                B0 = super_base(g0);
                RS_WAVE_SYNC();
                ((uint4 *)S.pacbuf)[lane] = pre; if (lane < 4) ((uint4 *)S.pacbuf)[64 + lane] = pre_t;
                if (g0 + RS_SUPER <= g_hi) {
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
            const uint32_t w0 = __builtin_bswap32(S.pacbuf[m]), w1 = __builtin_bswap32(S.pacbuf[m + 1]);
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
        for (int j = 0; j < RS_PPL; j++) {                          // position j's filter bit enters at the top and moves down: one v_alignbit instead of and / shift / or
            const uint32_t fw = (y >> (16 - 2 * j)) & ((1u << RS_FLT_BITS) - 1u);
            pass = __builtin_amdgcn_alignbit(S.flt[fw >> 5] >> (fw & 31), pass, 1);
        }
        pass >>= 32 - RS_PPL;
        pass &= n_valid >= RS_PPL ? (1u << RS_PPL) - 1u : (n_valid > 0 ? (1u << n_valid) - 1u : 0u);
        while (pass) {
            const int j = __ffs((int)pass) - 1;
            pass &= pass - 1;
            const uint32_t w = (y >> (16 - 2 * j)) & 0xFFFFu;
            uint32_t slot = d_rs_hash(w), e = S.tab[slot];
            for (int t = 0; e != RS_EMPTY && (e >> 9) != w && t < RS_TAB; t++) { slot = (slot + 1) & (RS_TAB - 1); e = S.tab[slot]; }
            if (e == RS_EMPTY || (e >> 9) != w) continue;                          // only the low bits of the id were in the read gap
            const int p = p0 + j;
            for (int lo = (int)(e & 511u); lo < nk; lo++) {
                const uint32_t key = S.km[lo];
                if ((key >> 9) != w) break;
                const int rp = (int)(key & 511u);
                const int64_t d = (int64_t)p - rp;
                if (EMIT && (d < d_lo || d >= d_hi)) continue;                     // another chunk's diagonal
                atomicOr(&S.ring[(rp >> 6) * RS_RING + (int)((uint64_t)d & (RS_RING - 1))], 1ull << (rp & 63));
                atomicOr(&S.dirty, 1u << ((uint32_t)((uint64_t)d >> 6) & (RS_RING / 64 - 1)));
            }
        }
        RS_WAVE_SYNC();
    }
    d_rs_finalize<WORDS, EMIT>(S.ring, &S.dirty, st, em, pool, d_hi, true, lane);
    return n_trips;
}

// WORDS = 64-bit bitmap words per diagonal: a job with read gap rl needs (rl - 8) / 64 + 1 of them; the kernel is
// instantiated for 1, 2 and 4 (8 / 16 / 32 KB of ring + 6.5 KB of tables: 10 / 6 / 4 waves per CU) and each instance takes the
// items of its size (the host only launches the sizes the batch's longest read allows: a gap is at most rlen - 32, two 16-base seeds around it).
// Keys of the read gap's 8-mers are (id << 9 | position): ids stay below 2^17 (CreateKmerID adds
// nst_nt4_table values up to 5 without masking), positions below 512.
template <int WORDS>
__global__ void __launch_bounds__(64)
k_reseed(const DIndex ix, const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off,
         DJob *__restrict__ jobs, const unsigned long long *__restrict__ list, const unsigned int *__restrict__ count_p, const unsigned int *__restrict__ chunk_p, unsigned int *ticket,
         RsChunkOut *__restrict__ outs, const RsPool pool, int inline_max /* RS_ENT_INLINE (a test hook lowers it: more chunks go through the pool) */, unsigned long long *ctr, const int *__restrict__ abort_p)
{
    __shared__ RsLds<WORDS> S;
    uint32_t *tmpk = (uint32_t *)S.ring;                     // the unsorted keys live in the ring's space until the ring is cleared
    const int lane = threadIdx.x;
    if (*abort_p >= DG_ABORT) return;
    const unsigned int n_items = *count_p;             // this ring size's (job, chunk) items (k_order_jobs)
    const uint32_t C = *chunk_p;
    const uint64_t t_start = wall_clock64();
    unsigned long long n_done = 0, w_done = 0, n_trips = 0, n_whole_again = 0, n_items_done = 0, n_pool_chunks = 0;
    while (true) {
        __syncthreads();
        if (lane == 0) S.next = atomicAdd(ticket, 1u);
        __syncthreads();
        if (S.next >= n_items) break;
        const unsigned long long item = list[S.next];
        const unsigned int jb = (unsigned int)(item >> 32), ck = (unsigned int)item;
        const DJob job = jobs[jb];
        const int rl = job.rl, glen = job.glen;
        int thr = (int)(rl * 0.85); if (thr < 8) thr = 8;
        n_items_done++;
        if (ck == 0) { n_done += 1; w_done += (unsigned long long)(glen > 0 ? glen : 0); }
        const unsigned char *rd = seq + seq_off[job.read] + job.rBegin;
        bool plain = true;                                   // only A, C, G, T in the gap (either case): the ids need no carries, and the reference's restart
                                                             // after an 'N' (which leaves its window one base behind its label from there on) does not happen
        for (int i = lane; i < rl + 8; i += 64) {
            const unsigned char ch = i < rl ? rd[i] : (unsigned char)'N';
            S.rs[i] = ch;
            if (i < rl && d_nt4(ch) > 3) plain = false;
        }
        for (int i = lane; i < (1 << RS_FLT_BITS) / 32; i += 64) S.flt[i] = 0;
        for (int i = lane; i < RS_TAB; i += 64) S.tab[i] = RS_EMPTY;
        if (lane == 0) S.dirty = 0;
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
                for (int i = 0; i < 8; i++) wid = (wid << 2) + (d_nt4(S.rs[head + i]) & 3u);
                const unsigned long long mk = __ballot(ok);
                if (ok) tmpk[nk + __popcll(mk & ((1ull << lane) - 1ull))] = (wid << 9) | (uint32_t)head;
                nk += __popcll(mk);
            }
            if (lane == 0) S.nk = nk;
        } else if (lane == 0) {   // the reference's rolling form, literally
            int nk = 0, count = 0, head, tail = 0;
            uint32_t wid = 0;
            while (count < 8 && tail < rl) { if (S.rs[tail++] != 'N') count++; else count = 0; }
            if (count == 8) {
                head = tail - 8; wid = 0;
                for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(S.rs[i]);
                tmpk[nk++] = (wid << 9) | (uint32_t)head;
                for (head += 1; tail < rl; head++, tail++) {
                    if (S.rs[tail] != 'N') {
                        wid = ((wid & 0x3FFF) << 2) + d_nt4(S.rs[tail]);
                        tmpk[nk++] = (wid << 9) | (uint32_t)head;
                    } else {
                        count = 0; tail++;
                        while (count < 8 && tail < rl) { if (S.rs[tail++] != 'N') count++; else count = 0; }
                        if (count == 8) {
                            head = tail - 8; wid = 0;
                            for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(S.rs[i]);
                            tmpk[nk++] = (wid << 9) | (uint32_t)head;
                        } else break;
                    }
                }
            }
            S.nk = nk;
        }
        __syncthreads();
        const int nk = S.nk;
        int found = 0;
        RsFold st; st.s = 1; st.max_len = 0; st.best_r = 0; st.best_g = 0; st.next_fin = 0;
        bool result = job.n_chunks <= 1;                     // this wave writes the job's result (a chunk's wave: only the one that finishes the last chunk)
        if (nk > 0 && glen >= 8) {
            for (int e = lane; e < nk; e += 64) {           // rank sort by (id, position); keys are distinct
                const uint32_t key = tmpk[e];
                int rank = 0;
                for (int j = 0; j < nk; j++) rank += tmpk[j] < key ? 1 : 0;
                S.km[rank] = key;
                const uint32_t fw = (key >> 9) & ((1u << RS_FLT_BITS) - 1u);
                atomicOr(&S.flt[fw >> 5], 1u << (fw & 31));
            }
            __syncthreads();
            for (int i = lane; i < RS_RING * WORDS; i += 64) S.ring[i] = 0;       // (tmpk is dead from here)
            for (int e = lane; e < nk; e += 64) {            // the first entry of every id goes into the table
                const uint32_t id = S.km[e] >> 9;
                if (e == 0 || (S.km[e - 1] >> 9) != id) {
                    uint32_t slot = d_rs_hash(id);
                    while (atomicCAS(&S.tab[slot], RS_EMPTY, (id << 9) | (uint32_t)e) != RS_EMPTY) slot = (slot + 1) & (RS_TAB - 1);
                }
            }
            __syncthreads();
            const int span = rl - 8;
            const int64_t D0 = -(int64_t)(((span + 63) >> 6) << 6);          // the window's lowest diagonal, aligned to the 64-diagonal groups
            RsEmit em; em.buf = S.ent; em.n_buf = 0; em.n_total = 0; em.have_single = false; em.lost = false; em.first_blk = em.last_blk = RS_NO_BLOCK; em.D0 = D0; em.inline_max = inline_max;
            bool whole = job.n_chunks <= 1;
            if (!whole) {
                // ---- one chunk of a shared window ----
                const int64_t d_lo = ck == 0 ? D0 : (int64_t)ck * C;
                const int64_t d_hi = (int64_t)(ck + 1) * C < (int64_t)glen - 7 ? (int64_t)(ck + 1) * C : (int64_t)glen - 7;
                const int64_t g_top = d_hi - 1 + span;
                st.next_fin = d_lo;
                n_trips += d_rs_scan<WORDS, true>(ix, S, job, nk, span, ck == 0 ? 0 : (int)((int64_t)ck * C), (int)(g_top < (int64_t)glen - 8 ? g_top : (int64_t)glen - 8), d_lo, d_hi, st, em, pool, lane);
                RS_WAVE_SYNC();
                RsChunkOut *o = outs + job.out_first + ck;
                const bool via_pool = em.first_blk != RS_NO_BLOCK || (int)em.n_total > inline_max;
                if (via_pool && !em.lost) d_rs_flush_block(em, pool, lane, true);                       // the last, partial block
                if (via_pool) n_pool_chunks++;
                if (lane <= RS_ENT_INLINE)
                    rs_store(&o->w[lane], lane == 0 ? (em.lost ? RS_OVERFLOW : ((unsigned long long)(via_pool ? em.first_blk : RS_NO_BLOCK) << 32) | em.n_total)
                                                    : ((!via_pool && lane - 1 < (int)em.n_total) ? S.ent[lane - 1] : 0ull));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the record (and the pool blocks) are at the device-coherent level before the count says so
                unsigned int before = 0;
                if (lane == 0) before = __hip_atomic_fetch_add(&jobs[jb].done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
                if (before + 1u == job.n_chunks) {
                    // ---- the window's last chunk: replay every chunk's entries in diagonal order ----
                    result = true;
                    st.s = 1; st.max_len = 0; st.best_r = 0; st.best_g = 0;
                    bool over = false;
                    auto fold_lanes = [&](const unsigned long long v, const int first_lane, const int n_ent) {        // entries held by lanes first_lane .. first_lane + n_ent - 1, in lane order
                        for (int e = 0; e < n_ent; e++) {
                            const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, first_lane + e), hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), first_lane + e);
                            const unsigned long long en = ((unsigned long long)hi32 << 32) | lo32;
                            d_rs_fold_step(st, (int)(en & 511u), (int)((en >> 9) & 511u), (int)((en >> 18) & 511u), D0 + (int64_t)(en >> 27));
                        }
                    };
                    for (uint32_t c0 = 0; c0 < job.n_chunks && !over; c0 += 4) {       // four records per wave-wide load
                        const uint32_t cc = c0 + (uint32_t)(lane >> 4);
                        const unsigned long long v = cc < job.n_chunks ? rs_load(&outs[job.out_first + cc].w[lane & 15]) : 0ull;
                        for (int q = 0; q < 4 && c0 + q < job.n_chunks && !over; q++) {
                            const uint32_t ne = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 16 * q), blk0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 16 * q);
                            if (ne == 0xFFFFFFFFu && blk0 == 0xFFFFFFFFu) { over = true; break; }
                            if (blk0 == RS_NO_BLOCK) { fold_lanes(v, 16 * q + 1, (int)ne); continue; }
                            uint32_t bk = blk0;
                            for (uint32_t left = ne; left > 0; ) {                      // the chunk's chain of pool blocks
                                const int nb = left < 64u ? (int)left : 64;
                                const unsigned long long bv = lane < nb ? rs_load(pool.ent + (size_t)bk * 64 + lane) : 0ull;
                                const uint32_t nx = rs_load32(pool.next + bk);
                                fold_lanes(bv, 0, nb);
                                left -= (uint32_t)nb; bk = nx;
                                if (left && bk == RS_NO_BLOCK) { over = true; break; }            // (cannot happen: a chain is as long as its count says)
                            }
                        }
                    }
                    if (over) {                                      // the pool ran out under some chunk: the whole window by this wave (its tables are in LDS)
                        whole = true; n_whole_again++;
                        for (int i = lane; i < RS_RING * WORDS; i += 64) S.ring[i] = 0;
                        if (lane == 0) S.dirty = 0;
                        __syncthreads();
                    }
                }
            }
            if (whole) {
                st.s = 1; st.max_len = 0; st.best_r = 0; st.best_g = 0; st.next_fin = D0;
                n_trips += d_rs_scan<WORDS, false>(ix, S, job, nk, span, 0, glen - 8, D0, (int64_t)(glen - 8) + 1, st, em, pool, lane);
            }
            found = (st.max_len >= thr && st.max_len > 0) ? 1 : 0;
        } else if (job.n_chunks > 1) {                       // (no k-mer in the gap: every chunk's wave finds that out; the last one reports)
            unsigned int before = 0;
            if (lane == 0) before = __hip_atomic_fetch_add(&jobs[jb].done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            result = (unsigned int)__builtin_amdgcn_readfirstlane((int)before) + 1u == job.n_chunks;
        }
        if (result && lane == 0) {
            DJob &o = jobs[jb];
            o.found = found; o.len = st.max_len; o.rPos = st.best_r + job.rBegin; o.gPos = st.best_g + job.Lb;
        }
    }
    if (lane == 0) {
        if (n_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED, n_done);
        if (w_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEEDW, w_done);
        if (n_trips) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_TRIPS, n_trips);
        if (n_whole_again) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_WHOLE, n_whole_again);
        if (n_items_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_ITEMS, n_items_done);
        if (n_pool_chunks) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_POOLED, n_pool_chunks);
        atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED_TICKS, (unsigned long long)(wall_clock64() - t_start));
    }
}
