// dart_amd/csrc/dg_chain.h -- seeds -> alignment candidates, mate pairing, redundancy filter.
//
// Replaces the tail of IdentifySeedPairs (the sort, AlignmentCandidates.cpp:212),
// GenerateAlignmentCandidate (:241-288), CheckPairedAlignmentCandidates (Mapping.cpp:403-450),
// RemoveUnMatedAlignmentCandidates (:452-477) and RemoveRedundantCandidates (:371-401).
//
// One lane = one read pair (or one single read).  A candidate is a contiguous run of the
// gPos-sorted seed list, so it is stored as (first,count) into the read's seed segment: nothing is
// copied.  Per-read arrays are tiny (typically 1-5 seeds); the kernel is latency-, not
// bandwidth-bound and costs a few percent of k_seed.
#pragma once
#include "dg_common.h"

// in-place sort of a lane-private seed segment by (gPos,rPos): insertion sort for short lists,
// heap sort beyond (both give the unique order of a total order up to identical elements)
__host__ __device__ inline void d_sort_seeds(DSeed *a, int n)
{
    if (n < 2) return;
    if (n == 2) { const DSeed x = a[0], y = a[1]; if (d_seed_less(y, x)) { a[0] = y; a[1] = x; } return; }
    if (n == 3) {
        DSeed x = a[0], y = a[1], z = a[2], t;
        bool sw = false;
        if (d_seed_less(y, x)) { t = x; x = y; y = t; sw = true; }
        if (d_seed_less(z, y)) { t = y; y = z; z = t; sw = true; if (d_seed_less(y, x)) { t = x; x = y; y = t; } }
        if (sw) { a[0] = x; a[1] = y; a[2] = z; }
        return;
    }
    if (n <= 32) {
        for (int i = 1; i < n; i++) {
            DSeed x = a[i];
            int j = i;
            while (j > 0 && d_seed_less(x, a[j - 1])) { a[j] = a[j - 1]; j--; }
            a[j] = x;
        }
        return;
    }
    for (int st = n / 2 - 1; st >= 0; st--) {          // heapify
        int root = st;
        DSeed x = a[root];
        while (true) {
            int ch = 2 * root + 1;
            if (ch >= n) break;
            if (ch + 1 < n && d_seed_less(a[ch], a[ch + 1])) ch++;
            if (!d_seed_less(x, a[ch])) break;
            a[root] = a[ch]; root = ch;
        }
        a[root] = x;
    }
    for (int end = n - 1; end > 0; end--) {
        DSeed x = a[end];
        a[end] = a[0];
        int root = 0;
        while (true) {
            int ch = 2 * root + 1;
            if (ch >= end) break;
            if (ch + 1 < end && d_seed_less(a[ch], a[ch + 1])) ch++;
            if (!d_seed_less(x, a[ch])) break;
            a[root] = a[ch]; root = ch;
        }
        a[root] = x;
    }
}

// GenerateAlignmentCandidate :241-288.  seeds = the read's sorted segment (absolute base `base`).
// Every seed is fetched once (whole 24-byte record) and the chain tail is kept in registers.
__host__ __device__ inline int d_gen_candidates(const DIndex &ix, const DParams &pr, int rlen, const DSeed *__restrict__ s, int num, uint32_t base, DCand *__restrict__ out)
{
    int nc = 0;
    if (num == 0) return 0;
    const int thr = (int)(rlen * 0.3);
    int i = 0;
    DSeed si = s[0];
    while (si.gPos - si.rPos < 0) { if (++i >= num) return 0; si = s[i]; }
    while (i < num) {
        int score = si.rLen, k;
        int64_t pd_j = si.gPos - si.rPos, g_j = si.gPos;
        int r_j = si.rPos;
        const int64_t pd0 = pd_j;
        DSeed sk = si;
        for (k = i + 1; k < num; k++) {
            sk = s[k];
            const int64_t pd_k = sk.gPos - sk.rPos;
            int64_t pd = pd_k - pd_j;
            if (pd < 0) pd = -pd;
            bool ok = pd < pr.max_gaps;
            if (!ok && pd < pr.max_intron) {
                const int lb = d_loc_lower_bound(ix, g_j);
                ok = sk.gPos < ix.loc_key[lb] && sk.rPos > r_j;
            }
            if (!ok) break;
            score += sk.rLen;
            pd_j = pd_k; g_j = sk.gPos; r_j = sk.rPos;
        }
        if (score > thr) {
            DCand c;
            c.PosDiff = pd0 < 0 ? 0 : pd0;
            c.first = (int32_t)(base + i); c.count = k - i; c.Score = score; c.PairedIdx = -1; c.SJtype = -1;
            c.work_off = 0; c.final_n = 0; c.n_a = 0; c.job_first = 0; c.job_count = 0;
            out[nc++] = c;
        }
        i = k;
        si = sk;            // the seed that broke the chain starts the next one (unused when k == num)
    }
    return nc;
}

__host__ __device__ inline void d_remove_redundant(DCand *c, int n)   // Mapping.cpp:371-401
{
    if (n <= 1) return;
    int s1 = 0, s2 = 0;
    for (int i = 0; i < n; i++) {
        const int sc = c[i].Score;
        if (sc > s2) {
            if (sc >= s1) { s2 = s1; s1 = sc; }
            else s2 = sc;
        } else if (sc == s2) s2 = s1;
    }
    const int thr = (s1 == s2 || s1 - s2 > 20) ? s1 : s2;
    for (int i = 0; i < n; i++) if (c[i].Score < thr) c[i].Score = 0;
}

__host__ __device__ inline bool d_check_paired(DCand *c1, int n1, DCand *c2, int n2)   // Mapping.cpp:403-450
{
    bool pairing = false;
    if (n1 * n2 > 1000) { d_remove_redundant(c1, n1); d_remove_redundant(c2, n2); }
    for (int i = 0; i < n1; i++) {
        if (c1[i].Score == 0) continue;
        int best = -1;
        int64_t min_dist = 2000000;
        for (int j = 0; j < n2; j++) {
            if (c2[j].Score == 0 || c2[j].PosDiff < c1[i].PosDiff) continue;
            const int64_t d = c2[j].PosDiff - c1[i].PosDiff;   // >= 0 here
            if (d < min_dist) { best = j; min_dist = d; }
        }
        if (best != -1) {
            const int j = best;
            if (c2[j].PairedIdx == -1) {
                pairing = true;
                c1[i].PairedIdx = j; c2[j].PairedIdx = i;
            } else if (c1[i].Score > c1[c2[j].PairedIdx].Score) {
                c1[c2[j].PairedIdx].PairedIdx = -1;
                c1[i].PairedIdx = j; c2[j].PairedIdx = i;
            }
        }
    }
    return pairing;
}

__host__ __device__ inline void d_remove_unmated(DCand *c1, int n1, DCand *c2, int n2)   // Mapping.cpp:452-477
{
    for (int i = 0; i < n1; i++) {
        if (c1[i].PairedIdx == -1) c1[i].Score = 0;
        else { const int j = c1[i].PairedIdx; c1[i].Score = c2[j].Score = c1[i].Score + c2[j].Score; }
    }
    for (int j = 0; j < n2; j++) if (c2[j].PairedIdx == -1) c2[j].Score = 0;
}

// working-region size of one candidate in the report stage: tandem/translocation clean-up never
// grows the list; re-seeding adds <= n-1, gap filling <= 2 per adjacent pair, normal pairs <= 1 per
// adjacent pair, plus merge slack -> 14n+8 seeds is a safe bound
__device__ __forceinline__ uint32_t d_work_need(int count) { return 14u * (uint32_t)count + 8u; }

// a unit (pair / single read) is "heavy" when a mate has more than CH_HEAVY seeds: repeat families.
// Heavy units are a fraction of a percent of the input but their per-read arrays are 10-100x
// longer; one lane walking them through global memory used to set the kernel's run time, so they
// get a wave each (k_chain_heavy) and k_chain skips them.
#define CH_HEAVY 16
__device__ __forceinline__ bool d_unit_is_heavy(const uint32_t *seed_off, int paired, int u)
{
    const int r1 = paired ? 2 * u : u;
    if (seed_off[r1 + 1] - seed_off[r1] > CH_HEAVY) return true;
    return paired && seed_off[r1 + 2] - seed_off[r1 + 1] > CH_HEAVY;
}

__global__ void __launch_bounds__(256)
k_chain(const DIndex ix, const DParams pr, int n_units, int paired, const uint16_t *__restrict__ rlen,
        const uint32_t *__restrict__ seed_off, DSeed *__restrict__ seeds, DCand *__restrict__ cands,
        uint32_t *__restrict__ ncand, uint32_t *__restrict__ nrep, uint32_t *__restrict__ work_need, unsigned long long *ctr)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nc_total = 0;
    if (u < n_units && !d_unit_is_heavy(seed_off, paired, u)) {
        const int r1 = paired ? 2 * u : u;
        const uint32_t b1 = seed_off[r1], e1 = seed_off[r1 + 1];
        d_sort_seeds(seeds + b1, (int)(e1 - b1));
        DCand *c1 = cands + b1;
        const int n1 = d_gen_candidates(ix, pr, rlen[r1], seeds + b1, (int)(e1 - b1), b1, c1);
        if (paired) {
            const int r2 = r1 + 1;
            const uint32_t b2 = seed_off[r2], e2 = seed_off[r2 + 1];
            d_sort_seeds(seeds + b2, (int)(e2 - b2));
            DCand *c2 = cands + b2;
            const int n2 = d_gen_candidates(ix, pr, rlen[r2], seeds + b2, (int)(e2 - b2), b2, c2);
            if (d_check_paired(c1, n1, c2, n2)) d_remove_unmated(c1, n1, c2, n2);
            d_remove_redundant(c1, n1); d_remove_redundant(c2, n2);
            uint32_t w = 0;
            for (int i = 0; i < n2; i++) if (c2[i].Score > 0) w += d_work_need(c2[i].count);
            ncand[r2] = (uint32_t)n2; nrep[r2] = n2 > 0 ? (uint32_t)n2 : 1u; work_need[r2] = w;
            nc_total += (unsigned long long)n2;
        } else d_remove_redundant(c1, n1);
        uint32_t w = 0;
        for (int i = 0; i < n1; i++) if (c1[i].Score > 0) w += d_work_need(c1[i].count);
        ncand[r1] = (uint32_t)n1; nrep[r1] = n1 > 0 ? (uint32_t)n1 : 1u; work_need[r1] = w;
        nc_total += (unsigned long long)n1;
    }
    d_wave_add(ctr + CTR_CANDS, nc_total);
}

// ---------------------------------------------------------------------------------------------
// k_chain_heavy: one wave = one heavy unit.  Seeds of both mates are staged in LDS and sorted with
// a wave-wide bitonic network; candidates are built by lane 0 from LDS; the O(n1*n2) mate search
// of CheckPairedAlignmentCandidates runs 64 candidates at a time with a wave min-reduction that
// keeps the reference's tie-break (first minimum).  Anything larger than the LDS staging
// (> CH_MAXS seeds or > CH_MAXC candidates per mate) takes the serial global-memory route.
// ---------------------------------------------------------------------------------------------
#define CH_MAXS 512
#define CH_MAXC 192

__device__ inline void d_bitonic_sort_lds(DSeed *a, int n, int lane)   // n <= CH_MAXS, one wave
{
    int m = 1; while (m < n) m <<= 1;
    for (int i = n + lane; i < m; i += 64) { a[i].gPos = 0x7FFFFFFFFFFFFFFFll; a[i].rPos = 0x7FFFFFFF; }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < m / 2; t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const bool up = (i & k) == 0;
                const DSeed x = a[i], y = a[l];
                if (d_seed_less(y, x) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
}

#ifdef DG_PROFILE_CLASSES
__device__ unsigned long long g_chain_max, g_chain_units, g_chain_cycles;
#endif
__global__ void __launch_bounds__(64)
k_chain_heavy(const DIndex ix, const DParams pr, int n_units, int paired, const uint16_t *__restrict__ rlen,
              const uint32_t *__restrict__ seed_off, DSeed *__restrict__ seeds, DCand *__restrict__ cands,
              uint32_t *__restrict__ ncand, uint32_t *__restrict__ nrep, uint32_t *__restrict__ work_need,
              const uint32_t *__restrict__ heavy_list, const unsigned int *__restrict__ n_heavy_p, unsigned long long *ctr)
{
    __shared__ DSeed ls[2][CH_MAXS];
    __shared__ DCand lc[2][CH_MAXC];
    __shared__ int s_n[2], s_pairing;
    const int lane = threadIdx.x;
    const unsigned int n_heavy = *n_heavy_p;
    unsigned long long nc_total = 0;
    for (unsigned int hi = blockIdx.x; hi < n_heavy; hi += gridDim.x) {
        const int u = (int)heavy_list[hi];
        const int nm = paired ? 2 : 1;
#ifdef DG_PROFILE_CLASSES
        const long long t_begin = clock64();
#endif
        const int r1 = paired ? 2 * u : u;
        uint32_t b[2], n[2];
        bool fits = true;
        for (int m = 0; m < nm; m++) { b[m] = seed_off[r1 + m]; n[m] = seed_off[r1 + m + 1] - b[m]; fits = fits && n[m] <= CH_MAXS; }
        __syncthreads();
        if (fits) {
            for (int m = 0; m < nm; m++) {
                for (uint32_t i = lane; i < n[m]; i += 64) ls[m][i] = seeds[b[m] + i];
                __syncthreads();
                d_bitonic_sort_lds(ls[m], (int)n[m], lane);
                for (uint32_t i = lane; i < n[m]; i += 64) seeds[b[m] + i] = ls[m][i];
            }
            __syncthreads();
        }
        // candidates: lane 0, from LDS when staged (GenerateAlignmentCandidate :241-288)
        if (lane == 0) {
            for (int m = 0; m < nm; m++) {
                if (!fits) d_sort_seeds(seeds + b[m], (int)n[m]);
                const DSeed *src = fits ? ls[m] : seeds + b[m];
                int nc = d_gen_candidates(ix, pr, rlen[r1 + m], src, (int)n[m], b[m], cands + b[m]);
                s_n[m] = nc;
            }
        }
        __syncthreads();
        const int n1 = s_n[0], n2 = paired ? s_n[1] : 0;
        DCand *c1 = cands + b[0], *c2 = paired ? cands + b[1] : nullptr;
        if (paired) {
            const bool stage = n1 <= CH_MAXC && n2 <= CH_MAXC;
            if (stage) {
                for (int i = lane; i < n1; i += 64) lc[0][i] = c1[i];
                for (int i = lane; i < n2; i += 64) lc[1][i] = c2[i];
                __syncthreads();
                DCand *a1 = lc[0], *a2 = lc[1];
                if (n1 * n2 > 1000) { if (lane == 0) { d_remove_redundant(a1, n1); d_remove_redundant(a2, n2); } __syncthreads(); }
                if (lane == 0) s_pairing = 0;
                __syncthreads();
                for (int i = 0; i < n1; i++) {                       // CheckPairedAlignmentCandidates :416-448
                    if (a1[i].Score == 0) continue;                  // uniform: read from LDS by every lane
                    const int64_t pd1 = a1[i].PosDiff;
                    int64_t best_d = 2000000; int best_j = 0x7FFFFFFF;
                    for (int j = lane; j < n2; j += 64) {
                        if (a2[j].Score == 0 || a2[j].PosDiff < pd1) continue;
                        const int64_t d = a2[j].PosDiff - pd1;
                        if (d < best_d) { best_d = d; best_j = j; }   // per lane: first minimum in increasing j
                    }
                    for (int o = 32; o > 0; o >>= 1) {               // wave minimum of (dist, j): first minimum overall
                        const int64_t od = __shfl_xor(best_d, o, 64); const int oj = __shfl_xor(best_j, o, 64);
                        if (od < best_d || (od == best_d && oj < best_j)) { best_d = od; best_j = oj; }
                    }
                    if (lane == 0 && best_d < 2000000) {
                        const int j = best_j;
                        if (a2[j].PairedIdx == -1) { s_pairing = 1; a1[i].PairedIdx = j; a2[j].PairedIdx = i; }
                        else if (a1[i].Score > a1[a2[j].PairedIdx].Score) { a1[a2[j].PairedIdx].PairedIdx = -1; a1[i].PairedIdx = j; a2[j].PairedIdx = i; }
                    }
                    __syncthreads();
                }
                if (lane == 0) {
                    if (s_pairing) d_remove_unmated(a1, n1, a2, n2);
                    d_remove_redundant(a1, n1); d_remove_redundant(a2, n2);
                }
                __syncthreads();
                for (int i = lane; i < n1; i += 64) c1[i] = lc[0][i];
                for (int i = lane; i < n2; i += 64) c2[i] = lc[1][i];
            } else if (lane == 0) {
                if (d_check_paired(c1, n1, c2, n2)) d_remove_unmated(c1, n1, c2, n2);
                d_remove_redundant(c1, n1); d_remove_redundant(c2, n2);
            }
        } else if (lane == 0) d_remove_redundant(c1, n1);
        __syncthreads();
        if (lane == 0) {
            for (int m = 0; m < nm; m++) {
                const DCand *c = cands + b[m];
                const int nc = m == 0 ? n1 : n2;
                uint32_t w = 0;
                for (int i = 0; i < nc; i++) if (c[i].Score > 0) w += d_work_need(c[i].count);
                ncand[r1 + m] = (uint32_t)nc; nrep[r1 + m] = nc > 0 ? (uint32_t)nc : 1u; work_need[r1 + m] = w;
                nc_total += (unsigned long long)nc;
            }
#ifdef DG_PROFILE_CLASSES
            const unsigned long long cyc = (unsigned long long)(clock64() - t_begin);
            auto cl = [](unsigned long long v) { return v > 1023 ? 1023ull : v; };
            atomicMax(&g_chain_max, (cyc << 40) | (cl(n[0]) << 30) | (cl(paired ? n[1] : 0) << 20) | (cl(n1) << 10) | cl(n2));
            atomicAdd(&g_chain_units, 1ull); atomicAdd(&g_chain_cycles, cyc);
#endif
        }
    }
    if (lane == 0 && nc_total) atomicAdd(d_ctr_stripe(ctr) + CTR_CANDS, nc_total);
}

// list of heavy units (order irrelevant: every unit writes only its own slots)
__global__ void __launch_bounds__(256)
k_heavy_list(int n_units, int paired, const uint32_t *__restrict__ seed_off, uint32_t *__restrict__ heavy_list, unsigned int *n_heavy)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n_units && d_unit_is_heavy(seed_off, paired, u)) heavy_list[atomicAdd(n_heavy, 1u)] = (uint32_t)u;
}
