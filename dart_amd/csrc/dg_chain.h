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
__device__ inline void d_sort_seeds(DSeed *a, int n)
{
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
__device__ inline int d_gen_candidates(const DIndex &ix, const DParams &pr, int rlen, const DSeed *s, int num, uint32_t base, DCand *out)
{
    int nc = 0;
    if (num == 0) return 0;
    const int thr = (int)(rlen * 0.3);
    int i = 0;
    while (i < num && s[i].gPos - s[i].rPos < 0) i++;
    while (i < num) {
        int score = s[i].rLen, j = i, k;
        for (k = i + 1; k < num; k++) {
            int64_t pd = (s[k].gPos - s[k].rPos) - (s[j].gPos - s[j].rPos);
            if (pd < 0) pd = -pd;
            bool ok = pd < pr.max_gaps;
            if (!ok && pd < pr.max_intron) {
                const int lb = d_loc_lower_bound(ix, s[j].gPos);
                ok = s[k].gPos < ix.loc_key[lb] && s[k].rPos > s[j].rPos;
            }
            if (!ok) break;
            score += s[k].rLen;
            j = k;
        }
        if (score > thr) {
            DCand c;
            const int64_t pd0 = s[i].gPos - s[i].rPos;
            c.PosDiff = pd0 < 0 ? 0 : pd0;
            c.first = (int32_t)(base + i); c.count = k - i; c.Score = score; c.PairedIdx = -1; c.SJtype = -1;
            c.work_off = 0; c.final_n = 0; c.n_a = 0; c.job_first = 0; c.job_count = 0;
            out[nc++] = c;
        }
        i = k;
    }
    return nc;
}

__device__ inline void d_remove_redundant(DCand *c, int n)   // Mapping.cpp:371-401
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

__device__ inline bool d_check_paired(DCand *c1, int n1, DCand *c2, int n2)   // Mapping.cpp:403-450
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

__device__ inline void d_remove_unmated(DCand *c1, int n1, DCand *c2, int n2)   // Mapping.cpp:452-477
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

__global__ void __launch_bounds__(256)
k_chain(const DIndex ix, const DParams pr, int n_units, int paired, const uint16_t *__restrict__ rlen,
        const uint32_t *__restrict__ seed_off, DSeed *__restrict__ seeds, DCand *__restrict__ cands,
        uint32_t *__restrict__ ncand, uint32_t *__restrict__ nrep, uint32_t *__restrict__ work_need, unsigned long long *ctr)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nc_total = 0;
    if (u < n_units) {
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
