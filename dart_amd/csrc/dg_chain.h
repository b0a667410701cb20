// dart_amd/csrc/dg_chain.h -- seeds -> alignment candidates, mate pairing, redundancy filter: the pieces shared by
// k_pair (dg_pair.h: one lane per unit, everything in LDS) and k_chain_heavy (here: one wave per unit with many seeds).
//
// What the reference does at this stage: the tail of IdentifySeedPairs (sort by (gPos,rPos),
// AlignmentCandidates.cpp:212), GenerateAlignmentCandidate (:241-288), CheckPairedAlignmentCandidates (Mapping.cpp:403-450),
// RemoveUnMatedAlignmentCandidates (:452-477), RemoveRedundantCandidates (:371-401).
//
// Seeds are SKey values (dg_common.h): sorting the 64-bit words is the reference's order.  A candidate is a run of the sorted
// seed segment.  The three candidate-list rules are written once, against a small "view" interface
//   n(), score(i), set_score(i, v), diag(i) [PosDiff], mate(i) [-1 = none], set_mate(i, v)
// with one view over DCand arrays in memory (CandMem, below) and one over k_pair's packed LDS words (dg_pair.h).
#pragma once
#include "dg_common.h"

// ---------------------------------------------------------------------------------------------
// sorting SKey segments in memory (insertion for short lists, heap sort beyond)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline void d_sort_keys(SKey *a, int n)
{
    if (n <= 24) {
        for (int i = 1; i < n; i++) {
            const SKey x = a[i];
            int j = i;
            for (; j > 0 && a[j - 1] > x; j--) a[j] = a[j - 1];
            a[j] = x;
        }
        return;
    }
    auto sift = [&](SKey x, int root, int end) {
        for (int ch = 2 * root + 1; ch < end; ch = 2 * root + 1) {
            if (ch + 1 < end && a[ch] < a[ch + 1]) ch++;
            if (!(x < a[ch])) break;
            a[root] = a[ch]; root = ch;
        }
        a[root] = x;
    };
    for (int st = n / 2 - 1; st >= 0; st--) sift(a[st], st, n);
    for (int end = n - 1; end > 0; end--) { const SKey x = a[end]; a[end] = a[0]; sift(x, 0, end); }
}

// ---------------------------------------------------------------------------------------------
// Clustering of a sorted seed segment into candidates (GenerateAlignmentCandidate :241-288).
// The walk: seeds whose diagonal (gPos - rPos) is negative are skipped at the front; a cluster grows by the next seed while
// the diagonals of the two neighbours differ by less than MaxGaps, or by less than MaxIntronSize when the next seed still lies
// before the end of the chromosome half that holds the cluster's last seed (the smallest ChrLocMap key >= its gPos) and
// further along the read; a cluster is a candidate when its seeds cover more than 30 % of the read.
// key(i) -> SKey of seed i; emit(first, count, score, PosDiff) is called per candidate, in order.  Returns their number.
// ---------------------------------------------------------------------------------------------
template <class KeyAt, class Emit>
__host__ __device__ inline int d_cluster_seeds(const LocTab &lt, const DParams &pr, int rlen, int num, KeyAt key, Emit emit)
{
    const int need = (int)(rlen * 0.3);
    int made = 0, head = 0;
    while (head < num && sk_diag(key(head)) < 0) head++;
    while (head < num) {
        SKey tail = key(head);
        const int64_t d_head = sk_diag(tail);
        int covered = sk_rlen(tail), next = head + 1;
        for (; next < num; next++) {
            const SKey cand = key(next);
            int64_t jump = sk_diag(cand) - sk_diag(tail);
            if (jump < 0) jump = -jump;
            bool joins = jump < pr.max_gaps;
            if (!joins && jump < pr.max_intron)
                joins = sk_gpos(cand) < lt.key[d_loc_lower_bound(lt, sk_gpos(tail))] && sk_rpos(cand) > sk_rpos(tail);
            if (!joins) break;
            covered += sk_rlen(cand);
            tail = cand;
        }
        if (covered > need) { emit(head, next - head, covered, d_head < 0 ? (int64_t)0 : d_head); made++; }
        head = next;
    }
    return made;
}
template <class KeyAt, class Emit>
__host__ __device__ inline int d_cluster_seeds(const DIndex &ix, const DParams &pr, int rlen, int num, KeyAt key, Emit emit)
{
    return d_cluster_seeds(d_loc_tab(ix), pr, rlen, num, key, emit);
}

// ---------------------------------------------------------------------------------------------
// The three list rules, over a view
// ---------------------------------------------------------------------------------------------
// RemoveRedundantCandidates (Mapping.cpp:371-401).  The scores are folded into (top, runner): a score above the runner
// replaces it -- and pushes the old top down when it is at least the top --, a score EQUAL to the runner lifts the runner to
// the top (the reference's tie rule: a second-best that occurs twice counts as a tie for first).  Candidates below the
// cut are dropped (score 0); the cut is the top when top == runner or when they are more than 20 apart, else the runner.
__host__ __device__ __forceinline__ void d_top2_push(int &top, int &runner, int sc)
{
    if (sc == runner) runner = top;
    else if (sc > runner) { const int t = top; top = t > sc ? t : sc; runner = t < sc ? t : sc; }
}
template <class V>
__host__ __device__ inline void d_keep_top(V &v)
{
    const int n = v.n();
    if (n < 2) return;
    int top = 0, runner = 0;
    for (int i = 0; i < n; i++) d_top2_push(top, runner, v.score(i));
    const int cut = (top == runner || top - runner > 20) ? top : runner;
    for (int i = 0; i < n; i++) if (v.score(i) < cut) v.set_score(i, 0);
}

// CheckPairedAlignmentCandidates (Mapping.cpp:403-450): every live candidate of mate 1, in order, looks for the live candidate
// of mate 2 that lies downstream of it (PosDiff not smaller) at the smallest distance below 2 000 000 (the first one wins a
// tie).  A free partner is taken; a taken partner changes hands when the newcomer scores higher than its current holder.
// Returns whether any pair was formed.
template <class V1, class V2>
__host__ __device__ inline bool d_pair_mates(V1 &a, V2 &b)
{
    const int na = a.n(), nb = b.n();
    if (na * nb > 1000) { d_keep_top(a); d_keep_top(b); }
    bool formed = false;
    for (int i = 0; i < na; i++) {
        const int mine = a.score(i);
        if (mine == 0) continue;
        const int64_t here = a.diag(i);
        int pick = -1;
        int64_t reach = 2000000;
        for (int j = 0; j < nb; j++) {
            if (b.score(j) == 0) continue;
            const int64_t ahead = b.diag(j) - here;
            if (ahead >= 0 && ahead < reach) { reach = ahead; pick = j; }
        }
        if (pick < 0) continue;
        const int holder = b.mate(pick);
        if (holder < 0) formed = true;
        else if (mine > a.score(holder)) a.set_mate(holder, -1);
        else continue;
        a.set_mate(i, pick); b.set_mate(pick, i);
    }
    return formed;
}

// RemoveUnMatedAlignmentCandidates (Mapping.cpp:452-477): partners both get the sum of their scores, singles get 0
template <class V1, class V2>
__host__ __device__ inline void d_settle_mates(V1 &a, V2 &b)
{
    for (int i = 0, na = a.n(); i < na; i++) {
        const int m = a.mate(i);
        if (m < 0) { a.set_score(i, 0); continue; }
        const int sum = a.score(i) + b.score(m);
        a.set_score(i, sum); b.set_score(m, sum);
    }
    for (int j = 0, nb = b.n(); j < nb; j++) if (b.mate(j) < 0) b.set_score(j, 0);
}

// the candidate stage of one unit after clustering: pairing (paired-end), then the redundancy filter per mate
template <class V1, class V2>
__host__ __device__ inline void d_candidate_rules(bool paired, V1 &a, V2 &b)
{
    if (paired) {
        if (d_pair_mates(a, b)) d_settle_mates(a, b);
        d_keep_top(a); d_keep_top(b);
    } else d_keep_top(a);
}

// view over DCand records in memory (global or LDS)
struct CandMem {
    DCand *c; int cnt;
    __host__ __device__ int n() const { return cnt; }
    __host__ __device__ int score(int i) const { return c[i].Score; }
    __host__ __device__ void set_score(int i, int v) { c[i].Score = v; }
    __host__ __device__ int64_t diag(int i) const { return c[i].PosDiff; }
    __host__ __device__ int mate(int i) const { return c[i].PairedIdx; }
    __host__ __device__ void set_mate(int i, int v) { c[i].PairedIdx = v; }
};

__host__ __device__ __forceinline__ DCand d_new_cand(uint32_t first_abs, int count, int score, int64_t pos_diff)
{
    DCand c;
    c.PosDiff = pos_diff; c.first = (int32_t)first_abs; c.count = count; c.Score = score; c.PairedIdx = -1; c.SJtype = -1;
    c.work_off = 0; c.final_n = 0; c.n_a = 0; c.job_first = 0; c.job_count = 0;
    return c;
}

// clustering of a segment that lies in memory, candidates appended to out[]
__host__ __device__ inline int d_gen_candidates(const DIndex &ix, const DParams &pr, int rlen, const SKey *s, int num, uint32_t base, DCand *out)
{
    int k = 0;
    return d_cluster_seeds(ix, pr, rlen, num, [&](int i) { return s[i]; },
                           [&](int first, int count, int score, int64_t pd) { out[k++] = d_new_cand(base + (uint32_t)first, count, score, pd); });
}

// working-region size of one candidate in the report stage: tandem/translocation clean-up never grows the list; re-seeding
// adds <= n-1, gap filling <= 2 per adjacent pair, normal pairs <= 1 per adjacent pair, plus merge slack -> 14n+8 seeds
__host__ __device__ __forceinline__ uint32_t d_work_need(int count) { return 14u * (uint32_t)count + 8u; }

// A unit (pair / single read) goes to k_chain_heavy when its seeds do not fit k_pair's per-lane LDS slice: reads from repeat
// families.  They are a small share of the input but their lists are 10-100x longer; a wave each.
#ifndef UNIT_MAX_SEEDS
#define UNIT_MAX_SEEDS 16
#endif
__device__ __forceinline__ bool d_unit_is_heavy(const uint32_t *seed_off, int paired, int u)
{
    const int r1 = paired ? 2 * u : u;
    return seed_off[r1 + (paired ? 2 : 1)] - seed_off[r1] > UNIT_MAX_SEEDS;
}

// ---------------------------------------------------------------------------------------------
// k_chain_heavy: one wave = one heavy unit.  Seeds of both mates are staged in LDS and sorted with a wave-wide bitonic
// network; candidates are built by the whole wave from LDS (d_gen_candidates_wave); the O(n1*n2) mate search runs 64 candidates at a time with a wave
// min-reduction that keeps the first-minimum rule.  Anything larger than the LDS staging (> CH_MAXS seeds or > CH_MAXC
// candidates per mate) takes the serial route through global memory (same functions).
// ---------------------------------------------------------------------------------------------
#define CH_MAXS 1024
#define CH_MAXC 192
#define CH_LOC_MAX 64         // ChrLocMap keys k_chain_heavy keeps in LDS

// d_gen_candidates by one wave over sorted keys in LDS (num <= CH_MAXS).  Whether seed i starts a new candidate depends on its
// predecessor only (the clustering walk compares every seed with the one it has just taken), so the starts, the covered-bases prefix
// and every candidate's end come from ballots and one running sum; candidates are numbered in seed order as the serial walk does.
// scratch: pref[CH_MAXS + 1] u32 and starts[CH_MAXS / 64] u64 (LDS).  Every lane returns the count.
__device__ inline int d_gen_candidates_wave(const LocTab &lt, const DParams &pr, int rlen, const SKey *s, int num, uint32_t base, DCand *out,
                                            uint32_t *pref, unsigned long long *starts, int lane)
{
    const int need = (int)(rlen * 0.3), rounds = (num + 63) >> 6;
    // the walk skips the leading seeds on negative diagonals
    int head0 = num;
    for (int r = 0; r < rounds && head0 == num; r++) {
        const int i = r * 64 + lane;
        const unsigned long long m = __ballot(i < num && sk_diag(s[i]) >= 0);
        if (m) head0 = r * 64 + __ffsll((long long)m) - 1;
    }
    uint32_t run = 0;
    if (lane == 0) pref[0] = 0;
    for (int r = 0; r < rounds; r++) {
        const int i = r * 64 + lane;
        bool start = false;
        uint32_t len = 0;
        if (i < num && i >= head0) {
            const SKey cand = s[i];
            len = (uint32_t)sk_rlen(cand);
            if (i == head0) start = true;
            else {
                const SKey tail = s[i - 1];
                int64_t jump = sk_diag(cand) - sk_diag(tail);
                if (jump < 0) jump = -jump;
                bool joins = jump < pr.max_gaps;
                if (!joins && jump < pr.max_intron)
                    joins = sk_gpos(cand) < lt.key[d_loc_lower_bound(lt, sk_gpos(tail))] && sk_rpos(cand) > sk_rpos(tail);
                start = !joins;
            }
        }
        const unsigned long long m = __ballot(start);
        if (lane == 0) starts[r] = m;
        uint32_t incl = len;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += v; }
        if (i < num) pref[i + 1] = run + incl;
        run += (uint32_t)__shfl((int)incl, 63, 64);
    }
    __syncthreads();
    int made = 0;
    for (int r = 0; r < rounds; r++) {
        const int i = r * 64 + lane;
        const unsigned long long sm = starts[r];
        const bool start = (sm >> lane) & 1ull;
        int end = num;
        if (start) {
            const unsigned long long later = lane == 63 ? 0ull : sm >> (lane + 1);
            if (later) end = i + 1 + (__ffsll((long long)later) - 1);
            else for (int q = r + 1; q < rounds; q++) if (starts[q]) { end = q * 64 + __ffsll((long long)starts[q]) - 1; break; }
        }
        const int covered = start ? (int)(pref[end] - pref[i]) : 0;
        const bool emit = start && covered > need;
        const unsigned long long em = __ballot(emit);
        if (emit) {
            const int64_t d = sk_diag(s[i]);
            out[made + __popcll(em & ((1ull << lane) - 1ull))] = d_new_cand(base + (uint32_t)i, end - i, covered, d < 0 ? (int64_t)0 : d);
        }
        made += __popcll(em);
    }
    __syncthreads();
    return made;
}

__device__ inline void d_bitonic_sort_keys(SKey *a, int n, int lane)   // n <= CH_MAXS, one wave
{
    int m = 1; while (m < n) m <<= 1;
    for (int i = n + lane; i < m; i += 64) a[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < m / 2; t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const bool up = (i & k) == 0;
                const SKey x = a[i], y = a[l];
                if ((y < x) == up) { a[i] = y; a[l] = x; }
            }
            __syncthreads();
        }
    }
}

#ifdef DG_PROFILE_CLASSES
__device__ unsigned long long g_chain_max, g_chain_units, g_chain_cycles;
#endif
// what the candidate rules read and write of a DCand, staged in LDS (16 of its 48 bytes)
struct __attribute__((aligned(8))) CandLite { int64_t PosDiff; int32_t Score, PairedIdx; };
struct CandLiteView {
    CandLite *c; int cnt;
    __host__ __device__ int n() const { return cnt; }
    __host__ __device__ int score(int i) const { return c[i].Score; }
    __host__ __device__ void set_score(int i, int v) { c[i].Score = v; }
    __host__ __device__ int64_t diag(int i) const { return c[i].PosDiff; }
    __host__ __device__ int mate(int i) const { return c[i].PairedIdx; }
    __host__ __device__ void set_mate(int i, int v) { c[i].PairedIdx = v; }
};
// LDS of one wave, 12.4 KB (round 4: 39 KB -- both mates' seeds, whole DCand records and the clustering scratch side by side; a wave of this kernel held a
// quarter of a CU's LDS, 17 % of the GPU's LDS-time on a genome with human-like repeat content): the sort + clustering of one mate at a time needs
// the seeds, the covered-bases prefix and the start masks; the pairing afterwards needs the two candidate lists and the picks -- the two phases share the bytes.
#define CH_LDS_SORT (CH_MAXS * 8 + (CH_MAXS + 1) * 4 + 4 + (CH_MAXS / 64) * 8)
#define CH_LDS_PAIR (2 * CH_MAXC * 16 + CH_MAXC * 2)
#define CH_LDS_BYTES (CH_LDS_SORT > CH_LDS_PAIR ? CH_LDS_SORT : CH_LDS_PAIR)
__global__ void __launch_bounds__(64)
k_chain_heavy(const DIndex ix, const DParams pr, int n_units, int paired, const uint16_t *__restrict__ rlen,
              const uint32_t *__restrict__ seed_off, SKey *__restrict__ seeds, DCand *__restrict__ cands,
              uint32_t *__restrict__ ncand, const uint32_t *__restrict__ heavy_list, const unsigned int *__restrict__ n_heavy_p,
              int16_t *__restrict__ picks, uint8_t *__restrict__ todo, unsigned long long *ctr, const int *__restrict__ abort_p)
{
    const unsigned long long t_wave0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[CH_LDS_BYTES];
    SKey *ls = (SKey *)s_mem;                                                       // phase 1: one mate's seeds ...
    uint32_t *s_pref = (uint32_t *)(s_mem + CH_MAXS * 8);                            // ... the prefix of covered bases ...
    unsigned long long *s_starts = (unsigned long long *)(s_mem + CH_MAXS * 8 + (CH_MAXS + 1) * 4 + 4);      // ... and the start masks
    CandLite *lc0 = (CandLite *)s_mem, *lc1 = lc0 + CH_MAXC;                          // phase 2: the two candidate lists ...
    int16_t *s_pick = (int16_t *)(s_mem + 2 * CH_MAXC * 16);                          // ... and every candidate's pick
    __shared__ int s_n[2];
    // the ChrLocMap keys in LDS when they fit (32 chromosomes: 512 bytes): the clustering asks "does this seed lie
    // before the end of its neighbour's chromosome half" through a binary search, six dependent loads from memory otherwise (dg_common.h, LocTab)
    __shared__ int64_t s_lkey[CH_LOC_MAX];
    const bool tab_in_lds = 2 * ix.n_chr <= CH_LOC_MAX;
    if (tab_in_lds) for (int i = threadIdx.x; i < 2 * ix.n_chr; i += 64) s_lkey[i] = ix.loc_key[i];
    __syncthreads();
    const LocTab lt = tab_in_lds ? LocTab{s_lkey, ix.loc_chr, ix.chr_off, 2 * ix.n_chr} : d_loc_tab(ix);
    if (*abort_p >= DG_ABORT) return;
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
        uint32_t b[2] = {0, 0}, n[2] = {0, 0};
        for (int m = 0; m < nm; m++) { b[m] = seed_off[r1 + m]; n[m] = seed_off[r1 + m + 1] - b[m]; }
        __syncthreads();
        for (int m = 0; m < nm; m++) {                       // one mate at a time: sort, write back, cluster
            if (n[m] <= 64) {
                // at most one key per lane (most heavy units: 17 .. 64 seeds per mate): the bitonic network in registers, partners by lane shuffles -- a third of the
                // instructions of the LDS form below (no index arithmetic, no LDS round trips, no barriers between the steps)
                SKey key = (uint32_t)lane < n[m] ? seeds[b[m] + lane] : ~0ull;
                int mm = 2; while ((uint32_t)mm < n[m]) mm <<= 1;
                for (int k = 2; k <= mm; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        const SKey other = d_u64((uint32_t)__shfl_xor((int)(uint32_t)key, j, 64), (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), j, 64));
                        const bool keep_small = ((lane & k) == 0) == ((lane & j) == 0);
                        key = (other < key) == keep_small ? other : key;
                    }
                if ((uint32_t)lane < n[m]) { ls[lane] = key; seeds[b[m] + lane] = key; }
                __syncthreads();
                const int made = d_gen_candidates_wave(lt, pr, rlen[r1 + m], ls, (int)n[m], b[m], cands + b[m], s_pref, s_starts, lane);
                if (lane == 0) s_n[m] = made;
            } else if (n[m] <= CH_MAXS) {
                for (uint32_t i = lane; i < n[m]; i += 64) ls[i] = seeds[b[m] + i];
                __syncthreads();
                d_bitonic_sort_keys(ls, (int)n[m], lane);
                for (uint32_t i = lane; i < n[m]; i += 64) seeds[b[m] + i] = ls[i];
                __syncthreads();
                const int made = d_gen_candidates_wave(lt, pr, rlen[r1 + m], ls, (int)n[m], b[m], cands + b[m], s_pref, s_starts, lane);
                if (lane == 0) s_n[m] = made;
            } else if (lane == 0) {                          // longer than the staging: the serial route through memory (same functions)
                d_sort_keys(seeds + b[m], (int)n[m]);
                s_n[m] = d_gen_candidates(ix, pr, rlen[r1 + m], seeds + b[m], (int)n[m], b[m], cands + b[m]);
            }
            __syncthreads();
        }
        const int n1 = s_n[0], n2 = paired ? s_n[1] : 0;
        DCand *c1 = cands + b[0], *c2 = paired ? cands + b[1] : nullptr;
        __syncthreads();                                       // (the candidates are in memory; the sort phase's LDS is free)
        if (paired) {
            const bool stage = n1 <= CH_MAXC && n2 <= CH_MAXC;
            if (stage) {
                for (int i = lane; i < n1; i += 64) { const DCand &c = c1[i]; lc0[i] = CandLite{c.PosDiff, c.Score, c.PairedIdx}; }
                for (int i = lane; i < n2; i += 64) { const DCand &c = c2[i]; lc1[i] = CandLite{c.PosDiff, c.Score, c.PairedIdx}; }
                __syncthreads();
                CandLiteView a1{lc0, n1}, a2{lc1, n2};
                if (n1 * n2 > 1000) { if (lane == 0) { d_keep_top(a1); d_keep_top(a2); } __syncthreads(); }
                // d_pair_mates in two steps: which partner a candidate would pick depends only on scores and PosDiffs, which the
                // pairing does not change -- so all picks are made first, 64 candidates of mate 1 at a time (every lane scans mate 2's
                // list: broadcast LDS reads), and lane 0 then replays the take / change-hands decisions in candidate order
                for (int i = lane; i < n1; i += 64) {
                    int pick = -1;
                    if (a1.score(i) != 0) {
                        const int64_t here = a1.diag(i);
                        int64_t reach = 2000000;
                        for (int j = 0; j < n2; j++) {
                            if (a2.score(j) == 0) continue;
                            const int64_t ahead = a2.diag(j) - here;
                            if (ahead >= 0 && ahead < reach) { reach = ahead; pick = j; }
                        }
                    }
                    s_pick[i] = (int16_t)pick;
                }
                __syncthreads();
                // The rest of the candidate rules is sequential in candidate order (who takes or loses a partner, RemoveUnMatedAlignmentCandidates, the
                // top / runner fold of RemoveRedundantCandidates): one lane's work.  Rounds 2-4 let lane 0 of this wave do it -- ~1000 wave-instructions per
                // unit, 40 % of this kernel's -- ; now the picks go to memory and k_chain_rules finishes 64 units per wave, lane = unit.
                for (int i = lane; i < n1; i += 64) picks[b[0] + i] = s_pick[i];
                if (n1 * n2 > 1000) {                         // (the scores the early redundancy filter zeroed)
                    for (int i = lane; i < n1; i += 64) c1[i].Score = lc0[i].Score;
                    for (int i = lane; i < n2; i += 64) c2[i].Score = lc1[i].Score;
                }
                if (lane == 0) todo[hi] = 1;
            } else if (lane == 0) { CandMem a1{c1, n1}, a2{c2, n2}; d_candidate_rules(true, a1, a2); todo[hi] = 0; }      // (lists longer than the staging: everything here, by one lane)
        } else if (lane == 0) todo[hi] = 2;                          // single-end: the redundancy filter alone (k_chain_rules)
        __syncthreads();
        if (lane == 0) {
            ncand[r1] = (uint32_t)n1; nc_total += (unsigned long long)n1;
            if (paired) { ncand[r1 + 1] = (uint32_t)n2; nc_total += (unsigned long long)n2; }
#ifdef DG_PROFILE_CLASSES
            const unsigned long long cyc = (unsigned long long)(clock64() - t_begin);
            auto cl = [](unsigned long long v) { return v > 1023 ? 1023ull : v; };
            atomicMax(&g_chain_max, (cyc << 40) | (cl(n[0]) << 30) | (cl(paired ? n[1] : 0) << 20) | (cl(n1) << 10) | cl(n2));
            atomicAdd(&g_chain_units, 1ull); atomicAdd(&g_chain_cycles, cyc);
#endif
        }
    }
    if (lane == 0 && nc_total) atomicAdd(d_ctr_stripe(ctr) + CTR_CANDS, nc_total);
    d_wave_resident(ctr, CTR_WT_CHAIN, t_wave0);
}

// ---------------------------------------------------------------------------------------------
// k_chain_rules: lane = one unit of k_chain_heavy's list -- the sequential part of the candidate stage (CheckPairedAlignmentCandidates' take / change-hands
// decisions over the picks k_chain_heavy made, Mapping.cpp:403-450; RemoveUnMatedAlignmentCandidates :452-477; RemoveRedundantCandidates :371-401) on the
// DCand records in memory.  One lane's work per unit either way; here 64 units share an instruction.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_chain_rules(int paired, const uint32_t *__restrict__ seed_off, DCand *__restrict__ cands, const uint32_t *__restrict__ ncand, const uint32_t *__restrict__ heavy_list,
              const unsigned int *__restrict__ n_heavy_p, const int16_t *__restrict__ picks, const uint8_t *__restrict__ todo, const int *__restrict__ abort_p)
{
    if (*abort_p >= DG_ABORT) return;
    const unsigned int n_heavy = *n_heavy_p;
    for (unsigned int hi = blockIdx.x * blockDim.x + threadIdx.x; hi < n_heavy; hi += gridDim.x * blockDim.x) {
        const int what = todo[hi];
        if (what == 0) continue;
        const int u = (int)heavy_list[hi], r1 = paired ? 2 * u : u;
        const uint32_t b0 = seed_off[r1];
        CandMem a1{cands + b0, (int)ncand[r1]};
        if (what == 2) { d_keep_top(a1); continue; }
        const uint32_t b1 = seed_off[r1 + 1];
        CandMem a2{cands + b1, (int)ncand[r1 + 1]};
        bool pairing = false;
        for (int i = 0; i < a1.cnt; i++) {
            const int pick = picks[b0 + i];
            if (pick < 0) continue;
            const int holder = a2.mate(pick);
            if (holder < 0) pairing = true;
            else if (a1.score(i) > a1.score(holder)) a1.set_mate(holder, -1);
            else continue;
            a1.set_mate(i, pick); a2.set_mate(pick, i);
        }
        if (pairing) d_settle_mates(a1, a2);
        d_keep_top(a1); d_keep_top(a2);
    }
}

