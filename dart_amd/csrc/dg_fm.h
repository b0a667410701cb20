// dart_amd/csrc/dg_fm.h -- FM-index kernels: maximal-exact-match search and SA locate.
//
// Replaces bwt_search.cpp:43-182 (bwt_occ, bwt_occ4, bwt_2occ4, bwt_invPsi, bwt_sa, BWT_Search)
// and the search loop of IdentifySeedPairs (AlignmentCandidates.cpp:181-215).
//
// Bound: HBM / Infinity-Cache random 64-byte reads.  One Occ query = one aligned 64-byte block
// (4 x global_load_dwordx4 from one lane, one cache line); the counting is ~100 VALU ops of
// popcounts, far below the memory time.  One lane owns one read and walks its dependent chain of
// ~120 queries; throughput comes from the >= 10^5 chains in flight (8 waves/SIMD).
#pragma once
#include "dg_common.h"

struct OccBlock { uint4 q0, q1, q2, q3; };

__device__ __forceinline__ OccBlock d_load_block(const DIndex &ix, uint64_t blk)
{
    const uint4 *p = ix.bwt + (blk << 2);
    OccBlock b;
    b.q0 = p[0]; b.q1 = p[1]; b.q2 = p[2]; b.q3 = p[3];
    return b;
}

__device__ __forceinline__ uint64_t d_sel4(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3, int i)
{
    uint64_t lo = (i & 1) ? a1 : a0, hi = (i & 1) ? a3 : a2;
    return (i & 2) ? hi : lo;
}
__device__ __forceinline__ uint64_t d_L2(const DIndex &ix, int i)   // i in 0..4
{
    return i == 4 ? ix.L2[4] : d_sel4(ix.L2[0], ix.L2[1], ix.L2[2], ix.L2[3], i);
}

// number of symbols equal to b among the first n (0..32) symbols of a 64-bit word holding 32
// 2-bit symbols, first symbol in the top bits (two consecutive u32 of the .bwt, MSB first)
__device__ __forceinline__ uint32_t d_cnt_word(uint64_t w, uint64_t rep, uint32_t n)
{
    uint64_t t = ~(w ^ rep);
    t = t & (t >> 1) & 0x5555555555555555ull;
    uint64_t m = n >= 32 ? ~0ull : ~(~0ull >> (2 * n));   // n == 0 -> 0
    return (uint32_t)__popcll(t & m);
}

// Occ(b, row) for the four bases, rows counted inclusively up to in-block offset o (bwt_occ4 :67-84)
__device__ __forceinline__ void d_occ4(const OccBlock &B, uint32_t o, uint64_t cnt[4])
{
    const uint64_t w0 = ((uint64_t)B.q2.x << 32) | B.q2.y, w1 = ((uint64_t)B.q2.z << 32) | B.q2.w;
    const uint64_t w2 = ((uint64_t)B.q3.x << 32) | B.q3.y, w3 = ((uint64_t)B.q3.z << 32) | B.q3.w;
    const uint32_t n = o + 1;                       // symbols included, 1..128
    const uint32_t n0 = n > 32 ? 32 : n;
    const uint32_t n1 = n > 64 ? 32 : (n > 32 ? n - 32 : 0);
    const uint32_t n2 = n > 96 ? 32 : (n > 64 ? n - 64 : 0);
    const uint32_t n3 = n > 96 ? n - 96 : 0;
    uint32_t c1 = d_cnt_word(w0, 0x5555555555555555ull, n0) + d_cnt_word(w1, 0x5555555555555555ull, n1)
                + d_cnt_word(w2, 0x5555555555555555ull, n2) + d_cnt_word(w3, 0x5555555555555555ull, n3);
    uint32_t c2 = d_cnt_word(w0, 0xAAAAAAAAAAAAAAAAull, n0) + d_cnt_word(w1, 0xAAAAAAAAAAAAAAAAull, n1)
                + d_cnt_word(w2, 0xAAAAAAAAAAAAAAAAull, n2) + d_cnt_word(w3, 0xAAAAAAAAAAAAAAAAull, n3);
    uint32_t c3 = d_cnt_word(w0, ~0ull, n0) + d_cnt_word(w1, ~0ull, n1) + d_cnt_word(w2, ~0ull, n2) + d_cnt_word(w3, ~0ull, n3);
    uint32_t c0 = n - c1 - c2 - c3;
    cnt[0] = (((uint64_t)B.q0.y << 32) | B.q0.x) + c0;
    cnt[1] = (((uint64_t)B.q0.w << 32) | B.q0.z) + c1;
    cnt[2] = (((uint64_t)B.q1.y << 32) | B.q1.x) + c2;
    cnt[3] = (((uint64_t)B.q1.w << 32) | B.q1.z) + c3;
}

// LF mapping, bwt_invPsi :119-125 (one block: the symbol and its Occ come from the same row)
__device__ __forceinline__ uint64_t d_lf(const DIndex &ix, uint64_t k)
{
    if (k == ix.primary) return 0;
    const uint64_t x = k - (k > ix.primary);
    const OccBlock B = d_load_block(ix, x >> 7);
    const uint32_t o = (uint32_t)(x & 127);
    const uint32_t wi = o >> 4;
    const uint32_t w = wi < 4 ? (wi == 0 ? B.q2.x : wi == 1 ? B.q2.y : wi == 2 ? B.q2.z : B.q2.w)
                              : (wi == 4 ? B.q3.x : wi == 5 ? B.q3.y : wi == 6 ? B.q3.z : B.q3.w);
    const int c = (int)((w >> ((~o & 15) << 1)) & 3);
    uint64_t cnt[4];
    d_occ4(B, o, cnt);
    return d_L2(ix, c) + d_sel4(cnt[0], cnt[1], cnt[2], cnt[3], c);
}

// ---------------------------------------------------------------------------------------------
// k_seed: one lane = one read.  Greedy left-to-right tiling with maximal exact matches
// (IdentifySeedPairs + BWT_Search).  The two nested loops of the reference are flattened into one
// loop whose every trip is exactly one bi-interval extension, so the lanes of a wave stay
// converged on the memory-bound step whatever their read positions are.
// Output: per read up to H intervals (a hit is >= 16 long, so H = max_rlen/16 + 1 always fits).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_seed(const DIndex ix, const DParams pr, const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off,
       const uint16_t *__restrict__ rlen, int n_reads, int H, DHit *__restrict__ hits, uint32_t *__restrict__ nhits,
       uint32_t *__restrict__ nseeds, unsigned long long *ctr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long steps = 0, blocks = 0;
    if (r < n_reads) {
        const unsigned char *s = seq + seq_off[r];
        const int len = rlen[r], end_pos = len - 13;
        int pos = 0, start = 0, p = 0, nh = 0;
        uint32_t ns = 0;
        bool searching = false;
        uint64_t x0 = 0, x1 = 0, x2 = 0;
        while (true) {
            if (!searching) {
                while (pos < end_pos && d_nt4(s[pos]) > 3) pos++;
                if (pos >= end_pos) break;
                const int c = d_nt4(s[pos]);
                start = pos; p = pos + 1; searching = true;
                x0 = d_L2(ix, c) + 1; x1 = d_L2(ix, 3 - c) + 1; x2 = d_L2(ix, c + 1) - d_L2(ix, c);
            }
            bool stop = p >= len;
            int c = 4;
            if (!stop) { c = d_nt4(s[p]); stop = c > 3; }
            if (!stop) {
                const uint64_t k = x1 - 1, l = k + x2;
                const uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
                uint64_t tk[4], tl[4];
                const OccBlock B = d_load_block(ix, kk >> 7);
                d_occ4(B, (uint32_t)(kk & 127), tk);
                if ((ll >> 7) != (kk >> 7)) {
                    const OccBlock B2 = d_load_block(ix, ll >> 7);
                    d_occ4(B2, (uint32_t)(ll & 127), tl);
                    blocks += 2;
                } else {
                    d_occ4(B, (uint32_t)(ll & 127), tl);
                    blocks += 1;
                }
                steps++;
                const int b = 3 - c;
                const uint64_t n2 = d_sel4(tl[0], tl[1], tl[2], tl[3], b) - d_sel4(tk[0], tk[1], tk[2], tk[3], b);
                if (n2 == 0) stop = true;
                else {
                    uint64_t nx0 = x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0);
                    if (b <= 2) nx0 += tl[3] - tk[3];
                    if (b <= 1) nx0 += tl[2] - tk[2];
                    if (b == 0) nx0 += tl[1] - tk[1];
                    x0 = nx0;
                    x1 = d_L2(ix, b) + 1 + d_sel4(tk[0], tk[1], tk[2], tk[3], b);
                    x2 = n2;
                    p++;
                }
            }
            if (stop) {
                const int l = p - start;
                if (x2 <= (uint64_t)pr.max_dup && l >= 16) {
                    if (nh < H) {
                        DHit h; h.x0 = x0; h.freq = (uint32_t)x2; h.rPos = (uint16_t)start; h.len = (uint16_t)l;
                        hits[(size_t)r * H + nh] = h;
                    }
                    nh++; ns += (uint32_t)x2;
                    pos = start + l;
                } else pos = start + 1;
                searching = false;
            }
        }
        nhits[r] = (uint32_t)nh;
        nseeds[r] = ns;
    }
    d_wave_add(ctr + CTR_STEPS, steps);
    d_wave_add(ctr + CTR_BLOCKS, blocks);
}

// ---------------------------------------------------------------------------------------------
// k_locate: one lane = one seed occurrence (one row of a hit's SA interval): LF-walk to the
// nearest sampled row (bwt_sa :127-137), then emit the seed.  Lane -> (read, hit, j) by binary
// search of the per-read seed offsets (exclusive scan of nseeds).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_locate(const DIndex ix, int n_reads, int H, const DHit *__restrict__ hits, const uint32_t *__restrict__ seed_off,
         uint32_t total, DSeed *__restrict__ seeds, unsigned long long *ctr)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long lf = 0;
    if (t < total) {
        int lo = 0, hi = n_reads;           // largest r with seed_off[r] <= t
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (seed_off[mid] <= t) lo = mid; else hi = mid; }
        const int r = lo;
        uint32_t u = t - seed_off[r];
        const DHit *h = hits + (size_t)r * H;
        while (u >= h->freq) { u -= h->freq; h++; }
        uint64_t k = h->x0 + u;
        const uint64_t mask = (uint64_t)ix.sa_intv - 1;
        uint64_t steps = 0;
        while (k & mask) { k = d_lf(ix, k); steps++; }
        lf = steps;
        DSeed s;
        s.gPos = (int64_t)(steps + ix.sa[k / (uint64_t)ix.sa_intv]);
        s.rPos = h->rPos; s.rLen = s.gLen = h->len; s.flags = SEED_SIMPLE;
        seeds[t] = s;
    }
    d_wave_add(ctr + CTR_LF, lf);
    d_wave_add(ctr + CTR_SA, t < total ? 1 : 0);
}
