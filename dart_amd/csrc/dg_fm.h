// dart_amd/csrc/dg_fm.h -- FM-index kernels: maximal-exact-match search and SA locate.
//
// Replaces bwt_search.cpp:43-182 (bwt_occ, bwt_occ4, bwt_2occ4, bwt_invPsi, bwt_sa, BWT_Search)
// and the search loop of IdentifySeedPairs (AlignmentCandidates.cpp:181-215).
//
// Bound: HBM / Infinity-Cache random 64-byte reads -- one Occ query = one aligned 64-byte block,
// exactly as in the reference's layout (so the algorithmic bytes are the reference's).  Round-1
// profiling showed the first version issue-bound instead (1445 VALU instructions per wave-step), so
// the device block is a re-layout of the same 64 bytes that makes counting cheap:
//
//   reference block (bwtindex.c:53-75)      device block (k_relayout_bwt, built once in dg_init)
//   4 x u64 counts A,C,G,T                  4 x u64 counts A,C,G,T
//   8 x u32, 16 symbols each, MSB first     4 x u32 HIGH bits + 4 x u32 LOW bits of the 128 symbols,
//                                           symbol j -> word j>>5, bit j&31
//
// With bit planes, "symbols whose high bit is h" is one XOR, and the two bases that share h are
// counted with two popcounts per 32 symbols.  One bi-interval extension only needs Occ of the
// extension base b and of its plane partner b^1 at two rows (see d_extend) -- the sum over j>b that
// moves x0 follows from sum_j Occ(j,k) = k+1.  Reads are pre-encoded to 4 bit/base (k_encode) and
// staged in LDS word-major, so the per-step base fetch is one conflict-free ds_read.
#pragma once
#include "dg_common.h"

struct OccBlock { uint4 q0, q1, q2, q3; };   // q0 = C[A],C[C]; q1 = C[G],C[T]; q2 = high plane; q3 = low plane

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
__device__ __forceinline__ uint64_t d_u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// in-block counts over symbols [0..o] of one block: P = #symbols with high bit == hi,
// Q = #symbols with high bit == hi and low bit == 1
__device__ __forceinline__ void d_plane_counts(const uint4 H, const uint4 Lo, uint32_t hmask, uint32_t o, uint32_t &P, uint32_t &Q)
{
    const uint32_t wi = o >> 5;
    const uint32_t pm = (2u << (o & 31)) - 1u;          // low (o&31)+1 bits; wraps to all ones at 31
    const uint32_t m0 = wi == 0 ? pm : 0xFFFFFFFFu;
    const uint32_t m1 = wi == 1 ? pm : (wi > 1 ? 0xFFFFFFFFu : 0u);
    const uint32_t m2 = wi == 2 ? pm : (wi > 2 ? 0xFFFFFFFFu : 0u);
    const uint32_t m3 = wi == 3 ? pm : 0u;
    const uint32_t e0 = (H.x ^ hmask) & m0, e1 = (H.y ^ hmask) & m1, e2 = (H.z ^ hmask) & m2, e3 = (H.w ^ hmask) & m3;
    P = __popc(e0) + __popc(e1) + __popc(e2) + __popc(e3);
    Q = __popc(e0 & Lo.x) + __popc(e1 & Lo.y) + __popc(e2 & Lo.z) + __popc(e3 & Lo.w);
}

// Occ(b, row) and Occ(b^1, row) for the row at in-block offset o (bwt_occ4 :67-84, two of the four)
__device__ __forceinline__ void d_occ_pair(const OccBlock &B, int b, uint32_t o, uint64_t &cb, uint64_t &cp)
{
    const bool hi = (b & 2) != 0, lo = (b & 1) != 0;
    uint32_t P, Q;
    d_plane_counts(B.q2, B.q3, hi ? 0u : 0xFFFFFFFFu, o, P, Q);
    const uint4 C = hi ? B.q1 : B.q0;                   // (C[2h] lo,hi, C[2h+1] lo,hi)
    const uint64_t c_even = d_u64(C.x, C.y) + (P - Q), c_odd = d_u64(C.z, C.w) + Q;
    cb = lo ? c_odd : c_even;
    cp = lo ? c_even : c_odd;
}

// LF mapping, bwt_invPsi :119-125 (the symbol and its Occ come from the same block)
__device__ __forceinline__ uint64_t d_lf(const DIndex &ix, uint64_t k)
{
    if (k == ix.primary) return 0;
    const uint64_t x = k - (k > ix.primary);
    const OccBlock B = d_load_block(ix, x >> 7);
    const uint32_t o = (uint32_t)(x & 127), wi = o >> 5, bit = o & 31;
    const uint32_t hw = wi == 0 ? B.q2.x : wi == 1 ? B.q2.y : wi == 2 ? B.q2.z : B.q2.w;
    const uint32_t lw = wi == 0 ? B.q3.x : wi == 1 ? B.q3.y : wi == 2 ? B.q3.z : B.q3.w;
    const int c = (int)(((hw >> bit) & 1u) * 2u + ((lw >> bit) & 1u));
    uint64_t cb, cp;
    d_occ_pair(B, c, o, cb, cp);
    return d_L2(ix, c) + cb;
}

// ---------------------------------------------------------------------------------------------
// k_relayout_bwt: reference .bwt blocks -> device blocks (once per dg_init)
// ---------------------------------------------------------------------------------------------
__global__ void k_relayout_bwt(const uint32_t *__restrict__ src, uint64_t src_words, uint4 *__restrict__ dst, uint64_t n_blocks)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t *s = src + b * 16;
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = (b * 16 + i) < src_words ? s[i] : 0u;
    uint32_t H[4] = {0, 0, 0, 0}, L[4] = {0, 0, 0, 0};
    for (int j = 0; j < 128; j++) {
        const uint32_t sym = (w[8 + (j >> 4)] >> ((~j & 15) << 1)) & 3u;     // bwt_B0, bwt_search.cpp:30-32
        H[j >> 5] |= (sym >> 1) << (j & 31);
        L[j >> 5] |= (sym & 1u) << (j & 31);
    }
    dst[b * 4 + 0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[b * 4 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
    dst[b * 4 + 2] = make_uint4(H[0], H[1], H[2], H[3]);
    dst[b * 4 + 3] = make_uint4(L[0], L[1], L[2], L[3]);
}

// ---------------------------------------------------------------------------------------------
// k_encode: ASCII reads -> 4 bit/base (nst_nt4_table codes 0..5), 8 bases per u32, read-major
// (enc[r * W + w]).  One thread per output word, so the 8-byte source groups of a read are
// fetched by neighbouring lanes (coalesced) and the stores are contiguous.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_encode(const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off, const uint16_t *__restrict__ rlen,
         int n_reads, int W, uint32_t *__restrict__ enc)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_reads * W) return;
    const int r = (int)(t / W), w = (int)(t % W);
    const unsigned char *s = seq + seq_off[r];
    const int len = rlen[r];
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = w * 8 + j;
        const uint32_t c = p < len ? d_nt4(s[p]) : 4u;
        v |= c << (4 * j);
    }
    enc[t] = v;
}

// ---------------------------------------------------------------------------------------------
// k_seed: one lane = one read.  Greedy left-to-right tiling with maximal exact matches
// (IdentifySeedPairs + BWT_Search).  The two nested loops of the reference are flattened into one
// loop whose every trip is exactly one bi-interval extension, so the lanes of a wave stay
// converged on the memory-bound step whatever their read positions are.
// Output: per read up to H intervals (a hit is >= 16 long, so H = max_rlen/16 + 1 always fits).
// ---------------------------------------------------------------------------------------------
template <bool USE_LDS>
__global__ void __launch_bounds__(256)
k_seed(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int n_reads, int W, int H,
       DHit *__restrict__ hits, uint32_t *__restrict__ nhits, uint32_t *__restrict__ nseeds, unsigned long long *ctr)
{
    extern __shared__ uint32_t sh[];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long steps = 0, blocks = 0;
    if (USE_LDS) {
        if (r < n_reads) for (int w = 0; w < W; w++) sh[w * 256 + threadIdx.x] = enc[(size_t)r * W + w];
        __syncthreads();
    }
    if (r < n_reads) {
        const int len = rlen[r], end_pos = len - 13;
        auto code = [&](int p) -> int {
            const uint32_t w = USE_LDS ? sh[(p >> 3) * 256 + threadIdx.x] : enc[(size_t)r * W + (p >> 3)];
            return (int)((w >> ((p & 7) << 2)) & 15u);
        };
        int pos = 0, start = 0, p = 0, nh = 0;
        uint32_t ns = 0;
        bool searching = false;
        uint64_t x0 = 0, x1 = 0, x2 = 0;
        while (true) {
            if (!searching) {
                while (pos < end_pos && code(pos) > 3) pos++;
                if (pos >= end_pos) break;
                const int c = code(pos);
                start = pos; p = pos + 1; searching = true;
                x0 = d_L2(ix, c) + 1; x1 = d_L2(ix, 3 - c) + 1; x2 = d_L2(ix, c + 1) - d_L2(ix, c);
            }
            bool stop = p >= len;
            int c = 4;
            if (!stop) { c = code(p); stop = c > 3; }
            if (!stop) {
                // one step of BWT_Search :152-170 for the single base b = 3 - c
                const int b = 3 - c;
                const uint64_t k = x1 - 1, l = k + x2;
                const uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
                uint64_t tkb, tkp, tlb, tlp;
                const OccBlock B = d_load_block(ix, kk >> 7);
                d_occ_pair(B, b, (uint32_t)(kk & 127), tkb, tkp);
                if ((ll >> 7) != (kk >> 7)) {
                    const OccBlock B2 = d_load_block(ix, ll >> 7);
                    d_occ_pair(B2, b, (uint32_t)(ll & 127), tlb, tlp);
                    blocks += 2;
                } else {
                    d_occ_pair(B, b, (uint32_t)(ll & 127), tlb, tlp);
                    blocks += 1;
                }
                steps++;
                const uint64_t n2 = tlb - tkb;
                if (n2 == 0) stop = true;
                else {
                    // sum over j > b of (tl[j] - tk[j]), using sum_j t[j] = row + 1
                    uint64_t above;
                    if (b == 0) above = (ll - kk) - n2;
                    else if (b == 1) above = (ll - kk) - n2 - (tlp - tkp);
                    else if (b == 2) above = tlp - tkp;
                    else above = 0;
                    x0 = x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0) + above;
                    x1 = d_L2(ix, b) + 1 + tkb;
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
