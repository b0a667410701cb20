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

// one step of BWT_Search (:152-170) with base c: returns false when the extension is empty
__device__ __forceinline__ bool d_extend(const DIndex &ix, int c, uint64_t &x0, uint64_t &x1, uint64_t &x2, uint32_t &nblk)
{
    const int b = 3 - c;
    const uint64_t k = x1 - 1, l = k + x2;
    const uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
    uint64_t tkb, tkp, tlb, tlp;
    const OccBlock B = d_load_block(ix, kk >> 7);
    d_occ_pair(B, b, (uint32_t)(kk & 127), tkb, tkp);
    if ((ll >> 7) != (kk >> 7)) {
        const OccBlock B2 = d_load_block(ix, ll >> 7);
        d_occ_pair(B2, b, (uint32_t)(ll & 127), tlb, tlp);
        nblk = 2;
    } else {
        d_occ_pair(B, b, (uint32_t)(ll & 127), tlb, tlp);
        nblk = 1;
    }
    const uint64_t n2 = tlb - tkb;
    if (n2 == 0) return false;
    // sum over j > b of (tl[j] - tk[j]), using sum_j t[j] = row + 1
    uint64_t above;
    if (b == 0) above = (ll - kk) - n2;
    else if (b == 1) above = (ll - kk) - n2 - (tlp - tkp);
    else if (b == 2) above = tlp - tkp;
    else above = 0;
    x0 = x0 + ((x1 <= ix.primary && x1 + x2 - 1 >= ix.primary) ? 1 : 0) + above;
    x1 = d_L2(ix, b) + 1 + tkb;
    x2 = n2;
    return true;
}

// ---------------------------------------------------------------------------------------------
// K-mer prefix table (built once in dg_init by k_build_ktab): entry id = sum_i base[i] << 2i of
// the first K bases of a search (first base in the lowest bits); it holds the bi-interval after
// those K bases, i.e. after K-1 steps of BWT_Search, plus how many steps / Occ blocks the
// reference's loop would have spent getting there (for the algorithmic-byte accounting).
//   e[0] = x0, e[1] = x1, e[2] = x2 (40 bits) | ref_steps << 40 | ref_blocks << 48
// x2 == 0: the K-mer does not occur; because K <= 16 such a search can never yield a seed
// (bwt_search.cpp:173 needs len >= 16), so it is skipped with the reference's step count.
// Results are unchanged by construction: the table is the reference's own loop, memoised.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_build_ktab(const DIndex ix, int K, uint64_t *__restrict__ tab)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (1u << (2 * K))) return;
    int c = (int)(id & 3u);
    uint64_t x0 = d_L2(ix, c) + 1, x1 = d_L2(ix, 3 - c) + 1, x2 = d_L2(ix, c + 1) - d_L2(ix, c);
    uint32_t steps = 0, blocks = 0;
    for (int i = 1; i < K; i++) {
        c = (int)((id >> (2 * i)) & 3u);
        uint32_t nb;
        steps++;
        const bool ok = d_extend(ix, c, x0, x1, x2, nb);
        blocks += nb;
        if (!ok) { x2 = 0; break; }
    }
    tab[(size_t)id * 3 + 0] = x0; tab[(size_t)id * 3 + 1] = x1;
    tab[(size_t)id * 3 + 2] = (x2 & 0xFFFFFFFFFFull) | ((uint64_t)steps << 40) | ((uint64_t)blocks << 48);
}

// ---------------------------------------------------------------------------------------------
// k_seed: one lane = one read at a time.  Greedy left-to-right tiling with maximal exact matches
// (IdentifySeedPairs + BWT_Search).  The two nested loops of the reference are flattened into one
// loop whose every trip is at most one bi-interval extension, so the lanes of a wave stay
// converged on the memory-bound step whatever their read positions are.
// Reads differ a lot in work (every failed search restarts one base further, :209), so waves are
// persistent and refill their idle lanes from a global read queue as soon as SEED_REFILL of them
// are idle: one atomic per refill, each refilled lane stages its read's 4-bit words into its own
// LDS column (the lane is the only reader of that column, so no barrier is needed).
// Output: per read up to H intervals (a hit is >= 16 long, so H = max_rlen/16 + 1 always fits).
// ---------------------------------------------------------------------------------------------
#define SEED_REFILL 8
template <bool USE_LDS>
__global__ void __launch_bounds__(64)
k_seed(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int n_reads, int W, int H,
       DHit *__restrict__ hits, uint32_t *__restrict__ nhits, uint32_t *__restrict__ nseeds, unsigned int *next_read, unsigned long long *ctr)
{
    extern __shared__ uint32_t sh[];
    const int lane = threadIdx.x;
    unsigned long long steps = 0, blocks = 0, steps_act = 0, blocks_act = 0, ktab_reads = 0;
    const int K = ix.ktab ? ix.ktab_k : 0;
    int r = -1, len = 0, end_pos = 0, pos = 0, start = 0, p = 0, nh = 0;
    uint32_t ns = 0;
    bool searching = false, exhausted = false;
    uint64_t x0 = 0, x1 = 0, x2 = 0;
    auto word = [&](int w) -> uint32_t { return w < W ? (USE_LDS ? sh[w * 64 + lane] : enc[(size_t)r * W + w]) : 0x44444444u; };
    auto code = [&](int q) -> int { return (int)((word(q >> 3) >> ((q & 7) << 2)) & 15u); };
    while (true) {
        const unsigned long long idle = __ballot(r < 0);
        if (idle) {
            const int n_idle = __popcll(idle);
            if (!exhausted && (n_idle >= SEED_REFILL || idle == ~0ull)) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(next_read, (unsigned int)n_idle);
                base = (unsigned int)__shfl((int)base, 0, 64);
                if (r < 0) {
                    const unsigned int mine = base + (unsigned int)__popcll(idle & ((1ull << lane) - 1ull));
                    if (mine < (unsigned int)n_reads) {
                        r = (int)mine;
                        if (USE_LDS) for (int w = 0; w < W; w++) sh[w * 64 + lane] = enc[(size_t)r * W + w];
                        len = rlen[r]; end_pos = len - 13; pos = 0; nh = 0; ns = 0; searching = false;
                    }
                }
                if (base + (unsigned int)n_idle >= (unsigned int)n_reads) exhausted = true;
            }
            if (__ballot(r >= 0) == 0) { if (exhausted) break; continue; }
        }
        if (r >= 0) {
            bool finished = false;
            if (!searching) {
                while (pos < end_pos && code(pos) > 3) pos++;
                if (pos >= end_pos) finished = true;
                else {
                    start = pos;
                    bool from_table = false;
                    if (K) {
                        // the K 4-bit codes from `pos` on: 48 bits out of three staged words
                        const int w0 = pos >> 3, sft = (pos & 7) << 2;
                        const uint64_t lo = (uint64_t)word(w0) | ((uint64_t)word(w0 + 1) << 32);
                        uint64_t v = sft ? ((lo >> sft) | ((uint64_t)word(w0 + 2) << (64 - sft))) : lo;
                        v &= (1ull << (4 * K)) - 1ull;
                        if ((v & 0x4444444444444444ull) == 0) {           // no N among them (codes 4,5 have bit 2 set)
                            uint64_t x = v & 0x3333333333333333ull;       // nibbles -> 2-bit pairs, first base lowest
                            x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
                            x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
                            x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
                            x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
                            const uint64_t *e = ix.ktab + (size_t)x * 3;
                            const uint64_t e2 = e[2];
                            ktab_reads++;
                            steps += (e2 >> 40) & 0xFF; blocks += (e2 >> 48) & 0xFF;
                            from_table = true;
                            if ((e2 & 0xFFFFFFFFFFull) == 0) pos = start + 1;   // cannot reach 16: no seed here, next start
                            else { x0 = e[0]; x1 = e[1]; x2 = e2 & 0xFFFFFFFFFFull; p = start + K; searching = true; }
                        }
                    }
                    if (!from_table) {
                        const int c = code(pos);
                        p = pos + 1; searching = true;
                        x0 = d_L2(ix, c) + 1; x1 = d_L2(ix, 3 - c) + 1; x2 = d_L2(ix, c + 1) - d_L2(ix, c);
                    }
                }
            }
            if (searching) {
                bool stop = p >= len;
                int c = 4;
                if (!stop) { c = code(p); stop = c > 3; }
                if (!stop) {
                    uint32_t nb;
                    const bool ok = d_extend(ix, c, x0, x1, x2, nb);
                    steps++; blocks += nb; steps_act++; blocks_act += nb;
                    if (ok) p++; else stop = true;
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
            if (finished) { nhits[r] = (uint32_t)nh; nseeds[r] = ns; r = -1; }
        }
    }
    d_wave_add(ctr + CTR_STEPS, steps);
    d_wave_add(ctr + CTR_BLOCKS, blocks);
    d_wave_add(ctr + CTR_STEPS_ACT, steps_act);
    d_wave_add(ctr + CTR_BLOCKS_ACT, blocks_act);
    d_wave_add(ctr + CTR_KTAB, ktab_reads);
}

// ---------------------------------------------------------------------------------------------
// k_build_sa_dense (once per dg_init): SA of every `intv`-th row, derived from the reference's
// SA/32 by the reference's own walk (bwt_sa :127-137), so results cannot change.  Each entry
// also remembers how many LF steps the reference needs from that row (algorithmic-byte accounting).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_build_sa_dense(const DIndex ix, int intv, uint64_t n_entries, uint64_t *__restrict__ dense)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    uint64_t k = i * (uint64_t)intv, steps = 0;
    const uint64_t mask = (uint64_t)ix.sa_intv - 1;
    while (k & mask) { k = d_lf(ix, k); steps++; }
    const uint64_t pos = steps + ix.sa[k / (uint64_t)ix.sa_intv];      // sa[0] = -1 wraps as in the reference
    dense[i] = ((pos + 1) & 0xFFFFFFFFFFull) | (steps << 40);
}

// ---------------------------------------------------------------------------------------------
// k_locate: SA interval rows -> text positions (bwt_sa :127-137) -> seeds.
// One wave owns 64 consecutive reads; their seed counts are prefix-summed with wave shuffles and
// the wave then walks its seeds 64 at a time, one lane per seed occurrence (so a read with 100
// repeat copies is spread over lanes, wavefront-compaction style): lane -> read by a 6-step binary
// search over the in-register prefix (ds_bpermute), no global search; seeds come out coalesced.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_locate(const DIndex ix, int n_reads, int H, const DHit *__restrict__ hits, const uint32_t *__restrict__ nseeds,
         const uint32_t *__restrict__ seed_off, DSeed *__restrict__ seeds, unsigned long long *ctr)
{
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;      // first read of this wave
    unsigned long long lf = 0, lf_act = 0, nsa = 0;
    if (r0 < n_reads) {
        const int r = r0 + lane;
        const uint32_t mine = r < n_reads ? nseeds[r] : 0u;
        uint32_t incl = mine;                                                        // inclusive prefix over the wave
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        const uint32_t total = __shfl(incl, 63, 64);
        const uint32_t base = seed_off[r0];
        for (uint32_t c0 = 0; c0 < total; c0 += 64) {
            const uint32_t u = c0 + lane;
            int lo = 0;                                     // smallest lane j with incl[j] > u
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint32_t v = __shfl(incl, lo + step - 1, 64);
                if (v <= u) lo += step;
            }
            lo = lo > 63 ? 63 : lo;
            const uint32_t excl = __shfl(incl, lo, 64) - __shfl(mine, lo, 64);      // all lanes take part in the shuffles
            if (u < total) {
                uint32_t w = u - excl;
                const DHit *h = hits + (size_t)(r0 + lo) * H;
                while (w >= h->freq) { w -= h->freq; h++; }
                uint64_t k = h->x0 + w;
                uint64_t steps = 0, pos;
                if (ix.sa_dense) {
                    const uint64_t mask = (uint64_t)ix.sa_dense_intv - 1;
                    while (k & mask) { k = d_lf(ix, k); steps++; }
                    const uint64_t e = ix.sa_dense[k / (uint64_t)ix.sa_dense_intv];
                    pos = steps + (e & 0xFFFFFFFFFFull) - 1;
                    lf += steps + (e >> 40); lf_act += steps;
                } else {
                    const uint64_t mask = (uint64_t)ix.sa_intv - 1;
                    while (k & mask) { k = d_lf(ix, k); steps++; }
                    pos = steps + ix.sa[k / (uint64_t)ix.sa_intv];
                    lf += steps; lf_act += steps;
                }
                nsa++;
                DSeed s;
                s.gPos = (int64_t)pos;
                s.rPos = h->rPos; s.rLen = s.gLen = h->len; s.flags = SEED_SIMPLE;
                seeds[base + u] = s;
            }
        }
    }
    d_wave_add(ctr + CTR_LF, lf);
    d_wave_add(ctr + CTR_LF_ACT, lf_act);
    d_wave_add(ctr + CTR_SA, nsa);
}
