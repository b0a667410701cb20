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
// (blocks are passed by value everywhere: a block chosen through a reference becomes a pointer select, which
// keeps both blocks in scratch memory)
__device__ __forceinline__ uint4 d_pick4(bool first, const uint4 a, const uint4 b)
{
    return make_uint4(first ? a.x : b.x, first ? a.y : b.y, first ? a.z : b.z, first ? a.w : b.w);
}
__device__ __forceinline__ void d_occ_pair(const OccBlock B, int b, uint32_t o, uint64_t &cb, uint64_t &cp)
{
    const bool hi = (b & 2) != 0, lo = (b & 1) != 0;
    uint32_t P, Q;
    d_plane_counts(B.q2, B.q3, hi ? 0u : 0xFFFFFFFFu, o, P, Q);
    const uint4 C = d_pick4(hi, B.q1, B.q0);            // (C[2h] lo,hi, C[2h+1] lo,hi)
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
// IN PLACE: a block is 64 bytes in both forms, a thread reads its whole block before it writes it, and the counts stay where they are --
// so the start-up can copy the file's bytes to their final place chunk by chunk and re-lay each chunk behind its copy (the bytes past
// the end of the file are the zeros the array was filled with).
__global__ void __launch_bounds__(256) k_relayout_bwt(uint4 *blk, uint64_t n_blocks)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint4 s2 = blk[b * 4 + 2], s3 = blk[b * 4 + 3];
    const uint32_t w[8] = { s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w };
    uint32_t H[4] = {0, 0, 0, 0}, L[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; q++) {
        // 16 symbols of word q, first symbol in the top two bits (bwt_B0, bwt_search.cpp:30-32) -> bits 16 (q & 1) ... + 15 of plane word q >> 1
        uint32_t h = 0, l = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t sym = (w[q] >> ((15 - j) << 1)) & 3u;
            h |= (sym >> 1) << j;
            l |= (sym & 1u) << j;
        }
        H[q >> 1] |= h << (16 * (q & 1));
        L[q >> 1] |= l << (16 * (q & 1));
    }
    blk[b * 4 + 2] = make_uint4(H[0], H[1], H[2], H[3]);
    blk[b * 4 + 3] = make_uint4(L[0], L[1], L[2], L[3]);
}

// ---------------------------------------------------------------------------------------------
// k_encode: ASCII reads -> the pac format of the text, read-major, W = 2*W2 words per read:
//   words [0, W2): 2 bit/base (nst_nt4_table codes 0..3), 16 bases per u32, FIRST base in the TOP bits
//   words [W2, W): the same positions, 0b11 where the read has no A/C/G/T there (N, or past the end)
// so that 64 read bases and 64 text bases can be XORed word against word, and a K-mer is a shift.
// One thread per output word: neighbouring lanes read neighbouring 16-byte groups and store contiguously.
// ---------------------------------------------------------------------------------------------
typedef uint4 __attribute__((aligned(1))) uint4_a1;
// four characters at once (SWAR): x = 4 ASCII bytes in string order, keep = 0xFF in the bytes that belong to the read.
// b8 = their four 2-bit codes, first character in the top bits; m8 = 0b11 where the character is not A/C/G/T (either case).
// (One base at a time through nst_nt4_table this kernel was 278 M wave-instructions per 2 M reads, half of k_seed's.)
__host__ __device__ __forceinline__ void d_enc4(uint32_t x, uint32_t keep, uint32_t &b8, uint32_t &m8)
{
    const uint32_t u = x & 0xDFDFDFDFu;                                             // upper case
    auto nz = [](uint32_t z) { return ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z; };     // bit 7 of a byte set <=> the byte is not zero
    const uint32_t none = nz(u ^ 0x41414141u) & nz(u ^ 0x43434343u) & nz(u ^ 0x47474747u) & nz(u ^ 0x54545454u);
    const uint32_t inv = ((none & 0x80808080u) >> 7) | (~keep & 0x01010101u);       // 1 in every byte that is not a base of the read
    const uint32_t code = ((u >> 1) ^ (u >> 2)) & 0x03030303u & ~(inv * 3u);        // d_nt4's formula, byte-wise (the mask keeps bits that came from the same byte)
    b8 = (code * 0x40100401u) >> 24;                                                // bytes b0..b3 -> b0<<6 | b1<<4 | b2<<2 | b3 (no two partial products meet)
    m8 = ((inv * 0x40100401u) >> 24) * 3u;
}

__global__ void __launch_bounds__(256)
k_encode(const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off, const uint16_t *__restrict__ rlen,
         int n_reads, int W, int lg, uint32_t *__restrict__ enc)
{
    const int W2 = W >> 1;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread = 16 bases = one base word + one mask word;
    const size_t rr = t >> lg;                                           // 2^lg >= W2 threads per read (no division)
    const int ww = (int)(t & ((1u << lg) - 1u));
    if (rr >= (size_t)n_reads || ww >= W2) return;
    const int r = (int)rr;
    const int left = (int)rlen[r] - ww * 16;                            // bases of the read from this group on
    uint4 q = make_uint4(0, 0, 0, 0);
    if (left > 0) q = *(const uint4_a1 *)(seq + seq_off[r] + ww * 16);  // may run up to 15 bytes past the read: the batch buffer is padded
    auto keep = [left](int w) -> uint32_t { const int k = left - 4 * w; return k >= 4 ? 0xFFFFFFFFu : (k <= 0 ? 0u : (1u << (8 * k)) - 1u); };
    uint32_t b0, b1, b2, b3, m0, m1, m2, m3;
    d_enc4(q.x, keep(0), b0, m0); d_enc4(q.y, keep(1), b1, m1); d_enc4(q.z, keep(2), b2, m2); d_enc4(q.w, keep(3), b3, m3);
    enc[(size_t)r * W + ww] = (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
    enc[(size_t)r * W + W2 + ww] = (m0 << 24) | (m1 << 16) | (m2 << 8) | m3;
}

// one step of BWT_Search (:152-170) with base c, given the already loaded Occ blocks of rows kk and ll
// (B2 is only read when they differ): returns false when the extension is empty
__device__ __forceinline__ bool d_extend_finish(const DIndex &ix, int c, const OccBlock B, const OccBlock B2, uint64_t kk, uint64_t ll,
                                                uint64_t &x0, uint64_t &x1, uint64_t &x2)
{
    const int b = 3 - c;
    uint64_t tkb, tkp, tlb, tlp;
    d_occ_pair(B, b, (uint32_t)(kk & 127), tkb, tkp);
    const bool same = (ll >> 7) == (kk >> 7);
    OccBlock E;
    E.q0 = d_pick4(same, B.q0, B2.q0); E.q1 = d_pick4(same, B.q1, B2.q1); E.q2 = d_pick4(same, B.q2, B2.q2); E.q3 = d_pick4(same, B.q3, B2.q3);
    d_occ_pair(E, b, (uint32_t)(ll & 127), tlb, tlp);
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
__device__ __forceinline__ void d_extend_rows(const DIndex &ix, uint64_t x1, uint64_t x2, uint64_t &kk, uint64_t &ll)
{
    const uint64_t k = x1 - 1, l = k + x2;
    kk = k - (k >= ix.primary); ll = l - (l >= ix.primary);
}
__device__ __forceinline__ bool d_extend(const DIndex &ix, int c, uint64_t &x0, uint64_t &x1, uint64_t &x2, uint32_t &nblk)
{
    uint64_t kk, ll;
    d_extend_rows(ix, x1, x2, kk, ll);
    const OccBlock B = d_load_block(ix, kk >> 7);
    OccBlock B2;
    B2.q0 = B2.q1 = B2.q2 = B2.q3 = make_uint4(0, 0, 0, 0);
    nblk = 1;
    if ((ll >> 7) != (kk >> 7)) { B2 = d_load_block(ix, ll >> 7); nblk = 2; }
    return d_extend_finish(ix, c, B, B2, kk, ll, x0, x1, x2);
}
// LF mapping given the loaded block of row x = k - (k > primary), k != primary
__device__ __forceinline__ uint64_t d_lf_finish(const DIndex &ix, const OccBlock B, uint64_t x)
{
    const uint32_t o = (uint32_t)(x & 127), wi = o >> 5, bit = o & 31;
    const uint32_t hw = wi == 0 ? B.q2.x : wi == 1 ? B.q2.y : wi == 2 ? B.q2.z : B.q2.w;
    const uint32_t lw = wi == 0 ? B.q3.x : wi == 1 ? B.q3.y : wi == 2 ? B.q3.z : B.q3.w;
    const int c = (int)(((hw >> bit) & 1u) * 2u + ((lw >> bit) & 1u));
    uint64_t cb, cp;
    d_occ_pair(B, c, o, cb, cp);
    return d_L2(ix, c) + cb;
}

// ---------------------------------------------------------------------------------------------
// K-mer prefix table (built once in dg_init by k_build_ktab): entry id = the first K bases of a
// search as a 2K-bit number, first base in the top bits; it holds the bi-interval after those K
// bases, i.e. after K-1 steps of BWT_Search, plus how many steps / Occ blocks the reference's
// loop would have spent getting there (for the algorithmic-byte accounting).
//   16-byte entries: w0 = x0 (40 bits) | x1[23:0] << 40
//                    w1 = x1[39:24] | x2 (31 bits) << 16 | ref_steps (5) << 47 | ref_blocks (6) << 52 | located << 62 | overflow << 63
//   located: a unique K-mer (x2 == 1, full/dense SA present) holds its text position + 1 in the x1 field and the
//   reference's LF count of bwt_sa in the x2 field: the search skips the locate trip
// x2 == 0: the K-mer does not occur; because K <= 16 such a search can never yield a seed
// (bwt_search.cpp:173 needs len >= 16), so it is skipped with the reference's step count.
// overflow (x2 >= 2^31, a K-mer with billions of copies): the search starts from its first base instead.
// Results are unchanged by construction: the table is the reference's own loop, memoised.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_build_ktab(const DIndex ix, int K, uint64_t *__restrict__ tab)
{
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < (1ull << (2 * K)); id += (uint64_t)gridDim.x * blockDim.x) {   // grid-stride (K = 16: 2^32 entries)
    int c = (int)((id >> (2 * (K - 1))) & 3u);
    uint64_t x0 = d_L2(ix, c) + 1, x1 = d_L2(ix, 3 - c) + 1, x2 = d_L2(ix, c + 1) - d_L2(ix, c);
    uint32_t steps = 0, blocks = 0;
    for (int i = 1; i < K; i++) {
        c = (int)((id >> (2 * (K - 1 - i))) & 3u);
        uint32_t nb;
        steps++;
        const bool ok = d_extend(ix, c, x0, x1, x2, nb);
        blocks += nb;
        if (!ok) { x2 = 0; break; }
    }
    uint64_t lfc = 0;
    if (x2 == 1 && ix.sa_dense) {
        // a unique K-mer: the search goes straight to the text comparison, so the entry carries the text position
        // (bwt_sa of row x0, same encoding as an SA entry: pos+1 and the reference's LF count) in place of x1
        uint64_t k = x0, st = 0;
        const uint64_t mask = (uint64_t)ix.sa_dense_intv - 1;
        while (k & mask) { k = d_lf(ix, k); st++; }
        const uint64_t e = ix.sa_dense[k >> ix.sa_dense_shift];
        x1 = (st + (e & 0xFFFFFFFFFFull)) & 0xFFFFFFFFFFull;          // (pos + 1), pos = st + SA - 1
        lfc = st + (e >> 40);                                          // the walk to a sampled row has no fixed bound
    }
    uint64_t w0 = (x0 & 0xFFFFFFFFFFull) | ((x1 & 0xFFFFFFull) << 40);
    const bool located = x2 == 1 && ix.sa_dense;
    uint64_t w1 = ((x1 >> 24) & 0xFFFFull) | (((located ? lfc : x2) & 0x7FFFFFFFull) << 16) | ((uint64_t)steps << 47) | ((uint64_t)blocks << 52);
    if (located) w1 |= 1ull << 62;
    if (x2 >> 31) w1 |= 1ull << 63;
    tab[id * 2 + 0] = w0; tab[id * 2 + 1] = w1;
    }
}

// ---------------------------------------------------------------------------------------------
// k_seed: one lane = one read at a time.  Greedy left-to-right tiling with maximal exact matches
// (IdentifySeedPairs + BWT_Search).  The two nested loops of the reference are flattened into one
// loop whose every trip is at most one bi-interval extension, so the lanes of a wave stay
// converged on the memory-bound step whatever their read positions are.
// Reads differ a lot in work (every failed search restarts one base further, :209), so waves are
// persistent and refill their idle lanes from a global read queue as soon as SEED_REFILL of them
// are idle: one atomic per refill, each refilled lane gets its read's words (k_encode format) staged into its own
// LDS column (the lane is the only reader of that column, so no barrier is needed).
// Output: per read up to H intervals (a hit is >= 16 long, so H = max_rlen/16 + 1 always fits).
// ---------------------------------------------------------------------------------------------
#define SEED_REFILL 8

// slow path of the text comparison (strand boundary, end of the text): up to 16 symbols T[t..t+16) in the
// read's format (first symbol in the top bits); *nv = how many exist.  T = forward strand + reverse
// complement, read from the forward pac.
__device__ __forceinline__ uint32_t d_text16_slow(const DIndex &ix, int64_t t, int &nv)
{
    uint32_t y = 0;
    nv = 0;
    for (int j = 0; j < 16; j++) {
        const char ch = d_refchar(ix, t + j);
        if (ch == 0) break;
        y |= (uint32_t)d_nt4((unsigned char)ch) << (30 - 2 * j);
        nv = j + 1;
    }
    return y;
}

// One BWT_Search (bwt_search.cpp:139-182) in flight, advanced one memory access per trip:
// mode 1 = FM steps, 3 = locating the unique row (LF steps), 2 = comparing with the text, 0 = done.
struct Search {
    uint64_t x0, x1, x2, lk;
    int64_t tpos;
    uint32_t lsteps;
    int start, p, mode;
    // what the reference's loop would have spent on this search (algorithmic-byte accounting)
    uint32_t ref_steps, ref_blocks;
    // result (valid when mode == 0): hit_len 0 = no seed; located = x0 already holds the text position
    int hit_len; bool located;
};
struct SeedCtr { unsigned long long steps, blocks, steps_act, blocks_act, ktab, lf_ref, lf_act, n_direct; };

// the read as two word accessors: rb(w) = 16 bases (2 bit each, first base on top), rm(w) = 0b11 where not A/C/G/T
// 16 positions starting at q, first on top
template <class F>
__device__ __forceinline__ uint32_t d_win16(F &f, int q)
{
    const int w = q >> 4;
    return __funnelshift_l(f(w + 1), f(w), (uint32_t)((q & 15) << 1));
}
template <class F>
__device__ __forceinline__ uint32_t d_at(F &f, int q) { return (f(q >> 4) >> (30 - ((q & 15) << 1))) & 3u; }
__device__ __forceinline__ uint32_t d_rev2(uint32_t x)      // reverse the order of the 16 2-bit groups
{
    const uint32_t y = __brev(x);
    return ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
}

__device__ __forceinline__ void d_search_end(const DParams &pr, Search &s)   // bwt_search.cpp:173-179
{
    const int l = s.p - s.start;
    s.hit_len = (s.x2 <= (uint64_t)pr.max_dup && l >= 16) ? l : 0;
    s.located = s.mode == 2;
    s.mode = 0;
}

// Every trip of a search is split in three so that a wave whose lanes are in different modes pays ONE
// memory latency per trip: d_begin_issue / d_trip_issue only compute the lane's addresses (TripAddr),
// d_trip_load issues all loads from ONE place in the program, d_begin_finish / d_trip_finish consume them.
// (With a load inside each divergent mode body the bodies load-and-wait one after the other; with
// the loads in the bodies but the uses later, the compiler still waits in each body to copy the
// value into the merged register.)
typedef uint4 __attribute__((aligned(4))) uint4_a4;
typedef uint2 __attribute__((aligned(4))) uint2_a4;
struct TripAddr { const uint4 *pa, *pb; const uint4_a4 *p16; const uint2_a4 *p8; };     // nullptr = not needed
struct TripData { OccBlock a, b; uint4 s16; uint2 s8; uint64_t kk, ll; uint32_t aux; };
enum { T_NONE = 0, T_TABLE, T_SINGLE, T_STEP, T_STOP, T_LF, T_LF_PRIMARY, T_SA, T_CMP, T_CMP_SLOW };

__device__ __forceinline__ void d_trip_load(const TripAddr &ta, TripData &t)
{
    if (ta.pa) { t.a.q0 = ta.pa[0]; t.a.q1 = ta.pa[1]; t.a.q2 = ta.pa[2]; t.a.q3 = ta.pa[3]; }
    if (ta.pb) { t.b.q0 = ta.pb[0]; t.b.q1 = ta.pb[1]; t.b.q2 = ta.pb[2]; t.b.q3 = ta.pb[3]; }
    if (ta.p16) t.s16 = *ta.p16;
    if (ta.p8) t.s8 = *ta.p8;
}

// mode 0 -> begin a search at `start` (a position holding A/C/G/T): prefix table or the single-base interval
template <class RB, class RM>
__device__ __forceinline__ void d_begin_issue(const DIndex &ix, int K, RB &rb, RM &rm, int start, Search &s, SeedCtr &c, TripAddr &ta, TripData &t)
{
    s.start = start; s.ref_steps = s.ref_blocks = 0; s.lsteps = 0; s.hit_len = 0; s.located = false;
    t.aux = T_SINGLE;
    if (K) {
        const uint32_t sft = 32u - 2u * (uint32_t)K;
        if ((d_win16(rm, start) >> sft) == 0) {            // K bases, none of them N or past the end
            ta.p16 = (const uint4_a4 *)(ix.ktab + (size_t)(d_win16(rb, start) >> sft) * 2);
            c.ktab++;
            t.aux = T_TABLE;
        }
    }
}
template <class RB>
__device__ __forceinline__ void d_begin_finish(const DIndex &ix, int K, RB &rb, Search &s, SeedCtr &c, const TripData &t)
{
    if (t.aux == T_TABLE) {
        const uint64_t w0 = d_u64(t.s16.x, t.s16.y), w1 = d_u64(t.s16.z, t.s16.w);
        if (!(w1 >> 63)) {
            const bool located = (w1 >> 62) & 1ull;       // unique K-mer: the entry already is the located position (see k_build_ktab)
            const uint64_t f2 = (w1 >> 16) & 0x7FFFFFFFull, x2 = located ? 1ull : f2;
            s.ref_steps = (uint32_t)((w1 >> 47) & 31u); s.ref_blocks = (uint32_t)((w1 >> 52) & 63u);
            if (x2 == 0) { s.mode = 0; return; }          // cannot reach 16: no seed from this start
            s.x0 = w0 & 0xFFFFFFFFFFull; s.x1 = (w0 >> 40) | ((w1 & 0xFFFFull) << 24); s.x2 = x2; s.p = s.start + K; s.mode = 1;
            if (located) {
                s.lk = s.x1 | (f2 << 40);
                s.tpos = (int64_t)s.x1 - 1; s.lsteps = 0; s.mode = 2; c.n_direct++;
            }
            return;
        }
    }
    const int cc = (int)d_at(rb, s.start);
    s.p = s.start + 1; s.mode = 1;
    s.x0 = d_L2(ix, cc) + 1; s.x1 = d_L2(ix, 3 - cc) + 1; s.x2 = d_L2(ix, cc + 1) - d_L2(ix, cc);
}

// modes 1,3,2: the addresses of this trip.  MODES: bit m set = the search may be in mode m (only those modes' code is instantiated;
// a search in another mode is left alone)
#define TM_STEP 2
#define TM_CMP 4
#define TM_LOC 8
#define TM_ALL 14
template <int MODES = TM_ALL, class RM>
__device__ __forceinline__ void d_trip_issue(const DIndex &ix, RM &rm, int len, bool direct, Search &s, SeedCtr &c, TripAddr &ta, TripData &t)
{
    const int mode = MODES == TM_STEP ? 1 : MODES == TM_CMP ? 2 : MODES == TM_LOC ? 3 : s.mode;
    if ((MODES & TM_STEP) && mode == 1) {
        if (s.p >= len || d_at(rm, s.p)) { t.aux = T_STOP; return; }
        d_extend_rows(ix, s.x1, s.x2, t.kk, t.ll);
        ta.pa = ix.bwt + ((t.kk >> 7) << 2);
        if ((t.ll >> 7) != (t.kk >> 7)) ta.pb = ix.bwt + ((t.ll >> 7) << 2);
        t.aux = T_STEP;
    } else if ((MODES & TM_LOC) && mode == 3) {    // bwt_sa on the unique row, one LF step per trip
        if (s.lk & ((uint64_t)ix.sa_dense_intv - 1)) {
            if (s.lk == ix.primary) { t.aux = T_LF_PRIMARY; return; }
            t.kk = s.lk - (s.lk > ix.primary);
            ta.pa = ix.bwt + ((t.kk >> 7) << 2);
            t.aux = T_LF;
        } else {
            ta.p8 = (const uint2_a4 *)(ix.sa_dense + (s.lk >> ix.sa_dense_shift));
            t.aux = T_SA;
        }
    } else if ((MODES & TM_CMP) && mode == 2) {    // 64 text symbols = five pac words
        const int64_t tt = s.tpos + (s.p - s.start), L = ix.l_pac;
        const uint32_t *pw = (const uint32_t *)ix.pac;
        int64_t f0 = -1;                           // first forward-strand symbol of the window
        if (tt >= 0 && tt + 64 <= L) { f0 = tt; t.ll = 0; }
        else if (tt >= L && tt + 64 <= 2 * L) { f0 = 2 * L - 1 - tt - 63; t.ll = 1; }
        if (f0 >= 0) {
            ta.p16 = (const uint4_a4 *)(pw + (f0 >> 4)); ta.p8 = (const uint2_a4 *)(pw + (f0 >> 4) + 4);
            t.kk = (uint64_t)((f0 & 15) << 1); t.aux = T_CMP;
        } else t.aux = T_CMP_SLOW;
    }
}

// modes 1,3,2: consume the loads; when the search finishes, mode becomes 0 and hit_len/located hold the result.
// MODES: the modes the trip can have been issued in (step: T_STOP / T_STEP, locate: T_LF / T_LF_PRIMARY / T_SA, compare: T_CMP / T_CMP_SLOW); only
// their code is instantiated -- with a run-time `aux` alone every instance carried all of it (the per-base slow text comparison included)
template <int MODES = TM_ALL, class RB, class RM>
__device__ __forceinline__ void d_trip_finish(const DIndex &ix, const DParams &pr, RB &rb, RM &rm, int len, Search &s, SeedCtr &c, const TripData &t)
{
    if ((MODES & TM_STEP) && t.aux == T_STOP) d_search_end(pr, s);
    else if ((MODES & TM_STEP) && t.aux == T_STEP) {
        const int cc = (int)d_at(rb, s.p);
        const uint32_t nb = (t.ll >> 7) != (t.kk >> 7) ? 2u : 1u;
        const bool ok = d_extend_finish(ix, cc, t.a, t.b, t.kk, t.ll, s.x0, s.x1, s.x2);
        s.ref_steps++; s.ref_blocks += nb; c.steps_act++; c.blocks_act += nb;
        if (ok) s.p++; else d_search_end(pr, s);
    } else if ((MODES & TM_LOC) && t.aux == T_LF) { s.lk = d_lf_finish(ix, t.a, t.kk); s.lsteps++; c.lf_act++; }
    else if ((MODES & TM_LOC) && t.aux == T_LF_PRIMARY) { s.lk = 0; s.lsteps++; c.lf_act++; }
    else if ((MODES & TM_LOC) && t.aux == T_SA) {
        const uint64_t e = d_u64(t.s8.x, t.s8.y);
        s.tpos = (int64_t)(s.lsteps + (e & 0xFFFFFFFFFFull) - 1);
        s.lk = e;                                  // keeps the memoised reference LF count (bits 40..)
        s.mode = 2; c.n_direct++;
    } else if ((MODES & TM_CMP) && (t.aux == T_CMP || t.aux == T_CMP_SLOW)) {      // the interval is one text position: compare up to 64 bases
        int nv = 64, chunk = 64;
        uint32_t T0, T1 = 0, T2 = 0, T3 = 0;
        if (t.aux == T_CMP) {
            const uint32_t d0 = __builtin_bswap32(t.s16.x), d1 = __builtin_bswap32(t.s16.y), d2 = __builtin_bswap32(t.s16.z),
                           d3 = __builtin_bswap32(t.s16.w), d4 = __builtin_bswap32(t.s8.x), o = (uint32_t)t.kk;
            const uint32_t s0 = __funnelshift_l(d1, d0, o), s1 = __funnelshift_l(d2, d1, o), s2 = __funnelshift_l(d3, d2, o), s3 = __funnelshift_l(d4, d3, o);
            if (t.ll) { T0 = ~d_rev2(s3); T1 = ~d_rev2(s2); T2 = ~d_rev2(s1); T3 = ~d_rev2(s0); }   // T[t+j] = 3 - fwd[2L-1-t-j]
            else { T0 = s0; T1 = s1; T2 = s2; T3 = s3; }
        } else { T0 = d_text16_slow(ix, s.tpos + (s.p - s.start), nv); chunk = 16; }
        const int w = s.p >> 4;
        const uint32_t o = (uint32_t)((s.p & 15) << 1);
        const uint32_t b0 = rb(w), b1 = rb(w + 1), b2 = rb(w + 2), b3 = rb(w + 3), b4 = rb(w + 4);
        const uint32_t m0 = rm(w), m1 = rm(w + 1), m2 = rm(w + 2), m3 = rm(w + 3), m4 = rm(w + 4);
        const uint32_t N0 = __funnelshift_l(m1, m0, o), N1 = __funnelshift_l(m2, m1, o), N2 = __funnelshift_l(m3, m2, o), N3 = __funnelshift_l(m4, m3, o);
        const int in_read = len - s.p < chunk ? len - s.p : chunk;                 // bases left in the read
        const int lim = in_read < nv ? in_read : nv;
        auto tail = [&](int k) -> uint32_t { const int rem = lim - 16 * k; return rem >= 16 ? 0u : (rem <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu >> (2 * rem)); };
        const uint32_t e0 = (__funnelshift_l(b1, b0, o) ^ T0) | N0 | tail(0), e1 = (__funnelshift_l(b2, b1, o) ^ T1) | N1 | tail(1),
                       e2 = (__funnelshift_l(b3, b2, o) ^ T2) | N2 | tail(2), e3 = (__funnelshift_l(b4, b3, o) ^ T3) | N3 | tail(3);
        const int j = e0 ? __clz((int)e0) >> 1 : e1 ? 16 + (__clz((int)e1) >> 1) : e2 ? 32 + (__clz((int)e2) >> 1) : e3 ? 48 + (__clz((int)e3) >> 1) : 64;
        s.ref_steps += (uint32_t)j; s.ref_blocks += (uint32_t)j;                   // the reference: one step (one block, +<1 % crossings) per base
        s.p += j;
        if (j < chunk) {
            // a mismatch or the end of the text costs the reference one more (failing) step; N / end of read do not
            const uint32_t nsel = j < 16 ? N0 : j < 32 ? N1 : j < 48 ? N2 : N3;
            if (j < in_read && !((nsel >> (30 - ((j & 15) << 1))) & 1u)) { s.ref_steps++; s.ref_blocks++; }
            d_search_end(pr, s);
        }
    }
}

// progress of a read handed from k_seed to k_seed_heavy
struct __attribute__((aligned(8))) DHeavy { uint32_t read; int32_t pos, nh; uint32_t ns; };

#define SEED_BAIL 20             // searches (starts tried) after which an unfinished read moves to the heavy queue:
                                 // a clean read needs 3-8; only reads whose starts keep failing need more, and
                                 // those are the ones 64-wide speculation helps (long single searches are not)

template <bool USE_LDS>
__global__ void __launch_bounds__(64)
k_seed(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int n_reads, int W, int H,
       DHit *__restrict__ hits, uint32_t *__restrict__ nhits, uint32_t *__restrict__ nseeds, unsigned int *next_read,
       DHeavy *__restrict__ heavy, unsigned int *n_heavy, unsigned long long *ctr, int bail_trips, int both_thr)
{
    extern __shared__ uint32_t sh[];
    const int lane = threadIdx.x;
    SeedCtr c = {0, 0, 0, 0, 0, 0, 0, 0};
    const int K = ix.ktab ? ix.ktab_k : 0;
    const bool direct = ix.sa_dense != nullptr;      // unique intervals are finished by direct text comparison
    int r = -1, len = 0, end_pos = 0, pos = 0, nh = 0;
    uint32_t ns = 0, trips = 0, max_trips = 0, nsearch = 0;
    bool exhausted = false;
    unsigned int pool_next = 0, pool_end = 0;
    Search s; s.mode = 0;
    uint32_t wtrips = 0;
    const int W2 = W >> 1;
    // USE_LDS: every lane owns two LDS columns of W words, the read it is working on (cur) and the next one,
    // which is fetched while the lane is still busy (see the refill event below)
    int cur = 0, r_nxt = -1, len_nxt = 0, pos_nxt = 0, nh_nxt = 0;
    uint32_t ns_nxt = 0;
    uint32_t pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0;      // words of the reads being prefetched, in flight
    unsigned int pend_total = 0;
    uint32_t *const tab = sh + 2 * W * 64;             // rank in the batch -> lane | buffer << 8
    // unconditional reads with a clamped index + a select: `w < W2 ? load : const` would be a branch around every load
    auto rb = [&](int w) -> uint32_t {
        const int wc = w < W2 ? w : W2 - 1;
        const uint32_t v = USE_LDS ? sh[(cur * W + wc) * 64 + lane] : enc[(size_t)(r < 0 ? 0 : r) * W + wc];
        return w < W2 ? v : 0u;
    };
    auto rm = [&](int w) -> uint32_t {
        const int wc = w < W2 ? w : W2 - 1;
        const uint32_t v = USE_LDS ? sh[(cur * W + W2 + wc) * 64 + lane] : enc[(size_t)(r < 0 ? 0 : r) * W + W2 + wc];
        return w < W2 ? v : 0xFFFFFFFFu;
    };
    // guided self-scheduling: one atomic buys this wave a run of reads (many at the start, few near the end),
    // so 2 M reads cost thousands of atomics on `next_read`, not one per refill
    auto grab = [&]() {
        unsigned int base = 0, chunk = 0;
        if (lane == 0) {
            const unsigned int seen = *(volatile unsigned int *)next_read;
            const unsigned int left = seen < (unsigned int)n_reads ? (unsigned int)n_reads - seen : 0u;
            chunk = left / (2u * gridDim.x);
            chunk = chunk < (unsigned int)SEED_REFILL ? (unsigned int)SEED_REFILL : (chunk > 512u ? 512u : chunk);
            base = atomicAdd(next_read, chunk);
        }
        base = (unsigned int)__shfl((int)base, 0, 64); chunk = (unsigned int)__shfl((int)chunk, 0, 64);
        pool_next = base < (unsigned int)n_reads ? base : (unsigned int)n_reads;
        pool_end = base + chunk < (unsigned int)n_reads ? base + chunk : (unsigned int)n_reads;
        if (pool_next == pool_end) exhausted = true;
    };
    const unsigned int w_magic = (65536u + (unsigned int)W - 1u) / (unsigned int)W;      // i / W == (i * w_magic) >> 16 for i < 256
    const unsigned int refill_cap = 256u / (unsigned int)W ? 256u / (unsigned int)W : 1u;      // reads per prefetch batch (4 words per lane in flight); a division: once, not per event
    while (true) {
        const unsigned long long idle = __ballot(r < 0);
        bool event = false;
        if (idle) {
            event = (__popcll(idle) >= SEED_REFILL || idle == ~0ull) && (!exhausted || __ballot(r < 0 && r_nxt >= 0) != 0);
            if (!event && idle == ~0ull) break;          // nothing in work, nothing prefetched, nothing left to claim
        }
        wtrips++;
        bool live = false, finished = false, beginning = false;
        TripData t; t.aux = T_NONE;
        TripAddr ta = {nullptr, nullptr, nullptr, nullptr};
        // Per trip the wave serves the cheap modes of all its lanes (0 = begin: one table entry; 3 = locate: one SA
        // entry) and ONE of the two heavy ones (1 = Occ step, 2 = 64-base text comparison), whichever more lanes are
        // in; the lanes in the other heavy mode wait a trip.  Running every mode's code in every trip made the kernel
        // instruction-fetch bound (the I-cache of a CU pair was busy 85 % of the time); serving only the single fullest
        // mode left 58 % of the lanes idle per trip.
        uint32_t serve;                                       // bit m: mode m's code runs in this trip
        {
            const int n1 = __popcll(__ballot(r >= 0 && s.mode == 1)), n2 = __popcll(__ballot(r >= 0 && s.mode == 2));
            if (both_thr <= 0) serve = 0x9u | (n1 >= n2 ? 2u : 4u);
            else {                                            // experimental: a mode's code runs only for at least both_thr lanes (else the fullest mode)
                const int n0 = __popcll(__ballot(r >= 0 && s.mode == 0)), n3 = __popcll(__ballot(r >= 0 && s.mode == 3));
                serve = (n0 >= both_thr ? 1u : 0u) | (n1 >= both_thr ? 2u : 0u) | (n2 >= both_thr ? 4u : 0u) | (n3 >= both_thr ? 8u : 0u);
                if (!serve) {
                    const int m01 = n0 >= n1 ? n0 : n1, m23 = n2 >= n3 ? n2 : n3;
                    serve = m01 >= m23 ? (n0 >= n1 ? 1u : 2u) : (n2 >= n3 ? 4u : 8u);
                }
            }
        }
        // ---- issue phase: every chosen lane computes its address and issues its load, nobody waits ----
        if (r >= 0 && ((serve >> s.mode) & 1u)) {
            trips++;
            live = true;
            if (s.mode == 0) {                       // IdentifySeedPairs :191-211: next start
                while (pos < end_pos && d_at(rm, pos)) pos++;
                if (pos >= end_pos) finished = true;
                else if (nsearch >= SEED_BAIL || trips >= (uint32_t)bail_trips) {     // a long walk: let a whole wave finish this read
                    DHeavy hv; hv.read = (uint32_t)r; hv.pos = pos; hv.nh = nh; hv.ns = ns;
                    heavy[atomicAdd(n_heavy, 1u)] = hv;
                    max_trips = trips > max_trips ? trips : max_trips;
                    r = -1; live = false;
                } else { nsearch++; beginning = true; d_begin_issue(ix, K, rb, rm, pos, s, c, ta, t); }
            } else d_trip_issue(ix, rm, len, direct, s, c, ta, t);
        }
        d_trip_load(ta, t);
        // ---- refill event, placed where the wave has to wait for memory anyway ----
        if (event) {
            if (USE_LDS) {
                // (a) the batch prefetched at the previous event has long arrived: scatter it into the owners' `next` columns
                if (pend_total) {
                    const uint32_t pv[4] = {pv0, pv1, pv2, pv3};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned int i = (unsigned int)(k * 64 + lane);
                        if (i < pend_total) {
                            const unsigned int rk = (i * w_magic) >> 16, e = tab[rk];
                            sh[(((e >> 8) & 1u) * W + (i - rk * W)) * 64 + (e & 63u)] = pv[k];
                        }
                    }
                    pend_total = 0;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                // (b) idle lanes switch to their prefetched read: no memory access, no wait
                if (r < 0 && r_nxt >= 0) {
                    r = r_nxt; r_nxt = -1; cur ^= 1;
                    len = len_nxt; end_pos = len - 13; pos = pos_nxt; nh = nh_nxt; ns = ns_nxt; s.mode = 0; trips = 0; nsearch = 0;
                }
                // (c) lanes without a next read claim one; the wave issues coalesced loads for the whole batch (the
                //     reads are consecutive, so their words are one contiguous run of enc) and goes back to work
                const unsigned long long need = __ballot(r_nxt < 0);
                if (need && !exhausted) {
                    if (pool_next == pool_end) grab();
                    const unsigned int avail = pool_end - pool_next, cap = refill_cap;
                    unsigned int take = (unsigned int)__popcll(need);
                    take = take < avail ? take : avail; take = take < cap ? take : cap;
                    const unsigned int rank = (unsigned int)__popcll(need & ((1ull << lane) - 1ull));
                    if (r_nxt < 0 && rank < take) {
                        r_nxt = (int)(pool_next + rank); pos_nxt = 0; nh_nxt = 0; ns_nxt = 0;
                        tab[rank] = (uint32_t)lane | ((uint32_t)(cur ^ 1) << 8);
                        len_nxt = rlen[r_nxt];
                    }
                    const uint32_t *src = enc + (size_t)pool_next * W;
                    pend_total = take * (unsigned int)W;
                    auto fetch = [&](unsigned int i) -> uint32_t { return src[i]; };   // word i of the batch: consecutive reads are one contiguous run of enc
                    if ((unsigned int)lane < pend_total) pv0 = fetch((unsigned int)lane);
                    if ((unsigned int)lane + 64 < pend_total) pv1 = fetch((unsigned int)lane + 64);
                    if ((unsigned int)lane + 128 < pend_total) pv2 = fetch((unsigned int)lane + 128);
                    if ((unsigned int)lane + 192 < pend_total) pv3 = fetch((unsigned int)lane + 192);
                    pool_next += take;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            } else if (!exhausted) {
                if (pool_next == pool_end) grab();
                const unsigned long long idl = __ballot(r < 0);
                const unsigned int avail = pool_end - pool_next, n_idl = (unsigned int)__popcll(idl);
                const unsigned int rank = (unsigned int)__popcll(idl & ((1ull << lane) - 1ull));
                if (r < 0 && rank < avail) {
                    pos = 0; nh = 0; ns = 0;
                    r = (int)(pool_next + rank);
                    len = rlen[r]; end_pos = len - 13; s.mode = 0; trips = 0; nsearch = 0;
                }
                pool_next += n_idl < avail ? n_idl : avail;
            }
        }
        // ---- finish phase ----
        if (live) {
            if (beginning) d_begin_finish(ix, K, rb, s, c, t);
            else if (t.aux != T_NONE) d_trip_finish(ix, pr, rb, rm, len, s, c, t);
            if (direct && s.mode == 1 && s.x2 == 1) { s.mode = 3; s.lk = s.x0; s.lsteps = 0; }   // unique: locate, then compare with the text
            if (!finished && s.mode == 0) {          // a search just ended (or the table said "absent")
                c.steps += s.ref_steps; c.blocks += s.ref_blocks;
                if (s.hit_len) {
                    if (nh < H) {
                        DHit h; h.rPos = (uint16_t)s.start; h.len = (uint16_t)s.hit_len;
                        if (s.located) { h.x0 = (uint64_t)s.tpos; h.freq = 1u | 0x80000000u; c.lf_ref += s.lsteps + (uint32_t)(s.lk >> 40); }
                        else { h.x0 = s.x0; h.freq = (uint32_t)s.x2; }
                        hits[(size_t)r * H + nh] = h;
                    }
                    nh++; ns += (uint32_t)s.x2;
                    pos = s.start + s.hit_len;
                } else pos = s.start + 1;
            }
            if (finished) { nhits[r] = (uint32_t)nh; nseeds[r] = ns; r = -1; max_trips = trips > max_trips ? trips : max_trips; }
        }
    }
    atomicMax(d_ctr_stripe(ctr) + CTR_MAXTRIPS, (unsigned long long)max_trips);
    if (lane == 0) { atomicMax(d_ctr_stripe(ctr) + CTR_WTRIPS_MAX, (unsigned long long)wtrips); atomicAdd(d_ctr_stripe(ctr) + CTR_WTRIPS_SUM, (unsigned long long)wtrips); }
    d_wave_add(ctr + CTR_STEPS, c.steps);
    d_wave_add(ctr + CTR_BLOCKS, c.blocks);
    d_wave_add(ctr + CTR_STEPS_ACT, c.steps_act);
    d_wave_add(ctr + CTR_BLOCKS_ACT, c.blocks_act);
    d_wave_add(ctr + CTR_KTAB, c.ktab);
    d_wave_add(ctr + CTR_LF, c.lf_ref);
    d_wave_add(ctr + CTR_LF_ACT, c.lf_act);
    d_wave_add(ctr + CTR_DIRECT, c.n_direct);
}

// ---------------------------------------------------------------------------------------------
// k_seed_heavy_walk: lane = one read of k_seed_heavy's list.
// The tail of a read inside a high-copy repeat (satellites, microsatellites, young interspersed copies): the reference restarts its search at
// every base (a failed search advances by one, AlignmentCandidates.cpp:209) and each of those searches runs to the end of the read --
// O(rlen^2) Occ steps, two thirds of k_seed_heavy's work on a human-like genome.  A search from s that matches through to the read's end e
// stops there with the interval of read[s, e) (bwt_search.cpp:152-171), and that interval only grows as s moves right.  So ONE backward
// walk from e (prepending a base = extending the reverse complement on the right: the text holds both strands) finds the leftmost s_f
// whose read[s_f, e) occurs more than max_dup times -- and at least 128 times, so that every step of those searches fetched two Occ
// blocks, which keeps the reference-equivalent block count exact: every start in [s_f, e) fails (bwt_search.cpp:173) after e - s - 1 steps.
// Round 4 made this walk inside k_seed_heavy, once per read by the whole wave: its values are wave-uniform, so the compiler turned it into ~80 scalar
// instructions per step -- up to ~85 steps per read, 0.37 G instructions per 2 M reads of the human-like genome (a quarter of the batch's, 64 % of
// them SALU) for one lane's worth of work.  Here 64 reads share an instruction.  sfail_out[hi] = s_f (the read's length: nothing fails for free).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_seed_heavy_walk(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int W,
                  const DHeavy *__restrict__ heavy, const unsigned int *__restrict__ n_heavy_p, int32_t *__restrict__ sfail_out, unsigned long long *ctr)
{
    const int W2 = W >> 1;
    const unsigned int n_heavy = *n_heavy_p;
    SeedCtr c = {0, 0, 0, 0, 0, 0, 0, 0};
    const int K = ix.ktab ? ix.ktab_k : 0;
    for (unsigned int hi = blockIdx.x * blockDim.x + threadIdx.x; hi < n_heavy; hi += gridDim.x * blockDim.x) {
        const DHeavy hv = heavy[hi];
        const int r = (int)hv.read, len = rlen[r], pos = hv.pos;
        const uint32_t *ew = enc + (size_t)r * W;
        auto rb = [&](int w) -> uint32_t { const uint32_t v = ew[w < W2 ? w : W2 - 1]; return w < W2 ? v : 0u; };
        auto rm = [&](int w) -> uint32_t { const uint32_t v = ew[W2 + (w < W2 ? w : W2 - 1)]; return w < W2 ? v : 0xFFFFFFFFu; };
        int sfail = len;
        {
            const uint64_t thr = (uint64_t)(pr.max_dup + 1 > 128 ? pr.max_dup + 1 : 128);
            int q = len - 1;
            if (q >= pos && d_at(rm, q) == 0) {
                uint64_t x0 = 0, x1 = 0, x2 = 0;
                bool have = false;
                if (K && len - K >= pos) {
                    const uint32_t sft = 32u - 2u * (uint32_t)K;
                    if ((d_win16(rm, len - K) >> sft) == 0) {           // the read's last K bases, none of them N
                        const uint4 e4 = *(const uint4_a4 *)(ix.ktab + (size_t)(d_win16(rb, len - K) >> sft) * 2);
                        const uint64_t w0 = d_u64(e4.x, e4.y), w1 = d_u64(e4.z, e4.w);
                        c.ktab++;
                        if (!(w1 >> 63)) {                               // (an overflowing entry: start from the last base instead)
                            have = true;
                            if ((w1 >> 62) & 1ull) x2 = 1;               // a located (unique) K-mer
                            else { x2 = (w1 >> 16) & 0x7FFFFFFFull; x0 = w0 & 0xFFFFFFFFFFull; x1 = (w0 >> 40) | ((w1 & 0xFFFFull) << 24); q = len - K; }
                        }
                    }
                }
                if (!have) { const int cc = (int)d_at(rb, q); x0 = d_L2(ix, cc) + 1; x1 = d_L2(ix, 3 - cc) + 1; x2 = d_L2(ix, cc + 1) - d_L2(ix, cc); }
                if (x2 >= thr) {
                    sfail = q;
                    while (q > pos) {
                        const int qq = q - 1;
                        if (d_at(rm, qq)) break;
                        uint64_t y0 = x1, y1 = x0, y2 = x2;              // the bi-interval of the reverse complement
                        uint32_t nb;
                        const bool ok = d_extend(ix, 3 - (int)d_at(rb, qq), y0, y1, y2, nb);
                        c.steps_act++; c.blocks_act += nb;
                        if (!ok || y2 < thr) break;
                        x1 = y0; x0 = y1; x2 = y2; q = qq; sfail = q;
                    }
                }
            }
        }
        sfail_out[hi] = sfail;
    }
    d_wave_add(ctr + CTR_STEPS_ACT, c.steps_act);
    d_wave_add(ctr + CTR_BLOCKS_ACT, c.blocks_act);
    d_wave_add(ctr + CTR_KTAB, c.ktab);
}

// ---------------------------------------------------------------------------------------------
// k_seed_heavy: one wave = one read whose greedy walk is long (almost every start fails: an
// unmappable or very noisy read).  BWT_Search(start) does not depend on earlier searches -- only
// WHICH starts get searched does -- so the wave evaluates 64 consecutive starts at once, one per
// lane, and then replays the reference's walk (hit: pos += len, else pos++; :198-209) over the 64
// results with shuffles.  Results identical; only searches the walk really visits are charged to
// the reference-equivalent counters (the speculative ones count as executed work only).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_seed_heavy(const DIndex ix, const DParams pr, const uint32_t *__restrict__ enc, const uint16_t *__restrict__ rlen, int W, int H,
             DHit *__restrict__ hits, uint32_t *__restrict__ nhits, uint32_t *__restrict__ nseeds,
             const DHeavy *__restrict__ heavy, const unsigned int *__restrict__ n_heavy_p, const int32_t *__restrict__ sfail_in, unsigned long long *ctr)
{
    const unsigned long long t_wave0 = wall_clock64();
    extern __shared__ uint32_t sh[];                 // the read's words (k_encode format), shared by the wave
    const int W2 = W >> 1;
    const int lane = threadIdx.x;
    const unsigned int n_heavy = *n_heavy_p;
    SeedCtr c = {0, 0, 0, 0, 0, 0, 0, 0};
    const int K = ix.ktab ? ix.ktab_k : 0;
    const bool direct = ix.sa_dense != nullptr;
    for (unsigned int hi = blockIdx.x; hi < n_heavy; hi += gridDim.x) {
        const DHeavy hv = heavy[hi];
        const int r = (int)hv.read, len = rlen[r], end_pos = len - 13;
        __syncthreads();
        for (int w = lane; w < W; w += 64) sh[w] = enc[(size_t)r * W + w];
        __syncthreads();
        auto rb = [&](int w) -> uint32_t { const uint32_t v = sh[w < W2 ? w : W2 - 1]; return w < W2 ? v : 0u; };
        auto rm = [&](int w) -> uint32_t { const uint32_t v = sh[W2 + (w < W2 ? w : W2 - 1)]; return w < W2 ? v : 0xFFFFFFFFu; };
        int pos = hv.pos, nh = hv.nh;
        uint32_t ns = hv.ns;
        const int sfail = sfail_in[hi];              // from where on every start fails without being searched (k_seed_heavy_walk, below the comment there)
        while (pos < end_pos) {                      // uniform
            if (pos >= sfail) {                      // every remaining start fails after len - s - 1 steps: the sum over s = pos .. end_pos - 1
                if (lane == 0) {
                    const unsigned long long n = (unsigned long long)(end_pos - pos), st2 = n * (unsigned long long)((len - pos - 1) + (len - end_pos));
                    c.steps += st2 / 2; c.blocks += st2;
                }
                pos = end_pos;
                break;
            }
            Search s; s.mode = 0; s.hit_len = 0; s.located = false; s.ref_steps = s.ref_blocks = 0; s.start = pos + lane; s.x2 = 0; s.x0 = 0; s.tpos = 0; s.lsteps = 0; s.lk = 0;
            const int st = pos + lane;
            const bool doomed = st >= sfail && st < end_pos;             // (fails without being searched)
            const bool acgt = st < end_pos && st < sfail && d_at(rm, st) == 0;
            {   // the 64 searches advance in lock step: one issue phase, one finish phase per trip
                TripData t; t.aux = T_NONE;
                TripAddr ta = {nullptr, nullptr, nullptr, nullptr};
                if (acgt) d_begin_issue(ix, K, rb, rm, st, s, c, ta, t);
                d_trip_load(ta, t);
                if (acgt) d_begin_finish(ix, K, rb, s, c, t);
                if (acgt && direct && s.mode == 1 && s.x2 == 1) { s.mode = 3; s.lk = s.x0; s.lsteps = 0; }
                while (__ballot(acgt && s.mode != 0)) {
                    t.aux = T_NONE; ta.pa = ta.pb = ta.p16 = nullptr; ta.p8 = nullptr;
                    // as in k_seed: the cheap locate mode every trip, and the fuller of the two heavy modes
                    const int sel = __popcll(__ballot(acgt && s.mode == 1)) >= __popcll(__ballot(acgt && s.mode == 2)) ? 1 : 2;
                    if (acgt && (s.mode == 3 || s.mode == sel)) d_trip_issue(ix, rm, len, direct, s, c, ta, t);
                    d_trip_load(ta, t);
                    if (t.aux != T_NONE) d_trip_finish(ix, pr, rb, rm, len, s, c, t);
                    if (acgt && direct && s.mode == 1 && s.x2 == 1) { s.mode = 3; s.lk = s.x0; s.lsteps = 0; }
                }
            }
            // Replay the reference's walk (hit: pos += len, else pos++; AlignmentCandidates.cpp:198-209) over the 64 results.  Only the starts with a hit
            // change where the walk goes, so it jumps from hit to hit (round 5; rounds 1-4 stepped through the starts one by one with shuffles: ~40 scalar
            // instructions x 64 per round, two thirds of this kernel's 0.37 G instructions on a genome with human-like repeats) and marks what it
            // passed over; the visited lanes then add their own counters, the visited hits take their places by popcount.
            const int nlim = (pos + 64 < end_pos ? pos + 64 : end_pos) - pos;          // starts of this round (1 .. 64)
            const unsigned long long in_round = nlim >= 64 ? ~0ull : ((1ull << nlim) - 1ull);
            const unsigned long long hitm = __ballot(acgt && s.hit_len > 0) & in_round;
            unsigned long long vis = 0;
            int cur = 0;                             // relative to pos
            while (cur < nlim) {                     // uniform; one iteration per visited hit
                const unsigned long long m = hitm >> cur;
                if (!m) { vis |= in_round & ~((1ull << cur) - 1ull); cur = nlim; break; }
                const int j = cur + (__ffsll((long long)m) - 1);
                vis |= (j >= 63 ? ~0ull : ((1ull << (j + 1)) - 1ull)) & ~((1ull << cur) - 1ull);
                cur = j + __shfl(s.hit_len, j, 64);
            }
            const bool seen = (vis >> lane) & 1ull;
            if (seen && doomed) { c.steps += (unsigned long long)(len - st - 1); c.blocks += 2ull * (unsigned long long)(len - st - 1); }
            if (seen && acgt) { c.steps += s.ref_steps; c.blocks += s.ref_blocks; }      // a search the reference performs
            const unsigned long long vh = vis & hitm;
            if ((vh >> lane) & 1ull) {
                const int at = nh + __popcll(vh & ((1ull << lane) - 1ull));
                if (at < H) {
                    DHit h; h.rPos = (uint16_t)s.start; h.len = (uint16_t)s.hit_len;
                    if (s.located) { h.x0 = (uint64_t)s.tpos; h.freq = 1u | 0x80000000u; c.lf_ref += s.lsteps + (uint32_t)(s.lk >> 40); }
                    else { h.x0 = s.x0; h.freq = (uint32_t)s.x2; }
                    hits[(size_t)r * H + at] = h;
                }
            }
            uint32_t fr = ((vh >> lane) & 1ull) ? (uint32_t)s.x2 : 0u;
            for (int o = 32; o > 0; o >>= 1) fr += (uint32_t)__shfl_xor((int)fr, o, 64);
            nh += __popcll(vh); ns += fr;
            cur += pos;
            pos = cur;
        }
        if (lane == 0) { nhits[r] = (uint32_t)nh; nseeds[r] = ns; }
    }
    d_wave_add(ctr + CTR_STEPS, c.steps);
    d_wave_add(ctr + CTR_BLOCKS, c.blocks);
    d_wave_add(ctr + CTR_STEPS_ACT, c.steps_act);
    d_wave_add(ctr + CTR_BLOCKS_ACT, c.blocks_act);
    d_wave_add(ctr + CTR_KTAB, c.ktab);
    d_wave_add(ctr + CTR_LF, c.lf_ref);
    d_wave_add(ctr + CTR_LF_ACT, c.lf_act);
    d_wave_add(ctr + CTR_DIRECT, c.n_direct);
    d_wave_resident(ctr, CTR_WT_SEEDH, t_wave0);
}

// ---------------------------------------------------------------------------------------------
// k_build_sa_dense (once per dg_init): SA of every `intv`-th row (intv = 1: the full suffix array),
// derived from the reference's SA/32 so that every entry is exactly what the reference's walk
// (bwt_sa :127-137: LF steps to the next sampled row, then sa[] + steps) returns -- and each entry
// also remembers that step count (algorithmic-byte accounting).
// One thread per SAMPLED row r0: the rows LF(r0), LF^2(r0), ... up to the next sampled row r_T are
// exactly the rows whose reference walk ends at r_T, after T-1, T-2, ... steps.  So the chain is
// walked twice (once to find T and r_T, once to write) -- 2 LF evaluations per row instead of the
// 15.5 on average that a walk from every row costs (GRCh38-sized: 3.3 s -> see DESIGN.md).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_build_sa_dense(const DIndex ix, int intv, uint64_t n_entries, uint64_t *__restrict__ dense)
{
    const uint64_t smask = (uint64_t)ix.sa_intv - 1, dmask = (uint64_t)intv - 1;
    const uint64_t n_sampled = (ix.seq_len + (uint64_t)ix.sa_intv) / (uint64_t)ix.sa_intv;      // rows 0, 32, 64, ... <= seq_len
    int dsh = 0;
    while ((1 << dsh) < intv) dsh++;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_sampled; q += (uint64_t)gridDim.x * blockDim.x) {   // grid-stride: 2^32 work-items do not fit a launch
        const uint64_t r0 = q * (uint64_t)ix.sa_intv;
        if ((r0 >> dsh) < n_entries) dense[r0 >> dsh] = ((ix.sa[q] + 1) & 0xFFFFFFFFFFull);    // a sampled row: 0 steps (sa[0] = -1 wraps as in the reference)
        uint64_t k = d_lf(ix, r0), T = 1;
        while (k & smask) { k = d_lf(ix, k); T++; }
        const uint64_t base = ix.sa[k / (uint64_t)ix.sa_intv];
        k = d_lf(ix, r0);
        for (uint64_t t = 1; t < T; t++) {
            const uint64_t steps = T - t;
            if (!(k & dmask) && (k >> dsh) < n_entries) dense[k >> dsh] = ((steps + base + 1) & 0xFFFFFFFFFFull) | (steps << 40);
            k = d_lf(ix, k);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_locate: SA interval rows -> text positions (bwt_sa :127-137) -> seeds.
// One wave = 64 consecutive seed OCCURRENCES (a repeat-family read has hundreds, a clean read one
// or two: per-read work units leave the kernel waiting for its heaviest wave).  k_seed_offsets (dg_api.hip)
// records, for every tile of 64 seeds, the read that holds the tile's first seed; the wave loads
// the scan values of the next 64 reads with one coalesced load and each lane finds its read with a
// 6-step shuffle search (windows of 64 reads are walked when a tile spans more, i.e. across reads
// without seeds), then its hit among the read's <= H hits.  Seeds come out coalesced, in
// (read, hit, SA-interval row) order as in the reference.
// ---------------------------------------------------------------------------------------------

// grid = enough waves for the buffer's capacity; the real seed count is read from the scan (seed_off[n_reads])
__global__ void __launch_bounds__(256)
k_locate(const DIndex ix, int n_reads, int H, const DHit *__restrict__ hits, const uint32_t *__restrict__ tile_read,
         const uint32_t *__restrict__ seed_off, SKey *__restrict__ seeds, unsigned long long *ctr, const int *__restrict__ abort_p)
{
    if (*abort_p >= DG_ABORT) return;
    const uint32_t total = seed_off[n_reads];
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t u = (tile << 6) + (uint32_t)lane;
    unsigned long long lf = 0, lf_act = 0, nsa = 0;
    if ((tile << 6) < total) {                               // wave-uniform
        int r0 = (int)tile_read[tile];
        const uint32_t u_last = (tile << 6) + 63 < total ? (tile << 6) + 63 : total - 1;
        int r = -1;
        uint32_t first = 0;
        while (true) {                                       // uniform: windows of 64 reads
            const int rr = r0 + lane;
            const uint32_t nxt = rr + 1 <= n_reads ? seed_off[rr + 1] : 0xFFFFFFFFu;   // seeds before read rr+1
            // number of reads in the window that end at or before u = offset of u's read in the window
            int lo = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint32_t v = __shfl(nxt, lo + step - 1, 64);
                if (v <= u) lo += step;
            }
            if (__shfl(nxt, lo, 64) <= u) lo++;              // lo = 64: u's read lies beyond this window
            const uint32_t win_end = __shfl(nxt, 63, 64);    // seeds before read r0+64
            if (r < 0 && u < total && lo < 64) { r = r0 + lo; first = seed_off[r]; }
            if (win_end > u_last) break;
            r0 += 64;
        }
        if (u < total) {
            uint32_t w = u - first;
            const DHit *h = hits + (size_t)r * H;
            uint32_t fr = h->freq;
            while (w >= (fr & 0x7FFFFFFFu)) { w -= fr & 0x7FFFFFFFu; h++; fr = h->freq; }
            uint64_t k = h->x0 + w;
            uint64_t steps = 0, pos;
            if (fr & 0x80000000u) pos = h->x0;               // k_seed already located this unique hit
            else if (ix.sa_dense) {
                const uint64_t mask = (uint64_t)ix.sa_dense_intv - 1;
                while (k & mask) { k = d_lf(ix, k); steps++; }
                const uint64_t e = ix.sa_dense[k >> ix.sa_dense_shift];
                pos = steps + (e & 0xFFFFFFFFFFull) - 1;
                lf += steps + (e >> 40); lf_act += steps;
            } else {
                const uint64_t mask = (uint64_t)ix.sa_intv - 1;
                while (k & mask) { k = d_lf(ix, k); steps++; }
                pos = steps + ix.sa[k / (uint64_t)ix.sa_intv];
                lf += steps; lf_act += steps;
            }
            nsa++;
            seeds[u] = sk_make((int64_t)pos, (int)h->rPos, (int)h->len);
        }
    }
    d_wave_add(ctr + CTR_LF, lf);
    d_wave_add(ctr + CTR_LF_ACT, lf_act);
    d_wave_add(ctr + CTR_SA, nsa);
}
