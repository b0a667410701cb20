// dart_amd/csrc/index/dg_index.hip -> dart_amd/libdartindex.so: the device side of the offline indexer (include/dartindex.h).
//
// Replaces, for texts the GPU box has to index itself, what the reference's `bwt_index` computes (BWT_Index/bwtindex.c:77-148):
// the suffix order of forward + reverse complement (bwt_gen.c / QSufSort.c there; bucketed prefix doubling here) and the .bwt
// body with its interleaved Occ counters (bwtindex.c:53-75).  dart_amd/index_build.py drives these kernels and the radix sorter
// (dg_sort_pairs, ../dg_sort.h); the files it writes are compared byte for byte with the reference indexer's
// (tests/test_gpu_index.py).  gfx950 only.  Everything here is HBM-bound integer work: no MFMA.
//
// The suffix sorter, as a whole (N = n + 1 suffixes, 6.2 G at GRCh38 size):
//   round 0   the suffixes are split by their first two symbols into 16 buckets (plus the two singletons that meet the '$'
//             at once); a bucket's members are listed in text order together with a 63-bit key = the next 29 symbols and how
//             many of them exist, one radix sort orders them by their first 31 symbols, and k_grp_* turn the sorted keys into
//             ranks (= the row of the first suffix with the same key) and the list of rows that still share a key;
//   round r   only those rows are touched again: key = (own rank, rank of the suffix k symbols further on), k = 31, 62, ...:
//             the same sort, the same k_grp_*.  Ranks are refined in place (a row's new rank never contradicts the true order,
//             so a later bucket of the same round may already see it).
// Memory: sa and rank (2 x 8 N bytes), the text (N / 4), and four bucket-sized arrays for the sorter.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../../include/dartindex.h"

static thread_local char g_err[256] = "";
extern "C" const char *di_last_error(void) { return g_err; }
static int hip_fail(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -2;
}
#define DI_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(#x, e_); } while (0)
#define DI_DONE(name) do { hipError_t e_ = hipGetLastError(); if (e_ == hipSuccess) e_ = hipStreamSynchronize(0); if (e_ != hipSuccess) return hip_fail(name, e_); return 0; } while (0)

extern "C" size_t di_text_words(uint64_t n) { return (size_t)((n + 31) / 32 + 2); }

// 32 symbols from position p on, first symbol in the top bits
__device__ __forceinline__ uint64_t d_sym32(const uint64_t *__restrict__ T, uint64_t p)
{
    const uint64_t w = p >> 5; const uint32_t o = (uint32_t)(p & 31u) * 2u;
    const uint64_t hi = T[w], lo = T[w + 1];
    return o ? (hi << o) | (lo >> (64u - o)) : hi;
}
__device__ __forceinline__ uint32_t d_sym(const uint64_t *__restrict__ T, uint64_t p)
{
    return (uint32_t)(T[p >> 5] >> (62u - 2u * (uint32_t)(p & 31u))) & 3u;
}

// ------------------------------------------------------------------------------------------ the text
__global__ void __launch_bounds__(256) k_pack_text(const uint8_t *__restrict__ fwd, uint64_t L, uint64_t *__restrict__ T, uint64_t words)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (w >= words) return;
    uint64_t v = 0;
    const uint64_t p0 = w * 32u, n = 2 * L;
#pragma unroll 8
    for (int j = 0; j < 32; j++) {
        const uint64_t p = p0 + (uint64_t)j;
        uint32_t c = 0;
        if (p < L) c = fwd[p] & 3u;
        else if (p < n) c = 3u - (fwd[n - 1 - p] & 3u);
        v = (v << 2) | c;
    }
    T[w] = v;
}
extern "C" int di_pack_text(int device, const uint8_t *fwd, uint64_t l_pac, uint64_t *T)
{
    if (!fwd || !T || l_pac == 0) { snprintf(g_err, sizeof g_err, "di_pack_text: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    const uint64_t words = di_text_words(2 * l_pac);
    k_pack_text<<<(uint32_t)((words + 255) / 256), 256, 0, 0>>>(fwd, l_pac, T, words);
    DI_DONE("k_pack_text");
}

// ------------------------------------------------------------------------------------------ round 0: buckets and keys
__global__ void __launch_bounds__(256) k_bucket_hist(const uint64_t *__restrict__ T, uint64_t n, uint32_t tiles, uint32_t *__restrict__ table)
{
    __shared__ uint32_t lh[4][16];
    if (threadIdx.x < 64) lh[threadIdx.x >> 4][threadIdx.x & 15] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * DI_TILE;
    const uint32_t wv = threadIdx.x >> 6;
    for (int s = 0; s < 16; s++) {
        const uint64_t i = base + (uint32_t)s * 256u + threadIdx.x;
        if (i + 2 <= n) atomicAdd(&lh[wv][d_sym32(T, i) >> 60], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) table[(size_t)threadIdx.x * tiles + blockIdx.x] = lh[0][threadIdx.x] + lh[1][threadIdx.x] + lh[2][threadIdx.x] + lh[3][threadIdx.x];
}
extern "C" int di_bucket_hist(int device, const uint64_t *T, uint64_t n, uint32_t *table)
{
    if (!T || !table || n < 2) { snprintf(g_err, sizeof g_err, "di_bucket_hist: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    const uint64_t tiles = (n + 1 + DI_TILE - 1) / DI_TILE;
    if (tiles >= 0x7FFFFFFFull) { snprintf(g_err, sizeof g_err, "di_bucket_hist: text too long"); return -1; }
    k_bucket_hist<<<(uint32_t)tiles, 256, 0, 0>>>(T, n, (uint32_t)tiles, table);
    DI_DONE("k_bucket_hist");
}

// a wave owns 1024 consecutive positions of the tile (16 steps of 64), so the members leave in text order
__global__ void __launch_bounds__(256) k_bucket_keys(const uint64_t *__restrict__ T, uint64_t n, uint32_t pair, const uint32_t *__restrict__ tile_base,
                                                     uint64_t *__restrict__ keys, int64_t *__restrict__ vals)
{
    __shared__ uint32_t wtot[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * DI_TILE + (uint64_t)wv * 1024u;
    uint64_t msk[16];
    uint32_t total = 0;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const uint64_t i = base + (uint32_t)s * 64u + lane;
        const bool hit = i + 2 <= n && (uint32_t)(d_sym32(T, i) >> 60) == pair;
        msk[s] = __ballot(hit);
        total += (uint32_t)__popcll(msk[s]);
    }
    if (lane == 0) wtot[wv] = total;
    __syncthreads();
    uint32_t out = tile_base[blockIdx.x];
    for (uint32_t w = 0; w < wv; w++) out += wtot[w];
    const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const uint64_t i = base + (uint32_t)s * 64u + lane;
        if ((msk[s] >> lane) & 1ull) {
            const uint64_t v = d_sym32(T, i);                                 // symbols i .. i + 31
            const uint64_t left = n - i - 2;                                  // real symbols behind the pair
            const uint64_t f = left < 29 ? left : 29;
            const uint32_t o = out + (uint32_t)__popcll(msk[s] & below);
            keys[o] = (((v << 4) >> 6) << 5) | f;
            vals[o] = (int64_t)i;
        }
        out += (uint32_t)__popcll(msk[s]);
    }
}
extern "C" int di_bucket_keys(int device, const uint64_t *T, uint64_t n, int pair, const uint32_t *tile_base, uint64_t *keys, int64_t *vals)
{
    if (!T || !tile_base || !keys || !vals || n < 2 || pair < 0 || pair > 15) { snprintf(g_err, sizeof g_err, "di_bucket_keys: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    const uint64_t tiles = (n + 1 + DI_TILE - 1) / DI_TILE;
    k_bucket_keys<<<(uint32_t)tiles, 256, 0, 0>>>(T, n, (uint32_t)pair, tile_base, keys, vals);
    DI_DONE("k_bucket_keys");
}

// ------------------------------------------------------------------------------------------ later rounds: the rank pair
__global__ void __launch_bounds__(256) k_doubling_keys(const int64_t *__restrict__ sa, const int64_t *__restrict__ rank, uint64_t lo, const uint32_t *__restrict__ pos,
                                                       uint32_t m, uint64_t k, uint64_t N, int r2_bits, uint64_t *__restrict__ keys, int64_t *__restrict__ vals)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= m) return;
    const uint64_t s = (uint64_t)sa[lo + pos[j]];
    const uint64_t r1 = (uint64_t)rank[s] - lo;
    const uint64_t nxt = s + k;
    const uint64_t r2 = nxt < N ? (uint64_t)rank[nxt] + 1u : 0u;
    keys[j] = (r1 << r2_bits) | r2;
    vals[j] = (int64_t)s;
}
extern "C" int di_doubling_keys(int device, const int64_t *sa, const int64_t *rank, uint64_t lo, const uint32_t *pos, uint32_t m, uint64_t k, uint64_t N,
                                int r2_bits, uint64_t *keys, int64_t *vals)
{
    if (!sa || !rank || !pos || !keys || !vals || r2_bits < 1 || r2_bits > 62) { snprintf(g_err, sizeof g_err, "di_doubling_keys: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    if (m == 0) return 0;
    k_doubling_keys<<<(m + 255u) / 256u, 256, 0, 0>>>(sa, rank, lo, pos, m, k, N, r2_bits, keys, vals);
    DI_DONE("k_doubling_keys");
}

// ------------------------------------------------------------------------------------------ sorted keys -> rows, ranks, what is still tied
// A lane owns 16 consecutive pairs of the tile.  "head" = the pair's key differs from its predecessor's (pair 0 is a head, and so
// is the virtual pair m); a pair is tied unless it and its successor are both heads.  The value carried for the ranks is
// P(head) + 1 (0 = no head seen yet), a running maximum since P ascends.
struct GrpLane { uint32_t heads, tied; };          // bit q: pair q of the lane
__device__ __forceinline__ GrpLane d_grp_flags(const uint64_t *__restrict__ keys, uint32_t m, uint32_t j0)
{
    GrpLane g = {0u, 0u};
    if (j0 >= m) return g;
    uint64_t prev = j0 ? keys[j0 - 1] : 0;
    uint32_t heads = 0;                             // bit q for q = 0..16 (16 = the successor of the lane's last pair)
#pragma unroll
    for (int q = 0; q <= 16; q++) {
        const uint32_t j = j0 + (uint32_t)q;
        bool h = true;
        if (j < m) { const uint64_t k = keys[j]; h = j == 0 || k != prev; prev = k; }
        heads |= (uint32_t)h << q;
        if (j >= m) break;                          // pair m is the virtual head; nothing behind it
    }
    uint32_t present = m - j0 >= 16 ? 0xFFFFu : (1u << (m - j0)) - 1u;
    g.heads = heads & present;
    g.tied = ~(heads & (heads >> 1)) & present;
    return g;
}
__device__ __forceinline__ uint32_t d_P(const uint32_t *__restrict__ pos, uint32_t j) { return pos ? pos[j] : j; }

__global__ void __launch_bounds__(256) k_grp_a(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ pos, uint32_t m,
                                               uint32_t *__restrict__ tile_last, uint32_t *__restrict__ tile_tied)
{
    __shared__ uint32_t s_last[4], s_tied[4];
    const uint32_t j0 = blockIdx.x * DI_TILE + threadIdx.x * 16u;
    const GrpLane g = d_grp_flags(keys, m, j0);
    uint32_t last = g.heads ? d_P(pos, j0 + (31u - (uint32_t)__clz(g.heads))) + 1u : 0u;
    uint32_t tied = (uint32_t)__popc(g.tied);
    for (int o = 32; o; o >>= 1) { last = max(last, (uint32_t)__shfl_xor((int)last, o, 64)); tied += (uint32_t)__shfl_xor((int)tied, o, 64); }
    if ((threadIdx.x & 63u) == 0) { s_last[threadIdx.x >> 6] = last; s_tied[threadIdx.x >> 6] = tied; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_last[blockIdx.x] = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
        tile_tied[blockIdx.x] = s_tied[0] + s_tied[1] + s_tied[2] + s_tied[3];
    }
}

// one workgroup: exclusive running maximum of tile_last, exclusive running sum of tile_tied, in place; the sum's total to *n_tied
__global__ void __launch_bounds__(256) k_grp_b(uint32_t *__restrict__ tile_last, uint32_t *__restrict__ tile_tied, uint32_t tiles, uint32_t *__restrict__ n_tied)
{
    __shared__ uint32_t s_mx[256], s_sm[256];
    __shared__ uint32_t c_mx, c_sm;
    if (threadIdx.x == 0) { c_mx = 0; c_sm = 0; }
    __syncthreads();
    for (uint32_t b = 0; b < tiles; b += 256u) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t mx = i < tiles ? tile_last[i] : 0u, sm = i < tiles ? tile_tied[i] : 0u;
        s_mx[threadIdx.x] = mx; s_sm[threadIdx.x] = sm;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const uint32_t a = threadIdx.x >= (uint32_t)o ? s_mx[threadIdx.x - o] : 0u, c = threadIdx.x >= (uint32_t)o ? s_sm[threadIdx.x - o] : 0u;
            __syncthreads();
            s_mx[threadIdx.x] = max(s_mx[threadIdx.x], a); s_sm[threadIdx.x] += c;
            __syncthreads();
        }
        if (i < tiles) {
            tile_last[i] = max(c_mx, threadIdx.x ? s_mx[threadIdx.x - 1] : 0u);
            tile_tied[i] = c_sm + s_sm[threadIdx.x] - sm;
        }
        __syncthreads();
        if (threadIdx.x == 255) { c_mx = max(c_mx, s_mx[255]); c_sm += s_sm[255]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_tied = c_sm;
}

__global__ void __launch_bounds__(256) k_grp_c(const uint64_t *__restrict__ keys, const int64_t *__restrict__ vals, const uint32_t *__restrict__ pos, uint32_t m, uint64_t lo,
                                               const uint32_t *__restrict__ tile_last, const uint32_t *__restrict__ tile_tied,
                                               int64_t *__restrict__ rank, int64_t *__restrict__ sa, uint32_t *__restrict__ new_pos)
{
    __shared__ uint32_t s_last[4], s_tied[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t j0 = blockIdx.x * DI_TILE + threadIdx.x * 16u;
    const GrpLane g = d_grp_flags(keys, m, j0);
    const uint32_t n_here = j0 >= m ? 0u : (m - j0 >= 16u ? 16u : m - j0);
    uint32_t P[16];
#pragma unroll
    for (int q = 0; q < 16; q++) P[q] = (uint32_t)q < n_here ? d_P(pos, j0 + (uint32_t)q) : 0u;
    uint32_t my_last = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) if ((g.heads >> q) & 1u) my_last = P[q] + 1u;
    const uint32_t my_tied = (uint32_t)__popc(g.tied);
    // inclusive scans over the wave (maximum / sum), then the exclusive values of this lane
    uint32_t in_mx = my_last, in_sm = my_tied;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = (uint32_t)__shfl_up((int)in_mx, o, 64), c = (uint32_t)__shfl_up((int)in_sm, o, 64);
        if (lane >= (uint32_t)o) { in_mx = max(in_mx, a); in_sm += c; }
    }
    if (lane == 63) { s_last[wv] = in_mx; s_tied[wv] = in_sm; }
    uint32_t ex_mx = (uint32_t)__shfl_up((int)in_mx, 1, 64), ex_sm = in_sm - my_tied;
    if (lane == 0) ex_mx = 0;
    __syncthreads();
    uint32_t cur = max(tile_last[blockIdx.x], ex_mx), at = tile_tied[blockIdx.x] + ex_sm;
    for (uint32_t w = 0; w < wv; w++) { cur = max(cur, s_last[w]); at += s_tied[w]; }
#pragma unroll
    for (int q = 0; q < 16; q++) {
        if ((uint32_t)q < n_here) {
            if ((g.heads >> q) & 1u) cur = P[q] + 1u;
            const int64_t s = vals[j0 + (uint32_t)q];
            sa[lo + P[q]] = s;
            rank[s] = (int64_t)(lo + (cur - 1u));
            if ((g.tied >> q) & 1u) new_pos[at++] = P[q];
        }
    }
}
extern "C" int di_regroup(int device, const uint64_t *keys, const int64_t *vals, const uint32_t *pos, uint32_t m, uint64_t lo,
                          int64_t *rank, int64_t *sa, uint32_t *new_pos, uint32_t *scratch, uint32_t *n_tied)
{
    if (!keys || !vals || !rank || !sa || !new_pos || !scratch || !n_tied || new_pos == pos) { snprintf(g_err, sizeof g_err, "di_regroup: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    *n_tied = 0;
    if (m == 0) return 0;
    const uint32_t tiles = (m + DI_TILE - 1) / DI_TILE;
    uint32_t *tile_last = scratch, *tile_tied = scratch + tiles, *total = scratch + 2 * (size_t)tiles;
    k_grp_a<<<tiles, 256, 0, 0>>>(keys, pos, m, tile_last, tile_tied);
    k_grp_b<<<1, 256, 0, 0>>>(tile_last, tile_tied, tiles, total);
    k_grp_c<<<tiles, 256, 0, 0>>>(keys, vals, pos, m, lo, tile_last, tile_tied, rank, sa, new_pos);
    DI_CHK(hipGetLastError());
    DI_CHK(hipMemcpyAsync(n_tied, total, 4, hipMemcpyDeviceToHost, 0));
    DI_CHK(hipStreamSynchronize(0));
    return 0;
}

// ------------------------------------------------------------------------------------------ BWT + the Occ blocks' symbol words
// a lane packs 16 BWT symbols (one u32 of the .bwt body); the 8 lanes of a 128-symbol block add up their symbol counts
__global__ void __launch_bounds__(256) k_bwt_blocks(const int64_t *__restrict__ sa, const uint64_t *__restrict__ T, uint64_t n, uint64_t primary, uint64_t n_words,
                                                    uint32_t *__restrict__ blocks, uint32_t *__restrict__ counts)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t w = 0, c4 = 0;
    if (g < n_words) {
        const uint64_t o0 = g * 16u;
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const uint64_t o = o0 + (uint64_t)j;
            if (o < n) {
                const uint64_t p = (uint64_t)sa[o + (o >= primary ? 1u : 0u)];     // never 0: row `primary` is skipped
                const uint32_t c = d_sym(T, p - 1);
                w |= c << (30 - 2 * j);
                c4 += 1u << (8u * c);
            }
        }
    }
    c4 += (uint32_t)__shfl_xor((int)c4, 1, 64);
    c4 += (uint32_t)__shfl_xor((int)c4, 2, 64);
    c4 += (uint32_t)__shfl_xor((int)c4, 4, 64);
    if (g < n_words) {
        blocks[(g >> 3) * 16u + 8u + (g & 7u)] = w;
        if ((g & 7u) == 0) counts[g >> 3] = c4;
    }
}
extern "C" int di_bwt_blocks(int device, const int64_t *sa, const uint64_t *T, uint64_t n, uint64_t primary, uint32_t *blocks, uint32_t *counts)
{
    if (!sa || !T || !blocks || !counts || n == 0 || primary > n) { snprintf(g_err, sizeof g_err, "di_bwt_blocks: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    const uint64_t n_words = ((n + 127) / 128) * 8;
    k_bwt_blocks<<<(uint32_t)((n_words + 255) / 256), 256, 0, 0>>>(sa, T, n, primary, n_words, blocks, counts);
    DI_DONE("k_bwt_blocks");
}
