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

#define DI_API __attribute__((visibility("default")))        // built with -fvisibility=hidden: libdartgpu.so carries the sorter's kernels too,
                                                             // and a kernel's host stub must not be interposed by the other library's
static thread_local char g_err[256] = "";
extern "C" DI_API const char *di_last_error(void) { return g_err; }
static int hip_fail(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -2;
}
#define DI_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(#x, e_); } while (0)
#define DI_DONE(name) do { hipError_t e_ = hipGetLastError(); if (e_ == hipSuccess) e_ = hipStreamSynchronize(0); if (e_ != hipSuccess) return hip_fail(name, e_); return 0; } while (0)

extern "C" DI_API size_t di_text_words(uint64_t n) { return (size_t)((n + 31) / 32 + 2); }

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
extern "C" DI_API int di_pack_text(int device, const uint8_t *fwd, uint64_t l_pac, uint64_t *T)
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
extern "C" DI_API int di_bucket_hist(int device, const uint64_t *T, uint64_t n, uint32_t *table)
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
extern "C" DI_API int di_bucket_keys(int device, const uint64_t *T, uint64_t n, int pair, const uint32_t *tile_base, uint64_t *keys, int64_t *vals)
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
extern "C" DI_API int di_doubling_keys(int device, const int64_t *sa, const int64_t *rank, uint64_t lo, const uint32_t *pos, uint32_t m, uint64_t k, uint64_t N,
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
extern "C" DI_API int di_regroup(int device, const uint64_t *keys, const int64_t *vals, const uint32_t *pos, uint32_t m, uint64_t lo,
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
extern "C" DI_API int di_bwt_blocks(int device, const int64_t *sa, const uint64_t *T, uint64_t n, uint64_t primary, uint32_t *blocks, uint32_t *counts)
{
    if (!sa || !T || !blocks || !counts || n == 0 || primary > n) { snprintf(g_err, sizeof g_err, "di_bwt_blocks: bad argument"); return -1; }
    DI_CHK(hipSetDevice(device));
    const uint64_t n_words = ((n + 127) / 128) * 8;
    k_bwt_blocks<<<(uint32_t)((n_words + 255) / 256), 256, 0, 0>>>(sa, T, n, primary, n_words, blocks, counts);
    DI_DONE("k_bwt_blocks");
}

// ==========================================================================================
// The whole build, natively: di_build_files (what `dart index` and dart_amd/index_build.py call).  Host orchestration of the
// kernels above plus the radix sorter's kernels (../dg_sort.h, the same ones dg_sort_pairs launches) and three small scans.
// ==========================================================================================
#include "../dg_sort.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

// ---- exclusive scan of n u32 (three phases; tiles of 2048)
#define S32_TILE 2048u
__global__ void __launch_bounds__(256) k_s32_a(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *__restrict__ sums, uint32_t n)
{
    __shared__ uint32_t ws[4];
    const uint32_t base = blockIdx.x * S32_TILE + threadIdx.x * 8u, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = base + i < n ? in[base + i] : 0u; sum += v[i]; }
    uint32_t incl = sum;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
    if (lane == 63) ws[wv] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < wv; w++) run += ws[w];
#pragma unroll
    for (int i = 0; i < 8; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255) sums[blockIdx.x] = run;
}
__global__ void __launch_bounds__(256) k_s32_b(uint32_t *__restrict__ sums, uint32_t n_tiles, uint32_t *__restrict__ total)
{
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < n_tiles; b += 256u) {
        const uint32_t i = b + threadIdx.x, v = i < n_tiles ? sums[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_tiles) sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(256) k_s32_c(uint32_t *__restrict__ out, const uint32_t *__restrict__ sums, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] += sums[i / S32_TILE];
}
// scratch: ceil(n / 2048) + 1 u32; the total lands in scratch[ceil(n / 2048)]
static void scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *scratch)
{
    const uint32_t tiles = (n + S32_TILE - 1) / S32_TILE;
    k_s32_a<<<tiles, 256, 0, 0>>>(in, out, scratch, n);
    k_s32_b<<<1, 256, 0, 0>>>(scratch, tiles, scratch + tiles);
    k_s32_c<<<(n + 255u) / 256u, 256, 0, 0>>>(out, scratch, n);
}

// ---- the sorter: the kernels of ../dg_sort.h under the same driver as dg_sort_pairs (dg_api.hip), with caller-owned scratch
static void sort_pairs(uint64_t *keys, int64_t *vals, uint64_t *tk, int64_t *tv, uint32_t m, int key_bits, uint32_t *hist, uint32_t *scan_scratch)
{
    if (m < 2) return;
    const uint32_t tiles = (m + RS_TILE - 1) / RS_TILE, cells = 16u * tiles;
    uint64_t *ka = keys, *kb = tk; int64_t *va = vals, *vb = tv;
    for (int shift = 0; shift < key_bits; shift += 4) {
        k_rs_hist<<<tiles, 256, 0, 0>>>(ka, m, shift, tiles, hist);
        scan_u32(hist, hist, cells, scan_scratch);
        k_rs_scatter<<<tiles, 256, 0, 0>>>(ka, va, kb, vb, m, shift, tiles, hist);
        std::swap(ka, kb); std::swap(va, vb);
    }
    if (ka != keys) {
        (void)hipMemcpyAsync(keys, ka, (size_t)m * 8, hipMemcpyDeviceToDevice, 0);
        (void)hipMemcpyAsync(vals, va, (size_t)m * 8, hipMemcpyDeviceToDevice, 0);
    }
}

// ---- .pac bytes, sampled rows, the Occ counters in front of each block
__global__ void __launch_bounds__(256) k_pac(const uint8_t *__restrict__ fwd, uint64_t L, uint8_t *__restrict__ pac, uint64_t n_bytes)
{
    const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (b >= n_bytes) return;
    uint32_t v = 0;
    for (int j = 0; j < 4; j++) { const uint64_t p = b * 4u + (uint64_t)j; v = (v << 2) | (p < L ? fwd[p] & 3u : 0u); }
    pac[b] = (uint8_t)v;
}
__global__ void __launch_bounds__(256) k_sa_sample(const int64_t *__restrict__ sa, uint64_t *__restrict__ out, uint64_t count)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < count) out[i] = (uint64_t)sa[(i + 1) * 32u];
}
// Occ: four running counts (u64) over the blocks; a lane owns 4 consecutive blocks, a tile is 1024 blocks
#define OCC_TILE 1024u
__device__ __forceinline__ void d_occ_lane(const uint32_t *__restrict__ counts, uint64_t nblk, uint64_t b0, uint32_t c[4][4], uint32_t sum[4])
{
#pragma unroll
    for (int f = 0; f < 4; f++) sum[f] = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t v = b0 + (uint64_t)q < nblk ? counts[b0 + (uint64_t)q] : 0u;
#pragma unroll
        for (int f = 0; f < 4; f++) { c[q][f] = (v >> (8 * f)) & 255u; sum[f] += c[q][f]; }
    }
}
__global__ void __launch_bounds__(256) k_occ_a(const uint32_t *__restrict__ counts, uint64_t nblk, uint64_t *__restrict__ tsum)
{
    __shared__ uint32_t ws[4][4];
    uint32_t c[4][4], sum[4];
    d_occ_lane(counts, nblk, (uint64_t)blockIdx.x * OCC_TILE + threadIdx.x * 4u, c, sum);
#pragma unroll
    for (int f = 0; f < 4; f++) for (int o = 32; o; o >>= 1) sum[f] += (uint32_t)__shfl_xor((int)sum[f], o, 64);
    if ((threadIdx.x & 63u) == 0) for (int f = 0; f < 4; f++) ws[threadIdx.x >> 6][f] = sum[f];
    __syncthreads();
    if (threadIdx.x < 4) tsum[(size_t)blockIdx.x * 4u + threadIdx.x] = (uint64_t)ws[0][threadIdx.x] + ws[1][threadIdx.x] + ws[2][threadIdx.x] + ws[3][threadIdx.x];
}
__global__ void __launch_bounds__(256) k_occ_b(uint64_t *__restrict__ tsum, uint32_t tiles, uint64_t *__restrict__ total)
{
    __shared__ uint64_t sh[256];
    __shared__ uint64_t carry;
    for (int f = 0; f < 4; f++) {
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (uint32_t b = 0; b < tiles; b += 256u) {
            const uint32_t i = b + threadIdx.x;
            const uint64_t v = i < tiles ? tsum[(size_t)i * 4u + f] : 0ull;
            sh[threadIdx.x] = v;
            __syncthreads();
            for (int o = 1; o < 256; o <<= 1) {
                const uint64_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0ull;
                __syncthreads();
                sh[threadIdx.x] += t;
                __syncthreads();
            }
            if (i < tiles) tsum[(size_t)i * 4u + f] = carry + sh[threadIdx.x] - v;
            __syncthreads();
            if (threadIdx.x == 255) carry += sh[255];
            __syncthreads();
        }
        if (threadIdx.x == 0) total[f] = carry;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_occ_c(const uint32_t *__restrict__ counts, uint64_t nblk, const uint64_t *__restrict__ tsum, uint64_t *__restrict__ blocks64)
{
    __shared__ uint32_t ws[4][4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t b0 = (uint64_t)blockIdx.x * OCC_TILE + threadIdx.x * 4u;
    uint32_t c[4][4], sum[4], ex[4];
    d_occ_lane(counts, nblk, b0, c, sum);
#pragma unroll
    for (int f = 0; f < 4; f++) {
        uint32_t incl = sum[f];
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
        if (lane == 63) ws[wv][f] = incl;
        ex[f] = incl - sum[f];
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 4; f++) {
        uint64_t run = tsum[(size_t)blockIdx.x * 4u + f] + ex[f];
        for (uint32_t w = 0; w < wv; w++) run += ws[w][f];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (b0 + (uint64_t)q < nblk) blocks64[(b0 + (uint64_t)q) * 8u + (uint64_t)f] = run;     // a block = 16 u32 = 8 u64: the first four are the counts
            run += c[q][f];
        }
    }
}

namespace {
struct Dev {                                   // device allocations of one build, freed together
    std::vector<void *> all;
    ~Dev() { for (void *p : all) if (p) (void)hipFree(p); }
    template <class T> hipError_t get(T **out, size_t count) { void *p = nullptr; hipError_t e = hipMalloc(&p, count * sizeof(T) + 16); if (e == hipSuccess) { all.push_back(p); *out = (T *)p; } return e; }
    void drop(void *p) { for (void *&q : all) if (q == p) { (void)hipFree(q); q = nullptr; } }
};
struct Clock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double s() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
bool write_all(FILE *f, const void *p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }
// device memory -> file through two page-locked 32 MB buffers: the copy of chunk k + 1 runs while chunk k is written (no host copy of a 3 GB array)
struct DeviceToFile {
    static constexpr size_t CHUNK = 32u << 20;
    void *buf[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; bool ok = false;
    DeviceToFile()
    {
        ok = hipHostMalloc(&buf[0], CHUNK) == hipSuccess && hipHostMalloc(&buf[1], CHUNK) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
    }
    ~DeviceToFile()
    {
        for (int i = 0; i < 2; i++) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (buf[i]) (void)hipHostFree(buf[i]); }
        if (st) (void)hipStreamDestroy(st);
    }
    bool write(FILE *f, const void *dev, size_t bytes)                       // the device data must be complete (the caller has synchronised)
    {
        if (!ok) return false;
        const size_t n = (bytes + CHUNK - 1) / CHUNK;
        for (size_t k = 0; k <= n; k++) {
            if (k < n) {
                const size_t len = std::min(CHUNK, bytes - k * CHUNK);
                if (hipMemcpyAsync(buf[k & 1], (const char *)dev + k * CHUNK, len, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
                if (hipEventRecord(ev[k & 1], st) != hipSuccess) return false;
            }
            if (k > 0) {
                const size_t j = k - 1, len = std::min(CHUNK, bytes - j * CHUNK);
                if (hipEventSynchronize(ev[j & 1]) != hipSuccess) return false;
                if (!write_all(f, buf[j & 1], len)) return false;
            }
        }
        return true;
    }
};
}

extern "C" DI_API int di_build_files(int device, const uint8_t *fwd, uint64_t l_pac, const char *prefix, di_log_fn log, void *log_arg, uint64_t *primary_out)
{
    if (!fwd || !prefix || l_pac < 32) { snprintf(g_err, sizeof g_err, "di_build_files: bad argument (a text of fewer than 64 symbols is not indexed here)"); return -1; }
    DI_CHK(hipSetDevice(device));
    Clock clk;
    char line[320];
    double sort_s = 0;
    auto say = [&](const char *what) {
        if (!log) return;
        (void)hipDeviceSynchronize();
        snprintf(line, sizeof line, "  [%6.1f s; sorter %5.1f s] %s", clk.s(), sort_s, what);
        log(line, log_arg);
    };
#define DI_LAUNCHED(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return hip_fail(name, e_); } while (0)
    const uint64_t L = l_pac, n = 2 * L, N = n + 1;
    const std::string pre(prefix);
    Dev d;
    // ---- text, .pac
    uint8_t *fwd_d = nullptr, *pac_d = nullptr; uint64_t *T = nullptr;
    const uint64_t words = di_text_words(n), pac_bytes = (L + 3) / 4;
    DI_CHK(d.get(&fwd_d, L));
    DI_CHK(hipMemcpy(fwd_d, fwd, L, hipMemcpyHostToDevice));
    DI_CHK(d.get(&T, words));
    k_pack_text<<<(uint32_t)((words + 255) / 256), 256, 0, 0>>>(fwd_d, L, T, words);
    DI_CHK(d.get(&pac_d, pac_bytes));
    k_pac<<<(uint32_t)((pac_bytes + 255) / 256), 256, 0, 0>>>(fwd_d, L, pac_d, pac_bytes);
    DI_LAUNCHED("k_pack_text / k_pac");
    DeviceToFile out;
    if (!out.ok) { snprintf(g_err, sizeof g_err, "di_build_files: no page-locked staging buffers"); return -2; }
    {
        DI_CHK(hipStreamSynchronize(0));
        FILE *f = fopen((pre + ".pac").c_str(), "wb");
        const uint8_t zero = 0, tail = (uint8_t)(L % 4);
        bool ok = f && out.write(f, pac_d, pac_bytes) && (L % 4 != 0 || write_all(f, &zero, 1)) && write_all(f, &tail, 1);      // bntseq.c:192-201
        if (f) ok = fclose(f) == 0 && ok;
        if (!ok) { snprintf(g_err, sizeof g_err, "di_build_files: cannot write %s.pac", prefix); return -3; }
    }
    const uint32_t last_sym = 3u - (fwd[0] & 3u);
    d.drop(fwd_d); d.drop(pac_d);
    say("text packed (forward + reverse complement, 2 bits per symbol), .pac written");
    // ---- buckets
    const uint64_t tiles64 = (N + DI_TILE - 1) / DI_TILE;
    if (tiles64 >= 0x7FFFFFFFull / 16) { snprintf(g_err, sizeof g_err, "di_build_files: text too long"); return -1; }
    const uint32_t tiles = (uint32_t)tiles64;
    uint32_t *table = nullptr, *bases = nullptr, *scan_scr = nullptr;
    DI_CHK(d.get(&table, (size_t)16 * tiles));
    DI_CHK(d.get(&bases, (size_t)16 * tiles));
    k_bucket_hist<<<tiles, 256, 0, 0>>>(T, n, tiles, table);
    DI_LAUNCHED("k_bucket_hist");
    uint64_t counts[16], lows[16], m_max = 0;
    {
        uint32_t *scr = nullptr;
        DI_CHK(d.get(&scr, (size_t)(tiles / S32_TILE + 2)));
        for (int p = 0; p < 16; p++) {
            scan_u32(table + (size_t)p * tiles, bases + (size_t)p * tiles, tiles, scr);
            uint32_t tot = 0;
            DI_CHK(hipMemcpy(&tot, scr + (tiles + S32_TILE - 1) / S32_TILE, 4, hipMemcpyDeviceToHost));
            counts[p] = tot; if (tot > m_max) m_max = tot;
        }
        d.drop(scr); d.drop(table);
    }
    if (m_max >= 0x7FFFF000ull) { snprintf(g_err, sizeof g_err, "di_build_files: a two-symbol bucket holds >= 2^31 suffixes"); return -1; }
    int r2_bits = 0; while ((N >> r2_bits) != 0) r2_bits++;                     // a rank + 1 is at most N
    int m_bits = 0; while ((m_max >> m_bits) != 0) m_bits++;
    if (r2_bits + m_bits > 64) { snprintf(g_err, sizeof g_err, "di_build_files: the rank pair does not fit 64 bits"); return -1; }
    int64_t *sa = nullptr, *rank = nullptr, *vals = nullptr, *tv = nullptr; uint64_t *keys = nullptr, *tk = nullptr;
    uint32_t *new_pos = nullptr, *grp_scr = nullptr, *hist = nullptr;
    DI_CHK(d.get(&sa, N)); DI_CHK(d.get(&rank, N));
    DI_CHK(d.get(&keys, m_max)); DI_CHK(d.get(&vals, m_max)); DI_CHK(d.get(&tk, m_max)); DI_CHK(d.get(&tv, m_max));
    DI_CHK(d.get(&new_pos, m_max));
    const uint32_t m_tiles = (uint32_t)((m_max + DI_TILE - 1) / DI_TILE);
    DI_CHK(d.get(&grp_scr, (size_t)2 * m_tiles + 4));
    DI_CHK(d.get(&hist, (size_t)16 * m_tiles + 16));
    DI_CHK(d.get(&scan_scr, (size_t)(16 * (size_t)m_tiles / S32_TILE + 4)));
    say("bucket sizes known, buffers allocated");
    // rows: '$' first, then per first symbol c0 the suffix "c0 $" (if the text ends in c0), then the buckets c0 A, c0 C, c0 G, c0 T
    {
        uint64_t row = 1;
        const int64_t s_n = (int64_t)n, zero = 0;
        DI_CHK(hipMemcpy(sa, &s_n, 8, hipMemcpyHostToDevice));
        DI_CHK(hipMemcpy(rank + n, &zero, 8, hipMemcpyHostToDevice));
        for (uint32_t c0 = 0; c0 < 4; c0++) {
            if (c0 == last_sym) {
                const int64_t s1 = (int64_t)n - 1, r1 = (int64_t)row;
                DI_CHK(hipMemcpy(sa + row, &s1, 8, hipMemcpyHostToDevice));
                DI_CHK(hipMemcpy(rank + (n - 1), &r1, 8, hipMemcpyHostToDevice));
                row++;
            }
            for (uint32_t c1 = 0; c1 < 4; c1++) { lows[c0 * 4 + c1] = row; row += counts[c0 * 4 + c1]; }
        }
        if (row != N) { snprintf(g_err, sizeof g_err, "di_build_files: bucket counts do not add up"); return -2; }
    }
    auto sorted = [&](uint32_t m, int bits) {
        (void)hipStreamSynchronize(0);
        Clock c;
        sort_pairs(keys, vals, tk, tv, m, bits, hist, scan_scr);
        (void)hipStreamSynchronize(0);
        sort_s += c.s();
    };
    auto regroup = [&](const uint32_t *pos, uint32_t m, uint64_t lo, uint32_t *n_tied) -> int {
        const uint32_t t = (m + DI_TILE - 1) / DI_TILE;
        k_grp_a<<<t, 256, 0, 0>>>(keys, pos, m, grp_scr, grp_scr + t);
        k_grp_b<<<1, 256, 0, 0>>>(grp_scr, grp_scr + t, t, grp_scr + 2 * (size_t)t);
        k_grp_c<<<t, 256, 0, 0>>>(keys, vals, pos, m, lo, grp_scr, grp_scr + t, rank, sa, new_pos);
        DI_LAUNCHED("k_grp_*");
        DI_CHK(hipMemcpy(n_tied, grp_scr + 2 * (size_t)t, 4, hipMemcpyDeviceToHost));
        return 0;
    };
    uint32_t *pending[16] = {nullptr}; uint32_t n_pending[16] = {0};
    auto keep_tied = [&](int p, uint32_t t) -> int {                           // the bucket's still-tied rows, in a buffer of their own
        if (pending[p]) { d.drop(pending[p]); pending[p] = nullptr; }
        n_pending[p] = t;
        if (t == 0) return 0;
        DI_CHK(d.get(&pending[p], t));
        DI_CHK(hipMemcpy(pending[p], new_pos, (size_t)t * 4, hipMemcpyDeviceToDevice));
        return 0;
    };
    uint64_t tied_total = 0;
    for (int p = 0; p < 16; p++) {                                             // round 0: the first 31 symbols
        const uint32_t m = (uint32_t)counts[p];
        if (m == 0) continue;
        k_bucket_keys<<<tiles, 256, 0, 0>>>(T, n, (uint32_t)p, bases + (size_t)p * tiles, keys, vals);
        DI_LAUNCHED("k_bucket_keys");
        sorted(m, 63);
        uint32_t t = 0;
        int rc = regroup(nullptr, m, lows[p], &t); if (rc) return rc;
        rc = keep_tied(p, t); if (rc) return rc;
        tied_total += t;
        if (getenv("DART_INDEX_VERBOSE")) {
            snprintf(line + 200, 100, "bucket %c%c: %u suffixes, %u still tied after 31 symbols", "ACGT"[p >> 2], "ACGT"[p & 3], m, t);
            const std::string msg(line + 200); say(msg.c_str());
        }
    }
    d.drop(bases);
    snprintf(line + 200, 100, "round 0 (31 symbols): %llu of %llu suffixes still tied", (unsigned long long)tied_total, (unsigned long long)N);
    { const std::string msg(line + 200); say(msg.c_str()); }
    for (uint64_t k = 31; tied_total; k *= 2) {
        tied_total = 0;
        for (int p = 0; p < 16; p++) {
            const uint32_t m = n_pending[p];
            if (m == 0) continue;
            k_doubling_keys<<<(m + 255u) / 256u, 256, 0, 0>>>(sa, rank, lows[p], pending[p], m, k, N, r2_bits, keys, vals);
            DI_LAUNCHED("k_doubling_keys");
            int bits = 0; while ((counts[p] >> bits) != 0) bits++;
            sorted(m, std::min(64, r2_bits + bits));
            uint32_t t = 0;
            int rc = regroup(pending[p], m, lows[p], &t); if (rc) return rc;
            rc = keep_tied(p, t); if (rc) return rc;
            tied_total += t;
        }
        snprintf(line + 200, 100, "after %llu symbols: %llu suffixes still tied", (unsigned long long)(2 * k), (unsigned long long)tied_total);
        { const std::string msg(line + 200); say(msg.c_str()); }
    }
    int64_t primary_i = 0;
    DI_CHK(hipMemcpy(&primary_i, rank, 8, hipMemcpyDeviceToHost));             // the row of suffix 0
    const uint64_t primary = (uint64_t)primary_i;
    if (primary_out) *primary_out = primary;
    d.drop(rank); d.drop(keys); d.drop(vals); d.drop(tk); d.drop(tv); d.drop(new_pos); d.drop(grp_scr); d.drop(hist); d.drop(scan_scr);
    say("suffix array done");
    // ---- BWT, Occ
    const uint64_t nblk = (n + 127) / 128, n_words = nblk * 8;
    uint32_t *blocks = nullptr, *c4 = nullptr; uint64_t *tsum = nullptr, *sample = nullptr;
    const uint32_t occ_tiles = (uint32_t)((nblk + OCC_TILE - 1) / OCC_TILE);
    DI_CHK(d.get(&blocks, nblk * 16)); DI_CHK(d.get(&c4, nblk)); DI_CHK(d.get(&tsum, (size_t)occ_tiles * 4 + 4));
    k_bwt_blocks<<<(uint32_t)((n_words + 255) / 256), 256, 0, 0>>>(sa, T, n, primary, n_words, blocks, c4);
    k_occ_a<<<occ_tiles, 256, 0, 0>>>(c4, nblk, tsum);
    k_occ_b<<<1, 256, 0, 0>>>(tsum, occ_tiles, tsum + (size_t)occ_tiles * 4);
    k_occ_c<<<occ_tiles, 256, 0, 0>>>(c4, nblk, tsum, (uint64_t *)blocks);
    DI_LAUNCHED("k_bwt_blocks / k_occ_*");
    uint64_t occ_last[4], L2[5] = {0, 0, 0, 0, 0};
    DI_CHK(hipMemcpy(occ_last, tsum + (size_t)occ_tiles * 4, 32, hipMemcpyDeviceToHost));
    for (int c = 0; c < 4; c++) L2[c + 1] = L2[c] + occ_last[c];               // the BWT is a permutation of the text: its totals are the symbol counts
    if (L2[4] != n) { snprintf(g_err, sizeof g_err, "di_build_files: BWT symbol counts do not add up"); return -2; }
    const uint64_t n_samples = n / 32;                                         // rows 32, 64, ... (bwt.c:101-123)
    DI_CHK(d.get(&sample, n_samples + 1));
    k_sa_sample<<<(uint32_t)((n_samples + 255) / 256), 256, 0, 0>>>(sa, sample, n_samples);
    DI_LAUNCHED("k_sa_sample");
    say("BWT words, Occ counters and sampled rows on the device");
    {
        const uint64_t nwords16 = (n + 15) / 16;
        const uint64_t body_words = (nblk - 1) * 16 + 8 + (nwords16 - (nblk - 1) * 8);           // the last block keeps only the words that hold symbols (bwtindex.c:53-75)
        DI_CHK(hipStreamSynchronize(0));
        FILE *f = fopen((pre + ".bwt").c_str(), "wb");
        bool ok = f && write_all(f, &primary, 8) && write_all(f, L2 + 1, 32) && out.write(f, blocks, body_words * 4) && write_all(f, occ_last, 32);
        if (f) ok = fclose(f) == 0 && ok;
        if (!ok) { snprintf(g_err, sizeof g_err, "di_build_files: cannot write %s.bwt", prefix); return -3; }
    }
    {
        const uint64_t hdr[2] = {32, n};
        FILE *f = fopen((pre + ".sa").c_str(), "wb");
        bool ok = f && write_all(f, &primary, 8) && write_all(f, L2 + 1, 32) && write_all(f, hdr, 16) && out.write(f, sample, n_samples * 8);
        if (f) ok = fclose(f) == 0 && ok;
        if (!ok) { snprintf(g_err, sizeof g_err, "di_build_files: cannot write %s.sa", prefix); return -3; }
    }
    say(".bwt and .sa written");
    return 0;
#undef DI_LAUNCHED
}
