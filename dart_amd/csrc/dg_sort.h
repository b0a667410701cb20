// dart_amd/csrc/dg_sort.h -- stable LSD radix sort of (u64 key, i64 value) pairs in HBM: the sorter of the index builder.
//
// Replaces the sorting inside the reference's offline indexer (BWT_Index/bwtindex.c:77-148 -> bwt_gen.c / QSufSort.c: suffix
// sorting of the 2-bit text) for texts the GPU box has to index itself: dart_amd/index_build.py does prefix
// doubling, and every round is one sort of (rank pair, suffix) -- up to 386 M pairs per bucket of a GRCh38-sized text.
//
// 4 bits per pass, least significant digit first, tiles of 4096 pairs (256 lanes x 16):
//   k_rs_hist     per tile, how many keys fall in each of the 16 bins (bin-major table, so one scan gives every tile its bases)
//   (scan)        exclusive scan of the 16 x tiles table
//   k_rs_scatter  per tile: coalesced load into LDS; every lane takes 16 CONSECUTIVE pairs (stability = order by lane, then item),
//                 counts them per bin in its own LDS column; one scan over the 16 x 256 counts (bin-major) gives every lane its
//                 start in every bin; the pairs are put in tile-sorted order in LDS and leave with coalesced stores, each run of a
//                 bin going to  base[bin][tile] + (position - start of the bin in the tile)
// HBM-bound: 8 + 32 bytes per pair and pass.  No MFMA anywhere (integer keys).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RS_TILE 4096
#define RS_PAD(i) ((i) + ((i) >> 4))            // one slot of padding per 16: a lane's 16 consecutive pairs do not all hit one bank

__global__ void __launch_bounds__(256)
k_rs_hist(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t n_tiles, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t lh[4][16];
    if (threadIdx.x < 64) lh[threadIdx.x >> 4][threadIdx.x & 15] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE, wv = threadIdx.x >> 6;
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
        const uint32_t i = base + (uint32_t)j * 256u + threadIdx.x;
        if (i < n) atomicAdd(&lh[wv][(keys[i] >> shift) & 15u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = lh[0][threadIdx.x] + lh[1][threadIdx.x] + lh[2][threadIdx.x] + lh[3][threadIdx.x];
}

__global__ void __launch_bounds__(256)
k_rs_scatter(const uint64_t *__restrict__ keys, const int64_t *__restrict__ vals, uint64_t *__restrict__ keys_out, int64_t *__restrict__ vals_out,
             uint32_t n, int shift, uint32_t n_tiles, const uint32_t *__restrict__ bases)
{
    __shared__ uint64_t sk[RS_PAD(RS_TILE) + 1];
    __shared__ int64_t sv[RS_PAD(RS_TILE) + 1];
    __shared__ uint16_t cnt[16 * 256];
    __shared__ uint32_t wsum[4], bstart[16], gbase[16];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * RS_TILE;
    const uint32_t n_valid = n - base < RS_TILE ? n - base : RS_TILE;
    // 1. coalesced load (pairs past the end: largest key, so they stay last in the tile's order and are never stored)
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
        const uint32_t li = (uint32_t)j * 256u + tid, i = base + li;
        sk[RS_PAD(li)] = i < n ? keys[i] : ~0ull;
        sv[RS_PAD(li)] = i < n ? vals[i] : 0;
    }
    if (tid < 16) gbase[tid] = bases[(size_t)tid * n_tiles + blockIdx.x];
#pragma unroll
    for (int b = 0; b < 16; b++) cnt[b * 256 + tid] = 0;
    __syncthreads();
    // 2. this lane's 16 consecutive pairs, counted per bin in its own column
    uint64_t k[16]; int64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t li = tid * 16u + (uint32_t)j;
        k[j] = sk[RS_PAD(li)]; v[j] = sv[RS_PAD(li)];
        const uint32_t d = li < n_valid ? (uint32_t)(k[j] >> shift) & 15u : 15u;
        cnt[d * 256 + tid]++;
    }
    __syncthreads();
    // 3. exclusive scan of the 4096 counts in bin-major order: lane t owns entries [16 t, 16 t + 16)
    uint32_t c[16], mine = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) { c[q] = cnt[tid * 16 + q]; mine += c[q]; }
    uint32_t incl = mine;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += up; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t w = 0; w < wv; w++) run += wsum[w];
    if ((tid & 15u) == 0) bstart[tid >> 4] = run;              // entry (bin, lane 0): where the bin starts in the sorted tile
#pragma unroll
    for (int q = 0; q < 16; q++) { cnt[tid * 16 + q] = (uint16_t)run; run += c[q]; }
    __syncthreads();
    // 4. the pairs in tile-sorted order
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t li = tid * 16u + (uint32_t)j;
        const uint32_t d = li < n_valid ? (uint32_t)(k[j] >> shift) & 15u : 15u;
        const uint32_t p = cnt[d * 256 + tid]++;
        sk[RS_PAD(p)] = k[j]; sv[RS_PAD(p)] = v[j];
    }
    __syncthreads();
    // 5. coalesced stores
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
        const uint32_t p = (uint32_t)j * 256u + tid;
        if (p < n_valid) {
            const uint64_t key = sk[RS_PAD(p)];
            const uint32_t d = (uint32_t)(key >> shift) & 15u;
            const uint32_t dst = gbase[d] + (p - bstart[d]);
            keys_out[dst] = key; vals_out[dst] = sv[RS_PAD(p)];
        }
    }
}
