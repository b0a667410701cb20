// dart_amd/csrc/dg_scan.h -- single-pass prefix sums across the workgroups of ONE launch (decoupled look-back).
//
// Why: the record arrays of a batch are laid out in read order (reports of read r start at rep_off[r], CIGAR ops follow the
// reports), and the sizes come out of the same kernel that produces the records (k_pair: candidates per read, CIGAR ops per
// report, which units need the general path).  Counting in one launch, scanning in three more and emitting in another made
// every record cross HBM twice and cost seven launches per scan; here a workgroup publishes its totals, sums its
// predecessors' and goes on to write its records at their final place.
//
// Protocol (per scanned launch): workgroups take a ticket (tile number = order of arrival, so every predecessor of a tile has
// started and never waits for a later one -- no deadlock whatever order the hardware starts workgroups in).  A tile's state is
// two 64-bit words, each (status << 62) | payload, written with one atomic store each:
//   word A: two 31-bit counters x | y << 31      word B: one 62-bit counter z
//   status 1 = payload is the tile's own total, 2 = payload is the inclusive prefix up to and including the tile.
// A reader accepts a tile when both words carry the same non-zero status (the writer updates A then B, so a torn pair shows
// different statuses and is polled again).  The first wave of the workgroup looks back 64 tiles at a time: it adds totals
// until it meets an inclusive prefix.  The arrays must be zero before the launch (status 0 = nothing yet).
// A poll budget turns a protocol failure into an error code instead of a hung GPU; the host then runs the batch again (seen once in ~10^5
// batches with a dozen contexts in flight, cause not found: finish_run in dg_api.hip).
#pragma once
#include "dg_common.h"

struct TileScan { unsigned long long *a, *b; unsigned int *ticket; };
struct Triple { uint32_t x, y; uint64_t z; };

// RELAXED on purpose: a state word carries its whole message (status + payload in one 64-bit atomic access that goes to the
// device-coherent level, `sc1`), nothing else is published through it.  Acquire / release at agent scope cost a `buffer_inv sc1`
// per poll and a `buffer_wbl2 sc1` -- a write-back of the XCD's whole L2 -- per store on gfx950: with those the scan was 90 % of
// k_pair's time.
__device__ __forceinline__ unsigned long long ts_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ts_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ts_pack_a(int st, uint32_t x, uint32_t y) { return ((unsigned long long)st << 62) | ((unsigned long long)(y & 0x7FFFFFFFu) << 31) | (unsigned long long)(x & 0x7FFFFFFFu); }
__device__ __forceinline__ unsigned long long ts_pack_b(int st, uint64_t z) { return ((unsigned long long)st << 62) | (z & 0x3FFFFFFFFFFFFFFFull); }

// ticket of this workgroup (call once, by every thread; sh = one u32 of LDS)
__device__ __forceinline__ unsigned int d_tile_ticket(const TileScan &ts, unsigned int *sh)
{
    if (threadIdx.x == 0) *sh = atomicAdd(ts.ticket, 1u);
    __syncthreads();
    return *sh;
}

// exclusive prefix of this tile's totals over all earlier tiles; every thread of the workgroup calls it with the tile's totals
// (only thread 0's copy is published) and gets the same result.  sh = 4 u64 of LDS.  Workgroups must have >= 64 threads.
__device__ inline Triple d_tile_exclusive(const TileScan &ts, unsigned int tile, Triple own, unsigned long long *sh, int *err)
{
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        uint64_t sx = 0, sy = 0, sz = 0;
        if (tile > 0) {
            if (lane == 0) { ts_store(ts.a + tile, ts_pack_a(1, own.x, own.y)); ts_store(ts.b + tile, ts_pack_b(1, own.z)); }
            long long base = (long long)tile - 1;
            unsigned int polls = 0;
            while (true) {
                const long long t = base - lane;
                unsigned long long va = ts_pack_a(2, 0, 0), vb = ts_pack_b(2, 0);      // before tile 0: an inclusive prefix of nothing
                if (t >= 0) { va = ts_load(ts.a + t); vb = ts_load(ts.b + t); }
                const int sa = (int)(va >> 62), sb = (int)(vb >> 62);
                const bool ready = sa != 0 && sa == sb;
                const unsigned long long m_incl = __ballot(ready && sa == 2), m_wait = __ballot(!ready);
                int upto = -1;                                                          // lanes 0..upto are summed; -1 = poll again
                bool done = false;
                if (m_incl) {
                    const int f = __ffsll((long long)m_incl) - 1;
                    if (!(m_wait & ((1ull << f) - 1ull))) { upto = f; done = true; }
                } else if (!m_wait) upto = 63;
                if (upto >= 0) {
                    uint64_t x = lane <= upto ? (va & 0x7FFFFFFFull) : 0, y = lane <= upto ? ((va >> 31) & 0x7FFFFFFFull) : 0, z = lane <= upto ? (vb & 0x3FFFFFFFFFFFFFFFull) : 0;
                    for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); z += __shfl_xor(z, o, 64); }
                    sx += x; sy += y; sz += z;
                    if (done) break;
                    base -= 64;
                } else {
                    polls++;
                    if ((polls & 1023u) == 0u) {
                        // another kernel stage gave the batch up (the host will run it again): nobody needs this prefix any more
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= DG_ABORT) break;
                        // belt and braces: a state word is read with a device-coherent load, but should a stale copy ever sit in a cache
                        // of this XCD, this drops it (one invalidate per thousand polls costs nothing; one per poll tripled k_pair's time)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    if (polls > (1u << 20)) { if (lane == 0) atomicMax(err, DG_E_SCAN); break; }       // ~2 s: the host runs the batch again
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        if (lane == 0) {
            ts_store(ts.a + tile, ts_pack_a(2, (uint32_t)sx + own.x, (uint32_t)sy + own.y)); ts_store(ts.b + tile, ts_pack_b(2, sz + own.z));
            sh[0] = sx; sh[1] = sy; sh[2] = sz;
        }
    }
    __syncthreads();
    Triple r; r.x = (uint32_t)sh[0]; r.y = (uint32_t)sh[1]; r.z = sh[2];
    __syncthreads();
    return r;
}

// exclusive prefix of three per-thread counts inside a 256-thread workgroup + the workgroup totals.  sh = 3 * 4 u64 of LDS.
__device__ inline Triple d_block_exclusive(Triple v, Triple &total, unsigned long long *sh)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint64_t x = v.x, y = v.y, z = v.z;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t px = __shfl_up(x, o, 64), py = __shfl_up(y, o, 64), pz = __shfl_up(z, o, 64);
        if (lane >= o) { x += px; y += py; z += pz; }
    }
    if (lane == 63) { sh[wv] = x; sh[4 + wv] = y; sh[8 + wv] = z; }
    __syncthreads();
    uint64_t bx = 0, by = 0, bz = 0, tx = 0, ty = 0, tz = 0;
    for (int w = 0; w < nw; w++) {
        if (w < wv) { bx += sh[w]; by += sh[4 + w]; bz += sh[8 + w]; }
        tx += sh[w]; ty += sh[4 + w]; tz += sh[8 + w];
    }
    __syncthreads();
    total.x = (uint32_t)tx; total.y = (uint32_t)ty; total.z = tz;
    Triple r; r.x = (uint32_t)(bx + x - v.x); r.y = (uint32_t)(by + y - v.y); r.z = bz + z - v.z;
    return r;
}
