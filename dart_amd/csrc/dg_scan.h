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
// four 64-bit words, one per counter, each self-describing and written with one store:
//   word k = tag << 32 | value_k          tag = epoch << 2 | status
//   status 1 = the values are the tile's own totals, 2 = the inclusive prefix up to and including the tile
//   epoch    = the number of the run (dg_ctx::scan_epoch, 30 bits, never 0, different for every enqueued run of a context)
// A reader accepts a tile when its four words carry the SAME tag and that tag's epoch is the reader's own: a word left by an
// earlier run of the context (any status, any value) is "nothing yet", so the arrays are never zeroed between runs (they are
// zeroed once, when allocated) and no stale word can be taken for a result -- round 2 zeroed them with a fill launch per batch and
// trusted that no cached copy of a previous batch's word survived it.  The first wave of the workgroup looks back 64 tiles at a
// time: it adds totals until it meets an inclusive prefix.
// A time budget (wall-clock ticks, ~2 s; the tests' hook is a poll count) turns a predecessor that does not publish into an error code instead
// of a hung GPU and records what the poller saw AND what the stuck tile's workgroup last said about itself (its trace words: where it runs --
// HW_ID, XCC_ID -- and when it took its ticket, published its totals, published its prefix); the host logs that and runs the batch again
// (finish_run in dg_api.hip).
#pragma once
#include "dg_common.h"

struct TileScan { unsigned long long *w; unsigned int *ticket; uint32_t epoch, budget /* polls; 0 = none (the time budget alone) */; unsigned long long *dbg;
                  unsigned long long ticks /* wall_clock64 ticks a look-back may wait; 0 = no limit */; unsigned long long *trace /* SCAN_TRACE_WORDS per tile, or null */; };
struct Triple { uint32_t x, y; uint64_t z; uint32_t w; };   // four counters (the name is older than the fourth); z stays 64-bit for the callers' arithmetic, a published z is < 2^32: it counts CIGAR ops of one batch
#define SCAN_WORDS 4            // state words per tile
#define SCAN_DBG_WORDS 16       // what a poller that ran out of budget saw: [0] its tile + 1, [1] the stuck predecessor, [2..4] that tile's state words, [5] epoch, [6] polls, [7] lane,
                                // [8..11] the stuck tile's trace words, [12] the time the poller gave up, [13] the time it started to wait
#define SCAN_TRACE_WORDS 4      // per tile, written by thread 0 of its workgroup: [0] epoch << 32 | HW_ID (wave, SIMD, CU, SH, SE), [1] wall clock at the ticket,
                                // [2] at the publication of its own totals (0: not yet), [3] XCC_ID << 56 | wall clock at the publication of its prefix (low bits 0: not yet)

// RELAXED on purpose: a state word carries its whole message (tag + value in one 64-bit access that goes to the device-coherent
// level, `sc1`), nothing else is published through it.  Acquire / release at agent scope cost a `buffer_inv sc1` per poll and a
// `buffer_wbl2 sc1` -- a write-back of the XCD's whole L2 -- per store on gfx950: with those the scan was 90 % of k_pair's time.
__device__ __forceinline__ unsigned long long ts_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ts_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__host__ __device__ __forceinline__ uint32_t ts_tag(uint32_t epoch, int st) { return (epoch << 2) | (uint32_t)st; }
__host__ __device__ __forceinline__ unsigned long long ts_pack(uint32_t tag, uint32_t v) { return ((unsigned long long)tag << 32) | (unsigned long long)v; }
__device__ __forceinline__ void ts_publish(const TileScan &ts, unsigned int tile, int st, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    const uint32_t tag = ts_tag(ts.epoch, st);
    unsigned long long *p = ts.w + (size_t)tile * SCAN_WORDS;
    ts_store(p, ts_pack(tag, x)); ts_store(p + 1, ts_pack(tag, y)); ts_store(p + 2, ts_pack(tag, z)); ts_store(p + 3, ts_pack(tag, w));
}

// Ticket of this workgroup (call once, by every thread; sh = one u32 of LDS) -- unless thread 0 says the workgroup leaves: kernels leave a batch whose status
// word says "this run is void" (dg_common.h, DG_ABORT).  That decision is thread 0's alone, made BEFORE it takes the ticket and handed to the workgroup through
// LDS behind the ticket's barrier, so the waves of a workgroup can never disagree: a status raised INSIDE the launch by another tile's look-back (DG_E_SCAN)
// reaches the waves of one workgroup at different times, and with every wave reading *err on its own (round 4) some left while the others went on with a tile
// number or block sums nobody had written.  Returns 0xFFFFFFFF = leave: no ticket was taken, so no tile exists that a successor could wait for.
#define SCAN_LEAVE 0xFFFFFFFFu
__device__ __forceinline__ unsigned int d_tile_ticket_unless(const TileScan &ts, unsigned int *sh, bool leave_says_thread0)
{
    if (threadIdx.x == 0) {
        if (leave_says_thread0) *sh = SCAN_LEAVE;
        else {
            const unsigned int t = atomicAdd(ts.ticket, 1u);
            *sh = t;
            if (ts.trace) {
                unsigned long long *p = ts.trace + (size_t)t * SCAN_TRACE_WORDS;
                const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_HW_ID, HW_REG_XCC_ID
                p[0] = ((unsigned long long)ts.epoch << 32) | hw; p[1] = wall_clock64(); p[2] = 0ull; p[3] = (unsigned long long)(xcc & 0xFFu) << 56;
            }
        }
    }
    __syncthreads();
    return *sh;
}
__device__ __forceinline__ unsigned int d_tile_ticket(const TileScan &ts, unsigned int *sh) { return d_tile_ticket_unless(ts, sh, false); }

// exclusive prefix of this tile's totals over all earlier tiles; every thread of the workgroup calls it with the tile's totals
// (only thread 0's copy is published) and gets the same result.  sh = 4 u64 of LDS (behind d_block_exclusive's 16).  Workgroups must have >= 64 threads.
// Only DG_E_SCAN (another poller of this run gave up: the host runs the batch again) ends a look-back early; a capacity overflow
// raised inside the same launch does not -- the totals the host grows its buffers from stay exact.
__device__ inline Triple d_tile_exclusive(const TileScan &ts, unsigned int tile, Triple own, unsigned long long *sh, int *err)
{
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        uint64_t sx = 0, sy = 0, sz = 0, sw = 0;
        if (tile > 0) {
            if (lane == 0) { ts_publish(ts, tile, 1, own.x, own.y, (uint32_t)own.z, own.w); if (ts.trace) ts.trace[(size_t)tile * SCAN_TRACE_WORDS + 2] = wall_clock64(); }
            long long base = (long long)tile - 1;
            unsigned int polls = 0;
            const unsigned long long t_wait0 = wall_clock64();
            bool late = false;
            const uint32_t want1 = ts_tag(ts.epoch, 1), want2 = ts_tag(ts.epoch, 2);
            while (true) {
                const long long t = base - lane;
                unsigned long long va = ts_pack(want2, 0), vb = va, vc = va, vd = va;    // before tile 0: an inclusive prefix of nothing
                if (t >= 0) { const unsigned long long *p = ts.w + (size_t)t * SCAN_WORDS; va = ts_load(p); vb = ts_load(p + 1); vc = ts_load(p + 2); vd = ts_load(p + 3); }
                const uint32_t ta = (uint32_t)(va >> 32), tb = (uint32_t)(vb >> 32), tc = (uint32_t)(vc >> 32), td = (uint32_t)(vd >> 32);
                // (a torn state shows different tags and is polled again; the tests' hook -- a poll budget of 1 -- makes tile 1 see nothing at all, so
                //  that one tile of every hooked scan gives up whatever the timing: elsewhere a predecessor has usually published already)
                const bool ready = !(ts.budget == 1u && tile == 1u) && ta == tb && tb == tc && tc == td && (ta == want1 || ta == want2);
                const unsigned long long m_incl = __ballot(ready && ta == want2), m_wait = __ballot(!ready);
                int upto = -1;                                                          // lanes 0..upto are summed; -1 = poll again
                bool done = false;
                if (m_incl) {
                    const int f = __ffsll((long long)m_incl) - 1;
                    if (!(m_wait & ((1ull << f) - 1ull))) { upto = f; done = true; }
                } else if (!m_wait) upto = 63;
                if (upto >= 0) {
                    uint64_t x = lane <= upto ? (va & 0xFFFFFFFFull) : 0, y = lane <= upto ? (vb & 0xFFFFFFFFull) : 0, z = lane <= upto ? (vc & 0xFFFFFFFFull) : 0,
                             w = lane <= upto ? (vd & 0xFFFFFFFFull) : 0;
                    for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); z += __shfl_xor(z, o, 64); w += __shfl_xor(w, o, 64); }
                    sx += x; sy += y; sz += z; sw += w;
                    if (done) break;
                    base -= 64;
                } else {
                    polls++;
                    if ((polls & 1023u) == 0u) {
                        // another poller of this run gave up (the host will run the batch again): nobody needs this prefix any more
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == DG_E_SCAN) break;
                        // a state word is read with a device-coherent load; should a copy ever sit in this CU's vector cache, this drops it
                        // (one invalidate per thousand polls costs nothing; one per poll tripled k_pair's time)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        late = ts.ticks && wall_clock64() - t_wait0 > ts.ticks;
                    }
                    if (late || (ts.budget && polls > ts.budget)) {
                        // the nearest predecessor this wave still waits for, and what its words look like from here
                        const int stuck = __ffsll((long long)m_wait) - 1;
                        if (lane == stuck && ts.dbg && atomicCAS(ts.dbg, 0ull, (unsigned long long)tile + 1ull) == 0ull) {
                            ts.dbg[1] = (unsigned long long)t; ts.dbg[2] = va; ts.dbg[3] = vb; ts.dbg[4] = vc ^ (vd & 0xFFFFFFFFull);      // (the fourth word's value folded in: it carries the same tag)
                            ts.dbg[5] = ts.epoch; ts.dbg[6] = polls; ts.dbg[7] = (unsigned long long)lane;
                            if (ts.trace && t >= 0) for (int k = 0; k < SCAN_TRACE_WORDS; k++) ts.dbg[8 + k] = __hip_atomic_load(ts.trace + (size_t)t * SCAN_TRACE_WORDS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ts.dbg[12] = wall_clock64(); ts.dbg[13] = t_wait0;
                        }
                        if (lane == 0) atomicMax(err, DG_E_SCAN);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        if (lane == 0) {
            ts_publish(ts, tile, 2, (uint32_t)sx + own.x, (uint32_t)sy + own.y, (uint32_t)(sz + own.z), (uint32_t)sw + own.w);
            if (ts.trace) { unsigned long long *q = ts.trace + (size_t)tile * SCAN_TRACE_WORDS + 3; *q = (*q & 0xFF00000000000000ull) | (wall_clock64() & 0x00FFFFFFFFFFFFFFull); }
            sh[0] = sx; sh[1] = sy; sh[2] = sz; sh[3] = sw;
        }
    }
    __syncthreads();
    Triple r; r.x = (uint32_t)sh[0]; r.y = (uint32_t)sh[1]; r.z = sh[2]; r.w = (uint32_t)sh[3];
    __syncthreads();
    return r;
}

// exclusive prefix of four per-thread counts inside a 256-thread workgroup + the workgroup totals.  sh = 4 * 4 u64 of LDS.
__device__ inline Triple d_block_exclusive(Triple v, Triple &total, unsigned long long *sh)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    uint64_t x = v.x, y = v.y, z = v.z, q = v.w;
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t px = __shfl_up(x, o, 64), py = __shfl_up(y, o, 64), pz = __shfl_up(z, o, 64), pq = __shfl_up(q, o, 64);
        if (lane >= o) { x += px; y += py; z += pz; q += pq; }
    }
    if (lane == 63) { sh[wv] = x; sh[4 + wv] = y; sh[8 + wv] = z; sh[12 + wv] = q; }
    __syncthreads();
    uint64_t bx = 0, by = 0, bz = 0, bq = 0, tx = 0, ty = 0, tz = 0, tq = 0;
    for (int w = 0; w < nw; w++) {
        if (w < wv) { bx += sh[w]; by += sh[4 + w]; bz += sh[8 + w]; bq += sh[12 + w]; }
        tx += sh[w]; ty += sh[4 + w]; tz += sh[8 + w]; tq += sh[12 + w];
    }
    __syncthreads();
    total.x = (uint32_t)tx; total.y = (uint32_t)ty; total.z = tz; total.w = (uint32_t)tq;
    Triple r; r.x = (uint32_t)(bx + x - v.x); r.y = (uint32_t)(by + y - v.y); r.z = bz + z - v.z; r.w = (uint32_t)(bq + q - v.w);
    return r;
}
