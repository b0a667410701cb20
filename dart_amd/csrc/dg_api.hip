// dart_amd/csrc/dg_api.hip -- libdartgpu.so: the C ABI of include/dartgpu.h over the HIP kernels.
// gfx950 (MI355X) only; no CPU fallback anywhere: every entry point needs a live HIP device.
#include "../../include/dartgpu.h"
#include "dg_common.h"
#include "dg_fm.h"
#include "dg_seedq.h"
#include "dg_chain.h"
#include "dg_report.h"
#include "dg_pair.h"
#include "dg_reseed.h"
#include "dg_sort.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <vector>
#include <atomic>
#include <mutex>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>

static char g_init_error[512] = "";

template <typename T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

#define N_TIMERS 20
#define N_TOPS 128              // small device counters of a batch (bump tops, tickets, list sizes, class histogram), zeroed per run
enum { TOP_CIG = 0, TOP_SJ, TOP_JOBS, TOP_REPORT_MAIN, TOP_REPORT_JOBS, TOP_HEAVY_UNITS, TOP_SEED_NEXT, TOP_SEED_HEAVY,
       TOP_WORK = 16, TOP_TICKET_PAIR = 17, TOP_TICKET_EMIT = 18, TOP_RESEED_COUNT = 19 /* 19..21: items per ring size; 22: diagonals per chunk */, TOP_RESEED_TICKET = 23 /* 23..25 */, TOP_TICKET_SEED = 26, TOP_RS_POOL = 27, TOP_RS_OUT = 31,
       TOP_ORDER_INFO = 28 /* 28..30 */, TOP_CLASS_HIST = 32 /* 32..63 */, TOP_CLASS_FILL = 64 /* 64..95 */ };        // (explicit values: every index names its own word)

// The index of a device as its contexts see it: the root context owns it, dg_clone()d contexts point to it.  The look-up aids may still be
// under construction when the first batches run (DG_INIT_ASYNC_AIDS): the aids thread publishes a new DIndex (gen + 1) whenever one is
// complete, and every context compares its generation at the start of a run.
struct IndexShared {
    std::mutex mu;
    std::condition_variable cv;
    DIndex ix{};
    std::atomic<uint32_t> gen{1};
    std::thread aids;
    std::atomic<int> aids_state{0};          // 0 = no aids thread, 1 = allocating / building, 2 = complete, -1 = failed (the index works without)
    char aids_msg[256] = "";
    bool upload_done = false, upload_ok = false;
    void *d_ktab = nullptr, *d_sa_dense = nullptr;
    std::string report, report_out;
    int device = 0;
    // the re-seeding kernels' streams, shared by the contexts of this index (DG_S2_SHARED, make_ctx_objects)
    hipStream_t s2[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n_s2 = -1;                            // -1 = not decided yet
    std::atomic<int> n_ctx{0};
};

struct dg_ctx {
    int device = 0;
    IndexShared *shared_ix = nullptr; uint32_t ix_gen = 0;
    bool owns_index = true;       // false for dg_clone()d contexts: the index arrays belong to the parent
    hipStream_t stream = nullptr, stream2 = nullptr;
    bool owns_stream2 = true;
    hipEvent_t ev_dl = nullptr, ev_dl_block = nullptr;      // this context's place in the device's copy stream (copy_stream below); _block: the host thread sleeps (DG_BLOCKING_SYNC)
    bool dl_on_copy_stream = false;
    int env_copy_stream = 1;                           // DG_COPY_STREAM: 0 = the downloads on the context's own stream, 1 = on the device's copy stream
    hipEvent_t ev_prep = nullptr, ev_reseed0 = nullptr, ev_reseed1 = nullptr, ev_wait = nullptr;
    float reseed_ms = 0;
    char err[512] = "";
    DIndex ix{};
    DParams pr{};
    // index storage (the aids belong to shared_ix)
    void *d_bwt = nullptr, *d_sa = nullptr, *d_pac = nullptr, *d_lockey = nullptr, *d_locchr = nullptr, *d_chroff = nullptr;
    // batch inputs
    int n_reads = 0, max_rlen = 0;
    size_t seq_bytes = 0;
    bool enc_ready = false;       // the packed entry point filled enc itself (no k_encode)
    DBuf<unsigned char> seq; DBuf<uint32_t> seq_off; DBuf<uint16_t> rlen; DBuf<uint32_t> enc; DBuf<uint32_t> packed_in, nlist_in;
    // pipeline buffers; the four whose size depends on the data have sticky capacities (cap_*): a batch is enqueued against
    // them without asking the device for a size first, the device reports what it needed (DSizes) and flags an overflow
    // (d_err >= DG_ABORT), in which case the host grows the buffer and runs the batch again
    DBuf<DHit> hits; DBuf<uint32_t> nhits, nseeds, seed_off, ncand, rep_off, tile_sums, tile_read, slow_units;
    DBuf<SKey> seeds; DBuf<DSeed> work; DBuf<DCand> cands; DBuf<DJob> jobs; DBuf<unsigned long long> job_items, job_pool; DBuf<RsChunkOut> job_outs; DBuf<uint32_t> job_pool_next; DBuf<unsigned long long> items; DBuf<uint32_t> hist, heavy; DBuf<DHeavy> seed_heavy; DBuf<int32_t> seed_heavy_sfail; DBuf<int16_t> chain_picks; DBuf<uint8_t> chain_todo;
    DBuf<dg_read_out> reads_out; DBuf<dg_report_out> reports; DBuf<uint32_t> cigpool, cigfinal;
    DBuf<dg_sj_out> sjpool, sjfinal;
    DBuf<dg_read_c> reads_c; DBuf<dg_report_c> reports_c; DBuf<uint32_t> cig_c;     // compact records and their stored CIGAR ops (written by k_pair / k_emit_slow)
    DBuf<unsigned char> ws;
    DBuf<unsigned long long> scan_state, scan_trace;      // look-back state words / what every tile's workgroup last said about itself (dg_scan.h)
    int wall_khz = 100000;                                // rate of wall_clock64() on this device
    uint32_t scan_epoch = 0;      // number of the enqueued run, carried by every state word of its single-pass scans (dg_scan.h); never 0
    size_t cap_seeds = 0, cap_rep = 0, cap_work = 0, cap_cig = 0;
    // what any context of this index has learned about capacities (the root owns it, clones point to it): a clone does not have to overflow
    // and run its first batch again to find out what its siblings already know
    struct SharedCaps { std::atomic<size_t> seeds{0}, rep{0}, work{0}, cig{0}; } *shared_caps = nullptr;
    bool owns_shared_caps = false;
    unsigned long long *d_ctr = nullptr; unsigned int *d_tops = nullptr; int *d_err = nullptr; DSizes *d_sizes = nullptr; unsigned int *d_input_bad = nullptr;
    struct HostTail { DSizes sizes; int err; unsigned int input_bad; unsigned int tops[N_TOPS]; unsigned long long ctr[CTR_STRIDE]; __host__ __device__ uint32_t *sizes_words() { return (uint32_t *)&sizes; } } *h_tail = nullptr;   // page-locked; written by k_batch_end
    bool state_zeroed = false;   // the small device state (counters, tops, status, sizes) is zero: k_batch_end of the previous run left it so
    size_t used[3] = {0, 0, 0};
    bool enqueued = false;
    // timings
    hipEvent_t ev[N_TIMERS + 1]; const char *tname[N_TIMERS]; int n_t = 0; float tms[N_TIMERS];
    uint64_t counters[CTR_N];
    uint64_t reruns_capacity = 0, reruns_scan = 0;      // since dg_init / dg_clone
    bool want_compact = true, packed_valid = false;     // compact records: written by this run's kernels too / present for the batch that ran last
    bool want_full = true, full_valid = false;          // the full record types of the units k_pair finishes: written by this run (dg_map_batch_compact does not want them) / present for the batch that ran last
    int n_cu = 256, runs_of_last_batch = 0, attempt_no = 0;
    // environment switches, read once per context (not per batch)
    int env_seed_waves = 4, env_bail_trips = 0 /* 0: 64 trips in k_seed_qf (a trip there is up to three dependent accesses), 128 in the other two */, env_both = 0, env_report_bpc = 8, env_no_fast = 0, env_seed_legacy = 0, env_seed_slots_lg = 0, env_seed_wgs = 0, env_blocking_sync = 0;
    int env_seed_phases = 0, env_seed_wg_waves = 4, env_seed_partial = 32, env_seed_multi = 4;   // DG_SEED_PHASES=1: round 2's barrier-phased queue kernel (k_seed_q) instead of the free-running one (k_seed_qf)
    bool seed_qf_used = false;     // the last run's seeding kernel was k_seed_qf (its own-work counters are derived from its slot counts)
    int env_scan_mask = 7, env_one_stream = 1, env_packed_pair = 1, env_drain_bail = 0;
    int env_chain_bpc = 8, env_seedh_bpc = 8, env_reseed_pct = 100;      // persistent one-wave workgroups per CU of k_chain_heavy / k_seed_heavy; k_reseed's grids in per cent (sweeps: DG_CHAIN_BPC, DG_SEEDH_BPC, DG_RESEED_PCT)
    int env_rs_chunk = RS_CHUNK_DIAGS, env_rs_inline = RS_ENT_INLINE, env_rs_pool = 0;      // test hooks (DG_RS_CHUNK, DG_RS_ENT_MAX, DG_RS_POOL_BLOCKS): diagonals per chunk of a shared re-seeding window, entries a chunk record holds inline, blocks of the entry pool
    int env_scan_budget = 0;      // DG_SCAN_POLL_BUDGET: poll budget of a look-back on the FIRST attempt of a batch (test hook: forces the DG_E_SCAN re-run path)
};

static void read_env(dg_ctx *c)
{
    auto geti = [](const char *k, int dflt) { const char *v = getenv(k); return v ? atoi(v) : dflt; };
    c->env_seed_waves = geti("DG_SEED_WAVES", 4); c->env_bail_trips = geti("DG_SEED_BAIL_TRIPS", 0); c->env_both = geti("DG_SEED_BOTH", 0);
    c->env_seed_legacy = geti("DG_SEED_LEGACY", 0); c->env_seed_slots_lg = geti("DG_SEED_SLOTS_LG", 0); c->env_seed_wgs = geti("DG_SEED_WGS", 0); c->env_blocking_sync = geti("DG_BLOCKING_SYNC", 0);
    c->env_report_bpc = geti("DG_REPORT_BPC", 8); c->env_no_fast = geti("DG_NO_FAST_PAIR", 0);
    c->env_scan_budget = geti("DG_SCAN_POLL_BUDGET", 0); c->env_scan_mask = geti("DG_SCAN_POLL_SCANS", 7);
    c->env_seed_phases = geti("DG_SEED_PHASES", 0); c->env_seed_wg_waves = geti("DG_SEED_WG_WAVES", 4); c->env_seed_partial = geti("DG_SEED_PARTIAL_MIN", 32);
    c->env_copy_stream = geti("DG_COPY_STREAM", 1); c->env_one_stream = geti("DG_ONE_STREAM", 1); c->env_packed_pair = geti("DG_PACKED_PAIR", 1); c->env_drain_bail = geti("DG_SEED_DRAIN_BAIL", 0);
    c->env_chain_bpc = std::max(1, geti("DG_CHAIN_BPC", 8)); c->env_seedh_bpc = std::max(1, geti("DG_SEEDH_BPC", 8)); c->env_reseed_pct = std::max(10, geti("DG_RESEED_PCT", 100));
    c->env_rs_chunk = geti("DG_RS_CHUNK", RS_CHUNK_DIAGS); c->env_rs_chunk = std::min(1 << 24, std::max(RS_SUPER, c->env_rs_chunk / RS_SUPER * RS_SUPER));      // (a multiple of the pac super-chunk)
    c->env_rs_inline = std::min(RS_ENT_INLINE, std::max(0, geti("DG_RS_ENT_MAX", RS_ENT_INLINE))); c->env_rs_pool = std::max(0, geti("DG_RS_POOL_BLOCKS", 0));
    c->env_seed_multi = geti("DG_SEED_MULTI", 4); if (c->env_seed_multi < 0 || c->env_seed_multi > SQF_MULTI_MAX) c->env_seed_multi = SQF_MULTI_MAX;   // rows of an interval that are located and compared with the text at once (0: single rows only)
}

// a context adopts the aids that were completed since its last run
static inline void refresh_index(dg_ctx *c)
{
    IndexShared *sh = c->shared_ix;
    if (sh && c->ix_gen != sh->gen.load(std::memory_order_acquire)) { std::lock_guard<std::mutex> lk(sh->mu); c->ix = sh->ix; c->ix_gen = sh->gen.load(); }
}

static int fail(dg_ctx *c, int code, const char *what, hipError_t e)
{
    snprintf(c ? c->err : g_init_error, 512, "%s: %s", what, e == hipSuccess ? "error" : hipGetErrorString(e));
    return code;
}
#define HIPCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail(c, DG_ERR_HIP, #call, _e); } while (0)

// ------------------------------------------------------------------------------------------
// small utility kernels: exclusive scan in three phases (the index builder's sorter, dg_sort_pairs; the mapping path scans in single passes, dg_scan.h)
// ------------------------------------------------------------------------------------------
#define SCAN_TILE 2048
__global__ void __launch_bounds__(256) k_scan_tiles(const uint32_t *in, uint32_t *out, uint32_t *tile_sums, uint32_t n)
{
    __shared__ uint32_t sh[256];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
    uint32_t v[8], sum = 0;
    for (int i = 0; i < 8; i++) { v[i] = base + i < n ? in[base + i] : 0; sum += v[i]; }
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - sum;
    for (int i = 0; i < 8; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = sh[255];
}
__global__ void __launch_bounds__(256) k_scan_top(uint32_t *tile_sums, uint32_t n_tiles, uint32_t *total_out, uint32_t *total_copy, uint32_t cap, int *err, int code)
{
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < n_tiles; b += 256) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_tiles) tile_sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (total_copy) *total_copy = carry;
        if (err && carry > cap) atomicMax(err, code);          // the consumers of this scan return at once (DG_ABORT)
    }
}
__global__ void __launch_bounds__(256) k_scan_add(uint32_t *out, const uint32_t *tile_sums, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += tile_sums[i / SCAN_TILE];
}

// seeds per read -> where every read's seeds start (exclusive scan; seed_off[n] = total, flagged when it exceeds the capacity), the read
// that owns the first seed of every 64-seed tile (k_locate's entry point) and the list of units with more seeds than a lane of k_pair
// holds (k_chain_heavy's work list) -- ONE launch with a single-pass scan (dg_scan.h).  Round 2: three scan launches, k_tile_reads and
// k_heavy_list, each a pass over the 2 M counts.  Thread = 8 consecutive reads (so both mates of a pair are in one thread).
#define SO_PER 8
__global__ void __launch_bounds__(256)
k_seed_offsets(int n_reads, int paired, const uint32_t *__restrict__ nseeds, uint32_t *__restrict__ seed_off, uint32_t *__restrict__ tile_read, uint32_t n_tile_read,
               uint32_t *__restrict__ heavy_list, unsigned int *n_heavy, uint32_t *total_copy, uint32_t cap, TileScan ts, int *err)
{
    __shared__ unsigned long long s_scan[20];
    __shared__ unsigned int s_tile;
    // a status raised before this launch (the seeding kernel's safety net): nseeds cannot be trusted.  DG_E_SEEDS is raised below by the workgroup with the LAST
    // ticket, when every other workgroup is past this line; DG_E_SCAN may come from a look-back of this very launch: the decision is thread 0's (d_tile_ticket_unless)
    const unsigned int tile = d_tile_ticket_unless(ts, &s_tile, threadIdx.x == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= DG_ABORT);
    if (tile == SCAN_LEAVE) return;
    const uint32_t first = tile * (256u * SO_PER) + threadIdx.x * SO_PER;
    uint32_t v[SO_PER], sum = 0;
    if (first + SO_PER <= (uint32_t)n_reads) {
        const uint4 a = *(const uint4 *)(nseeds + first), b = *(const uint4 *)(nseeds + first + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < SO_PER; i++) v[i] = first + i < (uint32_t)n_reads ? nseeds[first + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < SO_PER; i++) sum += v[i];
    Triple mine, tot;
    mine.x = sum; mine.y = 0; mine.z = 0; mine.w = 0;
    const Triple inb = d_block_exclusive(mine, tot, s_scan);
    const Triple base = d_tile_exclusive(ts, tile, tot, s_scan + 16, err);
    uint32_t run = base.x + inb.x;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < SO_PER; i++) {
        const uint32_t r = first + i;
        if (r < (uint32_t)n_reads) {
            seed_off[r] = run;
            for (uint32_t t = (run + 63u) >> 6; (t << 6) < run + v[i] && t < n_tile_read; t++) tile_read[t] = r;       // (a read with seeds; tiles beyond the capacity: the batch runs again)
        }
        // units with more seeds than k_pair's lanes hold: a pair = reads 2k, 2k+1 of this thread; order of the list is irrelevant
        const bool unit_end = paired ? (i & 1) == 1 : true;
        const uint32_t useeds = paired ? ((i & 1) ? v[i - ((i & 1) ? 1 : 0)] + v[i] : 0u) : v[i];
        const bool heavy = unit_end && r < (uint32_t)n_reads && useeds > (uint32_t)UNIT_MAX_SEEDS;
        const unsigned long long m = __ballot(heavy);
        if (m) {
            unsigned int hb = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) hb = atomicAdd(n_heavy, (unsigned int)__popcll(m));
            hb = (unsigned int)__shfl((int)hb, leader, 64);
            if (heavy) heavy_list[hb + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = paired ? r >> 1 : r;
        }
        run += v[i];
    }
    if (tile == gridDim.x - 1 && threadIdx.x == 0) {
        const uint32_t total = base.x + tot.x;
        seed_off[n_reads] = total;
        if (total_copy) *total_copy = total;
        if (total > cap) atomicMax(err, DG_E_SEEDS);       // the consumers of these offsets return at once (DG_ABORT); the host grows the buffer
    }
}

static std::atomic<uint64_t> g_phase_ns[4];
static void caps_publish(std::atomic<size_t> &a, size_t v) { size_t cur = a.load(); while (cur < v && !a.compare_exchange_weak(cur, v)) { } }
static void caps_adopt(size_t &mine, const std::atomic<size_t> &a) { const size_t v = a.load(); if (v > mine) mine = v; }
// waiting for a context's stream on the per-batch path.  DG_BLOCKING_SYNC=1: through an event created with hipEventBlockingSync, so the
// host thread sleeps instead of spinning (one thread per context: a dozen spinning threads per GPU is a dozen busy cores)
// One copy stream per device, with the highest priority, shared by all contexts of the device (dg_clone), for the downloads.  A context's
// download is four copies; on its own stream each of them is a blit launch that queues among a dozen batches' kernels (a third of a
// context's cycle was spent there).  On the device's copy stream they run ahead of the kernels and one batch after the other: 853
// against 797-838 M reads/s (profiles/r03/h_copy_stream.txt).  One such stream PER CONTEXT is far worse (621): too many queues; the
// uploads on a stream of the same kind (the context's stream waiting for an event): 827 against 850.
// The host has waited for the run before it downloads, so the copy stream needs no dependency on the context's stream; the context waits
// for its own event behind its copies.  The streams live as long as the process.
static std::mutex g_copy_mu[64];
static hipStream_t g_copy_stream[64][2];
static hipError_t copy_stream(int device, int dir, hipStream_t *out)
{
    hipStream_t &sl = g_copy_stream[device & 63][dir];
    if (!sl) {
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&sl, hipStreamNonBlocking, hi);
        if (e != hipSuccess) { sl = nullptr; return e; }
    }
    *out = sl;
    return hipSuccess;
}

// The shared copy stream pays off for page-locked destinations (true asynchronous DMA).  A copy into ordinary memory is staged and blocks its caller --
// under the device-wide mutex every other context's download would wait behind it (`dart`'s streaming path passes ordinary arrays): those go
// through the context's own stream.
static bool host_page_locked(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static hipError_t wait_stream(dg_ctx *c)
{
    if (!c->env_blocking_sync || !c->ev_wait) return hipStreamSynchronize(c->stream);
    const hipError_t e = hipEventRecord(c->ev_wait, c->stream);
    return e != hipSuccess ? e : hipEventSynchronize(c->ev_wait);
}

// ------------------------------------------------------------------------------------------
// k_report: persistent waves; one lane = one read at a time (GenMappingReport,
// AlignmentCandidates.cpp:1079-1207); results go to the read's dg_read_out / dg_report_out slots
// ------------------------------------------------------------------------------------------
#ifdef DG_PROFILE_CLASSES          // diagnostic build only (profiles/probes/class_profile.sh): shader cycles per cost class
__device__ unsigned long long g_class_cycles[DG_COST_CLASSES + 1], g_class_chunks[DG_COST_CLASSES + 1];
#endif
template <int MINW>
__global__ void __launch_bounds__(64, MINW)
k_report(const DIndex ix, const DParams pr, int n_reads, int paired, const unsigned char *__restrict__ seq,
         const uint32_t *__restrict__ seq_off, const uint16_t *__restrict__ rlen, const uint32_t *__restrict__ seed_off,
         const DJob *__restrict__ jobs, DCand *__restrict__ cands,
         const uint32_t *__restrict__ rep_off, DSeed *__restrict__ work, const unsigned long long *__restrict__ items,
         const uint32_t *__restrict__ n_jobitems_p, const uint32_t *__restrict__ n_items_p, int job_part, dg_report_out *__restrict__ reports, uint32_t *cigpool, uint32_t cigcap,
         unsigned int *tops, unsigned char *ws, const WSLayout L, unsigned long long *ctr, int *err)
{
    const unsigned long long t_wave0 = wall_clock64();
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t lds_pm[64 * PM_LDS_WORDS];
    if (*err >= DG_ABORT) return;
    LaneCtx cx;
#ifdef DG_PROFILE_CLASSES
    __shared__ unsigned long long ph_acc[DG_NCLS * DG_NPHASE], cls_acc[2 * DG_NCLS];
    for (int q = threadIdx.x; q < DG_NCLS * DG_NPHASE; q += 64) ph_acc[q] = 0;
    for (int q = threadIdx.x; q < 2 * DG_NCLS; q += 64) cls_acc[q] = 0;
    __syncthreads();
    cx.ph = ph_acc;
#endif
    cx.lds = lds_pm + (threadIdx.x & 63) * PM_LDS_WORDS;
    cx.ix = &ix; cx.pr = &pr; cx.L = &L;
    cx.ws = ws + (size_t)lane * L.stride;
    cx.n_nw = cx.nw_cells = cx.n_reseed = cx.reseed_w = 0;
    // items = [candidates that wait for k_reseed | all other live candidates of the listed reads, heaviest class first] (k_order_items); this launch takes
    // one part, 64 items per ticket: lane = candidate.  (Rounds 1-4: lane = read, and the reads with dozens of candidates a wave each.)
    const unsigned int n_jobitems = *n_jobitems_p;
    const unsigned int lo = job_part ? 0u : n_jobitems, hi = job_part == 1 ? n_jobitems : *n_items_p;       // job_part 2: the whole list (k_reseed has run)
    unsigned int *next = tops + (job_part == 1 ? 4 : 3);
    while (true) {
        unsigned int ticket = 0;
        if ((threadIdx.x & 63) == 0) ticket = atomicAdd(next, 1u);
        ticket = (unsigned int)__shfl((int)ticket, 0, 64);
        const unsigned int base = lo + ticket * 64u;
        if (base >= hi) break;
        const unsigned int idx = base + (threadIdx.x & 63);
        const bool valid = idx < hi;                               // every lane enters d_gen_mapping_report (it has wave-wide steps)
        const unsigned long long item = valid ? items[idx] : 0ull;
        const int r = (int)(item >> 32);
        const uint32_t ci = (uint32_t)item, rep_i = valid ? rep_off[r] + (ci - seed_off[r]) : 0u;
        DRead rd;
        rd.sub_score = 0; rd.mis_num = 0; rd.mapq = 0;
        cx.seq = seq + seq_off[r]; cx.rlen = rlen[r];
#ifdef DG_PROFILE_CLASSES
        const long long t_begin = clock64();
        cx.cls = (int)(cands[(uint32_t)items[base]].final_n & 31); cx.t_last = t_begin;
#endif
        d_gen_mapping_report(cx, valid, paired ? (r & 1) == 0 : true, rd, cands + ci, valid ? 1 : 0, jobs, work,
                             reports + rep_i, rep_i, cigpool, tops + 0, cigcap, err, 0, 1, true);
#ifdef DG_PROFILE_CLASSES
        if ((threadIdx.x & 63) == 0) { cls_acc[cx.cls] += (unsigned long long)(clock64() - t_begin); cls_acc[DG_NCLS + cx.cls] += 1ull; }
#endif
    }
#ifdef DG_PROFILE_CLASSES
    __syncthreads();
    for (int q = threadIdx.x; q < DG_NCLS * DG_NPHASE; q += 64) if (ph_acc[q]) atomicAdd(&g_phase[q / DG_NPHASE][q % DG_NPHASE], ph_acc[q]);
    if (threadIdx.x < DG_NCLS && cls_acc[DG_NCLS + threadIdx.x]) { atomicAdd(&g_class_cycles[threadIdx.x], cls_acc[threadIdx.x]); atomicAdd(&g_class_chunks[threadIdx.x], cls_acc[DG_NCLS + threadIdx.x]); }
#endif
    d_wave_add(ctr + CTR_NW, cx.n_nw);
    d_wave_add(ctr + CTR_NWCELLS, cx.nw_cells);
    d_wave_add(ctr + CTR_RESEED, cx.n_reseed);
    d_wave_add(ctr + CTR_RESEEDW, cx.reseed_w);
    d_wave_resident(ctr, CTR_WT_REPORT, t_wave0);
}

// ------------------------------------------------------------------------------------------
// k_finalize: one lane = one unit of the general path (slow_units[]): CheckPairedFinalAlignments, FLAGs, MAPQ,
// splice-junction tuples (Mapping.cpp:74-206,479-565,615-621) on the records k_report left in memory
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_finalize(const DIndex ix, const DParams pr, int paired, const uint32_t *__restrict__ slow_units, const DSizes *__restrict__ sizes,
           const uint32_t *__restrict__ seed_off, const DCand *__restrict__ cands, const uint32_t *__restrict__ ncand, const uint32_t *__restrict__ rep_off, const DSeed *__restrict__ work,
           dg_read_out *__restrict__ rout, dg_report_out *__restrict__ reports, dg_sj_out *sjpool, uint32_t sjcap, unsigned int *tops, int *err)
{
    if (*err >= DG_ABORT) return;
    const unsigned int n_units = sizes->n_slow_units;
    for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < n_units; it += gridDim.x * blockDim.x) {
        const int u = (int)slow_units[it];
        const int nm = paired ? 2 : 1;
        DRead rd[2];
        dg_read_out o[2];
        RepMem<dg_report_out> rp[2];
        for (int m = 0; m < nm; m++) {
            const int r = paired ? 2 * u + m : u;
            const int nc = (int)ncand[r];
            dg_report_out *rep = reports + rep_off[r];
            // the running best / second best of GenMappingReport :1161-1172, in candidate order, over what k_report's lanes (one candidate each) left:
            // aln_score, and the candidate's mismatch count parked in `flag`.  SURVEY F6: sub_score, mis_num, mapq start at 0
            rd[m].score = 0; rd[m].sub_score = 0; rd[m].mis_num = 0; rd[m].mapq = 0; rd[m].iBest = 0; rd[m].CanNum = nc > 0 ? nc : 1;
            for (int i = 0; i < nc; i++) {
                const int aln = rep[i].aln_score, mis_num = rep[i].flag;
                if (mis_num) rep[i].flag = 0;                       // (also where the reference zeroes aln late, :1157: the parked value must not stay)
                if (aln <= 0) continue;                            // such an aln changes nothing: score and sub_score are 0 or stay
                if (aln > rd[m].score) { rd[m].iBest = i; rd[m].mis_num = mis_num; rd[m].sub_score = rd[m].score; rd[m].score = aln; }
                else if (aln == rd[m].score) rd[m].sub_score = rd[m].score;
            }
            o[m].score = rd[m].score; o[m].sub_score = rd[m].sub_score; o[m].mis_num = rd[m].mis_num; o[m].mapq = 0;
            o[m].n_rep = rd[m].CanNum; o[m].best = rd[m].iBest; o[m].rep_off = (int32_t)rep_off[r]; o[m].sj_off = 0; o[m].n_sj = 0;
            rp[m].p = rep;
        }
        if (paired) {
            d_settle_pair(pr, rd[0], rp[0], rd[1], rp[1]);
            d_flag_pair(rd[0], rp[0], rd[1], rp[1]);
        } else d_flag_single(rd[0], rp[0]);
        for (int m = 0; m < nm; m++) {
            const int r = paired ? 2 * u + m : u;
            d_mapq(rd[m], rp[m]);
            o[m].score = rd[m].score; o[m].mapq = rd[m].mapq; o[m].best = rd[m].iBest;
            if (ncand[r] > 0 && (rd[m].mapq == 50 || (pr.all_sj && rd[m].score > 0)))
                d_collect_sj(ix, pr, cands[seed_off[r] + rd[m].iBest], work, r, sjpool, tops + TOP_SJ, sjcap, o[m].sj_off, o[m].n_sj, err);
            rout[r] = o[m];
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_emit_slow: the variable-length records of the general path's reads, laid out deterministically behind k_pair's: one lane
// = one read of slow_units[] (ascending); a single-pass scan (dg_scan.h) of (CIGAR ops, junction tuples) per read gives the
// places; CIGAR ops move from the report kernel's pool to cigfinal[cig_fast + ...], tuples from the bump pool to sjfinal.
// The grid covers the worst case (every unit on the list); workgroups beyond the list leave at once.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_emit_slow(int paired, const uint32_t *__restrict__ slow_units, DSizes *sizes, dg_read_out *__restrict__ rout, dg_report_out *__restrict__ reports,
            const uint32_t *__restrict__ cigpool, uint32_t *__restrict__ cigfinal, uint32_t cap_cig, const dg_sj_out *__restrict__ sjpool, dg_sj_out *__restrict__ sjfinal,
            TileScan ts, int *err, const uint16_t *__restrict__ rlen, const CompactOut co)
{
    __shared__ unsigned long long s_scan[20];
    __shared__ unsigned int s_tile;
    bool leave = false;                                           // (DG_E_CIGFINAL may have been raised by an earlier workgroup of this launch; thread 0 decides for the workgroup)
    if (threadIdx.x == 0) { const int e0 = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); leave = e0 >= DG_ABORT && e0 != DG_E_CIGFINAL; }
    const unsigned int tile = d_tile_ticket_unless(ts, &s_tile, leave);
    if (tile == SCAN_LEAVE) return;
    const unsigned int n_items = sizes->n_slow_units * (paired ? 2u : 1u);
    if (tile > 0 && tile * 256u >= n_items) return;
    const unsigned int it = tile * 256u + threadIdx.x;
    const bool on = it < n_items;
    const int r = on ? (paired ? (int)(2u * slow_units[it >> 1] + (it & 1u)) : (int)slow_units[it]) : 0;
    dg_read_out o;
    o.n_rep = 0; o.rep_off = 0; o.n_sj = 0; o.sj_off = 0;
    if (on) o = rout[r];
    uint32_t ncig = 0, ncigc = 0;
    const uint32_t full_match = on ? (uint32_t)rlen[r] << 4 : 0u;                      // "<rlen>M": not stored in the compact op array
    for (int i = 0; i < (on ? o.n_rep : 0); i++) {
        const dg_report_out &rp = reports[o.rep_off + i];
        ncig += rp.n_cigar;
        if (co.reads) ncigc += (rp.n_cigar == 1u && cigpool[rp.cigar_off] == full_match) ? 0u : rp.n_cigar;
    }
    Triple mine, tot;
    mine.x = ncig; mine.y = on ? (uint32_t)o.n_sj : 0u; mine.z = 0; mine.w = ncigc;
    const Triple inb = d_block_exclusive(mine, tot, s_scan);
    const Triple base = d_tile_exclusive(ts, tile, tot, s_scan + 16, err);
    const uint32_t cig_fast = sizes->cig_fast, cigc_fast = sizes->pad[2];
    if (on) {
        uint32_t dst = cig_fast + base.x + inb.x, dstc = cigc_fast + base.w + inb.w;
        bool fits = true;
        if ((uint64_t)dst + ncig > cap_cig) atomicMax(err, DG_E_CIGFINAL);
        else for (int i = 0; i < o.n_rep; i++) {
            dg_report_out &rp = reports[o.rep_off + i];
            const uint32_t m = rp.n_cigar, src = rp.cigar_off;
            const bool plain = m == 1u && cigpool[src] == full_match;
            for (uint32_t k = 0; k < m; k++) cigfinal[dst + k] = cigpool[src + k];
            if (co.reads) {                                        // the compact form of the same report; its stored ops lie in the op array's second region
                if (!plain) { for (uint32_t k = 0; k < m; k++) co.cigar[dstc + k] = cigpool[src + k]; dstc += m; }
                dg_report_c q; fits = d_compact_report(rp, plain, true, q) && fits; co.reports[o.rep_off + i] = q;
            }
            rp.cigar_off = dst; dst += m;
        }
        if (o.n_sj > 0) {
            const uint32_t sd = base.y + inb.y;
            for (int k = 0; k < o.n_sj; k++) sjfinal[sd + k] = sjpool[o.sj_off + k];
            rout[r].sj_off = (int32_t)sd;
        }
        if (co.reads) { dg_read_c oc; fits = d_compact_read(o, oc) && fits; co.reads[r] = oc; if (!fits) *co.bad = 1u; }
    }
    const unsigned int last = n_items ? (n_items - 1u) / 256u : 0u;
    if (tile == last && threadIdx.x == 0) { sizes->total_cig = cig_fast + base.x + tot.x; sizes->total_sj = base.y + tot.y; sizes->pad[0] = cigc_fast + base.w + tot.w; }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
extern "C" void dg_params_default(dg_params *p)
{
    p->max_gaps = 5; p->max_dup = 100; p->max_intron = 500000; p->min_intron = 5;
    p->max_mismatch = 0; p->multi_hit = 0; p->all_sj = 0; p->paired = 0;
}

static void set_params(dg_ctx *c, const dg_params *p)
{
    c->pr.max_gaps = p->max_gaps; c->pr.max_dup = p->max_dup; c->pr.max_intron = p->max_intron; c->pr.min_intron = p->min_intron;
    c->pr.max_mismatch = p->max_mismatch; c->pr.multi_hit = p->multi_hit; c->pr.all_sj = p->all_sj; c->pr.paired = p->paired;
}

extern "C" int dg_set_params(dg_ctx *c, const dg_params *p)
{
    if (!c || !p) return DG_ERR_ARG;
    set_params(c, p);
    read_env(c);                    // the DG_* tuning switches are read here and at init, never per batch
    return DG_OK;
}

extern "C" int dg_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" const char *dg_last_error(const dg_ctx *c) { return c ? c->err : g_init_error; }

extern "C" void dg_destroy(dg_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->owns_index && c->shared_ix) for (hipStream_t &q : c->shared_ix->s2) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); q = nullptr; }   // (clones go before their parent)
    if (c->owns_index && c->shared_ix) {          // the aids thread may still be allocating or building: it ends by itself (a failed upload tells it so)
        IndexShared *sh = c->shared_ix;
        { std::lock_guard<std::mutex> lk(sh->mu); if (!sh->upload_done) { sh->upload_done = true; sh->upload_ok = false; } }
        sh->cv.notify_all();
        if (sh->aids.joinable()) sh->aids.join();
        if (sh->d_ktab) (void)hipFree(sh->d_ktab);
        if (sh->d_sa_dense) (void)hipFree(sh->d_sa_dense);
        delete sh; c->shared_ix = nullptr;
    }
    void *ptrs[] = { c->d_bwt, c->d_sa, c->d_pac, c->d_lockey, c->d_locchr, c->d_chroff };
    if (c->owns_index) for (void *p : ptrs) if (p) (void)hipFree(p);
    void *own[] = { c->d_ctr, c->d_tops, c->d_err, c->d_sizes, c->d_input_bad };
    for (void *p : own) if (p) (void)hipFree(p);
    if (c->h_tail) (void)hipHostFree(c->h_tail);
    c->seq.release(); c->seq_off.release(); c->rlen.release(); c->enc.release(); c->packed_in.release(); c->nlist_in.release(); c->hits.release(); c->nhits.release(); c->nseeds.release();
    c->seed_off.release(); c->ncand.release(); c->rep_off.release(); c->tile_sums.release(); c->tile_read.release(); c->slow_units.release();
    c->seeds.release(); c->work.release(); c->cands.release(); c->jobs.release(); c->job_items.release(); c->job_outs.release(); c->job_pool.release(); c->job_pool_next.release(); c->items.release(); c->hist.release(); c->heavy.release(); c->seed_heavy.release(); c->seed_heavy_sfail.release(); c->chain_picks.release(); c->chain_todo.release();
    c->reads_out.release(); c->reports.release(); c->cigpool.release(); c->cigfinal.release(); c->sjpool.release(); c->sjfinal.release();
    c->ws.release(); c->scan_state.release(); c->scan_trace.release(); c->reads_c.release(); c->reports_c.release(); c->cig_c.release();
    for (int i = 0; i <= N_TIMERS; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_prep) (void)hipEventDestroy(c->ev_prep);
    if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
    if (c->ev_dl) (void)hipEventDestroy(c->ev_dl);
    if (c->ev_dl_block) (void)hipEventDestroy(c->ev_dl_block);
    if (c->ev_reseed0) (void)hipEventDestroy(c->ev_reseed0);
    if (c->ev_reseed1) (void)hipEventDestroy(c->ev_reseed1);
    if (c->stream2 && c->owns_stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->owns_shared_caps) delete c->shared_caps;
    delete c;
}

// per-context objects that dg_init and dg_clone both need
static hipError_t make_ctx_objects(dg_ctx *c)
{
    hipError_t e;
    for (int i = 0; i <= N_TIMERS; i++) c->ev[i] = nullptr;
    // DG_CU_PARTS=P (measurement switch, profiles/r04/y_*): context k's two streams only get the compute units of part k % P (DG_CU_LAYOUT 0: P contiguous
    // ranges of the mask's bits, 1: bit i belongs to part (i % 8) % P) -- fewer different kernels share a CU's instruction cache and LDS at a time
    static std::atomic<int> n_made{0};
#ifdef DG_EXPERIMENTS      /* measurement builds only (profiles/r04/x_modes_*): a stray environment variable must not take HBM away from a product run */
    if (getenv("DG_EXP_CTX_PAD_KB")) {         // context k's allocations start behind a pad of (k + 1) x this many KB (never freed: an experiment)
        void *pad = nullptr; static std::atomic<int> n_pad{0};
        (void)hipMalloc(&pad, (size_t)(n_pad.fetch_add(1) + 1) * (size_t)atoll(getenv("DG_EXP_CTX_PAD_KB")) * 1024);
    }
#endif
    const int parts = getenv("DG_CU_PARTS") ? atoi(getenv("DG_CU_PARTS")) : 0, layout = getenv("DG_CU_LAYOUT") ? atoi(getenv("DG_CU_LAYOUT")) : 0;
    if (parts > 1) {
        int n_cu = 0; (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        const int part = n_made.fetch_add(1) % parts;
        std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
        for (int i = 0; i < n_cu; i++) { const int pi = layout == 1 ? (i % 8) % parts : (int)((long long)i * parts / n_cu); if (pi == part) mask[(size_t)i / 32] |= 1u << (i % 32); }
        if ((e = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data())) != hipSuccess || (e = hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)mask.size(), mask.data())) != hipSuccess) return e;
    } else {
        // A device's contexts share a few streams for their re-seeding kernels (four launches per batch) instead of owning one each: the runtime maps streams
        // onto GPU_MAX_HW_QUEUES hardware queues, least-used first with ties broken by the queues' addresses -- with twelve contexts x two streams on sixteen
        // queues, which contexts' MAIN streams ended up on one queue (and then ran their kernels one after the other) differed from process to process:
        // 826 to 985 M reads/s for the same command (profiles/r04/x_modes_*).  Twelve main streams + DG_S2_SHARED (3) shared ones + the caller's own stream
        // fit the sixteen queues without sharing.  0 = a private stream per context (rounds 2-3).
        IndexShared *sh = c->shared_ix;
        if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return e;
        const bool one_stream = !getenv("DG_ONE_STREAM") || atoi(getenv("DG_ONE_STREAM")) != 0;      // (the default since round 5: no second stream exists at all -- a stream that is never used still takes its place among the hardware queues)
        if (sh && !one_stream) {
            std::lock_guard<std::mutex> lk(sh->mu);
            if (sh->n_s2 < 0) { const int n = getenv("DG_S2_SHARED") ? atoi(getenv("DG_S2_SHARED")) : 3; sh->n_s2 = n < 0 ? 0 : (n > 8 ? 8 : n); }
            if (sh->n_s2 > 0) {
                const int k = sh->n_ctx.fetch_add(1) % sh->n_s2;
                if (!sh->s2[k] && (e = hipStreamCreate(&sh->s2[k])) != hipSuccess) return e;
                c->stream2 = sh->s2[k]; c->owns_stream2 = false;
            }
        }
        if (!one_stream && !c->stream2 && (e = hipStreamCreate(&c->stream2)) != hipSuccess) return e;
    }
    if ((e = hipEventCreate(&c->ev_prep)) != hipSuccess || (e = hipEventCreate(&c->ev_reseed0)) != hipSuccess || (e = hipEventCreate(&c->ev_reseed1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_wait, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_dl, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&c->ev_dl_block, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) return e;
    for (int i = 0; i <= N_TIMERS; i++) if ((e = hipEventCreate(&c->ev[i])) != hipSuccess) return e;
    if ((e = hipMalloc((void **)&c->d_ctr, CTR_STRIPES * CTR_STRIDE * 8)) != hipSuccess || (e = hipMalloc((void **)&c->d_tops, N_TOPS * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_err, 4)) != hipSuccess || (e = hipMalloc((void **)&c->d_sizes, sizeof(DSizes))) != hipSuccess ||
        (e = hipMalloc((void **)&c->d_input_bad, 4)) != hipSuccess) return e;
    if ((e = hipHostMalloc((void **)&c->h_tail, sizeof(dg_ctx::HostTail), hipHostMallocDefault)) != hipSuccess) return e;
    memset(c->h_tail, 0, sizeof(dg_ctx::HostTail));
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->wall_khz = khz; }
    read_env(c);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------
// start-up: index bytes -> HBM through page-locked staging chunks, look-up aids built beside (or behind) it
// ------------------------------------------------------------------------------------------
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// where a chunk's bytes come from: a file (pread: page cache -> staging chunk) or the caller's host array (memcpy)
struct UpSrc { int fd = -1; const uint8_t *mem = nullptr; };
struct UpChunk { UpSrc src; uint64_t src_off; uint8_t *dst; size_t bytes; uint64_t relayout_blocks; };

// Reader threads take chunks in order; each owns a copy stream and two page-locked staging buffers, so a chunk's read from the page
// cache overlaps the previous chunk's DMA, and the threads' reads overlap each other (one thread copies ~4 GB/s out of the page cache;
// the link takes 50).  The Occ blocks of a .bwt chunk are re-laid in place on the same stream, behind the chunk's copy.
static hipError_t upload_chunks(int device, const std::vector<UpChunk> &chunks, size_t chunk_cap, int n_threads, std::string &what)
{
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::mutex mu;
    hipError_t first = hipSuccess;
    auto fail_with = [&](hipError_t e, const char *w) { std::lock_guard<std::mutex> lk(mu); if (!failed.exchange(1)) { first = e == hipSuccess ? hipErrorUnknown : e; what = w; } };
    auto work = [&]() {
        hipError_t e;
        if ((e = hipSetDevice(device)) != hipSuccess) { fail_with(e, "hipSetDevice (index upload)"); return; }
        hipStream_t s = nullptr; uint8_t *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false};
        if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) { fail_with(e, "hipStreamCreate (index upload)"); return; }
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            if ((e = hipHostMalloc((void **)&pin[b], chunk_cap, hipHostMallocDefault)) == hipSuccess) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
        }
        if (e != hipSuccess) fail_with(e, "page-locked staging buffers (index upload)");
        int b = 0;
        while (e == hipSuccess && !failed.load()) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size()) break;
            const UpChunk &ch = chunks[i];
            if (used[b] && (e = hipEventSynchronize(ev[b])) != hipSuccess) { fail_with(e, "hipEventSynchronize (index upload)"); break; }
            if (ch.src.mem) memcpy(pin[b], ch.src.mem + ch.src_off, ch.bytes);
            else {
                size_t got = 0;
                while (got < ch.bytes) {
                    const ssize_t r = pread(ch.src.fd, pin[b] + got, ch.bytes - got, (off_t)(ch.src_off + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got < ch.bytes) { fail_with(hipErrorUnknown, "an index file is shorter than its header says"); break; }
            }
            if (ch.bytes && (e = hipMemcpyAsync(ch.dst, pin[b], ch.bytes, hipMemcpyHostToDevice, s)) != hipSuccess) { fail_with(e, "hipMemcpyAsync (index upload)"); break; }
            if (ch.relayout_blocks) {
                k_relayout_bwt<<<(unsigned)((ch.relayout_blocks + 255) / 256), 256, 0, s>>>((uint4 *)ch.dst, ch.relayout_blocks);
                if ((e = hipGetLastError()) != hipSuccess) { fail_with(e, "k_relayout_bwt"); break; }
            }
            if ((e = hipEventRecord(ev[b], s)) != hipSuccess) { fail_with(e, "hipEventRecord (index upload)"); break; }
            used[b] = true; b ^= 1;
        }
        if (s && (e = hipStreamSynchronize(s)) != hipSuccess) fail_with(e, "index upload");
        for (int k = 0; k < 2; k++) { if (ev[k]) (void)hipEventDestroy(ev[k]); if (pin[k]) (void)hipHostFree(pin[k]); }
        if (s) (void)hipStreamDestroy(s);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
    return failed.load() ? first : hipSuccess;
}

static void add_chunks(std::vector<UpChunk> &out, UpSrc src, uint64_t src_off, uint8_t *dst, size_t bytes, size_t chunk, bool relayout, uint64_t total_blocks)
{
    for (size_t o = 0; o < bytes || (relayout && o / 64 < total_blocks); o += chunk) {
        const size_t nb = o < bytes ? std::min(chunk, bytes - o) : 0;
        uint64_t rb = 0;
        if (relayout) { const uint64_t first = o / 64; rb = std::min<uint64_t>(chunk / 64, total_blocks > first ? total_blocks - first : 0); }
        out.push_back(UpChunk{src, src_off + o, dst + o, nb, rb});
    }
}

// the look-up aids (full suffix array, K-mer prefix table): allocated while the index loads, built once it is in HBM, on a stream of the
// lowest priority so that batches already being mapped (DG_INIT_ASYNC_AIDS) keep the GPU.  Each aid is published when complete.
static void aids_thread(IndexShared *sh, int sa_file_intv, uint64_t n_sa, bool async, uint64_t expected_reads)
{
    auto failed = [&](const char *w, hipError_t e) {
        std::lock_guard<std::mutex> lk(sh->mu);
        snprintf(sh->aids_msg, sizeof sh->aids_msg, "%s: %s", w, hipGetErrorString(e));
        sh->aids_state.store(-1); sh->cv.notify_all();
    };
    hipError_t e;
    if ((e = hipSetDevice(sh->device)) != hipSuccess) return failed("hipSetDevice (aids)", e);
    DIndex ix;
    { std::lock_guard<std::mutex> lk(sh->mu); ix = sh->ix; }
    const uint64_t seq_len = ix.seq_len;
    double t0 = now_s();
    // full suffix array (8 bytes per text symbol: 1 GB for chr20, 50 GB for a human genome -- this is what 288 GB are for): locating a row is
    // one load instead of a walk of up to 31 LF steps.  Texts too large for that fall back to every 2nd / 4th row; DG_SA_DENSE=0 turns it
    // off, =2/4/8/16 forces an interval
    // A job known to be short (dg_index_files::expected_reads) gets LEAN aids -- every 4th row of the SA and K <= 14: 17 GB instead of 118 GB for
    // a human genome, the seeding stage twice as long (1.85 + 0.25 against 0.97 + 0.10 ms per 2 M reads, profiles/r04/b_init_probe_aids_vs_startup.txt)
    // -- because the full ones cost more than they save below some hundred million reads: 0.7 s of build kernels, and up to 3-4 s of hipMalloc
    // when the device's memory was in use a moment ago (the driver clears VRAM behind a release; an allocation that needs those pages waits).
    const bool lean = expected_reads != 0 && expected_reads < 400000000ull;
    int intv = seq_len <= (12ull << 30) ? 1 : (seq_len <= (24ull << 30) ? 2 : 4);
    if (lean && intv < 4) intv = 4;
    if (const char *v = getenv("DG_SA_DENSE")) intv = atoi(v);
    const bool want_dense = intv >= 1 && intv < sa_file_intv && (intv & (intv - 1)) == 0 && seq_len < (1ull << 39);
    const uint64_t n_entries = want_dense ? seq_len / (uint64_t)intv + 1 : 0;
    int K = 8;
    while (K < 16 && (1ull << (2 * K)) < seq_len) K++;
    if (lean && K > 14) K = 14;
    if (async) {
        // the caller maps batches meanwhile: the device allocations below take a lock every other allocation of the process needs (a context's
        // first batch sizes its buffers) and, on recently used memory, seconds -- so they start only when the index itself is in HBM
        std::unique_lock<std::mutex> lk(sh->mu);
        sh->cv.wait(lk, [&]() { return sh->upload_done; });
        if (!sh->upload_ok) { lk.unlock(); sh->aids_state.store(-1); sh->cv.notify_all(); return; }
    }
    if (want_dense && (e = hipMalloc(&sh->d_sa_dense, n_entries * 8)) != hipSuccess) return failed("hipMalloc dense SA", e);
    // K-mer prefix table: the smallest K with 4^K >= text length (so that most K-mers are unique or absent and a search needs the table plus
    // a step or two), 8 <= K <= 16: 16 bytes per entry = 4.3 GB at K = 14 (chr20), 69 GB at K = 16 (human; K = 14: k_seed 1.31 instead of
    // 0.97 ms).  DG_KTAB_K=0 turns it off, =2..16 forces K.
    {   // the table must leave room for the batches in flight: step down while it would not leave 48 GB free
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (K > 8 && ((size_t)16 << (2 * K)) + ((size_t)48 << 30) > free_b) K--;
    }
    if (const char *v = getenv("DG_KTAB_K")) K = atoi(v);
    if (K > 16) K = 16;
    if (K >= 2 && (e = hipMalloc(&sh->d_ktab, ((size_t)16) << (2 * K))) != hipSuccess) return failed("hipMalloc k-mer table", e);
    const double t_alloc = now_s() - t0;
    int lo = 0, hi = 0;
    hipStream_t s = nullptr;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                    // (lo = the least priority)
    if ((e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo)) != hipSuccess) return failed("hipStreamCreate (aids)", e);
    {   // wait for the index bytes
        std::unique_lock<std::mutex> lk(sh->mu);
        sh->cv.wait(lk, [&]() { return sh->upload_done; });
        if (!sh->upload_ok) { lk.unlock(); (void)hipStreamDestroy(s); sh->aids_state.store(-1); sh->cv.notify_all(); return; }
    }
    double t_dense = 0, t_ktab = 0;
    // Built beside running batches (async), the two kernels take a quarter of the wave slots (4 workgroups of 256 per CU, grid-stride): the
    // stream's low priority alone did not keep them from crowding the batches out (a batch waited for the whole build, profiles/r04/a_*).
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, sh->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }
    int per_cu = 4;
    if (const char *v = getenv("DG_AIDS_WGS_PER_CU")) per_cu = std::max(1, atoi(v));
    const uint64_t grid_cap = async ? (uint64_t)n_cu * (uint64_t)per_cu : (uint64_t)1 << 22;
    if (want_dense) {
        t0 = now_s();
        k_build_sa_dense<<<(unsigned)std::min<uint64_t>((n_sa + 255) / 256, grid_cap), 256, 0, s>>>(ix, intv, n_entries, (uint64_t *)sh->d_sa_dense);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) { (void)hipStreamDestroy(s); return failed("k_build_sa_dense", e); }
        ix.sa_dense = (const uint64_t *)sh->d_sa_dense; ix.sa_dense_intv = intv; ix.sa_dense_shift = 0;
        for (int b = 0; (1 << b) < intv; b++) ix.sa_dense_shift = b + 1;
        { std::lock_guard<std::mutex> lk(sh->mu); sh->ix = ix; sh->gen.fetch_add(1, std::memory_order_release); }
        t_dense = now_s() - t0;
    }
    if (K >= 2) {
        t0 = now_s();
        const size_t entries = (size_t)1 << (2 * K);
        k_build_ktab<<<(unsigned)std::min<uint64_t>((entries + 255) / 256, grid_cap), 256, 0, s>>>(ix, K, (uint64_t *)sh->d_ktab);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(s)) != hipSuccess) { (void)hipStreamDestroy(s); return failed("k_build_ktab", e); }
        ix.ktab = (const uint64_t *)sh->d_ktab; ix.ktab_k = K;
        { std::lock_guard<std::mutex> lk(sh->mu); sh->ix = ix; sh->gen.fetch_add(1, std::memory_order_release); }
        t_ktab = now_s() - t0;
    }
    (void)hipStreamDestroy(s);
    {
        std::lock_guard<std::mutex> lk(sh->mu);
        char b[256];
        snprintf(b, sizeof b, "; %saids%s: hipMalloc %.3f s (full SA every %d row(s) %.1f GB, K=%d table %.1f GB), k_build_sa_dense %.3f s, k_build_ktab %.3f s",
                 lean ? "lean " : "", async ? " (built beside the first batches)" : "", t_alloc, want_dense ? intv : 0, n_entries * 8 / 1e9, K >= 2 ? K : 0, K >= 2 ? (double)((size_t)16 << (2 * K)) / 1e9 : 0.0, t_dense, t_ktab);
        sh->report += b;
        sh->aids_state.store(2);
    }
    sh->cv.notify_all();
}

struct IndexMeta { uint64_t bwt_words, primary, L2[5], seq_len, n_sa; int sa_intv; int64_t l_pac; int n_chr; const int64_t *chr_off, *chr_len; uint64_t expected_reads; };

// everything of dg_init / dg_init_files behind the argument checks: contexts objects, index arrays, the upload, the aids
static dg_ctx *init_index(const IndexMeta &m, UpSrc bwt, uint64_t bwt_off, UpSrc sa, uint64_t sa_off, uint64_t sa_first, uint64_t sa_count, UpSrc pac, size_t pac_copy,
                          const dg_params *p, int device, int flags, int *status)
{
    int ndev = 0;
    dg_ctx *c = nullptr;
    auto bail = [&](int code, const char *what, hipError_t e) -> dg_ctx * {
        snprintf(g_init_error, sizeof g_init_error, "%s: %s", what, e == hipSuccess ? "invalid" : hipGetErrorString(e));
        if (c) dg_destroy(c);
        if (status) *status = code;
        return nullptr;
    };
    const double t_start = now_s();
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return bail(DG_ERR_NO_DEVICE, "no HIP device (libdartgpu has no CPU fallback)", e);
    if (device < 0 || device >= ndev) return bail(DG_ERR_NO_DEVICE, "device ordinal out of range", hipSuccess);
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(DG_ERR_HIP, "hipSetDevice", e);
#ifdef DG_EXPERIMENTS      /* measurement builds only: where the allocations land (profiles/r04/x_modes_*) */
    if (getenv("DG_EXP_PREALLOC_GB")) {        // measurement switch (where the allocations land: profiles/r04/x_modes_*): take and give back this much memory first
        void *dummy = nullptr;
        if (hipMalloc(&dummy, (size_t)atoll(getenv("DG_EXP_PREALLOC_GB")) << 30) == hipSuccess) { (void)hipMemset(dummy, 0, 1 << 20); (void)hipDeviceSynchronize(); (void)hipFree(dummy); }
    }
    if (getenv("DG_EXP_HOLD_GB")) { void *hold = nullptr; (void)hipMalloc(&hold, (size_t)atoll(getenv("DG_EXP_HOLD_GB")) << 30); }      // (kept: shifts what follows)
    if (getenv("DG_EXP_CHURN_GB")) {           // this much memory in 4 GB blocks, every other one given back first, then the rest: a free list in pieces
        std::vector<void *> blk((size_t)atoll(getenv("DG_EXP_CHURN_GB")) / 4, nullptr);
        for (auto &b : blk) (void)hipMalloc(&b, (size_t)4 << 30);
        for (size_t i = 0; i < blk.size(); i += 2) (void)hipFree(blk[i]);
        for (size_t i = 1; i < blk.size(); i += 2) (void)hipFree(blk[i]);
    }
#endif
    c = new dg_ctx();
    c->device = device;
    c->shared_caps = new dg_ctx::SharedCaps(); c->owns_shared_caps = true;
    c->shared_ix = new IndexShared(); c->shared_ix->device = device;
    if ((e = make_ctx_objects(c)) != hipSuccess) return bail(DG_ERR_HIP, "stream / event / counter allocation", e);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
    set_params(c, p);
    const double t_ctx = now_s();

    // ---- index arrays: the .bwt blocks at a 64-byte aligned base (+ two blocks of padding so the last, possibly partial, block can be
    //      fetched whole), sampled SA, pac, chromosome keys
    const size_t bwt_bytes = (size_t)m.bwt_words * 4, sa_bytes = (size_t)m.n_sa * 8, pac_bytes = (size_t)(m.l_pac / 4 + 1);
    const uint64_t n_blocks = (m.seq_len + 127) / 128;
    if (bwt_bytes > (n_blocks + 1) * 64) return bail(DG_ERR_ARG, "the .bwt data is larger than its header's text length allows", hipSuccess);
    if ((e = hipMalloc(&c->d_bwt, (n_blocks + 2) * 64)) != hipSuccess || (e = hipMemsetAsync(c->d_bwt, 0, (n_blocks + 2) * 64, c->stream)) != hipSuccess) return bail(DG_ERR_HIP, "hipMalloc bwt", e);
    if ((e = hipMalloc(&c->d_sa, sa_bytes + 8)) != hipSuccess || (e = hipMemsetAsync(c->d_sa, 0, sa_bytes + 8, c->stream)) != hipSuccess) return bail(DG_ERR_HIP, "hipMalloc sa", e);
    if ((e = hipMalloc(&c->d_pac, pac_bytes + 4096)) != hipSuccess || (e = hipMemsetAsync(c->d_pac, 0, pac_bytes + 4096, c->stream)) != hipSuccess) return bail(DG_ERR_HIP, "hipMalloc pac", e);
    {
        const int n = m.n_chr;
        std::vector<int64_t> key(2 * n), off(n);
        std::vector<int32_t> chr(2 * n);
        for (int i = 0; i < n; i++) {       // ChrLocMap, bwt_index.cpp:246-250
            off[i] = m.chr_off[i];
            key[i] = m.chr_off[i] + m.chr_len[i] - 1; chr[i] = i;
            key[2 * n - 1 - i] = 2 * m.l_pac - m.chr_off[i] - 1; chr[2 * n - 1 - i] = i;
        }
        if ((e = hipMalloc(&c->d_lockey, 16 * (size_t)n)) != hipSuccess || (e = hipMalloc(&c->d_locchr, 8 * (size_t)n)) != hipSuccess ||
            (e = hipMalloc(&c->d_chroff, 8 * (size_t)n)) != hipSuccess) return bail(DG_ERR_HIP, "hipMalloc chr tables", e);
        if ((e = hipMemcpy(c->d_lockey, key.data(), 16 * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(c->d_locchr, chr.data(), 8 * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(c->d_chroff, off.data(), 8 * (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) return bail(DG_ERR_HIP, "upload chr tables", e);
        if (sa_first) {                    // (from the file: sa[0] = -1 is not stored, bwt.c:185-196)
            const uint64_t minus1 = ~0ull;
            if ((e = hipMemcpyAsync(c->d_sa, &minus1, 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return bail(DG_ERR_HIP, "upload sa[0]", e);
        }
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(DG_ERR_HIP, "index array fills", e);
    c->ix.bwt = (const uint4 *)c->d_bwt; c->ix.sa = (const uint64_t *)c->d_sa; c->ix.pac = (const uint8_t *)c->d_pac;
    c->ix.loc_key = (const int64_t *)c->d_lockey; c->ix.loc_chr = (const int32_t *)c->d_locchr; c->ix.chr_off = (const int64_t *)c->d_chroff;
    c->ix.primary = m.primary; for (int i = 0; i < 5; i++) c->ix.L2[i] = m.L2[i]; c->ix.seq_len = m.seq_len;
    c->ix.l_pac = m.l_pac; c->ix.n_chr = m.n_chr; c->ix.sa_intv = m.sa_intv;
    c->ix.ktab = nullptr; c->ix.ktab_k = 0; c->ix.sa_dense = nullptr; c->ix.sa_dense_intv = 0; c->ix.sa_dense_shift = 0;
    IndexShared *sh = c->shared_ix;
    sh->ix = c->ix; c->ix_gen = sh->gen.load();
    const double t_alloc = now_s();
    sh->aids_state.store(1);
    sh->aids = std::thread(aids_thread, sh, m.sa_intv, m.n_sa, (flags & DG_INIT_ASYNC_AIDS) != 0, m.expected_reads);   // (sync: allocates the aids while the index bytes travel)

    const size_t chunk = (size_t)32 << 20;
    std::vector<UpChunk> chunks;
    add_chunks(chunks, bwt, bwt_off, (uint8_t *)c->d_bwt, bwt_bytes, chunk, true, n_blocks);
    add_chunks(chunks, sa, sa_off, (uint8_t *)c->d_sa + 8 * sa_first, (size_t)sa_count * 8, chunk, false, 0);
    add_chunks(chunks, pac, 0, (uint8_t *)c->d_pac, pac_copy, chunk, false, 0);
    int n_threads = 6;
    if (const char *v = getenv("DG_INIT_THREADS")) n_threads = std::max(1, std::min(32, atoi(v)));
    n_threads = (int)std::min<size_t>((size_t)n_threads, std::max<size_t>(1, chunks.size()));
    std::string what;
    e = upload_chunks(device, chunks, chunk, n_threads, what);
    { std::lock_guard<std::mutex> lk(sh->mu); sh->upload_done = true; sh->upload_ok = e == hipSuccess; }
    sh->cv.notify_all();
    if (e != hipSuccess) return bail(what.find("shorter") != std::string::npos ? DG_ERR_ARG : DG_ERR_HIP, what.c_str(), what.find("shorter") != std::string::npos ? hipSuccess : e);
    const double t_up = now_s();
    {
        std::lock_guard<std::mutex> lk(sh->mu);
        char b[320];
        snprintf(b, sizeof b, "context %.3f s, index arrays (hipMalloc + fill) %.3f s, %s -> HBM + Occ re-layout %.3f s (%.2f GB, %d reader threads)",
                 t_ctx - t_start, t_alloc - t_ctx, bwt.mem ? "host arrays" : "index files", t_up - t_alloc, (bwt_bytes + sa_count * 8 + pac_copy) / 1e9, n_threads);
        sh->report = std::string(b) + sh->report;
    }
    if (!(flags & DG_INIT_ASYNC_AIDS)) {
        const int rc = dg_index_wait(c);
        if (rc != DG_OK) { char msg[256]; { std::lock_guard<std::mutex> lk(sh->mu); snprintf(msg, sizeof msg, "%s", sh->aids_msg); } snprintf(g_init_error, sizeof g_init_error, "%s", msg); dg_destroy(c); if (status) *status = rc; return nullptr; }
    }
    if (getenv("DG_INIT_TIMING")) fprintf(stderr, "[libdartgpu init] %s%s (total %.3f s)\n", dg_init_report(c), (flags & DG_INIT_ASYNC_AIDS) ? "; aids: building in the background" : "", now_s() - t_start);
    if (status) *status = DG_OK;
    return c;
}

extern "C" int dg_index_wait(dg_ctx *c)
{
    if (!c || !c->shared_ix) return DG_ERR_ARG;
    IndexShared *sh = c->shared_ix;
    {
        std::unique_lock<std::mutex> lk(sh->mu);
        sh->cv.wait(lk, [&]() { const int s = sh->aids_state.load(); return s != 1; });
    }
    refresh_index(c);
    if (sh->aids_state.load() < 0) { std::lock_guard<std::mutex> lk(sh->mu); snprintf(c->err, 512, "look-up aids not built: %s", sh->aids_msg); return DG_ERR_HIP; }
    return DG_OK;
}

extern "C" const char *dg_init_report(const dg_ctx *c)
{
    if (!c || !c->shared_ix) return "";
    IndexShared *sh = c->shared_ix;
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->report_out = sh->report;           // (a copy the caller may keep reading while the aids thread appends)
    return sh->report_out.c_str();
}

extern "C" dg_ctx *dg_init(const dg_index_view *v, const dg_params *p, int device, int *status)
{
    if (!v || !p || !v->bwt || !v->sa || !v->pac || v->n_chr <= 0 || v->n_sa == 0) {
        snprintf(g_init_error, sizeof g_init_error, "dg_init arguments: invalid");
        if (status) *status = DG_ERR_ARG;
        return nullptr;
    }
    IndexMeta m;
    m.bwt_words = v->bwt_words; m.primary = v->primary; for (int i = 0; i < 5; i++) m.L2[i] = v->L2[i]; m.seq_len = v->seq_len; m.n_sa = v->n_sa; m.sa_intv = v->sa_intv;
    m.l_pac = v->l_pac; m.n_chr = v->n_chr; m.chr_off = v->chr_off; m.chr_len = v->chr_len; m.expected_reads = 0;
    UpSrc b, s, q;
    b.mem = (const uint8_t *)v->bwt; s.mem = (const uint8_t *)v->sa; q.mem = v->pac;
    return init_index(m, b, 0, s, 0, 0, v->n_sa, q, (size_t)(v->l_pac / 4 + 1), p, device, 0, status);
}

extern "C" dg_ctx *dg_init_files(const dg_index_files *f, const dg_params *p, int device, int flags, int *status)
{
    auto bad = [&](int code, const char *fmt, const char *arg) -> dg_ctx * {
        snprintf(g_init_error, sizeof g_init_error, fmt, arg);
        if (status) *status = code;
        return nullptr;
    };
    if (!f || !p || !f->bwt_path || !f->sa_path || !f->pac_path || f->n_chr <= 0 || !f->chr_off || !f->chr_len || f->l_pac <= 0) return bad(DG_ERR_ARG, "dg_init_files arguments: %s", "invalid");
    struct Fd { int fd = -1; ~Fd() { if (fd >= 0) close(fd); } } fb, fs, fp;
    struct stat st;
    IndexMeta m;
    // .bwt: primary, L2[1..4], then the Occ-interleaved words (bwt_index.cpp:102-121)
    if ((fb.fd = open(f->bwt_path, O_RDONLY)) < 0 || fstat(fb.fd, &st) != 0 || st.st_size < 40) return bad(DG_ERR_ARG, "cannot read %s", f->bwt_path);
    uint64_t hdr[7];
    if (pread(fb.fd, hdr, 40, 0) != 40) return bad(DG_ERR_ARG, "cannot read %s", f->bwt_path);
    m.primary = hdr[0]; m.L2[0] = 0; for (int i = 1; i < 5; i++) m.L2[i] = hdr[i]; m.seq_len = m.L2[4];
    m.bwt_words = ((uint64_t)st.st_size - 40) / 4;
    // .sa: primary, L2[1..4], sa_intv, seq_len, then sa[1 .. n_sa) (bwt_index.cpp:15-35)
    if ((fs.fd = open(f->sa_path, O_RDONLY)) < 0 || fstat(fs.fd, &st) != 0 || st.st_size < 56) return bad(DG_ERR_ARG, "cannot read %s", f->sa_path);
    if (pread(fs.fd, hdr, 56, 0) != 56) return bad(DG_ERR_ARG, "cannot read %s", f->sa_path);
    m.sa_intv = (int)hdr[5];
    if (m.sa_intv <= 0 || m.seq_len == 0) return bad(DG_ERR_ARG, "%s: bad header", f->sa_path);
    m.n_sa = (m.seq_len + (uint64_t)m.sa_intv) / (uint64_t)m.sa_intv;
    const uint64_t sa_count = std::min<uint64_t>(m.n_sa - 1, ((uint64_t)st.st_size - 56) / 8);
    if ((fp.fd = open(f->pac_path, O_RDONLY)) < 0 || fstat(fp.fd, &st) != 0) return bad(DG_ERR_ARG, "cannot read %s", f->pac_path);
    const size_t pac_copy = std::min<size_t>((size_t)st.st_size, (size_t)(f->l_pac / 4 + 1));
    m.l_pac = f->l_pac; m.n_chr = f->n_chr; m.chr_off = f->chr_off; m.chr_len = f->chr_len; m.expected_reads = f->expected_reads;
    UpSrc b, s, q;
    b.fd = fb.fd; s.fd = fs.fd; q.fd = fp.fd;
    return init_index(m, b, 40, s, 56, 1, sa_count, q, pac_copy, p, device, flags, status);
}

// A second context on the same device that shares the parent's index (no copy): its own streams, batch buffers
// and counters, so that two batches can be in flight at once (one host thread per context).  The parent must
// outlive its clones.
// A context drives two HIP streams and a host keeps several contexts in flight (dg_clone); the runtime's default of 4 hardware
// queues makes those streams wait for each other.  The variable is read when the HIP runtime initialises, so this only helps a
// host that loads the library before its first HIP call; others export GPU_MAX_HW_QUEUES themselves.
__attribute__((constructor)) static void dg_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

extern "C" dg_ctx *dg_clone(dg_ctx *parent, int *status)
{
    if (status) *status = DG_ERR_ARG;
    if (!parent) return nullptr;
    hipError_t e;
    if ((e = hipSetDevice(parent->device)) != hipSuccess) { if (status) *status = DG_ERR_HIP; return nullptr; }
    dg_ctx *c = new dg_ctx();
    c->device = parent->device; c->owns_index = false; c->n_cu = parent->n_cu;
    c->shared_ix = parent->shared_ix;
    c->ix = parent->ix; c->ix_gen = parent->ix_gen; c->pr = parent->pr;
    refresh_index(c);
    c->shared_caps = parent->shared_caps;
    if ((e = make_ctx_objects(c)) != hipSuccess) {
        snprintf(g_init_error, sizeof g_init_error, "dg_clone: %s", hipGetErrorString(e)); dg_destroy(c); if (status) *status = DG_ERR_HIP; return nullptr;
    }
    if (status) *status = DG_OK;
    return c;
}

// ------------------------------------------------------------------------------------------
// batch input: ASCII reads (as the reference's ReadItem_t::seq) or packed reads (2 bit/base + a list of the non-ACGT positions)
// ------------------------------------------------------------------------------------------
static int enqueue_upload(dg_ctx *c, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq)
{
    if (!c || n_reads < 0 || (n_reads > 0 && (!seq_off || !rlen || !seq))) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    size_t bytes = 0; int mx = 0;
    for (int i = 0; i < n_reads; i++) {
        const size_t end = (size_t)seq_off[i] + rlen[i];
        bytes = end > bytes ? end : bytes;
        mx = rlen[i] > mx ? rlen[i] : mx;
    }
    if (mx > DG_MAX_RLEN) { snprintf(c->err, 512, "a read is longer than DG_MAX_RLEN (%d)", DG_MAX_RLEN); return DG_ERR_ARG; }
    c->n_reads = n_reads; c->max_rlen = mx; c->seq_bytes = bytes; c->enc_ready = false; c->enqueued = false;
    HIPCHK(c->seq.ensure(bytes + 64));     /* the kernels read up to 24 bytes at a read position in one go */
    HIPCHK(c->seq_off.ensure((size_t)n_reads + 1)); HIPCHK(c->rlen.ensure((size_t)n_reads + 1));
    if (n_reads) {
        HIPCHK(hipMemcpyAsync(c->seq.p, seq, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->seq_off.p, seq_off, (size_t)n_reads * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->rlen.p, rlen, (size_t)n_reads * 2, hipMemcpyHostToDevice, c->stream));
    }
    return DG_OK;
}

extern "C" int dg_batch_upload(dg_ctx *c, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq)
{
    const int rc = enqueue_upload(c, n_reads, seq_off, rlen, seq);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return DG_OK;
}

// packed reads -> the ASCII buffer the report stage reads (A/C/G/T, then 'N' at the listed positions) and the 2-bit + mask
// words the seeding stage reads (k_encode's format).  One thread = 16 bases = one input word.
__global__ void __launch_bounds__(256)
k_unpack(const uint32_t *__restrict__ words, uint32_t n_words, int W2, double inv_w2, int rlen_all, const uint16_t *rlen_in, unsigned char *__restrict__ seq,
         uint32_t *__restrict__ seq_off, uint16_t *rlen_out, uint32_t *__restrict__ enc, unsigned int *bad)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_words) return;
    if (t == 0) *bad = 0u;                                              // (k_unpack_n, the next launch, raises it)
    uint32_t r = (uint32_t)((double)t * inv_w2);                        // t / W2 without an integer division: the product is within one of it
    if (r * (uint32_t)W2 > t) r--; else if ((r + 1u) * (uint32_t)W2 <= t) r++;
    const int ww = (int)(t - r * (uint32_t)W2);
    const int len = rlen_in ? rlen_in[r] : rlen_all, left = len - 16 * ww;
    const uint32_t w = words[t];
    // (no ASCII copy of the batch: the seeding stage and the fused pair kernel read the words; k_prep rebuilds the characters of the
    //  units that take the general path -- a twentieth of a DNA batch)
    const uint32_t past = left >= 16 ? 0u : (left <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu >> (2 * left));          // mask 0b11 past the end, as k_encode
    enc[(size_t)r * 2 * W2 + ww] = w & ~past;
    enc[(size_t)r * 2 * W2 + W2 + ww] = past;
    if (ww == 0) { seq_off[r] = r * 16u * (uint32_t)W2; if (!rlen_in) rlen_out[r] = (uint16_t)len; }      // (given lengths are already in place)
}
__global__ void __launch_bounds__(256)
k_unpack_n(const uint32_t *__restrict__ nlist, uint32_t n_n, int W2, uint32_t n_bases, unsigned char *__restrict__ seq, uint32_t *__restrict__ enc, unsigned int *bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_n) return;
    const uint32_t flat = nlist[i], per = 16u * (uint32_t)W2, r = flat / per, pos = flat - r * per;
    if (flat >= n_bases) { atomicMax(bad, 1u); return; }       // a list entry outside the batch (a buggy packer) must not write outside it: dg_batch_run reports DG_ERR_ARG
    const uint32_t bit = 3u << (30 - 2 * (pos & 15u));
    atomicAnd(&enc[(size_t)r * 2 * W2 + (pos >> 4)], ~bit);
    atomicOr(&enc[(size_t)r * 2 * W2 + W2 + (pos >> 4)], bit);
}

// DG_PACKED_PAIR=0 (a measurement switch): the ASCII copy of the WHOLE packed batch, as rounds 2-3 made it, for k_pair<false>
__global__ void __launch_bounds__(256)
k_unpack_all(uint32_t n_words, int W2, const uint32_t *__restrict__ enc, unsigned char *__restrict__ seq)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_words) return;
    const uint32_t r = t / (uint32_t)W2, ww = t - r * (uint32_t)W2;
    const uint32_t w = enc[(size_t)r * 2 * W2 + ww], m = enc[(size_t)r * 2 * W2 + W2 + ww];
    uint32_t out[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t a = (w >> (24 - 8 * q)) & 0xFFu, b = (m >> (24 - 8 * q)) & 0xFFu;
        const uint32_t sel = ((a * 0x40100401u) >> 6) & 0x03030303u, nb = ((b * 0x40100401u) >> 6) & 0x03030303u;
        const uint32_t isn = ((nb | (nb >> 1)) & 0x01010101u) * 0xFFu;
        out[q] = (__builtin_amdgcn_perm(0u, 0x54474341u, sel) & ~isn) | (0x4E4E4E4Eu & isn);
    }
    *(uint4 *)(seq + (size_t)r * 16 * W2 + 16 * ww) = make_uint4(out[0], out[1], out[2], out[3]);
}

static int enqueue_upload_packed(dg_ctx *c, int n_reads, int rlen_all, const uint16_t *rlen, int words_per_read, const uint32_t *words, const uint32_t *nlist, size_t n_n)
{
    if (!c || n_reads < 0 || words_per_read < 1 || (n_reads > 0 && !words) || (n_n > 0 && !nlist)) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const int W2 = words_per_read;
    int mx = rlen_all;
    if (rlen) { mx = 0; for (int i = 0; i < n_reads; i++) mx = rlen[i] > mx ? rlen[i] : mx; }
    if (mx > DG_MAX_RLEN || mx > 16 * W2 || (size_t)n_reads * 16 * W2 > 0xFFFFFFF0ull) { snprintf(c->err, 512, "packed batch: read length %d does not fit %d words (or exceeds DG_MAX_RLEN / 2^32 bases)", mx, W2); return DG_ERR_ARG; }
    if (n_reads && (mx + 15) / 16 != W2) { snprintf(c->err, 512, "packed batch: words_per_read must be ceil(longest read / 16) = %d", (mx + 15) / 16); return DG_ERR_ARG; }
    const size_t nw = (size_t)n_reads * W2, bytes = nw * 16;
    c->n_reads = n_reads; c->max_rlen = mx; c->seq_bytes = bytes; c->enqueued = false;
    HIPCHK(c->seq.ensure(bytes + 64)); HIPCHK(c->seq_off.ensure((size_t)n_reads + 1)); HIPCHK(c->rlen.ensure((size_t)n_reads + 1));
    HIPCHK(c->enc.ensure(2 * nw + 16)); HIPCHK(c->packed_in.ensure(nw + 1)); HIPCHK(c->nlist_in.ensure(n_n + 1));
    c->enc_ready = true;
    if (n_reads == 0) return DG_OK;
    HIPCHK(hipMemcpyAsync(c->packed_in.p, words, nw * 4, hipMemcpyHostToDevice, c->stream));
    if (rlen) HIPCHK(hipMemcpyAsync(c->rlen.p, rlen, (size_t)n_reads * 2, hipMemcpyHostToDevice, c->stream));
    if (n_n) HIPCHK(hipMemcpyAsync(c->nlist_in.p, nlist, n_n * 4, hipMemcpyHostToDevice, c->stream));
    k_unpack<<<(unsigned)((nw + 255) / 256), 256, 0, c->stream>>>(c->packed_in.p, (uint32_t)nw, W2, 1.0 / (double)W2, rlen_all, rlen ? c->rlen.p : nullptr, c->seq.p, c->seq_off.p, c->rlen.p, c->enc.p, c->d_input_bad);
    if (n_n) {
        k_unpack_n<<<(unsigned)((n_n + 255) / 256), 256, 0, c->stream>>>(c->nlist_in.p, (uint32_t)n_n, W2, (uint32_t)(nw * 16), c->seq.p, c->enc.p, c->d_input_bad);
        HIPCHK(hipMemcpyAsync(&c->h_tail->input_bad, c->d_input_bad, 4, hipMemcpyDeviceToHost, c->stream));
    } else c->h_tail->input_bad = 0;
    if (!c->env_packed_pair) k_unpack_all<<<(unsigned)((nw + 255) / 256), 256, 0, c->stream>>>((uint32_t)nw, W2, c->enc.p, c->seq.p);
    HIPCHK(hipGetLastError());
    return DG_OK;
}

extern "C" int dg_batch_upload_packed(dg_ctx *c, int n_reads, int rlen_all, const uint16_t *rlen, int words_per_read, const uint32_t *words, const uint32_t *nlist, size_t n_n)
{
    const int rc = enqueue_upload_packed(c, n_reads, rlen_all, rlen, words_per_read, words, nlist, n_n);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return DG_OK;
}

extern "C" void *dg_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
extern "C" void dg_host_free(void *p) { if (p) (void)hipHostFree(p); }

static WSLayout make_ws_layout(int R)
{
    WSLayout L;
    auto al = [](uint32_t x) { return (x + 15u) & ~15u; };
    uint32_t o = 0;
    const uint32_t nmax = 2u * (uint32_t)R + 32u;                       // longest genome fragment of a segment pair
    L.max_rlen = (uint32_t)R;
    L.cig_off = o; L.cig_cap = 4u * (uint32_t)R + 64u; o = al(o + L.cig_cap * 4u);
    L.nwbits_off = o; L.nwbits_words = ((uint32_t)R + 1u) * ((nmax + 7u) / 8u); o = al(o + L.nwbits_words * 4u);
    L.rows_off = o; L.row_cap = (uint32_t)R + 8u; o = al(o + 2u * L.row_cap * 4u);          // strip boundary column: s and r per row
    L.str_off = o; L.str_cap = al(3u * (uint32_t)R + 64u); o = al(o + 6u * L.str_cap);
    L.kmer_off = o; L.kmer_cap = (uint32_t)R + 8u > 48u ? (uint32_t)R + 8u : 48u; o = al(o + L.kmer_cap * 8u);   // also holds 4 x (4 + 6) u64 of pair strings/results
    uint32_t rd = 64; while (rd < (uint32_t)R + 1u) rd <<= 1;
    L.ring_diag = rd; L.ring_words = ((uint32_t)R + 64u) / 64u; L.ring_off = o; o = al(o + rd * L.ring_words * 8u);
    L.stride = (o + 255u) & ~255u;
    return L;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function on a device, not to a context: it only ever grows here (a
// context with shorter reads used to lower what a sibling with longer reads had set, whose next launch then asked for more than allowed)
#define DG_MAX_DEVICES 64
static std::mutex g_lds_mu;
static size_t g_lds_set[2][DG_MAX_DEVICES];
static hipError_t raise_dynamic_lds(const void *fn, int which, int device, size_t lds)
{
    std::lock_guard<std::mutex> lk(g_lds_mu);
    size_t &cur = g_lds_set[which][device & (DG_MAX_DEVICES - 1)];
    if (lds <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) cur = lds;
    return e;
}

// k_encode + k_seed (reads staged in LDS when 256 lanes x W words fit comfortably)
static hipError_t launch_seed(dg_ctx *c, int n, int H, hipEvent_t after_encode = nullptr)
{
    const int W = 2 * ((c->max_rlen + 15) / 16 > 0 ? (c->max_rlen + 15) / 16 : 1);   // 2-bit words + N-mask words per read
    hipError_t e = c->enc.ensure((size_t)W * n + 16);
    if (e != hipSuccess) return e;
    int lg = 0;
    while ((1 << lg) < W / 2) lg++;
    if (!c->enc_ready) k_encode<<<(unsigned)((((size_t)n << lg) + 255) / 256), 256, 0, c->stream>>>(c->seq.p, c->seq_off.p, c->rlen.p, n, W, lg, c->enc.p);
    if (after_encode) { e = hipEventRecord(after_encode, c->stream); if (e != hipSuccess) return e; }
    // persistent one-wave workgroups pulling reads from a queue (TOP_SEED_NEXT); long walks go to TOP_SEED_HEAVY's list
    if ((e = c->seed_heavy.ensure((size_t)n + 16)) != hipSuccess) return e;
    unsigned blocks = (unsigned)c->n_cu * (unsigned)c->env_seed_waves;   // one wave per SIMD: the kernel is instruction-fetch bound, more waves add ~10 % alone but cost more than
                                                                         // that to the other batches in flight (measured 5.4 vs 6.0 ms per step with four batches)
    if ((size_t)blocks * 64 > (size_t)n) blocks = (unsigned)((n + 63) / 64);
    const bool use_qf = !c->env_seed_legacy && !c->env_seed_phases && W <= 62 && c->pr.max_dup <= 30000;      // (k_seed_qf's slot state counts occurrences in 20 bits: 31 hits x max_dup)
    const int bail_trips = c->env_bail_trips > 0 ? c->env_bail_trips : (use_qf ? 64 : 128), both_thr = c->env_both;
    unsigned int *tops = c->d_tops;
    // default: the queue kernel (dg_seedq.h).  DG_SEED_LEGACY=1 or reads too long for its LDS slots (> 496 bases): the lane-per-read kernel
    c->seed_qf_used = false;
    if (use_qf) {
        // the free-running queue kernel (dg_seedq.h, k_seed_qf): workgroups of `nw` waves around 2^lg read slots
        int lg = c->env_seed_slots_lg >= 6 && c->env_seed_slots_lg <= 11 ? c->env_seed_slots_lg : 9;
        int nw = c->env_seed_wg_waves >= 1 && c->env_seed_wg_waves <= 8 ? c->env_seed_wg_waves : 4;
        while (lg > 6 && sqf_lds_bytes(lg, W, nw) > (size_t)156 * 1024) lg--;
        if (c->env_seed_slots_lg == 0) while (lg > 6 && sqf_lds_bytes(lg, W, nw) > (size_t)80 * 1024) lg--;      // default: two workgroups per CU
        const size_t lds = sqf_lds_bytes(lg, W, nw);
        if ((e = raise_dynamic_lds((const void *)k_seed_qf, 0, c->device, lds)) != hipSuccess) return e;
        unsigned per_cu = (unsigned)(((size_t)160 * 1024) / lds);
        if (per_cu * (unsigned)nw > 16u) per_cu = 16u / (unsigned)nw;
        // Two workgroups per CU although three fit: the launch lasts as long as its slowest reads' chains of trips whatever the number of waves
        // (alone 0.98 ms with two or three), and the third workgroup's 50 KB of LDS is what another batch's k_pair (72 KB) needs to start on
        // the same CU: 840 against 806 M reads/s with twelve batches in flight (profiles/r03/d_variants_wgs_per_cu.txt).  DG_SEED_WGS overrides.
        if (per_cu > 2u) per_cu = 2u;
        if (per_cu < 1u) per_cu = 1u;
        if (c->env_seed_wgs > 0) per_cu = (unsigned)c->env_seed_wgs;
        unsigned wgs = (unsigned)c->n_cu * per_cu;
        const unsigned need = (unsigned)(((size_t)n + ((size_t)1 << lg) - 1) >> lg);
        if (wgs > need) wgs = need;
        c->seed_qf_used = true;
        k_seed_qf<<<wgs, nw * 64, lds, c->stream>>>(c->ix, c->pr, c->enc.p, c->rlen.p, n, W, H, lg, c->hits.p, c->nhits.p, c->nseeds.p, tops + TOP_SEED_NEXT,
                                                    c->seed_heavy.p, tops + TOP_SEED_HEAVY, c->d_ctr, bail_trips, c->env_seed_partial, c->env_seed_multi, c->d_err,
                                                    c->env_drain_bail > 0 ? c->env_drain_bail : bail_trips);
    } else
    if (!c->env_seed_legacy && W <= 62) {
        int lg = c->env_seed_slots_lg >= 6 && c->env_seed_slots_lg <= 10 ? c->env_seed_slots_lg : 9;
        while (lg > 6 && sq_lds_bytes(lg, W) > (size_t)80 * 1024) lg--;
        const size_t lds = sq_lds_bytes(lg, W);
        if ((e = raise_dynamic_lds((const void *)k_seed_q, 1, c->device, lds)) != hipSuccess) return e;
        unsigned per_cu = (unsigned)(((size_t)160 * 1024) / lds);
        if (per_cu > 4u) per_cu = 4u;
        if (c->env_seed_wgs > 0) per_cu = (unsigned)c->env_seed_wgs;
        unsigned wgs = (unsigned)c->n_cu * per_cu;
        const unsigned need = (unsigned)(((size_t)n + ((size_t)1 << lg) - 1) >> lg);
        if (wgs > need) wgs = need;
        k_seed_q<<<wgs, SQ_THREADS, lds, c->stream>>>(c->ix, c->pr, c->enc.p, c->rlen.p, n, W, H, lg, c->hits.p, c->nhits.p, c->nseeds.p, tops + TOP_SEED_NEXT,
                                                      c->seed_heavy.p, tops + TOP_SEED_HEAVY, c->d_ctr, bail_trips, c->d_err);
    } else
    if (W <= 78) k_seed<true><<<blocks, 64, ((size_t)2 * W * 64 + 64) * 4, c->stream>>>(c->ix, c->pr, c->enc.p, c->rlen.p, n, W, H, c->hits.p, c->nhits.p, c->nseeds.p, tops + TOP_SEED_NEXT, c->seed_heavy.p, tops + TOP_SEED_HEAVY, c->d_ctr, bail_trips, both_thr);
    else k_seed<false><<<blocks, 64, 0, c->stream>>>(c->ix, c->pr, c->enc.p, c->rlen.p, n, W, H, c->hits.p, c->nhits.p, c->nseeds.p, tops + TOP_SEED_NEXT, c->seed_heavy.p, tops + TOP_SEED_HEAVY, c->d_ctr, bail_trips, both_thr);
    // the backward walk of every listed read, lane = read (dg_fm.h): from which start on all searches fail without being made
    if ((e = c->seed_heavy_sfail.ensure((size_t)n + 16)) != hipSuccess) return e;
    k_seed_heavy_walk<<<(unsigned)c->n_cu * 8u, 64, 0, c->stream>>>(c->ix, c->pr, c->enc.p, c->rlen.p, W, c->seed_heavy.p, tops + TOP_SEED_HEAVY, c->seed_heavy_sfail.p, c->d_ctr);
    k_seed_heavy<<<(unsigned)c->n_cu * (unsigned)c->env_seedh_bpc, 64, (size_t)W * 4 + 16, c->stream>>>(c->ix, c->pr, c->enc.p, c->rlen.p, W, H, c->hits.p, c->nhits.p, c->nseeds.p, c->seed_heavy.p, tops + TOP_SEED_HEAVY, c->seed_heavy_sfail.p, c->d_ctr);
    return hipGetLastError();
}

// A look-back may wait ~2 s of wall clock for a predecessor (DG_SCAN_POLL_BUDGET, the tests' hook: a poll count on a batch's FIRST attempt instead).
// `first_tile`: where this scan's tiles start in the context's state / trace arrays.
static TileScan make_tile_scan(dg_ctx *c, size_t first_tile, unsigned int *ticket, int which /* 1 = seed offsets, 2 = k_pair, 4 = k_emit_slow: DG_SCAN_POLL_SCANS picks the hooked ones */)
{
    TileScan ts;
    ts.w = c->scan_state.p + SCAN_WORDS * first_tile; ts.ticket = ticket; ts.epoch = c->scan_epoch;
    ts.budget = (c->env_scan_budget > 0 && c->attempt_no == 0 && (c->env_scan_mask & which)) ? (uint32_t)c->env_scan_budget : 0u;
    ts.dbg = c->d_sizes->scan_dbg;
    ts.ticks = (unsigned long long)c->wall_khz * 2000ull;
    ts.trace = c->scan_trace.p ? c->scan_trace.p + SCAN_TRACE_WORDS * first_tile : nullptr;
    return ts;
}

#define TICK(name) do { if (c->n_t < N_TIMERS) { c->tname[c->n_t] = name; HIPCHK(hipEventRecord(c->ev[c->n_t + 1], c->stream)); c->n_t++; } } while (0)

// the kernels of the seeding + locate stage up to the located, unsorted seeds (shared by dg_batch_run and dg_probe_seeds)
static int enqueue_seeding(dg_ctx *c, int n, int H, bool timed, bool paired_units)
{
    HIPCHK(c->hits.ensure((size_t)n * H)); HIPCHK(c->nhits.ensure(n)); HIPCHK(c->nseeds.ensure(n)); HIPCHK(c->seed_off.ensure((size_t)n + 1));
    if (c->shared_caps) caps_adopt(c->cap_seeds, c->shared_caps->seeds);
    if (c->cap_seeds < (size_t)n * 3 + 1024) c->cap_seeds = (size_t)n * 3 + 1024;              // first guess: ~2.5 seeds per read; grows by itself
    HIPCHK(c->seeds.ensure(c->cap_seeds + 1)); HIPCHK(c->cands.ensure(c->cap_seeds + 1)); HIPCHK(c->tile_read.ensure(c->cap_seeds / 64 + 16));
    if (timed) c->tname[c->n_t] = "k_encode";
    HIPCHK(launch_seed(c, n, H, timed ? c->ev[c->n_t + 1] : nullptr));
    if (timed) { c->n_t++; TICK("k_seed"); }
    {
        const int paired = (paired_units && (n % 2 == 0)) ? 1 : 0;
        HIPCHK(c->heavy.ensure((size_t)n + 16));
        const TileScan ts_seed = make_tile_scan(c, 0, c->d_tops + TOP_TICKET_SEED, 1);
        k_seed_offsets<<<(unsigned)((n + 256 * SO_PER - 1) / (256 * SO_PER)), 256, 0, c->stream>>>(n, paired, c->nseeds.p, c->seed_off.p, c->tile_read.p, (uint32_t)c->tile_read.cap,
                                                                                                 c->heavy.p, c->d_tops + TOP_HEAVY_UNITS, &c->d_sizes->total_seeds, (uint32_t)c->cap_seeds, ts_seed, c->d_err);
        HIPCHK(hipGetLastError());
    }
    if (timed) TICK("seed_offsets");
    k_locate<<<(unsigned)((c->cap_seeds + 255) / 256), 256, 0, c->stream>>>(c->ix, n, H, c->hits.p, c->tile_read.p, c->seed_off.p, c->seeds.p, c->d_ctr, c->d_err);
    HIPCHK(hipGetLastError());
    if (timed) TICK("k_locate");
    return DG_OK;
}

// the small per-batch state -- work counters, bump tops and tickets, status word, sizes -- zeroed by ONE launch (round 2: five fills;
// the four blocks stay separate allocations: the status word every workgroup reads must not share lines with the tops every workgroup bumps)
__global__ void __launch_bounds__(256)
k_batch_begin(unsigned long long *ctr, unsigned int *tops, int *err, DSizes *sizes)
{
    for (int i = threadIdx.x; i < CTR_STRIPES * CTR_STRIDE; i += 256) ctr[i] = 0ull;
    if (threadIdx.x < N_TOPS) tops[threadIdx.x] = 0u;
    if (threadIdx.x < (int)(sizeof(DSizes) / 4)) ((uint32_t *)sizes)[threadIdx.x] = 0u;
    if (threadIdx.x == 0) *err = 0;
}

// the look-back state of the batch's three single-pass scans in c->scan_state: [seed offsets | k_pair | k_emit_slow]
static size_t scan_tiles_seed(int n_reads) { return (size_t)(n_reads + 256 * SO_PER - 1) / (256 * SO_PER) + 1; }
static size_t scan_tiles_pair(int n_units) { return (size_t)(n_units + PU_THREADS - 1) / PU_THREADS + 1; }

// the tail of a batch: sizes, status word, tops and the work counters (their 64 stripes summed -- or, for the two maxima, maximised -- here)
// written straight into the context's page-locked host block by ONE launch (round 2: four device-to-host copies) -- and, since round 5, left ZEROED for
// the next run of the context, which then needs no k_batch_begin in front (one launch per batch less; every thread clears what it has just read)
__global__ void __launch_bounds__(256)
k_batch_end(DSizes *__restrict__ sizes, int *__restrict__ err, unsigned int *__restrict__ tops, unsigned long long *__restrict__ ctr, dg_ctx::HostTail *out)
{
    const int t = threadIdx.x;
    if (t < (int)(sizeof(DSizes) / 4)) { out->sizes_words()[t] = ((const uint32_t *)sizes)[t]; ((uint32_t *)sizes)[t] = 0u; }
    if (t < N_TOPS) { out->tops[t] = tops[t]; tops[t] = 0u; }
    if (t == 0) { out->err = *err; *err = 0; }
    if (t < CTR_STRIDE) {
        const bool is_max = t == CTR_MAXTRIPS || t == CTR_WTRIPS_MAX;
        unsigned long long v = 0;
        for (int s_ = 0; s_ < CTR_STRIPES; s_++) { const unsigned long long x = ctr[s_ * CTR_STRIDE + t]; ctr[s_ * CTR_STRIDE + t] = 0ull; v = is_max ? (x > v ? x : v) : v + x; }
        out->ctr[t] = v;
    }
}

static int zero_batch_state(dg_ctx *c, int n_reads, int n_units)
{
    if (!c->state_zeroed) {                     // (a context's first run, or the run after one whose enqueue failed half-way: k_batch_end of every complete run leaves the state zeroed)
        k_batch_begin<<<1, 256, 0, c->stream>>>(c->d_ctr, c->d_tops, c->d_err, c->d_sizes);
        HIPCHK(hipGetLastError());
    }
    c->state_zeroed = false;
    // the look-back state of the two single-pass scans is NOT zeroed per batch: its words carry the run's epoch (dg_scan.h).
    // Zeroed once, when (re)allocated: epoch 0 is never used.
    const size_t tiles2 = (size_t)(2 * n_units + 255) / 256 + 1;
    const size_t words = SCAN_WORDS * (scan_tiles_seed(n_reads) + scan_tiles_pair(n_units) + tiles2);
    if (words > c->scan_state.cap) {
        HIPCHK(c->scan_state.ensure(words));
        HIPCHK(hipMemsetAsync(c->scan_state.p, 0, c->scan_state.cap * 8, c->stream));
        static_assert(SCAN_TRACE_WORDS == SCAN_WORDS, "the trace array is sized like the state array");
        HIPCHK(c->scan_trace.ensure(c->scan_state.cap));
        HIPCHK(hipMemsetAsync(c->scan_trace.p, 0, c->scan_trace.cap * 8, c->stream));
    }
    c->scan_epoch = (c->scan_epoch + 1u) & 0x3FFFFFFFu;
    if (c->scan_epoch == 0) c->scan_epoch = 1;
    return DG_OK;
}


// Enqueues the whole path of the uploaded batch on the context's stream and returns without waiting: no size is read back
// in between.  finish_run() waits, learns the sizes, and runs the batch again if a capacity was too small.
static int enqueue_run(dg_ctx *c)
{
    const int n = c->n_reads;
    const int paired = (c->pr.paired && (n % 2 == 0)) ? 1 : 0;      // Mapping.cpp:598
    const int n_units = paired ? n / 2 : n;
    refresh_index(c);
    c->n_t = 0;
    const int H = c->max_rlen / 16 + 1;
    const uint32_t nb = (uint32_t)((n + 255) / 256);
    { const int zr = zero_batch_state(c, n, n_units); if (zr) return zr; }
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    { const int rc = enqueue_seeding(c, n, H, true, c->pr.paired != 0); if (rc) return rc; }

    // capacities of the data-dependent buffers (sticky; see dg_ctx)
    if (c->shared_caps) { caps_adopt(c->cap_rep, c->shared_caps->rep); caps_adopt(c->cap_cig, c->shared_caps->cig); caps_adopt(c->cap_work, c->shared_caps->work); }
    if (c->cap_rep < (size_t)n + (size_t)n / 4 + 1024) c->cap_rep = (size_t)n + (size_t)n / 4 + 1024;
    if (c->cap_cig < 3 * c->cap_rep) c->cap_cig = 3 * c->cap_rep;
    if (c->cap_work < (size_t)n * 8 + 65536) c->cap_work = (size_t)n * 8 + 65536;
    const size_t cigcap = (size_t)n * 48 + c->cap_rep * (16 + CIG_SLOT) + 4096, sjcap = (size_t)n * 4 + 1024;
    HIPCHK(c->ncand.ensure(n)); HIPCHK(c->rep_off.ensure((size_t)n + 1)); HIPCHK(c->reads_out.ensure(n));
    HIPCHK(c->heavy.ensure((size_t)n_units + 16)); HIPCHK(c->slow_units.ensure((size_t)n_units + 16));
    HIPCHK(c->reports.ensure(c->cap_rep + 1)); HIPCHK(c->work.ensure(c->cap_work + 16)); HIPCHK(c->jobs.ensure(c->cap_seeds + 16));
    // k_reseed's work list: (window, chunk) items per ring size, and one record per chunk of the windows that several waves share (dg_reseed.h)
    const uint32_t rs_out_cap = (uint32_t)std::max<size_t>(16384, (size_t)n / 8), rs_list_cap = (uint32_t)std::min<size_t>(c->jobs.cap + rs_out_cap, 0xFFFFFFF0u);
    HIPCHK(c->job_outs.ensure(rs_out_cap)); HIPCHK(c->job_items.ensure(3 * (size_t)rs_list_cap + 16));
    const uint32_t rs_pool_cap = (uint32_t)std::max<size_t>(16384, (size_t)n / 32);           // blocks of 64 entries for the chunks that report more than their record holds
    HIPCHK(c->job_pool.ensure((size_t)rs_pool_cap * 64)); HIPCHK(c->job_pool_next.ensure(rs_pool_cap));
    HIPCHK(c->cigpool.ensure(cigcap)); HIPCHK(c->cigfinal.ensure(c->cap_cig + 16)); HIPCHK(c->sjpool.ensure(sjcap)); HIPCHK(c->sjfinal.ensure(sjcap));
    HIPCHK(c->items.ensure(c->cap_seeds + 16));                 // k_report's work list: one item per live candidate of the general path (candidates <= seeds)
    unsigned int *tops = c->d_tops;

    // units with more seeds than a lane of k_pair holds: a wave each, before k_pair (which needs their candidate counts)
    HIPCHK(c->chain_picks.ensure(c->cap_seeds + 16)); HIPCHK(c->chain_todo.ensure((size_t)n_units + 16));
    k_chain_heavy<<<c->n_cu * c->env_chain_bpc, 64, 0, c->stream>>>(c->ix, c->pr, n_units, paired, c->rlen.p, c->seed_off.p, c->seeds.p, c->cands.p, c->ncand.p,
                                                     c->heavy.p, tops + TOP_HEAVY_UNITS, c->chain_picks.p, c->chain_todo.p, c->d_ctr, c->d_err);
    // the sequential rest of the candidate rules of those units, lane = unit (dg_chain.h)
    k_chain_rules<<<c->n_cu * 4, 64, 0, c->stream>>>(paired, c->seed_off.p, c->cands.p, c->ncand.p, c->heavy.p, tops + TOP_HEAVY_UNITS, c->chain_picks.p, c->chain_todo.p, c->d_err);
    HIPCHK(hipGetLastError());
#ifdef DG_PROFILE_CLASSES
    {
        unsigned long long v[3], z = 0;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpyFromSymbol(&v[0], HIP_SYMBOL(g_chain_max), 8)); HIPCHK(hipMemcpyFromSymbol(&v[1], HIP_SYMBOL(g_chain_units), 8)); HIPCHK(hipMemcpyFromSymbol(&v[2], HIP_SYMBOL(g_chain_cycles), 8));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_max), &z, 8)); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_units), &z, 8)); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_cycles), &z, 8));
        fprintf(stderr, "[chain_heavy] units %llu  mean cycles %llu  max cycles %llu (seeds %llu/%llu cands %llu/%llu)\n", v[1], v[1] ? v[2] / v[1] : 0, v[0] >> 40,
                (v[0] >> 30) & 1023, (v[0] >> 20) & 1023, (v[0] >> 10) & 1023, v[0] & 1023);
    }
#endif
    TICK("k_chain_heavy");
    const size_t tiles_s = scan_tiles_seed(n), tiles = scan_tiles_pair(n_units);
    const TileScan ts_pair = make_tile_scan(c, tiles_s, tops + TOP_TICKET_PAIR, 2), ts_emit = make_tile_scan(c, tiles_s + tiles, tops + TOP_TICKET_EMIT, 4);
    const int try_fast = (c->env_no_fast || c->ix.n_chr > 0xFFFF) ? 0 : 1;
    // the compact record types are written by the kernels that write the full ones (unless the caller is known to take the full ones: dg_map_batch)
    CompactOut co{nullptr, nullptr, nullptr, nullptr};
    if (c->want_compact) {
        HIPCHK(c->reads_c.ensure((size_t)n + 1)); HIPCHK(c->reports_c.ensure(c->cap_rep + 1)); HIPCHK(c->cig_c.ensure(c->cap_cig + 16));
        co = CompactOut{c->reads_c.p, c->reports_c.p, c->cig_c.p, &c->d_sizes->pad[1]};
    }
    const int W2p = (c->max_rlen + 15) / 16 > 0 ? (c->max_rlen + 15) / 16 : 1;
    const bool packed_pair = c->enc_ready && c->env_packed_pair;
    if (packed_pair)        // a packed batch: the characters follow from its words; only the general path's units get an ASCII copy (below)
        k_pair<true><<<(unsigned)((n_units + PU_THREADS - 1) / PU_THREADS), PU_THREADS, 0, c->stream>>>(
            c->ix, c->pr, n_units, paired, try_fast, (c->want_full || !c->want_compact) ? 0 : 2, c->seq.p, c->seq_off.p, c->enc.p, W2p, c->rlen.p, c->seed_off.p, c->seeds.p, c->cands.p, c->ncand.p, c->rep_off.p,
            c->slow_units.p, c->reads_out.p, c->reports.p, c->cigfinal.p, (uint32_t)c->cap_rep, (uint32_t)c->cap_cig, ts_pair, c->d_sizes, tops + TOP_CIG, c->d_ctr, c->d_err, co);
    else
        k_pair<false><<<(unsigned)((n_units + PU_THREADS - 1) / PU_THREADS), PU_THREADS, 0, c->stream>>>(
            c->ix, c->pr, n_units, paired, try_fast, (c->want_full || !c->want_compact) ? 0 : 2, c->seq.p, c->seq_off.p, c->enc.p, W2p, c->rlen.p, c->seed_off.p, c->seeds.p, c->cands.p, c->ncand.p, c->rep_off.p,
            c->slow_units.p, c->reads_out.p, c->reports.p, c->cigfinal.p, (uint32_t)c->cap_rep, (uint32_t)c->cap_cig, ts_pair, c->d_sizes, tops + TOP_CIG, c->d_ctr, c->d_err, co);
    HIPCHK(hipGetLastError());
    TICK("k_pair");

    // ---- the general path, on the units k_pair listed ----
    const WSLayout L = make_ws_layout(c->max_rlen < 32 ? 32 : c->max_rlen);
    int blocks = c->n_cu * c->env_report_bpc;       // 8 one-wave workgroups per CU (2 per SIMD: the kernel needs ~220 VGPRs to stay out of scratch)
    if ((size_t)blocks * 64 > (size_t)n) blocks = (n + 63) / 64;
    {   // the lane workspace grows with the square of the longest read: keep it under ~12 GB by running fewer persistent waves
        const size_t budget = (size_t)12 << 30, per_block = (size_t)64 * L.stride;
        if ((size_t)blocks * per_block > budget) blocks = (int)(budget / per_block);
        if (blocks < 1) blocks = 1;
    }
    HIPCHK(c->ws.ensure((size_t)blocks * 64 * L.stride));
    unsigned slow_grid = (unsigned)c->n_cu * 4u;
    if ((size_t)slow_grid * 256 > (size_t)n) slow_grid = nb;
    k_prep<dg_report_out><<<slow_grid, 256, 0, c->stream>>>(c->pr, paired, c->slow_units.p, c->d_sizes, c->seed_off.p, c->seeds.p, c->cands.p, c->ncand.p, c->work.p, tops + TOP_WORK,
                                             (uint32_t)c->cap_work, c->jobs.p, tops + TOP_JOBS, (uint32_t)c->jobs.cap, c->d_err, c->rlen.p, c->rep_off.p, c->reports.p, tops + TOP_CLASS_HIST,
                                             packed_pair ? c->enc.p : nullptr, W2p, c->seq.p);      // (a packed batch: the listed reads get their ASCII copy here)
    HIPCHK(hipGetLastError());
    TICK("k_prep");
    // Round 5: the re-seeding kernels run on the context's main stream, before the report kernel, which then takes ALL candidates in one launch.  Rounds 1-4
    // gave a wave a whole window (up to 500 kb = ~1000 serial trips): the kernel was one long tail, so it ran on a second stream beside the report of the
    // candidates without jobs -- whose persistent waves held 113 of a CU's 160 KB of LDS and left k_reseed three waves per CU (cfg5: 3.8 s of wave time in
    // 5.4 ms).  With windows shared by several waves (dg_reseed.h) it is a balanced throughput kernel that wants the GPU for itself for a short time; a
    // context owns ONE stream (the sixteen hardware queues of a process hold twelve contexts, the caller's stream and -- on a node -- RCCL's).
    // DG_ONE_STREAM=0: the second stream as before (k_report in two launches: candidates without jobs beside k_reseed, the others behind it).
    if (!c->stream2) c->env_one_stream = 1;                       // (the context was made without a second stream)
    const hipStream_t s2 = c->env_one_stream ? c->stream : c->stream2;
    const uint32_t jobcap = (uint32_t)c->jobs.cap;
    const RsPool rs_pool{c->job_pool.p, c->job_pool_next.p, tops + TOP_RS_POOL, c->env_rs_pool > 0 ? std::min<uint32_t>((uint32_t)c->env_rs_pool, rs_pool_cap) : rs_pool_cap};
    const int rs_gap_max = c->max_rlen - 32, rs_need = rs_gap_max >= 8 ? (rs_gap_max - 8) / 64 + 1 : 1;        // (see below)
    // k_order: k_reseed's work list ((window, chunk) items per ring size) and k_report's (the candidates grouped by cost class)
    const RsOrder rs_order{c->jobs.p, tops + TOP_JOBS, jobcap, c->job_items.p, rs_list_cap, rs_out_cap, rs_need <= 1 ? 1 : (rs_need <= 2 ? 2 : 4), (uint32_t)c->env_rs_chunk, tops + TOP_RESEED_COUNT, tops + TOP_RS_OUT};
    k_order<<<slow_grid, 256, 0, c->stream>>>(paired, c->slow_units.p, c->d_sizes, c->seed_off.p, c->cands.p, c->ncand.p, tops + TOP_CLASS_HIST, tops + TOP_CLASS_FILL, c->items.p, tops + TOP_ORDER_INFO, rs_order, c->d_err);
    HIPCHK(hipGetLastError());
    TICK("order");
    HIPCHK(hipEventRecord(c->ev_prep, c->stream));
    if (!c->env_one_stream) HIPCHK(hipStreamWaitEvent(s2, c->ev_prep, 0));
    HIPCHK(hipEventRecord(c->ev_reseed0, s2));
    // One launch per ring size the batch can need: a read gap lies between two seeds of >= 16 bases (bwt_search.cpp:165), so it is at most rlen - 32 long and
    // needs (gap - 16) / 64 + 1 bitmap words per diagonal -- one for reads of up to 103 bases (round 4 launched all three sizes for every batch), two up to 167.
    // (one stream for the ring sizes: side by side on streams of their own they cost the step 8 % with eight batches in flight, profiles/r02)
    k_reseed<1><<<c->n_cu * 10 * c->env_reseed_pct / 100, 64, 0, s2>>>(c->ix, c->seq.p, c->seq_off.p, c->jobs.p, c->job_items.p, tops + TOP_RESEED_COUNT, tops + TOP_RESEED_COUNT + 3, tops + TOP_RESEED_TICKET, c->job_outs.p, rs_pool, c->env_rs_inline, c->d_ctr, c->d_err);
    if (rs_need > 1) k_reseed<2><<<c->n_cu * 6 * c->env_reseed_pct / 100, 64, 0, s2>>>(c->ix, c->seq.p, c->seq_off.p, c->jobs.p, c->job_items.p + rs_list_cap, tops + TOP_RESEED_COUNT + 1, tops + TOP_RESEED_COUNT + 3, tops + TOP_RESEED_TICKET + 1, c->job_outs.p, rs_pool, c->env_rs_inline, c->d_ctr, c->d_err);
    if (rs_need > 2) k_reseed<4><<<c->n_cu * 4 * c->env_reseed_pct / 100, 64, 0, s2>>>(c->ix, c->seq.p, c->seq_off.p, c->jobs.p, c->job_items.p + 2 * (size_t)rs_list_cap, tops + TOP_RESEED_COUNT + 2, tops + TOP_RESEED_COUNT + 3, tops + TOP_RESEED_TICKET + 2, c->job_outs.p, rs_pool, c->env_rs_inline, c->d_ctr, c->d_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_reseed1, s2));
    if (c->env_one_stream) TICK("k_reseed");
    const uint32_t *n_jobitems_p = tops + TOP_ORDER_INFO;             // items of class 0 (they wait for k_reseed)
    const uint32_t *n_items_p = tops + TOP_ORDER_INFO + 1;            // end of the list
    // (two streams: one wave slot per CU is left free so that k_reseed's waves are resident beside the persistent report waves)
    const int blocks_main = (!c->env_one_stream && blocks > c->n_cu * 4) ? blocks - c->n_cu : blocks;
    k_report<2><<<blocks_main, 64, 0, c->stream>>>(c->ix, c->pr, n, paired, c->seq.p, c->seq_off.p, c->rlen.p, c->seed_off.p, c->jobs.p, c->cands.p,
                                               c->rep_off.p, c->work.p, c->items.p, n_jobitems_p, n_items_p, c->env_one_stream ? 2 : 0, c->reports.p, c->cigpool.p,
                                               (uint32_t)cigcap, tops, c->ws.p, L, c->d_ctr, c->d_err);
    HIPCHK(hipGetLastError());
#ifdef DG_PROFILE_CLASSES
    {
        unsigned long long cyc[DG_COST_CLASSES + 1], cnt[DG_COST_CLASSES + 1], z[DG_COST_CLASSES + 1] = {0};
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_class_cycles), sizeof cyc)); HIPCHK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_class_chunks), sizeof cnt));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_class_cycles), z, sizeof z)); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_class_chunks), z, sizeof z));
        unsigned long long tot = 0; for (int k = 0; k <= DG_COST_CLASSES; k++) tot += cyc[k];
        unsigned long long ph[DG_NCLS][DG_NPHASE], zp[DG_NCLS][DG_NPHASE] = {{0}};
        HIPCHK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof ph)); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zp, sizeof zp));
        for (int k = 0; k < DG_COST_CLASSES; k++) if (cnt[k]) {
            fprintf(stderr, "[class %2d] chunks %7llu  cycles/chunk %8llu  share %5.1f %%  phases/chunk:", k, cnt[k], cyc[k] / cnt[k], 100.0 * cyc[k] / (tot ? tot : 1));
            for (int q = 0; q < DG_NPHASE; q++) fprintf(stderr, " %llu", ph[k][q] / cnt[k]);
            fprintf(stderr, "\n");
        }
    }
#endif
    TICK("k_report");
    if (!c->env_one_stream) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_reseed1, 0));
    // (the candidates that waited for k_reseed: a handful on DNA, a third of a spliced batch -- the full persistent grid; waves without work leave at once)
    k_report<2><<<blocks_main, 64, 0, c->stream>>>(c->ix, c->pr, n, paired, c->seq.p, c->seq_off.p, c->rlen.p, c->seed_off.p, c->jobs.p, c->cands.p,
                                               c->rep_off.p, c->work.p, c->items.p, n_jobitems_p, n_items_p, 1, c->reports.p, c->cigpool.p,
                                               (uint32_t)cigcap, tops, c->ws.p, L, c->d_ctr, c->d_err);
    HIPCHK(hipGetLastError());
    TICK("k_report_jobs");
    }
    k_finalize<<<slow_grid, 256, 0, c->stream>>>(c->ix, c->pr, paired, c->slow_units.p, c->d_sizes, c->seed_off.p, c->cands.p, c->ncand.p, c->rep_off.p, c->work.p,
                                                 c->reads_out.p, c->reports.p, c->sjpool.p, (uint32_t)sjcap, tops, c->d_err);
    k_emit_slow<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(paired, c->slow_units.p, c->d_sizes, c->reads_out.p, c->reports.p, c->cigpool.p, c->cigfinal.p,
                                                                    (uint32_t)c->cap_cig, c->sjpool.p, c->sjfinal.p, ts_emit, c->d_err, c->rlen.p, co);
    HIPCHK(hipGetLastError());
    TICK("k_finalize");
    // the tail: sizes, status, counters -> pinned host memory, one copy each
    k_batch_end<<<1, 256, 0, c->stream>>>(c->d_sizes, c->d_err, c->d_tops, c->d_ctr, c->h_tail);
    HIPCHK(hipGetLastError());
    c->state_zeroed = true;
    TICK("tail");
    c->enqueued = true;
    return DG_OK;
}

// waits for the enqueued batch; on a capacity overflow grows the buffer from what the device reported and runs the batch again
static int finish_run(dg_ctx *c, size_t used[3])
{
    c->runs_of_last_batch = 0;
    for (int attempt = 0; attempt < 6; attempt++) {
        c->runs_of_last_batch++;
        HIPCHK(wait_stream(c));
        if (c->owns_stream2 && c->stream2) HIPCHK(hipStreamSynchronize(c->stream2));      // (a shared one holds other contexts' kernels too; this context's are behind ev_reseed1, which its main stream waited for)
        if (c->enc_ready && c->h_tail->input_bad) { snprintf(c->err, 512, "packed batch: the N list holds a position outside the batch"); c->enqueued = false; return DG_ERR_ARG; }
        const DSizes &sz = c->h_tail->sizes;
        const int derr = c->h_tail->err;
        if (derr < DG_ABORT) break;
        auto grow = [](size_t need) { return need + need / 4 + 1024; };
        if (derr == DG_E_SEEDS) c->cap_seeds = grow(sz.total_seeds);
        else if (derr == DG_E_REPORTS) {        // (never shrinks: should the reported total ever be short, doubling still converges)
            c->cap_rep = std::max(grow(sz.total_rep), 2 * c->cap_rep);
            if (c->cap_cig < 3 * c->cap_rep) c->cap_cig = 3 * c->cap_rep;
        }
        else if (derr == DG_E_WORK) c->cap_work = grow(c->h_tail->tops[TOP_WORK]);
        else if (derr == DG_E_CIGFINAL) c->cap_cig = grow(sz.total_cig > 2 * c->cap_cig ? sz.total_cig : 2 * c->cap_cig);
        if (c->shared_caps) { caps_publish(c->shared_caps->seeds, c->cap_seeds); caps_publish(c->shared_caps->rep, c->cap_rep); caps_publish(c->shared_caps->work, c->cap_work); caps_publish(c->shared_caps->cig, c->cap_cig); }
        if (derr == DG_E_SEEDS || derr == DG_E_REPORTS || derr == DG_E_WORK || derr == DG_E_CIGFINAL) { /* grown above */ }
        else if ((derr == DG_E_SCAN || derr == DG_E_SEEDQ) && attempt < 5) {
            // a look-back that ran out of its time budget (dg_scan.h) or the seeding kernel's safety net: nothing to grow, the batch runs again (up to
            // five times: an event is rare -- three in 600 000 batches -- and independent of the batch) -- but never silently: what the poller saw and
            // what the stuck tile's workgroup last said about itself go to stderr and into the context's error text, the count into dg_last_counters [40]
            c->reruns_scan++;
            const unsigned long long *d = sz.scan_dbg;
            const double tick_ms = 1.0 / (double)c->wall_khz;
            const bool same_epoch = (d[8] >> 32) == d[5];
            snprintf(c->err, 512, "batch run again (device status %d): look-back of tile %llu gave up on tile %lld after %llu polls / %.1f ms (lane %llu, epoch %llu): words %016llx %016llx %016llx; "
                     "stuck tile's trace: epoch %llu%s HW_ID %08llx (wave %llu simd %llu cu %llu sh %llu se %llu) XCC %llu, ticket %.3f ms before the give-up, own totals %s, prefix %s",
                     derr, d[0] ? d[0] - 1 : 0ull, (long long)d[1], d[6], (double)(d[12] - d[13]) * tick_ms, d[7], d[5], d[2], d[3], d[4],
                     d[8] >> 32, same_epoch ? "" : " (NOT this run's: it never took its ticket here)", d[8] & 0xFFFFFFFFull, d[8] & 15, (d[8] >> 4) & 3, (d[8] >> 8) & 15, (d[8] >> 12) & 1, (d[8] >> 13) & 7, d[11] >> 56,
                     (double)(d[12] - d[9]) * tick_ms, d[10] ? "published" : "NOT published", (d[11] & 0x00FFFFFFFFFFFFFFull) ? "published" : "NOT published");
            if (!c->env_scan_budget) fprintf(stderr, "[libdartgpu] %s\n", c->err);
        }
        else { snprintf(c->err, 512, "device-side scan did not complete (status %d)", derr); return DG_ERR_INTERNAL; }
        if (attempt == 5) { snprintf(c->err, 512, "buffer capacities did not converge (status %d)", derr); return DG_ERR_INTERNAL; }
        if (derr != DG_E_SCAN && derr != DG_E_SEEDQ) c->reruns_capacity++;
        c->attempt_no = attempt + 1;
        const int rc = enqueue_run(c);
        if (rc) return rc;
    }
    c->enqueued = false;
    for (int i = 0; i < c->n_t; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]); c->tms[i] = ms; }
    if (!c->env_one_stream && c->n_t < N_TIMERS) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev_reseed0, c->ev_reseed1); c->tname[c->n_t] = "k_reseed(overlapped)"; c->tms[c->n_t] = ms; c->n_t++; }
    static_assert(CTR_N <= CTR_STRIDE, "the work counters fill one stripe");
    for (int k = 0; k < CTR_N; k++) c->counters[k] = c->h_tail->ctr[k];
    if (c->seed_qf_used) {
        // k_seed_qf keeps no per-lane tallies of its own work (they cost registers and moves in its loop): what it executed follows from the
        // slots its trips served -- one Occ step (one block; two when the interval straddles blocks, not counted) per slot of a step trip, one
        // prefix-table entry per slot of a begin trip, one SA entry and one text window per slot of a locate trip (k_seed_heavy adds its own)
        c->counters[CTR_STEPS_ACT] += c->counters[CTR_SQ_LANES + SQ_STEP]; c->counters[CTR_BLOCKS_ACT] += c->counters[CTR_SQ_LANES + SQ_STEP];
        c->counters[CTR_KTAB] += c->counters[CTR_SQ_LANES + SQ_BEGIN]; c->counters[CTR_DIRECT] += c->counters[CTR_SQ_LANES + SQ_LOC] + c->counters[CTR_SQ_LANES + SQ_CMP];
    }
    const DSizes &sz = c->h_tail->sizes;
    c->counters[CTR_SEEDS] = sz.total_seeds;
    const int derr = c->h_tail->err;
    if (derr) { snprintf(c->err, 512, "device pool exhausted (%s)", derr == DG_E_CIGAR ? "cigar" : (derr == DG_E_SJ ? "splice junction" : "reseed jobs")); return DG_ERR_INTERNAL; }
    c->used[0] = sz.total_rep; c->used[1] = sz.total_cig; c->used[2] = sz.total_sj;
    c->packed_valid = c->want_compact; c->full_valid = c->want_full || !c->want_compact;
    if (used) { used[0] = c->used[0]; used[1] = c->used[1]; used[2] = c->used[2]; }
    return DG_OK;
}

extern "C" int dg_batch_run(dg_ctx *c, size_t used[3])
{
    if (!c) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    c->used[0] = c->used[1] = c->used[2] = 0;
    memset(c->counters, 0, sizeof c->counters);
    if (used) used[0] = used[1] = used[2] = 0;
    c->n_t = 0;
    c->packed_valid = false; c->full_valid = false;
    if (c->n_reads == 0) return DG_OK;
    c->attempt_no = 0;
    const int rc = enqueue_run(c);
    if (rc) return rc;
    return finish_run(c, used);
}

static int enqueue_download(dg_ctx *c, dg_read_out *ro, dg_report_out *po, uint32_t *cig, dg_sj_out *so, const size_t caps[3])
{
    if (caps[0] < c->used[0] || caps[1] < c->used[1] || caps[2] < c->used[2]) {
        snprintf(c->err, 512, "output capacity too small: reports %zu of %zu, cigar ops %zu of %zu, junction tuples %zu of %zu", c->used[0], caps[0], c->used[1], caps[1], c->used[2], caps[2]);
        return DG_ERR_CAPACITY;
    }
    c->dl_on_copy_stream = false;
    if (c->env_copy_stream >= 1 && c->ev_dl && host_page_locked(ro)) {   // (as in dg_batch_download_compact)
        std::lock_guard<std::mutex> lk(g_copy_mu[c->device & 63]);
        hipStream_t cs = nullptr;
        HIPCHK(copy_stream(c->device, 0, &cs));
        if (c->n_reads && ro) HIPCHK(hipMemcpyAsync(ro, c->reads_out.p, (size_t)c->n_reads * sizeof(dg_read_out), hipMemcpyDeviceToHost, cs));
        if (c->used[0] && po) HIPCHK(hipMemcpyAsync(po, c->reports.p, c->used[0] * sizeof(dg_report_out), hipMemcpyDeviceToHost, cs));
        if (c->used[1] && cig) HIPCHK(hipMemcpyAsync(cig, c->cigfinal.p, c->used[1] * 4, hipMemcpyDeviceToHost, cs));
        if (c->used[2] && so) HIPCHK(hipMemcpyAsync(so, c->sjfinal.p, c->used[2] * sizeof(dg_sj_out), hipMemcpyDeviceToHost, cs));
        HIPCHK(hipEventRecord(c->env_blocking_sync ? c->ev_dl_block : c->ev_dl, cs));
        c->dl_on_copy_stream = true;
        return DG_OK;
    }
    if (c->n_reads && ro) HIPCHK(hipMemcpyAsync(ro, c->reads_out.p, (size_t)c->n_reads * sizeof(dg_read_out), hipMemcpyDeviceToHost, c->stream));
    if (c->used[0] && po) HIPCHK(hipMemcpyAsync(po, c->reports.p, c->used[0] * sizeof(dg_report_out), hipMemcpyDeviceToHost, c->stream));
    if (c->used[1] && cig) HIPCHK(hipMemcpyAsync(cig, c->cigfinal.p, c->used[1] * 4, hipMemcpyDeviceToHost, c->stream));
    if (c->used[2] && so) HIPCHK(hipMemcpyAsync(so, c->sjfinal.p, c->used[2] * sizeof(dg_sj_out), hipMemcpyDeviceToHost, c->stream));
    return DG_OK;
}
// waits for what enqueue_download enqueued
static hipError_t wait_download(dg_ctx *c)
{
    if (c->dl_on_copy_stream) { c->dl_on_copy_stream = false; return hipEventSynchronize(c->env_blocking_sync ? c->ev_dl_block : c->ev_dl); }
    return wait_stream(c);
}

extern "C" int dg_batch_download_compact(dg_ctx *c, dg_read_c *ro, dg_report_c *po, uint32_t *cig, dg_sj_out *so, const size_t caps[3], size_t *n_ops_out)
{
    if (!c || !caps) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (n_ops_out) *n_ops_out = 0;
    if (caps[0] < c->used[0] || caps[2] < c->used[2]) {
        snprintf(c->err, 512, "output capacity too small: reports %zu of %zu, junction tuples %zu of %zu", c->used[0], caps[0], c->used[2], caps[2]);
        return DG_ERR_CAPACITY;
    }
    const size_t n = (size_t)c->n_reads, nr = c->used[0];
    if (n == 0) return DG_OK;
    if (!c->packed_valid) {                       // (dg_batch_run and dg_map_batch_compact write the compact records beside the full ones; dg_map_batch* do not)
        snprintf(c->err, 512, "dg_batch_download_compact: the last batch was mapped through dg_map_batch / dg_map_batch_packed, which write the full records only");
        return DG_ERR_ARG;
    }
    if (c->h_tail->sizes.pad[1]) { snprintf(c->err, 512, "a record field does not fit the compact types: use dg_batch_download"); return DG_ERR_RANGE; }
    const size_t n_ops = (size_t)c->h_tail->sizes.pad[0];
    if (n_ops_out) *n_ops_out = n_ops;
    if (caps[1] < n_ops) { snprintf(c->err, 512, "output capacity too small: cigar ops %zu of %zu", n_ops, caps[1]); return DG_ERR_CAPACITY; }
    if (c->env_copy_stream >= 1 && c->ev_dl && host_page_locked(ro)) {
        {
            std::lock_guard<std::mutex> lk(g_copy_mu[c->device & 63]);        // (a context's copies and its event stay together in the shared stream)
            hipStream_t cs = nullptr;
            HIPCHK(copy_stream(c->device, 0, &cs));
            if (ro) HIPCHK(hipMemcpyAsync(ro, c->reads_c.p, n * sizeof(dg_read_c), hipMemcpyDeviceToHost, cs));
            if (nr && po) HIPCHK(hipMemcpyAsync(po, c->reports_c.p, nr * sizeof(dg_report_c), hipMemcpyDeviceToHost, cs));
            if (n_ops && cig) HIPCHK(hipMemcpyAsync(cig, c->cig_c.p, n_ops * 4, hipMemcpyDeviceToHost, cs));
            if (c->used[2] && so) HIPCHK(hipMemcpyAsync(so, c->sjfinal.p, c->used[2] * sizeof(dg_sj_out), hipMemcpyDeviceToHost, cs));
            HIPCHK(hipEventRecord(c->env_blocking_sync ? c->ev_dl_block : c->ev_dl, cs));
        }
        HIPCHK(hipEventSynchronize(c->env_blocking_sync ? c->ev_dl_block : c->ev_dl));
        return DG_OK;
    }
    if (ro) HIPCHK(hipMemcpyAsync(ro, c->reads_c.p, n * sizeof(dg_read_c), hipMemcpyDeviceToHost, c->stream));
    if (nr && po) HIPCHK(hipMemcpyAsync(po, c->reports_c.p, nr * sizeof(dg_report_c), hipMemcpyDeviceToHost, c->stream));
    if (n_ops && cig) HIPCHK(hipMemcpyAsync(cig, c->cig_c.p, n_ops * 4, hipMemcpyDeviceToHost, c->stream));
    if (c->used[2] && so) HIPCHK(hipMemcpyAsync(so, c->sjfinal.p, c->used[2] * sizeof(dg_sj_out), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(wait_stream(c));
    return DG_OK;
}

extern "C" int dg_map_batch_compact(dg_ctx *c, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq,
                                    int rlen_all, int words_per_read, const uint32_t *words, const uint32_t *nlist, size_t n_n,
                                    dg_read_c *ro, dg_report_c *po, uint32_t *cig, dg_sj_out *so, const size_t caps[3], size_t used[3])
{
    if (!c || !caps) return DG_ERR_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = words ? enqueue_upload_packed(c, n_reads, rlen_all, rlen, words_per_read, words, nlist, n_n) : enqueue_upload(c, n_reads, seq_off, rlen, seq);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    c->want_full = false;                                   // this caller takes the compact records: the units k_pair finishes get no full ones (dg_batch_download would map the batch again)
    rc = dg_batch_run(c, used);
    c->want_full = true;
    if (rc) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    size_t n_ops = 0;
    rc = dg_batch_download_compact(c, ro, po, cig, so, caps, &n_ops);
    const auto t3 = std::chrono::steady_clock::now();
    g_phase_ns[0] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    g_phase_ns[1] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
    g_phase_ns[2] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t3 - t2).count();
    g_phase_ns[3] += 1;
    if (getenv("DG_HOST_PHASES") && (g_phase_ns[3] % 100) == 0)
        fprintf(stderr, "[dg host phases] batches %llu: upload enqueue %.3f ms, run (enqueue + wait) %.3f ms, download %.3f ms per batch\n", (unsigned long long)g_phase_ns[3],
                g_phase_ns[0] / 1e6 / g_phase_ns[3], g_phase_ns[1] / 1e6 / g_phase_ns[3], g_phase_ns[2] / 1e6 / g_phase_ns[3]);
    if (used && (rc == DG_OK || rc == DG_ERR_CAPACITY)) used[1] = n_ops ? n_ops : (rc == DG_OK ? 0 : used[1]);
    return rc;
}

// The full record types of the units k_pair finished are not written when the caller asked for the compact records only (dg_map_batch_compact): a caller that
// wants them after all (the compact types could not hold a field: DG_ERR_RANGE) gets the batch mapped again, with them -- the uploaded batch is still in HBM.
static int ensure_full_records(dg_ctx *c)
{
    if (c->full_valid || c->n_reads == 0) return DG_OK;
    const bool keep = c->want_full;
    c->want_full = true;
    size_t used[3];
    const int rc = dg_batch_run(c, used);
    c->want_full = keep;
    return rc;
}

extern "C" int dg_batch_download(dg_ctx *c, dg_read_out *ro, dg_report_out *po, uint32_t *cig, dg_sj_out *so, const size_t caps[3])
{
    if (!c || !caps) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    { const int rf = ensure_full_records(c); if (rf) return rf; }
    const int rc = enqueue_download(c, ro, po, cig, so, caps);
    if (rc) return rc;
    HIPCHK(wait_download(c));
    return DG_OK;
}

// host buffers in, host records out: the copies and the kernels are enqueued back to back (two waits per batch: sizes, then
// records).  With page-locked caller buffers (dg_host_alloc) the copies are DMA transfers that overlap other contexts' kernels.
extern "C" int dg_map_batch(dg_ctx *c, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq,
                            dg_read_out *ro, dg_report_out *po, uint32_t *cig, dg_sj_out *so, const size_t caps[3], size_t used[3])
{
    if (!c || !caps) return DG_ERR_ARG;
    int rc = enqueue_upload(c, n_reads, seq_off, rlen, seq);
    if (rc) return rc;
    c->want_compact = false;                                // this caller takes the full records
    rc = dg_batch_run(c, used);
    c->want_compact = true;
    if (rc) return rc;
    return dg_batch_download(c, ro, po, cig, so, caps);
}

extern "C" int dg_map_batch_packed(dg_ctx *c, int n_reads, int rlen_all, const uint16_t *rlen, int words_per_read, const uint32_t *words,
                                   const uint32_t *nlist, size_t n_n, dg_read_out *ro, dg_report_out *po, uint32_t *cig, dg_sj_out *so,
                                   const size_t caps[3], size_t used[3])
{
    if (!c || !caps) return DG_ERR_ARG;
    int rc = enqueue_upload_packed(c, n_reads, rlen_all, rlen, words_per_read, words, nlist, n_n);
    if (rc) return rc;
    c->want_compact = false;
    rc = dg_batch_run(c, used);
    c->want_compact = true;
    if (rc) return rc;
    return dg_batch_download(c, ro, po, cig, so, caps);
}

extern "C" int dg_batch_device_ptrs(dg_ctx *c, void *ptrs[4])
{
    if (!c || !ptrs) return DG_ERR_ARG;
    { const int rf = ensure_full_records(c); if (rf) return rf; }
    ptrs[0] = c->reads_out.p; ptrs[1] = c->reports.p; ptrs[2] = c->cigfinal.p; ptrs[3] = c->sjfinal.p;
    return DG_OK;
}

extern "C" int dg_batch_device_ptrs_compact(dg_ctx *c, void *ptrs[2])
{
    if (!c || !ptrs) return DG_ERR_ARG;
    ptrs[0] = c->reads_c.p; ptrs[1] = c->reports_c.p;
    return DG_OK;
}

extern "C" int dg_batch_device_records_compact(dg_ctx *c, void *ptrs[4], size_t counts[4])
{
    if (!c || !ptrs || !counts) return DG_ERR_ARG;
    if (!c->packed_valid) { snprintf(c->err, 512, "dg_batch_device_records_compact: the last batch has no compact records (use dg_map_batch_compact / dg_batch_download_compact)"); return DG_ERR_ARG; }
    ptrs[0] = c->reads_c.p; ptrs[1] = c->reports_c.p; ptrs[2] = c->cig_c.p; ptrs[3] = c->sjfinal.p;
    counts[0] = (size_t)c->n_reads; counts[1] = c->used[0]; counts[2] = (size_t)c->h_tail->sizes.pad[0]; counts[3] = c->used[2];
    return DG_OK;
}

// ---- roofline calibration (not in the public header): random 64-byte block reads over the resident
// Occ array, the access pattern of k_seed / k_locate without any of their arithmetic.
// dependent = 1: the next block index depends on the loaded data (a chain per lane, like an FM walk);
// dependent = 0: indices are a pure function of (lane, iteration).
__global__ void __launch_bounds__(256)
k_rand_blocks(const uint4 *__restrict__ bwt, uint64_t n_blocks, int iters, int dependent, uint32_t *__restrict__ sink)
{
    uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
        const uint4 *p = bwt + ((x % n_blocks) << 2);
        const uint4 a = p[0], b = p[1], cc = p[2], d = p[3];
        const uint32_t v = a.x ^ b.y ^ cc.z ^ d.w;
        acc += v;
        if (dependent) x += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// quad-cooperative variant: the 4 lanes of a quad fetch the four 16-byte pieces of ONE block with one
// instruction (one 64-byte line per quad per load instruction instead of four instructions per lane)
__global__ void __launch_bounds__(256)
k_rand_blocks_quad(const uint4 *__restrict__ bwt, uint64_t n_blocks, int iters, int dependent, uint32_t *__restrict__ sink)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = (uint64_t)(tid >> 2) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
        const uint4 a = bwt[((x % n_blocks) << 2) + (tid & 3)];
        uint32_t v = a.x ^ a.y ^ a.z ^ a.w;
        v ^= __shfl_xor((int)v, 1, 64); v ^= __shfl_xor((int)v, 2, 64);      // every lane of the quad sees the whole block
        acc += v;
        if (dependent) x += v;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int dg_debug_random_blocks(dg_ctx *c, int iters, int dependent, int waves_per_cu, float *ms, unsigned long long *n_loads)
{
    if (!c) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const uint64_t n_blocks = (c->ix.seq_len + 127) / 128;
    const unsigned blocks = (unsigned)(c->n_cu * waves_per_cu / 4);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const int quad = dependent >> 1; dependent &= 1;
    if (quad) k_rand_blocks_quad<<<blocks, 256, 0, c->stream>>>(c->ix.bwt, n_blocks, 4, dependent, (uint32_t *)c->d_err);
    else k_rand_blocks<<<blocks, 256, 0, c->stream>>>(c->ix.bwt, n_blocks, 4, dependent, (uint32_t *)c->d_err);   // warm
    HIPCHK(hipEventRecord(e0, c->stream));
    if (quad) k_rand_blocks_quad<<<blocks, 256, 0, c->stream>>>(c->ix.bwt, n_blocks, iters, dependent, (uint32_t *)c->d_err);
    else k_rand_blocks<<<blocks, 256, 0, c->stream>>>(c->ix.bwt, n_blocks, iters, dependent, (uint32_t *)c->d_err);
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    *n_loads = (unsigned long long)blocks * (quad ? 64ull : 256ull) * (unsigned long long)iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return DG_OK;
}

// debugging aid (not in the public header): the re-seed job queue of the last run
extern "C" int dg_debug_jobs(dg_ctx *c, void *out, int cap)
{
    unsigned int n = c->h_tail ? c->h_tail->tops[TOP_JOBS] : 0u;       // (the device copy is zeroed by k_batch_end)
    if ((int)n > cap) n = (unsigned)cap;
    if (n && hipMemcpy(out, c->jobs.p, (size_t)n * sizeof(DJob), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}

extern "C" int dg_last_timings(dg_ctx *c, const char **names, float *ms, int cap)
{
    if (!c) return 0;
    int k = c->n_t < cap ? c->n_t : cap;
    for (int i = 0; i < k; i++) { names[i] = c->tname[i]; ms[i] = c->tms[i]; }
    return k;
}

extern "C" int dg_last_counters(dg_ctx *c, uint64_t *out, int cap)
{
    if (!c) return 0;
    int k = CTR_R4_N < cap ? CTR_R4_N : cap;
    for (int i = 0; i < k; i++) out[i] = c->counters[i];
    // [36] units that took the general path, [37] units chained by a wave each, [38] times the batch was enqueued, [39] / [40] re-runs of this context so far: capacity grown / scan not completed (> 1: a buffer grew)
    const uint64_t extra[5] = { c->h_tail ? c->h_tail->sizes.n_slow_units : 0u, c->h_tail ? c->h_tail->tops[TOP_HEAVY_UNITS] : 0u, (uint64_t)c->runs_of_last_batch,
                                c->reruns_capacity, c->reruns_scan };
    for (int i = 0; i < 5 && k < cap; i++) out[k++] = extra[i];
    for (int i = CTR_R4_N; i < CTR_N && k < cap; i++) out[k++] = c->counters[i];        // [41] windows scanned again whole, [42] (window, chunk) items of k_reseed
    return k;
}

// diagnostic: the per-tile trace of one of the last run's single-pass scans (dg_scan.h, SCAN_TRACE_WORDS u64 per tile: epoch << 32 | HW_ID, wall clock at the
// ticket, at the publication of the tile's own totals, XCC_ID << 56 | wall clock at the publication of its prefix), for tests/probes/kpair_wait.py:
// how long does a workgroup of k_pair work before its look-back, and how long does the look-back wait?  which: 0 = k_seed_offsets, 1 = k_pair, 2 = k_emit_slow
extern "C" int dg_debug_scan_trace(dg_ctx *c, int which, uint64_t *out, size_t cap_tiles, size_t *n_tiles, double *ticks_per_ms)
{
    if (!c || !out || !n_tiles || which < 0 || which > 2) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    const int n = c->n_reads, paired = (c->pr.paired && (n % 2 == 0)) ? 1 : 0, n_units = paired ? n / 2 : n;
    const size_t ts = scan_tiles_seed(n), tp = scan_tiles_pair(n_units), te = (size_t)(2 * n_units + 255) / 256 + 1;
    const size_t first = which == 0 ? 0 : (which == 1 ? ts : ts + tp), cnt = (which == 0 ? ts : (which == 1 ? tp : te)) - 1;
    *n_tiles = cnt;
    if (ticks_per_ms) *ticks_per_ms = (double)c->wall_khz;
    if (!c->scan_trace.p || cnt > cap_tiles) return cnt > cap_tiles ? DG_ERR_CAPACITY : DG_ERR_ARG;
    HIPCHK(hipMemcpy(out, c->scan_trace.p + SCAN_TRACE_WORDS * first, cnt * SCAN_TRACE_WORDS * 8, hipMemcpyDeviceToHost));
    return DG_OK;
}

// ---- the index builder's sorter (dg_sort.h): stable radix sort of n (key, value) pairs in device memory, ascending by the low
// key_bits bits of the key.  keys/vals hold the input and the result; *_tmp are scratch of the same size.  Runs on the NULL
// stream of `device` (so it is ordered with a caller that uses the default stream, e.g. torch) and returns when it is done.
extern "C" int dg_sort_pairs(int device, uint64_t *keys, int64_t *vals, uint64_t *keys_tmp, int64_t *vals_tmp, size_t n, int key_bits)
{
    dg_ctx *c = nullptr;                                   // (HIPCHK reports through the init error string)
    if (!keys || !vals || !keys_tmp || !vals_tmp || key_bits < 1 || key_bits > 64 || n >= 0xFFFFF000ull) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(device));
    if (n < 2) return DG_OK;
    const uint32_t tiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE), m = 16u * tiles;
    const uint32_t scan_tiles = (m + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t *hist = nullptr, *sums = nullptr;
    HIPCHK(hipMalloc((void **)&hist, ((size_t)m + 1) * 4));
    if (hipMalloc((void **)&sums, ((size_t)scan_tiles + 1) * 4) != hipSuccess) { (void)hipFree(hist); return DG_ERR_HIP; }
    uint64_t *ka = keys, *kb = keys_tmp; int64_t *va = vals, *vb = vals_tmp;
    hipError_t e = hipSuccess;
    for (int shift = 0; shift < key_bits && e == hipSuccess; shift += 4) {
        k_rs_hist<<<tiles, 256, 0, 0>>>(ka, (uint32_t)n, shift, tiles, hist);
        k_scan_tiles<<<scan_tiles, 256, 0, 0>>>(hist, hist, sums, m);
        k_scan_top<<<1, 256, 0, 0>>>(sums, scan_tiles, hist + m, nullptr, 0, nullptr, 0);
        k_scan_add<<<(m + 255) / 256, 256, 0, 0>>>(hist, sums, m);
        k_rs_scatter<<<tiles, 256, 0, 0>>>(ka, va, kb, vb, (uint32_t)n, shift, tiles, hist);
        e = hipGetLastError();
        std::swap(ka, kb); std::swap(va, vb);
    }
    if (e == hipSuccess && ka != keys) {
        e = hipMemcpyAsync(keys, ka, n * 8, hipMemcpyDeviceToDevice, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(vals, va, n * 8, hipMemcpyDeviceToDevice, 0);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    (void)hipFree(hist); (void)hipFree(sums);
    if (e != hipSuccess) return fail(nullptr, DG_ERR_HIP, "dg_sort_pairs", e);
    return DG_OK;
}

// ---- stage probes ----
extern "C" int dg_probe_seeds(dg_ctx *c, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq,
                              uint32_t *seed_off, int32_t *rpos, int32_t *slen, int64_t *gpos, size_t cap, size_t *used)
{
    int rc = dg_batch_upload(c, n_reads, seq_off, rlen, seq);
    if (rc) return rc;
    const int n = n_reads;
    if (used) *used = 0;
    if (n == 0) return DG_OK;
    const int H = c->max_rlen / 16 + 1;
    uint32_t total = 0;
    for (int attempt = 0; ; attempt++) {
        c->n_t = 0;
        if ((rc = zero_batch_state(c, n, n))) return rc;
        if ((rc = enqueue_seeding(c, n, H, false, false))) return rc;
        // the production sorters, every read on its own (unpaired): k_chain_heavy for the long lists, k_pair (candidate stage
        // only, sorted seeds written back for every read) for the rest
        HIPCHK(c->ncand.ensure(n)); HIPCHK(c->rep_off.ensure((size_t)n + 1)); HIPCHK(c->reads_out.ensure(n));
        HIPCHK(c->heavy.ensure((size_t)n + 16)); HIPCHK(c->slow_units.ensure((size_t)n + 16)); HIPCHK(c->reports.ensure(16)); HIPCHK(c->cigfinal.ensure(16));
        HIPCHK(c->chain_picks.ensure(c->cap_seeds + 16)); HIPCHK(c->chain_todo.ensure((size_t)n + 16));
        k_chain_heavy<<<c->n_cu * 4, 64, 0, c->stream>>>(c->ix, c->pr, n, 0, c->rlen.p, c->seed_off.p, c->seeds.p, c->cands.p, c->ncand.p, c->heavy.p, c->d_tops + TOP_HEAVY_UNITS, c->chain_picks.p, c->chain_todo.p, c->d_ctr, c->d_err);
        k_chain_rules<<<c->n_cu * 4, 64, 0, c->stream>>>(0, c->seed_off.p, c->cands.p, c->ncand.p, c->heavy.p, c->d_tops + TOP_HEAVY_UNITS, c->chain_picks.p, c->chain_todo.p, c->d_err);
        const TileScan ts = make_tile_scan(c, scan_tiles_seed(n), c->d_tops + TOP_TICKET_PAIR, 2);
        k_pair<false><<<(unsigned)((n + PU_THREADS - 1) / PU_THREADS), PU_THREADS, 0, c->stream>>>(
            c->ix, c->pr, n, 0, 0, 1, c->seq.p, c->seq_off.p, c->enc.p, 1, c->rlen.p, c->seed_off.p, c->seeds.p, c->cands.p, c->ncand.p, c->rep_off.p,
            c->slow_units.p, c->reads_out.p, c->reports.p, c->cigfinal.p, 0xFFFFFFFFu, 0xFFFFFFFFu, ts, c->d_sizes, c->d_tops + TOP_CIG, c->d_ctr, c->d_err, CompactOut{nullptr, nullptr, nullptr, nullptr});
        HIPCHK(hipGetLastError());
        k_batch_end<<<1, 256, 0, c->stream>>>(c->d_sizes, c->d_err, c->d_tops, c->d_ctr, c->h_tail);
        HIPCHK(hipGetLastError());
        c->state_zeroed = true;
        HIPCHK(hipStreamSynchronize(c->stream));
        total = c->h_tail->sizes.total_seeds;
        if (c->h_tail->err == DG_E_SEEDS && attempt < 3) { c->cap_seeds = (size_t)total + total / 4 + 1024; continue; }
        if (c->h_tail->err) { snprintf(c->err, 512, "dg_probe_seeds: device status %d", c->h_tail->err); return DG_ERR_INTERNAL; }
        break;
    }
    if (used) *used = total;
    if (total > cap) return DG_ERR_CAPACITY;
    std::vector<SKey> h(total);
    HIPCHK(hipMemcpyAsync(seed_off, c->seed_off.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (total) HIPCHK(hipMemcpyAsync(h.data(), c->seeds.p, (size_t)total * sizeof(SKey), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < total; i++) { rpos[i] = sk_rpos(h[i]); slen[i] = sk_rlen(h[i]); gpos[i] = sk_gpos(h[i]); }
    return DG_OK;
}

// ---- nw_alignment probes: every form of nw_alignment.cpp:18-82 that runs in production, one pair per lane ----
// mode 0: the serial strip form d_nw (string path of d_process_pair)
__global__ void __launch_bounds__(64)
k_probe_nw(int n, const uint32_t *a_off, const uint32_t *b_off, const char *a, const char *b, const uint32_t *out_off,
           uint32_t *out_len, char *out_a, char *out_b, unsigned char *ws, const WSLayout L, const DIndex ix, const DParams pr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LaneCtx cx;
    cx.ix = &ix; cx.pr = &pr; cx.L = &L; cx.ws = ws + (size_t)i * L.stride; cx.seq = nullptr; cx.rlen = 0; cx.lds = nullptr;
    cx.n_nw = cx.nw_cells = cx.n_reseed = cx.reseed_w = 0;
    out_len[i] = (uint32_t)d_nw(cx, a + a_off[i], (int)(a_off[i + 1] - a_off[i]), b + b_off[i], (int)(b_off[i + 1] - b_off[i]), out_a + out_off[i], out_b + out_off[i]);
}
// mode 1: the register-strip form d_pair_nw (pairs up to PM_MAX x PM_MAX: strings as bytes in registers, traceback bits in
// the lane's LDS slice, result = a column list); the gapped strings are rebuilt from the column list
__global__ void __launch_bounds__(64)
k_probe_nw_strips(int n, const uint32_t *a_off, const uint32_t *b_off, const char *a, const char *b, const uint32_t *out_off,
                  uint32_t *out_len, char *out_a, char *out_b, const DIndex ix, const DParams pr)
{
    __shared__ uint32_t lds_pm[64 * PM_LDS_WORDS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    LaneCtx cx;
    cx.ix = &ix; cx.pr = &pr; cx.L = nullptr; cx.ws = nullptr; cx.seq = nullptr; cx.rlen = 0; cx.lds = lds_pm + (threadIdx.x & 63) * PM_LDS_WORDS;
    cx.n_nw = cx.nw_cells = cx.n_reseed = cx.reseed_w = 0;
    const char *pa = a + a_off[i], *pb = b + b_off[i];
    const int m = (int)(a_off[i + 1] - a_off[i]), nn = (int)(b_off[i + 1] - b_off[i]);
    uint64_t w[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < m; k++) w[k >> 3] |= (uint64_t)(unsigned char)pa[k] << ((k & 7) << 3);
    for (int k = 0; k < nn; k++) w[3 + (k >> 3)] |= (uint64_t)(unsigned char)pb[k] << ((k & 7) << 3);
    PairStr ps; ps.A0 = w[0]; ps.A1 = w[1]; ps.A2 = w[2]; ps.B0 = w[3]; ps.B1 = w[4]; ps.B2 = w[5];
    ColList cl;
    d_pair_nw(cx, m, nn, ps, cl);
    char *oa = out_a + out_off[i], *ob = out_b + out_off[i];
    int x = 0, y = 0;
    for (int p_ = 0; p_ < cl.K; p_++) {
        const uint32_t ty = d_colcode(cl, p_) & 3u;
        oa[p_] = ty == 1 ? '-' : pa[x++];
        ob[p_] = ty == 2 ? '-' : pb[y++];
    }
    out_len[i] = (uint32_t)cl.K;
}
// modes 2 and 3: the wave-wide service d_nw_wave (8-lane groups up to 64 columns, the whole wave beyond; mode 3: the whole-wave
// form for every pair) + the owner's traceback d_tb_traceback.  The genome side comes from a pac (here: the b strings packed
// 2 bit/base), exactly as in k_report.
__global__ void __launch_bounds__(64)
k_probe_nw_wave(int n, int all_wide, const uint32_t *a_off, const uint32_t *b_off, const char *a, const uint32_t *out_off,
                uint32_t *out_len, char *out_a, char *out_b, unsigned char *ws, const WSLayout L, const DIndex ix, const DParams pr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const bool has = i < n;
    LaneCtx cx;
    cx.ix = &ix; cx.pr = &pr; cx.L = &L; cx.ws = ws + (size_t)(has ? i : 0) * L.stride; cx.seq = nullptr; cx.rlen = 0; cx.lds = nullptr;
    cx.n_nw = cx.nw_cells = cx.n_reseed = cx.reseed_w = 0;
    const int m = has ? (int)(a_off[i + 1] - a_off[i]) : 0, nn = has ? (int)(b_off[i + 1] - b_off[i]) : 0;
    const unsigned char *pa = (const unsigned char *)a + (has ? a_off[i] : 0);
    const int64_t gPos = has ? (int64_t)b_off[i] : 0;
    d_nw_wave(cx, has, pa, m, gPos, nn, lane, all_wide != 0);
    if (has) {
        char *g = ws_str(cx, 0);
        d_ref_fill(ix, gPos, nn, g);
        out_len[i] = (uint32_t)d_tb_traceback(cx, (const char *)pa, m, g, nn, out_a + out_off[i], out_b + out_off[i]);
    }
}

struct DevTmp {                       // frees its device buffers on every return path
    std::vector<void *> ptrs;
    hipError_t alloc(void **p, size_t bytes) { hipError_t e = hipMalloc(p, bytes ? bytes : 16); if (e == hipSuccess) ptrs.push_back(*p); return e; }
    ~DevTmp() { for (void *p : ptrs) (void)hipFree(p); }
};

extern "C" int dg_probe_nw_mode(dg_ctx *c, int mode, int n, const uint32_t *a_off, const uint32_t *b_off, const char *a, const char *b,
                                uint32_t *out_off, uint32_t *out_len, char *out_a, char *out_b, size_t cap)
{
    if (!c || n < 0 || mode < 0 || mode > 3) return DG_ERR_ARG;
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return DG_OK;
    size_t tot = 0; int mx = 32;
    for (int i = 0; i < n; i++) {
        const int m = (int)(a_off[i + 1] - a_off[i]), nn = (int)(b_off[i + 1] - b_off[i]);
        out_off[i] = (uint32_t)tot; tot += (size_t)m + nn;
        if (m > mx) mx = m;
        if ((nn + 1) / 2 > mx) mx = (nn + 1) / 2;      // workspace bit matrix holds 2R+32 columns
        if (mode == 1 && (m > PM_MAX || nn > PM_MAX || m < 1 || nn < 1)) { snprintf(c->err, 512, "dg_probe_nw_mode 1: pair %d is not within %d x %d", i, PM_MAX, PM_MAX); return DG_ERR_ARG; }
    }
    if (tot > cap) return DG_ERR_CAPACITY;
    const size_t la = a_off[n], lb = b_off[n];
    std::vector<uint8_t> pac;
    if (mode >= 2) {                                   // the genome side as a forward-strand pac (2 bit/base, first base on top)
        pac.assign(lb / 4 + 1 + 64, 0);
        for (size_t k = 0; k < lb; k++) {
            const uint8_t code = d_nt4((unsigned char)b[k]);
            if (code > 3) { snprintf(c->err, 512, "dg_probe_nw_mode %d: the genome side must be ACGT (RefSequence holds nothing else)", mode); return DG_ERR_ARG; }
            pac[k >> 2] |= (uint8_t)(code << ((~k & 3) << 1));
        }
    }
    const WSLayout L = make_ws_layout(mx);
    DevTmp tmp;
    unsigned char *d_ws = nullptr, *d_pac = nullptr; char *d_a = nullptr, *d_b = nullptr, *d_oa = nullptr, *d_ob = nullptr; uint32_t *d_off = nullptr;
    const size_t lanes = ((size_t)n + 63) / 64 * 64;
    if (mode != 1) HIPCHK(tmp.alloc((void **)&d_ws, lanes * L.stride));
    HIPCHK(tmp.alloc((void **)&d_a, la + 16)); HIPCHK(tmp.alloc((void **)&d_b, lb + 16));
    HIPCHK(tmp.alloc((void **)&d_oa, tot + 16)); HIPCHK(tmp.alloc((void **)&d_ob, tot + 16)); HIPCHK(tmp.alloc((void **)&d_off, (size_t)(4 * n + 4) * 4));
    uint32_t *d_aoff = d_off, *d_boff = d_off + (n + 1), *d_ooff = d_off + 2 * (n + 1), *d_olen = d_off + 3 * (n + 1);
    HIPCHK(hipMemcpy(d_a, a, la, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_b, b, lb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_aoff, a_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_boff, b_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_ooff, out_off, (size_t)n * 4, hipMemcpyHostToDevice));
    const unsigned nblk = (unsigned)((n + 63) / 64);
    if (mode == 0) k_probe_nw<<<nblk, 64, 0, c->stream>>>(n, d_aoff, d_boff, d_a, d_b, d_ooff, d_olen, d_oa, d_ob, d_ws, L, c->ix, c->pr);
    else if (mode == 1) k_probe_nw_strips<<<nblk, 64, 0, c->stream>>>(n, d_aoff, d_boff, d_a, d_b, d_ooff, d_olen, d_oa, d_ob, c->ix, c->pr);
    else {
        HIPCHK(tmp.alloc((void **)&d_pac, pac.size()));
        HIPCHK(hipMemcpy(d_pac, pac.data(), pac.size(), hipMemcpyHostToDevice));
        DIndex fx = c->ix;                              // a text that is just the b strings, forward strand only
        fx.pac = d_pac; fx.l_pac = (int64_t)lb;
        k_probe_nw_wave<<<nblk, 64, 0, c->stream>>>(n, mode == 3 ? 1 : 0, d_aoff, d_boff, d_a, d_ooff, d_olen, d_oa, d_ob, d_ws, L, fx, c->pr);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out_len, d_olen, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out_a, d_oa, tot, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(out_b, d_ob, tot, hipMemcpyDeviceToHost));
    return DG_OK;
}

extern "C" int dg_probe_nw(dg_ctx *c, int n, const uint32_t *a_off, const uint32_t *b_off, const char *a, const char *b,
                           uint32_t *out_off, uint32_t *out_len, char *out_a, char *out_b, size_t cap)
{
    return dg_probe_nw_mode(c, 0, n, a_off, b_off, a, b, out_off, out_len, out_a, out_b, cap);
}
