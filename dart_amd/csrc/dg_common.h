// dart_amd/csrc/dg_common.h -- device-side data layout shared by the kernels of libdartgpu.
// gfx950 only (wave = 64). Names follow the reference's domain (seeds, candidates, reports).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DG_WAVE 64

__host__ __device__ __forceinline__ uint64_t d_u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// One segment pair (SeedPair_t, structure.h:106-115).  PosDiff is not stored: every use in the
// reference happens while PosDiff == gPos - rPos still holds, so it is derived.
struct __attribute__((aligned(8))) DSeed {
    int64_t  gPos;
    int32_t  rPos;
    int32_t  rLen;
    int32_t  gLen;
    uint32_t flags;       // bit0 bSimple, bit1 bAcceptorSite
};
#define SEED_SIMPLE   1u
#define SEED_ACCEPTOR 2u

// An exact seed as it leaves the FM-index stage (bSimple, gLen == rLen), packed so that sorting the 64-bit values IS the
// (gPos, rPos) order of CompByGenomePos (AlignmentCandidates.cpp:21-25): gPos < 2^40 (texts up to 10^12 symbols),
// rPos, rLen < 4096.  k_locate writes these, the chain stage sorts them; DSeed (24 bytes) only exists in the working
// regions of the candidates that go through the general report path.
typedef uint64_t SKey;
__host__ __device__ __forceinline__ SKey sk_make(int64_t gPos, int rPos, int rLen) { return ((uint64_t)gPos << 24) | ((uint64_t)(uint32_t)rPos << 12) | (uint64_t)(uint32_t)rLen; }
__host__ __device__ __forceinline__ int64_t sk_gpos(SKey k) { return (int64_t)(k >> 24); }
__host__ __device__ __forceinline__ int sk_rpos(SKey k) { return (int)((k >> 12) & 0xFFFu); }
__host__ __device__ __forceinline__ int sk_rlen(SKey k) { return (int)(k & 0xFFFu); }
__host__ __device__ __forceinline__ int64_t sk_diag(SKey k) { return sk_gpos(k) - sk_rpos(k); }      // PosDiff

// One maximal exact match before SA lookup: the SA interval of BWT_Search (bwt_search.cpp:139-182)
struct __attribute__((aligned(16))) DHit {
    uint64_t x0;          // first row of the forward interval
    uint32_t freq;        // interval size (<= MaxDupNum)
    uint16_t rPos, len;
};

// AlignmentCandidate_t (structure.h:125-132); SeedVec = seeds[first .. first+count)
struct __attribute__((aligned(8))) DCand {
    int64_t PosDiff;
    int32_t first, count;
    int32_t Score, PairedIdx;
    int32_t SJtype;
    uint32_t work_off;    // offset of this candidate's private working seed region
    int32_t final_n;      // seeds left in the working region after the report stage
    int32_t n_a;          // seeds left after the clean-up phase (k_prep)
    uint32_t job_first;   // this candidate's re-seeding jobs = jobs[job_first .. job_first+job_count)
    int32_t job_count;
};

// One ReseedingWithSpecificRegion call (AlignmentCandidates.cpp:596-624), queued by k_prep, executed by k_reseed -- one wave when the
// genome window has at most `chunk` diagonals, else one wave per chunk of diagonals (dg_reseed.h) --, consumed by k_report.
struct __attribute__((aligned(16))) DJob {
    int64_t Lb;           // genome window [Lb, Lb+glen)
    int64_t gPos;         // result: seed genome position
    int32_t glen;
    int32_t rBegin, rl;   // read gap [rBegin, rBegin+rl)
    uint32_t read;        // read index (for the sequence)
    int32_t found;        // result: 1 = seed accepted
    int32_t rPos, len;    // result
    uint32_t n_chunks;    // k_order_jobs: waves that share this window (1: the whole window by one wave, nothing below is used)
    uint32_t out_first;   // first of its n_chunks RsChunkOut records
    uint32_t done;        // chunks finished so far (the wave that finishes the last one folds all of them)
    uint32_t pad[2];
};
static_assert(sizeof(DJob) == 64, "DJob is four 16-byte words");

// The index on the device.  bwt keeps the reference's Occ-interleaved layout (one 64-byte block
// per 128 rows = 4 x u64 cumulative counts + 8 x u32 of 2-bit symbols, bwtindex.c:53-75) at a
// 64-byte aligned base, so one Occ query is exactly one aligned 64-byte line.
struct DIndex {
    const uint4    *bwt;          // 4 x uint4 per block
    const uint64_t *sa;           // sampled SA, sa[0] = -1
    const uint64_t *sa_dense;     // SA of every sa_dense_intv-th row: (pos+1) in 40 bits | reference LF steps << 40; NULL = off
    const uint64_t *ktab;         // K-mer prefix table (3 x u64 per entry), see dg_fm.h; NULL = off
    const uint8_t  *pac;
    const int64_t  *loc_key;      // ChrLocMap keys, ascending (2*n_chr)
    const int32_t  *loc_chr;
    const int64_t  *chr_off;
    uint64_t primary, L2[5], seq_len;
    int64_t  l_pac;
    int32_t  n_chr, sa_intv;
    int32_t  ktab_k, sa_dense_intv;
    int32_t  sa_dense_shift;      // log2(sa_dense_intv): a 64-bit division by a run-time value is ~150 instructions per lane
};

struct DParams {
    int32_t max_gaps, max_dup, max_intron, min_intron, max_mismatch, multi_hit, all_sj, paired;
};

// Device-side status word of a batch (dg_ctx::d_err).  1..3: a bump pool ran out (results of this run are incomplete);
// >= DG_ABORT: a capacity estimate was too small -- every later kernel of the batch returns at once, the host grows the
// buffer from the sizes the device reported and runs the batch again (dg_batch_run never syncs mid-batch to learn a size).
#define DG_E_CIGAR 1
#define DG_E_SJ 2
#define DG_E_JOBS 3
#define DG_ABORT 10
#define DG_E_SEEDS 10
#define DG_E_REPORTS 11
#define DG_E_WORK 12
#define DG_E_CIGFINAL 13
#define DG_E_SCAN 14             // a look-back of a single-pass scan ran out of its poll budget (dg_scan.h): the host logs what it saw and runs the batch again
#define DG_E_SEEDQ 15            // the seeding kernel's safety net (dg_seedq.h): raised BEFORE k_pair is launched, which then returns at once like on DG_E_SEEDS
#define DG_COST_CLASSES 32       // k_report's work order: cost classes of the reads on the general path's list (k_prep, dg_reseed.h)
// sizes the device reports at the end of a batch (one small D2H copy)
struct DSizes { uint32_t total_seeds, total_rep, total_work, total_cig, total_sj, n_jobs, n_slow_units, n_heavy_units, cig_fast, pad[3];
                unsigned long long scan_dbg[16]; };     // scan_dbg: what a look-back that ran out of budget saw (dg_scan.h, SCAN_DBG_WORDS)

// work counters (dg_last_counters)
enum { CTR_STEPS = 0, CTR_BLOCKS, CTR_LF, CTR_SA, CTR_SEEDS, CTR_CANDS, CTR_NW, CTR_NWCELLS, CTR_RESEED, CTR_RESEEDW,
       CTR_STEPS_ACT, CTR_BLOCKS_ACT, CTR_KTAB, CTR_LF_ACT, CTR_DIRECT, CTR_MAXTRIPS, CTR_WTRIPS_MAX, CTR_WTRIPS_SUM,
       CTR_RESEED_TRIPS, CTR_RESEED_TICKS,
       CTR_SQ_TRIPS, CTR_SQ_LANES = CTR_SQ_TRIPS + 5, CTR_SQ_PHASES = CTR_SQ_LANES + 5,
       CTR_WT_SEEDQF, CTR_WT_SEEDH, CTR_WT_CHAIN, CTR_WT_PAIR, CTR_WT_REPORT, CTR_R4_N /* dg_last_counters lists these, then its five host-side values, then the ones below */,
       CTR_RESEED_WHOLE = CTR_R4_N /* windows shared by several waves that one wave had to scan again whole (the entry pool ran out) */, CTR_RESEED_ITEMS /* (window, chunk) items k_reseed served */, CTR_RESEED_POOLED /* chunks whose entries went through the pool */, CTR_N };             // CTR_WT_*: time the kernel's waves were resident, summed over its waves, in wall-clock ticks (10 ns): which resource do a dozen batches in flight fill?   // k_seed_q: wave-trips and slots served per queue (begin, step, compare, locate, refill)   // *_ACT: steps/blocks this implementation really executed

__host__ __device__ __forceinline__ uint8_t d_nt4(unsigned char c)   // nst_nt4_table, BWT_Index/bntseq.c:40: ACGT/acgt -> 0..3, '-' -> 5, else 4
{
    // branch-free (a switch compiles to a tree of divergent branches): A=0x41 C=0x43 G=0x47 T=0x54 after clearing the case bit
    const uint32_t u = c & 0xDFu;
    const uint32_t code = ((u >> 1) ^ (u >> 2)) & 3u;
    const bool acgt = (u == 0x41u) | (u == 0x43u) | (u == 0x47u) | (u == 0x54u);
    return (uint8_t)(acgt ? code : (c == '-' ? 5u : 4u));
}

// RefSequence[g] (bwt_index.cpp:193-212,252) computed from the forward 2-bit pac; 0 outside [0,2L)
__host__ __device__ __forceinline__ char d_refchar(const DIndex &ix, int64_t g)
{
    const int64_t L = ix.l_pac;
    if (g < 0 || g >= 2 * L) return 0;
    const bool rev = g >= L;
    const int64_t f = rev ? 2 * L - 1 - g : g;
    const uint32_t c = (ix.pac[f >> 2] >> ((~f & 3) << 1)) & 3u;
    return (char)((rev ? 0x41434754u : 0x54474341u) >> (8u * c));     // "TGCA"[c] : "ACGT"[c] without a table in memory
}

typedef uint2 __attribute__((aligned(1))) uint2_a1;
typedef uint32_t __attribute__((aligned(1))) uint32_a1;

// up to 8 reference characters RefSequence[g0 .. g0+8) packed low byte first; positions outside the text give 0
__host__ __device__ __forceinline__ uint64_t d_ref8(const DIndex &ix, int64_t g0)
{
    const int64_t L = ix.l_pac;
    uint64_t out = 0;
    if (g0 >= 0 && g0 + 8 <= L) {                              // forward strand: 8 bases = 16 bits out of 3 pac bytes
        const uint32_t w = __builtin_bswap32(*(const uint32_a1 *)(ix.pac + (g0 >> 2))) << ((g0 & 3) << 1);
#pragma unroll
        for (int k = 0; k < 8; k++) out |= (uint64_t)((0x54474341u >> (8u * ((w >> (30 - 2 * k)) & 3u))) & 0xFFu) << (8 * k);
        return out;
    }
    if (g0 >= L && g0 + 8 <= 2 * L) {                          // reverse strand: Ref[g0+k] = comp(fwd[f0-k])
        const int64_t f0 = 2 * L - 1 - g0, lo = f0 - 7;        // fwd[lo..f0]
        const uint32_t w = __builtin_bswap32(*(const uint32_a1 *)(ix.pac + (lo >> 2))) << ((lo & 3) << 1);   // fwd[lo+q] at bits 31-2q
#pragma unroll
        for (int k = 0; k < 8; k++) out |= (uint64_t)((0x41434754u >> (8u * ((w >> (16 + 2 * k)) & 3u))) & 0xFFu) << (8 * k);
        return out;
    }
    for (int k = 0; k < 8; k++) out |= (uint64_t)(unsigned char)d_refchar(ix, g0 + k) << (8 * k);
    return out;
}

// RefSequence[g0 .. g0+n) into dst, eight bases per pac fetch (a per-base d_refchar is a dependent load each, and on a
// GRCh38-sized pac those miss the caches)
__host__ __device__ __forceinline__ void d_ref_fill(const DIndex &ix, int64_t g0, int n, char *dst)
{
    for (int i = 0; i < n; i += 8) {
        const uint64_t w = d_ref8(ix, g0 + i);
        const int m = n - i < 8 ? n - i : 8;
        for (int k = 0; k < m; k++) dst[i + k] = (char)(w >> (8 * k));
    }
}

// ChrLocMap.lower_bound(g): index of the smallest key >= g
__host__ __device__ __forceinline__ int d_loc_lower_bound(const DIndex &ix, int64_t g)
{
    int lo = 0, hi = 2 * ix.n_chr;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (ix.loc_key[mid] < g) lo = mid + 1; else hi = mid; }
    return lo;
}
// The chromosome table (ChrLocMap keys, their chromosomes, the chromosomes' offsets) as three pointers: the index's arrays in memory, or a copy a
// workgroup keeps in LDS -- a look-up is seven dependent loads (a binary search over 2 n keys, then the chromosome and its offset), and k_pair
// makes one per seed-joining question and one per reported candidate: from memory they were half of a tile's processing time.
struct LocTab { const int64_t *key; const int32_t *chr; const int64_t *off; int n2; };
__host__ __device__ __forceinline__ LocTab d_loc_tab(const DIndex &ix) { return LocTab{ix.loc_key, ix.loc_chr, ix.chr_off, 2 * ix.n_chr}; }
__host__ __device__ __forceinline__ int d_loc_lower_bound(const LocTab &t, int64_t g)
{
    int lo = 0, hi = t.n2;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (t.key[mid] < g) lo = mid + 1; else hi = mid; }
    return lo;
}

__host__ __device__ __forceinline__ bool d_seed_less(const DSeed &a, const DSeed &b)   // CompByGenomePos, AlignmentCandidates.cpp:21-25
{
    return a.gPos == b.gPos ? a.rPos < b.rPos : a.gPos < b.gPos;
}

// The work counters live in CTR_STRIPES copies, 384 bytes apart, chosen by workgroup: atomics on ONE address
// serialise at ~6 ns each on this chip (three counters updated by each of k_locate's 56 k waves cost 1 ms,
// more than the kernel's work).  dg_batch_run sums (or, for the two maxima, maximises) the stripes.
#define CTR_STRIPES 64
#define CTR_STRIDE  48
__device__ __forceinline__ unsigned long long *d_ctr_stripe(unsigned long long *ctr) { return ctr + (blockIdx.x & (CTR_STRIPES - 1)) * CTR_STRIDE; }
// the time this wave was resident, added to a striped counter by its first lane (the last statement of a kernel)
__device__ __forceinline__ void d_wave_resident(unsigned long long *ctr, int which, unsigned long long t_begin)
{
    if ((threadIdx.x & 63) == 0) atomicAdd(d_ctr_stripe(ctr) + which, (unsigned long long)(wall_clock64() - t_begin));
}
// wave-level sum of a per-lane counter, one atomic per wave
__device__ __forceinline__ void d_wave_add(unsigned long long *dst, unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(d_ctr_stripe(dst), v);
}
