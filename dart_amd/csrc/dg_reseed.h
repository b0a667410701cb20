// dart_amd/csrc/dg_reseed.h -- candidate clean-up (k_prep) and wave-cooperative 8-mer re-seeding
// (k_reseed).
//
// k_prep replaces RemoveTandemRepeatSeeds / RemoveTranslocatedSeeds (AlignmentCandidates.cpp:
// 817-902) and the *enumeration* half of IdentifyMissingSeeds (:685-700): every
// ReseedingWithSpecificRegion call the reference would make becomes one DJob.
// k_reseed replaces ReseedingWithSpecificRegion (:596-624) + KmerAnalysis.cpp:25-166.
//
// Why a separate kernel: a re-seeding window is up to MaxIntronSize (5e5) bases but only
// 0.03-0.8 % of reads need one.  Inside the lane-per-read report kernel one such read
// stalls its whole wave for tens of milliseconds; as a job queue the windows are spread over the
// chip, one wave per window, 64 window positions per step.
//
// Algorithm (order-equivalent to the reference's sort/join/sort, see dg_report.h d_reseed for the
// serial form): the read-gap 8-mers sit sorted in LDS behind a 64 Kbit presence filter; every lane
// extracts the 8-mer of its window position straight from the 2-bit pac (one unaligned 24-bit
// fetch; reverse-strand windows are read from the forward pac and reverse-complemented in
// registers), and hits set bit rPos of diagonal (g - rPos) in an LDS ring of diagonals.  A diagonal
// is complete once the window front has moved past it by the gap length; complete diagonals are
// summarised 64 at a time (count, first, last rPos) and folded in increasing order into the
// running (s, max_len) state of GenerateLongestSimplePairsFromFragmentPair -- bit-identical,
// including the carry-over quirk of `s` (KmerAnalysis.cpp:147-161).
// Bound: streaming pac reads (2 bit/base) + LDS; no MFMA.
#pragma once
#include "dg_common.h"
#include "dg_report.h"

#define RS_MAX_RL   263          // longest read gap handled cooperatively (span <= 255 -> 4 bitmap words)
#define RS_RING     1024         // diagonals in the ring (>= span + 63 + RS_CHUNK live at any time): 32 KB of LDS, 4 waves per CU
#define RS_WORDS    4

// One thread per read of the general path's units (slow_units[], two reads per unit when paired).  Each live candidate gets its
// private working region (bump-allocated per wave: the regions are internal, only the records' layout is deterministic), its
// seeds unpacked into it, the clean-up passes, and its re-seeding jobs queued.
__global__ void __launch_bounds__(256)
k_prep(const DParams pr, int paired, const uint32_t *__restrict__ slow_units, const DSizes *__restrict__ sizes, const uint32_t *__restrict__ seed_off,
       const SKey *__restrict__ seeds, DCand *__restrict__ cands, const uint32_t *__restrict__ ncand, DSeed *__restrict__ work, unsigned int *worktop, uint32_t workcap,
       DJob *__restrict__ jobs, unsigned int *jobtop, uint32_t jobcap, int *err, const uint16_t *__restrict__ rlen)
{
    if (*err >= DG_ABORT) return;
    const unsigned int n_items = sizes->n_slow_units * (paired ? 2u : 1u);
    const int lane = threadIdx.x & 63;
    for (unsigned int base = blockIdx.x * blockDim.x; base < n_items; base += gridDim.x * blockDim.x) {   // uniform per workgroup
        const unsigned int it = base + threadIdx.x;
        const bool on = it < n_items;
        const int r = on ? (paired ? (int)(2u * slow_units[it >> 1] + (it & 1u)) : (int)slow_units[it]) : 0;
        DCand *cd = cands + seed_off[r];
        const int nc = on ? (int)ncand[r] : 0;
        uint32_t need = 0;
        for (int i = 0; i < nc; i++) if (cd[i].Score != 0) need += d_work_need(cd[i].count);
        // the wave's regions in one atomic
        uint32_t incl = need;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        const uint32_t wave_total = __shfl(incl, 63, 64);
        uint32_t wave_base = 0;
        if (lane == 0 && wave_total) wave_base = atomicAdd(worktop, wave_total);
        wave_base = __shfl(wave_base, 0, 64);
        if ((uint64_t)wave_base + wave_total > workcap) { if (lane == 0) atomicMax(err, DG_E_WORK); continue; }   // (worktop keeps the need)
        uint32_t wo = wave_base + incl - need;
        for (int i = 0; i < nc; i++) {
            DCand &c = cd[i];
            c.final_n = 0; c.n_a = 0; c.job_count = 0; c.job_first = 0;
            if (c.Score == 0) continue;
            c.work_off = wo; wo += d_work_need(c.count);
            DSeed *s = work + c.work_off;
            int n = c.count;
            for (int q = 0; q < n; q++) {
                const SKey k = seeds[c.first + q];
                DSeed x; x.gPos = sk_gpos(k); x.rPos = sk_rpos(k); x.rLen = x.gLen = sk_rlen(k); x.flags = SEED_SIMPLE;
                s[q] = x;
            }
            n = d_untangle_seeds(s, n, (int)rlen[r], (uint32_t *)(s + 6 * c.count + 4));       // tail of the working region as scratch
            c.n_a = n;
            int cnt = 0;                                     // IdentifyMissingSeeds :691-697, enumeration only
            int big = n > 0 && (s[0].rPos > PM_MAX || (int)rlen[r] - (s[n - 1].rPos + s[n - 1].rLen) > PM_MAX) ? 1 : 0, indel = 0;
            for (int k = 1; k < n; k++) {
                const int pd = (int)((s[k].gPos - s[k].rPos) - (s[k - 1].gPos - s[k - 1].rPos));
                const int rGaps = s[k].rPos - s[k - 1].rPos - s[k - 1].rLen;
                if (pd > pr.max_gaps && rGaps > 20) cnt++;
                if (rGaps > PM_MAX || rGaps + pd > PM_MAX) big = 1;
                if (pd != 0) indel = 2;
            }
            c.final_n = big | indel;                         // scheduling hints for k_cost only (k_report sets the real value): bit 0 = some
                                                             // segment pair is longer than PM_MAX (string path, maybe a wave-wide alignment),
                                                             // bit 1 = two seeds on different diagonals (an nw_alignment is certain)
            if (cnt == 0) continue;
            const unsigned int first = atomicAdd(jobtop, (unsigned int)cnt);
            if (first + (unsigned int)cnt > jobcap) { atomicMax(err, DG_E_JOBS); continue; }
            c.job_first = first; c.job_count = cnt;
            int w = 0;
            for (int k = 1; k < n; k++) {
                const int pd = (int)((s[k].gPos - s[k].rPos) - (s[k - 1].gPos - s[k - 1].rPos));
                const int rGaps = s[k].rPos - s[k - 1].rPos - s[k - 1].rLen;
                if (pd > pr.max_gaps && rGaps > 20) {
                    DJob j;
                    j.Lb = s[k - 1].gPos + s[k - 1].gLen;
                    j.glen = (int32_t)(s[k].gPos - j.Lb);
                    j.rBegin = s[k - 1].rPos + s[k - 1].rLen; j.rl = rGaps;
                    j.read = (uint32_t)r; j.found = rGaps > RS_MAX_RL ? -1 : 0;   // -1: left to the serial path in k_report
                    j.gPos = 0; j.rPos = 0; j.len = 0; j.pad = 0;
                    jobs[first + w++] = j;
                }
            }
        }
    }
}

// 8-mer id (CreateKmerID, KmerAnalysis.cpp:25-32) of text positions t..t+7, t in [0, 2L)
__device__ __forceinline__ uint32_t d_window_kmer(const DIndex &ix, int64_t t)
{
    const int64_t L = ix.l_pac;
    if (t >= 0 && t + 7 < L) {
        const uint8_t *p = ix.pac + (t >> 2);
        const uint32_t v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[2];
        return (v >> (8 - 2 * (int)(t & 3))) & 0xFFFFu;
    }
    if (t >= L && t + 7 < 2 * L) {                      // reverse half: complement of the mirrored forward 8-mer
        const int64_t u = 2 * L - 1 - t - 7;
        const uint8_t *p = ix.pac + (u >> 2);
        const uint32_t v = ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[2];
        uint32_t x = (v >> (8 - 2 * (int)(u & 3))) & 0xFFFFu;
        x = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
        x = ((x & 0x0F0Fu) << 4) | ((x >> 4) & 0x0F0Fu);
        x = ((x << 8) | (x >> 8)) & 0xFFFFu;
        return x ^ 0xFFFFu;
    }
    uint32_t wid = 0;                                    // straddles the strand boundary or the end
    for (int i = 0; i < 8; i++) wid = (wid << 2) + d_nt4((unsigned char)d_refchar(ix, t + i));
    return wid;
}

// running state of GenerateLongestSimplePairsFromFragmentPair's scan (KmerAnalysis.cpp:146-163)
struct RsFold { int s, max_len, best_r; int64_t best_g, next_fin; };

// LDS ring of diagonals, word-major (ring[w * RS_RING + slot]) so that 64 consecutive diagonals are
// read conflict-free; `dirty` has one bit per group of 64 diagonals that received a hit.
// Folds the complete diagonal groups below `lim` (all remaining ones when `final`) into st, in
// increasing diagonal order; every lane of the wave calls it with the same arguments.
template <int WORDS>
__device__ __forceinline__ void d_rs_finalize(unsigned long long *ring, uint32_t *dirty, RsFold &st, const int64_t lim, const bool final, const int lane)
{
    while (st.next_fin + 64 <= lim || (final && st.next_fin < lim)) {
        const int64_t b0 = st.next_fin;
        const uint32_t gbit = 1u << ((uint32_t)((uint64_t)b0 >> 6) & (RS_RING / 64 - 1));
        if (*dirty & gbit) {
            const int slot = (int)((uint64_t)(b0 + lane) & (RS_RING - 1));
            int cnt = 0, first = -1, last = -1;
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                const unsigned long long v = ring[w * RS_RING + slot];
                if (v) {
                    cnt += __popcll(v);
                    if (first < 0) first = w * 64 + (__ffsll((long long)v) - 1);
                    last = w * 64 + 63 - __clzll((long long)v);
                    ring[w * RS_RING + slot] = 0;
                }
            }
            if (lane == 0) *dirty &= ~gbit;
            unsigned long long mask = __ballot(cnt > 0);
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                const int c = __shfl(cnt, l, 64), f = __shfl(first, l, 64), la = __shfl(last, l, 64);
                st.s += c - 1;
                const int len = 8 + (la - f);
                if (len > st.max_len && st.s > (len - 8) / 2) { st.best_r = f; st.best_g = b0 + l + f; st.max_len = len; st.s = 1; }
                mask &= mask - 1;
            }
        }
        st.next_fin = b0 + 64;
    }
}

// the block is ONE wave: LDS operations of a wave complete in order, so ordering needs only a
// compiler barrier + an LDS drain -- unlike __syncthreads() this leaves the prefetched global
// load in flight (hipcc drains vmcnt at every __syncthreads fence)
#define RS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define RS_CHUNK 512             // window positions per trip: one coalesced pac load, RS_PPL consecutive positions per lane;
                                 // live diagonals: span + 63 + RS_CHUNK <= RS_RING
#define RS_PPL (RS_CHUNK / 64)

// WORDS = 64-bit bitmap words per diagonal: a job with read gap rl needs (rl - 8) / 64 + 1 of them; the kernel is
// instantiated for 1, 2 and 4 (8 / 16 / 32 KB of ring, so 8 / 5 / 3 waves per CU) and each instance takes the jobs of its size
template <int WORDS>
__global__ void __launch_bounds__(64)
k_reseed(const DIndex ix, const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off,
         DJob *__restrict__ jobs, const unsigned int *__restrict__ jobtop, uint32_t jobcap, unsigned long long *ctr, const int *__restrict__ abort_p)
{
    __shared__ unsigned char rs[RS_MAX_RL + 9];
    __shared__ uint64_t tmpk[RS_MAX_RL + 1], km[RS_MAX_RL + 1];
    __shared__ uint32_t flt[2048];
    __shared__ unsigned long long ring[RS_RING * WORDS];
    __shared__ uint32_t pacbuf[68];
    __shared__ uint32_t s_dirty;
    __shared__ int s_nk;
    const int lane = threadIdx.x;
    if (*abort_p >= DG_ABORT) return;
    const unsigned int njobs = *jobtop < jobcap ? *jobtop : jobcap;
    const int64_t L = ix.l_pac;
    unsigned long long n_done = 0, w_done = 0;
    for (unsigned int jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
        const DJob job = jobs[jb];
        if (job.found < 0) continue;                    // too long for the LDS ring: serial path in k_report
        const int rl = job.rl, glen = job.glen;
        {
            const int need = rl >= 8 ? (rl - 8) / 64 + 1 : 1;
            if ((need <= 1 ? 1 : (need <= 2 ? 2 : 4)) != WORDS) continue;
        }
        int thr = (int)(rl * 0.85); if (thr < 8) thr = 8;
        n_done += 1; w_done += (unsigned long long)(glen > 0 ? glen : 0);
        const unsigned char *rd = seq + seq_off[job.read] + job.rBegin;
        __syncthreads();
        for (int i = lane; i < rl; i += 64) rs[i] = rd[i];
        for (int i = lane; i < 2048; i += 64) flt[i] = 0;
        for (int i = lane; i < RS_RING * WORDS; i += 64) ring[i] = 0;
        if (lane == 0) s_dirty = 0;
        __syncthreads();
        if (lane == 0) {   // CreateKmerVecFromReadSeq :34-80 on the read gap, position order
            int nk = 0, count = 0, head, tail = 0;
            uint32_t wid = 0;
            while (count < 8 && tail < rl) { if (rs[tail++] != 'N') count++; else count = 0; }
            if (count == 8) {
                head = tail - 8; wid = 0;
                for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(rs[i]);
                tmpk[nk++] = ((uint64_t)wid << 32) | (uint32_t)head;
                for (head += 1; tail < rl; head++, tail++) {
                    if (rs[tail] != 'N') {
                        wid = ((wid & 0x3FFF) << 2) + d_nt4(rs[tail]);
                        tmpk[nk++] = ((uint64_t)wid << 32) | (uint32_t)head;
                    } else {
                        count = 0; tail++;
                        while (count < 8 && tail < rl) { if (rs[tail++] != 'N') count++; else count = 0; }
                        if (count == 8) {
                            head = tail - 8; wid = 0;
                            for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(rs[i]);
                            tmpk[nk++] = ((uint64_t)wid << 32) | (uint32_t)head;
                        } else break;
                    }
                }
            }
            s_nk = nk;
        }
        __syncthreads();
        const int nk = s_nk;
        int found = 0, best_r = 0, max_len = 0;
        int64_t best_g = 0;
        if (nk > 0 && glen >= 8) {
            for (int e = lane; e < nk; e += 64) {       // rank sort by (wid,pos); keys are distinct
                const uint64_t key = tmpk[e];
                int rank = 0;
                for (int j = 0; j < nk; j++) rank += tmpk[j] < key ? 1 : 0;
                km[rank] = key;
                const uint32_t w16 = (uint32_t)(key >> 32) & 0xFFFFu;
                atomicOr(&flt[w16 >> 5], 1u << (w16 & 31));
            }
            __syncthreads();
            const int span = rl - 8;
            RsFold st; st.s = 1; st.max_len = 0; st.best_r = 0; st.best_g = 0;
            st.next_fin = -(int64_t)(((span + 63) >> 6) << 6);          // aligned to the 64-diagonal groups
            // window entirely inside one strand half -> k-mers come from coalesced pac dwords staged in LDS
            const bool fwd = job.Lb >= 0 && job.Lb + glen <= L;
            const bool rev = job.Lb >= L && job.Lb + glen <= 2 * L;
            auto chunk_base = [&](int g0) -> int64_t {                  // first pac byte (4-aligned) of chunk g0
                if (fwd) return (int64_t)(((job.Lb + g0) >> 2) & ~(int64_t)3);
                int64_t ulo = 2 * L - 1 - (job.Lb + g0 + RS_CHUNK - 1) - 7;
                if (ulo < 0) ulo = 0;
                return (int64_t)((ulo >> 2) & ~(int64_t)3);
            };
            uint32_t pre0 = 0, pre1 = 0;
            if (fwd || rev) {
                const uint32_t *src = (const uint32_t *)(ix.pac + chunk_base(0));
                pre0 = src[lane]; if (lane < 4) pre1 = src[64 + lane];
            }
            for (int g0 = 0; g0 + 8 <= glen; g0 += RS_CHUNK) {
                d_rs_finalize<WORDS>(ring, &s_dirty, st, (int64_t)g0 - span, false, lane);
                uint64_t x = 0;                 // the lane's RS_PPL + 7 bases, first base in the top bits
                int o_rev = 0;
                if (fwd || rev) {
                    const int64_t B0 = chunk_base(g0);
                    RS_WAVE_SYNC();
                    pacbuf[lane] = pre0; if (lane < 4) pacbuf[64 + lane] = pre1;
                    if (g0 + RS_CHUNK + 8 <= glen) {                    // prefetch the next chunk while this one is processed
                        const uint32_t *src = (const uint32_t *)(ix.pac + chunk_base(g0 + RS_CHUNK));
                        pre0 = src[lane]; if (lane < 4) pre1 = src[64 + lane];
                    }
                    RS_WAVE_SYNC();
                    // first forward base this lane needs, relative to base 4*B0 of the staged bytes
                    int64_t fb;
                    if (fwd) fb = job.Lb + g0 + RS_PPL * lane - 4 * B0;
                    else { fb = 2 * L - 1 - (job.Lb + g0 + RS_PPL * lane + RS_PPL - 1) - 7 - 4 * B0; if (fb < 0) { o_rev = (int)-fb; fb = 0; } }
                    const int m = (int)(fb >> 4), o = (int)(fb & 15);                 // dword index, base offset inside it
                    const uint32_t w0 = __builtin_bswap32(pacbuf[m]), w1 = __builtin_bswap32(pacbuf[m + 1]), w2 = __builtin_bswap32(pacbuf[m + 2]);
                    const uint64_t hi = ((uint64_t)w0 << 32) | w1;
                    x = o ? ((hi << (2 * o)) | ((uint64_t)w2 >> (32 - 2 * o))) : hi;
                    if (o_rev) x = o_rev > 23 ? 0 : x >> (2 * o_rev);     // window start of the reverse half clipped at forward base 0 (never a valid position)
                }
                uint32_t wid[RS_PPL];
                uint32_t pass = 0;
#pragma unroll
                for (int j = 0; j < RS_PPL; j++) {
                    const int p = g0 + RS_PPL * lane + j;
                    uint32_t w;
                    if (fwd) w = (uint32_t)(x >> (48 - 2 * j)) & 0xFFFFu;
                    else if (rev) {
                        uint32_t f = (uint32_t)(x >> (48 - 2 * (RS_PPL - 1 - j))) & 0xFFFFu;      // forward 8-mer, mirrored position
                        f = ((f & 0x3333u) << 2) | ((f >> 2) & 0x3333u);
                        f = ((f & 0x0F0Fu) << 4) | ((f >> 4) & 0x0F0Fu);
                        f = ((f << 8) | (f >> 8)) & 0xFFFFu;
                        w = f ^ 0xFFFFu;
                    } else w = p + 8 <= glen ? d_window_kmer(ix, job.Lb + p) : 0u;
                    wid[j] = w;
                }
#pragma unroll
                for (int j = 0; j < RS_PPL; j++) {
                    const uint32_t w16 = wid[j] & 0xFFFFu;
                    const bool ok = g0 + RS_PPL * lane + j + 8 <= glen;
                    pass |= (ok ? ((flt[w16 >> 5] >> (w16 & 31)) & 1u) : 0u) << j;
                }
                while (pass) {
                    const int j = __ffs((int)pass) - 1;
                    pass &= pass - 1;
                    const int p = g0 + RS_PPL * lane + j;
                    uint32_t w = wid[0];
#pragma unroll
                    for (int q = 1; q < RS_PPL; q++) w = j == q ? wid[q] : w;
                    int lo = 0, hi2 = nk;
                    while (lo < hi2) { const int mid = (lo + hi2) >> 1; if ((uint32_t)(km[mid] >> 32) < w) lo = mid + 1; else hi2 = mid; }
                    for (; lo < nk && (uint32_t)(km[lo] >> 32) == w; lo++) {
                        const int rp = (int)(uint32_t)km[lo];
                        const int64_t d = (int64_t)p - rp;
                        atomicOr(&ring[(rp >> 6) * RS_RING + (int)((uint64_t)d & (RS_RING - 1))], 1ull << (rp & 63));
                        atomicOr(&s_dirty, 1u << ((uint32_t)((uint64_t)d >> 6) & (RS_RING / 64 - 1)));
                    }
                }
                RS_WAVE_SYNC();
            }
            d_rs_finalize<WORDS>(ring, &s_dirty, st, (int64_t)(glen - 8) + 1, true, lane);
            max_len = st.max_len; best_r = st.best_r; best_g = st.best_g;
            found = (max_len >= thr && max_len > 0) ? 1 : 0;
        }
        if (lane == 0) {
            DJob &o = jobs[jb];
            o.found = found; o.len = max_len; o.rPos = best_r + job.rBegin; o.gPos = best_g + job.Lb;
        }
    }
    if (lane == 0) { if (n_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEED, n_done); if (w_done) atomicAdd(d_ctr_stripe(ctr) + CTR_RESEEDW, w_done); }
}
