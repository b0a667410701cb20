// dart_amd/csrc/dg_pair.h -- k_pair: everything between the located seeds and the finished records for the units (read pairs,
// or single reads) whose alignment is "exact seeds on one diagonal with substitutions between them" -- 70-98 % of a DNA batch --
// in ONE kernel, lane = unit, all intermediate state in LDS.  Units outside that pattern leave k_pair as candidates in
// memory, in the form the general path (k_prep / k_reseed / k_report / k_finalize) takes, on a compact list.
//
// Per unit the kernel does what the reference does in Mapping.cpp:598-639 for such a read pair:
//   sort the seeds by (gPos,rPos)                         IdentifySeedPairs tail, AlignmentCandidates.cpp:212
//   cluster them into candidates                          GenerateAlignmentCandidate :241-288
//   pair the mates' candidates, drop unmated / redundant  Mapping.cpp:371-477
//   report every live candidate                           GenMappingReport :1079-1207 where it reduces to [S] M [S]
//   settle the pair, FLAG, MAPQ                           Mapping.cpp:74-206, 479-530
// and writes dg_read_out / dg_report_out / CIGAR ops at their final places: the offsets (reports per read, CIGAR ops per
// report, position on the list of units for the general path) come from a single-pass scan inside the kernel (dg_scan.h).
//
// Before this kernel the same work was k_chain -> 2 scans -> k_prep -> k_report_diag -> k_finalize -> CIGAR compaction, with
// seeds, candidates, working regions and reports crossing HBM between them (3.4 GB per 2 M reads against 0.3 GB of input
// and output).
//
// The pair-settling / FLAG / MAPQ rules are written once, against a "report view" (n, aln, set_aln, mate, set_mate, bdir,
// set_flag, or_flag): RepLds over k_pair's packed LDS words, RepMem over dg_report_out records for the general path's k_finalize.
#pragma once
#include "../../include/dartgpu.h"
#include "dg_common.h"
#include "dg_chain.h"
#include "dg_report.h"

// ---------------------------------------------------------------------------------------------
// pair settling, FLAG, MAPQ over a report view
// ---------------------------------------------------------------------------------------------
// CheckPairedFinalAlignments (Mapping.cpp:479-530).  If the two best reports are each other's mates nothing changes (unless
// -m).  Otherwise, when both reads have a live report, the mated pair of live reports with the largest combined score
// becomes the best of both reads (first such pair on a tie).  With a mated pair settled, read 1's reports that do not
// carry the pair's scores lose their score and their link; without one, every link is cut and only reports scoring the
// read's best stay alive.
template <class R1, class R2>
__host__ __device__ inline void d_settle_pair(const DParams &pr, DRead &m1, R1 &p1, DRead &m2, R2 &p2)
{
    bool linked = p1.mate(m1.iBest) == m2.iBest;
    if (linked && !pr.multi_hit) return;
    if (!linked && m1.score > 0 && m2.score > 0) {
        int best_sum = 0;
        for (int i = 0; i < m1.CanNum; i++) {
            const int a = p1.aln(i), j = p1.mate(i);
            if (a <= 0 || j < 0) continue;
            const int b = p2.aln(j);
            if (b <= 0) continue;
            linked = true;
            if (a + b > best_sum) { best_sum = a + b; m1.iBest = i; m1.score = a; m2.iBest = j; m2.score = b; }
        }
    }
    if (linked) {
        for (int i = 0; i < m1.CanNum; i++) {
            const int j = p1.mate(i);
            if (p1.aln(i) != m1.score || (j >= 0 && p2.aln(j) != m2.score)) { p1.set_aln(i, 0); p1.set_mate(i, -1); }
        }
        return;
    }
    for (int i = 0; i < m1.CanNum; i++) { p1.set_mate(i, -1); const int a = p1.aln(i); if (a > 0 && a != m1.score) p1.set_aln(i, 0); }
    for (int j = 0; j < m2.CanNum; j++) { p2.set_mate(j, -1); const int b = p2.aln(j); if (b > 0 && b != m2.score) p2.set_aln(j, 0); }
}

template <class R>
__host__ __device__ inline void d_flag_single(const DRead &r, R &p)   // SetSingleAlignmentFlag, Mapping.cpp:74-99
{
    if (r.score > r.sub_score) p.set_flag(r.iBest, p.bdir(r.iBest) ? 0 : 0x10);
    else if (r.score > 0) { for (int i = 0; i < r.CanNum; i++) if (p.aln(i) > 0) p.set_flag(i, p.bdir(i) ? 0 : 0x10); }
    else p.set_flag(0, 0x4);
}

// one read of a pair whose FLAGs are not both unique-best (Mapping.cpp:124-153 / :155-184); base = 0x41 or 0x81
template <class RA, class RB>
__host__ __device__ inline void d_flag_mate(const DRead &a, RA &pa, const DRead &b, RB &pb, int base)
{
    auto mapped = [&](int i) {
        const int j = pa.mate(i);
        pa.set_flag(i, base | (pa.bdir(i) ? 0x20 : 0x10) | ((j >= 0 && pb.aln(j) > 0) ? 0x2 : 0x8));
    };
    if (a.score > a.sub_score) mapped(a.iBest);
    else if (a.score > 0) { for (int i = 0; i < a.CanNum; i++) if (pa.aln(i) > 0) mapped(i); }
    else pa.set_flag(0, base | 0x4 | (b.score == 0 ? 0x8 : (pb.bdir(b.iBest) ? 0x10 : 0x20)));
}

template <class R1, class R2>
__host__ __device__ inline void d_flag_pair(const DRead &r1, R1 &p1, const DRead &r2, R2 &p2)   // SetPairedAlignmentFlag, Mapping.cpp:101-186
{
    if (r1.score > r1.sub_score && r2.score > r2.sub_score) {
        const int i = r1.iBest, j = r2.iBest;
        const int proper = j == p1.mate(i) ? 0x2 : 0;
        p1.set_flag(i, 0x41 | proper | (p1.bdir(i) ? 0x20 : 0x10));
        p2.set_flag(j, 0x81 | proper | (p2.bdir(j) ? 0x20 : 0x10));
    } else {
        d_flag_mate(r1, p1, r2, p2, 0x41);
        d_flag_mate(r2, p2, r1, p1, 0x81);
    }
}

template <class R>
__host__ __device__ inline void d_mapq(DRead &r, const R &p)   // EvaluateMAPQ, Mapping.cpp:188-206
{
    if (r.score == 0 || r.score == r.sub_score) r.mapq = 0;
    else if (r.sub_score == 0 || r.score > r.sub_score) r.mapq = 50;
    else {
        int n = 0;
        for (int i = 0; i < r.CanNum; i++) if (p.aln(i) == r.score) n++;
        r.mapq = n >= 10 ? 0 : (n >= 4 ? 1 : (n == 3 ? 2 : (n == 2 ? 3 : 50)));
    }
}

// view over dg_report_out records in memory
template <typename ReportT>
struct RepMem {
    ReportT *p;
    __host__ __device__ int aln(int i) const { return p[i].aln_score; }
    __host__ __device__ void set_aln(int i, int v) { p[i].aln_score = v; }
    __host__ __device__ int mate(int i) const { return p[i].paired_idx; }
    __host__ __device__ void set_mate(int i, int v) { p[i].paired_idx = v; }
    __host__ __device__ int bdir(int i) const { return p[i].bdir; }
    __host__ __device__ void set_flag(int i, int v) { p[i].flag = v; }
};

// ---------------------------------------------------------------------------------------------
// k_pair's per-lane state: element e of a lane's array lives at base[e * S] (S = workgroup size on the device: neighbouring
// lanes touch neighbouring banks; S = 1 in the host-compiled checks)
//   key[16]  the unit's seeds, mate 1 then mate 2, each sorted
//   cw[16]   candidate words, mate 1's then mate 2's:
//            first seed (5, index inside the mate's segment) | seeds (5) << 5 | score (12) << 10 | mate (5, 31 = none) << 22 | slot (4, 15 = none) << 27
//   rw[2*6]  report slots of the LIVE candidates (score != 0 after the candidate rules), two words each:
//            w0 = POS (32) | chromosome (16) << 32 | AlnScore (12) << 48 | bDir << 60 | has coordinates << 61
//            w1 = leading S (12) | M (12) << 12 | trailing S (12) << 24 | FLAG (12) << 36 | mismatches (12) << 48
// ---------------------------------------------------------------------------------------------
#ifndef PU_THREADS
#define PU_THREADS 256
#endif
#define PU_SEEDS UNIT_MAX_SEEDS
#ifndef PU_SLOTS
#define PU_SLOTS 6
#endif
#define PU_LOC_MAX 256        // ChrLocMap keys a workgroup keeps in LDS (two per chromosome)                        // 2 workgroups of 256 lanes per CU: (16 x 8 + 16 x 4 + 2 x 6 x 8) bytes per lane = 72 KB each

__host__ __device__ __forceinline__ uint32_t cw_make(int first, int count, int score) { return (uint32_t)first | ((uint32_t)count << 5) | ((uint32_t)score << 10) | (31u << 22) | (15u << 27); }
__host__ __device__ __forceinline__ int cw_first(uint32_t w) { return (int)(w & 31u); }
__host__ __device__ __forceinline__ int cw_count(uint32_t w) { return (int)((w >> 5) & 31u); }
__host__ __device__ __forceinline__ int cw_score(uint32_t w) { return (int)((w >> 10) & 0xFFFu); }
__host__ __device__ __forceinline__ int cw_mate(uint32_t w) { const int m = (int)((w >> 22) & 31u); return m == 31 ? -1 : m; }
__host__ __device__ __forceinline__ int cw_slot(uint32_t w) { return (int)((w >> 27) & 15u); }

template <int S>
struct CandLds {                      // candidate view (dg_chain.h rules) of one mate
    const SKey *key; uint32_t *cw; int cnt;      // key, cw already point at the mate's first element
    __host__ __device__ int n() const { return cnt; }
    __host__ __device__ int score(int i) const { return cw_score(cw[i * S]); }
    __host__ __device__ void set_score(int i, int v) { cw[i * S] = (cw[i * S] & ~(0xFFFu << 10)) | ((uint32_t)v << 10); }
    __host__ __device__ int64_t diag(int i) const { const int64_t d = sk_diag(key[cw_first(cw[i * S]) * S]); return d < 0 ? 0 : d; }
    __host__ __device__ int mate(int i) const { return cw_mate(cw[i * S]); }
    __host__ __device__ void set_mate(int i, int v) { cw[i * S] = (cw[i * S] & ~(31u << 22)) | ((uint32_t)(v < 0 ? 31 : v) << 22); }
};

template <int S>
struct RepLds {                       // report view of one mate: report i = candidate i (report 0 exists even without candidates)
    uint32_t *cw; uint64_t *rw; int nc; int flag0;        // flag0: FLAG of report 0 when it has no slot (an unmapped read)
    __host__ __device__ int slot(int i) const { return i < nc ? cw_slot(cw[i * S]) : 15; }
    __host__ __device__ int aln(int i) const { const int s = slot(i); return s == 15 ? 0 : (int)((rw[(2 * s) * S] >> 48) & 0xFFFu); }
    __host__ __device__ void set_aln(int i, int v) { const int s = slot(i); if (s != 15) rw[(2 * s) * S] = (rw[(2 * s) * S] & ~(0xFFFull << 48)) | ((uint64_t)(uint32_t)v << 48); }
    __host__ __device__ int mate(int i) const { return i < nc ? cw_mate(cw[i * S]) : -1; }
    __host__ __device__ void set_mate(int i, int v) { if (i < nc) cw[i * S] = (cw[i * S] & ~(31u << 22)) | ((uint32_t)(v < 0 ? 31 : v) << 22); }
    __host__ __device__ int bdir(int i) const { const int s = slot(i); return s == 15 ? 0 : (int)((rw[(2 * s) * S] >> 60) & 1u); }
    __host__ __device__ void set_flag(int i, int v) {
        const int s = slot(i);
        if (s != 15) rw[(2 * s + 1) * S] = (rw[(2 * s + 1) * S] & ~(0xFFFull << 36)) | ((uint64_t)(uint32_t)v << 36);
        else if (i == 0) flag0 = v;
    }
    __host__ __device__ int flag(int i) const { const int s = slot(i); return s == 15 ? (i == 0 ? flag0 : 0) : (int)((rw[(2 * s + 1) * S] >> 36) & 0xFFFu); }
};

// what a unit looks like after d_unit_process
// -DDG_PAIR_PROF (profiles/probes/pair_phases.sh): cycle stamps between the phases of k_pair, summed per wave and printed for a few tiles
#if defined(DG_PAIR_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define PP_STAMP(pp, pt, i) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); (pp)[i] += _n - (pt); (pt) = _n; } while (0)
#else
#define PP_STAMP(pp, pt, i) do { } while (0)
#endif
struct UnitState {
#ifdef DG_PAIR_PROF
    unsigned long long pp[10], pt;
#endif
    DRead rd[2];
    int nc[2];            // candidates per mate
    int flag0[2];
    bool fast;            // records complete in LDS; else: candidates go to the general path
    uint32_t n_cig;       // CIGAR ops of its reports (fast only)
    uint32_t n_nw, n_cells;   // nw_alignment calls / cells the reference spends on this unit's segment pairs (counters)
};

// insertion sort of a lane's seed segment in its LDS slice
template <int S>
__host__ __device__ __forceinline__ void d_unit_sort(SKey *key, int n)
{
    for (int i = 1; i < n; i++) {
        const SKey x = key[i * S];
        int j = i;
        for (; j > 0 && key[(j - 1) * S] > x; j--) key[j * S] = key[(j - 1) * S];
        key[j * S] = x;
    }
}

// nw_alignment (nw_alignment.cpp:18-82, integers x2 as d_nw in dg_report.h) of two strings of the same length g <= 8, asked one
// question: is the traceback the plain diagonal (g columns of M, no gap)?  The DP runs row by row with the previous row in
// registers (one strip of 8 columns, as d_nw's strips); only the cells (i,i) decide: the walk from (g,g) stays on the diagonal
// exactly when none of them equals its r or t value.  a8 = read bases, b8 = genome bases (byte k = base k).
#define SMALL_NW 8
__host__ __device__ inline bool d_small_nw_is_diagonal(uint64_t a8, uint64_t b8, int g)
{
    int sp[SMALL_NW], tp[SMALL_NW];
    uint8_t cb[SMALL_NW];
#pragma unroll
    for (int q = 0; q < SMALL_NW; q++) { sp[q] = -2 - (q + 1); tp[q] = -131072; cb[q] = q < g ? d_nt4((unsigned char)(b8 >> (8 * q))) : (uint8_t)7; }
    int diag0 = 0;
    bool leaves = false;
    for (int i = 1; i <= g; i++) {
        int left_s = -2 - i, left_r = -131072;
        const uint8_t ca = d_nt4((unsigned char)(a8 >> (8 * (i - 1))));
        int diag = diag0;
        diag0 = left_s;
#pragma unroll
        for (int q = 0; q < SMALL_NW; q++) {
            int x = left_r - 1, y = left_s - 3;
            const int r = x > y ? x : y;
            x = tp[q] - 1; y = sp[q] - 3;
            const int t = x > y ? x : y;
            const int sv = d_tr2(d_max3(diag + (ca == cb[q] ? 3 : -3), r, t));
            leaves = leaves || (q == i - 1 && (sv == r || sv == t));
            diag = sp[q];
            sp[q] = sv; tp[q] = t;
            left_s = sv; left_r = r;
        }
    }
    return !leaves;
}

// A read as the fused kernel looks at it: the ASCII bytes of dg_map_batch (case, '-' and IUPAC letters matter to the reference: tools.cpp:40-104), or the
// 2-bit + mask words of a packed batch (dg_map_batch_packed: A/C/G/T/N by contract, so the characters follow from the words and the ASCII copy of the
// batch -- 202 MB per million pairs, written by k_unpack and read back here -- is only made for the units of the general path: k_prep).
// get8(i, e): the characters i .. i + e - 1 (e <= 8) as the bytes of a word, character i in the low byte -- the form d_ref8 gives the genome in
struct ReadAscii {
    const unsigned char *p;
    __host__ __device__ __forceinline__ uint64_t get8(int i, int e) const { uint64_t v = 0; for (int t = 0; t < e; t++) v |= (uint64_t)p[i + t] << (8 * t); return v; }
    // characters i .. i + e - 1 against RefSequence[g .. g + e): how many differ, and whether one of the read's is a '-' (tools.cpp:130-164 compares characters)
    __host__ __device__ __forceinline__ int mismatches8(const DIndex &ix, int i, int e, int64_t g, bool &dash) const {
        const uint64_t ref = d_ref8(ix, g), a8 = get8(i, e), keep = e == 8 ? ~0ull : (1ull << (8 * e)) - 1ull, lo7 = 0x7F7F7F7F7F7F7F7Full, hi1 = 0x8080808080808080ull;
        const uint64_t x = (a8 ^ ref) & keep, y = (a8 ^ 0x2D2D2D2D2D2D2D2Dull) | ~keep;
        dash = dash || ((~(((y & lo7) + lo7) | y)) & hi1) != 0ull;
        return __builtin_popcountll((((x & lo7) + lo7) | x) & hi1);
    }
};
struct ReadWords {
    const uint32_t *w; int W2;           // k_encode's format: W2 words of 2-bit codes (first base on top), then W2 words with 0b11 where the base is no A/C/G/T
    __host__ __device__ __forceinline__ uint64_t get8(int i, int e) const {
        // eight codes and eight masks from two words each (the window may straddle a word), then 'A' 'C' 'G' 'T' by code, 'N' where masked
        const int w0 = i >> 4, sh = (i & 15) << 1;
        const uint32_t chi = w[w0], clo = w0 + 1 < W2 ? w[w0 + 1] : 0u, mhi = w[W2 + w0], mlo = w0 + 1 < W2 ? w[W2 + w0 + 1] : 0xFFFFFFFFu;
        const uint32_t c16 = (sh ? (chi << sh) | (clo >> (32 - sh)) : chi) >> 16, m16 = (sh ? (mhi << sh) | (mlo >> (32 - sh)) : mhi) >> 16;
        uint64_t v = 0;
        for (int t = 0; t < e; t++) {
            const uint32_t c = (c16 >> (14 - 2 * t)) & 3u, m = (m16 >> (14 - 2 * t)) & 3u;
            v |= (uint64_t)(m ? 0x4Eu : ((0x54474341u >> (8u * c)) & 0xFFu)) << (8 * t);
        }
        return v;
    }
    // the same count in the 2-bit domain: a packed batch's reads are A/C/G/T/N by contract (never a '-'), RefSequence is A/C/G/T (pac), so characters differ
    // exactly where the codes differ or the read's base is masked -- no character is ever built.  Windows that are not wholly inside one strand of the
    // text (d_ref8's per-character path: 0 outside) take the character form.
    __host__ __device__ __forceinline__ int mismatches8(const DIndex &ix, int i, int e, int64_t g, bool &dash) const {
        const int64_t L = ix.l_pac;
        uint32_t g16;                              // eight genome codes, base k at bits 15-2k .. 14-2k
        if (g >= 0 && g + 8 <= L) g16 = (__builtin_bswap32(*(const uint32_a1 *)(ix.pac + (g >> 2))) << ((g & 3) << 1)) >> 16;
        else if (g >= L && g + 8 <= 2 * L) {
            const int64_t lo = 2 * L - 1 - g - 7;                                                                  // fwd[lo .. lo+7], Ref[g+k] = 3 - fwd[lo+7-k]
            const uint32_t w = (__builtin_bswap32(*(const uint32_a1 *)(ix.pac + (lo >> 2))) << ((lo & 3) << 1)) >> 16;   // fwd[lo+q] at bits 15-2q .. 14-2q
            uint32_t r = ((w & 0x3333u) << 2) | ((w >> 2) & 0x3333u); r = ((r & 0x0F0Fu) << 4) | ((r >> 4) & 0x0F0Fu); r = ((r & 0x00FFu) << 8) | (r >> 8);   // 2-bit groups reversed
            g16 = ~r & 0xFFFFu;
        } else {
            const uint64_t ref = d_ref8(ix, g), a8 = get8(i, e), keep = e == 8 ? ~0ull : (1ull << (8 * e)) - 1ull, lo7 = 0x7F7F7F7F7F7F7F7Full, hi1 = 0x8080808080808080ull;
            const uint64_t x = (a8 ^ ref) & keep;
            return __builtin_popcountll((((x & lo7) + lo7) | x) & hi1);
        }
        const int w0 = i >> 4, sh = (i & 15) << 1;
        const uint32_t chi = w[w0], clo = w0 + 1 < W2 ? w[w0 + 1] : 0u, mhi = w[W2 + w0], mlo = w0 + 1 < W2 ? w[W2 + w0 + 1] : 0xFFFFFFFFu;
        const uint32_t c16 = (sh ? (chi << sh) | (clo >> (32 - sh)) : chi) >> 16, m16 = (sh ? (mhi << sh) | (mlo >> (32 - sh)) : mhi) >> 16;
        const uint32_t x = c16 ^ g16, keep = (0xFFFF0000u >> (2 * e)) & 0xFFFFu;
        return __builtin_popcount(((x | (x >> 1)) | m16) & 0x5555u & keep);
    }
};

// GenMappingReport for the live candidates of one mate, where it reduces to "[S] M [S]": every seed of the candidate exact and on
// one diagonal, at least one read base between neighbours, and the bases between them either equal-length with <= 2 and <= 20 %
// mismatches (ProcessNormalSequencePair's M shortcut, tools.cpp:137-141) or a single substituted base (a 1 x 1 nw_alignment).
// Then the clean-up passes, re-seeding, gap filling, splice detection and overlap trimming are all the identity
// (they act on repeated rPos, order inversions and diagonal changes).  Returns false when some live candidate is outside
// the pattern (nothing of this unit is then kept: the general path redoes it).
template <int S, class RD>
__host__ __device__ inline bool d_unit_reports(const DIndex &ix, const LocTab &lt, const DParams &pr, bool first, const RD seq, int len,
                                               const SKey *key, uint32_t *cw, int nc, uint64_t *rw, int &n_slot, DRead &rd, uint32_t &n_cig, uint32_t &n_nw, uint32_t &n_cells)
{
    const int64_t L = ix.l_pac;
    rd.score = rd.sub_score = rd.mis_num = rd.mapq = rd.iBest = 0;
    rd.CanNum = nc > 0 ? nc : 1;
    for (int i = 0; i < nc; i++) {
        const uint32_t w = cw[i * S];
        if (cw_score(w) == 0) continue;
        const int n = cw_count(w), f = cw_first(w);
        const SKey k0 = key[f * S];
        const int64_t diag = sk_diag(k0);
        if (n > 1 && diag == -1) return false;
        SKey prev = k0;
        int aln = sk_rlen(k0), mis = 0;
        uint32_t calls = 0, cells = 0;
        for (int k = 1; k < n; k++) {
            const SKey cur = key[(f + k) * S];
            const int from = sk_rpos(prev) + sk_rlen(prev), g = sk_rpos(cur) - from;
            if (sk_diag(cur) != diag || g < 1) return false;
            const int64_t gp = sk_gpos(prev) + sk_rlen(prev);
            int nm = 0;
            bool dash = false;
            for (int q = 0; q < g; q += 8) nm += seq.mismatches8(ix, from + q, g - q < 8 ? g - q : 8, gp + q, dash);     // eight characters at once
            if (nm <= 2 && nm <= (int)(g * 0.2)) { aln += g - nm; mis += nm; }
            else if (g == 1 && !dash) { calls++; cells++; mis += 1; }
            else if (g <= SMALL_NW && !dash && d_small_nw_is_diagonal(seq.get8(from, g), d_ref8(ix, gp), g)) {
                // two substitutions a few bases apart: ProcessNormalSequencePair calls nw_alignment (tools.cpp:142-163), the
                // alignment is g columns of M, AddNewCigarElements scores the identical characters
                calls++; cells += (uint32_t)(g * g); aln += g - nm; mis += nm;
            }
            else return false;
            aln += sk_rlen(cur);
            prev = cur;
        }
        const int64_t gPos = sk_gpos(k0), end_gPos = sk_gpos(prev) + sk_rlen(prev) - 1;
        if (n > 1 && ((gPos < L) != (end_gPos < L))) return false;          // CheckCoordinateValidity :136-163 would reject it
        const int head = sk_rpos(k0), span = sk_rpos(prev) + sk_rlen(prev) - head, tail = len - head - span;
        if (mis > pr.max_mismatch) aln = 0;
        if (n_slot >= PU_SLOTS) return false;
        const int slot = n_slot++;
        uint64_t w0 = 0;
        if (aln > 0) {                                                       // GenCoordinateInfo :83-116
            const int lb = d_loc_lower_bound(lt, gPos);
            const int chr = lt.chr[lb];
            int64_t pos; int bdir;
            if (gPos < L) { bdir = first ? 1 : 0; pos = gPos + 1 - lt.off[chr]; }
            else { bdir = first ? 0 : 1; pos = lt.key[lb] - end_gPos + 1; }
            if (pos <= 0 || pos > 0xFFFFFFFFll || chr > 0xFFFF) return false;   // (a report the reference zeroes late, or fields too wide for the slot)
            w0 = (uint64_t)pos | ((uint64_t)chr << 32) | ((uint64_t)aln << 48) | ((uint64_t)bdir << 60) | (1ull << 61);
            n_cig += 1u + (head > 0) + (tail > 0);
            if (aln > rd.score) { rd.iBest = i; rd.mis_num = mis; rd.sub_score = rd.score; rd.score = aln; }
            else if (aln == rd.score) rd.sub_score = rd.score;
        }
        rw[(2 * slot) * S] = w0;
        rw[(2 * slot + 1) * S] = (uint64_t)head | ((uint64_t)span << 12) | ((uint64_t)tail << 24) | ((uint64_t)mis << 48);
        cw[i * S] = (w & ~(15u << 27)) | ((uint32_t)slot << 27);
        n_nw += calls; n_cells += cells;
    }
    return true;
}

// One unit from sorted-or-not seeds in key[0 .. n1+n2) to either finished records in LDS (st.fast) or candidate words for the
// general path.  try_fast = false: the candidate stage only (dg_probe_seeds, chr tables too wide for the slots).
template <int S, class RD>
__host__ __device__ inline void d_unit_process_rd(const DIndex &ix, const LocTab &lt, const DParams &pr, bool paired, int n1, int n2, int len1, int len2,
                                               const RD seq1, const RD seq2, SKey *key, uint32_t *cw, uint64_t *rw,
                                               bool try_fast, UnitState &st)
{
    d_unit_sort<S>(key, n1);
    if (paired) d_unit_sort<S>(key + n1 * S, n2);
    int k = 0;
    st.nc[0] = d_cluster_seeds(lt, pr, len1, n1, [&](int i) { return key[i * S]; },
                               [&](int first, int count, int score, int64_t) { cw[(k++) * S] = cw_make(first, count, score); });
    st.nc[1] = 0;
    CandLds<S> a{key, cw, st.nc[0]}, b{key + n1 * S, cw + st.nc[0] * S, 0};
    if (paired) {
        st.nc[1] = b.cnt = d_cluster_seeds(lt, pr, len2, n2, [&](int i) { return key[(n1 + i) * S]; },
                                           [&](int first, int count, int score, int64_t) { cw[(k++) * S] = cw_make(first, count, score); });
    }
    PP_STAMP(st.pp, st.pt, 1);       // sort + clustering
    d_candidate_rules(paired, a, b);
    PP_STAMP(st.pp, st.pt, 2);       // candidate rules (pairing, redundancy)
    st.fast = false; st.n_cig = 0; st.n_nw = st.n_cells = 0; st.flag0[0] = st.flag0[1] = 0;
    if (!try_fast) return;
    int n_slot = 0;
    uint32_t n_cig = 0, n_nw = 0, n_cells = 0;
    if (!d_unit_reports<S>(ix, lt, pr, true, seq1, len1, key, cw, st.nc[0], rw, n_slot, st.rd[0], n_cig, n_nw, n_cells)) return;
    if (paired && !d_unit_reports<S>(ix, lt, pr, false, seq2, len2, key + n1 * S, cw + st.nc[0] * S, st.nc[1], rw, n_slot, st.rd[1], n_cig, n_nw, n_cells)) return;
    PP_STAMP(st.pp, st.pt, 3);       // the reports of the live candidates (read bases, genome words, chromosome table)
    RepLds<S> p1{cw, rw, st.nc[0], 0}, p2{cw + st.nc[0] * S, rw, st.nc[1], 0};
    if (paired) {
        d_settle_pair(pr, st.rd[0], p1, st.rd[1], p2);
        d_flag_pair(st.rd[0], p1, st.rd[1], p2);
        d_mapq(st.rd[1], p2);
    } else d_flag_single(st.rd[0], p1);
    d_mapq(st.rd[0], p1);
    st.flag0[0] = p1.flag0; st.flag0[1] = p2.flag0;
    st.fast = true; st.n_cig = n_cig; st.n_nw = n_nw; st.n_cells = n_cells;
    PP_STAMP(st.pp, st.pt, 4);       // pair settling, FLAG, MAPQ
}
template <int S>
__host__ __device__ inline void d_unit_process(const DIndex &ix, const DParams &pr, bool paired, int n1, int n2, int len1, int len2,
                                               const unsigned char *seq1, const unsigned char *seq2, SKey *key, uint32_t *cw, uint64_t *rw,
                                               bool try_fast, UnitState &st)
{
    d_unit_process_rd<S, ReadAscii>(ix, d_loc_tab(ix), pr, paired, n1, n2, len1, len2, ReadAscii{seq1}, ReadAscii{seq2}, key, cw, rw, try_fast, st);
}

// ---- the compact record types (include/dartgpu.h: dg_read_c 12 bytes, dg_report_c 16 bytes), written by the kernels that write the full
// records (k_pair for the units it finishes, k_emit_slow for the general path's) -- round 2 re-read the full records in two more launches
// and a three-launch scan.  `slow` marks a report whose stored CIGAR ops lie in the second region of the op array (dartgpu.h).
// Returns false when a field does not fit (the host then answers DG_ERR_RANGE: the caller takes the full records).
struct CompactOut { dg_read_c *reads; dg_report_c *reports; uint32_t *cigar; unsigned int *bad; };
__host__ __device__ __forceinline__ bool d_compact_read(const dg_read_out &r, dg_read_c &o)
{
    o.score = (uint16_t)r.score; o.sub_score = (uint16_t)r.sub_score; o.mis_num = (uint16_t)r.mis_num; o.mapq = (uint8_t)r.mapq; o.n_sj = (uint8_t)r.n_sj;
    o.n_rep = (uint16_t)r.n_rep; o.best = (uint16_t)r.best;
    return !((uint32_t)r.score > 0xFFFFu || (uint32_t)r.sub_score > 0xFFFFu || (uint32_t)r.mis_num > 0xFFFFu || (uint32_t)r.mapq > 0xFFu || (uint32_t)r.n_sj > 0xFFu ||
             (uint32_t)r.n_rep > 0xFFFFu || (uint32_t)r.best > 0xFFFFu);
}
// plain = the CIGAR is the single op "<read length>M" and is not stored
__host__ __device__ __forceinline__ bool d_compact_report(const dg_report_out &p, bool plain, bool slow, dg_report_c &q)
{
    q.pos = (int32_t)p.pos; q.aln_score = (uint16_t)p.aln_score; q.flag = (uint16_t)p.flag; q.paired_idx = (int16_t)p.paired_idx;
    q.chr = p.chr < 0 ? (uint16_t)0xFFFFu : (uint16_t)p.chr; q.sj_type = (int8_t)p.sj_type; q.bdir = (uint8_t)p.bdir; q.pad = slow ? 1u : 0u;
    q.n_cigar = plain ? (uint8_t)DG_CIGAR_FULL_MATCH : (uint8_t)p.n_cigar;
    return !(p.pos != (int64_t)(int32_t)p.pos || (uint32_t)p.aln_score > 0xFFFFu || (uint32_t)p.flag > 0xFFFFu || p.paired_idx > 32767 || p.paired_idx < -1 ||
             p.chr >= 0xFFFF || p.n_cigar > 254u || p.sj_type < -128 || p.sj_type > 127 || (uint32_t)p.bdir > 1u);
}

// stored (compact) CIGAR ops of the reports of one read of a fast unit: 0 for a plain full-length match, else 1 + soft clips
template <int S>
__host__ __device__ inline uint32_t d_unit_compact_ops(const DRead &rd, const RepLds<S> &p)
{
    uint32_t n = 0;
    for (int i = 0; i < rd.CanNum; i++) {
        const int s = p.slot(i);
        if (s == 15) continue;
        const uint64_t w0 = p.rw[(2 * s) * S], w1 = p.rw[(2 * s + 1) * S];
        if (!((w0 >> 61) & 1ull)) continue;
        const uint32_t head = (uint32_t)(w1 & 0xFFFu), tail = (uint32_t)((w1 >> 24) & 0xFFFu);
        n += (head || tail) ? 1u + (head > 0) + (tail > 0) : 0u;
    }
    return n;
}

// the records of one read of a fast unit, written at their final places; returns the CIGAR ops written.  co.reads != nullptr: the compact
// records too (the read's at co.reads, its reports at co.reports + rep_off, its stored ops from co.cigar + cigc_off on)
template <int S>
__host__ __device__ inline uint32_t d_unit_emit_read(bool first, const DRead &rd, const RepLds<S> &p, uint32_t rep_off, uint32_t cig_off,
                                                     dg_read_out *rout_r, dg_report_out *reports, uint32_t *cigar, const CompactOut &co = CompactOut{nullptr, nullptr, nullptr, nullptr},
                                                     dg_read_c *rc_r = nullptr, uint32_t cigc_off = 0, bool full = true)
{
    // full = false (round 5): the caller takes the compact records only (dg_map_batch_compact) -- the 36 + 40 bytes per read of the full types and their CIGAR ops
    // are not written for the units this kernel finishes (the general path's reads always get theirs: k_finalize / k_emit_slow work on them)
    dg_read_out o;
    o.score = rd.score; o.sub_score = rd.sub_score; o.mis_num = rd.mis_num; o.mapq = rd.mapq; o.n_rep = rd.CanNum; o.best = rd.iBest;
    o.rep_off = (int32_t)rep_off; o.sj_off = 0; o.n_sj = 0;
    if (full) *rout_r = o;
    bool fits = true;
    if (co.reads) { dg_read_c oc; fits = d_compact_read(o, oc); *rc_r = oc; }
    uint32_t used = 0, used_c = 0;
    for (int i = 0; i < rd.CanNum; i++) {
        dg_report_out rp;
        rp.aln_score = 0; rp.sj_type = -1; rp.flag = p.flag(i); rp.paired_idx = p.mate(i); rp.chr = -1; rp.bdir = 0; rp.pos = 0;
        rp.cigar_off = cig_off + used; rp.n_cigar = 0;
        const int s = p.slot(i);
        if (s != 15) {
            const uint64_t w0 = p.rw[(2 * s) * S], w1 = p.rw[(2 * s + 1) * S];
            if ((w0 >> 61) & 1ull) {
                rp.aln_score = (int32_t)((w0 >> 48) & 0xFFFu); rp.chr = (int32_t)((w0 >> 32) & 0xFFFFu); rp.bdir = (int32_t)((w0 >> 60) & 1u);
                rp.pos = (int64_t)(w0 & 0xFFFFFFFFull);
                const uint32_t head = (uint32_t)(w1 & 0xFFFu), span = (uint32_t)((w1 >> 12) & 0xFFFu), tail = (uint32_t)((w1 >> 24) & 0xFFFu);
                const bool rev = (rp.bdir != 0) != first;                    // bDir = mate 1 on the forward half / mate 2 on the reverse half (:83-116);
                                                                             // a candidate on the reverse half has its CIGAR reversed (:1179)
                const uint32_t lead = rev ? tail : head, trail = rev ? head : tail;
                // the ops [S] M [S] in registers (no indexed local array), then to whichever op arrays are wanted
                const uint32_t o_mid = CIG(span, OP_M), o_trail = CIG(trail, OP_S);
                const uint32_t a0 = lead ? CIG(lead, OP_S) : o_mid, a1 = lead ? o_mid : o_trail, a2 = o_trail;
                const uint32_t m = 1u + (lead ? 1u : 0u) + (trail ? 1u : 0u);
                if (full) { uint32_t *c = cigar + cig_off + used; c[0] = a0; if (m > 1) c[1] = a1; if (m > 2) c[2] = a2; }
                rp.n_cigar = m; used += m;
                if (co.reads && m > 1) { uint32_t *c = co.cigar + cigc_off + used_c; c[0] = a0; c[1] = a1; if (m > 2) c[2] = a2; used_c += m; }   // ("<span>M" alone = the whole read: not stored)
            }
        }
        if (full) reports[rep_off + (uint32_t)i] = rp;
        if (co.reads) { dg_report_c q; fits = d_compact_report(rp, rp.n_cigar == 1u, false, q) && fits; co.reports[rep_off + (uint32_t)i] = q; }
    }
    if (co.reads && !fits) *co.bad = 1u;
    return used;
}

// ---------------------------------------------------------------------------------------------
// k_pair: lane = unit (workgroup t of arrival handles units [256 t, 256 t + 256): the ticket order IS the unit order, so the
// scanned offsets are the read-order layout of the records).  Units with more than PU_SEEDS seeds were chained by
// k_chain_heavy before this launch: they only take part in the scan (ncand from memory) and go on the general path's list.
//   scan counters: x = reports of the unit, y = 1 if the unit goes to the general path, z = CIGAR ops of a finished unit
// Outputs for every read: rep_off[r].  Finished units:
// rout / reports / cigar.  Other units: slow_units[] (ascending), their sorted seeds, DCand records and ncand.
// The workgroup that drew the last ticket also writes the totals (its inclusive prefix is the grand total).
// ---------------------------------------------------------------------------------------------
#include "dg_scan.h"

// the pac word a gap comparison that starts behind seed k will read first (ReadWords::mismatches8 / d_ref8: the same address on either strand); 0 for windows
// at the edges of the text, which take the per-character path
__device__ __forceinline__ uint32_t d_pac_touch(const DIndex &ix, SKey k)
{
    const int64_t L = ix.l_pac, g = sk_gpos(k) + sk_rlen(k);
    if (g >= 0 && g + 8 <= L) return *(const uint32_a1 *)(ix.pac + (g >> 2));
    if (g >= L && g + 8 <= 2 * L) return *(const uint32_a1 *)(ix.pac + ((2 * L - 1 - g - 7) >> 2));
    return 0u;
}

template <bool PACKED>      // the reads of the batch: their 2-bit + mask words (a packed batch: enc, W2) or their ASCII bytes (seq, seq_off)
__global__ void __launch_bounds__(PU_THREADS)
k_pair(const DIndex ix, const DParams pr, int n_units, int paired, int try_fast, int write_all_sorted /* bit 0: the sorted seeds and candidates of EVERY unit to memory (dg_probe_seeds); bit 1: no full record types for the units finished here (the caller takes the compact ones) */,
       const unsigned char *__restrict__ seq, const uint32_t *__restrict__ seq_off, const uint32_t *__restrict__ enc, int W2, const uint16_t *__restrict__ rlen,
       const uint32_t *__restrict__ seed_off, SKey *__restrict__ seeds, DCand *__restrict__ cands, uint32_t *__restrict__ ncand,
       uint32_t *__restrict__ rep_off, uint32_t *__restrict__ slow_units,
       dg_read_out *__restrict__ rout, dg_report_out *__restrict__ reports, uint32_t *__restrict__ cigar,
       uint32_t cap_rep, uint32_t cap_cig, TileScan ts, DSizes *sizes, unsigned int *pool_top, unsigned long long *ctr, int *err, const CompactOut co)
{
    const unsigned long long t_wave0 = wall_clock64();
    // (static LDS on purpose.  The compiler derives the kernel's occupancy from it and pads the register allocation to match -- 122 used, 169 claimed;
    //  k_chain_heavy 56 -> 257, k_reseed 44 -> 129 / 169 / 257 --; with dynamic LDS the padding goes away, which changed nothing measurable with twelve
    //  batches in flight (profiles/r04/g_variants_dynamic_lds_and_hw_queues.txt), and this kernel then faulted -- a global access at an LDS-sized
    //  address -- whenever the compact-record pointers were null (dg_map_batch); static LDS it stays)
    __shared__ SKey s_key[PU_SEEDS * PU_THREADS];
    __shared__ uint32_t s_cw[PU_SEEDS * PU_THREADS];
    __shared__ uint64_t s_rw[2 * PU_SLOTS * PU_THREADS];
    __shared__ unsigned long long s_scan[20];
    __shared__ unsigned int s_tile;
    // the chromosome table in LDS when it fits (up to 128 chromosomes; 4 KB: two workgroups still share a CU): d_loc_tab's comment says why
    __shared__ int64_t s_lkey[PU_LOC_MAX]; __shared__ int64_t s_coff[PU_LOC_MAX / 2]; __shared__ int32_t s_lchr[PU_LOC_MAX];
    const bool tab_in_lds = 2 * ix.n_chr <= PU_LOC_MAX;
    if (tab_in_lds) {
        for (int i = threadIdx.x; i < 2 * ix.n_chr; i += PU_THREADS) { s_lkey[i] = ix.loc_key[i]; s_lchr[i] = ix.loc_chr[i]; }
        for (int i = threadIdx.x; i < ix.n_chr; i += PU_THREADS) s_coff[i] = ix.chr_off[i];
    }                                                  // (d_tile_ticket's barrier below orders these stores before every use)
    const LocTab lt = tab_in_lds ? LocTab{s_lkey, s_lchr, s_coff, 2 * ix.n_chr} : d_loc_tab(ix);
    // SEEDS / SEEDQ are raised before this launch (seeds that do not fit, the seeding kernel's safety net), SCAN by k_seed_offsets before it or by a look-back
    // inside it: seed_off / nseeds cannot be trusted, the host runs the batch again.  Leaving BEFORE the ticket is safe: no tile exists that a successor could wait
    // for, and pollers of a DG_E_SCAN run give up by themselves.  The decision is thread 0's, the same for the whole workgroup (d_tile_ticket_unless).  Capacity
    // errors raised INSIDE this launch never make a workgroup leave: its successors wait for its totals
    bool leave = false;
    if (threadIdx.x == 0) { const int e0 = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); leave = e0 == DG_E_SEEDS || e0 == DG_E_SEEDQ || e0 == DG_E_SCAN; }
    const unsigned int tile = d_tile_ticket_unless(ts, &s_tile, leave);
    if (tile == SCAN_LEAVE) return;
    const int u = (int)(tile * PU_THREADS + threadIdx.x);
    const bool valid = u < n_units;
    SKey *key = s_key + threadIdx.x;
    uint32_t *cw = s_cw + threadIdx.x;
    uint64_t *rw = s_rw + threadIdx.x;
    UnitState st;
    st.fast = false; st.n_cig = 0; st.n_nw = st.n_cells = 0; st.nc[0] = st.nc[1] = 0; st.flag0[0] = st.flag0[1] = 0;
#ifdef DG_PAIR_PROF
    for (int i = 0; i < 10; i++) st.pp[i] = 0;
    st.pt = __builtin_amdgcn_s_memtime();
    const unsigned long long pp_t0 = st.pt;
#endif
    int r1 = 0, n1 = 0, n2 = 0, len1 = 0, len2 = 0;
    uint32_t b1 = 0;
    uint32_t pf = 0, pf2 = 0, pf3 = 0, pf4 = 0, pf5 = 0, pf6 = 0;
    bool heavy = false;
    if (valid) {
        r1 = paired ? 2 * u : u;
        len1 = rlen[r1]; len2 = paired ? rlen[r1 + 1] : 0;        // (asked for here, beside the seed offsets: not one more round trip behind the seeds)
        if (PACKED) { pf = enc[(size_t)r1 * 2 * W2 + (W2 >> 1)]; if (paired) pf2 = enc[(size_t)(r1 + 1) * 2 * W2 + (W2 >> 1)]; }     // touched early, see below
        b1 = seed_off[r1];
        const uint32_t e1 = seed_off[r1 + 1], e2 = paired ? seed_off[r1 + 2] : e1;
        n1 = (int)(e1 - b1); n2 = (int)(e2 - e1);
        heavy = n1 + n2 > PU_SEEDS;
        if (!heavy) {
            const SKey *src = seeds + b1;
            const int nt = n1 + n2;
            for (int i0 = 0; i0 < nt; i0 += 4) {      // four loads in flight
                const SKey k0 = src[i0], k1 = i0 + 1 < nt ? src[i0 + 1] : 0, k2 = i0 + 2 < nt ? src[i0 + 2] : 0, k3 = i0 + 3 < nt ? src[i0 + 3] : 0;
                key[i0 * PU_THREADS] = k0;
                if (i0 + 1 < nt) key[(i0 + 1) * PU_THREADS] = k1;
                if (i0 + 2 < nt) key[(i0 + 2) * PU_THREADS] = k2;
                if (i0 + 3 < nt) key[(i0 + 3) * PU_THREADS] = k3;
            }
#ifdef __HIP_DEVICE_COMPILE__
            __asm__ volatile("" ::: "memory");   // (the seeds are in LDS: nothing below may use their registers, or its wait for them would be a wait for the touches too)
#endif
            // Touch now what the reports phase will read after sorting, clustering and the candidate rules: the unit's read words (above) and the text behind the
            // first two seeds of each mate (where the gap to the next seed begins).  Those loads were the phase's dependent round trips to HBM, one per gap,
            // one after the other; touched here they are on their way while the lane works in LDS.  The values only keep the loads alive (see the end).
            if (n1 > 1) { pf3 = d_pac_touch(ix, key[0]); if (n1 > 2) pf4 = d_pac_touch(ix, key[1 * PU_THREADS]); }
            if (n2 > 1) { pf5 = d_pac_touch(ix, key[n1 * PU_THREADS]); if (n2 > 2) pf6 = d_pac_touch(ix, key[(n1 + 1) * PU_THREADS]); }
            PP_STAMP(st.pp, st.pt, 0);       // ticket, seed offsets, seeds into LDS
            if (PACKED) {
                const ReadWords a{enc + (size_t)r1 * 2 * W2, W2}, b{enc + (size_t)(r1 + (paired ? 1 : 0)) * 2 * W2, W2};
                d_unit_process_rd<PU_THREADS, ReadWords>(ix, lt, pr, paired != 0, n1, n2, len1, len2, a, b, key, cw, rw, try_fast != 0, st);
            } else d_unit_process_rd<PU_THREADS, ReadAscii>(ix, lt, pr, paired != 0, n1, n2, len1, len2, ReadAscii{seq + seq_off[r1]}, ReadAscii{seq + seq_off[r1 + (paired ? 1 : 0)]}, key, cw, rw, try_fast != 0, st);
        } else { st.nc[0] = (int)ncand[r1]; st.nc[1] = paired ? (int)ncand[r1 + 1] : 0; }
    }
    const uint32_t nrep1 = valid ? (uint32_t)(st.nc[0] > 0 ? st.nc[0] : 1) : 0u, nrep2 = (valid && paired) ? (uint32_t)(st.nc[1] > 0 ? st.nc[1] : 1) : 0u;
    Triple mine, tot;
    mine.x = nrep1 + nrep2; mine.y = (valid && !st.fast) ? 1u : 0u; mine.z = st.fast ? st.n_cig : 0u;
    uint32_t cc1 = 0;                                    // stored compact ops of mate 1 (mate 2's follow)
    mine.w = 0;
    if (co.reads && st.fast) {
        RepLds<PU_THREADS> q1{cw, rw, st.nc[0], st.flag0[0]}, q2{cw + st.nc[0] * PU_THREADS, rw, st.nc[1], st.flag0[1]};
        cc1 = d_unit_compact_ops<PU_THREADS>(st.rd[0], q1);
        mine.w = cc1 + (paired ? d_unit_compact_ops<PU_THREADS>(st.rd[1], q2) : 0u);
    }
    PP_STAMP(st.pp, st.pt, 5);               // (whatever the lane did since its last stamp: units that left d_unit_process early, the compact op count)
    const Triple inb = d_block_exclusive(mine, tot, s_scan);
    PP_STAMP(st.pp, st.pt, 6);               // the scan inside the workgroup (two barriers: the workgroup's slowest wave)
    const Triple base = d_tile_exclusive(ts, tile, tot, s_scan + 16, err);
    PP_STAMP(st.pp, st.pt, 7);               // the look-back
    const uint32_t rep0 = base.x + inb.x, slow_at = base.y + inb.y;
    const uint64_t cig0 = base.z + inb.z;
    unsigned long long n_cands = 0, n_nw = 0, n_cells = 0;
    if (valid) {
        rep_off[r1] = rep0;
        if (paired) rep_off[r1 + 1] = rep0 + nrep1;
        if (!heavy) n_cands = (unsigned long long)(st.nc[0] + st.nc[1]);
        if (st.fast) {
            if ((uint64_t)rep0 + mine.x > cap_rep) atomicMax(err, DG_E_REPORTS);
            else if (cig0 + st.n_cig > cap_cig) atomicMax(err, DG_E_CIGFINAL);
            else {
                RepLds<PU_THREADS> p1{cw, rw, st.nc[0], st.flag0[0]}, p2{cw + st.nc[0] * PU_THREADS, rw, st.nc[1], st.flag0[1]};
                const uint32_t cigc0 = base.w + inb.w;               // (stored ops never outnumber the full ones: the same capacity covers them)
#ifdef DG_EXP_EMIT2      /* measurement only: the records written twice -- what the second time costs is what the stores cost */
                d_unit_emit_read<PU_THREADS>(true, st.rd[0], p1, rep0, (uint32_t)cig0, rout + r1, reports, cigar, co, co.reads ? co.reads + r1 : nullptr, cigc0);
                if (paired) d_unit_emit_read<PU_THREADS>(false, st.rd[1], p2, rep0 + nrep1, (uint32_t)cig0 + d_unit_emit_read<PU_THREADS>(true, st.rd[0], p1, rep0, (uint32_t)cig0, rout + r1, reports, cigar, co, co.reads ? co.reads + r1 : nullptr, cigc0), rout + r1 + 1, reports, cigar, co, co.reads ? co.reads + r1 + 1 : nullptr, cigc0 + cc1);
                __asm__ volatile("" ::: "memory");
#endif
                const bool full = !(write_all_sorted & 2) || !co.reads;
                const uint32_t c1 = d_unit_emit_read<PU_THREADS>(true, st.rd[0], p1, rep0, (uint32_t)cig0, rout + r1, reports, cigar, co, co.reads ? co.reads + r1 : nullptr, cigc0, full);
                if (paired) d_unit_emit_read<PU_THREADS>(false, st.rd[1], p2, rep0 + nrep1, (uint32_t)cig0 + c1, rout + r1 + 1, reports, cigar, co, co.reads ? co.reads + r1 + 1 : nullptr, cigc0 + cc1, full);
                n_nw = st.n_nw; n_cells = st.n_cells;
            }
        } else {
            if (slow_at < (uint32_t)n_units) slow_units[slow_at] = (uint32_t)u;      // (a prefix from a look-back that gave up is garbage: the batch runs again, nothing may be written outside the list)
            if ((uint64_t)rep0 + mine.x > cap_rep) atomicMax(err, DG_E_REPORTS);       // its reports would not fit either
        }
        if (!heavy && (!st.fast || (write_all_sorted & 1))) {
            const int nt = n1 + n2;
            for (int i = 0; i < nt; i++) seeds[b1 + i] = key[i * PU_THREADS];
            auto put_cands = [&](int first_cw, int first_key, uint32_t seg, int count, int r) {     // (called with constants: st stays in registers)
                for (int i = 0; i < count; i++) {
                    const uint32_t w = cw[(first_cw + i) * PU_THREADS];
                    const int64_t d = sk_diag(key[(first_key + cw_first(w)) * PU_THREADS]);
                    DCand c = d_new_cand(seg + (uint32_t)cw_first(w), cw_count(w), cw_score(w), d < 0 ? 0 : d);
                    c.PairedIdx = cw_mate(w);
                    cands[seg + i] = c;
                }
                ncand[r] = (uint32_t)count;
            };
            put_cands(0, 0, b1, st.nc[0], r1);
            if (paired) put_cands(st.nc[0], n1, b1 + (uint32_t)n1, st.nc[1], r1 + 1);
        } else if (!heavy) { ncand[r1] = (uint32_t)st.nc[0]; if (paired) ncand[r1 + 1] = (uint32_t)st.nc[1]; }
    }
    if (tile == gridDim.x - 1 && threadIdx.x == 0) {
        sizes->total_rep = base.x + tot.x; sizes->n_slow_units = base.y + tot.y; sizes->cig_fast = (uint32_t)(base.z + tot.z);
        sizes->total_cig = sizes->cig_fast;                          // (k_emit_slow adds the general path's)
        sizes->pad[2] = base.w + tot.w;                              // stored compact ops of the finished units: the general path's follow behind them
        *pool_top = (base.x + tot.x) * CIG_SLOT;                     // the report kernel's CIGAR pool: one slot group per report, overflow area behind
    }
    PP_STAMP(st.pp, st.pt, 8);               // the records (or, for the general path, the sorted seeds and candidates) to memory
#ifdef DG_PAIR_PROF
    if ((threadIdx.x & 63) == 0 && (tile % 61u) == 0u)
        printf("kpair tile %u wave %d: total %llu | load %llu sort+cluster %llu rules %llu reports %llu settle %llu rest %llu blockscan %llu lookback %llu emit %llu\n", tile, (int)(threadIdx.x >> 6),
               __builtin_amdgcn_s_memtime() - pp_t0, st.pp[0], st.pp[1], st.pp[2], st.pp[3], st.pp[4], st.pp[5], st.pp[6], st.pp[7], st.pp[8]);
#endif
    if (n_units < 0 && (pf ^ pf2 ^ pf3 ^ pf4 ^ pf5 ^ pf6) == 0x5A17C3E1u) atomicMax(err, DG_E_SCAN);      // (never true: what keeps the touches above from being dropped)
    d_wave_add(ctr + CTR_CANDS, n_cands);
    d_wave_add(ctr + CTR_NW, n_nw);
    d_wave_add(ctr + CTR_NWCELLS, n_cells);
    d_wave_resident(ctr, CTR_WT_PAIR, t_wave0);
}
