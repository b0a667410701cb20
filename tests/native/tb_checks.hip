// Host run of the traceback consumers of dg_report.h that walk the nw_alignment bits (TbWalk: a window of the column-major bits in the lane's LDS slice, the two
// sequences as 8-character words) against the gapped-string forms they replace:
//   d_process_pair_tb                       against d_process_pair's string path on the same bits (d_tb_traceback, CheckLocalAlignmentQuality, the head / tail
//                                           trimming, AddNewCigarElements): CIGAR runs, score, the trimmed segment pair, in the three modes
//   d_gap_right_tb / d_gap_left_tb / d_gap_split_tb   against d_gap_right_strings / d_gap_left_strings / d_gap_split_strings (FillGapsBetweenAdjacentSeeds on wide read gaps)
// The bits come from a scalar restatement of the cell recurrence in the column-major layout of d_nw_group / d_nw_coop (realistic paths: substitutions, indels, ends
// that do not align) and, for a third of the cases, from a random source (any bit matrix is a valid traceback: every flag moves up, left or both).  Both strands, the
// strand boundary, the ends of the text.  Compiled with hipcc, run without a GPU (no HIP API call).  Test infrastructure.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../../dart_amd/csrc/dg_common.h"
#include "../../dart_amd/csrc/dg_report.h"

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static int rint(int lo, int hi) { return lo + (int)(rnd() % (uint64_t)(hi - lo + 1)); }

// nw_alignment's cells (the x2 integers of d_nw) -> 2 bits per cell, column-major: tb[(j-1) * RW + (i-1)/16]
static void fill_bits(const DIndex &ix, const unsigned char *a, int m, int64_t gpos, int n, uint32_t *tb, bool random_source)
{
    const int RW = (m + 15) >> 4;
    for (int k = 0; k < RW * n; k++) tb[k] = 0;
    if (random_source) {
        const int bias = rint(0, 2);
        for (int j = 1; j <= n; j++) for (int i = 1; i <= m; i++) {
            uint32_t fl = bias == 0 ? (uint32_t)(rnd() & 3) : ((rnd() % 10) < (bias == 1 ? 8u : 6u) ? 0u : (uint32_t)(1 + rnd() % 3));
            tb[(size_t)(j - 1) * RW + ((i - 1) >> 4)] |= fl << (((i - 1) & 15) << 1);
        }
        return;
    }
    std::vector<int> S((size_t)(m + 1) * (n + 1)), Rr((size_t)(m + 1) * (n + 1)), Tt((size_t)(m + 1) * (n + 1));
    auto at = [&](int i, int j) { return (size_t)i * (n + 1) + j; };
    S[at(0, 0)] = 0;
    for (int j = 1; j <= n; j++) { S[at(0, j)] = -2 - j; Tt[at(0, j)] = -131072; }
    for (int i = 1; i <= m; i++) { S[at(i, 0)] = -2 - i; Rr[at(i, 0)] = -131072; }
    for (int i = 1; i <= m; i++) for (int j = 1; j <= n; j++) {
        int x = Rr[at(i, j - 1)] - 1, y = S[at(i, j - 1)] - 3;
        const int r = x > y ? x : y;
        x = Tt[at(i - 1, j)] - 1; y = S[at(i - 1, j)] - 3;
        const int t = x > y ? x : y;
        const int d = S[at(i - 1, j - 1)] + (d_nt4(a[i - 1]) == d_nt4((unsigned char)d_refchar(ix, gpos + j - 1)) ? 3 : -3);
        const int sv = d_tr2(d_max3(d, r, t));
        S[at(i, j)] = sv; Rr[at(i, j)] = r; Tt[at(i, j)] = t;
        tb[(size_t)(j - 1) * RW + ((i - 1) >> 4)] |= ((sv == r ? 1u : 0u) | (sv == t ? 2u : 0u)) << (((i - 1) & 15) << 1);
    }
}

int main()
{
    const int64_t L = 6000;
    std::vector<uint8_t> pac(L / 4 + 64, 0);
    for (int64_t i = 0; i < L; i++) pac[i >> 2] |= (uint8_t)((rnd() & 3) << ((~i & 3) << 1));
    for (int64_t i = 2000; i < 2600; i++) { pac[i >> 2] &= (uint8_t)~(3u << ((~i & 3) << 1)); pac[i >> 2] |= (uint8_t)(((i / 2) & 1) << ((~i & 3) << 1)); }   // low complexity: ties everywhere
    DIndex ix; memset(&ix, 0, sizeof ix);
    ix.pac = pac.data(); ix.l_pac = L;
    DParams pr; memset(&pr, 0, sizeof pr);
    const int R = 300;
    WSLayout Lw; memset(&Lw, 0, sizeof Lw);
    uint32_t o = 0;
    Lw.cig_off = o; Lw.cig_cap = 4u * R + 64u; o += Lw.cig_cap * 4u;
    Lw.nwbits_off = o; Lw.nwbits_words = ((R + 15) / 16) * (2 * R + 64); o += Lw.nwbits_words * 4u;
    Lw.rows_off = o; Lw.row_cap = R + 8; o += 2u * Lw.row_cap * 4u;
    Lw.str_off = o; Lw.str_cap = 3u * R + 64u; o += 6u * Lw.str_cap;
    std::vector<unsigned char> ws(o + 64);
    uint32_t lds[PM_LDS_WORDS + 1];
    std::vector<unsigned char> seq(1024, 'A');
    LaneCtx cs; memset(&cs, 0, sizeof cs); cs.ws = ws.data(); cs.L = &Lw; cs.ix = &ix; cs.pr = &pr; cs.seq = seq.data(); cs.rlen = 600; cs.lds = nullptr;       // no LDS slice: the string forms
    LaneCtx cn = cs; cn.lds = lds;
    uint32_t *tb = (uint32_t *)(ws.data() + Lw.nwbits_off);
    long bad = 0, n_pair = 0, n_skipped = 0, by_mode[3] = {0, 0, 0}, n_quality_fail = 0, n_head_trim = 0, n_tail_trim = 0, n_gap = 0, n_gap_accept = 0, n_trail = 0, n_lead = 0;
    const char *nt = "ACGT";
    auto place = [&](int span) -> int64_t {
        const int k = rint(0, 15);
        if (k == 0) return rint(-3, 3);
        if (k == 1) return L - rint(0, span + 3);
        if (k == 2) return 2 * L - rint(0, span + 3);
        return rint(0, (int)(2 * L - 1));
    };
    auto make_read = [&](unsigned char *rd, int m, int64_t g, int n) {       // the genome segment with edits (few, some, many), or noise
        const int noise = rint(0, 19) == 0, span = (const int[]){2000, 400, 400, 80, 29}[rint(0, 4)];
        int gi = rint(0, 5) == 0 ? rint(0, 3) : 0;                            // sometimes the read starts a few genome bases in
        for (int k = 0; k < m; k++) {
            const int e = rint(0, span);
            if (e == 0 && gi + 1 < n) gi++;                                   // a deleted genome base
            char c = (e == 1 || noise) ? nt[rnd() & 3] : d_refchar(ix, g + (gi < n ? gi : n - 1));
            if (c == 0) c = nt[rnd() & 3];
            if (e != 2) gi++;                                                 // (e == 2: an inserted read base)
            if (e == 3) c = 'N'; else if (e == 4) c = (char)(c | 0x20);
            rd[k] = (unsigned char)c;
        }
    };
    // ---- A: segment pairs ----
    for (int it = 0; it < 60000; it++) {
        const int m = it % 3 == 0 ? rint(1, 40) : rint(20, 160), dn = rint(-8, 8), n = it % 5 == 0 ? rint(1, 200) : (m + dn < 1 ? 1 : m + dn);
        const int mode = it % 3;
        const int64_t g = place(n);
        const int rpos = rint(0, 200);
        unsigned char *rd = seq.data() + rpos;
        make_read(rd, m, g, n);
        for (int k = m; k < m + 24; k++) rd[k] = (unsigned char)"ACGTN-acgt"[rnd() % 10];
        DSeed s1; s1.flags = 0; s1.rPos = rpos; s1.rLen = m; s1.gPos = g; s1.gLen = n;
        if (!d_big_needs_nw(cs, s1, mode)) { n_skipped++; continue; }
        fill_bits(ix, rd, m, g, n, tb, (it / 3) % 4 == 1);
        if ((it / 3) % 4 == 2) {
            // a path of at most three runs laid over the bits -- one run of columns with both bases (the read made equal to the genome there, but for a few bases),
            // the unmatched genome bases as one run, the unmatched read bases as one run, in any order: what the head and tail trimming is about
            const int RW = (m + 15) >> 4, k = rint(0, m < n ? m : n);
            int order[3] = {0, 1, 2};                                             // 0 = both bases, 1 = genome only (D), 2 = read only (I); from the END of the alignment
            for (int q = 2; q > 0; q--) { const int r = rint(0, q), t = order[q]; order[q] = order[r]; order[r] = t; }
            int i = m, j = n;
            for (int q = 0; q < 3; q++) {
                const int cnt = order[q] == 0 ? k : (order[q] == 1 ? n - k : m - k);
                for (int c = 0; c < cnt; c++) {
                    if (i > 0 && j > 0) {
                        uint32_t &wd = tb[(size_t)(j - 1) * RW + ((i - 1) >> 4)];
                        const int sh = ((i - 1) & 15) << 1;
                        wd = (wd & ~(3u << sh)) | ((order[q] == 0 ? 0u : (order[q] == 1 ? (rnd() & 1 ? 1u : 3u) : 2u)) << sh);
                    }
                    if (order[q] == 0) { const char gc = d_refchar(ix, g + j - 1); if (gc && rint(0, 30) != 0) rd[i - 1] = (unsigned char)gc; i--; j--; }
                    else if (order[q] == 1) j--; else i--;
                }
            }
            if (!d_big_needs_nw(cs, s1, mode)) { n_skipped++; continue; }
        }
        DSeed s2 = s1;
        uint32_t c1[1400], c2[1400];
        int nc1 = rint(0, 3), nc2 = nc1;
        for (int k = 0; k < nc1; k++) c1[k] = c2[k] = CIG(7 + k, OP_M);
        const int sc1 = d_process_pair(cs, s1, mode, c1, nc1, true);
        const int sc2 = d_process_pair_tb(cn, s2, mode, c2, nc2);
        n_pair++; by_mode[mode]++;
        bool same = sc1 == sc2 && nc1 == nc2 && s1.rPos == s2.rPos && s1.rLen == s2.rLen && s1.gPos == s2.gPos && s1.gLen == s2.gLen;
        for (int k = 0; same && k < nc1; k++) same = c1[k] == c2[k];
        if (mode != 2 && sc1 == 0 && nc1 > 0 && (c1[nc1 - 1] & 15u) == OP_S && (int)(c1[nc1 - 1] >> 4) == m) n_quality_fail++;
        if (mode == 0 && (s1.rPos != rpos || s1.gPos != g)) n_head_trim++;
        if (mode == 1 && (s1.rLen != m || s1.gLen != n)) n_tail_trim++;
        if (!same) { if (bad < 8) printf("pair differs (it %d, mode %d, %d x %d): score %d / %d, ops %d / %d\n", it, mode, m, n, sc1, sc2, nc1, nc2); bad++; }
    }
    // ---- B: wide read gaps ----
    for (int it = 0; it < 30000; it++) {
        const int m = it % 4 == 0 ? rint(PM_MAX + 1, 263) : rint(PM_MAX + 1, 90);
        pr.max_mismatch = it % 3 == 0 ? 5 : rint(0, 12);
        const int64_t gR = place(m), gL = it % 6 == 0 ? gR + rint(0, 8) : place(m);
        unsigned char *rd = seq.data() + rint(0, 100);
        const int cut = it % 7 == 0 ? m : (it % 7 == 1 ? 0 : rint(0, m));
        make_read(rd, cut, gR, cut > 0 ? cut : 1);
        if (cut < m) make_read(rd + cut, m - cut, gL + cut, m - cut);
        for (int k = m; k < m + 24; k++) rd[k] = (unsigned char)"ACGTN-acgt"[rnd() % 10];
        char *g = ws_str(cs, 0), *f1 = ws_str(cs, 1), *f2 = ws_str(cs, 2), *f3 = ws_str(cs, 3), *f4 = ws_str(cs, 4);
        std::vector<int> Rv(m + 1, 0), Lv(m + 1, 0);
        std::vector<uint32_t> RJ(m + 1, 0xDEADBEEFu), LJ(m + 1, 0xDEADBEEFu);
        const bool rnd_src = (it / 3) % 4 == 1;
        fill_bits(ix, rd, m, gR, m, tb, rnd_src);
        d_ref_fill(ix, gR, m, g);
        const int len = d_tb_traceback(cs, (const char *)rd, m, g, m, f1, f2);
        if (f2[len - 1] == '-') n_trail++;
        d_gap_right_strings(ix, f1, f2, len, gR + m, Rv.data());
        d_gap_right_tb(cn, rd, m, gR, RJ.data());
        fill_bits(ix, rd, m, gL, m, tb, rnd_src);
        d_ref_fill(ix, gL, m, g);
        const int len3 = d_tb_traceback(cs, (const char *)rd, m, g, m, f3, f4);
        if (f4[0] == '-') n_lead++;
        d_gap_left_strings(ix, f3, f4, len3, gL, m, Lv.data());
        d_gap_left_tb(cn, rd, m, gL, LJ.data());
        int bp_s, re_s, le_s, bp_n, re_n, le_n;
        d_gap_split_strings(pr, Rv.data(), Lv.data(), m, f1, f2, f3, f4, len3, bp_s, re_s, le_s);
        d_gap_split_tb(pr, RJ.data(), LJ.data(), m, bp_n, re_n, le_n);
        n_gap++; if (re_s || le_s) n_gap_accept++;
        bool same = bp_s == bp_n && re_s == re_n && le_s == le_n;
        for (int q = 0; same && q <= m; q++) same = Rv[q] == (int)(RJ[q] & 0xFFFFu) && Lv[q] == (int)(LJ[q] & 0xFFFFu);
        if (!same) { if (bad < 8) printf("gap differs (it %d, m %d): strings bp %d ext %d / %d, bits bp %d ext %d / %d\n", it, m, bp_s, re_s, le_s, bp_n, re_n, le_n); bad++; }
    }
    if (cs.n_nw != cn.n_nw || cs.nw_cells != cn.nw_cells) { printf("work counters differ: %llu / %llu alignments, %llu / %llu cells\n", cs.n_nw, cn.n_nw, cs.nw_cells, cn.nw_cells); bad++; }
    printf("pairs: %ld compared (%ld head, %ld tail, %ld normal; %ld skipped: no alignment needed), %ld failed the local quality check, %ld head-trimmed, %ld tail-trimmed\n",
           n_pair, by_mode[0], by_mode[1], by_mode[2], n_skipped, n_quality_fail, n_head_trim, n_tail_trim);
    printf("wide gaps: %ld compared, %ld with an accepted split, %ld / %ld with read bases beyond the right / left window\n", n_gap, n_gap_accept, n_trail, n_lead);
    printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
