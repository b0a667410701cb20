// Host run of FillGapsBetweenAdjacentSeeds' two forms in dg_report.h: d_gap_small (read gaps of at most PM_MAX bases: both alignments by d_pair_nw, the split
// search on bit masks over the gap's read positions) against the string form (d_gap_right_strings / d_gap_left_strings / d_gap_split_strings on the gapped strings of
// d_nw -- which tests/native/host_checks.hip pins on the oracle's nw_alignment), on random read gaps against a random text: both strand halves, windows at the strand
// boundary and at the ends of the text, gaps that continue the left seed's genome, the right seed's, both (an intron in between) or neither, with substitutions,
// insertions, deletions, lower case and N.  Compiled with hipcc, run without a GPU (no HIP API call).  Test infrastructure.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../../dart_amd/csrc/dg_common.h"
#include "../../dart_amd/csrc/dg_report.h"

static uint64_t rng_s = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static int rint(int lo, int hi) { return lo + (int)(rnd() % (uint64_t)(hi - lo + 1)); }

int main()
{
    const int64_t L = 5000;
    std::vector<uint8_t> pac(L / 4 + 64, 0);
    for (int64_t i = 0; i < L; i++) pac[i >> 2] |= (uint8_t)((rnd() & 3) << ((~i & 3) << 1));
    for (int64_t i = 1000; i < 1400; i++) { pac[i >> 2] &= (uint8_t)~(3u << ((~i & 3) << 1)); pac[i >> 2] |= (uint8_t)(((i / 3) & 1) << ((~i & 3) << 1)); }     // a low-complexity stretch: many equally good alignments
    DIndex ix; memset(&ix, 0, sizeof ix);
    ix.pac = pac.data(); ix.l_pac = L;
    DParams pr; memset(&pr, 0, sizeof pr);
    const int R = 64;
    WSLayout Lw; memset(&Lw, 0, sizeof Lw);
    Lw.nwbits_off = 0; Lw.nwbits_words = (R + 1) * ((2 * R + 32 + 7) / 8);
    Lw.rows_off = Lw.nwbits_words * 4; Lw.row_cap = R + 8;
    std::vector<unsigned char> wsbuf((size_t)Lw.nwbits_words * 4 + 2 * (R + 8) * 4 + 64);
    uint32_t lds[PM_LDS_WORDS + 1];
    LaneCtx cs; memset(&cs, 0, sizeof cs); cs.ws = wsbuf.data(); cs.L = &Lw; cs.ix = &ix; cs.pr = &pr;
    LaneCtx cn; memset(&cn, 0, sizeof cn); cn.lds = lds; cn.ix = &ix; cn.pr = &pr;
    long bad = 0, n_cases = 0, n_accept = 0, n_trail = 0, n_lead = 0, by_bp[PM_MAX + 2] = {0};
    const char *nt = "ACGT";
    for (int it = 0; it < 400000; it++) {
        const int rGaps = it % 5 == 0 ? rint(1, 4) : rint(1, PM_MAX);
        pr.max_mismatch = it % 3 == 0 ? 5 : rint(0, 8);
        // where the two windows lie: anywhere on either strand, sometimes across the strand boundary or at the ends of the text
        int64_t gR, gL;
        const int place = it % 16;
        if (place == 0) gR = rint(-3, 3); else if (place == 1) gR = L - rint(0, PM_MAX + 3); else if (place == 2) gR = 2 * L - rint(0, PM_MAX + 3); else gR = rint(0, (int)(2 * L - 1));
        if (place == 3) gL = rint(-3, 3); else if (place == 4) gL = L - rint(0, PM_MAX + 3); else if (place == 5) gL = 2 * L - rint(0, PM_MAX + 3); else if (place == 6) gL = gR + rint(0, 6); else gL = rint(0, (int)(2 * L - 1));
        // the read gap: the genome behind the left seed for a while, then the genome in front of the right seed (a junction inside the gap), or noise
        unsigned char rd[64];
        memset(rd, 'A', sizeof rd);
        const int cut = it % 7 == 0 ? rGaps : (it % 7 == 1 ? 0 : rint(0, rGaps));
        for (int k = 0; k < rGaps; k++) {
            char c = k < cut ? d_refchar(ix, gR + k) : d_refchar(ix, gL + k);
            if (c == 0 || it % 11 == 0) c = nt[rnd() & 3];
            rd[k] = (unsigned char)c;
        }
        const int n_edit = it % 4 == 0 ? 0 : rint(0, 3);
        for (int e = 0; e < n_edit; e++) {
            const int k = rint(0, rGaps - 1), what = rint(0, 3);
            if (what == 0) rd[k] = (unsigned char)nt[rnd() & 3];
            else if (what == 1) { for (int q = rGaps - 1; q > k; q--) rd[q] = rd[q - 1]; rd[k] = (unsigned char)nt[rnd() & 3]; }      // an inserted base
            else if (what == 2) { for (int q = k; q + 1 < rGaps; q++) rd[q] = rd[q + 1]; rd[rGaps - 1] = (unsigned char)nt[rnd() & 3]; }   // a deleted one
            else rd[k] = (unsigned char)(rnd() & 1 ? 'N' : (rd[k] | 0x20));
        }
        for (int k = rGaps; k < 40; k++) rd[k] = (unsigned char)"ACGTN-acgt"[rnd() % 10];       // what follows the gap in the read must not matter
        // ---- the string form ----
        char g[64], f1[160], f2[160], f3[160], f4[160];
        int Rv[PM_MAX + 2], Lv[PM_MAX + 2];
        for (int q = 0; q <= rGaps; q++) Rv[q] = Lv[q] = 0;
        d_ref_fill(ix, gR, rGaps, g);
        const int len = d_nw(cs, (const char *)rd, rGaps, g, rGaps, f1, f2);
        if (f2[len - 1] == '-') n_trail++;
        d_gap_right_strings(ix, f1, f2, len, gR + rGaps, Rv);
        d_ref_fill(ix, gL, rGaps, g);
        const int len3 = d_nw(cs, (const char *)rd, rGaps, g, rGaps, f3, f4);
        if (f4[0] == '-') n_lead++;
        d_gap_left_strings(ix, f3, f4, len3, gL, rGaps, Lv);
        int bp_s, re_s, le_s;
        d_gap_split_strings(pr, Rv, Lv, rGaps, f1, f2, f3, f4, len3, bp_s, re_s, le_s);
        // ---- the form without strings ----
        int bp_n = -1, re_n = -1, le_n = -1;
        d_gap_small(cn, rd, rGaps, gR, gL, bp_n, re_n, le_n);
        n_cases++; by_bp[bp_s]++; if (re_s || le_s) n_accept++;
        if (bp_s != bp_n || re_s != re_n || le_s != le_n) {
            if (bad < 8) printf("differs (it %d, rGaps %d, gR %ld, gL %ld): strings bp %d ext %d / %d, small bp %d ext %d / %d\n", it, rGaps, (long)gR, (long)gL, bp_s, re_s, le_s, bp_n, re_n, le_n);
            bad++;
        }
    }
    if (cs.n_nw != cn.n_nw || cs.nw_cells != cn.nw_cells) { printf("work counters differ: %llu / %llu alignments, %llu / %llu cells\n", cs.n_nw, cn.n_nw, cs.nw_cells, cn.nw_cells); bad++; }
    printf("gap filling: %ld cases, %ld with an accepted split, %ld / %ld with read bases beyond the right / left window; split points:", n_cases, n_accept, n_trail, n_lead);
    for (int q = 0; q <= PM_MAX; q++) printf(" %ld", by_bp[q]);
    printf("\nbad=%ld\n", bad);
    return bad ? 1 : 0;
}
