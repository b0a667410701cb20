// Host run of the list passes of the report stage (dg_report.h: d_untangle_seeds = RemoveTandemRepeatSeeds + RemoveTranslocatedSeeds,
// d_identify_normal_pairs with d_trim_overlaps / d_resolve_overlap = IdentifyNormalPairs + CheckOverlappingSeeds +
// CheckSeedOverlapping, d_check_splice = CheckSpliceJunction) against the oracle's restatement of the same reference functions
// (orc_seed_stage) on random seed lists.  Compiled with hipcc, run without a GPU (no HIP API call).  Test infrastructure.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../dart_amd/csrc/dg_common.h"
#include "../../dart_amd/csrc/dg_report.h"
#include "../../oracle/dart_oracle.h"

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static int rint(int lo, int hi) { return lo + (int)(rnd() % (uint64_t)(hi - lo + 1)); }

struct Lists { std::vector<int32_t> rpos, rlen, glen; std::vector<int64_t> gpos; std::vector<uint32_t> flags; };

static bool same(const DSeed *s, int n, const Lists &o, int on)
{
    if (n != on) return false;
    for (int i = 0; i < n; i++)
        if (s[i].rPos != o.rpos[i] || s[i].rLen != o.rlen[i] || s[i].gLen != o.glen[i] || s[i].gPos != o.gpos[i] || s[i].flags != o.flags[i]) return false;
    return true;
}

int main()
{
    // a small two-chromosome text for the splice pass
    const int64_t L = 40000;
    std::vector<uint8_t> pac(L / 4 + 8, 0), codes(L);
    for (int64_t i = 0; i < L; i++) codes[i] = (uint8_t)(rnd() & 3);
    int64_t chr_off[2] = {0, 25000}, chr_len[2] = {25000, 15000};
    int64_t loc_key[4] = {24999, 39999, 2 * L - 25000 - 1, 2 * L - 1};
    int loc_chr_o[4] = {0, 1, 1, 0}; int32_t loc_chr_d[4] = {0, 1, 1, 0};
    orc_params pr; orc_params_default(&pr);
    DParams dp; dp.max_gaps = pr.max_gaps; dp.max_dup = pr.max_dup; dp.max_intron = pr.max_intron; dp.min_intron = pr.min_intron; dp.max_mismatch = 5; dp.multi_hit = 0; dp.all_sj = 0; dp.paired = 0;
    long bad = 0, n_untangle = 0, n_changed = 0, n_pairs = 0, n_splice = 0, n_sj = 0;
    const int CAP = 4096;
    std::vector<DSeed> s(CAP);
    std::vector<uint32_t> scratch(CAP * 4);
    Lists o; o.rpos.resize(CAP); o.rlen.resize(CAP); o.glen.resize(CAP); o.gpos.resize(CAP); o.flags.resize(CAP);
    auto to_oracle = [&](int n) { for (int i = 0; i < n; i++) { o.rpos[i] = s[i].rPos; o.rlen[i] = s[i].rLen; o.glen[i] = s[i].gLen; o.gpos[i] = s[i].gPos; o.flags[i] = s[i].flags; } };
    orc_index oix; memset(&oix, 0, sizeof oix);
    oix.pac = pac.data(); oix.l_pac = L; oix.n_chr = 2; oix.chr_off = chr_off; oix.chr_len = chr_len; oix.loc_key = loc_key; oix.loc_chr = loc_chr_o;
    DIndex dix; memset(&dix, 0, sizeof dix);
    dix.pac = pac.data(); dix.l_pac = L; dix.n_chr = 2; dix.loc_key = loc_key; dix.loc_chr = loc_chr_d; dix.chr_off = chr_off;

    // ---- pass 0: tandem + translocation clean-up (lists in genome order, exact seeds, read positions anywhere) ----
    for (int it = 0; it < 300000; it++) {
        const int rlen = rint(40, 1000), n = it % 50 == 0 ? rint(30, 400) : rint(1, 12);
        int64_t g = rint(0, 1000000);
        const int style = it % 4;
        int rp = rint(0, 30);
        for (int i = 0; i < n; i++) {
            g += rint(0, style == 3 ? 5 : 400);
            if (style == 0) rp = rint(0, rlen - 16);                       // anywhere
            else if (style == 1) { rp += rint(0, 60); if (rnd() % 5 == 0) rp = rint(0, rlen - 16); }     // mostly rising, sometimes a jump back, sometimes equal
            else if (style == 2) rp = (rnd() % 3 == 0) ? rp : rint(0, rlen - 16);   // repeats of the same read position
            else rp += rint(1, 40);                                        // strictly rising: the shortcut
            if (rp > rlen - 16) rp = rlen - 16;
            s[i].gPos = g; s[i].rPos = rp; s[i].rLen = s[i].gLen = rint(16, 60); s[i].flags = SEED_SIMPLE;
        }
        std::stable_sort(s.begin(), s.begin() + n, [](const DSeed &a, const DSeed &b) { return d_seed_less(a, b); });
        to_oracle(n);
        int on = n;
        orc_seed_stage(&oix, &pr, 0, &on, CAP, o.rpos.data(), o.rlen.data(), o.glen.data(), o.gpos.data(), o.flags.data());
        const int dn = d_untangle_seeds(s.data(), n, rlen, scratch.data());
        n_untangle++; n_changed += dn != n;
        if (!same(s.data(), dn, o, on)) { if (bad < 5) printf("untangle differs (it %d, n %d -> %d vs %d)\n", it, n, dn, on); bad++; }
    }
    // ---- pass 1: overlap trimming + normal pairs (exact seeds with read/genome overlaps, some non-exact pairs in between) ----
    for (int it = 0; it < 300000; it++) {
        const int n = it % 40 == 0 ? rint(20, 200) : rint(1, 9);
        int64_t g = rint(100, 100000);
        int rp = rint(0, 20);
        for (int i = 0; i < n; i++) {
            const int len = rint(1, 50);
            s[i].gPos = g; s[i].rPos = rp; s[i].rLen = len; s[i].gLen = len; s[i].flags = SEED_SIMPLE;
            if (rnd() % 9 == 0) { s[i].flags = 0; s[i].gLen = rint(0, 60); if (rnd() % 3 == 0) s[i].rLen = rint(0, 10); }    // what SeedExtension leaves behind
            const int kind = (int)(rnd() % 8);
            int dr = len + rint(0, 30), dg = dr + rint(-3, 3);
            if (kind == 0) { dr = len - rint(1, len); dg = dr; }                          // overlap on both
            else if (kind == 1) { dr = len - rint(0, len); dg = len + rint(0, 40); }      // overlap on the read only
            else if (kind == 2) { dg = len - rint(1, len); dr = len + rint(0, 20); }      // overlap on the genome only
            else if (kind == 3) { dg = dr + rint(31, 5000); }                             // an intron-sized genome gap
            else if (kind == 4) { dr = len; dg = len + rint(0, 80); }                     // touching on the read
            rp += dr; g += dg < 0 ? 0 : dg;
        }
        std::stable_sort(s.begin(), s.begin() + n, [](const DSeed &a, const DSeed &b) { return d_seed_less(a, b); });
        to_oracle(n);
        int on = n;
        orc_seed_stage(&oix, &pr, 1, &on, CAP, o.rpos.data(), o.rlen.data(), o.glen.data(), o.gpos.data(), o.flags.data());
        const int dn = d_identify_normal_pairs(s.data(), n);
        n_pairs++;
        if (!same(s.data(), dn, o, on)) { if (bad < 5) printf("normal pairs differ (it %d, n %d -> %d vs %d)\n", it, n, dn, on); bad++; }
    }
    // ---- pass 2: splice junctions (motifs planted at shifted boundaries, on both strands, some absent) ----
    LaneCtx cx; memset(&cx, 0, sizeof cx); cx.ix = &dix; cx.pr = &dp;
    std::vector<int2> vec(CAP);
    static const char *motif[4] = {"GTAG", "CTAC", "GCAG", "CTGC"};
    auto put = [&](int64_t gpos2, char ch) {            // writes text position gpos2 (either half) into the forward codes
        const int code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
        if (gpos2 < 0 || gpos2 >= 2 * L) return;
        if (gpos2 < L) codes[gpos2] = (uint8_t)code; else codes[2 * L - 1 - gpos2] = (uint8_t)(3 - code);
    };
    for (int it = 0; it < 60000; it++) {
        const int n = rint(2, 6);
        const bool rev = it & 1;
        int64_t g = (rev ? L : 0) + rint(50, 3000);
        int rp = 0;
        for (int i = 0; i < n; i++) {
            const int len = rint(10, 40);
            s[i].gPos = g; s[i].rPos = rp; s[i].rLen = s[i].gLen = len; s[i].flags = (rnd() % 12 == 0) ? 0u : SEED_SIMPLE;
            rp += len;
            const int64_t intron = (rnd() % 5 == 0) ? rint(0, 5) : rint(20, 3000);
            if (i + 1 < n && intron > 5 && rnd() % 4 != 0) {                 // plant a motif at a shifted boundary
                const int t = (int)(rnd() % 4), sh = rint(-9, 9);
                const int64_t Lg = g + len, Rg = g + len + intron;
                put(Lg + sh, motif[t][0]); put(Lg + sh + 1, motif[t][1]); put(Rg - 2 + sh, motif[t][2]); put(Rg - 1 + sh, motif[t][3]);
                // the shifted bases must be identical on both sides for the shift to be allowed: copy them
                if (sh > 0) for (int q = 0; q < sh; q++) { const int64_t a = Lg + q, b = Rg + q; if (a >= 0 && a < 2 * L && b < 2 * L) put(b, "ACGT"[a < L ? codes[a] : 3 - codes[2 * L - 1 - a]]); }
                if (sh < 0) for (int q = 1; q <= -sh; q++) { const int64_t a = Lg - q, b = Rg - q; if (a >= 0 && a < 2 * L && b < 2 * L) put(a, "ACGT"[b < L ? codes[b] : 3 - codes[2 * L - 1 - b]]); }
            }
            g += len + intron;
        }
        for (int64_t i = 0; i < (int64_t)pac.size(); i++) pac[i] = 0;
        for (int64_t i = 0; i < L; i++) pac[i >> 2] |= (uint8_t)(codes[i] << ((~i & 3) << 1));
        to_oracle(n);
        int on = n;
        const int otype = orc_seed_stage(&oix, &pr, 2, &on, CAP, o.rpos.data(), o.rlen.data(), o.glen.data(), o.gpos.data(), o.flags.data());
        const int dtype = d_check_splice(cx, s.data(), n, vec.data());
        n_splice++; n_sj += dtype >= 0;
        if (dtype != otype || !same(s.data(), n, o, on)) { if (bad < 5) printf("splice differs (it %d): type %d vs %d\n", it, dtype, otype); bad++; }
    }
    printf("untangle: %ld lists (%ld changed), normal pairs: %ld lists, splice: %ld lists (%ld with a junction type)  bad=%ld\n", n_untangle, n_changed, n_pairs, n_splice, n_sj, bad);
    return bad ? 1 : 0;
}
