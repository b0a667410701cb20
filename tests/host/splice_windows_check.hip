// host-side check: d_identify_sj (windows in registers) == d_identify_sj_chars (single characters) on random texts, junctions anywhere incl. both strand halves,
// the strand boundary and the ends of the text; planted motifs so that every outcome occurs
#include <hip/hip_runtime.h>
#include "../../include/dartgpu.h"
#include "../../dart_amd/csrc/dg_common.h"
#include "../../dart_amd/csrc/dg_fm.h"
#include "../../dart_amd/csrc/dg_seedq.h"
#include "../../dart_amd/csrc/dg_chain.h"
#include "../../dart_amd/csrc/dg_report.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
int main()
{
    srand48(7);
    long checked = 0, fast = 0, bad = 0, hist[21] = {0};
    for (int rep = 0; rep < 40; rep++) {
        const int64_t L = 200 + (lrand48() % 3000);
        std::vector<uint8_t> pac((L + 3) / 4 + 16, 0);
        std::vector<int> code(L);
        for (int64_t i = 0; i < L; i++) { code[i] = (int)(lrand48() & 3); if (rep % 3 == 0 && (i % 7) < 5) code[i] = (int)((i / 7) & 3); }   // low-complexity texts too: shifts pass the fragment test
        for (int64_t i = 0; i < L; i++) pac[i >> 2] |= (uint8_t)(code[i] << ((~i & 3) << 1));
        DIndex ix; memset(&ix, 0, sizeof ix);
        ix.pac = pac.data(); ix.l_pac = L;
        for (int t = 0; t < 60000; t++) {
            DSeed l, r; memset(&l, 0, sizeof l); memset(&r, 0, sizeof r);
            const int64_t Lg = (lrand48() % (2 * L + 40)) - 20, Rg = (lrand48() % (2 * L + 40)) - 20;
            l.gLen = 1 + (int)(lrand48() % 30); l.rLen = 1 + (int)(lrand48() % 30); l.gPos = Lg - l.gLen;
            r.gLen = 1 + (int)(lrand48() % 30); r.rLen = 1 + (int)(lrand48() % 30); r.gPos = Rg;
            for (int type = 0; type < 4; type++) {
                const int a = d_identify_sj(ix, type, l, r), b = d_identify_sj_chars(ix, type, l, r);
                bool okl, okr; d_ref_codes(ix, Lg - 9, 20, &okl); d_ref_codes(ix, Rg - 11, 20, &okr);
                fast += okl && okr; checked++;
                hist[a + 10 > 20 ? 20 : a + 10]++;
                if (a != b && bad++ < 10) printf("MISMATCH L=%lld Lg=%lld Rg=%lld type=%d fast=%d chars=%d\n", (long long)L, (long long)Lg, (long long)Rg, type, a, b);
            }
        }
    }
    // d_ref_codes against d_refchar, every position and length
    long cbad = 0, cchk = 0;
    {
        const int64_t L = 517;
        std::vector<uint8_t> pac((L + 3) / 4 + 16, 0);
        for (int64_t i = 0; i < L; i++) pac[i >> 2] |= (uint8_t)((lrand48() & 3) << ((~i & 3) << 1));
        DIndex ix; memset(&ix, 0, sizeof ix); ix.pac = pac.data(); ix.l_pac = L;
        for (int64_t g0 = -40; g0 < 2 * L + 40; g0++) for (int n = 1; n <= 28; n++) {
            bool ok; const uint64_t w = d_ref_codes(ix, g0, n, &ok);
            if (!ok) continue;
            cchk++;
            for (int j = 0; j < n; j++) if ("ACGT"[(w >> (62 - 2 * j)) & 3] != d_refchar(ix, g0 + j)) { cbad++; break; }
        }
    }
    printf("checked %ld (fast path %ld), mismatches %ld; windows checked %ld, bad %ld\nresults (shift -9..9, 10 = none):", checked, fast, bad, cchk, cbad);
    for (int k = 1; k <= 20; k++) printf(" %ld", hist[k]);
    printf("\n");
    return bad || cbad ? 1 : 0;
}
