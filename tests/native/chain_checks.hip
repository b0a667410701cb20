// Host run of the chain stage's code (dg_chain.h / dg_pair.h: sort, GenerateAlignmentCandidate, mate pairing, redundancy filter;
// the in-memory form of k_chain_heavy AND the per-lane packed form of k_pair) on
// the seeds of the REFERENCE's stage dumps; prints the candidates in the dump's own format so that the test can compare
// them with the reference's C1/C2 lines.  Compiled with hipcc, run without a GPU (no HIP API call).  Test infrastructure.
//   input (stdin): "H n_chr l_pac max_gaps max_intron paired" then n_chr lines "chr_off chr_len", then per unit
//   "U rlen1 n1 (rPos len gPos)*n1 [rlen2 n2 (rPos len gPos)*n2]"
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../dart_amd/csrc/dg_common.h"
#include "../../dart_amd/csrc/dg_chain.h"
#include "../../dart_amd/csrc/dg_pair.h"

int main()
{
    int n_chr, paired; long long l_pac; DParams pr; memset(&pr, 0, sizeof pr);
    if (scanf(" H %d %lld %d %d %d", &n_chr, &l_pac, &pr.max_gaps, &pr.max_intron, &paired) != 5) return 2;
    std::vector<int64_t> key(2 * n_chr), off(n_chr);
    std::vector<int32_t> chr(2 * n_chr);
    for (int i = 0; i < n_chr; i++) {
        long long o, l;
        if (scanf("%lld %lld", &o, &l) != 2) return 2;
        off[i] = o; key[i] = o + l - 1; chr[i] = i;                                       // as dg_init builds ChrLocMap (bwt_index.cpp:249-250)
        key[2 * n_chr - 1 - i] = 2 * l_pac - o - 1; chr[2 * n_chr - 1 - i] = i;
    }
    DIndex ix; memset(&ix, 0, sizeof ix);
    ix.l_pac = l_pac; ix.n_chr = n_chr; ix.loc_key = key.data(); ix.loc_chr = chr.data(); ix.chr_off = off.data();
    char tag;
    uint64_t rng = 88172645463325252ull;
    long both = 0;
    while (scanf(" %c", &tag) == 1 && tag == 'U') {
        std::vector<SKey> s[2]; std::vector<DCand> c[2]; int rl[2] = {0, 0}, nc[2] = {0, 0}, ns[2] = {0, 0};
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            int n;
            if (scanf("%d %d", &rl[m], &n) != 2) return 2;
            ns[m] = n;
            s[m].resize(n + 1); c[m].resize(n + 1);
            for (int i = 0; i < n; i++) {
                long long g; int r, l;
                if (scanf("%d %d %lld", &r, &l, &g) != 3) return 2;
                s[m][i] = sk_make(g, r, l);
            }
            for (int i = n - 1; i > 0; i--) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; std::swap(s[m][i], s[m][rng % (i + 1)]); }   // the sort has work to do
        }
        // (b) the per-lane form of k_pair (units whose seeds fit a lane's slice), on a copy of the unsorted seeds
        SKey lk[PU_SEEDS]; uint32_t lcw[PU_SEEDS]; uint64_t lrw[2 * PU_SLOTS];
        UnitState st; st.nc[0] = st.nc[1] = 0;
        const bool tiny = ns[0] + ns[1] <= PU_SEEDS;
        if (tiny) {
            for (int i = 0; i < ns[0]; i++) lk[i] = s[0][i];
            for (int i = 0; i < ns[1]; i++) lk[ns[0] + i] = s[1][i];
            d_unit_process<1>(ix, pr, paired != 0, ns[0], ns[1], rl[0], rl[1], nullptr, nullptr, lk, lcw, lrw, false, st);
        }
        // (a) the in-memory form (k_chain_heavy's serial route)
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            d_sort_keys(s[m].data(), ns[m]);
            nc[m] = d_gen_candidates(ix, pr, rl[m], s[m].data(), ns[m], 0u, c[m].data());
        }
        CandMem a{c[0].data(), nc[0]}, b{c[1].data(), nc[1]};
        d_candidate_rules(paired != 0, a, b);
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            printf("C%d %d", m + 1, nc[m]);
            for (int i = 0; i < nc[m]; i++) printf(" %d:%lld:%d:%d", c[m][i].Score, (long long)c[m][i].PosDiff, c[m][i].PairedIdx, c[m][i].count);
            printf("\n");
        }
        if (tiny) {                                   // both forms must agree field by field (the test compares form (a) with the reference)
            both++;
            for (int m = 0, q = 0; m < (paired ? 2 : 1); m++) {
                if (st.nc[m] != nc[m]) { printf("MISMATCH candidate count\n"); return 0; }
                for (int i = 0; i < nc[m]; i++, q++) {
                    const uint32_t w = lcw[q];
                    const int64_t d = sk_diag(lk[(m ? ns[0] : 0) + cw_first(w)]);
                    if (cw_score(w) != c[m][i].Score || cw_mate(w) != c[m][i].PairedIdx || cw_count(w) != c[m][i].count || (d < 0 ? 0 : d) != c[m][i].PosDiff ||
                        cw_first(w) != c[m][i].first) { printf("MISMATCH lane form vs memory form\n"); return 0; }
                }
                for (int i = 0; i < ns[m]; i++) if (lk[(m ? ns[0] : 0) + i] != s[m][i]) { printf("MISMATCH sort\n"); return 0; }
            }
        }
    }
    fprintf(stderr, "units through both forms: %ld\n", both);
    return 0;
}
