// Host run of the chain stage's lane code (dg_chain.h: sort, GenerateAlignmentCandidate, mate pairing, redundancy filter) on
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
    while (scanf(" %c", &tag) == 1 && tag == 'U') {
        std::vector<DSeed> s[2]; std::vector<DCand> c[2]; int rl[2] = {0, 0}, nc[2] = {0, 0};
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            int n;
            if (scanf("%d %d", &rl[m], &n) != 2) return 2;
            s[m].resize(n + 1); c[m].resize(n + 1);
            for (int i = 0; i < n; i++) {
                long long g; int r, l;
                if (scanf("%d %d %lld", &r, &l, &g) != 3) return 2;
                s[m][i].gPos = g; s[m][i].rPos = r; s[m][i].rLen = s[m][i].gLen = l; s[m][i].flags = SEED_SIMPLE;
            }
            for (int i = n - 1; i > 0; i--) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; std::swap(s[m][i], s[m][rng % (i + 1)]); }   // the sort has work to do
            d_sort_seeds(s[m].data(), n);
            nc[m] = d_gen_candidates(ix, pr, rl[m], s[m].data(), n, 0u, c[m].data());
        }
        if (paired) {
            if (d_check_paired(c[0].data(), nc[0], c[1].data(), nc[1])) d_remove_unmated(c[0].data(), nc[0], c[1].data(), nc[1]);
            d_remove_redundant(c[0].data(), nc[0]); d_remove_redundant(c[1].data(), nc[1]);
        } else d_remove_redundant(c[0].data(), nc[0]);
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            printf("C%d %d", m + 1, nc[m]);
            for (int i = 0; i < nc[m]; i++) printf(" %d:%lld:%d:%d", c[m][i].Score, (long long)c[m][i].PosDiff, c[m][i].PairedIdx, c[m][i].count);
            printf("\n");
        }
    }
    return 0;
}
