// Host run of k_pair's per-unit code (dg_pair.h: d_unit_process<1> = sort, candidates, mate pairing, redundancy filter, the
// "[S] M [S]" report, pair settling, FLAG, MAPQ; d_unit_emit_read = the records) on the oracle's seeds, compared with the
// oracle's records for every unit the code finishes itself; units it hands to the general path are only counted.
// Compiled with hipcc, run without a GPU (no HIP API call).  Test infrastructure.
// input: a binary file written by tests/test_kernel_arith_host.py (layout: see read order below)
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../../include/dartgpu.h"
#include "../../dart_amd/csrc/dg_common.h"
#include "../../dart_amd/csrc/dg_pair.h"

template <typename T> static std::vector<T> rd(FILE *f, size_t n) { std::vector<T> v(n + 8); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[16];
    if (fread(hdr, 4, 16, f) != 16) return 2;
    const int n_chr = hdr[0], paired = hdr[1], n_reads = hdr[2];
    DParams pr; pr.max_gaps = hdr[3]; pr.max_dup = hdr[4]; pr.max_intron = hdr[5]; pr.min_intron = hdr[6]; pr.max_mismatch = hdr[7]; pr.multi_hit = hdr[8]; pr.all_sj = hdr[9]; pr.paired = paired;
    const int64_t l_pac = ((int64_t)hdr[11] << 32) | (uint32_t)hdr[10];
    const size_t n_seeds = (uint32_t)hdr[12], n_rep = (uint32_t)hdr[13], n_cig = (uint32_t)hdr[14], seq_bytes = (uint32_t)hdr[15];
    auto chr_off = rd<int64_t>(f, n_chr), chr_len = rd<int64_t>(f, n_chr);
    auto pac = rd<uint8_t>(f, (size_t)(l_pac / 4 + 1));
    auto seq_off = rd<uint32_t>(f, n_reads); auto rlen = rd<uint16_t>(f, n_reads); auto seq = rd<uint8_t>(f, seq_bytes);
    auto seed_off = rd<uint32_t>(f, (size_t)n_reads + 1); auto rpos = rd<int32_t>(f, n_seeds); auto slen = rd<int32_t>(f, n_seeds); auto gpos = rd<int64_t>(f, n_seeds);
    auto e_reads = rd<dg_read_out>(f, n_reads); auto e_rep = rd<dg_report_out>(f, n_rep); auto e_cig = rd<uint32_t>(f, n_cig);
    fclose(f);
    pac.resize(pac.size() + 64, 0);
    std::vector<int64_t> key(2 * n_chr); std::vector<int32_t> chr(2 * n_chr);
    for (int i = 0; i < n_chr; i++) {
        key[i] = chr_off[i] + chr_len[i] - 1; chr[i] = i;
        key[2 * n_chr - 1 - i] = 2 * l_pac - chr_off[i] - 1; chr[2 * n_chr - 1 - i] = i;
    }
    DIndex ix; memset(&ix, 0, sizeof ix);
    ix.pac = pac.data(); ix.l_pac = l_pac; ix.n_chr = n_chr; ix.loc_key = key.data(); ix.loc_chr = chr.data(); ix.chr_off = chr_off.data();
    const int nm = paired ? 2 : 1, n_units = n_reads / nm;
    long n_fast = 0, n_slow = 0, n_big = 0, bad = 0, n_multi = 0, n_packed = 0;
    uint64_t rng = 88172645463325252ull;
    for (int u = 0; u < n_units; u++) {
        const int r1 = u * nm;
        const int n1 = (int)(seed_off[r1 + 1] - seed_off[r1]), n2 = paired ? (int)(seed_off[r1 + 2] - seed_off[r1 + 1]) : 0;
        if (n1 + n2 > PU_SEEDS) { n_big++; continue; }
        SKey lk[PU_SEEDS]; uint32_t lcw[PU_SEEDS]; uint64_t lrw[2 * PU_SLOTS];
        for (int i = 0; i < n1 + n2; i++) { const uint32_t q = seed_off[r1] + i; lk[i] = sk_make(gpos[q], rpos[q], slen[q]); }
        for (int m = 0; m < nm; m++) {                       // unsorted, as k_locate leaves them
            SKey *seg = lk + (m ? n1 : 0); const int n = m ? n2 : n1;
            for (int i = n - 1; i > 0; i--) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; const int j = (int)(rng % (uint64_t)(i + 1)); const SKey t = seg[i]; seg[i] = seg[j]; seg[j] = t; }
        }
        UnitState st;
        SKey lk2[PU_SEEDS]; uint32_t lcw2[PU_SEEDS]; uint64_t lrw2[2 * PU_SLOTS];
        for (int i = 0; i < n1 + n2; i++) lk2[i] = lk[i];
        d_unit_process<1>(ix, pr, paired != 0, n1, n2, rlen[r1], paired ? rlen[r1 + 1] : 0, seq.data() + seq_off[r1], seq.data() + seq_off[r1 + (paired ? 1 : 0)], lk, lcw, lrw, true, st);
        {   // the same unit with its reads as 2-bit + mask words (a packed batch, dg_map_batch_packed: reads of A/C/G/T/N only): the same state, word for word
            bool plain = true;
            std::vector<uint32_t> words[2];
            int W2m[2] = {1, 1};
            for (int m = 0; m < nm; m++) {
                const int len = rlen[r1 + m], W2 = (len + 15) / 16 > 0 ? (len + 15) / 16 : 1;
                W2m[m] = W2; words[m].assign(2 * (size_t)W2, 0u);
                for (int i = 0; i < 16 * W2; i++) {
                    const unsigned char ch = i < len ? seq[seq_off[r1 + m] + i] : (unsigned char)'N';
                    const char *f = ch ? strchr("ACGT", ch) : nullptr;
                    if (!f && ch != 'N') plain = false;
                    if (f) words[m][i >> 4] |= (uint32_t)(f - "ACGT") << (30 - 2 * (i & 15));
                    else words[m][W2 + (i >> 4)] |= 3u << (30 - 2 * (i & 15));
                }
            }
            if (plain) {
                UnitState s2;
                const ReadWords a{words[0].data(), W2m[0]}, b{words[paired ? 1 : 0].data(), W2m[paired ? 1 : 0]};
                d_unit_process_rd<1, ReadWords>(ix, d_loc_tab(ix), pr, paired != 0, n1, n2, rlen[r1], paired ? rlen[r1 + 1] : 0, a, b, lk2, lcw2, lrw2, true, s2);
                bool same = s2.fast == st.fast && s2.nc[0] == st.nc[0] && s2.nc[1] == st.nc[1];
                if (same && st.fast) {
                    same = s2.n_cig == st.n_cig && s2.n_nw == st.n_nw && s2.n_cells == st.n_cells && memcmp(&s2.rd[0], &st.rd[0], sizeof st.rd[0]) == 0 && (!paired || memcmp(&s2.rd[1], &st.rd[1], sizeof st.rd[1]) == 0);
                    for (int i = 0; same && i < st.nc[0] + st.nc[1]; i++) same = lcw2[i] == lcw[i];
                }
                if (!same) { if (bad < 5) printf("unit %d: the packed-read form of the unit code differs from the ASCII form\n", u); bad++; }
                n_packed++;
            }
        }
        if (!st.fast) { n_slow++; continue; }
        n_fast++;
        dg_read_out o[2]; std::vector<dg_report_out> rep(64); std::vector<uint32_t> cig(256);
        RepLds<1> p1{lcw, lrw, st.nc[0], st.flag0[0]}, p2{lcw + st.nc[0], lrw, st.nc[1], st.flag0[1]};
        const uint32_t nrep1 = (uint32_t)(st.nc[0] > 0 ? st.nc[0] : 1);
        uint32_t used = d_unit_emit_read<1>(true, st.rd[0], p1, 0, 0, &o[0], rep.data(), cig.data());
        if (paired) used += d_unit_emit_read<1>(false, st.rd[1], p2, nrep1, used, &o[1], rep.data(), cig.data());
        if (used != st.n_cig) { if (bad < 5) printf("unit %d: CIGAR op count %u vs scan value %u\n", u, used, st.n_cig); bad++; }
        for (int m = 0; m < nm; m++) {
            const dg_read_out &e = e_reads[r1 + m], &g = o[m];
            bool ok = e.score == g.score && e.sub_score == g.sub_score && e.mis_num == g.mis_num && e.mapq == g.mapq && e.n_rep == g.n_rep && e.best == g.best && e.n_sj == 0;
            if (e.n_rep > 1) n_multi++;
            for (int i = 0; ok && i < e.n_rep; i++) {
                const dg_report_out &er = e_rep[e.rep_off + i], &gr = rep[g.rep_off + i];
                ok = er.aln_score == gr.aln_score && er.sj_type == gr.sj_type && er.flag == gr.flag && er.paired_idx == gr.paired_idx && er.chr == gr.chr && er.bdir == gr.bdir &&
                     er.pos == gr.pos && er.n_cigar == gr.n_cigar;
                for (uint32_t k = 0; ok && k < er.n_cigar; k++) ok = e_cig[er.cigar_off + k] == cig[gr.cigar_off + k];
            }
            if (!ok) { if (bad < 5) printf("unit %d mate %d differs: score %d/%d sub %d/%d mis %d/%d mapq %d/%d n_rep %d/%d best %d/%d\n", u, m, e.score, g.score, e.sub_score, g.sub_score, e.mis_num, g.mis_num, e.mapq, g.mapq, e.n_rep, g.n_rep, e.best, g.best); bad++; }
        }
    }
    printf("units %d: finished here %ld, to the general path %ld, more than %d seeds %ld; reads with several reports %ld; units also run from 2-bit words %ld; bad=%ld\n", n_units, n_fast, n_slow, PU_SEEDS, n_big, n_multi, n_packed, bad);
    return bad ? 1 : 0;
}
