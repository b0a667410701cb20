// Host-side checks of the pure arithmetic helpers of the kernels (compiled with hipcc, run WITHOUT a GPU: no HIP API call).
// Test infrastructure only: `python -m pytest tests -m "not gpu"` builds and runs it (tests/test_kernel_arith_host.py).
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../../dart_amd/csrc/dg_fm.h"
#include "../../dart_amd/csrc/dg_report.h"
extern "C" int orc_nw(const char *s1, const char *s2, char *out1, char *out2, int cap);   // oracle/dart_oracle.h (the checker)

static int ref_nt4(unsigned char c)      // nst_nt4_table as BWT_Index/bntseq.c:40 defines it
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

int main()
{
    long bad = 0;
    for (int c = 0; c < 256; c++) if (d_nt4((unsigned char)c) != ref_nt4((unsigned char)c)) { printf("d_nt4(%d)\n", c); bad++; }
    // d_tr2: the x2 restatement of a float truncated to short (nw_alignment.cpp): 2*trunc(v/2) for v = 2*value
    for (int v = -70000; v <= 70000; v++) { const int want = 2 * (v / 2); if (d_tr2(v) != want) { if (bad < 5) printf("d_tr2(%d) = %d want %d\n", v, d_tr2(v), want); bad++; } }
    // one truncation per nw_alignment cell: the maximum of three truncated operands is the truncated maximum (d_tr2 is monotone); the sentinel of an
    // impossible gap (-131072 and a little below) among the operands
    {
        const int vals[] = {-131080, -131073, -131072, -131071, -40, -7, -6, -5, -4, -3, -2, -1, 0, 1, 2, 3, 4, 5, 6, 7, 33, 300, 301};
        const int nv = (int)(sizeof vals / sizeof vals[0]);
        for (int a = 0; a < nv; a++) for (int b = 0; b < nv; b++) for (int c = 0; c < nv; c++) {
            const int ta = d_tr2(vals[a]), tb = d_tr2(vals[b]), tc = d_tr2(vals[c]);
            const int want = ta > tb ? (ta > tc ? ta : tc) : (tb > tc ? tb : tc);
            if (d_tr2(d_max3(vals[a], vals[b], vals[c])) != want) { if (bad < 5) printf("d_tr2(max3(%d,%d,%d))\n", vals[a], vals[b], vals[c]); bad++; }
        }
        for (int a = -600; a <= 600; a += 1) for (int b = -600; b <= 600; b += 7) for (int c = -131075; c <= -131069; c++) {
            const int ta = d_tr2(a), tb = d_tr2(b), tc = d_tr2(c);
            const int want = ta > tb ? (ta > tc ? ta : tc) : (tb > tc ? tb : tc);
            if (d_tr2(d_max3(a, b, c)) != want || d_tr2(d_max3(c, a, b)) != want) bad++;
        }
    }
    // d_enc4 against one base at a time
    uint64_t s = 88172645463325252ull;
    const char *al = "ACGTacgtNn-\0\xC1\x21XB";
    for (long it = 0; it < 4000000; it++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        unsigned char c[4];
        for (int k = 0; k < 4; k++) c[k] = (it & 1) ? (unsigned char)(s >> (8 * k)) : (unsigned char)al[(s >> (8 * k)) & 15];
        const int nk = (int)((s >> 40) % 5);
        const uint32_t keep = nk >= 4 ? 0xFFFFFFFFu : (nk <= 0 ? 0u : (1u << (8 * nk)) - 1u);
        const uint32_t x = c[0] | (uint32_t)c[1] << 8 | (uint32_t)c[2] << 16 | (uint32_t)c[3] << 24;
        uint32_t b8, m8, eb = 0, em = 0;
        d_enc4(x, keep, b8, m8);
        for (int k = 0; k < 4; k++) { const int v = k < nk ? ref_nt4(c[k]) : 4; eb |= (uint32_t)(v > 3 ? 0 : v) << (6 - 2 * k); em |= (uint32_t)(v > 3 ? 3 : 0) << (6 - 2 * k); }
        if (eb != b8 || em != m8) { if (bad < 5) printf("d_enc4 x=%08x keep=%08x got %02x %02x want %02x %02x\n", x, keep, b8, m8, eb, em); bad++; }
    }
    // RefSequence windows (bwt_index.cpp:193-212): d_ref8 / d_ref_fill against one d_refchar per base, and d_refchar against the
    // definition (forward strand from the 2-bit pac, reverse half = complement of the mirrored base, 0 outside), across both
    // strand boundaries and the ends of the text
    {
        const int64_t L = 1003;                                     // odd on purpose: the last pac byte is partial
        static unsigned char pac[(1003 + 3) / 4 + 8];
        static char fwd[1003];
        for (int64_t i = 0; i < L; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const int b = (int)(s & 3); fwd[i] = "ACGT"[b]; pac[i >> 2] |= (unsigned char)(b << ((~i & 3) << 1)); }
        DIndex ix; memset(&ix, 0, sizeof ix); ix.pac = pac; ix.l_pac = L;
        auto want = [&](int64_t g) -> char {
            if (g < 0 || g >= 2 * L) return 0;
            if (g < L) return fwd[g];
            const char c = fwd[2 * L - 1 - g];
            return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
        };
        for (int64_t g = -20; g < 2 * L + 20; g++) {
            if (d_refchar(ix, g) != want(g)) { if (bad < 5) printf("d_refchar(%ld)\n", (long)g); bad++; }
            const uint64_t w = d_ref8(ix, g);
            for (int k = 0; k < 8; k++) if ((char)(w >> (8 * k)) != want(g + k)) { if (bad < 5) printf("d_ref8(%ld) byte %d\n", (long)g, k); bad++; }
            char buf[40];
            for (int n = 0; n <= 37; n += 37) {
                d_ref_fill(ix, g, n, buf);
                for (int k = 0; k < n; k++) if (buf[k] != want(g + k)) { if (bad < 5) printf("d_ref_fill(%ld,%d) at %d\n", (long)g, n, k); bad++; }
            }
        }
    }
    // d_nw (the strip-in-registers restatement of nw_alignment, x2 integers) against the oracle's nw_alignment on random
    // strings: same gapped strings, character for character (mixed case, N, '-' and odd characters included)
    {
        const int R = 96;
        WSLayout Lw; memset(&Lw, 0, sizeof Lw);
        Lw.nwbits_off = 0; Lw.nwbits_words = (R + 1) * ((2 * R + 32 + 7) / 8);
        Lw.rows_off = Lw.nwbits_words * 4; Lw.row_cap = R + 8;
        static unsigned char wsbuf[(96 + 1) * ((2 * 96 + 32 + 7) / 8) * 4 + 2 * (96 + 8) * 4 + 64];
        LaneCtx cx; memset(&cx, 0, sizeof cx); cx.ws = wsbuf; cx.L = &Lw;
        const char *alpha = "ACGTACGTACGTACGTacgtNn-X";
        for (int it = 0; it < 20000; it++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const int m = 1 + (int)(s % 60), n = 1 + (int)((s >> 20) % 90);
            char a[128], b[128], oa[300], ob[300], ra[300], rb2[300];
            uint64_t t = s;
            for (int i = 0; i < m; i++) { t ^= t << 13; t ^= t >> 7; t ^= t << 17; a[i] = alpha[(it % 7 == 0) ? t % 24 : t % 16]; }
            for (int i = 0; i < n; i++) { t ^= t << 13; t ^= t >> 7; t ^= t << 17; b[i] = (i < m && (t >> 8) % 10 < 7) ? a[i] : alpha[t % 16]; }
            a[m] = 0; b[n] = 0;
            const int len = d_nw(cx, a, m, b, n, oa, ob);
            const int rlen = orc_nw(a, b, ra, rb2, 300);
            bool same = len == rlen;
            for (int i = 0; same && i < len; i++) same = oa[i] == ra[i] && ob[i] == rb2[i];
            if (!same) { if (bad < 5) printf("d_nw differs from the oracle: m=%d n=%d len %d/%d\n", m, n, len, rlen); bad++; }
        }
        printf("d_nw: 20000 random alignments compared with the oracle\n");
    }
    printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
