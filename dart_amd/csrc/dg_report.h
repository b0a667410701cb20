// dart_amd/csrc/dg_report.h -- per-candidate gap filling, splice detection, CIGAR, coordinates,
// then pair finalisation, FLAG, MAPQ and splice-junction tuples.
//
// Replaces GenMappingReport and everything it calls (AlignmentCandidates.cpp:37-61,83-116,136-163,
// 299-306,385-467,547-624,685-1035,1052-1207), KmerAnalysis.cpp:25-166, tools.cpp:40-300,
// nw_alignment.cpp:18-82, and from Mapping.cpp: CheckPairedFinalAlignments :479-530,
// SetSingle/PairedAlignmentFlag :74-186, EvaluateMAPQ :188-206, UpdateLocalSJMap :532-565.
//
// One persistent lane = one read pair at a time (both mates, because pair finalisation and FLAGs
// need both).  All per-lane scratch (CIGAR elements, NW traceback bits and rows, aligned strings,
// re-seeding k-mer list and diagonal ring) lives in a lane-private slice of one HBM workspace;
// each candidate edits its seeds in its own working region (sized by k_chain), so nothing is
// allocated on the device.  Integer/byte work only -- no MFMA.
#pragma once
#include "dg_common.h"

// per-lane workspace layout (bytes), derived on the host from the batch's longest read R
struct WSLayout {
    uint32_t cig_off, cig_cap;          // u32 elements  len<<4|op
    uint32_t nwbits_off, nwbits_words;  // 2 bits per DP cell, rows padded to u32
    uint32_t rows_off, row_cap;         // 3 int rows of row_cap entries
    uint32_t str_off, str_cap;          // 6 char buffers of str_cap
    uint32_t kmer_off, kmer_cap;        // u64 (wid<<32|pos)
    uint32_t ring_off, ring_diag, ring_words;  // ring_diag diagonals x ring_words u64 bitmap words
    uint32_t stride;                    // bytes per lane
    uint32_t max_rlen;
};

struct LaneCtx {
#ifdef DG_PROFILE_CLASSES
    int cls; long long t_last; unsigned long long *ph;
#endif
    uint32_t *lds;                // PM_LDS_WORDS words of LDS owned by this lane (d_pair_nw)
    const DIndex *ix;
    const DParams *pr;
    const unsigned char *seq;   // this read, ASCII
    int rlen;
    unsigned char *ws;          // lane workspace base
    const WSLayout *L;
    unsigned long long n_nw, nw_cells, n_reseed, reseed_w;
};

#define OP_M 0u
#define OP_I 1u
#define OP_D 2u
#define OP_N 3u
#define OP_S 4u
#define CIG(len, op) ((((uint32_t)(len)) << 4) | (op))

__host__ __device__ __forceinline__ uint32_t *ws_cig(LaneCtx &cx) { return (uint32_t *)(cx.ws + cx.L->cig_off); }
__host__ __device__ __forceinline__ char *ws_str(LaneCtx &cx, int i) { return (char *)(cx.ws + cx.L->str_off + (uint32_t)i * cx.L->str_cap); }

__host__ __device__ __forceinline__ int d_tr2(int v2) { return (int)(((unsigned)v2 + ((unsigned)v2 >> 31)) & ~1u); }   // 2*trunc(v2/2)
// nw_alignment.cpp:40-47 takes the maximum of three operands that were each truncated to short first.  Truncation toward zero is monotone (a <= b => trunc(a) <= trunc(b)),
// so max(tr(d), tr(r), tr(t)) == tr(max(d, r, t)): one truncation per cell instead of three (tests/native/host_checks.hip checks the identity over a range of triples).
__host__ __device__ __forceinline__ int d_max3(int a, int b, int c) { const int m = a > b ? a : b; return m > c ? m : c; }

// ---------------------------------------------------------------------------------------------
// nw_alignment (nw_alignment.cpp:18-82) restated on integers x2: s is built from
// operands truncated toward zero to 16-bit integers; traceback needs only the predicates
// s==r and s==t per cell (2 bits).
// Layout of the DP: columns are processed in strips of 8 that live in registers (s and t of the
// previous row); only the strip's left boundary column (s, r per row) goes through the lane's
// scratch, and it is prefetched one row ahead -- a cell costs ~30 VALU instead of ~10 dependent
// memory round trips.  Matrices up to 8 columns (95 % of all calls) touch no row memory at all.
// a: m chars, b: n chars; oa/ob receive the gapped strings; returns their common length.
// ---------------------------------------------------------------------------------------------
#define NW_STRIP 8
__host__ __device__ inline int d_nw(LaneCtx &cx, const char *a, int m, const char *b, int n, char *oa, char *ob)
{
    uint32_t *bits = (uint32_t *)(cx.ws + cx.L->nwbits_off);         // bits[(i-1) * nstrips + strip]: 2 bits per cell
    int *colS = (int *)(cx.ws + cx.L->rows_off), *colR = colS + cx.L->row_cap;
    const int nstrips = (n + NW_STRIP - 1) / NW_STRIP;
    cx.n_nw++; cx.nw_cells += (unsigned long long)m * (unsigned long long)n;
    for (int st = 0; st < nstrips; st++) {
        const int j0 = st * NW_STRIP + 1;                             // first column of the strip (1-based)
        const bool first = st == 0, last = st == nstrips - 1;
        int sp[NW_STRIP], tp[NW_STRIP];
        uint8_t cb[NW_STRIP];
#pragma unroll
        for (int q = 0; q < NW_STRIP; q++) {
            const int j = j0 + q;
            sp[q] = -2 - j; tp[q] = -131072;                          // s[0][j], t[0][j]
            cb[q] = j <= n ? d_nt4((unsigned char)b[j - 1]) : 7;
        }
        int diag0 = first ? 0 : -2 - (j0 - 1);                        // s[0][j0-1]
        int nxtS = 0, nxtR = 0;
        if (!first && m > 0) { nxtS = colS[1]; nxtR = colR[1]; }
        for (int i = 1; i <= m; i++) {
            int left_s, left_r;
            if (first) { left_s = -2 - i; left_r = -131072; }          // s[i][0], r[i][0]
            else {
                left_s = nxtS; left_r = nxtR;
                if (i < m) { nxtS = colS[i + 1]; nxtR = colR[i + 1]; }  // prefetch the next row's boundary
            }
            const uint8_t ca = d_nt4((unsigned char)a[i - 1]);
            int diag = diag0;
            diag0 = left_s;
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < NW_STRIP; q++) {
                int x = left_r - 1, y = left_s - 3;
                const int r = x > y ? x : y;
                x = tp[q] - 1; y = sp[q] - 3;
                const int t = x > y ? x : y;
                const int sv = d_tr2(d_max3(diag + (ca == cb[q] ? 3 : -3), r, t));      // = max of the three truncated operands: the truncation is monotone
                acc |= ((sv == r ? 1u : 0u) | (sv == t ? 2u : 0u)) << (2 * q);
                diag = sp[q];
                sp[q] = sv; tp[q] = t;
                left_s = sv; left_r = r;
            }
            bits[(size_t)(i - 1) * nstrips + st] = acc;
            if (!last) { colS[i] = left_s; colR[i] = left_r; }
        }
    }
    // traceback :61-74, columns produced back to front
    int i = m, j = n, k = 0;
    while (i > 0 || j > 0) {
        uint32_t f;
        if (i == 0) f = 1;               // s[0][j] == r[0][j]
        else if (j == 0) f = 2;          // s[i][0] == t[i][0], r[i][0] is the sentinel
        else f = (bits[(size_t)(i - 1) * nstrips + ((j - 1) >> 3)] >> (((j - 1) & 7) << 1)) & 3u;
        if (f & 1u) { oa[k] = '-'; ob[k] = b[j - 1]; j--; }
        else if (f & 2u) { oa[k] = a[i - 1]; ob[k] = '-'; i--; }
        else { oa[k] = a[i - 1]; ob[k] = b[j - 1]; i--; j--; }
        k++;
    }
    for (int p = 0, q = k - 1; p < q; p++, q--) {
        char c = oa[p]; oa[p] = oa[q]; oa[q] = c;
        c = ob[p]; ob[p] = ob[q]; ob[q] = c;
    }
    return k;
}

// ---------------------------------------------------------------------------------------------
// 8-mer re-seeding between two seeds (ReseedingWithSpecificRegion :596-624 +
// GenerateLongestSimplePairsFromFragmentPair KmerAnalysis.cpp:134-166).
// The reference sorts every 8-mer of the genome window by id, joins, then sorts the pairs by
// (PosDiff,rPos).  Here the window is streamed once: window position g can only hit diagonals
// g-rPos in (g-rl, g], so a ring of rl diagonals, each a bitmap over rPos, is complete -- and can
// be folded into the running (s, max_len) state in increasing diagonal order -- as soon as g has
// moved rl past it.  Same pairs, same order, O(window) time and O(rl^2/8) bytes.
// ---------------------------------------------------------------------------------------------
__device__ inline bool d_reseed(LaneCtx &cx, int rBegin, int rEnd, int64_t Lb, int64_t Rb, DSeed *out)
{
    const DIndex &ix = *cx.ix;
    const int rl = rEnd - rBegin, gl = (int)(Rb - Lb);
    int thr = (int)(rl * 0.85); if (thr < 8) thr = 8;
    cx.n_reseed++; cx.reseed_w += (unsigned long long)(gl > 0 ? gl : 0);
    uint64_t *km = (uint64_t *)(cx.ws + cx.L->kmer_off);
    int nk = 0;
    const unsigned char *rs = cx.seq + rBegin;
    {   // read-gap k-mers in position order, with the reference's 'N' handling
        int count = 0, head, tail = 0;
        uint32_t wid = 0;
        while (count < 8 && tail < rl) { if (rs[tail++] != 'N') count++; else count = 0; }
        if (count == 8) {
            head = tail - 8; wid = 0;
            for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(rs[i]);
            km[nk++] = ((uint64_t)wid << 32) | (uint32_t)head;
            for (head += 1; tail < rl; head++, tail++) {
                if (rs[tail] != 'N') {
                    wid = ((wid & 0x3FFF) << 2) + d_nt4(rs[tail]);
                    km[nk++] = ((uint64_t)wid << 32) | (uint32_t)head;
                } else {
                    count = 0; tail++;
                    while (count < 8 && tail < rl) { if (rs[tail++] != 'N') count++; else count = 0; }
                    if (count == 8) {
                        head = tail - 8; wid = 0;
                        for (int i = head; i < head + 8; i++) wid = (wid << 2) + d_nt4(rs[i]);
                        km[nk++] = ((uint64_t)wid << 32) | (uint32_t)head;
                    } else break;
                }
            }
        }
    }
    if (nk == 0 || gl < 8) return false;
    for (int i = 1; i < nk; i++) {            // sort by (wid,pos)
        uint64_t x = km[i]; int j = i;
        while (j > 0 && km[j - 1] > x) { km[j] = km[j - 1]; j--; }
        km[j] = x;
    }
    uint64_t *ring = (uint64_t *)(cx.ws + cx.L->ring_off);
    const int RD = (int)cx.L->ring_diag, RW = (int)cx.L->ring_words;
    for (int i = 0; i < RD * RW; i++) ring[i] = 0;
    const int span = rl - 8;                  // rPos ranges over [0, span]
    int s = 1, max_len = 0, best_r = 0;
    int64_t best_g = 0;
    int64_t next_fin = -(int64_t)span;        // smallest possible diagonal
    auto finalize = [&](int64_t d) {
        uint64_t *bm = ring + (size_t)((uint64_t)d & (uint64_t)(RD - 1)) * RW;
        int cnt = 0, first = -1, last = -1;
        for (int w = 0; w < RW; w++) {
            const uint64_t v = bm[w];
            if (v) {
                cnt += __popcll(v);
                if (first < 0) first = w * 64 + (__ffsll((unsigned long long)v) - 1);
                last = w * 64 + 63 - __clzll((long long)v);
                bm[w] = 0;
            }
        }
        if (cnt == 0) return;
        s += cnt - 1;
        const int l = 8 + (last - first);
        if (l > max_len && s > (l - 8) / 2) { best_r = first; best_g = d + first; max_len = l; s = 1; }
    };
    // the window holds no 'N' (RefSequence is ACGT, or '\0' past the end which is not 'N' either)
    uint32_t wid = 0;
    for (int i = 0; i < 8; i++) wid = (wid << 2) + d_nt4((unsigned char)d_refchar(ix, Lb + i));
    for (int g = 0; g + 8 <= gl; g++) {
        if (g > 0) wid = ((wid & 0x3FFF) << 2) + d_nt4((unsigned char)d_refchar(ix, Lb + g + 7));
        while (next_fin < (int64_t)g - span) { finalize(next_fin); next_fin++; }
        int lo = 0, hi = nk;                  // first read k-mer with this id
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)(km[mid] >> 32) < wid) lo = mid + 1; else hi = mid; }
        for (; lo < nk && (uint32_t)(km[lo] >> 32) == wid; lo++) {
            const int rp = (int)(uint32_t)km[lo];
            const int64_t d = (int64_t)g - rp;
            ring[(size_t)((uint64_t)d & (uint64_t)(RD - 1)) * RW + (rp >> 6)] |= 1ull << (rp & 63);
        }
    }
    for (const int64_t endd = (int64_t)(gl - 8); next_fin <= endd; next_fin++) finalize(next_fin);
    if (max_len >= thr && max_len > 0) {
        out->rLen = out->gLen = max_len;
        out->rPos = best_r + rBegin;
        out->gPos = best_g + Lb;
        out->flags = SEED_SIMPLE;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// clean-up of a candidate's seed list before re-seeding (GenMappingReport :1104-1110)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int d_drop_dead_seeds(DSeed *s, int n)   // RemoveNullSeeds :299-306: seeds with rLen == 0 leave the list, order kept
{
    int w = 0;
    for (int i = 0; i < n; i++) { if (s[i].rLen == 0) continue; if (w != i) s[w] = s[i]; w++; }
    return w;
}

// RemoveTandemRepeatSeeds (:817-842) and RemoveTranslocatedSeeds (:844-902) in one entry point.  The list is in genome order.
//  * Both passes are the identity when the read positions already increase along the list (every seed's rPos larger than
//    its predecessor's): no two seeds share a read position and read order equals genome order.  That is the normal case
//    and is decided with one look at each seed.
//  * Otherwise, tandem pass: every seed whose rPos also belongs to another seed is dropped -- two bit sets over the read
//    positions (seen / seen twice) instead of the reference's sort by rPos.
//  * Then, translocation pass: ord[k] = list index of the seed with the k-th smallest rPos.  Where ord[k] != k a block of
//    ranks [k, end] that is closed under ord is collected (end grows to the largest ord inside); its seeds that come EARLIER
//    in the read than in the genome (k < ord[k]) are weighed against the others by read bases covered, and the lighter side's
//    displaced seeds are dropped (ties drop the early side).
// scratch: 2 * (rlen / 32 + 1) + n u32 words.
__host__ __device__ inline int d_untangle_seeds(DSeed *s, int n, int rlen, uint32_t *scratch)
{
    if (n < 2) return n;
    bool rising = true;
    for (int i = 1; i < n && rising; i++) rising = s[i].rPos > s[i - 1].rPos;
    if (rising) return n;
    {   // tandem pass
        const int words = rlen / 32 + 1;
        uint32_t *once = scratch, *twice = scratch + words;
        for (int w = 0; w < 2 * words; w++) scratch[w] = 0;
        for (int i = 0; i < n; i++) {
            const uint32_t bit = 1u << (s[i].rPos & 31), w = (uint32_t)s[i].rPos >> 5;
            twice[w] |= once[w] & bit;
            once[w] |= bit;
        }
        bool any = false;
        for (int i = 0; i < n; i++)
            if ((twice[(uint32_t)s[i].rPos >> 5] >> (s[i].rPos & 31)) & 1u) { s[i].rLen = s[i].gLen = 0; any = true; }
        if (any) n = d_drop_dead_seeds(s, n);
        if (n < 2) return n;
    }
    // translocation pass: ranks by rPos (ties cannot occur any more; the index in the low bits keeps the sort stable anyway)
    uint32_t *ord = scratch;
    for (int i = 0; i < n; i++) {
        const uint32_t x = ((uint32_t)s[i].rPos << 20) | (uint32_t)i;      // rPos < 4096, i < 2^20
        int j = i;
        for (; j > 0 && ord[j - 1] > x; j--) ord[j] = ord[j - 1];
        ord[j] = x;
    }
    bool any = false;
    for (int k = 0; k < n; k++) {
        if ((int)(ord[k] >> 20) == s[k].rPos) continue;               // the k-th seed of the read is the k-th of the genome
        int end = (int)(ord[k] & 0xFFFFFu);
        for (int q = k + 1; q <= end; q++) { const int g = (int)(ord[q] & 0xFFFFFu); end = g > end ? g : end; }
        int early = 0, rest = 0;
        for (int q = k; q <= end; q++) { const int g = (int)(ord[q] & 0xFFFFFu); if (q < g) early += s[g].rLen; else rest += s[g].rLen; }
        const bool keep_early = early > rest;
        for (int q = k; q <= end; q++) {
            const int g = (int)(ord[q] & 0xFFFFFu);
            if (keep_early ? q > g : q < g) { s[g].rLen = s[g].gLen = 0; any = true; }
        }
        k = end;
    }
    return any ? d_drop_dead_seeds(s, n) : n;
}

__device__ inline void d_insertion_sort_seeds(DSeed *a, int n)
{
    for (int i = 1; i < n; i++) {
        DSeed x = a[i];
        int j = i;
        while (j > 0 && d_seed_less(x, a[j - 1])) { a[j] = a[j - 1]; j--; }
        a[j] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// splice junctions (:6,702-815; main.cpp:18)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int d_shift_arr(int i) { return i == 0 ? 0 : ((i & 1) ? (i + 1) / 2 : -(i / 2)); }   // ShiftArr :6

__host__ __device__ inline bool d_check_seq_fragment(const DIndex &ix, int64_t Lg, int64_t Rg, int shift)   // :702-730
{
    if (shift <= 0) { shift = -shift; Lg -= shift; Rg -= shift; }
    // shift <= 9 (ShiftArr): eight bases of each side in one fetch, the ninth on its own
    const int n8 = shift < 8 ? shift : 8;
    if (n8 > 0) {
        const uint64_t m = n8 >= 8 ? ~0ull : ((1ull << (8 * n8)) - 1ull);
        if ((d_ref8(ix, Lg) ^ d_ref8(ix, Rg)) & m) return false;
    }
    for (int i = 8; i < shift; i++) if (d_refchar(ix, Lg + i) != d_refchar(ix, Rg + i)) return false;
    return true;
}

// RefSequence[g0 .. g0 + n) as 2-bit codes (A C G T = 0 1 2 3), n <= 28, first base in the top bits -- when the bases, and the eight pac bytes the fetch
// reads, lie inside one strand half; *ok says so.  One 8-byte fetch instead of n dependent d_refchar loads.
__host__ __device__ __forceinline__ uint64_t d_ref_codes(const DIndex &ix, int64_t g0, int n, bool *ok)
{
    typedef uint64_t __attribute__((aligned(1))) uint64_a1;
    const int64_t L = ix.l_pac;
    if (g0 >= 0 && g0 + 32 <= L) {
        *ok = true;
        return __builtin_bswap64(*(const uint64_a1 *)(ix.pac + (g0 >> 2))) << ((g0 & 3) << 1);
    }
    if (g0 >= L + 32 - n && g0 + n <= 2 * L) {            // RefSequence[g0 + j] = complement of forward base f0 - j
        const int64_t lo = 2 * L - 1 - g0 - (n - 1);      // the window's last base is forward base lo: 0 <= lo, lo + 32 <= L
        const uint64_t x = __builtin_bswap64(*(const uint64_a1 *)(ix.pac + (lo >> 2))) << ((lo & 3) << 1);       // forward base lo + k at bits 63-2k, 62-2k
        uint64_t y = __builtin_bitreverse64(x);                                                                  // ... at bits 2k, 2k+1 (swapped inside the pair)
        y = ((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull);
        *ok = true;
        return ~(y << (64 - 2 * n));                      // window base j = complement of forward base lo + (n-1-j), now at bits 63-2j, 62-2j
    }
    *ok = false;
    return 0;
}
__host__ __device__ __forceinline__ uint64_t d_codes_field(uint64_t w, int first, int count) { return (w << (2 * first)) >> (64 - 2 * count); }      // count >= 1

__host__ __device__ inline int d_identify_sj_chars(const DIndex &ix, int type, const DSeed &l, const DSeed &r);
__host__ __device__ inline int d_identify_sj(const DIndex &ix, int type, const DSeed &l, const DSeed &r)   // :732-756
{
    // every base the search can look at is within 9 of the two boundaries: 20 bases from Lg - 9 on (donor pair at Lg + shift, the bases a shift moves across)
    // and 20 from Rg - 11 on (acceptor pair at Rg - 2 + shift, the same bases on that side), fetched once; the loop then runs in registers
    const int64_t Lg = l.gPos + l.gLen, Rg = r.gPos;
    bool okl, okr;
    const uint64_t WL = d_ref_codes(ix, Lg - 9, 20, &okl), WR = d_ref_codes(ix, Rg - 11, 20, &okr);
    if (!(okl && okr)) return d_identify_sj_chars(ix, type, l, r);              // a window at a strand boundary or at the ends of the text
    // SpliceJunctionArr = { "GT/AG", "CT/AC", "GC/AG", "CT/GC" } as codes
    const uint64_t donor = type == 0 ? 0xBu : (type == 2 ? 0x9u : 0x7u), acceptor = type == 0 ? 0x2u : (type == 1 ? 0x1u : (type == 2 ? 0x2u : 0x9u));
    int i = l.rLen < r.rLen ? l.rLen : r.rLen;
    int j = l.gLen < r.gLen ? l.gLen : r.gLen;
    if (i < j) j = i;
    if (j > 9) j = 9;
    j <<= 1;
    int shift = 0;
    for (i = 0; i <= j; i++) {
        shift = d_shift_arr(i);
        if (shift > 0 && d_codes_field(WL, 9, shift) != d_codes_field(WR, 11, shift)) continue;                 // CheckSeqFragment :702-730
        if (shift < 0 && d_codes_field(WL, 9 + shift, -shift) != d_codes_field(WR, 11 + shift, -shift)) continue;
        if (d_codes_field(WL, 9 + shift, 2) == donor && d_codes_field(WR, 9 + shift, 2) == acceptor) break;
    }
    return i > j ? 10 : shift;
}

__host__ __device__ inline int d_identify_sj_chars(const DIndex &ix, int type, const DSeed &l, const DSeed &r)   // the same through single characters
{
    // SpliceJunctionArr = { "GT/AG", "CT/AC", "GC/AG", "CT/GC" }
    const char d0 = type == 0 ? 'G' : (type == 2 ? 'G' : 'C');
    const char d1 = type == 2 ? 'C' : 'T';
    const char a0 = type == 3 ? 'G' : 'A';
    const char a1 = (type == 0 || type == 2) ? 'G' : 'C';
    int i = l.rLen < r.rLen ? l.rLen : r.rLen;
    int j = l.gLen < r.gLen ? l.gLen : r.gLen;
    if (i < j) j = i;
    if (j > 9) j = 9;
    j <<= 1;
    const int64_t Lg = l.gPos + l.gLen, Rg = r.gPos;
    int shift = 0;
    for (i = 0; i <= j; i++) {
        shift = d_shift_arr(i);
        if (shift != 0 && !d_check_seq_fragment(ix, Lg, Rg, shift)) continue;
        const int64_t g1 = Lg + shift, g2 = Rg - 2 + shift;
        if (d_refchar(ix, g1) == d0 && d_refchar(ix, g1 + 1) == d1 && d_refchar(ix, g2) == a0 && d_refchar(ix, g2 + 1) == a1) break;
    }
    return i > j ? 10 : shift;
}

// CheckSpliceJunction (:758-815).  A junction = two neighbouring exact seeds whose diagonals are more than MinIntronSize apart.
// The four motifs are tried in their fixed order; a motif's cost is the sum of |boundary shift| over the junctions (a junction
// without the motif costs 10) and it competes only if it fits at least one junction; the cheapest wins (the earlier one on
// a tie) and the search stops at the first motif that fits every junction.  The winner's junctions get bAcceptorSite and
// their boundary moved by the shift.  Returns the motif (SJtype) or -1.  trial/keep: 2 * num int2 of scratch.
__host__ __device__ inline int d_check_splice(LaneCtx &cx, DSeed *s, int num, int2 *trial)
{
    const DIndex &ix = *cx.ix;
    int2 *keep = trial + num + 1;
    int kept = 0, kept_cost = 1000, motif_won = -1;
    for (int motif = 0; motif < 4; motif++) {
        int fits = 0, cost = 0, without = 0;
        for (int j = 1; j < num; j++) {
            const DSeed &up = s[j - 1], &dn = s[j];
            if (!(up.flags & dn.flags & SEED_SIMPLE) || (dn.gPos - dn.rPos) - (up.gPos - up.rPos) <= cx.pr->min_intron) continue;
            const int sh = d_identify_sj(ix, motif, up, dn);
            cost += sh < 0 ? -sh : sh;
            if (sh == 10) without++; else trial[fits++] = make_int2(j, sh);
        }
        if (fits > 0 && cost < kept_cost) { kept_cost = cost; motif_won = motif; kept = fits; for (int q = 0; q < fits; q++) keep[q] = trial[q]; }
        if (without == 0) break;
    }
    for (int q = 0; q < kept; q++) {
        DSeed &up = s[keep[q].x - 1], &dn = s[keep[q].x];
        const int sh = keep[q].y;
        dn.flags |= SEED_ACCEPTOR;
        up.rLen += sh; up.gLen += sh;                       // (sh == 0: nothing moves)
        dn.rLen -= sh; dn.gLen -= sh; dn.rPos += sh; dn.gPos += sh;
    }
    return motif_won;
}

// ---------------------------------------------------------------------------------------------
// overlaps and normal pairs (:904-1035)
// ---------------------------------------------------------------------------------------------
// CheckSeedOverlapping (:904-954): `lo` precedes `hi` in the list.  First along the read, then (if both are still alive) along
// the genome: when they overlap by ov, the shorter of the two in that dimension gives way -- `lo` loses its last ov bases,
// `hi` its first ov (a seed not longer than ov dies).  A trimmed seed is exact, so its gLen follows its rLen.
// Returns false when `lo` was the one that gave way (the caller then stops comparing it with later seeds).
__host__ __device__ inline bool d_resolve_overlap(DSeed &lo, DSeed &hi)
{
    bool lo_intact = true;
    auto trim_end = [](DSeed &x, int ov) { x.rLen = x.rLen > ov ? x.rLen - ov : 0; x.gLen = x.rLen; };
    auto trim_front = [](DSeed &x, int ov) { if (x.rLen > ov) { x.rPos += ov; x.gPos += ov; x.rLen -= ov; x.gLen = x.rLen; } else x.rLen = x.gLen = 0; };
    const int on_read = lo.rPos + lo.rLen - hi.rPos;
    if (on_read > 0) {
        if (lo.rLen < hi.rLen) { trim_end(lo, on_read); lo_intact = false; } else trim_front(hi, on_read);
    }
    if (lo.rLen > 0 && hi.rLen > 0) {
        const int on_genome = (int)(lo.gPos + lo.gLen - hi.gPos);
        if (on_genome > 0) {
            if (lo.gLen < hi.gLen) { trim_end(lo, on_genome); lo_intact = false; } else trim_front(hi, on_genome);
        }
    }
    return lo_intact;
}

// CheckOverlappingSeeds (:956-999): every live seed, in list order, is compared with the live seeds after it until one starts
// beyond its end in both coordinates (the ends as they were when its turn began) or it gave way itself.  A seed that died on its
// own turn sends the cursor back to the nearest live seed before it (or the list head), whose turn is taken again.
__host__ __device__ inline int d_trim_overlaps(DSeed *s, int num)
{
    if (num < 2) return num;
    bool holes = false;
    int at = 0;
    while (at < num) {
        if (s[at].rLen == 0) { holes = true; at++; continue; }
        const int r_last = s[at].rPos + s[at].rLen - 1;
        const int64_t g_last = s[at].gPos + s[at].gLen - 1;
        for (int nx = at + 1; nx < num; nx++) {
            if (s[nx].rLen == 0) continue;
            if (s[nx].rPos > r_last && s[nx].gPos > g_last) break;
            if (!d_resolve_overlap(s[at], s[nx])) break;
        }
        if (s[at].rLen != 0) { at++; continue; }
        holes = true;
        at--;
        while (at > 0 && s[at].rLen == 0) at--;
        if (at < 0) at = 0;
    }
    return holes ? d_drop_dead_seeds(s, num) : num;
}

// IdentifyNormalPairs (:1001-1035): after the overlap trimming, every two neighbours that do not touch on the read get a
// non-exact pair holding what lies between them: the read bases (none when the neighbours overlap there) and the genome bases
// -- none when negative, and none when there are more than 30 AND more than twice the read bases (that gap becomes an N
// in the CIGAR instead).  Pairs with nothing on either side are not made.  The new pairs are then merged into list order
// (stable: behind every older element that is not greater).
__host__ __device__ inline int d_identify_normal_pairs(DSeed *s, int n)
{
    if (n < 2) return n;
    n = d_trim_overlaps(s, n);
    const int old = n;
    for (int j = 1; j < old; j++) {
        const DSeed &up = s[j - 1], &dn = s[j];
        const int r_from = up.rPos + up.rLen, between_r = dn.rPos - r_from;
        if (between_r == 0) continue;                                   // (also skips a genome-only gap: it will be an N or D later)
        const int64_t g_from = up.gPos + up.gLen;
        const int rG = between_r > 0 ? between_r : 0;
        int gG = (int)(dn.gPos - g_from);
        if (gG < 0 || (gG > 30 && gG > 2 * rG)) gG = 0;
        if (rG == 0 && gG == 0) continue;
        DSeed fill; fill.flags = 0; fill.rPos = r_from; fill.gPos = g_from; fill.rLen = rG; fill.gLen = gG;
        s[n++] = fill;
    }
    // the two-way merge of std::inplace_merge(begin, begin + old, end) (:1033), done in place: the new pairs are taken in the
    // order they were made; each goes in front of the first not-yet-passed older element that is greater than it, and the
    // walk over the older elements never goes back (so the result is the reference's even if the new pairs are not in order)
    int at = 0;
    for (int t = old; t < n; t++) {
        const DSeed x = s[t];
        while (at < t && !d_seed_less(x, s[at])) at++;
        for (int j = t; j > at; j--) s[j] = s[j - 1];
        s[at++] = x;
    }
    return n;
}

__device__ inline bool d_check_coordinate_validity(const DIndex &ix, const DSeed *s, int n)   // :136-163
{
    int64_t g1 = 0, g2 = 2 * ix.l_pac;
    const int64_t L = ix.l_pac;
    for (int i = 0; i < n; i++) if (s[i].gLen > 0) { g1 = s[i].gPos; break; }
    for (int i = n - 1; i >= 0; i--) if (s[i].gLen > 0) { g2 = s[i].gPos + s[i].gLen - 1; break; }
    return !((g1 < L && g2 >= L) || (g1 >= L && g2 < L));
}

// ---------------------------------------------------------------------------------------------
// segment pair -> CIGAR elements (tools.cpp:40-104,130-300)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int d_add_cigar(const char *s1, const char *s2, int len, uint32_t *cig, int &nc)   // AddNewCigarElements :49-104
{
    uint32_t state = 99;
    int c = 0, score = 0;
    for (int i = 0; i < len; i++) {
        uint32_t st;
        if (s1[i] == '-') st = OP_D;
        else if (s2[i] == '-') st = OP_I;
        else { st = OP_M; if (s1[i] == s2[i]) score++; }
        if (state == st) c++;
        else { if (c > 0) cig[nc++] = CIG(c, state); c = 1; state = st; }
    }
    if (c > 0) cig[nc++] = CIG(c, state);
    return score;
}

__host__ __device__ inline bool d_local_quality(const char *a1, const char *a2, int len)   // CheckLocalAlignmentQuality :166-201
{
    int n = 0, mis = 0, type = -1, st = 0;
    for (int i = 0; i < len; i++) {
        if (a1[i] == '-') { if (type != 0) { type = 0; st++; } }
        else if (a2[i] == '-') { if (type != 1) { type = 1; st++; } }
        else { n++; if (a1[i] != a2[i]) mis++; if (type != 2) { type = 2; st++; } }
    }
    return !(st >= 4 || (mis >= 3 && mis >= (int)(n * 0.3)));
}

// ---------------------------------------------------------------------------------------------
// Register-only path of d_process_pair for pairs of at most 8 x 8 bases (90 % of all calls, 97 % of
// all nw_alignment calls: a substitution leaves a 1 x 1 pair between two exact seeds).  Same
// arithmetic as d_nw / d_add_cigar / d_local_quality, but the two strings are 8 ASCII bytes in a
// register pair, the traceback bits of the (at most) 8 rows are 4 registers, and the gapped
// alignment is a column list (3 bits per column: 0 = both bases, 1 = gap in the read, 2 = gap in
// the genome, +4 = the two bases are the same character) -- no round trip through the lane's
// scratch memory, which is what the generic path spends its time waiting for.
// ---------------------------------------------------------------------------------------------
// the gapped alignment as a column list, 4 bits per column, column 0 = the LAST column (traceback order)
struct ColList { uint64_t w0, w1, w2; int K; };
__host__ __device__ __forceinline__ uint32_t d_colcode(const ColList &c, int p)          // p counts from the first column
{
    const int k = c.K - 1 - p;
    const uint64_t w = k < 16 ? c.w0 : (k < 32 ? c.w1 : c.w2);
    return (uint32_t)(w >> ((k & 15) << 2)) & 7u;
}
__host__ __device__ __forceinline__ uint32_t d_byte3(uint64_t a0, uint64_t a1, uint64_t a2, int i)   // byte i of 24
{
    const uint64_t w = i < 8 ? a0 : (i < 16 ? a1 : a2);
    return (uint32_t)(w >> ((i & 7) << 3)) & 0xFFu;
}

__device__ inline int d_add_cigar_cols(const ColList &cl, int p0, int p1, uint32_t *cig, int &nc)   // d_add_cigar on a column list
{
    uint32_t state = 99;
    int c = 0, score = 0;
    for (int p = p0; p < p1; p++) {
        const uint32_t cd = d_colcode(cl, p), ty = cd & 3u;
        const uint32_t st = ty == 1 ? OP_D : (ty == 2 ? OP_I : OP_M);
        if (ty == 0 && (cd & 4u)) score++;
        if (state == st) c++;
        else { if (c > 0) cig[nc++] = CIG(c, state); c = 1; state = st; }
    }
    if (c > 0) cig[nc++] = CIG(c, state);
    return score;
}

// ---------------------------------------------------------------------------------------------
// ProcessNormal/Head/TailSequencePair for pairs of at most PM_MAX x PM_MAX bases, split in three so
// that the nw_alignment calls of a wave's lanes can be executed TOGETHER (d_gen_mapping_report):
//   d_pair_classify  the two strings as ASCII bytes in three register pairs each; the cheap
//                    outcomes (equal length with <= 2 and <= 20 % mismatches -> M; 1 x 1) or
//                    "needs nw_alignment"
//   d_pair_nw        nw_alignment in strips of 8 columns held in registers (as d_nw), traceback
//                    bits and strip boundary column in a 252-byte LDS slice of the lane, result =
//                    a column list in three registers (4 bits per column: 0 = both bases, 1 = gap
//                    in the read, 2 = gap in the genome, +4 = the two bases are the same character)
//   d_pair_finish    AddNewCigarElements / CheckLocalAlignmentQuality / head and tail trimming on it
// On a chr20-sized text almost every pair is 1 x 1 (a substituted base between two exact seeds);
// on a GRCh38-sized one a chance 16-mer hit elsewhere usually follows the substitution (4^16 <
// text length), the true locus resumes ~18 bases later and the typical pair is ~12 x 12: executed
// one lane at a time, wherever each lane's loop happened to be, those alignments were 75 % of k_report.
// ---------------------------------------------------------------------------------------------
#define PM_MAX 24
#define PM_LDS_WORDS 63          // per lane: 24 rows x 3 strips x u16 bits (144 B) + 2 x 25 x int16 boundary (100 B), odd stride
#define PM_MAXQ 4                // nw_alignment results kept per candidate (more pairs than that: the rest run in place)
struct PairStr { uint64_t A0, A1, A2, B0, B1, B2; };
enum { PC_GENERIC = 0, PC_TRIVIAL, PC_EQUAL, PC_ONE, PC_NW };

// which way does this pair go?  PC_GENERIC = too long or a literal '-' in the read (string path); PC_TRIVIAL = mode-2
// early outs (tools.cpp:132-141); PC_EQUAL = equal length, few mismatches (*nm set); PC_ONE = 1 x 1; PC_NW
__device__ inline int d_pair_classify(LaneCtx &cx, const DSeed &sp, int mode, PairStr &ps, int &nm)
{
    const DIndex &ix = *cx.ix;
    if (mode == 2 && (sp.gPos - sp.rPos == -1 || sp.rLen == 0 || sp.gLen == 0)) return PC_TRIVIAL;
    const int m = sp.rLen, n = sp.gLen;
    if (m > PM_MAX || n > PM_MAX) return PC_GENERIC;
    const unsigned char *rp = cx.seq + sp.rPos;
    const uint2 r0 = *(const uint2_a1 *)rp, r1 = *(const uint2_a1 *)(rp + 8), r2 = *(const uint2_a1 *)(rp + 16);
    auto keep = [](int len, int w) -> uint64_t { const int r = len - 8 * w; return r >= 8 ? ~0ull : (r <= 0 ? 0ull : ((1ull << (8 * r)) - 1ull)); };
    ps.A0 = d_u64(r0.x, r0.y) & keep(m, 0); ps.A1 = d_u64(r1.x, r1.y) & keep(m, 1); ps.A2 = d_u64(r2.x, r2.y) & keep(m, 2);
    {   // a literal '-' in the read changes what AddNewCigarElements sees: leave those to the string path
        const uint64_t k = 0x2D2D2D2D2D2D2D2Dull, o = 0x0101010101010101ull, h = 0x8080808080808080ull;
        const uint64_t z0 = ps.A0 ^ k, z1 = ps.A1 ^ k, z2 = ps.A2 ^ k;
        if ((((z0 - o) & ~z0 & h) & keep(m, 0)) | (((z1 - o) & ~z1 & h) & keep(m, 1)) | (((z2 - o) & ~z2 & h) & keep(m, 2))) return PC_GENERIC;
    }
    ps.B0 = d_ref8(ix, sp.gPos) & keep(n, 0);
    ps.B1 = n > 8 ? d_ref8(ix, sp.gPos + 8) & keep(n, 1) : 0ull;
    ps.B2 = n > 16 ? d_ref8(ix, sp.gPos + 16) & keep(n, 2) : 0ull;
    if (m == n) {
        nm = 0;
        for (int i = 0; i < m; i++) if (d_byte3(ps.A0, ps.A1, ps.A2, i) != d_byte3(ps.B0, ps.B1, ps.B2, i)) nm++;     // CalFragPairMismatchBases :40-47
        if (nm <= 2 && nm <= (int)(m * 0.2)) return PC_EQUAL;
    }
    // 1 x 1 with two different characters: s[1][1] = tr(-1.5) = -1 beats r = t = -3 whatever the characters are, so the
    // traceback is the diagonal: one M column with unequal characters, in every mode (one state change, one mismatch:
    // the local quality check passes and nothing is trimmed)
    if (m == 1 && n == 1) return PC_ONE;
    return PC_NW;
}

__host__ __device__ inline void d_pair_nw(LaneCtx &cx, int m, int n, const PairStr &ps, ColList &cl)
{
    cx.n_nw++; cx.nw_cells += (unsigned long long)m * (unsigned long long)n;
    uint16_t *bits = (uint16_t *)cx.lds;                     // bits[(i-1) * 3 + strip]
    int16_t *colS = (int16_t *)(cx.lds + 36), *colR = colS + 25;
    const int nstrips = (n + NW_STRIP - 1) / NW_STRIP;
    for (int st = 0; st < nstrips; st++) {
        const int j0 = st * NW_STRIP + 1;
        const bool first = st == 0, last = st == nstrips - 1;
        int sp_[NW_STRIP], tp_[NW_STRIP];
        uint8_t cb[NW_STRIP];
#pragma unroll
        for (int q = 0; q < NW_STRIP; q++) {
            const int j = j0 + q;
            sp_[q] = -2 - j; tp_[q] = -131072;
            cb[q] = j <= n ? d_nt4((unsigned char)d_byte3(ps.B0, ps.B1, ps.B2, j - 1)) : 7;
        }
        int diag0 = first ? 0 : -2 - (j0 - 1);
        for (int i = 1; i <= m; i++) {
            int left_s, left_r;
            if (first) { left_s = -2 - i; left_r = -131072; }
            else { left_s = colS[i]; left_r = colR[i]; }
            const uint8_t ca = d_nt4((unsigned char)d_byte3(ps.A0, ps.A1, ps.A2, i - 1));
            int diag = diag0;
            diag0 = left_s;
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < NW_STRIP; q++) {
                int x = left_r - 1, y = left_s - 3;
                const int r = x > y ? x : y;
                x = tp_[q] - 1; y = sp_[q] - 3;
                const int t = x > y ? x : y;
                const int sv = d_tr2(d_max3(diag + (ca == cb[q] ? 3 : -3), r, t));      // = max of the three truncated operands: the truncation is monotone
                acc |= ((sv == r ? 1u : 0u) | (sv == t ? 2u : 0u)) << (2 * q);
                diag = sp_[q];
                sp_[q] = sv; tp_[q] = t;
                left_s = sv; left_r = r;
            }
            bits[(i - 1) * 3 + st] = (uint16_t)acc;
            if (!last) { colS[i] = (int16_t)left_s; colR[i] = (int16_t)left_r; }
        }
    }
    // traceback :61-74 into the column list, last column first
    cl.w0 = cl.w1 = cl.w2 = 0; cl.K = 0;
    int i = m, j = n;
    while (i > 0 || j > 0) {
        uint32_t fl;
        if (i == 0) fl = 1;
        else if (j == 0) fl = 2;
        else fl = ((uint32_t)bits[(i - 1) * 3 + ((j - 1) >> 3)] >> (((j - 1) & 7) << 1)) & 3u;
        uint64_t code;
        if (fl & 1u) { code = 1; j--; }
        else if (fl & 2u) { code = 2; i--; }
        else { code = d_byte3(ps.A0, ps.A1, ps.A2, i - 1) == d_byte3(ps.B0, ps.B1, ps.B2, j - 1) ? 4u : 0u; i--; j--; }
        const int sh = (cl.K & 15) << 2;
        if (cl.K < 16) cl.w0 |= code << sh; else if (cl.K < 32) cl.w1 |= code << sh; else cl.w2 |= code << sh;
        cl.K++;
    }
}

__device__ inline int d_pair_finish(const ColList &cl, DSeed &sp, int mode, uint32_t *cig, int &nc)
{
    const int K = cl.K;
    if (mode == 2) return d_add_cigar_cols(cl, 0, K, cig, nc);
    {   // CheckLocalAlignmentQuality :166-201
        int nn = 0, mis = 0, type = -1, st = 0;
        for (int p = 0; p < K; p++) {
            const uint32_t cd = d_colcode(cl, p), ty = cd & 3u;
            if (ty == 1) { if (type != 0) { type = 0; st++; } }
            else if (ty == 2) { if (type != 1) { type = 1; st++; } }
            else { nn++; if (!(cd & 4u)) mis++; if (type != 2) { type = 2; st++; } }
        }
        if (st >= 4 || (mis >= 3 && mis >= (int)(nn * 0.3))) { cig[nc++] = CIG(sp.rLen, OP_S); return 0; }
    }
    if (mode == 0) {
        int p0 = 0, p = 0;
        while (p0 + p < K && (d_colcode(cl, p0 + p) & 3u) == 1) p++;
        if (p > 0) { p0 += p; sp.gPos += p; sp.gLen -= p; }
        p = 0;
        while (p0 + p < K && (d_colcode(cl, p0 + p) & 3u) == 2) p++;
        if (p > 0) { p0 += p; sp.rPos += p; sp.rLen -= p; cig[nc++] = CIG(p, OP_S); }
        return d_add_cigar_cols(cl, p0, K, cig, nc);
    }
    int len = K, p = len - 1, c = 0;
    while (p >= 0 && (d_colcode(cl, p) & 3u) == 1) { c++; p--; }
    if (c > 0) { len -= c; sp.gLen -= c; }
    p = len - 1; c = 0;
    while (p >= 0 && (d_colcode(cl, p) & 3u) == 2) { c++; p--; }
    if (c > 0) { len -= c; sp.rLen -= c; }
    const int score = d_add_cigar_cols(cl, 0, len, cig, nc);
    if (c > 0) cig[nc++] = CIG(c, OP_S);
    return score;
}

// ---------------------------------------------------------------------------------------------
// nw_alignment of ONE large pair by the whole wave (all 64 lanes call it with the same arguments):
// anti-diagonal wavefront, lane = column (blocks of 64 columns), the three values a cell needs from
// its left neighbour come by shuffle.  A 30 x 50 alignment is 80 steps of ~45 instructions instead of
// 1500 cells x 30 in one lane while 63 lanes wait: on a GRCh38-sized text those large pairs (a chance
// 16-mer hit elsewhere after a substitution, the true locus resuming ~30 bases later) hold 88 % of all
// nw_alignment cells.  Output: traceback bits in the OWNER lane's scratch, column-major:
// tb[(j-1) * RW + (i-1)/16], 2 bits per cell; the owner then runs the traceback itself.
// ---------------------------------------------------------------------------------------------
__device__ inline void d_nw_coop(const DIndex &ix, const unsigned char *a, int m, int64_t gPos, int n, unsigned char *ows, const WSLayout &L, int lane)
{
    uint32_t *tb = (uint32_t *)(ows + L.nwbits_off);
    int *colS = (int *)(ows + L.rows_off), *colR = colS + L.row_cap;   // s, r of the previous block's last column
    const int RW = (m + 15) >> 4;
    for (int j0 = 1; j0 <= n; j0 += 64) {
        const int nb = n - j0 + 1 < 64 ? n - j0 + 1 : 64;
        const int j = j0 + lane;
        const bool col_ok = lane < nb, firstblk = j0 == 1, lastblk = j0 + 64 > n;
        const uint8_t cb = col_ok ? d_nt4((unsigned char)d_refchar(ix, gPos + j - 1)) : 7;
        int up_s = -2 - j, up_t = -131072;                 // s[0][j], t[0][j]
        int cur_s = 0, cur_r = 0, prev_s = 0;              // this lane's cell of the previous step (row i) and the one before
        uint32_t acc = 0;
        __syncthreads();                                   // the previous block's boundary column is in memory
        unsigned char a_nxt = (1 - lane >= 1 && 1 - lane <= m) ? a[1 - lane - 1] : 0;   // the row's read character, fetched a step ahead
        for (int t = 1; t <= m + nb - 1; t++) {
            // the left neighbour's values: a DPP wave shift by one lane (wave_shr:1), not a ds_bpermute round trip through LDS
            const int nl_cur_s = __builtin_amdgcn_update_dpp(0, cur_s, 0x138, 0xF, 0xF, false), nl_cur_r = __builtin_amdgcn_update_dpp(0, cur_r, 0x138, 0xF, 0xF, false),
                      nl_prev_s = __builtin_amdgcn_update_dpp(0, prev_s, 0x138, 0xF, 0xF, false);
            const int i = t - lane;
            const unsigned char a_cur = a_nxt;
            a_nxt = (i + 1 >= 1 && i + 1 <= m) ? a[i] : 0;
            if (col_ok && i >= 1 && i <= m) {
                int left_s, left_r, diag;
                if (lane == 0) {
                    if (firstblk) { left_s = -2 - i; left_r = -131072; diag = i == 1 ? 0 : -2 - (i - 1); }
                    else { left_s = colS[i]; left_r = colR[i]; diag = i == 1 ? -2 - (j0 - 1) : colS[i - 1]; }
                } else { left_s = nl_cur_s; left_r = nl_cur_r; diag = i == 1 ? -2 - (j - 1) : nl_prev_s; }
                const uint8_t ca = d_nt4(a_cur);
                int x = left_r - 1, y = left_s - 3;
                const int r = x > y ? x : y;
                x = up_t - 1; y = up_s - 3;
                const int tt = x > y ? x : y;
                const int sv = d_tr2(d_max3(diag + (ca == cb ? 3 : -3), r, tt));
                acc |= ((sv == r ? 1u : 0u) | (sv == tt ? 2u : 0u)) << (((i - 1) & 15) << 1);
                if (((i - 1) & 15) == 15 || i == m) { tb[(size_t)(j - 1) * RW + ((i - 1) >> 4)] = acc; acc = 0; }
                prev_s = cur_s; cur_s = sv; cur_r = r; up_s = sv; up_t = tt;
                if (!lastblk && lane == 63) { colS[i] = sv; colR[i] = r; }
            }
        }
    }
    __syncthreads();                                       // the owner reads tb next
}

// ---------------------------------------------------------------------------------------------
// nw_alignment of up to EIGHT pairs at once: the wave's 64 lanes form 8 groups of 8, a group aligns one
// pair of n <= 64 columns.  Lane k of a group owns the c = ceil(n / 8) columns [k*c, (k+1)*c) for every
// row (their s/t values of the previous row stay in registers, as in d_nw's strips) and the group is a
// software pipeline along the rows: at step t lane k computes row t - k, taking the three values its
// first column needs from lane k-1 (which computed that row one step earlier) by a DPP row shift.
// A 35 x 35 pair costs 42 steps of ~170 instructions for 8 pairs, against 70 steps of ~45 for ONE pair in
// d_nw_coop: the per-phase profile (profiles/r01/i_k_report_class_profile.txt) had the one-pair-at-a-time
// loop at 60 % of the large-pair class.  Same output as d_nw_coop: traceback bits, column-major, in the
// owner lane's scratch.  All 64 lanes call it; the arguments are the GROUP's pair (has = false: none).
// NWG_LANES = 16 (four pairs at a time, up to 128 columns; a DPP row is 16 lanes, so the same row shift serves): the 65 .. 128-column pairs of 2x151 reads -- the
// gap between two exons' seeds, a long pair next to an indel -- went through d_nw_coop one after the other before (profiles/r05/al_k_report_by_class_and_phase.txt:
// the class of candidates that waited for re-seeding spent 3.8 M cycles per 64 of them there).
// ---------------------------------------------------------------------------------------------
#define NWG_MAXN  64
#define NWG_MAXN16 128
template <int NWG_LANES>
__device__ inline void d_nw_group(const DIndex &ix, bool has, const unsigned char *a, int m, int64_t gPos, int n, unsigned char *ows, const WSLayout &L, int lane)
{
    static_assert(NWG_LANES == 8 || NWG_LANES == 16, "a group is half a DPP row or a whole one");
    const int k = lane & (NWG_LANES - 1);
    const int c = has ? (n + NWG_LANES - 1) / NWG_LANES : 0;
    const int j_first = k * c + 1;
    int cn = has ? n - k * c : 0;                                  // this lane's valid columns
    cn = cn < 0 ? 0 : (cn > c ? c : cn);
    int steps = has ? m + NWG_LANES - 1 : 0, cmax = c;
    for (int o = 32; o > 0; o >>= 1) {
        const int v = __shfl_xor(steps, o, 64); steps = v > steps ? v : steps;
        const int w = __shfl_xor(cmax, o, 64); cmax = w > cmax ? w : cmax;
    }
    uint32_t *tb = (uint32_t *)(ows + L.nwbits_off);
    const int RW = (m + 15) >> 4;
    int up_s[8], up_t[8];
    uint32_t acc[8], cbp = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int j = j_first + q;
        up_s[q] = -2 - j; up_t[q] = -131072; acc[q] = 0;
        const uint32_t cb = q < cn ? (uint32_t)d_nt4((unsigned char)d_refchar(ix, gPos + j - 1)) : 7u;
        cbp |= cb << (4 * q);
    }
    int cur_s = 0, cur_r = 0, prev_s = 0;                          // s, r of this lane's LAST column in the row of the previous step; s one row earlier
    unsigned char a_nxt = (has && 1 - k >= 1 && 1 - k <= m) ? a[1 - k - 1] : 0;
    for (int t = 1; t <= steps; t++) {
        const int nl_cur_s = __builtin_amdgcn_update_dpp(0, cur_s, 0x111, 0xF, 0xF, false), nl_cur_r = __builtin_amdgcn_update_dpp(0, cur_r, 0x111, 0xF, 0xF, false),
                  nl_prev_s = __builtin_amdgcn_update_dpp(0, prev_s, 0x111, 0xF, 0xF, false);      // row_shr:1 (lane 8 of a row ignores what it gets: k == 0)
        const int i = t - k;
        const unsigned char a_cur = a_nxt;
        a_nxt = (has && i + 1 >= 1 && i + 1 <= m) ? a[i] : 0;
        if (cn > 0 && i >= 1 && i <= m) {
            int left_s, left_r, diag;
            if (k == 0) { left_s = -2 - i; left_r = -131072; diag = i == 1 ? 0 : -2 - (i - 1); }
            else { left_s = nl_cur_s; left_r = nl_cur_r; diag = i == 1 ? -2 - (j_first - 1) : nl_prev_s; }
            const uint32_t ca = d_nt4(a_cur);
            const int sh = ((i - 1) & 15) << 1;
            int new_prev = prev_s;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q >= cmax) break;                              // wave-uniform
                if (q < cn) {
                    int x = left_r - 1, y = left_s - 3;
                    const int r = x > y ? x : y;
                    x = up_t[q] - 1; y = up_s[q] - 3;
                    const int tt = x > y ? x : y;
                    const int sv = d_tr2(d_max3(diag + (ca == ((cbp >> (4 * q)) & 15u) ? 3 : -3), r, tt));
                    acc[q] |= ((sv == r ? 1u : 0u) | (sv == tt ? 2u : 0u)) << sh;
                    diag = up_s[q];
                    new_prev = diag;                               // after the last valid column: s[i-1][last]
                    up_s[q] = sv; up_t[q] = tt;
                    left_s = sv; left_r = r;
                }
            }
            prev_s = new_prev; cur_s = left_s; cur_r = left_r;
            if (sh == 30 || i == m) {
#pragma unroll
                for (int q = 0; q < 8; q++) if (q < cn) { tb[(size_t)(j_first + q - 1) * RW + ((i - 1) >> 4)] = acc[q]; acc[q] = 0; }
            }
        }
    }
    __syncthreads();                                               // the owners read tb next
}

// Wave-wide nw_alignment service: every lane may ask for one alignment (has; read characters a[0..m), genome
// gPos..gPos+n); requests of <= 64 columns run eight at a time, those of <= 128 four at a time (d_nw_group<8>, <16>), wider ones one after the other
// (d_nw_coop).  Traceback bits land in each requesting lane's own scratch.  Called by all 64 lanes.
__device__ inline void d_nw_wave(LaneCtx &cx, bool has, const unsigned char *a, int m, int64_t gPos, int n, int lane, bool all_wide = false)
{
    if (!__ballot(has)) return;
    const DIndex &ix = *cx.ix;
    const unsigned long long ap = (unsigned long long)a, wp = (unsigned long long)cx.ws;
    unsigned long long todo = __ballot(has && n <= NWG_MAXN && !all_wide);      // all_wide (dg_probe_nw_mode 3 only): every pair takes the whole-wave form
    while (todo) {
        int own = -1;
        for (int q = 0; q < 64 / 8; q++) {                   // group q takes the q-th requester
            const int b = todo ? __ffsll((long long)todo) - 1 : -1;
            if (q == (lane >> 3)) own = b;
            todo &= todo - 1;                                // (0 stays 0)
        }
        const int src = own < 0 ? 0 : own;
        const unsigned long long a_o = ((unsigned long long)(uint32_t)__shfl((int)(ap >> 32), src, 64) << 32) | (uint32_t)__shfl((int)ap, src, 64);
        const unsigned long long w_o = ((unsigned long long)(uint32_t)__shfl((int)(wp >> 32), src, 64) << 32) | (uint32_t)__shfl((int)wp, src, 64);
        const unsigned long long g_o = ((unsigned long long)(uint32_t)__shfl((int)((unsigned long long)gPos >> 32), src, 64) << 32) | (uint32_t)__shfl((int)gPos, src, 64);
        const int m_o = __shfl(m, src, 64), n_o = __shfl(n, src, 64);
        d_nw_group<8>(ix, own >= 0, (const unsigned char *)a_o, m_o, (int64_t)g_o, n_o, (unsigned char *)w_o, *cx.L, lane);
    }
    todo = __ballot(has && n > NWG_MAXN && n <= NWG_MAXN16 && !all_wide);         // 65 .. 128 columns: four at a time, sixteen lanes each
    while (todo) {
        int own = -1;
        for (int q = 0; q < 64 / 16; q++) {
            const int b = todo ? __ffsll((long long)todo) - 1 : -1;
            if (q == (lane >> 4)) own = b;
            todo &= todo - 1;
        }
        const int src = own < 0 ? 0 : own;
        const unsigned long long a_o = ((unsigned long long)(uint32_t)__shfl((int)(ap >> 32), src, 64) << 32) | (uint32_t)__shfl((int)ap, src, 64);
        const unsigned long long w_o = ((unsigned long long)(uint32_t)__shfl((int)(wp >> 32), src, 64) << 32) | (uint32_t)__shfl((int)wp, src, 64);
        const unsigned long long g_o = ((unsigned long long)(uint32_t)__shfl((int)((unsigned long long)gPos >> 32), src, 64) << 32) | (uint32_t)__shfl((int)gPos, src, 64);
        const int m_o = __shfl(m, src, 64), n_o = __shfl(n, src, 64);
        d_nw_group<16>(ix, own >= 0, (const unsigned char *)a_o, m_o, (int64_t)g_o, n_o, (unsigned char *)w_o, *cx.L, lane);
    }
    todo = __ballot(has && (n > NWG_MAXN16 || all_wide));
    while (todo) {
        const int owner = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const unsigned long long a_o = ((unsigned long long)(uint32_t)__shfl((int)(ap >> 32), owner, 64) << 32) | (uint32_t)__shfl((int)ap, owner, 64);
        const unsigned long long w_o = ((unsigned long long)(uint32_t)__shfl((int)(wp >> 32), owner, 64) << 32) | (uint32_t)__shfl((int)wp, owner, 64);
        const unsigned long long g_o = ((unsigned long long)(uint32_t)__shfl((int)((unsigned long long)gPos >> 32), owner, 64) << 32) | (uint32_t)__shfl((int)gPos, owner, 64);
        const int m_o = __shfl(m, owner, 64), n_o = __shfl(n, owner, 64);
        d_nw_coop(ix, (const unsigned char *)a_o, m_o, (int64_t)g_o, n_o, (unsigned char *)w_o, *cx.L, lane);
    }
}

// traceback (nw_alignment.cpp:61-74) from the column-major bits of d_nw_coop into the two gapped strings
__host__ __device__ inline int d_tb_traceback(LaneCtx &cx, const char *a, int m, const char *b, int n, char *oa, char *ob)
{
    const uint32_t *tb = (const uint32_t *)(cx.ws + cx.L->nwbits_off);
    const int RW = (m + 15) >> 4;
    cx.n_nw++; cx.nw_cells += (unsigned long long)m * (unsigned long long)n;
    int i = m, j = n, k = 0;
    while (i > 0 || j > 0) {
        uint32_t fl;
        if (i == 0) fl = 1;
        else if (j == 0) fl = 2;
        else fl = (tb[(size_t)(j - 1) * RW + ((i - 1) >> 4)] >> (((i - 1) & 15) << 1)) & 3u;
        if (fl & 1u) { oa[k] = '-'; ob[k] = b[j - 1]; j--; }
        else if (fl & 2u) { oa[k] = a[i - 1]; ob[k] = '-'; i--; }
        else { oa[k] = a[i - 1]; ob[k] = b[j - 1]; i--; j--; }
        k++;
    }
    for (int p = 0, q = k - 1; p < q; p++, q--) {
        char c = oa[p]; oa[p] = oa[q]; oa[q] = c;
        c = ob[p]; ob[p] = ob[q]; ob[q] = c;
    }
    return k;
}

// ---------------------------------------------------------------------------------------------
// The same traceback WITHOUT the strings.  d_tb_traceback pays a dependent load from the lane's global scratch per step (the bits are column-major and the
// path changes column almost every step), then the passes over the two gapped strings -- reversal, CheckLocalAlignmentQuality, AddNewCigarElements, or
// FillGapsBetweenAdjacentSeeds' five -- pay one per character again, because every iteration branches on what it loaded: for a 100-column alignment ~0.25 M
// cycles of one lane waiting (profiles/r05/as_*: "assemble" 0.6 M cycles per 64 candidates with a large pair on the spliced shape).  TbWalk keeps a window
// of the bits in the lane's LDS slice -- the words of TBW_COLS columns at the walk's block of 16 rows, fetched together: one memory latency per window
// instead of one per step -- and the two sequences as 8-character words in registers (one fetch per 8 steps each); what the callers want from the
// alignment (CIGAR runs, identical columns per read position, genome bases covered) is accumulated while walking.
// ---------------------------------------------------------------------------------------------
#define TBW_COLS 24
struct TbWalk {
    const uint32_t *tb; int RW;          // bits of cell (i, j): tb[(j - 1) * RW + ((i - 1) >> 4)] >> (((i - 1) & 15) << 1)
    uint32_t *win;                       // win[k] = the word of column jtop - k at row block rb (0 for columns < 1)
    int rb, jtop;
    const unsigned char *a; int64_t gpos; const DIndex *ix;
    uint64_t a8, g8; int ab, gb;         // characters 8 ab .. 8 ab + 7 of the read segment / of the genome segment (-1: none yet)
};
__host__ __device__ inline void tbw_init(TbWalk &w, LaneCtx &cx, const unsigned char *a, int m, int64_t gpos)
{
    w.tb = (const uint32_t *)(cx.ws + cx.L->nwbits_off); w.RW = (m + 15) >> 4; w.win = cx.lds; w.rb = -1; w.jtop = 0;
    w.a = a; w.gpos = gpos; w.ix = cx.ix; w.ab = w.gb = -1; w.a8 = w.g8 = 0;
}
__host__ __device__ inline uint32_t tbw_flag(TbWalk &w, int i, int j)        // i, j >= 1
{
    const int rb = (i - 1) >> 4;
    if (rb != w.rb || w.jtop - j >= TBW_COLS || j > w.jtop) {
        w.rb = rb; w.jtop = j;
#pragma unroll 8
        for (int k = 0; k < TBW_COLS; k++) { const int c = j - k; w.win[k] = c >= 1 ? w.tb[(size_t)(c - 1) * w.RW + rb] : 0u; }
    }
    return (w.win[w.jtop - j] >> (((i - 1) & 15) << 1)) & 3u;
}
__host__ __device__ inline unsigned char tbw_read_char(TbWalk &w, int i)     // a[i - 1]
{
    const int b = (i - 1) >> 3;
    if (b != w.ab) { w.ab = b; const uint2 v = *(const uint2_a1 *)(w.a + 8 * b); w.a8 = d_u64(v.x, v.y); }
    return (unsigned char)(w.a8 >> (((i - 1) & 7) << 3));
}
__host__ __device__ inline unsigned char tbw_genome_char(TbWalk &w, int j)   // RefSequence[gpos + j - 1]
{
    const int b = (j - 1) >> 3;
    if (b != w.gb) { w.gb = b; w.g8 = d_ref8(*w.ix, w.gpos + 8 * b); }
    return (unsigned char)(w.g8 >> (((j - 1) & 7) << 3));
}
// a literal '-' among the m read characters at a: there the character, not the traceback, decides what AddNewCigarElements makes of a column (string path)
__host__ __device__ inline bool d_has_dash(const unsigned char *a, int m)
{
    const uint64_t k = 0x2D2D2D2D2D2D2D2Dull, o = 0x0101010101010101ull, h = 0x8080808080808080ull;
    uint64_t any = 0;
    for (int w = 0; w < m; w += 8) {
        const uint2 rw = *(const uint2_a1 *)(a + w);
        const int r = m - w;
        const uint64_t z = d_u64(rw.x, rw.y) ^ k;
        any |= (z - o) & ~z & h & (r >= 8 ? ~0ull : ((1ull << (8 * r)) - 1ull));
    }
    return any != 0;
}

// ProcessNormal/Head/TailSequencePair (tools.cpp:130-164,203-300) for a pair whose nw_alignment bits lie in the lane's scratch (d_nw_wave), the caller having
// established that the pair reaches nw_alignment (d_big_needs_nw) and holds no literal '-' in the read: the CIGAR runs, the identical columns and what
// CheckLocalAlignmentQuality counts come from ONE walk; the runs land behind cig[nc] last one first and are turned round there.
__host__ __device__ inline int d_process_pair_tb(LaneCtx &cx, DSeed &sp, int mode, uint32_t *cig, int &nc)
{
    const int m = sp.rLen, n = sp.gLen;
    cx.n_nw++; cx.nw_cells += (unsigned long long)m * (unsigned long long)n;
    TbWalk w; tbw_init(w, cx, cx.seq + sp.rPos, m, sp.gPos);
    uint32_t *run = cig + nc;
    int R = 0, i = m, j = n, score = 0, nn = 0, cnt = 0;
    uint32_t state = 99;
    while (i > 0 || j > 0) {
        const uint32_t fl = i == 0 ? 1u : (j == 0 ? 2u : tbw_flag(w, i, j));
        uint32_t st;
        if (fl & 1u) { st = OP_D; j--; }
        else if (fl & 2u) { st = OP_I; i--; }
        else { st = OP_M; nn++; if (tbw_read_char(w, i) == tbw_genome_char(w, j)) score++; i--; j--; }
        if (st == state) cnt++;
        else { if (cnt > 0) run[R++] = CIG(cnt, state); cnt = 1; state = st; }
    }
    if (cnt > 0) run[R++] = CIG(cnt, state);
    for (int p = 0, q = R - 1; p < q; p++, q--) { const uint32_t t = run[p]; run[p] = run[q]; run[q] = t; }     // first run first
    if (mode == 2) { nc += R; return score; }
    const int mis = nn - score;
    if (R >= 4 || (mis >= 3 && mis >= (int)(nn * 0.3))) { cig[nc++] = CIG(sp.rLen, OP_S); return 0; }           // CheckLocalAlignmentQuality :166-201 (its st = the runs)
    if (mode == 0) {
        int k = 0, wr = 0;
        if (k < R && (run[k] & 15u) == OP_D) { const int p = (int)(run[k] >> 4); sp.gPos += p; sp.gLen -= p; k++; }
        if (k < R && (run[k] & 15u) == OP_I) { const int p = (int)(run[k] >> 4); sp.rPos += p; sp.rLen -= p; run[wr++] = CIG(p, OP_S); k++; }
        for (; k < R; k++) run[wr++] = run[k];                 // (wr <= k: nothing is overwritten before it is read)
        nc += wr;
        return score;
    }
    int e = R, c = 0;
    if (e > 0 && (run[e - 1] & 15u) == OP_D) { sp.gLen -= (int)(run[e - 1] >> 4); e--; }
    if (e > 0 && (run[e - 1] & 15u) == OP_I) { c = (int)(run[e - 1] >> 4); sp.rLen -= c; e--; }
    if (c > 0) run[e++] = CIG(c, OP_S);
    nc += e;
    return score;
}

// FillGapsBetweenAdjacentSeeds (called by SeedExtension :577-594) on the gapped strings of its two nw_alignment calls, in three steps the two forms below share
// with the host-side check (tests/native/gap_checks.hip).  Right: the read gap against the genome behind the left seed; the read bases its alignment leaves
// without a genome base at the end meet the genome that follows the window.  Rv[p] = identical columns among those of the gap's first p read bases.
__host__ __device__ inline void d_gap_right_strings(const DIndex &ix, const char *f1, char *f2, int len, int64_t g_after, int *Rv)
{
    int q = len - 1;
    while (q >= 0 && f2[q] == '-') q--;
    int64_t gp = g_after;
    for (q += 1; q < len; q++, gp++) f2[q] = d_refchar(ix, gp);
    int p = 0, sc = 0;
    for (q = 0; q < len; q++) { if (f1[q] == f2[q]) sc++; if (f1[q] != '-') p++; Rv[p] = sc; }
}
// Left: the read gap against the genome in front of the right seed, read from its end.  Lv[q] = identical columns among those of the gap's read bases q + 1 .. rGaps.
__host__ __device__ inline void d_gap_left_strings(const DIndex &ix, const char *f3, char *f4, int len3, int64_t g_first, int rGaps, int *Lv)
{
    int q = 0;
    while (q < len3 && f4[q] == '-') q++;
    int64_t gp = g_first;
    for (q -= 1; q >= 0; q--, gp--) f4[q] = d_refchar(ix, gp);
    int p = 0, sc = 0;
    for (q = len3 - 1; q >= 0; q--) { if (f3[q] == f4[q]) sc++; if (f3[q] != '-') p++; Lv[rGaps - p] = sc; }
}
// The split point with the most identical columns on both sides, and how much genome either side then covers
__host__ __device__ inline void d_gap_split_strings(const DParams &pr, const int *Rv, const int *Lv, int rGaps, const char *f1, const char *f2, const char *f3, const char *f4, int len3,
                                                    int &bp, int &right_ext, int &left_ext)
{
    int max_score = 0, p, q;
    bp = 0;
    for (q = 0; q <= rGaps; q++) { const int v = Rv[q] + Lv[q]; if (v > max_score) { max_score = v; bp = q; } }
    right_ext = left_ext = 0;
    if (!(max_score < (int)(rGaps * 0.8) || (rGaps - max_score) > pr.max_mismatch)) {
        for (p = bp, q = 0; p > 0; q++) { if (f1[q] != '-') p--; if (f2[q] != '-') right_ext++; }
        for (p = rGaps - bp, q = len3 - 1; p > 0; q--) { if (f3[q] != '-') p--; if (f4[q] != '-') left_ext++; }
    }
}

// The three steps on the traceback bits of a wide read gap (TbWalk) instead of the strings.  RJ[p], p = 0 .. m: low half = Rv[p], high half = the genome bases
// -- those beyond the window included -- that the columns up to the one of read base p cover (what the right_ext loop counts for bp = p); LJ[q]: low half =
// Lv[q], high half = the genome bases covered from the column of read base q + 1 to the end (the left_ext loop for bp = q).
__host__ __device__ inline void d_gap_right_tb(LaneCtx &cx, const unsigned char *rdp, int m, int64_t g_right, uint32_t *RJ)
{
    cx.n_nw++; cx.nw_cells += (unsigned long long)m * (unsigned long long)m;
    TbWalk w; tbw_init(w, cx, rdp, m, g_right);
    int i = m, j = m, T = 0;
    while (i > 0) {                                              // the read bases the alignment leaves without a genome base at the end ...
        const uint32_t fl = j == 0 ? 2u : tbw_flag(w, i, j);
        if ((fl & 1u) || !(fl & 2u)) break;
        i--; T++;
    }
    for (int u = 0; u < T; u++)                                  // ... meet the genome that follows the window
        RJ[m - T + 1 + u] = (rdp[m - T + u] == (unsigned char)d_refchar(*cx.ix, g_right + m + u) ? 1u : 0u) | ((uint32_t)(m + u + 1) << 16);
    while (i > 0 || j > 0) {
        const uint32_t fl = i == 0 ? 1u : (j == 0 ? 2u : tbw_flag(w, i, j));
        if (fl & 1u) j--;
        else if (fl & 2u) { RJ[i] = (uint32_t)j << 16; i--; }
        else { RJ[i] = (tbw_read_char(w, i) == tbw_genome_char(w, j) ? 1u : 0u) | ((uint32_t)j << 16); i--; j--; }
    }
    RJ[0] = 0;
    uint32_t acc = 0;
    for (int p = 1; p <= m; p++) { const uint32_t v = RJ[p]; acc += v & 1u; RJ[p] = (v & 0xFFFF0000u) | acc; }
}
__host__ __device__ inline void d_gap_left_tb(LaneCtx &cx, const unsigned char *rdp, int m, int64_t g_left, uint32_t *LJ)
{
    cx.n_nw++; cx.nw_cells += (unsigned long long)m * (unsigned long long)m;
    TbWalk w; tbw_init(w, cx, rdp, m, g_left);
    int i = m, j = m, H = -1;
    uint32_t sc = 0;
    LJ[m] = 0;
    while (i > 0 || j > 0) {
        const uint32_t fl = i == 0 ? 1u : (j == 0 ? 2u : tbw_flag(w, i, j));
        if (fl & 1u) { j--; continue; }
        uint32_t same, cov;
        if (fl & 2u) {
            if (j == 0) {                                        // the read bases without a genome base at the start meet the genome from the window's first base backwards
                if (H < 0) H = i;
                same = rdp[i - 1] == (unsigned char)d_refchar(*cx.ix, g_left - (H - i)) ? 1u : 0u; cov = (uint32_t)(m + H - i + 1);
            } else { same = 0u; cov = (uint32_t)(m - j); }
            sc += same; LJ[i - 1] = sc | (cov << 16); i--;
        } else {
            same = tbw_read_char(w, i) == tbw_genome_char(w, j) ? 1u : 0u; cov = (uint32_t)(m - j + 1);
            sc += same; LJ[i - 1] = sc | (cov << 16); i--; j--;
        }
    }
}
__host__ __device__ inline void d_gap_split_tb(const DParams &pr, const uint32_t *RJ, const uint32_t *LJ, int rGaps, int &bp, int &right_ext, int &left_ext)
{
    int max_score = 0;
    bp = 0;
    for (int q = 0; q <= rGaps; q++) { const int v = (int)(RJ[q] & 0xFFFFu) + (int)(LJ[q] & 0xFFFFu); if (v > max_score) { max_score = v; bp = q; } }
    right_ext = left_ext = 0;
    if (!(max_score < (int)(rGaps * 0.8) || (rGaps - max_score) > pr.max_mismatch)) {
        if (bp > 0) right_ext = (int)(RJ[bp] >> 16);
        if (rGaps - bp > 0) left_ext = (int)(LJ[bp] >> 16);
    }
}

// The same three steps for a read gap of at most PM_MAX bases WITHOUT the strings: both alignments by the lane itself (d_pair_nw: registers and the lane's LDS slice,
// result = a column list in three registers), Rv / Lv as two bit masks over the gap's read positions (Rv[q] = the set bits below q, Lv[q] = those from q up), the
// genome bases beyond the window fetched where a column needs one.  At a splice junction the two exact seeds around the intron end where the read stops matching
// the intron, so the gap between them is a few bases: through d_nw_wave every such gap cost a wave-wide alignment pass, a traceback through bits in the lane's
// global scratch (one dependent load per step) and five passes over strings in that scratch (profiles/r05/ao_*: 0.42 M cycles per 64 spliced candidates).
// The caller keeps read gaps that hold a literal '-' on the string path (there the character decides what a column is).
__host__ __device__ inline void d_gap_small(LaneCtx &cx, const unsigned char *rdp, int rGaps, int64_t g_right /* first base behind the left seed */, int64_t g_left /* first base of the window in front of the right seed */,
                                            int &bp, int &right_ext, int &left_ext)
{
    const DIndex &ix = *cx.ix;
    auto keep = [](int len, int w) -> uint64_t { const int r = len - 8 * w; return r >= 8 ? ~0ull : (r <= 0 ? 0ull : ((1ull << (8 * r)) - 1ull)); };
    auto code_k = [](const ColList &c, int k) -> uint32_t { const uint64_t w = k < 16 ? c.w0 : (k < 32 ? c.w1 : c.w2); return (uint32_t)(w >> ((k & 15) << 2)) & 7u; };   // k counts from the LAST column
    PairStr ps;
    const uint2 r0 = *(const uint2_a1 *)rdp, r1 = *(const uint2_a1 *)(rdp + 8), r2 = *(const uint2_a1 *)(rdp + 16);
    ps.A0 = d_u64(r0.x, r0.y) & keep(rGaps, 0); ps.A1 = d_u64(r1.x, r1.y) & keep(rGaps, 1); ps.A2 = d_u64(r2.x, r2.y) & keep(rGaps, 2);
    auto genome = [&](int64_t g0) {
        ps.B0 = d_ref8(ix, g0) & keep(rGaps, 0);
        ps.B1 = rGaps > 8 ? d_ref8(ix, g0 + 8) & keep(rGaps, 1) : 0ull;
        ps.B2 = rGaps > 16 ? d_ref8(ix, g0 + 16) & keep(rGaps, 2) : 0ull;
    };
    // ---- right ----
    ColList c1;
    genome(g_right);
    d_pair_nw(cx, rGaps, rGaps, ps, c1);
    int T = 0;                                                        // columns at the end without a genome base
    while (T < c1.K && (code_k(c1, T) & 3u) == 2u) T++;
    uint32_t RM = 0;
    for (int k = c1.K - 1, ri = 0; k >= 0; k--) {                     // first column first
        const uint32_t cd = code_k(c1, k), ty = cd & 3u;
        if (ty == 1u) continue;
        bool same = (cd & 4u) != 0u;                                  // (set on columns with both bases only)
        if (ty == 2u && k < T) same = (unsigned char)d_byte3(ps.A0, ps.A1, ps.A2, ri) == (unsigned char)d_refchar(ix, g_right + rGaps + (T - 1 - k));
        if (same) RM |= 1u << ri;
        ri++;
    }
    // ---- left ----
    ColList c3;
    genome(g_left);
    d_pair_nw(cx, rGaps, rGaps, ps, c3);
    int H = 0;                                                        // columns at the start without a genome base
    while (H < c3.K && (code_k(c3, c3.K - 1 - H) & 3u) == 2u) H++;
    uint32_t LM = 0;
    for (int k = 0, ri = rGaps - 1; k < c3.K; k++) {                  // last column first
        const uint32_t cd = code_k(c3, k), ty = cd & 3u;
        if (ty == 1u) continue;
        bool same = (cd & 4u) != 0u;
        const int pcol = c3.K - 1 - k;                                // the column's number from the start
        if (ty == 2u && pcol < H) same = (unsigned char)d_byte3(ps.A0, ps.A1, ps.A2, ri) == (unsigned char)d_refchar(ix, g_left - (H - 1 - pcol));
        if (same) LM |= 1u << ri;
        ri--;
    }
    // ---- the split ----
    int max_score = 0;
    bp = 0;
    for (int q = 0; q <= rGaps; q++) {
        const int v = __builtin_popcount(RM & ((1u << q) - 1u)) + __builtin_popcount(LM >> q);       // (q <= 24: the shifts are defined)
        if (v > max_score) { max_score = v; bp = q; }
    }
    right_ext = left_ext = 0;
    if (!(max_score < (int)(rGaps * 0.8) || (rGaps - max_score) > cx.pr->max_mismatch)) {
        for (int p = bp, k = c1.K - 1; p > 0; k--) { const uint32_t ty = code_k(c1, k) & 3u; if (ty != 1u) p--; if (ty != 2u || k < T) right_ext++; }
        for (int p = rGaps - bp, k = 0; p > 0; k++) { const uint32_t ty = code_k(c3, k) & 3u; if (ty != 1u) p--; if (ty != 2u || c3.K - 1 - k < H) left_ext++; }
    }
}

// SeedExtension :577-594 for the wave's lanes together (live = this lane has a candidate to extend).  The two nw_alignment calls of a
// FillGapsBetweenAdjacentSeeds whose read gap is longer than PM_MAX bases are served by d_nw_wave; the lane then reads its traceback, exactly the
// strings d_nw would have produced.  (On spliced reads these rGaps x rGaps alignments, one lane at a time, were most of k_report, and on plain reads
// the tail of the job reads: one lane, 1.9 M cycles.)  Shorter gaps -- most of them: see d_gap_small -- never leave the lane.
__device__ inline int d_seed_extension_wave(LaneCtx &cx, bool live, DSeed *s, int n, int lane)
{
    const DIndex &ix = *cx.ix;
    const int num = live ? n : 0;
    int i = 1;
    while (true) {
        bool has = false;
        for (; i < num; i++) {
            const int pd = (int)((s[i].gPos - s[i].rPos) - (s[i - 1].gPos - s[i - 1].rPos));
            if (pd > cx.pr->min_intron && s[i].rPos > s[i - 1].rPos + s[i - 1].rLen) { has = true; break; }
        }
        if (!__ballot(has)) break;
        DSeed Ls = s[has ? i - 1 : 0], Rs = s[has ? i : 0];
        int rGaps = has ? Rs.rPos - (Ls.rPos + Ls.rLen) : 0;
        const unsigned char *rdp = cx.seq + Ls.rPos + Ls.rLen;
        bool small = has && rGaps <= PM_MAX;
        if (small) {                                                  // a literal '-' in the read gap: the string path
            const uint64_t k = 0x2D2D2D2D2D2D2D2Dull, o = 0x0101010101010101ull, h = 0x8080808080808080ull;
            for (int w = 0; w < rGaps; w += 8) {
                const uint2 rw = *(const uint2_a1 *)(rdp + w);
                const int r = rGaps - w;
                const uint64_t z = d_u64(rw.x, rw.y) ^ k, m = r >= 8 ? ~0ull : ((1ull << (8 * r)) - 1ull);
                if ((z - o) & ~z & h & m) small = false;
            }
        }
        const bool wide = has && !small;
        const bool strings = wide && (!cx.lds || d_has_dash(rdp, rGaps));      // a literal '-' in a wide read gap (or a caller without the LDS slice): the gapped strings
        char *g = ws_str(cx, 0), *f1 = ws_str(cx, 1), *f2 = ws_str(cx, 2), *f3 = ws_str(cx, 3), *f4 = ws_str(cx, 4);
        int *Rv = (int *)ws_cig(cx), *Lv = Rv + rGaps + 1;      // the CIGAR scratch is idle at this stage
        int bp = 0, right_ext = 0, left_ext = 0;
        if (small) d_gap_small(cx, rdp, rGaps, Ls.gPos + Ls.gLen, Rs.gPos - rGaps, bp, right_ext, left_ext);
        int len = 0;
        d_nw_wave(cx, wide, rdp, rGaps, Ls.gPos + Ls.gLen, rGaps, lane);
        if (strings) {
            for (int q = 0; q <= rGaps; q++) Rv[q] = Lv[q] = 0;
            d_ref_fill(ix, Ls.gPos + Ls.gLen, rGaps, g);
            len = d_tb_traceback(cx, (const char *)rdp, rGaps, g, rGaps, f1, f2);
            d_gap_right_strings(ix, f1, f2, len, Ls.gPos + Ls.gLen + rGaps, Rv);
        } else if (wide) d_gap_right_tb(cx, rdp, rGaps, Ls.gPos + Ls.gLen, (uint32_t *)Rv);
        d_nw_wave(cx, wide, rdp, rGaps, Rs.gPos - rGaps, rGaps, lane);
        if (strings) {
            d_ref_fill(ix, Rs.gPos - rGaps, rGaps, g);
            const int len3 = d_tb_traceback(cx, (const char *)rdp, rGaps, g, rGaps, f3, f4);
            d_gap_left_strings(ix, f3, f4, len3, Rs.gPos - rGaps, rGaps, Lv);
            d_gap_split_strings(*cx.pr, Rv, Lv, rGaps, f1, f2, f3, f4, len3, bp, right_ext, left_ext);
        } else if (wide) {
            d_gap_left_tb(cx, rdp, rGaps, Rs.gPos - rGaps, (uint32_t *)Lv);
            d_gap_split_tb(*cx.pr, (const uint32_t *)Rv, (const uint32_t *)Lv, rGaps, bp, right_ext, left_ext);
        }
        if (has) {
            if (bp > 0) {
                DSeed x; x.flags = 0; x.rPos = Ls.rPos + Ls.rLen; x.gPos = Ls.gPos + Ls.gLen; x.rLen = bp; x.gLen = right_ext;
                s[n++] = x;
            }
            if ((rGaps -= bp) > 0) {
                DSeed x; x.flags = 0; x.rLen = rGaps; x.gLen = left_ext; x.rPos = Rs.rPos - x.rLen; x.gPos = Rs.gPos - x.gLen;
                s[n++] = x;
            }
            i++;
        }
    }
    if (live && n > num) d_insertion_sort_seeds(s, n);
    return n;
}

// does the string path reach nw_alignment for this pair (tools.cpp:130-164,203-300 up to the call)?
__host__ __device__ inline bool d_big_needs_nw(LaneCtx &cx, const DSeed &sp, int mode)
{
    if (mode == 2 && (sp.gPos - sp.rPos == -1 || sp.rLen == 0 || sp.gLen == 0)) return false;
    if (sp.rLen != sp.gLen) return true;
    int nm = 0;
    const unsigned char *rd = cx.seq + sp.rPos;
    for (int i = 0; i < sp.rLen; i += 8) {
        const uint64_t w = d_ref8(*cx.ix, sp.gPos + i);
        const int e = sp.rLen - i < 8 ? sp.rLen - i : 8;
        for (int k = 0; k < e; k++) if (rd[i + k] != (unsigned char)(w >> (8 * k))) nm++;
    }
    return !(nm <= 2 && nm <= (int)(sp.rLen * 0.2));
}

// mode 0 = head (ProcessHeadSequencePair :203-249), 1 = tail (:251-300), 2 = normal (:130-164)
__host__ __device__ inline int d_process_pair(LaneCtx &cx, DSeed &sp, int mode, uint32_t *cig, int &nc, bool have_tb = false)
{
    const DIndex &ix = *cx.ix;
    if (mode == 2) {
        if (sp.gPos - sp.rPos == -1) { cig[nc++] = CIG(sp.rLen, OP_S); return 0; }
        if (sp.rLen == 0 || sp.gLen == 0) {
            if (sp.rLen > 0) cig[nc++] = CIG(sp.rLen, OP_I);
            else if (sp.gLen > 0) cig[nc++] = CIG(sp.gLen, OP_D);
            return 0;
        }
    }
    const char *rd = (const char *)cx.seq + sp.rPos;
    if (have_tb && cx.lds && !d_has_dash(cx.seq + sp.rPos, sp.rLen)) return d_process_pair_tb(cx, sp, mode, cig, nc);     // (d_big_needs_nw was true: neither early-out below applies)
    char *g = ws_str(cx, 0);
    d_ref_fill(ix, sp.gPos, sp.gLen, g);
    if (sp.rLen == sp.gLen) {
        int n = 0;
        for (int i = 0; i < sp.rLen; i++) if (rd[i] != g[i]) n++;       // CalFragPairMismatchBases :40-47
        if (n <= 2 && n <= (int)(sp.rLen * 0.2)) { cig[nc++] = CIG(sp.rLen, OP_M); return sp.rLen - n; }
    }
    char *o1 = ws_str(cx, 1), *o2 = ws_str(cx, 3);     // each spans two string slots (rLen+gLen chars)
    int len = have_tb ? d_tb_traceback(cx, rd, sp.rLen, g, sp.gLen, o1, o2) : d_nw(cx, rd, sp.rLen, g, sp.gLen, o1, o2);
    if (mode == 2) return d_add_cigar(o1, o2, len, cig, nc);
    if (!d_local_quality(o1, o2, len)) { cig[nc++] = CIG(sp.rLen, OP_S); return 0; }
    if (mode == 0) {
        int p = 0;
        while (p < len && o1[p] == '-') p++;
        if (p > 0) { o1 += p; o2 += p; len -= p; sp.gPos += p; sp.gLen -= p; }
        p = 0;
        while (p < len && o2[p] == '-') p++;
        if (p > 0) { o1 += p; o2 += p; len -= p; sp.rPos += p; sp.rLen -= p; cig[nc++] = CIG(p, OP_S); }
        return d_add_cigar(o1, o2, len, cig, nc);
    }
    int p = len - 1, c = 0;
    while (p >= 0 && o1[p] == '-') { c++; p--; }
    if (c > 0) { len -= c; sp.gLen -= c; }
    p = len - 1; c = 0;
    while (p >= 0 && o2[p] == '-') { c++; p--; }
    if (c > 0) { len -= c; sp.rLen -= c; }
    const int score = d_add_cigar(o1, o2, len, cig, nc);
    if (c > 0) cig[nc++] = CIG(c, OP_S);
    return score;
}

#ifdef DG_PROFILE_CLASSES
#define DG_NPHASE 12
#define DG_NCLS 33
__device__ unsigned long long g_phase[DG_NCLS][DG_NPHASE];
#define PH(k) do { if ((threadIdx.x & 63) == 0) { const long long t_now = clock64(); cx.ph[cx.cls * DG_NPHASE + (k)] += (unsigned long long)(t_now - cx.t_last); cx.t_last = t_now; } } while (0)
#else
#define PH(k) do { } while (0)
#endif
// device-side per-read state (ReadItem_t)
#define CIG_SLOT 8
struct DRead {
    int score, sub_score, mis_num, mapq, CanNum, iBest;
};

// GenMappingReport :1079-1207 for one read per lane.  reports = this read's dg_report_out slots
// (n_rep = max(ncand,1)); cands/seeds = this read's candidates and the global seed array;
// work = global working-seed pool; cigpool/cigtop = bump pool for merged CIGAR ops.
// CALLED BY ALL 64 LANES OF THE WAVE TOGETHER (valid = false: a lane without a read): the candidate loop
// runs to the wave's largest candidate count with per-lane guards, because in its middle the wave
// aligns the lanes' LARGE segment pairs cooperatively, one after the other (d_nw_coop).
// Three lane layouts: one read per lane (cstride = 1, cstart = 0: the lane walks its read's candidates in order; k_pair's host-compiled
// checks and the probes), ONE read for the whole wave with lane = candidate (cstride = 64, cstart = lane), or -- k_report since round 5 --
// one CANDIDATE per lane, whichever read it belongs to (`cands` / `rep` point at that candidate, ncand = 1, park = true).  In the second
// layout the running best/second-best state of :1161-1172 is replayed in candidate order by lane 0 afterwards (rd is valid in lane 0
// only); in the third the candidate's mismatch count is parked in its report's flag and k_finalize replays the read's candidates.
template <typename ReportT>
__device__ inline void d_gen_mapping_report(LaneCtx &cx, bool valid, bool first, DRead &rd, DCand *cands, int ncand, const DJob *jobs,
                                            DSeed *work, ReportT *rep, uint32_t rep_index0, uint32_t *cigpool, unsigned int *cigtop, uint32_t cigcap, int *err,
                                            int cstart = 0, int cstride = 1, bool park = false)
{
    const DIndex &ix = *cx.ix;
    const int lane = threadIdx.x & 63;
    rd.score = rd.iBest = 0;
    if (!valid) ncand = 0;
    rd.CanNum = ncand > 0 ? ncand : 1;
    if (valid && ncand == 0 && cstart == 0) {
        ReportT rp;
        rp.aln_score = 0; rp.sj_type = -1; rp.flag = 0; rp.paired_idx = -1; rp.chr = -1; rp.bdir = 0;
        rp.pos = 0; rp.cigar_off = 0; rp.n_cigar = 0;
        rep[0] = rp;
    }
    int nmax = ncand > cstart ? (ncand - cstart + cstride - 1) / cstride : 0;      // this lane's iterations
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    uint32_t *cig = ws_cig(cx);
    PH(0);
    for (int it = 0; it < nmax; it++) {
        const int i = cstart + it * cstride;
        const bool act = i < ncand;
        // ---- part 1 (per lane): everything up to knowing which segment pairs need nw_alignment ----
        ReportT rp;
        rp.aln_score = 0; rp.sj_type = -1; rp.flag = 0; rp.paired_idx = -1; rp.chr = -1; rp.bdir = 0; rp.pos = 0;
        rp.cigar_off = 0; rp.n_cigar = 0;
        int final_n = 0, num = 0, nq = 0, bigj = -1, npl = 0;
        bool lane_nw_before = false;
        uint32_t nwj = 0;                                 // seed indices (8 bits each) of the first PM_MAXQ small alignments
        uint64_t pcl = 0;                                 // outcome of d_pair_classify for the first 12 pairs, 5 bits each (class | mismatches << 3)
        DSeed *s = work;
        bool go = false, live = false;
        int n = 0;
        if (act) {
            DCand &c = cands[i];
            rp.paired_idx = c.PairedIdx;
            if (c.Score != 0) {
                // the working region already went through k_prep (tandem / translocation clean-up);
                // IdentifyMissingSeeds :685-700: append the seeds k_reseed found, then re-sort
                live = true;
                s = work + c.work_off;
                n = c.n_a;
                const int n0 = n;
                for (int q = 0; q < c.job_count; q++) {
                    const DJob jb = jobs[c.job_first + q];
                    if (jb.found == 1) {
                        DSeed ns; ns.gPos = jb.gPos; ns.rPos = jb.rPos; ns.rLen = ns.gLen = jb.len; ns.flags = SEED_SIMPLE;
                        s[n++] = ns;
                    } else if (jb.found < 0) {               // read gap too long for the cooperative kernel
                        DSeed ns;
                        if (d_reseed(cx, jb.rBegin, jb.rBegin + jb.rl, jb.Lb, jb.Lb + jb.glen, &ns)) s[n++] = ns;
                    }
                }
                if (n > n0) d_insertion_sort_seeds(s, n);
            }
        }
        PH(1);
        n = d_seed_extension_wave(cx, live, s, n, lane);      // wave-wide: the gap-filling alignments of all lanes together
        PH(2);
        if (live) {
            {
                DCand &c = cands[i];
                int2 *vec = (int2 *)(s + n + 1);
                rp.sj_type = c.SJtype = d_check_splice(cx, s, n, vec);
                PH(3);
                n = d_identify_normal_pairs(s, n);
                final_n = num = n;
                go = !(num > 1 && !d_check_coordinate_validity(ix, s, num));
                PH(4);
                if (go) {
                    for (int j = 0; j < num; j++) {
                        const DSeed &sd = s[j];
                        if ((sd.rLen == 0 && sd.gLen == 0) || (sd.flags & SEED_SIMPLE) || j > 254) continue;
                        const int mode = j == 0 ? 0 : (j == num - 1 ? 1 : 2);
                        if (sd.rLen > PM_MAX || sd.gLen > PM_MAX) {              // large: the first one is aligned by the whole wave ...
                            if (bigj < 0 && d_big_needs_nw(cx, sd, mode)) { if (sd.gLen <= 64 * 64 && !lane_nw_before) bigj = j; else lane_nw_before = true; }
                            npl++;                                               // (slot stays PC_GENERIC)
                            continue;
                        }
                        PairStr ps; int nm = 0;
                        const int pc = d_pair_classify(cx, sd, mode, ps, nm);
                        // ... unless a pair in front of it runs the one-lane d_nw in part 2 (a pair with a literal '-' in the read, or one whose genome side is too long for the
                        // wave): d_nw keeps its traceback bits where the wave leaves the large pair's, and part 2 takes the pairs in order -- the large pair's traceback would
                        // read the other alignment's bits (found by test_gpu_literal_dashes_in_read_gaps_and_segment_pairs; such a large pair goes the one-lane way too)
                        if (pc == PC_GENERIC) lane_nw_before = true;
                        if (npl < 12) pcl |= (uint64_t)(pc | ((pc == PC_EQUAL ? nm : 0) << 3)) << (5 * npl);
                        npl++;
                        if (nq < PM_MAXQ && pc == PC_NW) {
                            uint64_t *pin = (uint64_t *)(cx.ws + cx.L->kmer_off) + 16 + 6 * nq;     // the strings, for the alignment pass
                            pin[0] = ps.A0; pin[1] = ps.A1; pin[2] = ps.A2; pin[3] = ps.B0; pin[4] = ps.B1; pin[5] = ps.B2;
                            nwj |= (uint32_t)j << (8 * nq); nq++;
                        }
                    }
                }
            }
        }
        PH(5);
        // ---- the wave aligns the lanes' large pairs: eight at a time (<= 64 columns), the rest one after the other ----
        if (__ballot(bigj >= 0)) {
            const DSeed sd = s[bigj < 0 ? 0 : bigj];         // only a requesting lane's copy is used
            d_nw_wave(cx, bigj >= 0, cx.seq + sd.rPos, sd.rLen, sd.gPos, sd.gLen, lane);
        }
        PH(6);
        // ---- part 2 (per lane): the lanes' small alignments together, then the reference's loop (:1134-1160) ----
        if (act && go) {
            int nc = 1, mis_num = 0, aln = 0;                 // cig[0] is kept free for a leading soft clip
            uint64_t *pmres = (uint64_t *)(cx.ws + cx.L->kmer_off);          // the k-mer list is idle at this stage: 4 u64 per result, then 6 per input
            for (int q = 0; q < nq; q++) {
                const int j = (int)((nwj >> (8 * q)) & 255u);
                PairStr ps; ColList cl;
                const uint64_t *pin = pmres + 16 + 6 * q;
                ps.A0 = pin[0]; ps.A1 = pin[1]; ps.A2 = pin[2]; ps.B0 = pin[3]; ps.B1 = pin[4]; ps.B2 = pin[5];
                d_pair_nw(cx, s[j].rLen, s[j].gLen, ps, cl);
                pmres[4 * q] = cl.w0; pmres[4 * q + 1] = cl.w1; pmres[4 * q + 2] = cl.w2; pmres[4 * q + 3] = (uint64_t)cl.K;
            }
            PH(7);
            int qn = 0, pi = 0;
            for (int j = 0; j < num; j++) {
                DSeed &sd = s[j];
                if (sd.rLen == 0 && sd.gLen == 0) continue;
                int g;
                if (j > 0 && (g = (int)(sd.gPos - (s[j - 1].gPos + s[j - 1].gLen))) > 0) cig[nc++] = CIG(g, OP_N);
                if (sd.flags & SEED_SIMPLE) { cig[nc++] = CIG(sd.rLen, OP_M); aln += sd.rLen; }
                else {
                    const int mode = j == 0 ? 0 : (j == num - 1 ? 1 : 2);
                    int score;
                    if (j == bigj) score = d_process_pair(cx, sd, mode, cig, nc, true);          // traceback bits are ready
                    else if (qn < nq && (int)((nwj >> (8 * qn)) & 255u) == j) {
                        ColList cl; cl.w0 = pmres[4 * qn]; cl.w1 = pmres[4 * qn + 1]; cl.w2 = pmres[4 * qn + 2]; cl.K = (int)pmres[4 * qn + 3];
                        qn++;
                        score = d_pair_finish(cl, sd, mode, cig, nc);
                    } else {
                        PairStr ps; int nm = 0, pc;
                        const int slot = j > 254 ? 99 : pi;      // pairs are numbered as in part 1 (which skips j > 254)
                        if (slot < 12 && ((pcl >> (5 * slot)) & 7u) != PC_NW) { pc = (int)((pcl >> (5 * slot)) & 7u); nm = (int)((pcl >> (5 * slot + 3)) & 3u); }
                        else pc = d_pair_classify(cx, sd, mode, ps, nm);
                        if (pc == PC_EQUAL) { cig[nc++] = CIG(sd.rLen, OP_M); score = sd.rLen - nm; }
                        else if (pc == PC_ONE) { cx.n_nw++; cx.nw_cells += 1; cig[nc++] = CIG(1, OP_M); score = 0; }
                        else if (pc == PC_NW) { ColList cl; d_pair_nw(cx, sd.rLen, sd.gLen, ps, cl); score = d_pair_finish(cl, sd, mode, cig, nc); }   // beyond PM_MAXQ
                        else score = d_process_pair(cx, sd, mode, cig, nc);                                                               // trivial cases and the string path
                    }
                    if (j <= 254) pi++;
                    aln += score;
                    mis_num += sd.rLen - score;
                }
            }
            PH(8);
            int c0 = 1;
            if (num > 0) {
                int j;
                if ((j = s[0].rPos) > 0) { cig[0] = CIG(j, OP_S); c0 = 0; }
                if ((j = cx.rlen - (s[num - 1].rPos + s[num - 1].rLen)) > 0) cig[nc++] = CIG(j, OP_S);
            }
            if (mis_num > cx.pr->max_mismatch || nc - c0 == 0) aln = 0;
            for (int j = c0; j < nc; j++) if ((cig[j] & 15u) == OP_N && (int)(cig[j] >> 4) < cx.pr->min_intron) { aln = 0; break; }   // CheckMinIntronSize :1052
            if (aln > 0) {
                const int64_t gPos = s[0].gPos, end_gPos = s[num - 1].gPos + s[num - 1].gLen - 1;   // GenCoordinateInfo :83-116
                const int lb = d_loc_lower_bound(ix, gPos);
                rp.chr = ix.loc_chr[lb];
                if (gPos < ix.l_pac) { rp.bdir = first ? 1 : 0; rp.pos = gPos + 1 - ix.chr_off[rp.chr]; }
                else { rp.bdir = first ? 0 : 1; rp.pos = ix.loc_key[lb] - end_gPos + 1; }
                if (rp.pos <= 0) aln = 0;
                else {
                    if (gPos >= ix.l_pac) for (int a = c0, b = nc - 1; a < b; a++, b--) { const uint32_t t = cig[a]; cig[a] = cig[b]; cig[b] = t; }
                    // GenerateCIGAR :37-61: merge equal neighbours, drop zero-length runs, in place
                    int m = 0, cnt = 0;
                    uint32_t state = 99;
                    for (int j = c0; j < nc; j++) {
                        const uint32_t op = cig[j] & 15u, ln = cig[j] >> 4;
                        if (op != state) { if (cnt > 0) cig[m++] = CIG(cnt, state); cnt = (int)ln; state = op; }
                        else cnt += (int)ln;
                    }
                    if (cnt > 0) cig[m++] = CIG(cnt, state);
                    // every report owns CIG_SLOT ops of the pool (no atomic: one hot bump counter serialises at ~6 ns per wave
                    // update); the rare longer CIGAR goes to the bump-allocated overflow area behind the slots
                    const unsigned int off = m <= CIG_SLOT ? (rep_index0 + (unsigned int)i) * CIG_SLOT : atomicAdd(cigtop, (unsigned int)m);
                    if (off + (unsigned int)m > cigcap) { atomicMax(err, DG_E_CIGAR); rp.n_cigar = 0; }
                    else { for (int j = 0; j < m; j++) cigpool[off + j] = cig[j]; rp.cigar_off = off; rp.n_cigar = (uint32_t)m; }
                }
                rp.aln_score = aln;
                if (cstride != 1 || park) rp.flag = mis_num;       // parked for the replay (below, or in k_finalize)
                else if (aln > rd.score) { rd.iBest = i; rd.mis_num = mis_num; rd.sub_score = rd.score; rd.score = aln; }
                else if (aln == rd.score) rd.sub_score = rd.score;
            }
        }
        PH(9);
        if (act) { cands[i].final_n = final_n; rep[i] = rp; }
        PH(10);
    }
    if (cstride != 1) {
        __syncthreads();                                           // the other lanes' reports are in memory
        if (lane == 0) {
            for (int i = 0; i < ncand; i++) {
                const int aln = rep[i].aln_score, mis_num = rep[i].flag;
                rep[i].flag = 0;                                   // (also where the reference zeroes aln late, :1157: the parked value must not stay)
                if (aln <= 0) continue;                            // such an aln changes nothing: score and sub_score are 0 or stay
                if (aln > rd.score) { rd.iBest = i; rd.mis_num = mis_num; rd.sub_score = rd.score; rd.score = aln; }
                else if (aln == rd.score) rd.sub_score = rd.score;
            }
        }
    }
}

// UpdateLocalSJMap, Mapping.cpp:532-565: tuples of the best candidate -> bump pool
template <typename SjT>
__device__ inline void d_collect_sj(const DIndex &ix, const DParams &pr, const DCand &c, const DSeed *work, int read_idx,
                                    SjT *sjpool, unsigned int *sjtop, uint32_t sjcap, int32_t &sj_off, int32_t &n_sj, int *err)
{
    if (c.SJtype == -1) return;
    const DSeed *s = work + c.work_off;
    const int64_t L = ix.l_pac;
    int cnt = 0;
    for (int pass = 0; pass < 2; pass++) {
        unsigned int off = 0;
        if (pass == 1) {
            if (cnt == 0) return;
            off = atomicAdd(sjtop, (unsigned int)cnt);
            if (off + (unsigned int)cnt > sjcap) { atomicMax(err, DG_E_SJ); return; }
            sj_off = (int32_t)off; n_sj = cnt;
        }
        int k = 0;
        for (int i = 1; i < c.final_n; i++) {
            if (!(s[i].flags & SEED_ACCEPTOR)) continue;
            int64_t g1, g2;
            if (c.PosDiff < L) { g1 = s[i - 1].gPos + s[i - 1].gLen; g2 = s[i].gPos - 1; }
            else { g1 = 2 * L - s[i].gPos; g2 = 2 * L - 1 - (s[i - 1].gPos + s[i - 1].gLen); }
            int64_t d = g2 - g1; if (d < 0) d = -d;
            if (d < pr.min_intron) continue;
            if (pass == 1) { SjT t; t.g1 = g1; t.g2 = g2; t.type = c.SJtype; t.read_idx = read_idx; sjpool[off + k] = t; }
            k++;
        }
        cnt = k;
    }
}
