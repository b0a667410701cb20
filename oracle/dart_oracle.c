/* oracle/dart_oracle.c -- TEST INFRASTRUCTURE (the parity checker), not product code.
 *
 * Plain-C CPU restatement of DART v1.4.6's per-read mapping path.  Citations are
 * path:line under /root/reference/src.  Deliberate, documented deviation: ReadItem_t's
 * sub_score / mis_num / mapq start at 0 (the reference leaves them uninitialised: SURVEY F6).
 *
 * Parity status: PINNED -- checked against the reference's own object code (oracle/_ref/
 * ref_harness, see tests/test_oracle_vs_ref.py) and against tests/golden/ fixtures that were
 * produced by that harness (tests/golden/make_golden.py).
 */
#define _GNU_SOURCE
#include "dart_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ------------------------------------------------------------------------------------------
 * small containers
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t simple, acceptor;   /* bSimple, bAcceptorSite   structure.h:106-115 */
    int rPos, rLen, gLen;
    int64_t gPos, PosDiff;
} seed_t;

typedef struct { seed_t *a; int n, m; } seedvec;

static void sv_reserve(seedvec *v, int need)
{
    if (need > v->m) {
        v->m = need < 8 ? 8 : (need * 3 / 2 + 4);
        v->a = (seed_t *)realloc(v->a, (size_t)v->m * sizeof(seed_t));
    }
}
static void sv_push(seedvec *v, const seed_t *s) { sv_reserve(v, v->n + 1); v->a[v->n++] = *s; }
static void sv_free(seedvec *v) { free(v->a); v->a = 0; v->n = v->m = 0; }

typedef struct { int len; char op; } cig_t;             /* pair<int,char>            */
typedef struct { cig_t *a; int n, m; } cigvec;
static void cv_push(cigvec *v, int len, char op)
{
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 16; v->a = (cig_t *)realloc(v->a, (size_t)v->m * sizeof(cig_t)); }
    v->a[v->n].len = len; v->a[v->n].op = op; v->n++;
}
static void cv_insert_front(cigvec *v, int len, char op)
{
    cv_push(v, 0, 0);
    memmove(v->a + 1, v->a, (size_t)(v->n - 1) * sizeof(cig_t));
    v->a[0].len = len; v->a[0].op = op;
}

typedef struct {
    int Score, SJtype, PairedIdx;        /* AlignmentCandidate_t structure.h:125-132 */
    int64_t PosDiff;
    seedvec seeds;
} cand_t;

typedef struct {
    int AlnScore, SJtype, iFrag, PairedIdx;   /* AlignmentReport_t structure.h:134-141 */
    int bDir, chr; int64_t gPos;              /* Coordinate_t      structure.h:117-123 */
    uint32_t *cig; int ncig;                  /* merged CIGAR: len<<4 | op(M0 I1 D2 N3 S4) */
} report_t;

typedef struct {
    int rlen; const char *seq; uint8_t *enc;
    int mapq, score, sub_score, mis_num, CanNum, iBest;   /* ReadItem_t structure.h:149-164 */
    report_t *rep;
} read_t;

typedef struct {
    const orc_index *ix; const orc_params *pr;
    orc_counters c;
} ctx_t;

/* nst_nt4_table, BWT_Index/bntseq.c:40: ACGT/acgt -> 0..3, '-' -> 5, everything else 4 */
static uint8_t nt4(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': return 5;
    default: return 4;
    }
}

/* RefSequence[g], bwt_index.cpp:193-212,252: forward half from .pac, reverse half its
 * complement mirrored; RefSequence[2L] = '\0'.  Outside [0,2L] the reference reads out of
 * bounds; we define '\0' there (equivalent to strncpy's zero padding at the end). */
char orc_refbase(const orc_index *ix, int64_t g)
{
    int64_t L = ix->l_pac;
    if (g < 0 || g >= 2 * L) return 0;
    if (g < L) return "ACGT"[ix->pac[g >> 2] >> ((~g & 3) << 1) & 3];
    g = 2 * L - 1 - g;
    return "TGCA"[ix->pac[g >> 2] >> ((~g & 3) << 1) & 3];
}
#define REF(cx, g) orc_refbase((cx)->ix, (g))

/* ChrLocMap.lower_bound(g): smallest key >= g (bwt_index.cpp:249-250) */
static int loc_lower_bound(const orc_index *ix, int64_t g)
{
    int lo = 0, hi = 2 * ix->n_chr;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (ix->loc_key[mid] < g) lo = mid + 1; else hi = mid; }
    return lo; /* == 2*n_chr means end() */
}

/* ------------------------------------------------------------------------------------------
 * FM-index primitives  (bwt_search.cpp:26-137)
 * ---------------------------------------------------------------------------------------- */
/* counts of A,C,G,T among the 16 symbols of word w whose index is <= last (0..15), MSB first */
static void count_word(uint32_t w, int last, uint64_t cnt[4])
{
    int i;
    for (i = 0; i <= last; i++) cnt[(w >> ((15 - i) << 1)) & 3]++;
}

/* bwt_occ4, bwt_search.cpp:67-84: Occ(b, k) for b=0..3, rows counted inclusively; k==-1 -> 0 */
static void occ4(ctx_t *cx, uint64_t k, uint64_t cnt[4])
{
    const orc_index *ix = cx->ix;
    const uint32_t *p;
    int w, nw;
    if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
    k -= (k >= ix->primary);
    p = ix->bwt + ((k >> 7) << 4);
    memcpy(cnt, p, 32);
    p += 8;
    nw = (int)((k & 127) >> 4);
    for (w = 0; w < nw; w++) count_word(p[w], 15, cnt);
    count_word(p[nw], (int)(k & 15), cnt);
    cx->c.n_occ_blocks++;
}

/* bwt_2occ4, bwt_search.cpp:86-117 (same values as two occ4; block-sharing only matters for
 * the traffic count: 1 block when both rows fall in the same 128-row block, else 2) */
static void occ4_pair(ctx_t *cx, uint64_t k, uint64_t l, uint64_t ck[4], uint64_t cl[4])
{
    const orc_index *ix = cx->ix;
    uint64_t _k = k - (k >= ix->primary), _l = l - (l >= ix->primary);
    uint64_t before = cx->c.n_occ_blocks;
    occ4(cx, k, ck);
    occ4(cx, l, cl);
    cx->c.n_2occ4++;
    if (!(_l >> 7 != _k >> 7 || k == (uint64_t)-1 || l == (uint64_t)-1)) cx->c.n_occ_blocks = before + 1;
}

/* bwt_occ, bwt_search.cpp:43-65 */
static uint64_t occ1(ctx_t *cx, uint64_t k, int c)
{
    const orc_index *ix = cx->ix;
    uint64_t cnt[4];
    if (k == ix->seq_len) return ix->L2[c + 1] - ix->L2[c];
    if (k == (uint64_t)-1) return 0;
    occ4(cx, k, cnt);
    return cnt[c];
}

/* bwt_invPsi, bwt_search.cpp:119-125 */
static uint64_t inv_psi(ctx_t *cx, uint64_t k)
{
    const orc_index *ix = cx->ix;
    uint64_t x = k - (k > ix->primary);
    int c = (int)((ix->bwt[((x >> 7) << 4) + 8 + ((x & 0x7f) >> 4)] >> ((~x & 0xf) << 1)) & 3);
    uint64_t r = ix->L2[c] + occ1(cx, k, c);
    cx->c.n_lf++;
    return k == ix->primary ? 0 : r;
}

/* bwt_sa, bwt_search.cpp:127-137 */
static uint64_t sa_lookup(ctx_t *cx, uint64_t k)
{
    const orc_index *ix = cx->ix;
    uint64_t sa = 0, mask = (uint64_t)ix->sa_intv - 1;
    while (k & mask) { ++sa; k = inv_psi(cx, k); }
    cx->c.n_sa++;
    return sa + ix->sa[k / (uint64_t)ix->sa_intv];
}

/* BWT_Search, bwt_search.cpp:139-182.  Returns freq; *len only meaningful when freq>0. */
static int bwt_search(ctx_t *cx, const uint8_t *seq, int start, int stop, int *len, uint64_t *locs)
{
    const orc_index *ix = cx->ix;
    uint64_t x0, x1, x2, tk[4], tl[4];
    int pos, p = seq[start], j;
    x0 = ix->L2[p] + 1;
    x1 = ix->L2[3 - p] + 1;
    x2 = ix->L2[p + 1] - ix->L2[p];
    cx->c.n_search++;
    for (pos = start + 1; pos < stop; pos++) {
        uint64_t o0[4], o1[4], o2[4];
        int b;
        if (seq[pos] > 3) break;
        occ4_pair(cx, x1 - 1, x1 - 1 + x2, tk, tl);
        for (b = 0; b < 4; b++) { o1[b] = ix->L2[b] + 1 + tk[b]; o2[b] = tl[b] - tk[b]; }
        o0[3] = x0 + (x1 <= ix->primary && x1 + x2 - 1 >= ix->primary);
        o0[2] = o0[3] + o2[3];
        o0[1] = o0[2] + o2[2];
        o0[0] = o0[1] + o2[1];
        b = 3 - seq[pos];
        if (o2[b] == 0) break;
        x0 = o0[b]; x1 = o1[b]; x2 = o2[b];
    }
    if (x2 <= (uint64_t)cx->pr->max_dup && (pos - start) >= 16) {
        *len = pos - start;
        for (j = 0; j < (int)x2; j++) locs[j] = sa_lookup(cx, x0 + (uint64_t)j);
        return (int)x2;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * seeds and candidates  (AlignmentCandidates.cpp:21-25,181-215,241-288)
 * ---------------------------------------------------------------------------------------- */
static int cmp_gpos(const void *a, const void *b)   /* CompByGenomePos :21-25 */
{
    const seed_t *p = (const seed_t *)a, *q = (const seed_t *)b;
    if (p->gPos != q->gPos) return p->gPos < q->gPos ? -1 : 1;
    if (p->rPos != q->rPos) return p->rPos < q->rPos ? -1 : 1;
    return 0;
}

/* IdentifySeedPairs :181-215 */
static void identify_seed_pairs(ctx_t *cx, int rlen, const uint8_t *enc, seedvec *out)
{
    int pos = 0, end_pos = rlen - 13, len = 0, freq, i;
    uint64_t *locs = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(cx->pr->max_dup + 1));
    seed_t s;
    memset(&s, 0, sizeof s);
    s.simple = 1; s.acceptor = 0;
    out->n = 0;
    while (pos < end_pos) {
        if (enc[pos] > 3) { pos++; continue; }
        freq = bwt_search(cx, enc, pos, rlen, &len, locs);
        if (freq > 0) {
            s.rPos = pos; s.rLen = s.gLen = len;
            for (i = 0; i < freq; i++) { s.gPos = (int64_t)locs[i]; s.PosDiff = s.gPos - s.rPos; sv_push(out, &s); }
            pos += len;
        } else pos++;
    }
    free(locs);
    if (out->n > 1) qsort(out->a, (size_t)out->n, sizeof(seed_t), cmp_gpos);
}

typedef struct { cand_t *a; int n, m; } candvec;
static cand_t *cand_new(candvec *v)
{
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 4; v->a = (cand_t *)realloc(v->a, (size_t)v->m * sizeof(cand_t)); }
    memset(&v->a[v->n], 0, sizeof(cand_t));
    return &v->a[v->n++];
}
static void candvec_free(candvec *v)
{
    int i;
    for (i = 0; i < v->n; i++) sv_free(&v->a[i].seeds);
    free(v->a); v->a = 0; v->n = v->m = 0;
}

static int64_t i64abs(int64_t x) { return x < 0 ? -x : x; }

/* GenerateAlignmentCandidate :241-288 */
static void generate_candidates(ctx_t *cx, int rlen, const seedvec *sv, candvec *out)
{
    const orc_params *pr = cx->pr;
    int i, j, k, num = sv->n, thr = (int)(rlen * 0.3);
    out->n = 0;
    if (num == 0) return;
    i = 0;
    while (i < num && sv->a[i].PosDiff < 0) i++;
    for (; i < num;) {
        seedvec cur = {0, 0, 0};
        int score = sv->a[i].rLen;
        sv_push(&cur, &sv->a[i]);
        for (j = i, k = i + 1; k < num; k++) {
            int64_t pd = i64abs(sv->a[k].PosDiff - sv->a[j].PosDiff);
            int ok = pd < pr->max_gaps;
            if (!ok && pd < pr->max_intron) {
                int lb = loc_lower_bound(cx->ix, sv->a[j].gPos);
                /* lower_bound never returns end() for gPos < 2L */
                if (sv->a[k].gPos < cx->ix->loc_key[lb] && sv->a[k].rPos > sv->a[j].rPos) ok = 1;
            }
            if (!ok) break;
            score += sv->a[k].rLen;
            sv_push(&cur, &sv->a[k]);
            j = k;
        }
        if (score > thr) {
            cand_t *c = cand_new(out);
            c->Score = score; c->PairedIdx = -1; c->SJtype = -1;
            c->PosDiff = cur.a[0].PosDiff < 0 ? 0 : cur.a[0].PosDiff;
            c->seeds = cur;
        } else sv_free(&cur);
        i = k;
    }
}

/* RemoveRedundantCandidates, Mapping.cpp:371-401 */
static void remove_redundant(candvec *v)
{
    int i, thr, s1 = 0, s2 = 0;
    if (v->n <= 1) return;
    for (i = 0; i < v->n; i++) {
        int sc = v->a[i].Score;
        if (sc > s2) {
            if (sc >= s1) { s2 = s1; s1 = sc; }
            else s2 = sc;
        } else if (sc == s2) s2 = s1;
    }
    thr = (s1 == s2 || s1 - s2 > 20) ? s1 : s2;
    for (i = 0; i < v->n; i++) if (v->a[i].Score < thr) v->a[i].Score = 0;
}

/* CheckPairedAlignmentCandidates, Mapping.cpp:403-450 */
static int check_paired_candidates(candvec *v1, candvec *v2)
{
    int pairing = 0, i, j, best, n1 = v1->n, n2 = v2->n;
    if (n1 * n2 > 1000) { remove_redundant(v1); remove_redundant(v2); }
    for (i = 0; i < n1; i++) {
        int64_t min_dist = 2000000;
        if (v1->a[i].Score == 0) continue;
        for (best = -1, j = 0; j < n2; j++) {
            int64_t d;
            if (v2->a[j].Score == 0 || v2->a[j].PosDiff < v1->a[i].PosDiff) continue;
            d = i64abs(v2->a[j].PosDiff - v1->a[i].PosDiff);
            if (d < min_dist) { best = j; min_dist = d; }
        }
        if (best != -1) {
            j = best;
            if (v2->a[j].PairedIdx == -1) {
                pairing = 1;
                v1->a[i].PairedIdx = j; v2->a[j].PairedIdx = i;
            } else if (v1->a[i].Score > v1->a[v2->a[j].PairedIdx].Score) {
                v1->a[v2->a[j].PairedIdx].PairedIdx = -1;
                v1->a[i].PairedIdx = j; v2->a[j].PairedIdx = i;
            }
        }
    }
    return pairing;
}

/* RemoveUnMatedAlignmentCandidates, Mapping.cpp:452-477 */
static void remove_unmated(candvec *v1, candvec *v2)
{
    int i, j;
    for (i = 0; i < v1->n; i++) {
        if (v1->a[i].PairedIdx == -1) v1->a[i].Score = 0;
        else { j = v1->a[i].PairedIdx; v1->a[i].Score = v2->a[j].Score = v1->a[i].Score + v2->a[j].Score; }
    }
    for (j = 0; j < v2->n; j++) if (v2->a[j].PairedIdx == -1) v2->a[j].Score = 0;
}

/* ------------------------------------------------------------------------------------------
 * nw_alignment  (nw_alignment.cpp:3-82)
 * All values are multiples of 0.5 -> stored x2.  The 3-argument max resolves to
 * max(short,short,short): each operand is truncated toward zero to a 16-bit integer first.
 * ---------------------------------------------------------------------------------------- */
static int tr2(int v2)  /* 2 * (short)(v2/2.0): truncation toward zero */
{
    return v2 >= 0 ? (v2 & ~1) : -((-v2) & ~1);
}

/* aligns a[0..m) with b[0..n); writes NUL-terminated gapped strings, returns aligned length */
static int nw_align(ctx_t *cx, const char *a, int m, const char *b, int n, char *oa, char *ob)
{
    int W = n + 1, i, j, k = 0, len;
    size_t cells = (size_t)(m + 1) * (size_t)(n + 1);
    int *s = (int *)malloc(cells * 3 * sizeof(int)), *r = s + cells, *t = r + cells;
    if (cx) { cx->c.n_nw++; cx->c.nw_cells += (uint64_t)m * (uint64_t)n; }
    s[0] = r[0] = t[0] = 0;
    for (i = 1; i <= m; i++) { r[i * W] = -131072; s[i * W] = t[i * W] = -2 - i; }
    for (j = 1; j <= n; j++) { t[j] = -131072; s[j] = r[j] = -2 - j; }
    for (i = 1; i <= m; i++) {
        for (j = 1; j <= n; j++) {
            int rr, tt, d, x, y;
            x = r[i * W + j - 1] - 1; y = s[i * W + j - 1] - 3; rr = x > y ? x : y;
            x = t[(i - 1) * W + j] - 1; y = s[(i - 1) * W + j] - 3; tt = x > y ? x : y;
            r[i * W + j] = rr; t[i * W + j] = tt;
            d = tr2(s[(i - 1) * W + j - 1] + (nt4((unsigned char)a[i - 1]) == nt4((unsigned char)b[j - 1]) ? 3 : -3));
            x = tr2(rr); y = tr2(tt);
            /* max(x,y,z) = x>y ? max(x,z) : max(y,z) */
            s[i * W + j] = d > x ? (d > y ? d : y) : (x > y ? x : y);
        }
    }
    /* traceback, :61-74 -- built back to front */
    i = m; j = n;
    while (i > 0 || j > 0) {
        if (s[i * W + j] == r[i * W + j]) { oa[k] = '-'; ob[k] = b[j - 1]; k++; j--; }
        else if (s[i * W + j] == t[i * W + j]) { oa[k] = a[i - 1]; ob[k] = '-'; k++; i--; }
        else { oa[k] = a[i - 1]; ob[k] = b[j - 1]; k++; i--; j--; }
    }
    len = k;
    for (i = 0, j = len - 1; i < j; i++, j--) {
        char c = oa[i]; oa[i] = oa[j]; oa[j] = c;
        c = ob[i]; ob[i] = ob[j]; ob[j] = c;
    }
    oa[len] = ob[len] = 0;
    free(s);
    return len;
}

int orc_nw(const char *s1, const char *s2, char *out1, char *out2, int cap)
{
    int m = (int)strlen(s1), n = (int)strlen(s2);
    if (m + n + 1 > cap) return -1;
    return nw_align(NULL, s1, m, s2, n, out1, out2);
}

/* ------------------------------------------------------------------------------------------
 * k-mer re-seeding  (KmerAnalysis.cpp:25-166, AlignmentCandidates.cpp:596-624,685-700)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t wid, pos; } kmer_t;
typedef struct { int PosDiff; uint32_t rPos, gPos; } kpair_t;

/* character source: either the read or RefSequence */
typedef struct { ctx_t *cx; const char *s; int64_t g0; } charsrc;
static char cs_at(const charsrc *c, int i) { return c->s ? c->s[i] : REF(c->cx, c->g0 + i); }

static uint32_t kmer_id(const charsrc *c, int pos)  /* CreateKmerID :25-32 */
{
    uint32_t id = 0; int i;
    for (i = pos; i < pos + 8; i++) id = (id << 2) + nt4((unsigned char)cs_at(c, i));
    return id;
}

/* CreateKmerVecFromReadSeq :34-80, without the final sort; calls emit(pos,wid) in position order */
typedef void (*kmer_emit)(void *u, uint32_t pos, uint32_t wid);
static void kmer_scan(const charsrc *c, int len, kmer_emit emit, void *u)
{
    int count = 0, head, tail = 0;
    uint32_t wid;
    while (count < 8 && tail < len) { if (cs_at(c, tail++) != 'N') count++; else count = 0; }
    if (count != 8) return;
    head = tail - 8; wid = kmer_id(c, head);
    emit(u, (uint32_t)head, wid);
    for (head += 1; tail < len; head++, tail++) {
        char ch = cs_at(c, tail);
        if (ch != 'N') {
            wid = ((wid & 0x3FFF) << 2) + nt4((unsigned char)ch);
            emit(u, (uint32_t)head, wid);
        } else {
            count = 0; tail++;
            while (count < 8 && tail < len) { if (cs_at(c, tail++) != 'N') count++; else count = 0; }
            if (count == 8) { head = tail - 8; wid = kmer_id(c, head); emit(u, (uint32_t)head, wid); }
            else break;
        }
    }
}

typedef struct { kmer_t *a; int n, m; } kmervec;
static void emit_collect(void *u, uint32_t pos, uint32_t wid)
{
    kmervec *v = (kmervec *)u;
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 128; v->a = (kmer_t *)realloc(v->a, (size_t)v->m * sizeof(kmer_t)); }
    v->a[v->n].pos = pos; v->a[v->n].wid = wid; v->n++;
}
static int cmp_kmer(const void *a, const void *b)
{
    const kmer_t *p = (const kmer_t *)a, *q = (const kmer_t *)b;
    if (p->wid != q->wid) return p->wid < q->wid ? -1 : 1;
    return p->pos < q->pos ? -1 : (p->pos > q->pos);
}
typedef struct { const kmervec *rk; kpair_t *a; int n, m; } joinst;
/* IdentifyCommonKmers :82-106 -- the same multiset of pairs, produced by probing the (small,
 * wid-sorted) read k-mer list with every window k-mer; the later sort by (PosDiff,rPos) is a
 * total order up to identical elements, so the sorted vector is identical. */
static void emit_join(void *u, uint32_t gpos, uint32_t wid)
{
    joinst *j = (joinst *)u;
    const kmervec *rk = j->rk;
    int lo = 0, hi = rk->n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (rk->a[mid].wid < wid) lo = mid + 1; else hi = mid; }
    for (; lo < rk->n && rk->a[lo].wid == wid; lo++) {
        if (j->n == j->m) { j->m = j->m ? j->m * 2 : 256; j->a = (kpair_t *)realloc(j->a, (size_t)j->m * sizeof(kpair_t)); }
        j->a[j->n].rPos = rk->a[lo].pos; j->a[j->n].gPos = gpos;
        j->a[j->n].PosDiff = (int)(gpos - rk->a[lo].pos);
        j->n++;
    }
}
static int cmp_kpair(const void *a, const void *b)   /* CompByKmerPosDiff :19-23 */
{
    const kpair_t *p = (const kpair_t *)a, *q = (const kpair_t *)b;
    if (p->PosDiff != q->PosDiff) return p->PosDiff < q->PosDiff ? -1 : 1;
    if (p->rPos != q->rPos) return p->rPos < q->rPos ? -1 : 1;
    return 0;
}

/* GenerateLongestSimplePairsFromFragmentPair :134-166 + ReseedingWithSpecificRegion
 * AlignmentCandidates.cpp:596-624.  Returns 1 and fills *out when a seed is accepted. */
static int reseed_region(ctx_t *cx, const char *seq, int rBegin, int rEnd, int64_t Lb, int64_t Rb, seed_t *out)
{
    int rlen = rEnd - rBegin, glen = (int)(Rb - Lb), thr, i, j, l, s, num, max_len = 0, found = 0;
    kmervec rk = {0, 0, 0};
    joinst js;
    charsrc c1, c2;
    uint32_t best_r = 0, best_g = 0;
    c1.cx = cx; c1.s = seq + rBegin; c1.g0 = 0;
    c2.cx = cx; c2.s = NULL; c2.g0 = Lb;
    cx->c.n_reseed++; cx->c.reseed_window += (uint64_t)(glen > 0 ? glen : 0); cx->c.ref_bases += (uint64_t)(glen > 0 ? glen : 0);
    if ((thr = (int)(rlen * 0.85)) < 8) thr = 8;
    kmer_scan(&c1, rlen, emit_collect, &rk);
    if (rk.n > 1) qsort(rk.a, (size_t)rk.n, sizeof(kmer_t), cmp_kmer);
    js.rk = &rk; js.a = 0; js.n = js.m = 0;
    if (rk.n > 0 && glen > 0) kmer_scan(&c2, glen, emit_join, &js);
    if (js.n > 1) qsort(js.a, (size_t)js.n, sizeof(kpair_t), cmp_kpair);
    num = js.n;
    for (s = 1, i = 0; i < num;) {
        int pd = js.a[i].PosDiff;
        for (j = i + 1; j < num; j++) { if (js.a[j].PosDiff != pd) break; else s++; }
        l = 8 + (int)(js.a[j - 1].rPos - js.a[i].rPos);
        if (l > max_len && s > (l - 8) / 2) {
            best_r = js.a[i].rPos; best_g = js.a[i].gPos; max_len = l; s = 1;
        }
        i = j;
    }
    if (max_len >= thr && max_len > 0) {
        memset(out, 0, sizeof *out);
        out->simple = 1; out->acceptor = 0;
        out->rLen = out->gLen = max_len;
        out->rPos = (int)best_r + rBegin;
        out->gPos = (int64_t)best_g + Lb;
        out->PosDiff = out->gPos - out->rPos;
        found = 1;
    }
    free(rk.a); free(js.a);
    return found;
}

/* ------------------------------------------------------------------------------------------
 * per-candidate clean-up  (AlignmentCandidates.cpp:299-306,817-902)
 * ---------------------------------------------------------------------------------------- */
static void remove_null_seeds(seedvec *v)   /* RemoveNullSeeds :299-306 */
{
    int i, k = 0;
    for (i = 0; i < v->n; i++) if (v->a[i].rLen != 0) v->a[k++] = v->a[i];
    v->n = k;
}

typedef struct { int first, second; } ipair;
static int cmp_ipair_first(const void *a, const void *b)
{
    const ipair *p = (const ipair *)a, *q = (const ipair *)b;
    if (p->first != q->first) return p->first < q->first ? -1 : 1;
    return p->second < q->second ? -1 : (p->second > q->second); /* ties are order-insensitive downstream */
}

static void remove_tandem_repeat_seeds(seedvec *v)   /* :817-842 */
{
    int i, j, k, num = v->n, any = 0;
    ipair *vec;
    if (num < 2) return;
    vec = (ipair *)malloc((size_t)num * sizeof(ipair));
    for (i = 0; i < num; i++) { vec[i].first = v->a[i].rPos; vec[i].second = i; }
    qsort(vec, (size_t)num, sizeof(ipair), cmp_ipair_first);
    for (i = 0; i < num;) {
        j = i + 1; while (j < num && vec[j].first == vec[i].first) j++;
        if (j - i > 1) { any = 1; for (k = i; k < j; k++) v->a[vec[k].second].rLen = v->a[vec[k].second].gLen = 0; }
        i = j;
    }
    free(vec);
    if (any) remove_null_seeds(v);
}

static void remove_translocated_seeds(seedvec *v)   /* :844-902 */
{
    int i, j, k, s1, s2, num = v->n, any = 0;
    ipair *vec;
    if (num < 2) return;
    vec = (ipair *)malloc((size_t)num * sizeof(ipair));
    for (i = 0; i < num; i++) { vec[i].first = v->a[i].rPos; vec[i].second = i; }
    qsort(vec, (size_t)num, sizeof(ipair), cmp_ipair_first);
    for (i = 0; i < num; i++) {
        if (vec[i].first != v->a[i].rPos) {
            int max_idx = vec[i].second;      /* IdentifyTranslocationRange :844-853 */
            any = 1;
            for (j = i + 1; j <= max_idx; j++) if (vec[j].second > max_idx) max_idx = vec[j].second;
            j = max_idx;
            s1 = s2 = 0;
            for (k = i; k <= j; k++) {
                if (k < vec[k].second) s1 += v->a[vec[k].second].rLen;
                else s2 += v->a[vec[k].second].rLen;
            }
            if (s1 > s2) { for (k = i; k <= j; k++) if (k > vec[k].second) v->a[vec[k].second].rLen = v->a[vec[k].second].gLen = 0; }
            else { for (k = i; k <= j; k++) if (k < vec[k].second) v->a[vec[k].second].rLen = v->a[vec[k].second].gLen = 0; }
            i = j;
        }
    }
    free(vec);
    if (any) remove_null_seeds(v);
}

/* IdentifyMissingSeeds :685-700 */
static void identify_missing_seeds(ctx_t *cx, const char *seq, seedvec *v)
{
    int i, num = v->n, rGaps, pd;
    seed_t s;
    for (i = 1; i < num; i++) {
        pd = (int)(v->a[i].PosDiff - v->a[i - 1].PosDiff);
        if (pd > cx->pr->max_gaps && (rGaps = v->a[i].rPos - v->a[i - 1].rPos - v->a[i - 1].rLen) > 20) {
            if (reseed_region(cx, seq, v->a[i - 1].rPos + v->a[i - 1].rLen, v->a[i].rPos, v->a[i - 1].gPos + v->a[i - 1].gLen, v->a[i].gPos, &s))
                sv_push(v, &s);
        }
    }
    if (v->n > num) qsort(v->a, (size_t)v->n, sizeof(seed_t), cmp_gpos);
}

/* IdentifyBestGappedPartition :385-467 + FillGapsBetweenAdjacentSeeds :547-575 */
static void fill_gaps(ctx_t *cx, const char *seq, const seed_t *L, const seed_t *R, seedvec *out)
{
    int rGaps = R->rPos - (L->rPos + L->rLen), len, len3, i, p, s, max_score = 0, gp = 0, right_ext = 0, left_ext = 0;
    char *f1 = (char *)malloc((size_t)(8 * rGaps + 16)), *f2 = f1 + 2 * rGaps + 4, *f3 = f2 + 2 * rGaps + 4, *f4 = f3 + 2 * rGaps + 4;
    char *g = (char *)malloc((size_t)rGaps + 1);
    int *Rv = (int *)calloc((size_t)(2 * (rGaps + 1)), sizeof(int)), *Lv = Rv + rGaps + 1;
    int64_t gPos;
    seed_t sd;

    for (i = 0; i < rGaps; i++) g[i] = REF(cx, L->gPos + L->gLen + i);
    cx->c.ref_bases += 2 * (uint64_t)rGaps;
    len = nw_align(cx, seq + L->rPos + L->rLen, rGaps, g, rGaps, f1, f2);
    i = len - 1; while (i >= 0 && f2[i] == '-') i--;                        /* :399 */
    for (i += 1, gPos = L->gPos + L->gLen + rGaps; i < len; i++, gPos++) f2[i] = REF(cx, gPos);
    for (p = s = 0, i = 0; i < len; i++) {                                 /* :403-409 */
        if (f1[i] == f2[i]) s++;
        if (f1[i] != '-') p++;
        Rv[p] = s;
    }
    for (i = 0; i < rGaps; i++) g[i] = REF(cx, R->gPos - rGaps + i);
    len3 = nw_align(cx, seq + L->rPos + L->rLen, rGaps, g, rGaps, f3, f4);
    i = 0; while (i < len3 && f4[i] == '-') i++;                           /* :424 */
    for (i -= 1, gPos = R->gPos - rGaps; i >= 0; i--, gPos--) f4[i] = REF(cx, gPos);
    for (p = 0, s = 0, i = len3 - 1; i >= 0; i--) {                        /* :428-434 */
        if (f3[i] == f4[i]) s++;
        if (f3[i] != '-') p++;
        Lv[rGaps - p] = s;
    }
    for (i = 0; i <= rGaps; i++) { s = Rv[i] + Lv[i]; if (s > max_score) { max_score = s; gp = i; } }
    if (max_score < (int)(rGaps * 0.8) || (rGaps - max_score) > cx->pr->max_mismatch) right_ext = left_ext = 0;
    else {
        for (right_ext = 0, p = gp, i = 0; p > 0; i++) { if (f1[i] != '-') p--; if (f2[i] != '-') right_ext++; }
        for (left_ext = 0, p = rGaps - gp, i = len3 - 1; p > 0; i--) { if (f3[i] != '-') p--; if (f4[i] != '-') left_ext++; }
    }
    memset(&sd, 0, sizeof sd);
    if (gp > 0) {
        sd.rPos = L->rPos + L->rLen; sd.gPos = L->gPos + L->gLen; sd.PosDiff = sd.gPos - sd.rPos;
        sd.rLen = gp; sd.gLen = right_ext;
        sv_push(out, &sd);
    }
    if ((rGaps -= gp) > 0) {
        sd.rLen = rGaps; sd.gLen = left_ext;
        sd.rPos = R->rPos - sd.rLen; sd.gPos = R->gPos - sd.gLen; sd.PosDiff = sd.gPos - sd.rPos;
        sv_push(out, &sd);
    }
    free(f1); free(g); free(Rv);
}

/* SeedExtension :577-594 */
static void seed_extension(ctx_t *cx, const char *seq, seedvec *v)
{
    int i, num = v->n, pd;
    seedvec add = {0, 0, 0};
    for (i = 1; i < num; i++) {
        pd = (int)(v->a[i].PosDiff - v->a[i - 1].PosDiff);
        if (pd > cx->pr->min_intron && v->a[i].rPos > v->a[i - 1].rPos + v->a[i - 1].rLen)
            fill_gaps(cx, seq, &v->a[i - 1], &v->a[i], &add);
    }
    if (add.n > 0) {
        for (i = 0; i < add.n; i++) sv_push(v, &add.a[i]);
        qsort(v->a, (size_t)v->n, sizeof(seed_t), cmp_gpos);
    }
    sv_free(&add);
}

/* ------------------------------------------------------------------------------------------
 * splice junctions  (AlignmentCandidates.cpp:6,702-815; main.cpp:18)
 * ---------------------------------------------------------------------------------------- */
static const int ShiftArr[19] = { 0, 1, -1, 2, -2, 3, -3, 4, -4, 5, -5, 6, -6, 7, -7, 8, -8, 9, -9 };
static const char *SJArr[4] = { "GT/AG", "CT/AC", "GC/AG", "CT/GC" };

static int check_seq_fragment(ctx_t *cx, int64_t Lg, int64_t Rg, int shift)   /* :702-730 */
{
    int i;
    if (shift <= 0) { shift = -shift; Lg -= shift; Rg -= shift; }
    for (i = 0; i < shift; i++, Lg++, Rg++) if (REF(cx, Lg) != REF(cx, Rg)) return 0;
    return 1;
}

static int identify_splice_junction(ctx_t *cx, int type, const seed_t *l, const seed_t *r)   /* :732-756 */
{
    int i, j, shift = 0;
    int64_t Lg, Rg, g1, g2;
    i = l->rLen < r->rLen ? l->rLen : r->rLen;
    j = l->gLen < r->gLen ? l->gLen : r->gLen;
    if (i < j) j = i;
    if (j > 9) j = 9;
    j <<= 1;
    Lg = l->gPos + l->gLen; Rg = r->gPos;
    for (i = 0; i <= j; i++) {
        shift = ShiftArr[i];
        if (shift != 0 && !check_seq_fragment(cx, Lg, Rg, shift)) continue;
        g1 = Lg + shift; g2 = Rg - 2 + shift;
        if (REF(cx, g1) == SJArr[type][0] && REF(cx, g1 + 1) == SJArr[type][1] && REF(cx, g2) == SJArr[type][3] && REF(cx, g2 + 1) == SJArr[type][4]) break;
    }
    return i > j ? 10 : shift;
}

static int check_splice_junction(ctx_t *cx, seedvec *v)   /* :758-815 */
{
    int i, j, type, shift, c, mis, min_cost = 1000, best_type = -1, num = v->n, nvec, nbest = 0;
    ipair *vec = (ipair *)malloc((size_t)(num + 1) * 2 * sizeof(ipair)), *best = vec + num + 1;
    for (type = 0; type < 4; type++) {
        nvec = 0; mis = 0;
        for (c = 0, i = 1; i < num; i++) {
            if ((v->a[i].PosDiff - v->a[i - 1].PosDiff) > cx->pr->min_intron && v->a[i - 1].simple && v->a[i].simple) {
                shift = identify_splice_junction(cx, type, &v->a[i - 1], &v->a[i]);
                if (shift != 10) { vec[nvec].first = i; vec[nvec].second = shift; nvec++; }
                else mis++;
                c += shift < 0 ? -shift : shift;
            }
        }
        if (nvec > 0 && c < min_cost) { min_cost = c; best_type = type; nbest = nvec; memcpy(best, vec, (size_t)nvec * sizeof(ipair)); }
        if (mis == 0) break;
    }
    if (best_type != -1) {
        for (i = 0; i < nbest; i++) {
            j = best[i].first; shift = best[i].second;
            v->a[j].acceptor = 1;
            if (shift != 0) {
                v->a[j - 1].rLen += shift; v->a[j - 1].gLen += shift;
                v->a[j].rLen -= shift; v->a[j].gLen -= shift;
                v->a[j].rPos += shift; v->a[j].gPos += shift;
            }
        }
    }
    free(vec);
    return best_type;
}

/* ------------------------------------------------------------------------------------------
 * overlaps and normal pairs  (AlignmentCandidates.cpp:904-1035)
 * ---------------------------------------------------------------------------------------- */
static int check_seed_overlapping(seed_t *p1, seed_t *p2)   /* :904-954 */
{
    int ov, master = 1;
    if ((ov = p1->rPos + p1->rLen - p2->rPos) > 0) {
        if (p1->rLen < p2->rLen) {
            master = 0;
            if (p1->rLen > ov) p1->gLen = (p1->rLen -= ov);
            else p1->rLen = p1->gLen = 0;
        } else {
            if (p2->rLen > ov) { p2->rPos += ov; p2->gPos += ov; p2->gLen = (p2->rLen -= ov); }
            else p2->rLen = p2->gLen = 0;
        }
    }
    if ((p1->rLen > 0 && p2->rLen > 0) && (ov = (int)(p1->gPos + p1->gLen - p2->gPos)) > 0) {
        if (p1->gLen < p2->gLen) {
            master = 0;
            if (p1->rLen > ov) p1->gLen = (p1->rLen -= ov);
            else p1->rLen = p1->gLen = 0;
        } else {
            if (p2->rLen > ov) { p2->rPos += ov; p2->gPos += ov; p2->gLen = (p2->rLen -= ov); }
            else p2->rLen = p2->gLen = 0;
        }
    }
    return master;
}

static void check_overlapping_seeds(seedvec *v)   /* :956-999 */
{
    int64_t gEnd;
    int i, j, num = v->n, rEnd, any = 0;
    if (num < 2) return;
    for (i = 0; i < num;) {
        if (v->a[i].rLen > 0) {
            rEnd = v->a[i].rPos + v->a[i].rLen - 1;
            gEnd = v->a[i].gPos + v->a[i].gLen - 1;
            for (j = i + 1; j < num; j++) {
                if (v->a[j].rLen == 0) continue;
                if (rEnd < v->a[j].rPos && gEnd < v->a[j].gPos) break;
                if (!check_seed_overlapping(&v->a[i], &v->a[j])) break;
            }
            if (v->a[i].rLen == 0) {
                any = 1;
                i = i - 1;                                  /* LocateThePreviousSeedIdx(i-1) :956-961 */
                while (i > 0 && v->a[i].rLen == 0) i--;
                if (i < 0) i = 0;
            } else i++;
        } else { any = 1; i++; }
    }
    if (any) remove_null_seeds(v);
}

static void identify_normal_pairs(seedvec *v)   /* :1001-1035 */
{
    int i, j, rGaps, gGaps, num;
    seed_t sp;
    if (v->n <= 1) return;
    check_overlapping_seeds(v);
    memset(&sp, 0, sizeof sp);
    num = v->n;
    for (i = 0, j = 1; j < num; i++, j++) {
        int64_t gg;
        if (v->a[j].rPos - v->a[i].rPos - v->a[i].rLen == 0) continue;
        rGaps = v->a[j].rPos - (v->a[i].rPos + v->a[i].rLen); if (rGaps < 0) rGaps = 0;
        gg = v->a[j].gPos - (v->a[i].gPos + v->a[i].gLen);
        gGaps = (int)gg; if (gGaps < 0) gGaps = 0; else if (gGaps > 30 && gGaps > (rGaps << 1)) gGaps = 0;
        if (rGaps > 0 || gGaps > 0) {
            sp.rPos = v->a[i].rPos + v->a[i].rLen;
            sp.gPos = v->a[i].gPos + v->a[i].gLen;
            sp.PosDiff = sp.gPos - sp.rPos;
            sp.rLen = rGaps; sp.gLen = gGaps;
            sv_push(v, &sp);
        }
    }
    if (v->n > num) {   /* inplace_merge(begin, begin+num, end, CompByGenomePos) :1033 */
        seed_t *tmp = (seed_t *)malloc((size_t)v->n * sizeof(seed_t));
        int a = 0, b = num, k = 0;
        while (a < num && b < v->n) {
            if (cmp_gpos(&v->a[b], &v->a[a]) < 0) tmp[k++] = v->a[b++];
            else tmp[k++] = v->a[a++];
        }
        while (a < num) tmp[k++] = v->a[a++];
        while (b < v->n) tmp[k++] = v->a[b++];
        memcpy(v->a, tmp, (size_t)v->n * sizeof(seed_t));
        free(tmp);
    }
}

/* CheckCoordinateValidity :136-163 */
static int check_coordinate_validity(const orc_index *ix, const seedvec *v)
{
    int64_t g1 = 0, g2 = 2 * ix->l_pac, L = ix->l_pac;
    int i;
    for (i = 0; i < v->n; i++) if (v->a[i].gLen > 0) { g1 = v->a[i].gPos; break; }
    for (i = v->n - 1; i >= 0; i--) if (v->a[i].gLen > 0) { g2 = v->a[i].gPos + v->a[i].gLen - 1; break; }
    if ((g1 < L && g2 >= L) || (g1 >= L && g2 < L)) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * segment pair -> CIGAR  (tools.cpp:40-104,130-300)
 * ---------------------------------------------------------------------------------------- */
static int add_new_cigar_elements(const char *s1, const char *s2, int len, cigvec *cv)   /* :49-104 */
{
    char state = '*';
    int i, c = 0, score = 0;
    for (i = 0; i < len; i++) {
        char st;
        if (s1[i] == '-') st = 'D';
        else if (s2[i] == '-') st = 'I';
        else { st = 'M'; if (s1[i] == s2[i]) score++; }
        if (state == st) c++;
        else { if (c > 0) cv_push(cv, c, state); c = 1; state = st; }
    }
    if (c > 0) cv_push(cv, c, state);
    return score;
}

static int check_local_alignment_quality(const char *a1, const char *a2, int len)   /* :166-201 */
{
    int i, n = 0, mis = 0, type = -1, st = 0;
    for (i = 0; i < len; i++) {
        if (a1[i] == '-') { if (type != 0) { type = 0; st++; } }
        else if (a2[i] == '-') { if (type != 1) { type = 1; st++; } }
        else { n++; if (a1[i] != a2[i]) mis++; if (type != 2) { type = 2; st++; } }
    }
    if (st >= 4 || (mis >= 3 && mis >= (int)(n * 0.3))) return 0;
    return 1;
}

/* fetch genome fragment; returns malloc'd buffer of gLen chars */
static char *ref_fragment(ctx_t *cx, int64_t gPos, int gLen)
{
    char *g = (char *)malloc((size_t)(gLen > 0 ? gLen : 0) + 1);
    int i;
    for (i = 0; i < gLen; i++) g[i] = REF(cx, gPos + i);
    g[gLen > 0 ? gLen : 0] = 0;
    cx->c.ref_bases += (uint64_t)(gLen > 0 ? gLen : 0);
    return g;
}

static int frag_mismatches(const char *a, const char *b, int len)   /* CalFragPairMismatchBases :40-47 */
{
    int i, c = 0;
    for (i = 0; i < len; i++) if (a[i] != b[i]) c++;
    return c;
}

static int process_normal_pair(ctx_t *cx, const char *seq, seed_t *sp, cigvec *cv)   /* :130-164 */
{
    int n, score = 0;
    if (sp->PosDiff == -1) cv_push(cv, sp->rLen, 'S');
    else if (sp->rLen == 0 || sp->gLen == 0) {
        if (sp->rLen > 0) cv_push(cv, sp->rLen, 'I');
        else if (sp->gLen > 0) cv_push(cv, sp->gLen, 'D');
    } else {
        char *g = ref_fragment(cx, sp->gPos, sp->gLen);
        if (sp->rLen == sp->gLen && (n = frag_mismatches(seq + sp->rPos, g, sp->rLen)) <= 2 && n <= (int)(sp->rLen * 0.2)) {
            score = sp->rLen - n;
            cv_push(cv, sp->rLen, 'M');
        } else {
            char *o1 = (char *)malloc((size_t)(2 * (sp->rLen + sp->gLen) + 4)), *o2 = o1 + sp->rLen + sp->gLen + 2;
            int len = nw_align(cx, seq + sp->rPos, sp->rLen, g, sp->gLen, o1, o2);
            score = add_new_cigar_elements(o1, o2, len, cv);
            free(o1);
        }
        free(g);
    }
    return score;
}

static int process_head_pair(ctx_t *cx, const char *seq, seed_t *sp, cigvec *cv)   /* :203-249 */
{
    int n, score;
    char *g = ref_fragment(cx, sp->gPos, sp->gLen);
    if (sp->rLen == sp->gLen && (n = frag_mismatches(seq + sp->rPos, g, sp->rLen)) <= 2 && n <= (int)(sp->rLen * 0.2)) {
        score = sp->rLen - n;
        cv_push(cv, sp->rLen, 'M');
    } else {
        char *o1 = (char *)malloc((size_t)(2 * (sp->rLen + sp->gLen) + 4)), *o2 = o1 + sp->rLen + sp->gLen + 2;
        int len = nw_align(cx, seq + sp->rPos, sp->rLen, g, sp->gLen, o1, o2);
        if (!check_local_alignment_quality(o1, o2, len)) { cv_push(cv, sp->rLen, 'S'); score = 0; }
        else {
            char *a1 = o1, *a2 = o2;
            int p = 0;
            while (p < len && a1[p] == '-') p++;
            if (p > 0) { a1 += p; a2 += p; len -= p; sp->gPos += p; sp->gLen -= p; }
            p = 0; while (p < len && a2[p] == '-') p++;
            if (p > 0) { a1 += p; a2 += p; len -= p; sp->rPos += p; sp->rLen -= p; cv_push(cv, p, 'S'); }
            score = add_new_cigar_elements(a1, a2, len, cv);
        }
        free(o1);
    }
    free(g);
    return score;
}

static int process_tail_pair(ctx_t *cx, const char *seq, seed_t *sp, cigvec *cv)   /* :251-300 */
{
    int n, score;
    char *g = ref_fragment(cx, sp->gPos, sp->gLen);
    if (sp->rLen == sp->gLen && (n = frag_mismatches(seq + sp->rPos, g, sp->rLen)) <= 2 && n <= (int)(sp->rLen * 0.2)) {
        score = sp->rLen - n;
        cv_push(cv, sp->rLen, 'M');
    } else {
        char *o1 = (char *)malloc((size_t)(2 * (sp->rLen + sp->gLen) + 4)), *o2 = o1 + sp->rLen + sp->gLen + 2;
        int len = nw_align(cx, seq + sp->rPos, sp->rLen, g, sp->gLen, o1, o2);
        if (!check_local_alignment_quality(o1, o2, len)) { cv_push(cv, sp->rLen, 'S'); score = 0; }
        else {
            int p, c;
            p = len - 1; c = 0; while (p >= 0 && o1[p] == '-') { c++; p--; }
            if (c > 0) { len -= c; sp->gLen -= c; }
            p = len - 1; c = 0; while (p >= 0 && o2[p] == '-') { c++; p--; }
            if (c > 0) { len -= c; sp->rLen -= c; }
            score = add_new_cigar_elements(o1, o2, len, cv);
            if (c > 0) cv_push(cv, c, 'S');
        }
        free(o1);
    }
    free(g);
    return score;
}

/* ------------------------------------------------------------------------------------------
 * GenMappingReport  (AlignmentCandidates.cpp:37-61,83-116,1052-1207)
 * ---------------------------------------------------------------------------------------- */
static int op_code(char c) { switch (c) { case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4; default: return 15; } }

/* GenerateCIGAR :37-61 -> merged op list */
static void generate_cigar(const cigvec *cv, report_t *rp)
{
    int i, c = 0, n = 0;
    char state = '\0';
    rp->cig = (uint32_t *)malloc((size_t)(cv->n + 1) * sizeof(uint32_t));
    for (i = 0; i < cv->n; i++) {
        if (cv->a[i].op != state) {
            if (c > 0) rp->cig[n++] = ((uint32_t)c << 4) | (uint32_t)op_code(state);
            c = cv->a[i].len; state = cv->a[i].op;
        } else c += cv->a[i].len;
    }
    if (c > 0) rp->cig[n++] = ((uint32_t)c << 4) | (uint32_t)op_code(state);
    rp->ncig = n;
}

static void gen_mapping_report(ctx_t *cx, int first, read_t *rd, candvec *cv)
{
    const orc_index *ix = cx->ix;
    int i, j, g, num, score, mis_num;
    cigvec cig = {0, 0, 0};
    rd->score = rd->iBest = 0;
    if ((rd->CanNum = cv->n) > 0) {
        rd->rep = (report_t *)calloc((size_t)rd->CanNum, sizeof(report_t));
        for (i = 0; i < cv->n; i++) {
            cand_t *c = &cv->a[i];
            report_t *rp = &rd->rep[i];
            seedvec *sv = &c->seeds;
            rp->SJtype = -1; rp->AlnScore = 0; rp->PairedIdx = c->PairedIdx; rp->chr = -1;
            if (c->Score == 0) continue;
            remove_tandem_repeat_seeds(sv);
            remove_translocated_seeds(sv);
            identify_missing_seeds(cx, rd->seq, sv);
            seed_extension(cx, rd->seq, sv);
            rp->SJtype = c->SJtype = check_splice_junction(cx, sv);
            identify_normal_pairs(sv);
            num = sv->n;
            if (num > 1 && !check_coordinate_validity(ix, sv)) continue;
            cig.n = 0; mis_num = 0;
            for (j = 0; j < num; j++) {
                seed_t *s = &sv->a[j];
                if (s->rLen == 0 && s->gLen == 0) continue;
                if (j > 0 && (g = (int)(s->gPos - (sv->a[j - 1].gPos + sv->a[j - 1].gLen))) > 0) cv_push(&cig, g, 'N');
                if (s->simple) { cv_push(&cig, s->rLen, 'M'); rp->AlnScore += s->rLen; }
                else {
                    if (j == 0) score = process_head_pair(cx, rd->seq, s, &cig);
                    else if (j == num - 1) score = process_tail_pair(cx, rd->seq, s, &cig);
                    else score = process_normal_pair(cx, rd->seq, s, &cig);
                    rp->AlnScore += score;
                    mis_num += s->rLen - score;
                }
            }
            if (num > 0) {
                if ((j = sv->a[0].rPos) > 0) cv_insert_front(&cig, j, 'S');
                if ((j = rd->rlen - (sv->a[num - 1].rPos + sv->a[num - 1].rLen)) > 0) cv_push(&cig, j, 'S');
            }
            if (mis_num > cx->pr->max_mismatch || cig.n == 0) rp->AlnScore = 0;
            for (j = 0; j < cig.n; j++) if (cig.a[j].op == 'N' && cig.a[j].len < cx->pr->min_intron) { rp->AlnScore = 0; break; }   /* CheckMinIntronSize :1052 */
            if (rp->AlnScore > 0) {
                /* GenCoordinateInfo :83-116 */
                int64_t gPos = sv->a[0].gPos, end_gPos = sv->a[num - 1].gPos + sv->a[num - 1].gLen - 1;
                int lb = loc_lower_bound(ix, gPos);
                if (gPos < ix->l_pac) {
                    rp->bDir = first ? 1 : 0;
                    rp->chr = ix->loc_chr[lb];
                    rp->gPos = gPos + 1 - ix->chr_off[rp->chr];
                } else {
                    rp->bDir = first ? 0 : 1;
                    rp->chr = ix->loc_chr[lb];
                    rp->gPos = ix->loc_key[lb] - end_gPos + 1;
                }
                if (rp->gPos <= 0) rp->AlnScore = 0;
                else {
                    if (sv->a[0].gPos >= ix->l_pac) {
                        int a, b;
                        for (a = 0, b = cig.n - 1; a < b; a++, b--) { cig_t t = cig.a[a]; cig.a[a] = cig.a[b]; cig.a[b] = t; }
                    }
                    generate_cigar(&cig, rp);
                }
                if (rp->AlnScore > rd->score) {
                    rd->iBest = i; rd->mis_num = mis_num; rd->sub_score = rd->score; rd->score = rp->AlnScore;
                } else if (rp->AlnScore == rd->score) rd->sub_score = rd->score;
            }
        }
    } else {
        rd->CanNum = 1; rd->iBest = 0;
        rd->rep = (report_t *)calloc(1, sizeof(report_t));
        rd->rep[0].AlnScore = 0; rd->rep[0].PairedIdx = -1; rd->rep[0].SJtype = -1; rd->rep[0].chr = -1;
    }
    free(cig.a);
}

/* ------------------------------------------------------------------------------------------
 * pairing of final alignments, FLAG, MAPQ  (Mapping.cpp:74-206,479-530)
 * ---------------------------------------------------------------------------------------- */
static void check_paired_final(const orc_params *pr, read_t *r1, read_t *r2)   /* :479-530 */
{
    int mated, i, j, s;
    mated = (r1->rep[r1->iBest].PairedIdx == r2->iBest);
    if (!pr->multi_hit && mated) return;
    if (!mated && r1->score > 0 && r2->score > 0) {
        for (s = 0, i = 0; i < r1->CanNum; i++) {
            if (r1->rep[i].AlnScore > 0 && (j = r1->rep[i].PairedIdx) != -1 && r2->rep[j].AlnScore > 0) {
                mated = 1;
                if (s < r1->rep[i].AlnScore + r2->rep[j].AlnScore) {
                    s = r1->rep[i].AlnScore + r2->rep[j].AlnScore;
                    r1->iBest = i; r1->score = r1->rep[i].AlnScore;
                    r2->iBest = j; r2->score = r2->rep[j].AlnScore;
                }
            }
        }
    }
    if (mated) {
        for (i = 0; i < r1->CanNum; i++) {
            if (r1->rep[i].AlnScore != r1->score || ((j = r1->rep[i].PairedIdx) != -1 && r2->rep[j].AlnScore != r2->score)) {
                r1->rep[i].AlnScore = 0; r1->rep[i].PairedIdx = -1;
            }
        }
    } else {
        for (i = 0; i < r1->CanNum; i++) {
            if (r1->rep[i].PairedIdx != -1) r1->rep[i].PairedIdx = -1;
            if (r1->rep[i].AlnScore > 0 && r1->rep[i].AlnScore != r1->score) r1->rep[i].AlnScore = 0;
        }
        for (j = 0; j < r2->CanNum; j++) {
            if (r2->rep[j].PairedIdx != -1) r2->rep[j].PairedIdx = -1;
            if (r2->rep[j].AlnScore > 0 && r2->rep[j].AlnScore != r2->score) r2->rep[j].AlnScore = 0;
        }
    }
}

static void set_single_flag(read_t *r)   /* :74-99 */
{
    int i;
    if (r->score > r->sub_score) { i = r->iBest; r->rep[i].iFrag = r->rep[i].bDir ? 0 : 0x10; }
    else if (r->score > 0) { for (i = 0; i < r->CanNum; i++) if (r->rep[i].AlnScore > 0) r->rep[i].iFrag = r->rep[i].bDir ? 0 : 0x10; }
    else r->rep[0].iFrag = 0x4;
}

static void set_one_mate_flags(read_t *a, read_t *b, int base)   /* :124-153 / :155-184 */
{
    int i, j;
    if (a->score > a->sub_score) {
        i = a->iBest;
        a->rep[i].iFrag = base | (a->rep[i].bDir ? 0x20 : 0x10);
        if ((j = a->rep[i].PairedIdx) != -1 && b->rep[j].AlnScore > 0) a->rep[i].iFrag |= 0x2;
        else a->rep[i].iFrag |= 0x8;
    } else if (a->score > 0) {
        for (i = 0; i < a->CanNum; i++) {
            if (a->rep[i].AlnScore > 0) {
                a->rep[i].iFrag = base | (a->rep[i].bDir ? 0x20 : 0x10);
                if ((j = a->rep[i].PairedIdx) != -1 && b->rep[j].AlnScore > 0) a->rep[i].iFrag |= 0x2;
                else a->rep[i].iFrag |= 0x8;
            }
        }
    } else {
        a->rep[0].iFrag = base | 0x4;
        if (b->score == 0) a->rep[0].iFrag |= 0x8;
        else a->rep[0].iFrag |= (b->rep[b->iBest].bDir ? 0x10 : 0x20);
    }
}

static void set_paired_flag(read_t *r1, read_t *r2)   /* :101-186 */
{
    int i, j;
    if (r1->score > r1->sub_score && r2->score > r2->sub_score) {
        i = r1->iBest; j = r2->iBest;
        r1->rep[i].iFrag = 0x41; r2->rep[j].iFrag = 0x81;
        if (j == r1->rep[i].PairedIdx) { r1->rep[i].iFrag |= 0x2; r2->rep[j].iFrag |= 0x2; }
        r1->rep[i].iFrag |= (r1->rep[i].bDir ? 0x20 : 0x10);
        r2->rep[j].iFrag |= (r2->rep[j].bDir ? 0x20 : 0x10);
    } else {
        set_one_mate_flags(r1, r2, 0x41);
        set_one_mate_flags(r2, r1, 0x81);
    }
}

static void evaluate_mapq(read_t *r)   /* :188-206 */
{
    int i, n;
    if (r->score == 0 || r->score == r->sub_score) r->mapq = 0;
    else if (r->sub_score == 0 || r->score > r->sub_score) r->mapq = 50;
    else {
        for (n = 0, i = 0; i < r->CanNum; i++) if (r->rep[i].AlnScore == r->score) n++;
        if (n >= 10) r->mapq = 0; else if (n >= 4) r->mapq = 1; else if (n == 3) r->mapq = 2; else if (n == 2) r->mapq = 3; else r->mapq = 50;
    }
}

/* UpdateLocalSJMap, Mapping.cpp:532-565 -> list of (g1,g2,type) */
typedef struct { orc_sj_out *a; int n, m; } sjvec;
static void collect_sj(ctx_t *cx, const cand_t *c, int read_idx, sjvec *out)
{
    int i;
    int64_t g1, g2, L = cx->ix->l_pac;
    if (c->SJtype == -1) return;
    for (i = 1; i < c->seeds.n; i++) {
        const seed_t *p = &c->seeds.a[i - 1], *s = &c->seeds.a[i];
        if (!s->acceptor) continue;
        if (c->PosDiff < L) { g1 = p->gPos + p->gLen; g2 = s->gPos - 1; }
        else { g1 = 2 * L - s->gPos; g2 = 2 * L - 1 - (p->gPos + p->gLen); }
        if (i64abs(g2 - g1) < cx->pr->min_intron) continue;
        if (out->n == out->m) { out->m = out->m ? out->m * 2 : 4; out->a = (orc_sj_out *)realloc(out->a, (size_t)out->m * sizeof(orc_sj_out)); }
        out->a[out->n].g1 = g1; out->a[out->n].g2 = g2; out->a[out->n].type = c->SJtype; out->a[out->n].read_idx = read_idx;
        out->n++;
    }
}

/* ------------------------------------------------------------------------------------------
 * the chunk body of ReadMapping  (Mapping.cpp:598-639)
 * ---------------------------------------------------------------------------------------- */
static void read_init(read_t *r, const char *seq, int rlen)
{
    int i;
    memset(r, 0, sizeof *r);
    r->rlen = rlen; r->seq = seq;
    r->enc = (uint8_t *)malloc((size_t)rlen + 1);
    for (i = 0; i < rlen; i++) r->enc[i] = nt4((unsigned char)seq[i]);   /* GetData.cpp:148 */
}
static void read_free(read_t *r)
{
    int i;
    for (i = 0; i < r->CanNum; i++) free(r->rep[i].cig);
    free(r->rep); free(r->enc);
}

typedef struct job_s {
    ctx_t cx;
    int n_reads, paired, begin, end;    /* read index range [begin,end), pair aligned */
    const uint32_t *seq_off; const uint16_t *rlen; const char *seq;
    read_t *reads;                      /* shared array; each thread touches its own range */
    sjvec sj;
    /* output phase */
    size_t n_rep, n_cig, n_sj, base_rep, base_cig, base_sj;
    orc_read_out *ro; orc_report_out *po; uint32_t *cigar_ops; orc_sj_out *so;
    const size_t *caps; int overflow;
    pthread_barrier_t *bar; struct job_s *all; int n_threads, tid;
} job_t;

static void map_range(job_t *jb)
{
    ctx_t *cx = &jb->cx;
    int i;
    seedvec s1 = {0, 0, 0}, s2 = {0, 0, 0};
    candvec c1 = {0, 0, 0}, c2 = {0, 0, 0};
    for (i = jb->begin; i < jb->end; i++) {   /* NUL-terminated private copy like ReadItem_t::seq */
        char *z = (char *)malloc((size_t)jb->rlen[i] + 1);
        memcpy(z, jb->seq + jb->seq_off[i], jb->rlen[i]); z[jb->rlen[i]] = 0;
        read_init(&jb->reads[i], z, jb->rlen[i]);
    }
    for (i = jb->begin; i < jb->end;) {
        if (jb->paired && i + 1 < jb->end) {
            read_t *r1 = &jb->reads[i], *r2 = &jb->reads[i + 1];
            identify_seed_pairs(cx, r1->rlen, r1->enc, &s1);
            generate_candidates(cx, r1->rlen, &s1, &c1);
            identify_seed_pairs(cx, r2->rlen, r2->enc, &s2);
            generate_candidates(cx, r2->rlen, &s2, &c2);
            if (check_paired_candidates(&c1, &c2)) remove_unmated(&c1, &c2);
            remove_redundant(&c1); remove_redundant(&c2);
            gen_mapping_report(cx, 1, r1, &c1);
            gen_mapping_report(cx, 0, r2, &c2);
            check_paired_final(cx->pr, r1, r2);
            set_paired_flag(r1, r2);
            evaluate_mapq(r1); evaluate_mapq(r2);
            if (r1->mapq == 50 || (cx->pr->all_sj && r1->score > 0)) collect_sj(cx, &c1.a[r1->iBest], i, &jb->sj);
            if (r2->mapq == 50 || (cx->pr->all_sj && r2->score > 0)) collect_sj(cx, &c2.a[r2->iBest], i + 1, &jb->sj);
            candvec_free(&c1); candvec_free(&c2);
            i += 2;
        } else {
            read_t *r1 = &jb->reads[i];
            identify_seed_pairs(cx, r1->rlen, r1->enc, &s1);
            generate_candidates(cx, r1->rlen, &s1, &c1);
            remove_redundant(&c1);
            gen_mapping_report(cx, 1, r1, &c1);
            set_single_flag(r1); evaluate_mapq(r1);
            if (r1->mapq == 50 || (cx->pr->all_sj && r1->score > 0)) collect_sj(cx, &c1.a[r1->iBest], i, &jb->sj);
            candvec_free(&c1);
            i += 1;
        }
    }
    sv_free(&s1); sv_free(&s2);
    for (i = jb->begin; i < jb->end; i++) {
        int k;
        jb->n_rep += (size_t)jb->reads[i].CanNum;
        for (k = 0; k < jb->reads[i].CanNum; k++) jb->n_cig += (size_t)jb->reads[i].rep[k].ncig;
    }
    jb->n_sj = (size_t)jb->sj.n;
}

/* every thread writes the flat records of its own read range at its prefix offsets */
static void emit_range(job_t *jb)
{
    size_t nrep = jb->base_rep, ncig = jb->base_cig, nsj = jb->base_sj;
    int i, k, sjpos = 0;
    if (nrep + jb->n_rep > jb->caps[0] || ncig + jb->n_cig > jb->caps[1] || nsj + jb->n_sj > jb->caps[2]) jb->overflow = 1;
    for (i = jb->begin; i < jb->end; i++) {
        read_t *r = &jb->reads[i];
        if (!jb->overflow) {
            orc_read_out *o = &jb->ro[i];
            o->score = r->score; o->sub_score = r->sub_score; o->mis_num = r->mis_num; o->mapq = r->mapq;
            o->n_rep = r->CanNum; o->best = r->iBest; o->rep_off = (int32_t)nrep;
            for (k = 0; k < r->CanNum; k++) {
                orc_report_out *p = &jb->po[nrep++];
                report_t *rp = &r->rep[k];
                p->aln_score = rp->AlnScore; p->sj_type = rp->SJtype; p->flag = rp->iFrag; p->paired_idx = rp->PairedIdx;
                p->chr = rp->chr; p->bdir = rp->bDir; p->pos = rp->gPos;
                p->cigar_off = (uint32_t)ncig; p->n_cigar = (uint32_t)rp->ncig;
                if (rp->ncig) memcpy(jb->cigar_ops + ncig, rp->cig, (size_t)rp->ncig * sizeof(uint32_t));
                ncig += (size_t)rp->ncig;
            }
            o->sj_off = (int32_t)nsj; o->n_sj = 0;
            while (sjpos < jb->sj.n && jb->sj.a[sjpos].read_idx == i) { jb->so[nsj++] = jb->sj.a[sjpos++]; o->n_sj++; }
        }
        free((char *)r->seq);
        read_free(r);
    }
}

static void *map_thread(void *p)
{
    job_t *jb = (job_t *)p;
    map_range(jb);
    if (jb->bar) pthread_barrier_wait(jb->bar);
    if (jb->tid == 0) {
        size_t a = 0, b = 0, c = 0; int t;
        for (t = 0; t < jb->n_threads; t++) { jb->all[t].base_rep = a; jb->all[t].base_cig = b; jb->all[t].base_sj = c; a += jb->all[t].n_rep; b += jb->all[t].n_cig; c += jb->all[t].n_sj; }
    }
    if (jb->bar) pthread_barrier_wait(jb->bar);
    emit_range(jb);
    return 0;
}

int orc_map_batch(const orc_index *ix, const orc_params *pr, int n_reads, const uint32_t *seq_off,
                  const uint16_t *rlen, const char *seq, orc_read_out *ro, orc_report_out *po,
                  uint32_t *cigar_ops, orc_sj_out *so, const size_t caps[3], size_t used[3],
                  int n_threads, orc_counters *ctr)
{
    int t, k, rc = 0, per;
    read_t *reads = (read_t *)calloc((size_t)(n_reads > 0 ? n_reads : 1), sizeof(read_t));
    job_t *jobs;
    pthread_t *th;
    pthread_barrier_t bar;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_reads / 2 + 1) n_threads = n_reads / 2 + 1;
    jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    per = ((n_reads + n_threads - 1) / n_threads + 1) & ~1;
    if (n_threads > 1) pthread_barrier_init(&bar, 0, (unsigned)n_threads);
    for (t = 0; t < n_threads; t++) {
        job_t *jb = &jobs[t];
        jb->cx.ix = ix; jb->cx.pr = pr;
        jb->paired = pr->paired && (n_reads % 2 == 0);
        jb->begin = t * per < n_reads ? t * per : n_reads;
        jb->end = (t + 1) * per < n_reads ? (t + 1) * per : n_reads;
        jb->reads = reads; jb->seq_off = seq_off; jb->rlen = rlen; jb->seq = seq;
        jb->ro = ro; jb->po = po; jb->cigar_ops = cigar_ops; jb->so = so; jb->caps = caps;
        jb->bar = n_threads > 1 ? &bar : 0; jb->all = jobs; jb->n_threads = n_threads; jb->tid = t;
    }
    if (n_threads > 1) {
        for (t = 0; t < n_threads; t++) pthread_create(&th[t], 0, map_thread, &jobs[t]);
        for (t = 0; t < n_threads; t++) pthread_join(th[t], 0);
        pthread_barrier_destroy(&bar);
    } else map_thread(&jobs[0]);
    if (ctr) memset(ctr, 0, sizeof *ctr);
    used[0] = used[1] = used[2] = 0;
    for (t = 0; t < n_threads; t++) {
        if (ctr) {
            uint64_t *d = (uint64_t *)ctr; const uint64_t *s = (const uint64_t *)&jobs[t].cx.c;
            for (k = 0; k < (int)(sizeof(orc_counters) / sizeof(uint64_t)); k++) d[k] += s[k];
        }
        if (jobs[t].overflow) rc = -1;
        used[0] += jobs[t].n_rep; used[1] += jobs[t].n_cig; used[2] += jobs[t].n_sj;
        free(jobs[t].sj.a);
    }
    free(reads); free(jobs); free(th);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * stage probes
 * ---------------------------------------------------------------------------------------- */
int orc_bwt_search(const orc_index *ix, const orc_params *pr, const uint8_t *enc, int start, int stop, int *len, uint64_t *locs)
{
    ctx_t cx; memset(&cx, 0, sizeof cx); cx.ix = ix; cx.pr = pr;
    return bwt_search(&cx, enc, start, stop, len, locs);
}

int orc_seeds(const orc_index *ix, const orc_params *pr, const char *seq, int rlen, int32_t *rpos, int32_t *slen, int64_t *gpos, int cap)
{
    ctx_t cx; seedvec sv = {0, 0, 0}; int i, n;
    uint8_t *enc = (uint8_t *)malloc((size_t)rlen + 1);
    memset(&cx, 0, sizeof cx); cx.ix = ix; cx.pr = pr;
    for (i = 0; i < rlen; i++) enc[i] = nt4((unsigned char)seq[i]);
    identify_seed_pairs(&cx, rlen, enc, &sv);
    n = sv.n < cap ? sv.n : cap;
    for (i = 0; i < n; i++) { rpos[i] = sv.a[i].rPos; slen[i] = sv.a[i].rLen; gpos[i] = sv.a[i].gPos; }
    n = sv.n;
    sv_free(&sv); free(enc);
    return n;
}

/* stage probe for the host-compiled checks of the kernels' list helpers (tests/native/report_checks.hip): one of the list
 * passes of GenMappingReport on a caller-supplied seed list, in place.  op 0 = RemoveTandemRepeatSeeds + RemoveTranslocatedSeeds,
 * 1 = IdentifyNormalPairs (with CheckOverlappingSeeds), 2 = CheckSpliceJunction (returns the SJ type).  flags: bit 0 bSimple,
 * bit 1 bAcceptorSite.  Returns the new list length through *n (cap = array capacity).                                          */
int orc_seed_stage(const orc_index *ix, const orc_params *pr, int op, int *n, int cap, int32_t *rpos, int32_t *rlen, int32_t *glen, int64_t *gpos, uint32_t *flags)
{
    ctx_t cx; seedvec v; int i, ret = 0;
    memset(&cx, 0, sizeof cx); memset(&v, 0, sizeof v);
    cx.ix = ix; cx.pr = pr;
    for (i = 0; i < *n; i++) {
        seed_t s; memset(&s, 0, sizeof s);
        s.rPos = rpos[i]; s.rLen = rlen[i]; s.gLen = glen[i]; s.gPos = gpos[i]; s.PosDiff = s.gPos - s.rPos;
        s.simple = (uint8_t)(flags[i] & 1u); s.acceptor = (uint8_t)((flags[i] >> 1) & 1u);
        sv_push(&v, &s);
    }
    if (op == 0) { remove_tandem_repeat_seeds(&v); remove_translocated_seeds(&v); }
    else if (op == 1) identify_normal_pairs(&v);
    else ret = check_splice_junction(&cx, &v);
    if (v.n > cap) { sv_free(&v); return -99; }
    for (i = 0; i < v.n; i++) {
        rpos[i] = v.a[i].rPos; rlen[i] = v.a[i].rLen; glen[i] = v.a[i].gLen; gpos[i] = v.a[i].gPos;
        flags[i] = (uint32_t)v.a[i].simple | ((uint32_t)v.a[i].acceptor << 1);
    }
    *n = v.n;
    sv_free(&v);
    return ret;
}

/* ------------------------------------------------------------------------------------------
 * index files  (bwt_index.cpp:15-121,229-251)
 * ---------------------------------------------------------------------------------------- */
static void *slurp(const char *fn, size_t *sz)
{
    FILE *f = fopen(fn, "rb");
    void *buf; long n;
    if (!f) return 0;
    fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    buf = malloc((size_t)n + 64);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return 0; }
    fclose(f); *sz = (size_t)n;
    return buf;
}

void orc_params_default(orc_params *p)
{
    p->max_gaps = 5; p->max_dup = 100; p->max_intron = 500000; p->min_intron = 5;
    p->max_mismatch = 0; p->multi_hit = 0; p->all_sj = 0; p->paired = 0;
}

orc_index *orc_index_load(const char *prefix)
{
    char fn[4096];
    size_t sz;
    orc_index *ix = (orc_index *)calloc(1, sizeof *ix);
    uint8_t *b; uint64_t *s; FILE *f;
    long long xx; int i, n_seqs; unsigned seed;
    int64_t *off, *len, total = 0;

    snprintf(fn, sizeof fn, "%s.bwt", prefix);
    if (!(b = (uint8_t *)slurp(fn, &sz))) goto fail;
    ix->owned[0] = b;
    memcpy(&ix->primary, b, 8); memcpy(&ix->L2[1], b + 8, 32); ix->L2[0] = 0;
    ix->bwt = (const uint32_t *)(b + 40); ix->bwt_words = (sz - 40) >> 2;
    ix->seq_len = ix->L2[4];

    snprintf(fn, sizeof fn, "%s.sa", prefix);
    if (!(s = (uint64_t *)slurp(fn, &sz))) goto fail;
    ix->owned[1] = s;
    ix->sa_intv = (int)s[5];
    ix->n_sa = (ix->seq_len + (uint64_t)ix->sa_intv) / (uint64_t)ix->sa_intv;
    s[6] = (uint64_t)-1;           /* sa[0] = -1 sits where the header's seq_len was */
    ix->sa = s + 6;

    snprintf(fn, sizeof fn, "%s.ann", prefix);
    if (!(f = fopen(fn, "r"))) goto fail;
    if (fscanf(f, "%lld%d%u", &xx, &n_seqs, &seed) != 3) { fclose(f); goto fail; }
    ix->l_pac = xx; ix->n_chr = n_seqs;
    off = (int64_t *)calloc((size_t)n_seqs, 8); len = (int64_t *)calloc((size_t)n_seqs, 8);
    ix->chr_name = (char **)calloc((size_t)n_seqs, sizeof(char *));
    ix->owned[2] = off; ix->owned[3] = len;
    for (i = 0; i < n_seqs; i++) {
        unsigned gi; char name[1024]; int c, l, namb;
        if (fscanf(f, "%u%1023s", &gi, name) != 2) { fclose(f); goto fail; }
        ix->chr_name[i] = strdup(name);
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &xx, &l, &namb) != 3) { fclose(f); goto fail; }
        len[i] = l;
        off[i] = total; total += l;       /* FowardLocation, bwt_index.cpp:246 */
    }
    fclose(f);
    ix->chr_off = off; ix->chr_len = len;

    snprintf(fn, sizeof fn, "%s.pac", prefix);
    if (!(b = (uint8_t *)slurp(fn, &sz))) goto fail;
    ix->owned[4] = b; ix->pac = b;

    /* ChrLocMap: key = last coordinate of each chromosome in both halves; std::map orders keys */
    ix->loc_key = (int64_t *)calloc((size_t)(2 * n_seqs), 8);
    ix->loc_chr = (int *)calloc((size_t)(2 * n_seqs), sizeof(int));
    for (i = 0; i < n_seqs; i++) {
        int64_t rev = 2 * ix->l_pac - (off[i] + len[i]);     /* ReverseLocation :247 */
        ix->loc_key[i] = off[i] + len[i] - 1; ix->loc_chr[i] = i;
        ix->loc_key[2 * n_seqs - 1 - i] = rev + len[i] - 1; ix->loc_chr[2 * n_seqs - 1 - i] = i;
    }
    return ix;
fail:
    orc_index_free(ix);
    return 0;
}

void orc_index_free(orc_index *ix)
{
    int i;
    if (!ix) return;
    for (i = 0; i < 8; i++) free(ix->owned[i]);
    if (ix->chr_name) { for (i = 0; i < ix->n_chr; i++) free(ix->chr_name[i]); free(ix->chr_name); }
    free(ix->loc_key); free(ix->loc_chr);
    free(ix);
}
