/* oracle/oracle_cli.c -- TEST INFRASTRUCTURE: `dart_oracle`, a CPU-only command line over the
 * oracle with DART's flags, used (a) to validate the oracle end to end against
 * oracle/_ref/ref_harness (SAM + junctions.tab byte for byte) and (b) as the "port" CPU
 * baseline in bench.py.  Restates main.cpp:96-239 (flags), GetData.cpp:44-247 (readers),
 * Mapping.cpp:208-369 (SAM records), :532-577,683-716 (junction table), :741-751 (header),
 * :812-822 (stats).
 */
#define _GNU_SOURCE
#include "dart_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <zlib.h>

typedef struct { char *header, *seq, *qual; int rlen; } entry_t;

static int fastq_format = 1, gz_compressed = 0, pair_end = 0;

static char comp_base(char c)   /* tools.cpp:3-17 */
{
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    default: return 'N';
    }
}
static void revcomp(int len, const char *seq, char *rseq)   /* tools.cpp:19-29 */
{
    int i;
    for (i = 0; i < len; i++) rseq[i] = comp_base(seq[len - 1 - i]);
}

static int hdr_beg(const char *s, int len) { int i; for (i = 1; i < len; i++) if (s[i] != '>' && s[i] != '@') return i; return len - 1; }   /* GetData.cpp:55-64 */
static int hdr_end(const char *s, int len) { int i; for (i = 1; i < len; i++) if (s[i] == ' ' || s[i] == '/' || s[i] == '\t') return i; return len - 1; }   /* :66-75 */

static entry_t next_entry(FILE *f)   /* GetNextEntry, GetData.cpp:77-132 */
{
    entry_t e; char *buf = NULL; size_t cap = 0; ssize_t len;
    memset(&e, 0, sizeof e);
    if ((len = getline(&buf, &cap, f)) != -1) {
        int p1 = hdr_beg(buf, (int)len), p2 = hdr_end(buf, (int)len), hl = p2 - p1;
        if (hl < 0) hl = 0;
        e.header = (char *)malloc((size_t)hl + 1); memcpy(e.header, buf + p1, (size_t)hl); e.header[hl] = 0;
        if (fastq_format) {
            ssize_t rl;
            if ((rl = getline(&buf, &cap, f)) != -1) {
                e.seq = (char *)malloc((size_t)rl + 1); memcpy(e.seq, buf, (size_t)rl);
                if (getline(&buf, &cap, f) == -1) {}
                if (getline(&buf, &cap, f) == -1) buf[0] = 0;
                e.qual = (char *)calloc((size_t)rl + 1, 1); strncpy(e.qual, buf, (size_t)rl);
                e.rlen = (int)rl - 1; e.seq[e.rlen] = 0; e.qual[e.rlen] = 0;
            } else e.rlen = 0;
        } else {
            size_t sl = 0, sc = 256; char *s = (char *)malloc(sc);
            s[0] = 0;
            while ((len = getline(&buf, &cap, f)) != -1) {
                if (buf[0] == '>') { fseek(f, 0 - (long)len, SEEK_CUR); break; }
                buf[len - 1] = 0;
                { size_t l = strlen(buf); if (sl + l + 1 > sc) { sc = (sl + l + 1) * 2; s = (char *)realloc(s, sc); } memcpy(s + sl, buf, l + 1); sl += l; }
            }
            if ((e.rlen = (int)sl) > 0) e.seq = s; else free(s);
        }
    }
    free(buf);
    return e;
}

static entry_t gz_next_entry(gzFile f)   /* gzGetNextEntry, GetData.cpp:181-210 */
{
    entry_t e; char buf[1024];
    memset(&e, 0, sizeof e);
    if (gzgets(f, buf, 1024) != NULL) {
        int len = (int)strlen(buf), p1 = hdr_beg(buf, len), p2 = hdr_end(buf, len);
        len = p2 - p1;
        if (len > 0 && (buf[0] == '@' || buf[0] == '>')) {
            e.header = (char *)malloc((size_t)len + 1); memcpy(e.header, buf + p1, (size_t)len); e.header[len] = 0;
            if (gzgets(f, buf, 1024) == NULL) buf[0] = '\n', buf[1] = 0;
            e.rlen = (int)strlen(buf) - 1; e.seq = (char *)malloc((size_t)e.rlen + 1); memcpy(e.seq, buf, (size_t)e.rlen); e.seq[e.rlen] = 0;
            if (fastq_format) {
                if (gzgets(f, buf, 1024) == NULL || gzgets(f, buf, 1024) == NULL) buf[0] = 0;
                e.qual = (char *)calloc((size_t)e.rlen + 1, 1); strncpy(e.qual, buf, (size_t)e.rlen);
            }
        }
    }
    return e;
}

typedef struct { entry_t *a; size_t n, m; } entvec;
static void ev_push(entvec *v, entry_t e) { if (v->n == v->m) { v->m = v->m ? v->m * 2 : 4096; v->a = (entry_t *)realloc(v->a, v->m * sizeof(entry_t)); } v->a[v->n++] = e; }

/* GetNextChunk / gzGetNextChunk, GetData.cpp:134-179,212-247: appends one chunk, returns its size */
static int next_chunk(int sep, FILE *f1, FILE *f2, gzFile g1, gzFile g2, entvec *out)
{
    int count = 0, base = 0;
    while (1) {
        entry_t e = gz_compressed ? gz_next_entry(g1) : next_entry(f1);
        if (e.rlen == 0) { free(e.header); free(e.seq); free(e.qual); break; }
        ev_push(out, e); base += e.rlen; count++;
        e = gz_compressed ? gz_next_entry(sep ? g2 : g1) : next_entry(sep ? f2 : f1);
        if (e.rlen == 0) { free(e.header); free(e.seq); free(e.qual); break; }
        if (pair_end) {
            char *r = (char *)malloc((size_t)e.rlen + 1);
            int i;
            revcomp(e.rlen, e.seq, r); memcpy(e.seq, r, (size_t)e.rlen); free(r);
            if (fastq_format) for (i = 0; i < e.rlen / 2; i++) { char c = e.qual[i]; e.qual[i] = e.qual[e.rlen - 1 - i]; e.qual[e.rlen - 1 - i] = c; }
        }
        ev_push(out, e); base += e.rlen; count++;
        if (count == 4000 || base > 1000000) break;
    }
    return count;
}

static int check_read_format(const char *fn)   /* Mapping.cpp:718-726 */
{
    char b[1] = {0}; gzFile f = gzopen(fn, "rb");
    if (!f) return 0;
    gzread(f, b, 1); gzclose(f);
    return b[0] == '@';
}

/* ---- junction table ---- */
typedef struct { int64_t g1, g2; int type, count; } sj_t;
static int cmp_sj(const void *a, const void *b)
{
    const sj_t *p = (const sj_t *)a, *q = (const sj_t *)b;
    if (p->g1 != q->g1) return p->g1 < q->g1 ? -1 : 1;
    if (p->g2 != q->g2) return p->g2 < q->g2 ? -1 : 1;
    return 0;
}

static const char *XS_A[3] = { "", " XS:A:+", " XS:A:-" };

typedef struct { char *s; size_t n, m; } sbuf;
static void sb_need(sbuf *b, size_t k) { if (b->n + k + 1 > b->m) { b->m = (b->n + k + 1) * 2; b->s = (char *)realloc(b->s, b->m); } }
static void sb_cigar(char *dst, const uint32_t *ops, int n)
{
    int i; char *p = dst;
    for (i = 0; i < n; i++) p += sprintf(p, "%u%c", ops[i] >> 4, "MIDNS??????????*"[ops[i] & 15]);
    *p = 0;
}

int main(int argc, char **argv)
{
    orc_params pr; orc_index *ix;
    const char *index = NULL, *out_name = "output.sam", *sj_name = "junctions.tab";
    char **f1 = (char **)calloc((size_t)argc, sizeof(char *)), **f2 = (char **)calloc((size_t)argc, sizeof(char *));
    int nf1 = 0, nf2 = 0, threads = 4, unique = 0, silent = 0, i, lib;
    int64_t total = 0, n_unique = 0, n_unmapped = 0, n_paired = 0;
    sj_t *sj = NULL; size_t nsj = 0, msj = 0;
    FILE *sam;
    double t_map = 0, t_load = 0;
    struct timeval tv0, tv1;

    orc_params_default(&pr);
    if (argc == 1 || !strcmp(argv[1], "-h")) { fprintf(stdout, "dart_oracle: CPU restatement of DART v1.4.6 (test oracle)\n"); return 0; }
    for (i = 1; i < argc; i++) {   /* main.cpp:136-205 */
        const char *p = argv[i];
        if (!strcmp(p, "-i")) index = argv[++i];
        else if (!strcmp(p, "-f")) { while (++i < argc && argv[i][0] != '-') f1[nf1++] = argv[i]; i--; }
        else if (!strcmp(p, "-f2")) { while (++i < argc && argv[i][0] != '-') f2[nf2++] = argv[i]; i--; }
        else if (!strcmp(p, "-t")) { if ((threads = atoi(argv[++i])) <= 0) { fprintf(stdout, "Warning! Thread number should be a positive number!\n"); threads = 4; } }
        else if (!strcmp(p, "-o")) out_name = argv[++i];
        else if (!strcmp(p, "-mis") && i + 1 < argc) pr.max_mismatch = atoi(argv[++i]);
        else if (!strcmp(p, "-max_dup") && i + 1 < argc) { pr.max_dup = atoi(argv[++i]); if (pr.max_dup < 100) pr.max_dup = 100; else if (pr.max_dup >= 10000) pr.max_dup = 10000; }
        else if (!strcmp(p, "-silent")) silent = 1;
        else if (!strcmp(p, "-j")) sj_name = argv[++i];
        else if (!strcmp(p, "-p")) pair_end = 1;
        else if (!strcmp(p, "-m")) pr.multi_hit = 1;
        else if (!strcmp(p, "-unique")) unique = 1;
        else if (!strcmp(p, "-all_sj")) pr.all_sj = 1;
        else if (!strcmp(p, "-max_intron")) { if ((pr.max_intron = atoi(argv[++i])) < 100000) pr.max_intron = 100000; }
        else if (!strcmp(p, "-min_intron")) pr.min_intron = atoi(argv[++i]);
        else if (!strcmp(p, "-v") || !strcmp(p, "--version")) { fprintf(stdout, "DART v1.4.6\n\n"); return 0; }
        else { fprintf(stderr, "Error! Unknow parameter: %s\n", p); return 1; }
    }
    (void)silent;
    if (nf1 == 0) { fprintf(stderr, "Error! Please specify a valid read input!\n"); return 1; }
    if (nf2 > 0 && nf1 != nf2) { fprintf(stderr, "Error! Paired-end reads input numbers do not match!\n"); return 1; }
    gettimeofday(&tv0, 0);
    if (!index || !(ix = orc_index_load(index))) { fprintf(stderr, "Error! Please specify a valid reference index!\n"); return 1; }
    gettimeofday(&tv1, 0);
    t_load = (double)(tv1.tv_sec - tv0.tv_sec) + 1e-6 * (double)(tv1.tv_usec - tv0.tv_usec);

    sam = fopen(out_name, "w");
    fprintf(sam, "@PG\tID:Dart\tPN:Dart\tVN:1.4.6\n");
    for (i = 0; i < ix->n_chr; i++) fprintf(sam, "@SQ\tSN:%s\tLN:%lld\n", ix->chr_name[i], (long long)ix->chr_len[i]);

    for (lib = 0; lib < nf1; lib++) {
        FILE *h1 = NULL, *h2 = NULL; gzFile g1 = NULL, g2 = NULL;
        int sep;
        const char *dot = strrchr(f1[lib], '.');
        gz_compressed = dot && !strcmp(dot + 1, "gz");
        fastq_format = check_read_format(f1[lib]);
        if (gz_compressed) g1 = gzopen(f1[lib], "rb"); else h1 = fopen(f1[lib], "r");
        if (nf1 == nf2) {
            sep = pair_end = 1;
            if (fastq_format != check_read_format(f2[lib])) { fprintf(stderr, "Error! %s and %s are with different format...\n", f1[lib], f2[lib]); return 1; }
            if (gz_compressed) g2 = gzopen(f2[lib], "rb"); else h2 = fopen(f2[lib], "r");
        } else sep = 0;
        if (!h1 && !g1) continue;
        if (sep && !h2 && !g2) continue;

        while (1) {
            /* accumulate reference-sized chunks into one super-batch; an odd chunk (only the last
             * one can be) is mapped on its own as single reads, as Mapping.cpp:598 does */
            entvec ev = {0, 0, 0};
            int n, k, odd = 0;
            size_t bases = 0, off;
            uint32_t *seq_off; uint16_t *rl; char *seq;
            orc_read_out *ro; orc_report_out *po; uint32_t *cig; orc_sj_out *so;
            size_t caps[3], used[3];
            sbuf sb = {0, 0, 0};
            while (ev.n < 400000) {
                int c = next_chunk(sep, h1, h2, g1, g2, &ev);
                if (c == 0) break;
                if (c & 1) { odd = c; break; }   /* only the final chunk can be odd */
            }
            if (ev.n == 0) break;
            n = (int)ev.n;
            for (k = 0; k < n; k++) bases += (size_t)ev.a[k].rlen;
            seq_off = (uint32_t *)malloc((size_t)n * 4); rl = (uint16_t *)malloc((size_t)n * 2); seq = (char *)malloc(bases + 1);
            for (k = 0, off = 0; k < n; k++) { seq_off[k] = (uint32_t)off; rl[k] = (uint16_t)ev.a[k].rlen; memcpy(seq + off, ev.a[k].seq, (size_t)ev.a[k].rlen); off += (size_t)ev.a[k].rlen; }
            caps[0] = (size_t)n * 64 + 1024; caps[1] = (size_t)n * 256 + 4096; caps[2] = (size_t)n * 8 + 64;
            ro = (orc_read_out *)calloc((size_t)n, sizeof *ro); po = (orc_report_out *)calloc(caps[0], sizeof *po);
            cig = (uint32_t *)calloc(caps[1], 4); so = (orc_sj_out *)calloc(caps[2], sizeof *so);
            {
                orc_params p2 = pr;
                int n_even = n, rc;
                p2.paired = pair_end;
                n_even = n - odd;
                gettimeofday(&tv0, 0);
                used[0] = used[1] = used[2] = 0;
                rc = 0;
                if (n_even > 0) rc = orc_map_batch(ix, &p2, n_even, seq_off, rl, seq, ro, po, cig, so, caps, used, threads, NULL);
                if (rc == 0 && n_even < n) {
                    size_t caps2[3], used2[3]; int kk;
                    orc_params p3 = p2; p3.paired = 0;
                    caps2[0] = caps[0] - used[0]; caps2[1] = caps[1] - used[1]; caps2[2] = caps[2] - used[2];
                    rc = orc_map_batch(ix, &p3, n - n_even, seq_off + n_even, rl + n_even, seq, ro + n_even, po + used[0], cig + used[1], so + used[2], caps2, used2, threads, NULL);
                    for (kk = n_even; kk < n; kk++) { ro[kk].rep_off += (int32_t)used[0]; ro[kk].sj_off += (int32_t)used[2]; }
                    for (kk = 0; kk < (int)used2[0]; kk++) po[used[0] + (size_t)kk].cigar_off += (uint32_t)used[1];
                    for (kk = 0; kk < (int)used2[2]; kk++) so[used[2] + (size_t)kk].read_idx += n_even;
                    used[0] += used2[0]; used[1] += used2[1]; used[2] += used2[2];
                }
                gettimeofday(&tv1, 0);
                t_map += (double)(tv1.tv_sec - tv0.tv_sec) + 1e-6 * (double)(tv1.tv_usec - tv0.tv_usec);
                if (rc) { fprintf(stderr, "oracle: output capacity exceeded\n"); return 2; }

                /* ---- SAM records: OutputPairedAlignments / OutputSingledAlignments ---- */
                for (k = 0; k < n; k++) {
                    int is_pair = pair_end && k < n_even;
                    int mate2 = is_pair && (k & 1);
                    const entry_t *e = &ev.a[k];
                    const orc_read_out *r = &ro[k], *m = is_pair ? &ro[k ^ 1] : NULL;
                    const orc_report_out *rp = po + r->rep_off, *mp = m ? po + m->rep_off : NULL;
                    const entry_t *me = is_pair ? &ev.a[k ^ 1] : NULL;
                    const char *q = fastq_format ? e->qual : "*";
                    char *alt = NULL, *altq = NULL, cigar[4096];
                    int j;
                    if (r->score == 0) {
                        n_unmapped++;
                        sb_need(&sb, (size_t)e->rlen * 2 + strlen(e->header) + 128);
                        sb.n += (size_t)sprintf(sb.s + sb.n, "%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tAS:i:0\tXS:i:0\n", e->header, rp[0].flag, e->seq, q);
                        continue;
                    }
                    if (!(!unique || r->mapq > 3)) continue;
                    if (r->mapq == 50) n_unique++;
                    for (j = r->best; j < r->n_rep; j++) {
                        int print = is_pair ? (rp[j].aln_score > 0) : (rp[j].aln_score == r->score);
                        if (print) {
                            int xs, use_alt, pj;
                            const char *s_out, *q_out;
                            if (rp[j].sj_type == -1) xs = 0;
                            else if (rp[j].sj_type == 0 || rp[j].sj_type == 2) xs = mate2 ? 2 : 1;
                            else xs = mate2 ? 1 : 2;
                            /* stored seq of mate 2 is the reverse complement of what was sequenced */
                            use_alt = mate2 ? (rp[j].bdir == 1) : (rp[j].bdir == 0);
                            if (use_alt && !alt) {
                                alt = (char *)malloc((size_t)e->rlen + 1); revcomp(e->rlen, e->seq, alt); alt[e->rlen] = 0;
                                if (fastq_format) { int z; altq = (char *)malloc((size_t)e->rlen + 1); for (z = 0; z < e->rlen; z++) altq[z] = e->qual[e->rlen - 1 - z]; altq[e->rlen] = 0; }
                            }
                            s_out = use_alt ? alt : e->seq;
                            q_out = fastq_format ? (use_alt ? altq : e->qual) : "*";
                            sb_cigar(cigar, cig + rp[j].cigar_off, (int)rp[j].n_cigar);
                            sb_need(&sb, (size_t)e->rlen * 2 + strlen(e->header) + strlen(cigar) + 256);
                            if (is_pair && (pj = rp[j].paired_idx) != -1 && mp[pj].aln_score > 0) {
                                const orc_report_out *a = mate2 ? &mp[pj] : &rp[j], *b = mate2 ? &rp[j] : &mp[pj];   /* a = read1's, b = read2's */
                                int r1len = mate2 ? me->rlen : e->rlen, r2len = mate2 ? e->rlen : me->rlen;
                                int dist = (int)(b->pos - a->pos + (a->bdir ? r2len : 0 - r1len));
                                if (mate2) dist = 0 - dist;
                                else if (j == r->best) n_paired += 2;
                                sb.n += (size_t)sprintf(sb.s + sb.n, "%s\t%d\t%s\t%lld\t%d\t%s\t=\t%lld\t%d\t%s\t%s\tNM:i:%d\tAS:i:%d\tXS:i:%d%s\n", e->header, rp[j].flag, ix->chr_name[rp[j].chr], (long long)rp[j].pos, r->mapq, cigar, (long long)mp[pj].pos, dist, s_out, q_out, r->mis_num, r->score, r->sub_score, XS_A[xs]);
                            } else
                                sb.n += (size_t)sprintf(sb.s + sb.n, "%s\t%d\t%s\t%lld\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\tAS:i:%d\tXS:i:%d%s\n", e->header, rp[j].flag, ix->chr_name[rp[j].chr], (long long)rp[j].pos, r->mapq, cigar, s_out, q_out, r->mis_num, r->score, r->sub_score, XS_A[xs]);
                            if (!is_pair && !pr.multi_hit) break;
                        }
                        if (is_pair && !pr.multi_hit) break;
                    }
                    free(alt); free(altq);
                }
                if (sb.n) fwrite(sb.s, 1, sb.n, sam);
                /* junction tuples -> map */
                for (k = 0; k < (int)used[2]; k++) {
                    if (nsj == msj) { msj = msj ? msj * 2 : 1024; sj = (sj_t *)realloc(sj, msj * sizeof(sj_t)); }
                    sj[nsj].g1 = so[k].g1; sj[nsj].g2 = so[k].g2; sj[nsj].type = so[k].type; sj[nsj].count = 1; nsj++;
                }
            }
            total += n;
            for (k = 0; k < n; k++) { free(ev.a[k].header); free(ev.a[k].seq); free(ev.a[k].qual); }
            free(ev.a); free(seq_off); free(rl); free(seq); free(ro); free(po); free(cig); free(so); free(sb.s);
            if (odd) break;
        }
        if (h1) fclose(h1);
        if (h2) fclose(h2);
        if (g1) gzclose(g1);
        if (g2) gzclose(g2);
    }
    fclose(sam);

    if (total > 0) {   /* Mapping.cpp:812-822 */
        FILE *jf; int nj = 0; size_t a, b;
        if (pair_end) fprintf(stdout, "\t# of total mapped reads = %lld (sensitivity = %.2f%%)\n\t# of paired sequences = %lld (%.2f%%)\n", (long long)(total - n_unmapped), (int)(10000 * (1.0 * (total - n_unmapped) / total) + 0.5) / 100.0, (long long)n_paired, (int)(10000 * (1.0 * n_paired / total) + 0.5) / 100.0);
        else fprintf(stdout, "\t# of total mapped reads = %lld (sensitivity = %.2f%%)\n", (long long)(total - n_unmapped), (int)(10000 * (1.0 * (total - n_unmapped) / total) + 0.5) / 100.0);
        fprintf(stdout, "\t# of unique mapped reads = %lld (%.2f%%)\n", (long long)n_unique, (int)(10000 * (1.0 * n_unique / total) + 0.5) / 100.0);
        if (!unique) fprintf(stdout, "\t# of multiple mapped reads = %lld (%.2f%%)\n", (long long)(total - n_unmapped - n_unique), (int)(10000 * (1.0 * (total - n_unmapped - n_unique) / total) + 0.5) / 100.0);
        fprintf(stdout, "\t# of unmapped reads = %lld (%.2f%%)\n", (long long)n_unmapped, (int)(10000 * (1.0 * n_unmapped / total) + 0.5) / 100.0);
        /* OutputSpliceJunctions :697-716 ; std::map order = (g1,g2); the type kept is the first inserted */
        jf = fopen(sj_name, "w");
        if (nsj) {
            /* stable: keep first type -> sort indices by key then original order (qsort is not stable, type is not printed) */
            qsort(sj, nsj, sizeof(sj_t), cmp_sj);
            for (a = 0; a < nsj; a = b) {
                int cnt = 0, lo = 0, hi = 2 * ix->n_chr, chr;
                for (b = a; b < nsj && sj[b].g1 == sj[a].g1 && sj[b].g2 == sj[a].g2; b++) cnt++;
                while (lo < hi) { int mid = (lo + hi) >> 1; if (ix->loc_key[mid] < sj[a].g1) lo = mid + 1; else hi = mid; }
                if (lo == 2 * ix->n_chr) continue;
                chr = ix->loc_chr[lo];
                fprintf(jf, "%s\t%lld\t%lld\t%d\n", ix->chr_name[chr], (long long)(sj[a].g1 + 1 - ix->chr_off[chr]), (long long)(sj[a].g2 + 1 - ix->chr_off[chr]), cnt);
                nj++;
            }
        }
        fclose(jf);
        fprintf(stdout, "\t# of splice junctions = %d (file: %s)\n", nj, sj_name);
        fprintf(stdout, "\tAlignment output: %s\n\n", out_name);
        fprintf(stderr, "[dart_oracle] index load %.3f s, mapping phase %.3f s, %lld reads, %d threads -> %.1f reads/s\n", t_load, t_map, (long long)total, threads, total / (t_map > 0 ? t_map : 1e-9));
    }
    orc_index_free(ix);
    return 0;
}
