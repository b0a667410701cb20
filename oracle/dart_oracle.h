/* oracle/dart_oracle.h -- TEST INFRASTRUCTURE (the parity checker), not product code.
 *
 * A plain-C CPU restatement of DART v1.4.6's per-read mapping path, written from the behaviour
 * documented in SURVEY.md section 8a; every function in dart_oracle.c cites the reference
 * file:line it restates.  Pinned against the reference's own object code (oracle/_ref, built by
 * oracle/Makefile from /root/reference) and against the committed fixtures in tests/golden/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 */
#ifndef DART_ORACLE_H
#define DART_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t primary, L2[5], seq_len;   /* .bwt header (bwt_index.cpp:102-121)            */
    const uint32_t *bwt; uint64_t bwt_words;
    const uint64_t *sa; uint64_t n_sa; int sa_intv; /* sa[0] = -1 (bwt_index.cpp:15-35)   */
    const uint8_t *pac; int64_t l_pac;  /* forward strand, 2 bit/base MSB first            */
    int n_chr; const int64_t *chr_off, *chr_len; char **chr_name;
    /* ChrLocMap as a sorted array (bwt_index.cpp:241-251): 2*n_chr keys               */
    int64_t *loc_key; int *loc_chr;
    void *owned[8];
} orc_index;

typedef struct {
    int max_gaps;      /* MaxGaps 5            main.cpp:101 */
    int max_dup;       /* MaxDupNum 100        main.cpp:102 */
    int max_intron;    /* MaxIntronSize 500000 main.cpp:110 */
    int min_intron;    /* MinIntronSize 5      main.cpp:111 */
    int max_mismatch;  /* MaxMismatch 0        main.cpp:17 (never initialised -> 0) */
    int multi_hit;     /* -m                   */
    int all_sj;        /* -all_sj              */
    int paired;        /* bPairEnd             */
} orc_params;

/* flat result records: same shape as include/dartgpu.h's dg_* records */
typedef struct { int32_t score, sub_score, mis_num, mapq, n_rep, best, rep_off, sj_off, n_sj; } orc_read_out;
typedef struct { int32_t aln_score, sj_type, flag, paired_idx, chr, bdir; int64_t pos; uint32_t cigar_off, n_cigar; } orc_report_out;
typedef struct { int64_t g1, g2; int32_t type, read_idx; } orc_sj_out;

/* traffic counters of the reference's algorithm+layout (SURVEY 8d "algorithmic bytes") */
typedef struct { uint64_t n_occ_blocks, n_lf, n_sa, n_search, n_2occ4, n_nw, nw_cells, n_reseed, reseed_window, ref_bases; } orc_counters;

orc_index *orc_index_load(const char *prefix);
void       orc_index_free(orc_index *);
void       orc_params_default(orc_params *);

/* Maps n_reads reads (pairs are reads 2i,2i+1; mate 2 already reverse-complemented as
 * GetData.cpp:157-162 does).  seq = ASCII bases, read i at seq[seq_off[i]] of length rlen[i].
 * caps/used: [0]=reports [1]=cigar ops [2]=sj.  Returns 0, or -1 when a capacity is too small. */
int orc_map_batch(const orc_index *, const orc_params *, int n_reads, const uint32_t *seq_off,
                  const uint16_t *rlen, const char *seq, orc_read_out *, orc_report_out *,
                  uint32_t *cigar_ops, orc_sj_out *, const size_t caps[3], size_t used[3],
                  int n_threads, orc_counters *ctr);

/* stage probes for unit parity */
int  orc_nw(const char *s1, const char *s2, char *out1, char *out2, int cap);
/* BWT_Search from `start`: returns freq (0 = no hit), *len set when freq>0; locs[<=max_dup] */
int  orc_bwt_search(const orc_index *, const orc_params *, const uint8_t *enc, int start, int stop, int *len, uint64_t *locs);
/* seeds of one read after the (gPos,rPos) sort; returns count (<= cap) */
int  orc_seeds(const orc_index *, const orc_params *, const char *seq, int rlen, int32_t *rpos, int32_t *slen, int64_t *gpos, int cap);
char orc_refbase(const orc_index *, int64_t g);
/* one list pass of GenMappingReport on a caller-supplied seed list (see dart_oracle.c) */
int  orc_seed_stage(const orc_index *, const orc_params *, int op, int *n, int cap, int32_t *rpos, int32_t *rlen, int32_t *glen, int64_t *gpos, uint32_t *flags);

#ifdef __cplusplus
}
#endif
#endif
