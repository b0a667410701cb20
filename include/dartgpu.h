/* include/dartgpu.h -- C ABI of libdartgpu.so: DART's per-read mapping path on one MI355X.
 *
 * The drop-in seam is the chunk body of the reference's ReadMapping (Mapping.cpp:598-643):
 * a batch of reads goes in, and for every read the fields that the reference leaves in
 * ReadItem_t (structure.h:149-164: score, sub_score, mis_num, mapq, CanNum, iBestAlnCanIdx,
 * AlnReportArr[]) plus the splice-junction tuples UpdateLocalSJMap (Mapping.cpp:532-565) would
 * have added come out as flat records.  Plain pointers and sizes only; the library owns all
 * device memory; the caller owns the host arrays.  Every entry point returns 0 on success or a
 * negative dg_status; nothing here ever exits the process (the reference's only error
 * behaviour on this path is exit(1) in main.cpp:199-227, which stays in the host program).
 *
 * There is no CPU fallback: without a HIP device dg_init fails with DG_ERR_NO_DEVICE.
 */
#ifndef DARTGPU_H
#define DARTGPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dg_ctx dg_ctx;

enum dg_status {
    DG_OK = 0,
    DG_ERR_NO_DEVICE = -1,   /* no HIP device / device index out of range                 */
    DG_ERR_HIP = -2,         /* a HIP call failed; text in dg_last_error                  */
    DG_ERR_ARG = -3,         /* bad argument (NULL, read longer than DG_MAX_RLEN, ...)    */
    DG_ERR_CAPACITY = -4,    /* caller's output arrays too small; `used` holds the need   */
    DG_ERR_INTERNAL = -5,
    DG_ERR_RANGE = -6        /* a field does not fit the compact record types: use the full ones  */
};

#define DG_MAX_RLEN 1000     /* the reference's gz reader caps lines at 1024 bytes (GetData.cpp:186) */

/* The index exactly as the reference loads it (bwt_index.cpp:15-35,102-121,147-159,229-251):
 *   bwt   = contents of PREFIX.bwt after its 40-byte header (Occ-interleaved, 64 B per 128 rows)
 *   sa    = sa[0..n_sa): sa[0] = (uint64)-1, sa[i] = contents of PREFIX.sa after its 56-byte header
 *   pac   = PREFIX.pac, forward strand, 2 bits/base, MSB first
 *   chr_* = per chromosome offset in the forward concatenation and length (PREFIX.ann)        */
typedef struct {
    const uint32_t *bwt; uint64_t bwt_words, primary, L2[5], seq_len;
    const uint64_t *sa;  uint64_t n_sa; int32_t sa_intv;
    const uint8_t  *pac; int64_t l_pac;
    int32_t n_chr; const int64_t *chr_off; const int64_t *chr_len;
} dg_index_view;

/* Globals of the reference that steer the path (main.cpp:101-117,169-192) */
typedef struct {
    int32_t max_gaps;      /* MaxGaps       = 5       */
    int32_t max_dup;       /* MaxDupNum     = 100     (-max_dup, 100..10000)  */
    int32_t max_intron;    /* MaxIntronSize = 500000  (-max_intron)           */
    int32_t min_intron;    /* MinIntronSize = 5       (-min_intron)           */
    int32_t max_mismatch;  /* MaxMismatch   = 0       (-mis)                  */
    int32_t multi_hit;     /* bMultiHit               (-m)                    */
    int32_t all_sj;        /* bFindAllJunction        (-all_sj)               */
    int32_t paired;        /* bPairEnd: reads 2i,2i+1 are mates; mate 2 already
                              reverse-complemented as GetData.cpp:157-162 does */
} dg_params;

/* ReadItem_t after ReadMapping's per-read work (structure.h:149-164) */
typedef struct { int32_t score, sub_score, mis_num, mapq, n_rep, best, rep_off, sj_off, n_sj; } dg_read_out;
/* AlignmentReport_t + Coordinate_t (structure.h:117-141); CIGAR as BAM-style ops len<<4|op,
 * op: M=0 I=1 D=2 N=3 S=4, already merged as GenerateCIGAR does (AlignmentCandidates.cpp:37-61) */
typedef struct { int32_t aln_score, sj_type, flag, paired_idx, chr, bdir; int64_t pos; uint32_t cigar_off, n_cigar; } dg_report_out;
/* one UpdateLocalSJMap key (Mapping.cpp:545-556): forward-strand g1,g2 and the SJ type */
typedef struct { int64_t g1, g2; int32_t type, read_idx; } dg_sj_out;

void        dg_params_default(dg_params *);
/* HIP devices this process can use (0: none, or the runtime failed): a multi-GPU host creates one root context per device and shards its
 * batches over them -- the reference's `-t` threads over one shared index (Mapping.cpp:792-793) */
int         dg_device_count(void);
/* uploads the index to device `device` (0-based HIP ordinal) and builds the device-side layout */
dg_ctx     *dg_init(const dg_index_view *, const dg_params *, int device, int *status);
/* ---- start-up from the index FILES (replaces RestoreReferenceInfo / bwa_idx_load, bwt_index.cpp:147-159,229-253, and main.cpp:231-235) ----
 * The reference reads PREFIX.bwt / .sa / .pac into host arrays and expands RefSequence before the first read is mapped.  Here the
 * three files go from the page cache through page-locked staging chunks straight to HBM (several reader threads, each with its own
 * copy stream; the Occ blocks are re-laid in place chunk by chunk behind their copy), so the host never holds the index.  The host
 * program parses the small text file PREFIX.ann itself (it needs the names for the SAM header) and passes the chromosome table.
 * The look-up aids the kernels build on top of the reference's index -- the full suffix array and the K-mer prefix table, which
 * change no result -- are allocated while the files load and
 *   flags = 0                   built before the call returns (what dg_init does)
 *   flags & DG_INIT_ASYNC_AIDS  allocated and built by a library thread (a quarter of the GPU's wave slots, lowest stream priority) while the
 *                               caller already maps batches: a context picks up each aid at its next batch.
 * dg_index_wait blocks until the aids are complete (DG_OK) or failed (the mapping still works without them; text in dg_last_error).
 * dg_init_report: one line of text with the start-up split in seconds (allocation, file -> HBM, each build kernel).              */
typedef struct {
    const char *bwt_path, *sa_path, *pac_path;
    int64_t l_pac; int32_t n_chr; const int64_t *chr_off; const int64_t *chr_len;
    uint64_t expected_reads;   /* size of the job if the host knows it (file sizes), 0 = unknown: below 400 M reads the aids are LEAN -- every 4th
                                  SA row, K <= 14: 17 GB instead of 118 GB for a human genome, seeding twice as long -- because the full ones
                                  cost 0.7 s of build kernels and, on memory that was in use a moment ago, seconds of allocation */
} dg_index_files;
#define DG_INIT_ASYNC_AIDS 1
dg_ctx     *dg_init_files(const dg_index_files *, const dg_params *, int device, int flags, int *status);
int         dg_index_wait(dg_ctx *);
const char *dg_init_report(const dg_ctx *);
void        dg_destroy(dg_ctx *);
/* a second context on the same device sharing the parent's index (no copy): its own stream and batch buffers (the re-seeding kernels of a
 * device's contexts run on a few shared streams: DG_S2_SHARED, INTEGRATION.md 5 -- keep contexts + 3 + the host's streams <= GPU_MAX_HW_QUEUES), so two
 * batches can be in flight at once, one host thread per context -- what the reference gets from running
 * ReadMapping in `-t` threads over one shared index (Mapping.cpp:760-790).  Destroy clones before the parent. */
dg_ctx     *dg_clone(dg_ctx *parent, int *status);
const char *dg_last_error(const dg_ctx *);   /* NULL ctx: error of the last failed dg_init */
int         dg_set_params(dg_ctx *, const dg_params *);

/* ---- the path: host buffers in, host records out (replaces Mapping.cpp:598-639) ----
 * read i = seq[seq_off[i] .. +rlen[i]) ASCII as read from the file.
 * caps/used index: 0 = reports, 1 = cigar ops, 2 = sj tuples.
 * Record layout: reports of read i are reports[rep_off .. rep_off + n_rep) and rep_off increases with i; a report's CIGAR
 * ops are cigar_ops[cigar_off .. cigar_off + n_cigar) -- the layout is the same for the same input (deterministic), but
 * cigar_off does NOT increase with the report index (reports finished by the fused kernel come first, then the others);
 * sj tuples are grouped by read in read order.
 * The whole batch is enqueued on the context's stream without asking the device for a size in between; the call waits twice:
 * for the sizes, then for the records.  If a data-dependent buffer was too small the library grows it and runs the batch
 * again before returning (first batches of a context, or a sudden change in the data).                                   */
int dg_map_batch(dg_ctx *, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq,
                 dg_read_out *, dg_report_out *, uint32_t *cigar_ops, dg_sj_out *,
                 const size_t caps[3], size_t used[3]);

/* Page-locked host memory for the arrays handed to dg_map_batch*: with it the copies in and out are DMA transfers that run
 * beside other contexts' kernels; with ordinary memory the calls still work, the runtime stages the copies. */
void *dg_host_alloc(size_t bytes);
void  dg_host_free(void *);

/* The same path for reads that hold nothing but A, C, G, T and N (upper case): 2 bit/base instead of a byte, which matters
 * once the host link is the limit (202 MB -> 56 MB per million 2x101 pairs).
 *   words   read i = words[i * words_per_read ..): 16 bases per word, FIRST base in the TOP two bits (A C G T = 0 1 2 3),
 *           words_per_read = ceil(longest read / 16); an N is stored as any base and listed in nlist
 *   nlist   the N positions as flat base indices  i * 16 * words_per_read + position  (n_n of them, any order)
 *   rlen    per-read lengths, or NULL when every read has length rlen_all
 * A read with any other character (lower case, IUPAC codes, '-') must go through dg_map_batch: the reference compares raw
 * characters in places (tools.cpp:40-47), so such a read maps differently from its upper-case form.                          */
int dg_map_batch_packed(dg_ctx *, int n_reads, int rlen_all, const uint16_t *rlen, int words_per_read, const uint32_t *words,
                        const uint32_t *nlist, size_t n_n, dg_read_out *, dg_report_out *, uint32_t *cigar_ops, dg_sj_out *,
                        const size_t caps[3], size_t used[3]);

/* ---- compact records: the same information in 12 + 16 bytes instead of 36 + 40, and no CIGAR for a plain full-length match ----
 * The host link carries ~57 GB/s per direction on an MI355X box (full duplex for the copy engines: profiles/probes/duplex_probe.hip),
 * so at several hundred million reads per second the bytes of the records load it: 85 -> 30 bytes per read (full records: 545 M
 * reads/s, compact: 860).  Nothing is lost; what the layout already says is not sent:
 *   rep_off    reports lie in read order: a read's reports start at the sum of n_rep of the reads before it
 *   sj_off     likewise the junction tuples: the sum of n_sj of the reads before it
 *   cigar_off  the stored CIGAR ops lie in TWO regions, each in report order: first those of the reports whose `pad` bit 0 is clear (finished
 *              by the fused kernel), behind them those of the reports with the bit set (the general report path); a report's ops start at
 *              the sum of the stored op counts of the reports before it IN ITS REGION (+ the size of the whole first region for the second)
 *   n_cigar    DG_CIGAR_FULL_MATCH (255): the CIGAR is the single op "<length of the report's read>M" and is not stored
 *              (19 of 20 reports of a DNA run); otherwise the number of stored ops (at most 254)
 * Lossless while the fields fit (scores and mismatches < 65536, at most 65535 reports per read / chromosomes, at most 254 CIGAR
 * ops per report, |POS| < 2^31); when one does not, dg_batch_download_compact returns DG_ERR_RANGE and the caller takes the
 * full records with dg_batch_download.  dart_amd/host.py::expand_compact is the reference expansion.                        */
#define DG_CIGAR_FULL_MATCH 255
typedef struct { uint16_t score, sub_score, mis_num; uint8_t mapq, n_sj; uint16_t n_rep, best; } dg_read_c;                             /* 12 bytes */
typedef struct { int32_t pos; uint16_t aln_score, flag; int16_t paired_idx; uint16_t chr /* 0xFFFF = none */;
                 uint8_t n_cigar; int8_t sj_type; uint8_t bdir, pad /* bit 0: stored ops in the second region */; } dg_report_c;            /* 16 bytes */
/* caps[1] counts stored ops (never more than the full records' op count, `used[1]` of dg_batch_run); *n_ops (may be NULL) = how many were written.
 * The compact records are written by the same kernels as the full ones during dg_batch_run / dg_map_batch_compact (not by dg_map_batch /
 * dg_map_batch_packed, whose callers take the full records: DG_ERR_ARG here after those).                                                    */
int dg_batch_download_compact(dg_ctx *, dg_read_c *, dg_report_c *, uint32_t *cigar_ops, dg_sj_out *, const size_t caps[3], size_t *n_ops);
/* upload (ASCII when words == NULL, else packed as dg_map_batch_packed) + run + compact download in one call; used[1] = stored ops */
int dg_map_batch_compact(dg_ctx *, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq,
                         int rlen_all, int words_per_read, const uint32_t *words, const uint32_t *nlist, size_t n_n,
                         dg_read_c *, dg_report_c *, uint32_t *cigar_ops, dg_sj_out *, const size_t caps[3], size_t used[3]);

/* ---- the same path split so a caller can keep the batch resident in HBM (bench.py) ----
 * dg_batch_upload copies reads to the device; dg_batch_run runs the whole path on the device
 * leaving the result records in HBM (sizes in `used`); dg_batch_download copies them out.     */
int dg_batch_upload(dg_ctx *, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq);
int dg_batch_upload_packed(dg_ctx *, int n_reads, int rlen_all, const uint16_t *rlen, int words_per_read, const uint32_t *words,
                           const uint32_t *nlist, size_t n_n);
int dg_batch_run(dg_ctx *, size_t used[3]);
int dg_batch_download(dg_ctx *, dg_read_out *, dg_report_out *, uint32_t *cigar_ops, dg_sj_out *, const size_t caps[3]);

/* raw device pointers of the last run's records (valid until the next upload/run on this ctx):
 * ptrs[0] dg_read_out[n_reads], [1] dg_report_out[used0], [2] cigar u32[used1], [3] dg_sj_out[used2].
 * Lets a caller hand the records to RCCL (torch.distributed) without a host round trip.       */
int dg_batch_device_ptrs(dg_ctx *, void *ptrs[4]);
/* the compact records of the last dg_batch_download_compact / dg_map_batch_compact in HBM: [0] dg_read_c[n_reads], [1] dg_report_c[used0] */
int dg_batch_device_ptrs_compact(dg_ctx *, void *ptrs[2]);
/* ALL compact records of the last dg_map_batch_compact / dg_batch_download_compact in HBM, with their element counts -- what a multi-GPU
 * host hands to RCCL for the SAM-order gather to the rank that writes (Mapping.cpp:644-664 is ONE ordered writer; bench.py --gather full):
 *   ptrs[0] dg_read_c[counts[0]]   ptrs[1] dg_report_c[counts[1]]   ptrs[2] stored CIGAR ops u32[counts[2]], in the two regions described above
 *   ptrs[3] dg_sj_out[counts[3]], grouped by read in read order (read_idx is the read's index inside the batch)                    */
int dg_batch_device_records_compact(dg_ctx *, void *ptrs[4], size_t counts[4]);

/* per-kernel device time of the last dg_batch_run, measured with HIP events on the library's
 * stream: names[i] -> ms[i]; returns the number of entries written (<= cap)                   */
int dg_last_timings(dg_ctx *, const char **names, float *ms, int cap);
/* work counters of the last run: [0] Occ-pair steps [1] Occ blocks touched [2] LF steps
 * [3] SA lookups [4] seeds [5] candidates [6] NW calls [7] NW cells [8] reseed calls
 * [9] reseed window bases -- [0..3] are counted as the REFERENCE's algorithm would execute them
 * (the basis of the algorithmic-byte figure); [10] Occ-pair steps and [11] Occ blocks this
 * implementation really executed, [12] k-mer prefix-table look-ups, [13] LF steps really executed, [14..17] seeding statistics,
 * [18] 512-position trips of k_reseed's waves, [19] the time its waves were resident, summed, in 10 ns ticks,
 * [20..24] wave-trips of the seeding kernel per queue (begin, Occ step, text comparison, locate, refill), [25..29] the slots they served,
 * [30] its phases, summed over the workgroups (a phase can serve four wave-trips),
 * [31..35] the time the waves of k_seed_qf, k_seed_heavy, k_chain_heavy, k_pair, k_report were resident, summed over the waves, in 10 ns ticks,
 * [36] units (pairs / single reads) that took the general report path, [37] units chained by a wave each (> 16 seeds),
 * [38] how many times the batch was enqueued (> 1: a capacity estimate was too small and the batch ran again),
 * [39] / [40] how often this context has run a batch again since it was created: capacity grown / a scan that did not complete   */
int dg_last_counters(dg_ctx *, uint64_t *out, int cap);

/* diagnostic (tests/probes/kpair_wait.py): the per-tile trace of one of the last run's single-pass scans -- 4 u64 per tile: epoch << 32 | HW_ID,
 * wall clock at the ticket, at the publication of the tile's own totals, XCC_ID << 56 | wall clock at the publication of its prefix.
 * which: 0 = seed offsets, 1 = the fused pair kernel, 2 = the general path's emit kernel.  DG_ERR_CAPACITY: *n_tiles holds the need. */
int dg_debug_scan_trace(dg_ctx *, int which, uint64_t *out, size_t cap_tiles, size_t *n_tiles, double *ticks_per_ms);

/* ---- the index builder's sorter (replaces the suffix sorting inside BWT_Index/bwtindex.c:77-148) ----
 * Stable radix sort of n < 2^32 (key, value) pairs in DEVICE memory, ascending by the low key_bits bits of the key; keys/vals hold the
 * input and the result, *_tmp are scratch of the same size.  Runs on the device's NULL stream and returns when done.
 * dart_amd/index_build.py drives it (prefix doubling: one sort of (rank pair, suffix) per round).                              */
int dg_sort_pairs(int device, uint64_t *keys, int64_t *vals, uint64_t *keys_tmp, int64_t *vals_tmp, size_t n, int key_bits);

/* ---- stage probes (parity tests of single kernels; mirror oracle/dart_oracle.h) ----
 * seeds of every read after the (gPos,rPos) sort (IdentifySeedPairs, AlignmentCandidates.cpp:181-215):
 * read i owns [seed_off[i], seed_off[i+1]) of rpos/slen/gpos; seed_off has n_reads+1 entries. */
int dg_probe_seeds(dg_ctx *, int n_reads, const uint32_t *seq_off, const uint16_t *rlen, const char *seq,
                   uint32_t *seed_off, int32_t *rpos, int32_t *slen, int64_t *gpos, size_t cap, size_t *used);
/* nw_alignment (nw_alignment.cpp:18-82) of n pairs: a[i]=s[a_off[i]..a_off[i+1]) etc.; outputs are
 * NUL-free gapped strings of equal length out_len[i] at out_a/out_b + out_off[i] (cap each = sum of lengths) */
int dg_probe_nw(dg_ctx *, int n, const uint32_t *a_off, const uint32_t *b_off, const char *a, const char *b,
                uint32_t *out_off, uint32_t *out_len, char *out_a, char *out_b, size_t cap);
/* the same through each form of nw_alignment the kernels run: mode 0 = serial strips (what dg_probe_nw runs), 1 = register
 * strips (pairs up to 24 x 24), 2 = the wave-wide service (8-lane groups up to 64 columns, the whole wave beyond) with the
 * owner's traceback, 3 = the whole-wave form for every pair.  Modes 2 and 3 take the b side from a 2-bit text, as the
 * kernels do (RefSequence, bwt_index.cpp:193-212): b must be ACGT, else DG_ERR_ARG.                                  */
int dg_probe_nw_mode(dg_ctx *, int mode, int n, const uint32_t *a_off, const uint32_t *b_off, const char *a, const char *b,
                     uint32_t *out_off, uint32_t *out_len, char *out_a, char *out_b, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
