/* include/dartindex.h -- C ABI of libdartindex.so: the device side of the offline indexer ("next").
 *
 * The reference's indexer is a program of its own (`bwt_index`, BWT_Index/bwtindex.c:77-148: pac -> BWT through bwt_gen.c /
 * QSufSort.c, Occ interleaving bwtindex.c:53-75, sampled SA bwt.c:101-123,185-196); its files are a pure function of the text, so
 * what has to be reproduced is the bytes, not the algorithm.  This library is the MI355X-side of dart_amd/index_build.py: suffix
 * sorting of the 2-bit text (forward + reverse complement, '$' behind it) by bucketed prefix doubling in HBM, then the BWT with its
 * Occ counters in the .bwt block layout.  It is separate from libdartgpu.so the way `bwt_index` is separate from `dart`; the radix
 * sorter both use is dg_sort_pairs (include/dartgpu.h).
 *
 * Conventions: every pointer is DEVICE memory of `device`; the calls run on the device's NULL stream and return when the work is
 * done; 0 = ok, -1 = bad argument, -2 = HIP error (di_last_error() has the text).  n = number of text symbols (2 x l_pac),
 * N = n + 1 suffixes (suffix n is the lone '$', the smallest).
 *
 * Text layout ("T"): 2 bits per symbol, 32 symbols per uint64_t, the FIRST symbol in the most significant bits, zero (= 'A') past
 * the end, di_text_words(n) words (two words of padding: every 32-symbol window that starts at a position <= n + 31 may be read).
 */
#ifndef DARTINDEX_H
#define DARTINDEX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DI_TILE 4096u                                   /* positions (or pairs) per workgroup in every tiled kernel below */

const char *di_last_error(void);
size_t di_text_words(uint64_t n);

/* fwd: l_pac codes 0..3, one per byte (bntseq.c:144,173-174: an ambiguous base already replaced by its random code).
 * T <- forward strand followed by its reverse complement (bntseq.c:184-190, called with for_only = 0 at bwtindex.c:93: the text
 * bwt_bwtgen2 is given at bwtindex.c:106). */
int di_pack_text(int device, const uint8_t *fwd, uint64_t l_pac, uint64_t *T);

/* How many suffixes start with each of the 16 symbol pairs, per tile of DI_TILE text positions: table[pair * tiles + tile],
 * tiles = ceil((n + 1) / DI_TILE), pair = 4 * first + second.  Only suffixes with two real symbols count (i <= n - 2): suffix n ('$')
 * and suffix n - 1 (one symbol, then '$') are singletons whose rows the caller knows. */
int di_bucket_hist(int device, const uint64_t *T, uint64_t n, uint32_t *table);

/* The members of pair bucket `pair`, in text order, with their round-0 keys: for tile t the members go to [tile_base[t], ...)
 * (tile_base = exclusive prefix sum of the bucket's row of di_bucket_hist's table).  vals[j] = the suffix, keys[j] = the 29 symbols
 * behind the pair (58 bits) << 5 | how many of those 29 exist (0..29): equal padded codes order shorter-first, as '$' < 'A' demands,
 * and no suffix that meets the '$' inside the key ties with any other. */
int di_bucket_keys(int device, const uint64_t *T, uint64_t n, int pair, const uint32_t *tile_base, uint64_t *keys, int64_t *vals);

/* A doubling round's keys for the m still-tied rows pos[0..m) (ascending, relative to `lo`) of one bucket: s = sa[lo + pos[j]],
 * keys[j] = (rank[s] - lo) << r2_bits | (s + k < N ? rank[s + k] + 1 : 0), vals[j] = s. */
int di_doubling_keys(int device, const int64_t *sa, const int64_t *rank, uint64_t lo, const uint32_t *pos, uint32_t m, uint64_t k, uint64_t N,
                     int r2_bits, uint64_t *keys, int64_t *vals);

/* After the sort of (keys, vals): the j-th pair belongs in row lo + P(j), P(j) = pos ? pos[j] : j.  Writes sa[lo + P(j)] = vals[j]
 * and rank[vals[j]] = lo + P(first pair with the same key); the rows whose key is shared with a neighbour go, ascending, to
 * new_pos[0..*n_tied) (room for m entries; must not be pos: tiles finish in any order).  scratch: 2 * ceil(m / DI_TILE) + 4 u32.
 * n_tied is a HOST pointer. */
int di_regroup(int device, const uint64_t *keys, const int64_t *vals, const uint32_t *pos, uint32_t m, uint64_t lo,
               int64_t *rank, int64_t *sa, uint32_t *new_pos, uint32_t *scratch, uint32_t *n_tied);

/* The .bwt body (bwtindex.c:53-75): for each block of 128 BWT symbols, blocks[16 * b + 8 .. + 15] = the symbols, 16 per word, first
 * symbol in the top bits; counts[b] = how many A / C / G / T the block holds, one byte each (A lowest).  BWT symbol o (o < n) =
 * T[sa[o + (o >= primary)] - 1]: the row of suffix 0 (`primary`) is left out (bwt.c's bwt_B0 convention).  The caller turns
 * counts into the running totals of blocks[16 * b + 0 .. + 7]. */
int di_bwt_blocks(int device, const int64_t *sa, const uint64_t *T, uint64_t n, uint64_t primary, uint32_t *blocks, uint32_t *counts);

/* The whole build (bwtindex.c:77-148 from the packed forward strand on): writes PREFIX.pac (bntseq.c:192-201), PREFIX.bwt
 * (bwtindex.c:53-75) and PREFIX.sa (bwt.c:101-123,185-196) for the l_pac codes 0..3 at `fwd` -- HOST memory, ambiguous bases already
 * replaced (bntseq.c:144,173-174); PREFIX.ann / .amb are the caller's (they are text about names and holes).  log, if given, gets a
 * line per phase.  -3 = a file could not be written.  HBM: about 17 x n bytes at the peak (n = 2 x l_pac).
 * `dart index ref.fa prefix` (main.cpp:125-127) and dart_amd/index_build.py both end here. */
typedef void (*di_log_fn)(const char *line, void *arg);
int di_build_files(int device, const uint8_t *fwd, uint64_t l_pac, const char *prefix, di_log_fn log, void *log_arg, uint64_t *primary_out);

#ifdef __cplusplus
}
#endif
#endif
