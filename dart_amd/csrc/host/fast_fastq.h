// dart_amd/csrc/host/fast_fastq.h -- the host side of `dart` for plain (uncompressed) FASTQ input, built for throughput:
// the reference reads, maps and prints under two global locks (GetData.cpp:134-179 under LibraryLock, Mapping.cpp:644-664 under
// OutputLock); with the mapping at hundreds of millions of reads per second those two serial sections are everything.
//
//   index pass     the input files are memory-mapped; T threads count the newlines of their share (pass A), then walk the lines
//                  of their share with the exact line number known (pass B): every fourth line starts a record.  No heuristics
//                  about '@': a record is four lines, as GetNextEntry (GetData.cpp:77-132) reads them.
//   assembly       per batch, T threads copy the bases into a page-locked arena (mate 2 reverse-complemented as GetData.cpp:157-166
//                  does) -- headers and qualities are never copied, the formatter reads them from the mapping
//   mapping        dg_map_batch on the batch's arena, one thread per context
//   format + write T threads format contiguous read ranges into their own buffers, the ranges' sizes give their file offsets,
//                  and the same threads pwrite() them: SAM bytes in input order without a serial writer
// The record semantics (header trimming, "an entry without bases ends the stream", qualities cut to the read length, the
// 4000-read / 1 000 000-base chunking that decides which reads of an odd-sized tail are mapped unpaired) are those of the
// reference's reader and of the streaming path in dart_main.cpp; tests/test_gpu_cli.py runs both paths against the oracle's
// command line.
#pragma once
#include <atomic>
#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include "libdeflate_dl.h"
#include <unistd.h>

// (libdeflate_dl.h: the system's libdeflate when it is there; it has no streaming interface, so a .gz file's output must fit a buffer: open_gz below)
struct MappedFile {
    const char *p = nullptr; size_t n = 0; int fd = -1;
    size_t cap = 0;                                   // > 0: p is an anonymous mapping of cap bytes holding an inflated file
    // A .gz file, inflated whole into memory (every member, as gzread concatenates them).  false = not possible here (no libdeflate, larger than
    // max_out, not a clean gzip file): the caller takes the streaming reader, which has gzread's behaviour for whatever this is.
    // What this process may still take: MemAvailable, and what the cgroup (v2, then v1) leaves -- a container's limit is not in /proc/meminfo.  0 = unknown.
    static size_t memory_left() {
        size_t avail = 0;
        if (FILE *f = fopen("/proc/meminfo", "r")) {
            char line[256];
            while (fgets(line, sizeof line, f)) { unsigned long long kb; if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (size_t)kb << 10; break; } }
            fclose(f);
        }
        if (!avail) { const long pg = sysconf(_SC_AVPHYS_PAGES), sz = sysconf(_SC_PAGESIZE); if (pg > 0 && sz > 0) avail = (size_t)pg * (size_t)sz; }
        auto read_u64 = [](const char *path, unsigned long long &v) { FILE *f = fopen(path, "r"); if (!f) return false; const bool ok = fscanf(f, "%llu", &v) == 1; fclose(f); return ok; };
        unsigned long long lim = 0, cur = 0;
        if ((read_u64("/sys/fs/cgroup/memory.max", lim) && read_u64("/sys/fs/cgroup/memory.current", cur)) ||
            (read_u64("/sys/fs/cgroup/memory/memory.limit_in_bytes", lim) && read_u64("/sys/fs/cgroup/memory/memory.usage_in_bytes", cur))) {
            const size_t room = lim > cur ? (size_t)(lim - cur) : 0;
            if (lim < (1ull << 60) && (!avail || room < avail)) avail = room;
        }
        return avail;
    }
    // A .gz file, inflated whole into memory (every member, as gzread concatenates them).  false = not possible here (no libdeflate, larger than
    // max_out or than a third of the memory this process may still take -- both mates are inflated at once and the batch buffers come on top; the
    // streaming reader runs such a job in constant memory --, not a clean gzip file): the caller takes the streaming reader, which has gzread's
    // behaviour for whatever this is.  The mapping grows in place (mremap) and the inflate goes on with the member that did not fit: the last
    // member's ISIZE is only a first guess (multi-member and bgzip files), and starting over doubled the work each time.
    bool open_gz(const char *fn, size_t max_out) {
        const LibDeflate &ld = lib_deflate();
        if (!ld.ok() || getenv("DART_GZ_STREAM")) return false;          // DART_GZ_STREAM=1: always the streaming reader
        MappedFile z;
        if (!z.open(fn) || z.n < 18 || (unsigned char)z.p[0] != 0x1f || (unsigned char)z.p[1] != 0x8b) return false;
        { const size_t left = memory_left(); if (left && left / 3 < max_out) max_out = left / 3; }
        uint32_t isize; memcpy(&isize, z.p + z.n - 4, 4);                       // the last member's size mod 2^32: the first guess
        size_t want = std::max<size_t>((size_t)isize + 64, z.n * 4);
        if (want > max_out) return false;
        void *d = ld.alloc();
        if (!d) return false;
        void *m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) { ld.release(d); return false; }
        size_t in_at = 0, out_at = 0; int rc = 0;
        while (in_at < z.n) {
            if (z.n - in_at < 18 || (unsigned char)z.p[in_at] != 0x1f || (unsigned char)z.p[in_at + 1] != 0x8b) { rc = 1; break; }    // bytes behind the last member: gzread's business
            size_t used = 0, made = 0;
            rc = ld.gunzip(d, z.p + in_at, z.n - in_at, (char *)m + out_at, want - out_at, &used, &made);
            if (rc == 3) {                                                      // LIBDEFLATE_INSUFFICIENT_SPACE: this member again, into a larger mapping (what the earlier members gave stays)
                const size_t bigger = want * 2;
                if (bigger > max_out) break;
                void *m2 = mremap(m, want, bigger, MREMAP_MAYMOVE);
                if (m2 == MAP_FAILED) break;
                m = m2; want = bigger; rc = 0;
                continue;
            }
            if (rc) break;
            in_at += used; out_at += made;
        }
        ld.release(d);
        if (rc != 0 || in_at < z.n) { munmap(m, want); return false; }
        p = (const char *)m; n = out_at; cap = want;
        return true;
    }
    bool open(const char *fn) {
        fd = ::open(fn, O_RDONLY);
        if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        madvise(m, n, MADV_SEQUENTIAL);
        p = (const char *)m;
        return true;
    }
    void close_now() { if (p && (cap || n)) munmap((void *)p, cap ? cap : n); if (fd >= 0) ::close(fd); p = nullptr; n = cap = 0; fd = -1; }
    ~MappedFile() { close_now(); }
};

// one FASTQ record: byte offset of its header line and the lengths of its four lines INCLUDING the newline (0 = the line does
// not exist: end of file)
struct FqRec { uint64_t off; uint32_t l0, l1, l2, l3; };

template <class F>
static void parallel_for(int nt, F f)
{
    if (nt <= 1) { f(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back([&f, t]() { f(t); });
    f(0);
    for (auto &x : th) x.join();
}

// the records of a mapped FASTQ file, in file order; *has_empty: some record has no bases (such an entry ends a chunk of the
// reference's reader, GetData.cpp:141,154: the caller then replays the chunking)
static void index_fastq(const MappedFile &mf, int nt, std::vector<FqRec> &recs, bool *has_empty, size_t min_share = (size_t)1 << 20)
{
    recs.clear(); *has_empty = false;
    const char *p = mf.p; const size_t n = mf.n;
    if (n == 0) return;
    nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)nt, n / min_share + 1));
    std::vector<size_t> cut(nt + 1), nl(nt + 1, 0);
    for (int k = 0; k <= nt; k++) cut[k] = n / nt * k;
    cut[nt] = n;
    parallel_for(nt, [&](int k) {                       // pass A: newlines per share
        size_t c = 0;
        for (const char *q = p + cut[k], *e = p + cut[k + 1]; q < e;) { const char *f = (const char *)memchr(q, '\n', e - q); if (!f) break; c++; q = f + 1; }
        nl[k + 1] = c;
    });
    for (int k = 0; k < nt; k++) nl[k + 1] += nl[k];    // newlines before each share
    std::vector<std::vector<FqRec>> part(nt);
    auto line_len = [&](size_t at) -> uint32_t {        // length of the line that starts at `at`, newline included; 0 past the end
        if (at >= n) return 0;
        const char *f = (const char *)memchr(p + at, '\n', n - at);
        return (uint32_t)(f ? (size_t)(f - (p + at)) + 1 : n - at);
    };
    parallel_for(nt, [&](int k) {                       // pass B: the records whose header line starts in this share
        const size_t c0 = cut[k], c1 = cut[k + 1];
        // a line starts at 0 and behind every newline; its number = the newlines before it.  First start in [c0, c1):
        size_t pos, line;
        if (c0 == 0) { pos = 0; line = 0; }
        else if (p[c0 - 1] == '\n') { pos = c0; line = nl[k]; }
        else { const char *f = (const char *)memchr(p + c0, '\n', n - c0); if (!f) return; pos = (size_t)(f - p) + 1; line = nl[k] + 1; }
        part[k].reserve((c1 - c0) / 200 + 16);
        while (pos < c1 && pos < n) {
            if (line & 3) { const uint32_t l = line_len(pos); pos += l; line++; if (l == 0) break; continue; }
            FqRec r; r.off = pos;
            r.l0 = line_len(pos); r.l1 = line_len(pos + r.l0); r.l2 = line_len(pos + r.l0 + r.l1); r.l3 = line_len(pos + r.l0 + r.l1 + r.l2);
            part[k].push_back(r);
            pos += (size_t)r.l0 + r.l1 + r.l2 + r.l3; line += 4;
            if (r.l3 == 0) break;                       // the file ended inside this record
        }
    });
    size_t total = 0;
    for (auto &v : part) total += v.size();
    recs.reserve(total);
    for (auto &v : part)
        for (const FqRec &r : v) {
            if ((int)r.l1 - 1 <= 0) *has_empty = true;           // rlen = line length - 1 (GetData.cpp:103)
            recs.push_back(r);
        }
}

// One read as the formatter sees it.  The STORED read (what ReadItem_t holds, GetData.cpp:140-166) is s[0..sl) and q[0..ql) as they
// are in the file, or -- rc: mate 2 of a pair -- the reverse complement (non-ACGT -> N) of s and the reversed q.
struct RView { const char *h; const char *s; const char *q; uint32_t hl, sl, ql; bool rc; };

static inline char comp_base_f(char c)   // tools.cpp:3-17
{
    switch (c) { case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A'; default: return 'N'; }
}

#define PUT_LIT(out, lit) (out).put(lit, sizeof(lit) - 1)

// complement tables: COMP1 = comp_base_f, COMP2 = comp_base_f applied twice (canonical upper-case base, or N)
struct CompTables { unsigned char c1[256], c2[256]; CompTables() { for (int i = 0; i < 256; i++) { c1[i] = (unsigned char)comp_base_f((char)i); c2[i] = (unsigned char)comp_base_f(comp_base_f((char)i)); } } };
static const CompTables g_comp;

// ---- byte-string kernels of the formatter: reversed copy and reverse-complement copy, 32 bytes at a time where the CPU has AVX2 ----
#include <immintrin.h>
__attribute__((target("avx2"))) static void rev_copy_avx2(char *d, const char *s, size_t l)
{
    const __m256i idx = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    size_t i = 0;
    for (; i + 32 <= l; i += 32) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(s + l - 32 - i));
        v = _mm256_permute2x128_si256(_mm256_shuffle_epi8(v, idx), _mm256_shuffle_epi8(v, idx), 1);
        _mm256_storeu_si256((__m256i *)(d + i), v);
    }
    for (; i < l; i++) d[i] = s[l - 1 - i];
}
// comp_base_f on 32 bytes: A/a -> T, C/c -> G, G/g -> C, T/t -> A, anything else -> N
__attribute__((target("avx2"))) static inline __m256i comp32_avx2(__m256i v)
{
    const __m256i lo = _mm256_or_si256(v, _mm256_set1_epi8(0x20));
    __m256i r = _mm256_set1_epi8('N');
    r = _mm256_blendv_epi8(r, _mm256_set1_epi8('T'), _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('a')));
    r = _mm256_blendv_epi8(r, _mm256_set1_epi8('G'), _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('c')));
    r = _mm256_blendv_epi8(r, _mm256_set1_epi8('C'), _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('g')));
    r = _mm256_blendv_epi8(r, _mm256_set1_epi8('A'), _mm256_cmpeq_epi8(lo, _mm256_set1_epi8('t')));
    return r;
}
__attribute__((target("avx2"))) static void revcomp_copy_avx2(char *d, const char *s, size_t l)
{
    const __m256i idx = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    size_t i = 0;
    for (; i + 32 <= l; i += 32) {
        __m256i v = comp32_avx2(_mm256_loadu_si256((const __m256i *)(s + l - 32 - i)));
        v = _mm256_shuffle_epi8(v, idx);
        _mm256_storeu_si256((__m256i *)(d + i), _mm256_permute2x128_si256(v, v, 1));
    }
    for (; i < l; i++) d[i] = (char)g_comp.c1[(unsigned char)s[l - 1 - i]];
}
__attribute__((target("avx2"))) static void comp2_copy_avx2(char *d, const char *s, size_t l)     // comp_base_f twice: the canonical base, or N
{
    size_t i = 0;
    for (; i + 32 <= l; i += 32) _mm256_storeu_si256((__m256i *)(d + i), comp32_avx2(comp32_avx2(_mm256_loadu_si256((const __m256i *)(s + i)))));
    for (; i < l; i++) d[i] = (char)g_comp.c2[(unsigned char)s[i]];
}
static const bool g_avx2 = __builtin_cpu_supports("avx2");
static const char g_digits2[201] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";

// (one per formatter thread, side by side in a vector: each on cache lines of its own -- `n` is written with every field, and neighbours that
//  shared a line made thirteen threads format slower than one: 890 ns per read and thread against 50, profiles/r03/g_formatter_false_sharing.txt)
struct alignas(128) TextBuf {
    char *b = nullptr; size_t n = 0, cap = 0;
    ~TextBuf() { free(b); }
    inline void need(size_t more) { if (n + more > cap) { cap = std::max(cap * 2, n + more + (1 << 16)); b = (char *)realloc(b, cap); } }
    inline void put(const char *s, size_t l) { memcpy(b + n, s, l); n += l; }
    inline void ch(char c) { b[n++] = c; }
    inline void put_rev(const char *s, size_t l) {                          // s reversed
        char *d = b + n;
        if (g_avx2) rev_copy_avx2(d, s, l); else for (size_t i = 0; i < l; i++) d[i] = s[l - 1 - i];
        n += l;
    }
    inline void put_comp2(const char *s, size_t l) {                        // comp(comp(s))
        char *d = b + n;
        if (g_avx2) comp2_copy_avx2(d, s, l); else for (size_t i = 0; i < l; i++) d[i] = (char)g_comp.c2[(unsigned char)s[i]];
        n += l;
    }
    inline void put_revcomp(const char *s, size_t l) {                      // reverse complement (non-ACGT -> N)
        char *d = b + n;
        if (g_avx2) revcomp_copy_avx2(d, s, l); else for (size_t i = 0; i < l; i++) d[i] = (char)g_comp.c1[(unsigned char)s[l - 1 - i]];
        n += l;
    }
    inline void num(long long v) {                                          // decimal, two digits per step
        char t[24]; int k = 24;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        while (u >= 100) { const unsigned r = (unsigned)(u % 100); u /= 100; k -= 2; t[k] = g_digits2[2 * r]; t[k + 1] = g_digits2[2 * r + 1]; }
        if (u >= 10) { k -= 2; t[k] = g_digits2[2 * u]; t[k + 1] = g_digits2[2 * u + 1]; } else t[--k] = (char)('0' + u);
        if (v < 0) b[n++] = '-';
        memcpy(b + n, t + k, (size_t)(24 - k)); n += (size_t)(24 - k);
    }
};

struct alignas(128) Counters { long long total = 0, unique = 0, unmapped = 0, paired = 0; };

// OutputPairedAlignments / OutputSingledAlignments (Mapping.cpp:208-369) for reads [lo,hi) of a batch.
// names/name_len: chromosome names; n_pair_mode: reads below this index are mates of pairs (2i, 2i+1).
static void format_views(const RView *V, int lo, int hi, int n_pair_mode, const dg_read_out *ro, const dg_report_out *po, const uint32_t *cig,
                         const std::vector<std::string> &names, bool unique_only, bool multi, bool fastq, TextBuf &out, Counters &ct)
{
    static const char *XS_A[3] = { "", " XS:A:+", " XS:A:-" };
    static const size_t XS_L[3] = { 0, 7, 7 };
    for (int k = lo; k < hi; k++) {
        const bool is_pair = k < n_pair_mode, mate2 = is_pair && (k & 1);
        const RView &e = V[k];
        // the reference prints the stored quality as a C string: it ends at a NUL byte, if there is one
        uint32_t ql = e.ql;
        if (ql && memchr(e.q, 0, ql)) { uint32_t z = 0; while (z < ql && (e.rc ? e.q[ql - 1 - z] : e.q[z]) != 0) z++; ql = z; }
        const dg_read_out &r = ro[k];
        const dg_report_out *rp = po + r.rep_off;
        auto put_stored_seq = [&]() { if (e.rc) out.put_revcomp(e.s, e.sl); else out.put(e.s, e.sl); };
        auto put_stored_qual = [&]() { if (e.rc) out.put_rev(e.q + (e.ql - ql), ql); else out.put(e.q, ql); };
        if (r.score == 0) {
            ct.unmapped++;
            out.need(e.hl + e.sl + ql + 96);
            out.put(e.h, e.hl); out.ch('\t'); out.num(rp[0].flag); PUT_LIT(out, "\t*\t0\t0\t*\t*\t0\t0\t"); put_stored_seq(); out.ch('\t');
            if (fastq) put_stored_qual(); else out.ch('*');
            PUT_LIT(out, "\tAS:i:0\tXS:i:0\n");
            continue;
        }
        if (!(!unique_only || r.mapq > 3)) continue;
        if (r.mapq == 50) ct.unique++;
        const dg_read_out *m = is_pair ? &ro[k ^ 1] : nullptr;
        const dg_report_out *mp = m ? po + m->rep_off : nullptr;
        for (int j = r.best; j < r.n_rep; j++) {
            const dg_report_out &pr = rp[j];
            const bool show = is_pair ? pr.aln_score > 0 : pr.aln_score == r.score;
            if (show) {
                int xs;
                if (pr.sj_type == -1) xs = 0; else if (pr.sj_type == 0 || pr.sj_type == 2) xs = mate2 ? 2 : 1; else xs = mate2 ? 1 : 2;
                const bool use_alt = mate2 ? pr.bdir == 1 : pr.bdir == 0;      // print the reverse complement of the stored read
                const std::string &cn = names[pr.chr];
                int pj;
                out.need(e.hl + cn.size() + e.sl + ql + 16 * (size_t)pr.n_cigar + 192);
                out.put(e.h, e.hl);
                out.ch('\t'); out.num(pr.flag); out.ch('\t'); out.put(cn.data(), cn.size()); out.ch('\t'); out.num((long long)pr.pos); out.ch('\t'); out.num(r.mapq); out.ch('\t');
                for (uint32_t c = 0; c < pr.n_cigar; c++) { const uint32_t op = cig[pr.cigar_off + c]; out.num(op >> 4); out.ch("MIDNS"[op & 15]); }
                if (is_pair && (pj = pr.paired_idx) != -1 && mp[pj].aln_score > 0) {
                    const dg_report_out &a = mate2 ? mp[pj] : pr, &b = mate2 ? pr : mp[pj];
                    const int l1 = mate2 ? (int)V[k ^ 1].sl : (int)e.sl, l2 = mate2 ? (int)e.sl : (int)V[k ^ 1].sl;
                    int dist = (int)(b.pos - a.pos + (a.bdir ? l2 : 0 - l1));
                    if (mate2) dist = 0 - dist; else if (j == r.best) ct.paired += 2;
                    PUT_LIT(out, "\t=\t"); out.num((long long)mp[pj].pos); out.ch('\t'); out.num(dist); out.ch('\t');
                } else PUT_LIT(out, "\t*\t0\t0\t");
                if (!use_alt) put_stored_seq();
                else if (e.rc) out.put_comp2(e.s, e.sl);                          // revcomp of the stored revcomp
                else out.put_revcomp(e.s, e.sl);
                out.ch('\t');
                if (!fastq) out.ch('*');
                else if (!use_alt) put_stored_qual();
                else if (e.rc) out.put(e.q + (e.ql - ql), ql);                  // the stored qualities reversed = the file's (the last ql of them)
                else out.put_rev(e.q, ql);
                PUT_LIT(out, "\tNM:i:"); out.num(r.mis_num); PUT_LIT(out, "\tAS:i:"); out.num(r.score); PUT_LIT(out, "\tXS:i:"); out.num(r.sub_score);
                out.put(XS_A[xs], XS_L[xs]); out.ch('\n');
                if (!is_pair && !multi) break;
            }
            if (is_pair && !multi) break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the pipeline
// ---------------------------------------------------------------------------------------------
static int hdr_beg_f(const char *s, int len) { for (int i = 1; i < len; i++) if (s[i] != '>' && s[i] != '@') return i; return len - 1; }
static int hdr_end_f(const char *s, int len) { for (int i = 1; i < len; i++) if (s[i] == ' ' || s[i] == '/' || s[i] == '\t') return i; return len - 1; }

struct FastSlot {                       // one batch travelling through the stages
    size_t first = 0; int n = 0, odd = 0; size_t seqno = 0;
    std::vector<RView> view; std::vector<uint32_t> soff; std::vector<uint16_t> rl;
    char *seq = nullptr; size_t seq_cap = 0, n_cap = 0;            // page-locked
    dg_read_out *ro = nullptr; dg_report_out *po = nullptr; uint32_t *cig = nullptr; dg_sj_out *sj = nullptr; size_t caps[3] = {0, 0, 0}, used[3] = {0, 0, 0};
    int rc = 0; std::string err;
};

struct FastStats { double t_index = 0, t_asm = 0, t_map = 0, t_fmt = 0, t_write = 0, t_alloc = 0; };

// Batch arenas: ordinary memory by default -- page-locking costs ~1 s per GB, more than a short job saves; DART_PINNED=1 page-locks them
// (then the copies to and from the GPU are plain DMA transfers: worth it for long runs with several contexts in flight)
static bool g_pinned_slots = false;
static void *arena_alloc(size_t bytes) { return g_pinned_slots ? dg_host_alloc(bytes) : malloc(bytes ? bytes : 1); }
static void arena_free(void *p) { if (!p) return; if (g_pinned_slots) dg_host_free(p); else free(p); }

static void slot_reserve(FastSlot &s, size_t n, size_t bases, const size_t need[3])
{
    if (bases + 64 > s.seq_cap) { arena_free(s.seq); s.seq_cap = bases + bases / 8 + 4096; s.seq = (char *)arena_alloc(s.seq_cap); }
    const size_t want[3] = { std::max(need[0], n * 2 + 1024), std::max(need[1], n * 6 + 4096), std::max(need[2], n + 1024) };
    if (!s.ro || s.n_cap < n || s.caps[0] < want[0] || s.caps[1] < want[1] || s.caps[2] < want[2]) {
        arena_free(s.ro); arena_free(s.po); arena_free(s.cig); arena_free(s.sj);
        s.n_cap = n + 16;
        s.ro = (dg_read_out *)arena_alloc(s.n_cap * sizeof(dg_read_out)); s.po = (dg_report_out *)arena_alloc(want[0] * sizeof(dg_report_out));
        s.cig = (uint32_t *)arena_alloc(want[1] * 4); s.sj = (dg_sj_out *)arena_alloc(want[2] * sizeof(dg_sj_out));
        s.caps[0] = want[0]; s.caps[1] = want[1]; s.caps[2] = want[2];
    }
}

// the batch slots; with DART_PINNED=1 they are page-locked in the background while the index loads
struct SlotPool {
    std::vector<FastSlot> slots; std::thread th;
    void start(size_t n_slots, size_t batch_reads, size_t bytes_per_read) {
        g_pinned_slots = getenv("DART_PINNED") && atoi(getenv("DART_PINNED")) != 0;
        slots.resize(n_slots);
        if (g_pinned_slots) th = std::thread([this, batch_reads, bytes_per_read]() { const size_t z[3] = {0, 0, 0}; for (auto &s : slots) slot_reserve(s, batch_reads, batch_reads * bytes_per_read, z); });
    }
    void wait() { if (th.joinable()) th.join(); }
    ~SlotPool() { wait(); for (auto &s : slots) { arena_free(s.seq); arena_free(s.ro); arena_free(s.po); arena_free(s.cig); arena_free(s.sj); } }
};

// The mapped read files of a library and the positions of their records.  The host program starts this for its first library before it
// loads the genome index and calls dg_init, so the two overlap (0.2 s per 16 M reads of the 1.6 s a job of that size takes).
// Would the reference's gz reader (gzGetNextEntry, GetData.cpp:181-210: gzgets into 1024 bytes, strlen) see these records as its plain reader
// (GetNextEntry, :77-132: getline) does?  Yes when every record has its four lines, none of them reaches 1024 bytes or holds a NUL, every header
// line begins with '@' and names something, and no entry is without bases.  Anything else goes through the streaming reader, which restates gzgets.
static bool gz_reader_sees_the_same(const MappedFile &mf, const std::vector<FqRec> &recs, int nt)
{
    if (mf.n && memchr(mf.p, 0, mf.n)) return false;
    std::atomic<bool> same(true);
    parallel_for(std::max(1, nt), [&](int k) {
        const size_t a = recs.size() * (size_t)k / (size_t)std::max(1, nt), b = recs.size() * (size_t)(k + 1) / (size_t)std::max(1, nt);
        for (size_t i = a; i < b && same.load(std::memory_order_relaxed); i++) {
            const FqRec &r = recs[i];
            bool good = r.l0 >= 2 && r.l1 >= 2 && r.l2 >= 1 && r.l3 >= 1 && r.l0 < 1024 && r.l1 < 1024 && r.l2 < 1024 && r.l3 < 1024 && mf.p[r.off] == '@';
            if (good) {                                     // IdentifyHeaderBegPos / EndPos (:55-75): the name must not be empty (:194)
                const char *h = mf.p + r.off; const int len = (int)r.l0;
                int p1 = len - 1, p2 = len - 1;
                for (int q = 0; q < len; q++) if (h[q] != '>' && h[q] != '@') { p1 = q; break; }
                for (int q = 1; q < len; q++) if (h[q] == ' ' || h[q] == '/' || h[q] == '\t') { p2 = q; break; }
                good = p2 - p1 > 0;
            }
            if (!good) same.store(false, std::memory_order_relaxed);
        }
    });
    return same.load();
}

struct FastqIndex {
    MappedFile m1, m2; std::vector<FqRec> r1, r2; bool e1 = false, e2 = false, ok = false, two = false; double t_index = 0;
    std::string f1, f2; std::thread th;
    // gz: the files are .gz and are inflated whole first (MappedFile::open_gz); ok then also says that the plain pipeline reads them as the gz reader would
    void run(const char *a, const char *b, int T, bool gz = false, size_t max_out = 0) {
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        f1 = a; f2 = b ? b : ""; two = b != nullptr;
        if (gz) {
            bool ok2 = true;
            if (b) { std::thread t2([&]() { ok2 = m2.open_gz(b, max_out); }); ok = m1.open_gz(a, max_out); t2.join(); ok = ok && ok2; }
            else ok = m1.open_gz(a, max_out);
        } else ok = m1.open(a) && (!b || m2.open(b));
        if (ok) {
            if (b) { std::thread t2([&]() { index_fastq(m2, std::max(1, T / 2), r2, &e2); }); index_fastq(m1, std::max(1, T - T / 2), r1, &e1); t2.join(); }
            else index_fastq(m1, T, r1, &e1);
        }
        if (ok && gz) ok = !e1 && !e2 && gz_reader_sees_the_same(m1, r1, T) && (!b || gz_reader_sees_the_same(m2, r2, T));
        if (!ok && gz) { m1.close_now(); m2.close_now(); r1.clear(); r2.clear(); r1.shrink_to_fit(); r2.shrink_to_fit(); }
        t_index = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t;
    }
    void start(const char *a, const char *b, int T, bool gz = false, size_t max_out = 0) { f1 = a; f2 = b ? b : ""; th = std::thread([this, a, b, T, gz, max_out]() { run(a, b, T, gz, max_out); }); }
    void wait() { if (th.joinable()) th.join(); }
    ~FastqIndex() { wait(); }
};

// Maps one library of plain FASTQ files.  f2 == nullptr: one file (single-end, or interlaced pairs when pair_end).
// Returns 0, or the failing dg status (message in err).  SAM text goes to fd at *file_off (advanced).
static int run_fast_library(const char *f1, const char *f2, bool pair_end, int threads, size_t batch_reads, const std::vector<dg_ctx *> &ctx, const dg_params &base_params,
                            const std::vector<std::string> &names, bool unique_only, bool multi, bool silent, int fd, uint64_t *file_off,
                            Counters &total, std::map<std::pair<int64_t, int64_t>, int> &sjmap, time_t t0, std::string &err, FastStats &st, SlotPool &pool,
                            FastqIndex *pre = nullptr)
{
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int T = std::max(1, threads);
    // T is the budget of ALL stages (the GPU boxes grant a CPU quota, not cores: more runnable threads than the quota means the whole process is
    // throttled): the formatter takes what the writer (TWR threads), the assembler and the mapping threads leave.  DART_FMT_THREADS / DART_WRITE_THREADS override.
    // (ONE writer: pwrites to one file serialise on its inode lock, and two threads taking turns at it are slower than one that keeps it --
    //  write 0.78 s against 1.4-1.6 s per 16 M reads on tmpfs, profiles/r03/g_formatter_false_sharing.txt)
    const int TWR = getenv("DART_WRITE_THREADS") ? std::max(1, atoi(getenv("DART_WRITE_THREADS"))) : 1;
    const int TF = getenv("DART_FMT_THREADS") ? std::max(1, atoi(getenv("DART_FMT_THREADS"))) : std::max(1, T - TWR - 1);
    const char *wm = getenv("DART_WRITE");
    const bool use_mmap = wm && strcmp(wm, "mmap") == 0;          // default pwrite (measured on tmpfs: 5.1 GB/s against 3.2 GB/s through a shared mapping)
    FastqIndex own;                                   // (unless the caller has indexed this library already, beside dg_init)
    if (pre) pre->wait();
    if (!pre || pre->f1 != f1 || pre->f2 != (f2 ? f2 : "")) { pre = &own; own.run(f1, f2, T); }
    if (!pre->ok) { err = "cannot map the read files"; return DG_ERR_ARG; }
    MappedFile &m1 = pre->m1, &m2 = pre->m2;
    std::vector<FqRec> &r1 = pre->r1, &r2 = pre->r2;
    const bool e1 = pre->e1, e2 = pre->e2;
    st.t_index += pre->t_index;
    // The reads in input order, as GetNextChunk (GetData.cpp:134-179) hands them out: from separate files r1[i], r2[i] alternate (the
    // stream ends with the shorter file, one read of file 1 more if it is the longer); from one file its records in order.
    // An entry without bases ends the reader's current chunk (it is consumed), a chunk without reads ends the stream, and a chunk with
    // an odd number of reads is mapped unpaired (Mapping.cpp:598) and ends the stream here (as in the streaming path of dart_main.cpp).
    // Without such entries the order is implicit; with them the chunking is replayed once to list the reads explicitly.
    std::vector<uint64_t> order;                       // (record index << 1) | file; used only when an entry without bases exists
    const bool explicit_order = e1 || e2;
    size_t n_total, odd_from;
    auto len_of = [&](const std::vector<FqRec> &v, size_t i) -> long long { return (long long)v[i].l1 - 1; };
    if (!explicit_order) {
        n_total = f2 ? (r1.size() > r2.size() ? 2 * r2.size() + 1 : 2 * r1.size()) : r1.size();
        odd_from = n_total;
        if (pair_end && (n_total & 1)) {               // only the last chunk can be odd: find where it starts
            size_t k = 0, start = 0; int count = 0; long long base = 0;
            auto rl_at = [&](size_t q) { return f2 ? len_of((q & 1) ? r2 : r1, q >> 1) : len_of(r1, q); };
            while (k < n_total) {
                base += rl_at(k); count++; k++;
                if (k >= n_total) break;
                base += rl_at(k); count++; k++;
                if (count == 4000 || base > 1000000) { start = k; count = 0; base = 0; }
            }
            odd_from = start;
        }
    } else {
        size_t p1 = 0, p2 = 0;                         // next record of file 1 / file 2
        odd_from = (size_t)-1;
        while (true) {
            const size_t chunk_start = order.size();
            int count = 0; long long base = 0;
            while (true) {
                if (p1 >= r1.size()) break;
                if (len_of(r1, p1) <= 0) { p1++; break; }
                order.push_back(((uint64_t)p1 << 1) | 0u); base += len_of(r1, p1); count++; p1++;
                std::vector<FqRec> &v2 = f2 ? r2 : r1; size_t &q = f2 ? p2 : p1;
                if (q >= v2.size()) break;
                if (len_of(v2, q) <= 0) { q++; break; }
                order.push_back(((uint64_t)q << 1) | (f2 ? 1u : 0u)); base += len_of(v2, q); count++; q++;
                if (count == 4000 || base > 1000000) break;
            }
            if (count == 0) break;
            if (count & 1) { if (pair_end) odd_from = chunk_start; break; }
        }
        n_total = order.size();
        if (odd_from == (size_t)-1) odd_from = n_total;
    }
    auto rec_of = [&](size_t k, const MappedFile *&mf) -> const FqRec & {
        if (explicit_order) { const uint64_t e = order[k]; mf = (e & 1) ? &m2 : &m1; return (e & 1) ? r2[e >> 1] : r1[e >> 1]; }
        if (f2) { mf = (k & 1) ? &m2 : &m1; return (k & 1) ? r2[k >> 1] : r1[k >> 1]; }
        mf = &m1; return r1[k];
    };
    if (batch_reads & 1) batch_reads++;
    // slots + queues
    pool.wait();
    if (pool.slots.size() < ctx.size() + 2) pool.slots.resize(ctx.size() + 2);
    std::vector<FastSlot> &slots = pool.slots;
    std::mutex mu; std::condition_variable cv;
    std::deque<FastSlot *> free_q, map_q; std::map<size_t, FastSlot *> fmt_q;
    for (auto &s : slots) free_q.push_back(&s);
    bool asm_done = false, failed = false; size_t mappers_left = ctx.size();
    int fail_rc = 0;

    std::thread assembler([&]() {
        size_t next = 0, seqno = 0;
        while (next < n_total) {
            FastSlot *s;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !free_q.empty() || failed; }); if (failed) break; s = free_q.front(); free_q.pop_front(); }
            const double ta = now();
            // a batch ends at the start of the unpaired tail, which is a batch of its own
            size_t end = std::min(n_total, next + batch_reads);
            int odd = 0;
            if (next < odd_from && end > odd_from) end = odd_from;
            if (next >= odd_from) { end = n_total; odd = (int)(end - next); }
            const int n = (int)(end - next);
            s->first = next; s->n = n; s->odd = odd; s->seqno = seqno++; s->rc = 0;
            s->view.resize(n); s->soff.resize((size_t)n + 1); s->rl.resize(n);
            size_t bases = 0;
            bool too_long = false;
            for (int k = 0; k < n; k++) {               // offsets: sequential (one add per read)
                const MappedFile *mf; const FqRec &r = rec_of(next + k, mf);
                const uint32_t rlen = r.l1 - 1;
                s->soff[k] = (uint32_t)bases; bases += rlen;
                if (rlen > DG_MAX_RLEN) too_long = true;
                s->rl[k] = (uint16_t)rlen;
            }
            s->soff[n] = (uint32_t)bases;
            if (too_long || bases > 0xFFFFFF00ull) { s->rc = DG_ERR_ARG; s->err = "read longer than DG_MAX_RLEN"; }
            else {
                { const double tl = now(); const size_t z[3] = {0, 0, 0}; slot_reserve(*s, (size_t)n, bases, z); st.t_alloc += now() - tl; }
                const int TA = std::max(1, T / 8);              // (copying bases is a tenth of the formatter's work; T is the budget of ALL stages)
                parallel_for(TA, [&](int tid) {
                    const int lo = (int)((long long)n * tid / TA), hi = (int)((long long)n * (tid + 1) / TA);
                    for (int k = lo; k < hi; k++) {
                        const MappedFile *mf; const FqRec &r = rec_of(next + k, mf);
                        const char *l0 = mf->p + r.off, *l1 = l0 + r.l0, *l3 = l1 + r.l1 + r.l2;
                        const int p1 = hdr_beg_f(l0, (int)r.l0), p2 = hdr_end_f(l0, (int)r.l0);
                        RView &v = s->view[k];
                        v.h = l0 + p1; v.hl = p2 > p1 ? (uint32_t)(p2 - p1) : 0u;
                        v.s = l1; v.sl = r.l1 - 1;
                        v.q = l3; v.ql = std::min<uint32_t>(v.sl, r.l3);                 // qualities cut to the read length (GetData.cpp:118-121)
                        v.rc = pair_end && ((next + k) & 1);                             // mate 2 is stored reverse-complemented (also in an unpaired tail)
                        char *dst = s->seq + s->soff[k];
                        if (v.rc) { for (uint32_t i = 0; i < v.sl; i++) dst[i] = comp_base_f(v.s[v.sl - 1 - i]); }
                        else memcpy(dst, v.s, v.sl);
                    }
                });
            }
            st.t_asm += now() - ta;
            next = end;
            { std::lock_guard<std::mutex> lk(mu); map_q.push_back(s); }
            cv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(mu); asm_done = true; }
        cv.notify_all();
    });

    std::vector<std::thread> mappers;
    for (size_t w = 0; w < ctx.size(); w++) mappers.emplace_back([&, w]() {
        while (true) {
            FastSlot *s;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !map_q.empty() || asm_done || failed; }); if (failed || map_q.empty()) break; s = map_q.front(); map_q.pop_front(); }
            const double tm = now();
            if (!s->rc) {
                const int n = s->n;
                for (int attempt = 0; attempt < 2; attempt++) {
                    if (attempt) { const size_t need[3] = { s->used[0] * 2 + 1024, s->used[1] * 2 + 1024, s->used[2] * 2 + 1024 }; slot_reserve(*s, (size_t)n, 0, need); }
                    dg_params p = base_params; p.paired = (pair_end && !s->odd) ? 1 : 0;
                    dg_set_params(ctx[w], &p);
                    const int rc = dg_map_batch(ctx[w], n, s->soff.data(), s->rl.data(), s->seq, s->ro, s->po, s->cig, s->sj, s->caps, s->used);
                    if (rc == DG_ERR_CAPACITY && attempt == 0) continue;      // `used` holds the need
                    s->rc = rc; if (rc) s->err = dg_last_error(ctx[w]);
                    break;
                }
            }
            { std::lock_guard<std::mutex> lk(mu); st.t_map += now() - tm; if (s->rc) { failed = true; fail_rc = s->rc; err = s->err; } fmt_q[s->seqno] = s; }
            cv.notify_all();
        }
        { std::lock_guard<std::mutex> lk(mu); mappers_left--; }
        cv.notify_all();
    });

    // format (this thread drives; T threads do the work) and write (its own thread, T threads again), batches in order; two sets of
    // text buffers, so batch b+1 is formatted while batch b is written
    struct TextSet { std::vector<TextBuf> bufs; std::vector<Counters> cts; FastSlot *slot = nullptr; bool full = false; };
    TextSet sets[2];
    for (auto &ts : sets) { ts.bufs.resize(TF); ts.cts.resize(TF); }
    bool fmt_done = false;
    std::thread writer([&]() {
        int cur = 0;
        while (true) {
            TextSet &ts = sets[cur];
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return ts.full || fmt_done || failed; }); if (!ts.full) break; }
            const double tw = now();
            FastSlot *s = ts.slot;
            std::vector<size_t> offs(TF + 1, 0);
            for (int k = 0; k < TF; k++) offs[k + 1] = offs[k] + ts.bufs[k].n;
            const uint64_t base = *file_off;
            char *win = nullptr; size_t win_len = 0; const uint64_t a0 = base & ~(uint64_t)4095;
            if (use_mmap && offs[TF] && ftruncate(fd, (off_t)(base + offs[TF])) == 0) {      // copy into a shared mapping of the file's new part
                win_len = (size_t)(base + offs[TF] - a0);
                void *m = mmap(nullptr, win_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a0);
                win = m == MAP_FAILED ? nullptr : (char *)m;
            }
            const int TW = win ? std::max(1, TF / 4) : TWR;     // writes to one file serialise in the kernel: more threads only burn the CPU share
            std::atomic<int> write_errno{0};                 // a full disk must not end as a holed SAM file and exit code 0
            if (use_mmap && offs[TF] && !win) write_errno = errno ? errno : EIO;
            else parallel_for(TW, [&](int wt) {
                for (int tid = wt; tid < TF; tid += TW) {
                    if (win) { memcpy(win + (base - a0) + offs[tid], ts.bufs[tid].b, ts.bufs[tid].n); continue; }
                    size_t done = 0;
                    while (done < ts.bufs[tid].n) {
                        const ssize_t w = pwrite(fd, ts.bufs[tid].b + done, ts.bufs[tid].n - done, (off_t)(base + offs[tid] + done));
                        if (w < 0 && errno == EINTR) continue;
                        if (w <= 0) { write_errno = w < 0 ? errno : ENOSPC; break; }
                        done += (size_t)w;
                    }
                }
            });
            if (win) munmap(win, win_len);
            if (write_errno) {
                { std::lock_guard<std::mutex> lk(mu); failed = true; fail_rc = DG_ERR_INTERNAL; err = std::string("writing the output failed: ") + strerror(write_errno); ts.full = false; }
                cv.notify_all();
                break;
            }
            *file_off = base + offs[TF];
            for (int k = 0; k < TF; k++) { total.unique += ts.cts[k].unique; total.unmapped += ts.cts[k].unmapped; total.paired += ts.cts[k].paired; }
            total.total += s->n;
            for (size_t k = 0; k < s->used[2]; k++) sjmap[std::make_pair(s->sj[k].g1, s->sj[k].g2)]++;   // UpdateLocal/GlobalSJMap, Mapping.cpp:532-577
            if (!silent) { fprintf(stdout, "\r%lld %s tags have been processed in %lld seconds...", total.total, pair_end ? "paired-end" : "singled-end", (long long)(time(NULL) - t0)); fflush(stdout); }
            st.t_write += now() - tw;
            { std::lock_guard<std::mutex> lk(mu); ts.full = false; free_q.push_back(s); }
            cv.notify_all();
            cur ^= 1;
        }
    });
    size_t next_out = 0;
    int fill = 0;
    while (true) {
        FastSlot *s = nullptr;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return fmt_q.count(next_out) || failed || (mappers_left == 0 && fmt_q.empty()); });
          if (failed) break;
          if (fmt_q.count(next_out)) { s = fmt_q[next_out]; fmt_q.erase(next_out); } else break; }
        TextSet &ts = sets[fill];
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !ts.full || failed; }); if (failed) break; }
        const double tf = now();
        const int n = s->n, n_pair_mode = (pair_end && !s->odd) ? n : 0;
        parallel_for(TF, [&](int tid) {
            int lo = (int)((long long)n * tid / TF) & ~1, hi = tid == TF - 1 ? n : ((int)((long long)n * (tid + 1) / TF) & ~1);
            ts.bufs[tid].n = 0; ts.cts[tid] = Counters();
            format_views(s->view.data(), lo, hi, n_pair_mode, s->ro, s->po, s->cig, names, unique_only, multi, true, ts.bufs[tid], ts.cts[tid]);
        });
        st.t_fmt += now() - tf;
        { std::lock_guard<std::mutex> lk(mu); ts.slot = s; ts.full = true; }
        cv.notify_all();
        next_out++; fill ^= 1;
    }
    { std::lock_guard<std::mutex> lk(mu); fmt_done = true; }
    cv.notify_all();
    writer.join();
    if (failed) cv.notify_all();
    assembler.join();
    for (auto &m : mappers) m.join();
    return failed ? (fail_rc ? fail_rc : DG_ERR_INTERNAL) : 0;
}
