// dart_amd/csrc/host/dart_main.cpp -- `dart`: DART's command line over libdartgpu (C ABI only).
//
// Host-side mirror of the reference's driver for the north-star path:
//   flags + defaults + messages + exit codes   main.cpp:96-239
//   index files                                 bwt_index.cpp:15-159,229-251
//   FASTA/FASTQ(.gz) readers, chunking, mate-2 reverse complement   GetData.cpp:44-247
//   SAM header/records, statistics, junctions.tab   Mapping.cpp:208-369,567-577,683-716,741-751,806-822
// Everything between "chunk read" and "records formatted" (Mapping.cpp:598-639) is dg_map_batch.
// -bo: BAM through bam_writer.h (the reference hands its SAM lines to htslib).  `dart index ref.fa prefix`: index_cmd.h over libdartindex.so.
// Out of scope and refused with a message: `update`.
//
// Batches are much larger than the reference's 4000-read chunks (a GPU launch needs >= 10^5 reads);
// output order is input order, which equals the reference at -t 1.  -t keeps its
// meaning for the host side: threads that format SAM text.  DART_GPUS=n spreads batches over n
// devices, DART_INFLIGHT=k (default 1) contexts per device (dg_clone); parsing (one thread per mate file),
// mapping + formatting (one worker per context) and the ordered writer run as a pipeline; DART_BATCH=reads sets the batch size; DART_TIMING=1 prints the stage times.
#include "dartgpu.h"
#include <zlib.h>
#include <sys/stat.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fast_fastq.h"
#include "bam_writer.h"
#include "index_cmd.h"

static const char *VersionStr = "1.4.6";

struct Options {
    const char *index = nullptr, *out = "output.sam";
    char sj[256] = "junctions.tab";
    std::vector<std::string> f1, f2;
    int threads = 4;
    bool pair_end = false, multi = false, unique = false, silent = false, all_sj = false, bam = false;
    dg_params p;
};

static void usage(const char *prog, const Options &o)   // ShowProgramUsage, main.cpp:20-40
{
    fprintf(stdout, "\nDART v%s (Hsin-Nan Lin & Wen-Lian Hsu)\n\n", VersionStr);
    fprintf(stdout, "Usage: %s -i Index_Prefix -f <ReadFile_A1 ReadFile_B1 ...> [-f2 <ReadFile_A2 ReadFile_B2 ...>] -o|-bo Alignment_Output\n\n", prog);
    fprintf(stdout, "Options: -t INT        number of threads [4]\n");
    fprintf(stdout, "         -f            files with #1 mates reads\n");
    fprintf(stdout, "         -f2           files with #2 mates reads\n");
    fprintf(stdout, "         -mis INT      maximal number of mismatches in an alignment\n");
    fprintf(stdout, "         -max_dup INT  maximal number of repetitive fragments (between 100-10000) [%d]\n", o.p.max_dup);
    fprintf(stdout, "         -o            alignment filename in SAM format\n");
    fprintf(stdout, "         -bo           alignment filename in BAM format\n");
    fprintf(stdout, "         -j            splice junction output filename [junctions.tab]\n");
    fprintf(stdout, "         -m            output multiple alignments [false]\n");
    fprintf(stdout, "         -all_sj       detect all splice junction regardless of mapq score [false]\n");
    fprintf(stdout, "         -p            paired-end reads are interlaced in the same file\n");
    fprintf(stdout, "         -unique       output unique alignments\n");
    fprintf(stdout, "         -max_intron   the maximal intron size [500000]\n");
    fprintf(stdout, "         -min_intron   the minimal intron size [10]\n");
    fprintf(stdout, "         -v            version\n");
    fprintf(stdout, "\n");
}

// ---------------------------------------------------------------------------------------------
// index files
// ---------------------------------------------------------------------------------------------
// What the host keeps of the index: the chromosome table of PREFIX.ann (names for the SAM header and the records, offsets for
// junctions.tab).  The three big files (.bwt, .sa, .pac) go from the page cache to HBM inside dg_init_files; the host never holds them.
struct HostIndex {
    std::vector<std::string> names; std::vector<int64_t> off, len;
    int64_t l_pac = 0;
};

static bool file_exists(const std::string &fn) { FILE *f = fopen(fn.c_str(), "r"); if (!f) return false; fclose(f); return true; }

static bool load_ann(const std::string &prefix, HostIndex &ix)      // bwt_index.cpp:37-89 (bns_restore_core)
{
    FILE *f = fopen((prefix + ".ann").c_str(), "r");
    if (!f) return false;
    long long xx; int n; unsigned seed;
    if (fscanf(f, "%lld%d%u", &xx, &n, &seed) != 3) { fclose(f); return false; }
    ix.l_pac = xx;
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        unsigned gi; char name[1024]; int c, l, na;
        if (fscanf(f, "%u%1023s", &gi, name) != 2) { fclose(f); return false; }
        while ((c = fgetc(f)) != '\n' && c != EOF) {}
        if (fscanf(f, "%lld%d%d", &xx, &l, &na) != 3) { fclose(f); return false; }
        ix.names.push_back(name); ix.off.push_back(total); ix.len.push_back(l); total += l;
    }
    fclose(f);
    return n > 0;
}

// ---------------------------------------------------------------------------------------------
// readers
// ---------------------------------------------------------------------------------------------
struct Entry { std::string header, seq, qual; int rlen = 0; };

static char comp_base(char c)   // tools.cpp:3-17
{
    switch (c) { case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A'; default: return 'N'; }
}
static std::string revcomp(const std::string &s) { std::string r(s.size(), 'N'); for (size_t i = 0; i < s.size(); i++) r[i] = comp_base(s[s.size() - 1 - i]); return r; }

static int hdr_beg(const char *s, int len) { for (int i = 1; i < len; i++) if (s[i] != '>' && s[i] != '@') return i; return len - 1; }
static int hdr_end(const char *s, int len) { for (int i = 1; i < len; i++) if (s[i] == ' ' || s[i] == '/' || s[i] == '\t') return i; return len - 1; }

struct Source {
    FILE *fp = nullptr; gzFile gz = nullptr; bool fastq = true;
    char *buf = nullptr; size_t cap = 0;
    // block reader for plain FASTQ: lines are handed out as (pointer into the block, length incl. newline), like getline
    // without the copy and the per-call locking; a line stays valid until the next call
    char *blk = nullptr; size_t blk_cap = 0, blk_beg = 0, blk_end = 0; bool blk_eof = false;
    ssize_t block_line(char **line) {
        while (true) {
            char *nl = blk_end > blk_beg ? (char *)memchr(blk + blk_beg, '\n', blk_end - blk_beg) : nullptr;
            if (nl) { *line = blk + blk_beg; const ssize_t len = nl - (blk + blk_beg) + 1; blk_beg += (size_t)len; return len; }
            if (blk_eof) {
                if (blk_end == blk_beg) return -1;
                *line = blk + blk_beg; const ssize_t len = (ssize_t)(blk_end - blk_beg); blk_beg = blk_end; return len;   // last line without newline
            }
            if (!blk) { blk_cap = (size_t)8 << 20; blk = (char *)malloc(blk_cap + 1); }
            if (blk_beg > 0) { memmove(blk, blk + blk_beg, blk_end - blk_beg); blk_end -= blk_beg; blk_beg = 0; }
            if (blk_end == blk_cap) { blk_cap *= 2; blk = (char *)realloc(blk, blk_cap + 1); }
            const size_t got = fill(blk_cap - blk_end);
            if (got == 0) blk_eof = true;
            blk_end += got;
        }
    }
    // gz: a thread of its own inflates 2 MB pieces ahead of the parser (zlib is the slower of the two: the parser then never waits for it to start)
    struct Inflater {
        std::thread th; std::mutex mu; std::condition_variable cv;
        std::deque<std::vector<char>> q; bool done = false, stop = false; std::vector<char> cur; size_t cur_pos = 0;
        void start(gzFile g) {
            th = std::thread([this, g]() {
                while (true) {
                    std::vector<char> b((size_t)2 << 20);
                    const int got = gzread(g, b.data(), (unsigned)b.size());
                    std::unique_lock<std::mutex> lk(mu);
                    if (got <= 0) { done = true; cv.notify_all(); return; }
                    b.resize((size_t)got);
                    cv.wait(lk, [this]() { return q.size() < 8 || stop; });
                    if (stop) return;
                    q.push_back(std::move(b));
                    cv.notify_all();
                }
            });
        }
        size_t read(char *dst, size_t room) {
            if (cur_pos == cur.size()) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this]() { return !q.empty() || done; });
                if (q.empty()) return 0;
                cur = std::move(q.front()); q.pop_front(); cur_pos = 0;
                cv.notify_all();
            }
            const size_t n = std::min(room, cur.size() - cur_pos);
            memcpy(dst, cur.data() + cur_pos, n); cur_pos += n;
            return n;
        }
        ~Inflater() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
    };
    std::unique_ptr<Inflater> inf;
    size_t fill(size_t room) {                       // the next bytes of the (inflated) stream behind blk_end
        if (!gz) return fread(blk + blk_end, 1, room, fp);
        if (!inf) { inf.reset(new Inflater()); inf->start(gz); }
        return inf->read(blk + blk_end, room);
    }
    // gzgets(file, buffer, 1024) as a view into the block (GetData.cpp:181-210 reads every line that way): up to 1023 bytes, ending behind
    // the first newline among them; 0 = end of the stream.  (zlib inflates in large blocks here instead of once per line.)
    size_t gets_view(char **line) {
        while (true) {
            const size_t avail = blk_end - blk_beg, want = std::min<size_t>(avail, 1023);
            char *nl = want ? (char *)memchr(blk + blk_beg, '\n', want) : nullptr;
            if (nl || want == 1023 || (blk_eof && avail)) {
                *line = blk + blk_beg;
                const size_t len = nl ? (size_t)(nl - (blk + blk_beg)) + 1 : want;
                blk_beg += len;
                return len;
            }
            if (blk_eof) return 0;
            if (!blk) { blk_cap = (size_t)8 << 20; blk = (char *)malloc(blk_cap + 1); }
            if (blk_beg > 0) { memmove(blk, blk + blk_beg, blk_end - blk_beg); blk_end -= blk_beg; blk_beg = 0; }
            if (blk_end == blk_cap) { blk_cap *= 2; blk = (char *)realloc(blk, blk_cap + 1); }
            const size_t got = fill(blk_cap - blk_end);
            if (got == 0) blk_eof = true;
            blk_end += got;
        }
    }
    ~Source() { free(buf); free(blk); }
    // GetNextEntry, GetData.cpp:77-132
    Entry next_plain() {
        Entry e; ssize_t len;
        if ((len = getline(&buf, &cap, fp)) != -1) {
            int p1 = hdr_beg(buf, (int)len), p2 = hdr_end(buf, (int)len);
            if (p2 > p1) e.header.assign(buf + p1, buf + p2);
            if (fastq) {
                ssize_t rl;
                if ((rl = getline(&buf, &cap, fp)) != -1) {
                    e.rlen = (int)rl - 1; e.seq.assign(buf, buf + e.rlen);
                    if (getline(&buf, &cap, fp) == -1) {}
                    ssize_t ql = getline(&buf, &cap, fp);
                    if (ql < 0) ql = 0;
                    size_t take = std::min<size_t>((size_t)e.rlen, strnlen(buf, (size_t)ql));
                    e.qual.assign(buf, buf + take);
                }
            } else {
                while ((len = getline(&buf, &cap, fp)) != -1) {
                    if (buf[0] == '>') { fseek(fp, 0 - (long)len, SEEK_CUR); break; }
                    buf[len - 1] = 0; e.seq += buf;
                }
                e.rlen = (int)e.seq.size();
            }
        }
        return e;
    }
    Entry next() { return next_plain(); }
};

// A run of reads in flat buffers (no per-read allocation: with one std::string per field the allocator, not the
// parsing, was the cost): read k = header [hoff[k],hoff[k+1]), bases [soff[k],soff[k+1]), qualities [qoff[k],qoff[k+1]).
struct Reads {
    std::string hdr, seq, qual;
    std::vector<uint32_t> hoff{0}, soff{0}, qoff{0};
    size_t size() const { return soff.size() - 1; }
    int rlen(size_t k) const { return (int)(soff[k + 1] - soff[k]); }
    void add(const char *h, size_t hl, const char *s, size_t sl, const char *q, size_t ql, bool rc, bool fastq) {
        hdr.append(h, hl); hoff.push_back((uint32_t)hdr.size());
        if (rc) { const size_t o = seq.size(); seq.resize(o + sl); for (size_t i = 0; i < sl; i++) seq[o + i] = comp_base(s[sl - 1 - i]); }
        else seq.append(s, sl);
        soff.push_back((uint32_t)seq.size());
        if (rc && fastq) { const size_t o = qual.size(); qual.resize(o + ql); for (size_t i = 0; i < ql; i++) qual[o + i] = q[ql - 1 - i]; }
        else qual.append(q, ql);
        qoff.push_back((uint32_t)qual.size());
    }
    void add_from(const Reads &r, size_t k) {             // copy read k of r
        hdr.append(r.hdr, r.hoff[k], r.hoff[k + 1] - r.hoff[k]); hoff.push_back((uint32_t)hdr.size());
        seq.append(r.seq, r.soff[k], r.soff[k + 1] - r.soff[k]); soff.push_back((uint32_t)seq.size());
        qual.append(r.qual, r.qoff[k], r.qoff[k + 1] - r.qoff[k]); qoff.push_back((uint32_t)qual.size());
    }
};

// one read from a Source appended to R (mate 2 of a pair: reverse-complemented, qualities reversed, GetData.cpp:160-166);
// false = end of the stream (an entry of length 0, as in the reference).  Plain FASTQ is parsed straight into the flat
// buffers; FASTA and gz go through the Entry readers above.
// gzGetNextEntry, GetData.cpp:181-210, parsed straight into the flat buffers: every line is a gzgets of at most 1023 bytes; a header line
// must start with '@' or '>' and name something; the sequence is the next line without its last byte; FASTQ: two more lines, the second
// holds the qualities (cut to the read length).  FASTA through this reader has ONE sequence line per record, as in the reference.
static bool read_into_gz(Source &s, Reads &R, bool rc)
{
    char *ln; size_t len = s.gets_view(&ln);
    if (len == 0) return false;
    len = strnlen(ln, len);                               // (the reference measures its buffer with strlen)
    const int p1 = hdr_beg(ln, (int)len), p2 = hdr_end(ln, (int)len);
    if (!(p2 - p1 > 0 && (ln[0] == '@' || ln[0] == '>'))) return false;      // an entry of length 0 ends the stream
    const size_t h0 = R.hdr.size(), s0 = R.seq.size();
    R.hdr.append(ln + p1, (size_t)(p2 - p1));
    size_t sl = s.gets_view(&ln);
    sl = sl ? strnlen(ln, sl) : 1;                        // (end of the stream behind a header: an empty line)
    const int rlen = (int)sl - 1;
    if (rlen <= 0) { R.hdr.resize(h0); return false; }
    if (rc) { R.seq.resize(s0 + (size_t)rlen); for (int i = 0; i < rlen; i++) R.seq[s0 + i] = comp_base(ln[rlen - 1 - i]); }
    else R.seq.append(ln, (size_t)rlen);
    if (s.fastq) {
        char *q = nullptr; size_t ql = 0;
        if (s.gets_view(&q) != 0) ql = s.gets_view(&q);
        if (ql == 0) q = nullptr;
        const size_t take = std::min<size_t>((size_t)rlen, q ? strnlen(q, ql) : 0), q0 = R.qual.size();
        if (rc) { R.qual.resize(q0 + take); for (size_t i = 0; i < take; i++) R.qual[q0 + i] = q[take - 1 - i]; }
        else if (take) R.qual.append(q, take);
    }
    R.hoff.push_back((uint32_t)R.hdr.size()); R.soff.push_back((uint32_t)R.seq.size()); R.qoff.push_back((uint32_t)R.qual.size());
    return true;
}

static bool read_into(Source &s, Reads &R, bool rc)
{
    if (s.gz) return read_into_gz(s, R, rc);
    if (!s.fastq) {
        Entry e = s.next();
        if (e.rlen == 0) return false;
        R.add(e.header.data(), e.header.size(), e.seq.data(), e.seq.size(), e.qual.data(), e.qual.size(), rc, s.fastq);
        return true;
    }
    char *ln;                                             // GetNextEntry, GetData.cpp:77-132 (FASTQ branch)
    ssize_t len = s.block_line(&ln);
    if (len == -1) return false;
    const int p1 = hdr_beg(ln, (int)len), p2 = hdr_end(ln, (int)len);
    const size_t h0 = R.hdr.size(), s0 = R.seq.size();
    if (p2 > p1) R.hdr.append(ln + p1, (size_t)(p2 - p1));
    ssize_t rl = s.block_line(&ln);
    const int rlen = rl == -1 ? 0 : (int)rl - 1;
    if (rlen > 0) {
        if (rc) { R.seq.resize(s0 + (size_t)rlen); for (int i = 0; i < rlen; i++) R.seq[s0 + i] = comp_base(ln[rlen - 1 - i]); }
        else R.seq.append(ln, (size_t)rlen);
    }
    if (rl != -1) {
        if (s.block_line(&ln) == -1) {}
        ssize_t ql = s.block_line(&ln);
        if (ql < 0) ql = 0;
        if (rlen > 0) {
            const size_t take = std::min<size_t>((size_t)rlen, ql > 0 ? strnlen(ln, (size_t)ql) : 0), q0 = R.qual.size();
            if (rc) { R.qual.resize(q0 + take); for (size_t i = 0; i < take; i++) R.qual[q0 + i] = ln[take - 1 - i]; }
            else R.qual.append(ln, take);
        }
    }
    if (rlen <= 0) { R.hdr.resize(h0); R.seq.resize(s0); return false; }     // an entry of length 0 ends the stream
    R.hoff.push_back((uint32_t)R.hdr.size()); R.soff.push_back((uint32_t)R.seq.size()); R.qoff.push_back((uint32_t)R.qual.size());
    return true;
}

// A Source parsed by its own thread, BLOCK reads at a time (the two mate files are parsed concurrently; mate 2 is
// reverse-complemented there too).  The consumer walks the blocks in file order.
struct Prefetch {
    static const size_t BLOCK = 16384;
    Source *src = nullptr; bool rc = false;
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::deque<std::unique_ptr<Reads>> q; bool eof = false, stop = false;
    std::unique_ptr<Reads> cur; size_t pos = 0;
    void start(Source *s, bool revcomp_mate) {
        src = s; rc = revcomp_mate;
        th = std::thread([this]() {
            while (true) {
                std::unique_ptr<Reads> blk(new Reads());
                blk->seq.reserve(BLOCK * 160); blk->qual.reserve(BLOCK * 160); blk->hdr.reserve(BLOCK * 24);
                bool end = false;
                while (blk->size() < BLOCK) if (!read_into(*src, *blk, rc)) { end = true; break; }
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this]() { return q.size() < 64 || stop; });
                if (stop) return;
                if (blk->size()) q.push_back(std::move(blk));
                if (end) { eof = true; cv.notify_all(); return; }
                cv.notify_all();
            }
        });
    }
    // appends the next read to R; false at the end of the stream
    bool next_into(Reads &R) {
        if (!cur || pos == cur->size()) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this]() { return !q.empty() || eof; });
            if (q.empty()) return false;
            cur = std::move(q.front()); q.pop_front(); pos = 0;
            cv.notify_all();
        }
        R.add_from(*cur, pos++);
        return true;
    }
    void finish() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (th.joinable()) th.join(); }
    ~Prefetch() { finish(); }
};

// GetNextChunk, GetData.cpp:134-179: one reference-sized chunk appended to `out`
static int next_chunk(Source &s1, Source *s2, bool pair_end, Reads &out, Prefetch *p1 = nullptr, Prefetch *p2 = nullptr)
{
    int count = 0, base = 0;
    while (true) {
        if (!(p1 ? p1->next_into(out) : read_into(s1, out, false))) break;
        base += out.rlen(out.size() - 1); count++;
        if (!(p2 ? p2->next_into(out) : read_into(s2 ? *s2 : s1, out, pair_end))) break;
        base += out.rlen(out.size() - 1); count++;
        if (count == 4000 || base > 1000000) break;
    }
    return count;
}

static bool check_read_format(const char *fn)   // CheckReadFormat, Mapping.cpp:718-726
{
    char b[1] = {0}; gzFile f = gzopen(fn, "rb");
    if (!f) return false;
    gzread(f, b, 1); gzclose(f);
    return b[0] == '@';
}

// ---------------------------------------------------------------------------------------------
// SAM text
// ---------------------------------------------------------------------------------------------
static const char *XS_A[3] = { "", " XS:A:+", " XS:A:-" };

// decimal text without snprintf (the formatter was spending most of its time there)
static inline void put_int(std::string &out, long long v)
{
    char b[24]; int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) out += '-';
    while (n) out += b[--n];
}
static inline void put_cigar(std::string &out, const uint32_t *ops, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) { put_int(out, ops[i] >> 4); out += "MIDNS"[ops[i] & 15 ? (ops[i] & 15) : 0]; }
}

// OutputPairedAlignments / OutputSingledAlignments (Mapping.cpp:208-369) for reads [lo,hi)
static void format_range(const Reads &R, int lo, int hi, int n_pair_mode, const dg_read_out *ro, const dg_report_out *po,
                         const uint32_t *cig, const HostIndex &ix, const Options &o, bool fastq, std::string &out, Counters &ct)
{
    out.reserve((size_t)(hi - lo) * 300);
    struct View { const char *header; size_t hl; const char *seq; size_t sl; const char *qual; size_t ql; int rlen; };
    for (int k = lo; k < hi; k++) {
        const bool is_pair = k < n_pair_mode, mate2 = is_pair && (k & 1);
        View e;
        e.header = R.hdr.data() + R.hoff[k]; e.hl = R.hoff[k + 1] - R.hoff[k];
        e.seq = R.seq.data() + R.soff[k]; e.sl = R.soff[k + 1] - R.soff[k]; e.rlen = (int)e.sl;
        e.qual = R.qual.data() + R.qoff[k]; e.ql = strnlen(e.qual, R.qoff[k + 1] - R.qoff[k]);   // the reference prints a C string
        const dg_read_out &r = ro[k];
        const dg_report_out *rp = po + r.rep_off;
        if (r.score == 0) {
            ct.unmapped++;
            out.append(e.header, e.hl); out += '\t'; put_int(out, rp[0].flag); out += "\t*\t0\t0\t*\t*\t0\t0\t"; out.append(e.seq, e.sl); out += '\t';
            if (fastq) out.append(e.qual, e.ql); else out += '*';
            out += "\tAS:i:0\tXS:i:0\n";
            continue;
        }
        if (!(!o.unique || r.mapq > 3)) continue;
        if (r.mapq == 50) ct.unique++;
        const dg_read_out *m = is_pair ? &ro[k ^ 1] : nullptr;
        const dg_report_out *mp = m ? po + m->rep_off : nullptr;
        for (int j = r.best; j < r.n_rep; j++) {
            const dg_report_out &pr = rp[j];
            const bool show = is_pair ? pr.aln_score > 0 : pr.aln_score == r.score;
            if (show) {
                int xs;
                if (pr.sj_type == -1) xs = 0; else if (pr.sj_type == 0 || pr.sj_type == 2) xs = mate2 ? 2 : 1; else xs = mate2 ? 1 : 2;
                const bool use_alt = mate2 ? pr.bdir == 1 : pr.bdir == 0;
                int pj;
                out.append(e.header, e.hl);
                out += '\t'; put_int(out, pr.flag); out += '\t'; out += ix.names[pr.chr]; out += '\t'; put_int(out, (long long)pr.pos); out += '\t'; put_int(out, r.mapq); out += '\t';
                put_cigar(out, cig + pr.cigar_off, pr.n_cigar);
                if (is_pair && (pj = pr.paired_idx) != -1 && mp[pj].aln_score > 0) {
                    const dg_report_out &a = mate2 ? mp[pj] : pr, &b = mate2 ? pr : mp[pj];
                    const int l1 = mate2 ? R.rlen(k ^ 1) : e.rlen, l2 = mate2 ? e.rlen : R.rlen(k ^ 1);
                    int dist = (int)(b.pos - a.pos + (a.bdir ? l2 : 0 - l1));
                    if (mate2) dist = 0 - dist; else if (j == r.best) ct.paired += 2;
                    out += "\t=\t"; put_int(out, (long long)mp[pj].pos); out += '\t'; put_int(out, dist); out += '\t';
                } else out += "\t*\t0\t0\t";
                if (use_alt) { for (size_t i = e.sl; i-- > 0;) out += comp_base(e.seq[i]); } else out.append(e.seq, e.sl);
                out += '\t';
                if (!fastq) out += '*';
                else if (use_alt) { for (size_t i = e.ql; i-- > 0;) out += e.qual[i]; }
                else out.append(e.qual, e.ql);
                out += "\tNM:i:"; put_int(out, r.mis_num); out += "\tAS:i:"; put_int(out, r.score); out += "\tXS:i:"; put_int(out, r.sub_score); out += XS_A[xs]; out += '\n';
                if (!is_pair && !o.multi) break;
            }
            if (is_pair && !o.multi) break;
        }
    }
}

int main(int argc, char *argv[])
{
    setenv("GPU_MAX_HW_QUEUES", "16", 0);      // before the first HIP call: a hardware queue per stream of the contexts in flight
    Options o;
    dg_params_default(&o.p);
    if (argc == 1 || strcmp(argv[1], "-h") == 0) { usage(argv[0], o); return 0; }
    if (strcmp(argv[1], "update") == 0) {          // main.cpp:120-124 runs git and make: not something a library drop-in does
        fprintf(stderr, "dart (MI355X): the 'update' sub-command is outside this build's scope\n");
        return 0;
    }
    if (strcmp(argv[1], "index") == 0) {           // main.cpp:125-132
        if (argc == 4) return index_cmd::run(argv[0], argv[2], argv[3]);
        fprintf(stderr, "usage: %s index ref.fa prefix\n", argv[0]);
        return 0;
    }
    for (int i = 1; i < argc; i++) {   // main.cpp:136-205
        std::string p = argv[i];
        if (p == "-i") o.index = argv[++i];
        else if (p == "-f") { while (++i < argc && argv[i][0] != '-') o.f1.push_back(argv[i]); i--; }
        else if (p == "-f2") { while (++i < argc && argv[i][0] != '-') o.f2.push_back(argv[i]); i--; }
        else if (p == "-t") { if ((o.threads = atoi(argv[++i])) <= 0) { fprintf(stdout, "Warning! Thread number should be a positive number!\n"); o.threads = 4; } }
        else if (p == "-o") { o.out = argv[++i]; o.bam = false; }                      // main.cpp:158-168
        else if (p == "-bo") { o.out = argv[++i]; o.bam = true; }
        else if (p == "-mis" && i + 1 < argc) o.p.max_mismatch = atoi(argv[++i]);
        else if (p == "-max_dup" && i + 1 < argc) { o.p.max_dup = atoi(argv[++i]); if (o.p.max_dup < 100) o.p.max_dup = 100; else if (o.p.max_dup >= 10000) o.p.max_dup = 10000; }
        else if (p == "-silent") o.silent = true;
        else if (p == "-j") { strncpy(o.sj, argv[++i], 255); o.sj[255] = 0; }
        else if (p == "-p") o.pair_end = true;
        else if (p == "-m") o.multi = true;
        else if (p == "-unique") o.unique = true;
        else if (p == "-all_sj") o.all_sj = true;
        else if (p == "-max_intron") { if ((o.p.max_intron = atoi(argv[++i])) < 100000) o.p.max_intron = 100000; }
        else if (p == "-min_intron") o.p.min_intron = atoi(argv[++i]);
        else if (p == "-d" || p == "-debug") {}   // debug prints are not reproduced
        else if (p == "-v" || p == "--version") { fprintf(stdout, "DART v%s\n\n", VersionStr); return 0; }
        else { fprintf(stderr, "Error! Unknow parameter: %s\n", argv[i]); usage(argv[0], o); return 1; }
    }
    o.p.multi_hit = o.multi; o.p.all_sj = o.all_sj;
    if (o.f1.empty()) { fprintf(stderr, "Error! Please specify a valid read input!\n"); usage(argv[0], o); return 1; }
    if (!o.f2.empty() && o.f1.size() != o.f2.size()) {
        fprintf(stderr, "Error! Paired-end reads input numbers do not match!\n");
        fprintf(stderr, "Read1:\n"); for (auto &s : o.f1) fprintf(stderr, "\t%s\n", s.c_str());
        fprintf(stderr, "Read2:\n"); for (auto &s : o.f2) fprintf(stderr, "\t%s\n", s.c_str());
        return 1;
    }
    {   // CheckInputFiles / CheckOutputFileName, main.cpp:42-94,220
        struct stat st; bool ok = true;
        for (auto &s : o.f1) if (stat(s.c_str(), &st) == -1) { ok = false; fprintf(stderr, "Cannot access file:[%s]\n", s.c_str()); }
        for (auto &s : o.f2) if (stat(s.c_str(), &st) == -1) { ok = false; fprintf(stderr, "Cannot access file:[%s]\n", s.c_str()); }
        if (strcmp(o.out, "output.sam") != 0 && stat(o.out, &st) == 0) {
            if (st.st_mode & S_IFDIR) { ok = false; fprintf(stdout, "Warning: %s is a directory!\n", o.out); }
            else if (!(st.st_mode & S_IFREG)) { ok = false; fprintf(stdout, "Warning: %s is not a regular file!\n", o.out); }
        }
        if (!ok) return 0;
    }
    setenv("DG_BLOCKING_SYNC", "1", 0);      // the mapping threads sleep while their batch is on the GPU instead of spinning: the CPU share belongs to the parser, the formatter and the writer
    const double t_proc0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    size_t batch_reads = getenv("DART_BATCH") ? (size_t)atoll(getenv("DART_BATCH")) : 500000; batch_reads = std::max<size_t>(4000, batch_reads & ~(size_t)1);
    const int inflight_cfg = std::max(1, getenv("DART_INFLIGHT") ? atoi(getenv("DART_INFLIGHT")) : 2);
    FastqIndex pre;                         // the first library's read files are mapped and indexed while the HIP runtime starts, the genome index loads and dg_init_files runs
    bool fast_first = false;
    // A library of .gz FASTQ files small enough to be inflated whole (libdeflate, ~3x zlib; DART_GZ_WHOLE_MAX_GB per file, default 8) goes through the
    // parallel pipeline too -- if the inflated text is one the reference's gz reader and its plain reader see alike (FastqIndex::run); else, and for
    // everything libdeflate cannot take, the streaming reader below.
    const size_t gz_whole_max = (size_t)((getenv("DART_GZ_WHOLE_MAX_GB") ? atof(getenv("DART_GZ_WHOLE_MAX_GB")) : 8.0) * (double)(1ull << 30));
    auto is_gz = [](const std::string &fn) { return fn.substr(fn.find_last_of('.') + 1) == "gz"; };
    auto gz_whole_candidate = [&](size_t lib) -> bool {
        if (o.bam || getenv("DART_STREAMING") || getenv("DART_GZ_STREAM") || !lib_deflate().ok() || gz_whole_max == 0) return false;
        const bool two = o.f1.size() == o.f2.size();
        struct stat st;
        const std::string *fns[2] = {&o.f1[lib], two ? &o.f2[lib] : nullptr};
        for (const std::string *fn : fns) {
            if (!fn) continue;
            if (!is_gz(*fn) || !check_read_format(fn->c_str()) || stat(fn->c_str(), &st) != 0 || (size_t)st.st_size > gz_whole_max / 2) return false;
        }
        return true;
    };
    bool gz_first = false;
    {
        const std::string &fn0 = o.f1[0];
        const bool two = o.f1.size() == o.f2.size();
        if (!o.bam && !getenv("DART_STREAMING") && !is_gz(fn0) && check_read_format(fn0.c_str())) {
            fast_first = true;
            if (!two || !is_gz(o.f2[0])) pre.start(fn0.c_str(), two ? o.f2[0].c_str() : nullptr, o.threads);
        } else if (gz_whole_candidate(0)) {
            fast_first = gz_first = true;
            pre.start(fn0.c_str(), two ? o.f2[0].c_str() : nullptr, o.threads, true, gz_whole_max);
        }
    }
    // every device of the node maps batches (reads shard, the index is replicated; the reference's -t threads over one shared index,
    // Mapping.cpp:792-793); DART_GPUS=n restricts it to the first n
    int n_gpu = dg_device_count();
    if (n_gpu < 1) { fprintf(stderr, "Error! No HIP device (this build of dart has no CPU path)\n"); return 1; }
    if (getenv("DART_GPUS")) n_gpu = std::min(std::max(1, atoi(getenv("DART_GPUS"))), n_gpu);      // (HIP_VISIBLE_DEVICES picks WHICH devices; a job of a few million reads is done before a second device has its index)
    // DART_SAME_DEVICE_TIMES=k (a test hook for one-GPU boxes): k "devices" that are all device 0 -- each with its own index replica, root context and clones --, so
    // that the multi-device pool (several roots, one ordered writer: Mapping.cpp:644-664) runs where only one GPU exists
    const int same_device_times = getenv("DART_SAME_DEVICE_TIMES") ? std::max(1, std::min(8, atoi(getenv("DART_SAME_DEVICE_TIMES")))) : 0;
    if (same_device_times) n_gpu = same_device_times;
    SlotPool pool;                          // batch slots of the parallel FASTQ pipeline (page-locked in the background with DART_PINNED=1)
    if (fast_first) pool.start((size_t)n_gpu * inflight_cfg + 2, batch_reads, 160);
    // The read files of a library.  Plain FASTQ goes through the parallel host pipeline (fast_fastq.h); FASTA, .gz, -bo and DART_STREAMING=1 through the
    // streaming one: one thread per mate file inflates (zlib, 1 MB at a time) and parses into flat buffers.  Library 0 is opened HERE, before the
    // index goes to the GPU, so that its files are read while the HIP runtime starts and dg_init_files runs.
    struct LibIO { Source s1, s2; Prefetch pf1, pf2; bool sep = false, gz = false, fastq = false, fast_host = false; int state = -1; };
    std::vector<std::unique_ptr<LibIO>> libs(o.f1.size());
    auto open_lib = [&](size_t lib) -> int {        // 0 = opened, 1 = a file cannot be opened (the library is skipped, Mapping.cpp:765-779), 2 = mates in different formats
        if (libs[lib]) return libs[lib]->state;
        libs[lib].reset(new LibIO());
        LibIO &L = *libs[lib];
        const std::string &fn = o.f1[lib];
        L.gz = fn.substr(fn.find_last_of('.') + 1) == "gz";
        L.fastq = check_read_format(fn.c_str());
        L.s1.fastq = L.s2.fastq = L.fastq;
        if (L.gz) L.s1.gz = gzopen(fn.c_str(), "rb"); else L.s1.fp = fopen(fn.c_str(), "r");
        if (o.f1.size() == o.f2.size()) {
            L.sep = true;
            if (L.fastq != check_read_format(o.f2[lib].c_str())) { fprintf(stderr, "Error! %s and %s are with different format...\n", fn.c_str(), o.f2[lib].c_str()); return L.state = 2; }
            if (L.gz) L.s2.gz = gzopen(o.f2[lib].c_str(), "rb"); else L.s2.fp = fopen(o.f2[lib].c_str(), "r");
        }
        if (!L.s1.fp && !L.s1.gz) return L.state = 1;
        if (L.sep && !L.s2.fp && !L.s2.gz) return L.state = 1;
        if (L.s1.gz) gzbuffer(L.s1.gz, 1u << 20);
        if (L.s2.gz) gzbuffer(L.s2.gz, 1u << 20);
        L.fast_host = L.fastq && !L.gz && !o.bam && !getenv("DART_STREAMING") && (!L.sep || o.f2[lib].substr(o.f2[lib].find_last_of('.') + 1) != "gz");
        if (L.sep && !L.fast_host) { L.pf1.start(&L.s1, false); L.pf2.start(&L.s2, true); }      // (separate mate files: paired, mate 2 reverse-complemented)
        return L.state = 0;
    };
    if (!fast_first && o.index && file_exists(std::string(o.index) + ".ann")) open_lib(0);      // (a whole-file .gz library that turns out not to qualify opens its streams late: rare)
    HostIndex ix;
    if (!o.index || !file_exists(std::string(o.index) + ".ann") || !file_exists(std::string(o.index) + ".amb") || !file_exists(std::string(o.index) + ".pac")) {
        fprintf(stderr, "Error! Please specify a valid reference index!\n"); usage(argv[0], o); return 1;
    }
    fprintf(stdout, "Load the genome index files...");
    if (!load_ann(o.index, ix)) { fprintf(stderr, "\n\nError! Index files are corrupt!\n"); return 0; }

    // The contexts: one root per device (the index files go to its HBM inside dg_init_files, main.cpp:231-235 / bwt_index.cpp:147-159), clones
    // for the batches in flight.  The look-up aids are built in the background while the first batches are mapped (DART_SYNC_AIDS=1: before).
    std::vector<dg_ctx *> ctx, roots, clones;
    {
        const std::string pb = std::string(o.index) + ".bwt", ps = std::string(o.index) + ".sa", pp = std::string(o.index) + ".pac";
        dg_index_files files;
        files.bwt_path = pb.c_str(); files.sa_path = ps.c_str(); files.pac_path = pp.c_str();
        files.l_pac = ix.l_pac; files.n_chr = (int32_t)ix.names.size(); files.chr_off = ix.off.data(); files.chr_len = ix.len.data();
        {   // the size of the job, from the sizes of its input files: ~250 bytes per FASTQ read of 2x101 (a quarter of that gzipped; FASTA: half).
            // A coarse figure is all the library needs: it decides between lean and full look-up aids at some hundred million reads.
            // An input whose size cannot be known (a FIFO, process substitution, /dev/stdin, a failed stat) makes the whole job "unknown" (0): the library then
            // builds the full aids -- a very large piped job must not run its seeding stage at half speed for its whole length
            uint64_t bytes = 0; struct stat st; bool known = true;
            for (auto &v : {&o.f1, &o.f2}) for (auto &fn : *v) {
                if (stat(fn.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) bytes += (uint64_t)st.st_size * (fn.size() > 3 && fn.substr(fn.size() - 3) == ".gz" ? 4u : 1u);
                else known = false;
            }
            files.expected_reads = known ? std::max<uint64_t>(1, bytes / 250) : 0;
            if (getenv("DART_EXPECTED_READS")) files.expected_reads = (uint64_t)atoll(getenv("DART_EXPECTED_READS"));
        }
        o.p.paired = (o.pair_end || o.f1.size() == o.f2.size()) ? 1 : 0;
        const int init_flags = (getenv("DART_SYNC_AIDS") && atoi(getenv("DART_SYNC_AIDS"))) ? 0 : DG_INIT_ASYNC_AIDS;
        roots.assign(n_gpu, nullptr);
        std::vector<int> st(n_gpu, 0); std::vector<std::string> msg(n_gpu);
        std::vector<std::thread> th;
        for (int d = 0; d < n_gpu; d++) th.emplace_back([&, d]() { roots[d] = dg_init_files(&files, &o.p, same_device_times ? 0 : d, init_flags, &st[d]); if (!roots[d]) msg[d] = dg_last_error(nullptr); });
        for (auto &t : th) t.join();
        for (int d = 0; d < n_gpu; d++) if (!roots[d]) {
            if (st[d] == DG_ERR_ARG) fprintf(stderr, "\n\nError! Index files are corrupt!\n"); else fprintf(stderr, "\nError! GPU %d: %s\n", d, msg[d].c_str());
            for (auto c : roots) if (c) dg_destroy(c);
            return st[d] == DG_ERR_ARG ? 0 : 1;
        }
        for (int d = 0; d < n_gpu; d++) {
            ctx.push_back(roots[d]);
            for (int k = 1; k < inflight_cfg; k++) {
                int st2 = 0; dg_ctx *cl = dg_clone(roots[d], &st2);
                if (!cl) { fprintf(stderr, "Error! GPU %d: %s\n", d, dg_last_error(nullptr)); return 1; }
                ctx.push_back(cl); clones.push_back(cl);
            }
        }
    }
    const double t_init1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stdout, "\nLoad the reference sequences...\n");

    std::string hdr_text = std::string("@PG\tID:Dart\tPN:Dart\tVN:") + VersionStr + "\n";      // Mapping.cpp:741-751
    for (size_t i = 0; i < ix.names.size(); i++) hdr_text += "@SQ\tSN:" + ix.names[i] + "\tLN:" + std::to_string((long long)ix.len[i]) + "\n";
    FILE *sam = nullptr;
    BamWriter bam;
    if (o.bam) {
        if (!bam.open(o.out, hdr_text, ix.names, std::vector<int64_t>(ix.len.begin(), ix.len.end()), o.threads)) { fprintf(stderr, "Cannot write %s\n", o.out); return 1; }
    } else {
        sam = fopen(o.out, "w+");     // (read access too: the parallel pipeline maps the file to write it)
        if (!sam) { fprintf(stderr, "Cannot write %s\n", o.out); return 1; }
        fwrite(hdr_text.data(), 1, hdr_text.size(), sam);
    }

    Counters total;
    std::map<std::pair<int64_t, int64_t>, int> sjmap;
    time_t t0 = time(NULL);
    if (o.silent) fprintf(stdout, "Start read mapping...\n");
    bool pair_end = o.pair_end;
    for (size_t lib = 0; lib < o.f1.size(); lib++) {
        {   // .gz FASTQ inflated whole: the parallel pipeline on the inflated text (library 0's files have been inflating since before the GPU was touched)
            std::unique_ptr<FastqIndex> own_gz;
            FastqIndex *gi = nullptr;
            if (lib == 0 && gz_first) { pre.wait(); gi = &pre; }
            else if (lib > 0 && gz_whole_candidate(lib)) { own_gz.reset(new FastqIndex()); own_gz->run(o.f1[lib].c_str(), o.f1.size() == o.f2.size() ? o.f2[lib].c_str() : nullptr, o.threads, true, gz_whole_max); gi = own_gz.get(); }
            if (gi && gi->ok) {
                const bool sep = o.f1.size() == o.f2.size();
                if (sep) pair_end = true;
                fflush(sam);
                uint64_t off = (uint64_t)ftello(sam);
                std::string ferr; FastStats fst;
                const int frc = run_fast_library(o.f1[lib].c_str(), sep ? o.f2[lib].c_str() : nullptr, pair_end, o.threads, batch_reads, ctx, o.p, ix.names, o.unique, o.multi, o.silent,
                                                 fileno(sam), &off, total, sjmap, t0, ferr, fst, pool, gi);
                fseeko(sam, (off_t)off, SEEK_SET);
                if (frc) { fprintf(stderr, "\nError! GPU mapping failed (%d): %s\n", frc, ferr.c_str()); return 1; }
                if (getenv("DART_TIMING")) fprintf(stderr, "[dart timing] start-up %.3f s (%s), inflate (libdeflate, whole files) + index %.3f s, assemble %.3f s (of which page-locked allocation %.3f s), map (sum over workers) %.3f s, format %.3f s, write %.3f s\n", t_init1 - t_proc0, dg_init_report(roots[0]), fst.t_index, fst.t_asm, fst.t_alloc, fst.t_map, fst.t_fmt, fst.t_write);
                gi->m1.close_now(); gi->m2.close_now();
                continue;
            }
        }
        const int orc = open_lib(lib);              // (library 0 was opened before the index went to the GPU: its files are being inflated and parsed since)
        if (orc == 2) return 1;
        if (orc == 1) continue;
        LibIO &L = *libs[lib];
        Source &s1 = L.s1, &s2 = L.s2; Prefetch &pf1 = L.pf1, &pf2 = L.pf2;
        const bool sep = L.sep, gz = L.gz, fastq = L.fastq, fast_host = L.fast_host;
        const std::string &fn = o.f1[lib];
        if (sep) pair_end = true;
        if (fast_host) {
            fflush(sam);
            uint64_t off = (uint64_t)ftello(sam);
            std::string ferr; FastStats fst;
            const int frc = run_fast_library(fn.c_str(), sep ? o.f2[lib].c_str() : nullptr, pair_end, o.threads, batch_reads, ctx, o.p, ix.names, o.unique, o.multi, o.silent,
                                             fileno(sam), &off, total, sjmap, t0, ferr, fst, pool, lib == 0 ? &pre : nullptr);
            fseeko(sam, (off_t)off, SEEK_SET);
            if (frc) { fprintf(stderr, "\nError! GPU mapping failed (%d): %s\n", frc, ferr.c_str()); return 1; }
            if (getenv("DART_TIMING")) fprintf(stderr, "[dart timing] start-up %.3f s (%s), index %.3f s, assemble %.3f s (of which page-locked allocation %.3f s), map (sum over workers) %.3f s, format %.3f s, write %.3f s\n", t_init1 - t_proc0, dg_init_report(roots[0]), fst.t_index, fst.t_asm, fst.t_alloc, fst.t_map, fst.t_fmt, fst.t_write);
            if (s1.fp) fclose(s1.fp);
            if (s2.fp) fclose(s2.fp);
            continue;
        }
        // ---- three-stage pipeline: reader thread -> mapping workers (one per context) -> ordered writer ----
        // Contexts = devices x DART_INFLIGHT (dg_clone: contexts of a device share its index), so the next batch is
        // parsed, and the previous one formatted and written, while the GPUs map.  Output order = input order.
        struct Batch { Reads rd; int odd = 0; std::unique_ptr<dg_read_out[]> ro; std::unique_ptr<dg_report_out[]> po; std::unique_ptr<uint32_t[]> cig; std::unique_ptr<dg_sj_out[]> sj;   // new T[n]: no zero fill
                       size_t n_reads = 0; size_t used[3] = {0, 0, 0}; int rc = 0; std::string err; size_t seq = 0; std::vector<std::string> outs; std::vector<Counters> cts; };
        std::mutex mu; std::condition_variable cv_in, cv_out, cv_space;
        std::deque<std::unique_ptr<Batch>> inq; std::map<size_t, std::unique_ptr<Batch>> done;
        bool reader_done = false, failed = false; size_t in_flight = 0;
        const size_t max_in_flight = ctx.size() + 2;
        double t_read = 0, t_map = 0, t_fmt = 0, t_write = 0;
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

        std::thread reader([&]() {
            bool eof = false; size_t seqno = 0;
            while (!eof) {
                // an odd chunk (only the last can be) is mapped on its own, unpaired (Mapping.cpp:598)
                const double t = now();
                std::unique_ptr<Batch> b(new Batch());
                b->rd.seq.reserve(batch_reads * 110); b->rd.qual.reserve(batch_reads * 110); b->rd.hdr.reserve(batch_reads * 16);
                while (b->rd.size() < batch_reads) {
                    int c = next_chunk(s1, sep ? &s2 : nullptr, pair_end, b->rd, sep ? &pf1 : nullptr, sep ? &pf2 : nullptr);
                    if (c == 0) { eof = true; break; }
                    if (c & 1) { b->odd = c; eof = true; break; }
                }
                t_read += now() - t;
                if (b->rd.size() == 0) break;
                b->seq = seqno++;
                std::unique_lock<std::mutex> lk(mu);
                cv_space.wait(lk, [&]() { return in_flight < max_in_flight || failed; });
                if (failed) break;
                in_flight++; inq.push_back(std::move(b));
                cv_in.notify_one();
            }
            std::lock_guard<std::mutex> lk(mu);
            reader_done = true; cv_in.notify_all(); cv_out.notify_all();
        });

        auto map_one = [&](Batch &b, dg_ctx *c) {
            const int n = (int)b.rd.size(), n_even = n - b.odd;
            const std::vector<uint32_t> &off = b.rd.soff; std::vector<uint16_t> rl(n); const std::string &flat = b.rd.seq;   // the parsed bases are the device input
            for (int k = 0; k < n; k++) {
                if (b.rd.rlen(k) > DG_MAX_RLEN) { b.rc = DG_ERR_ARG; b.err = "read longer than DG_MAX_RLEN"; return; }
                rl[k] = (uint16_t)b.rd.rlen(k);
            }
            b.ro.reset(new dg_read_out[n]); b.n_reads = (size_t)n;
            size_t caps[3] = { (size_t)n * 4 + 1024, (size_t)n * 16 + 4096, (size_t)n + 1024 };
            for (int attempt = 0; attempt < 2; attempt++) {
                b.po.reset(new dg_report_out[caps[0]]); b.cig.reset(new uint32_t[caps[1]]); b.sj.reset(new dg_sj_out[caps[2]]);
                size_t used1[3] = {0, 0, 0}, used2[3] = {0, 0, 0};
                dg_params p = o.p; p.paired = pair_end ? 1 : 0;
                dg_set_params(c, &p);
                int rc = n_even ? dg_map_batch(c, n_even, off.data(), rl.data(), flat.data(), b.ro.get(), b.po.get(), b.cig.get(), b.sj.get(), caps, used1) : 0;
                if (rc == 0 && b.odd) {
                    p.paired = 0; dg_set_params(c, &p);
                    size_t caps2[3] = { caps[0] - used1[0], caps[1] - used1[1], caps[2] - used1[2] };
                    rc = dg_map_batch(c, b.odd, off.data() + n_even, rl.data() + n_even, flat.data(), b.ro.get() + n_even, b.po.get() + used1[0],
                                      b.cig.get() + used1[1], b.sj.get() + used1[2], caps2, used2);
                    if (rc == DG_ERR_CAPACITY) for (int q = 0; q < 3; q++) used2[q] += used1[q];
                    else {
                        for (int k = n_even; k < n; k++) { b.ro[k].rep_off += (int32_t)used1[0]; b.ro[k].sj_off += (int32_t)used1[2]; }
                        for (size_t k = 0; k < used2[0]; k++) b.po[used1[0] + k].cigar_off += (uint32_t)used1[1];
                        for (size_t k = 0; k < used2[2]; k++) b.sj[used1[2] + k].read_idx += n_even;
                    }
                }
                if (rc == DG_ERR_CAPACITY && attempt == 0) {       // `used` holds the need: grow once and repeat
                    for (int q = 0; q < 3; q++) caps[q] = std::max(caps[q], used1[q] + used2[q]) * 2 + 1024;
                    continue;
                }
                b.rc = rc; if (rc) b.err = dg_last_error(c);
                for (int q = 0; q < 3; q++) b.used[q] = used1[q] + used2[q];
                return;
            }
        };
        auto format_one = [&](Batch &b) {
            const int n = (int)b.rd.size(), n_pair_mode = pair_end ? n - b.odd : 0;
            const int nt = std::max(1, std::min(o.threads, n / 2000 + 1));
            b.outs.assign(nt, std::string()); b.cts.assign(nt, Counters());
            std::vector<std::thread> ft;
            int per = ((n + nt - 1) / nt + 1) & ~1;
            for (int t = 0; t < nt; t++) {
                const int lo = std::min(n, t * per), hi = std::min(n, (t + 1) * per);
                ft.emplace_back([&, t, lo, hi]() { format_range(b.rd, lo, hi, n_pair_mode, b.ro.get(), b.po.get(), b.cig.get(), ix, o, fastq, b.outs[t], b.cts[t]); });
            }
            for (auto &t : ft) t.join();
            b.rd = Reads();                                      // the text is all the writer needs
        };
        std::vector<std::thread> workers;
        for (size_t w = 0; w < ctx.size(); w++) workers.emplace_back([&, w]() {
            while (true) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_in.wait(lk, [&]() { return !inq.empty() || reader_done || failed; });
                    if (failed || inq.empty()) return;
                    b = std::move(inq.front()); inq.pop_front();
                }
                double t = now();
                map_one(*b, ctx[w]);
                const double tm = now() - t; t = now();
                if (!b->rc) format_one(*b);
                const double tf = now() - t;
                std::lock_guard<std::mutex> lk(mu);
                t_map += tm; t_fmt += tf;
                if (b->rc) failed = true;
                const size_t k = b->seq;
                done[k] = std::move(b);
                cv_out.notify_all(); if (failed) { cv_in.notify_all(); cv_space.notify_all(); }
            }
        });
        // ordered writer (this thread)
        size_t next_out = 0; bool bad = false; std::string bad_msg; int bad_rc = 0;
        while (true) {
            std::unique_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_out.wait(lk, [&]() { return done.count(next_out) || failed || (reader_done && in_flight == 0); });
                if (done.count(next_out)) { b = std::move(done[next_out]); done.erase(next_out); }
                else if (failed) { for (auto &kv : done) if (kv.second->rc) { bad_rc = kv.second->rc; bad_msg = kv.second->err; } bad = true; }
                else break;
            }
            if (bad) break;
            if (b->rc) { bad = true; bad_rc = b->rc; bad_msg = b->err; std::lock_guard<std::mutex> lk(mu); failed = true; cv_in.notify_all(); cv_space.notify_all(); break; }
            const double t = now();
            const size_t n = b->n_reads;
            if (o.bam) {                                     // Mapping.cpp:655-662: every SAM line -> one BAM record
                std::vector<std::pair<const char *, size_t>> chunks;
                for (size_t t2 = 0; t2 < b->outs.size(); t2++) chunks.emplace_back(b->outs[t2].data(), b->outs[t2].size());
                bam.add_sam_chunks(chunks);
            }
            for (size_t t2 = 0; t2 < b->outs.size(); t2++) {
                if (!o.bam) fwrite(b->outs[t2].data(), 1, b->outs[t2].size(), sam);
                total.unique += b->cts[t2].unique; total.unmapped += b->cts[t2].unmapped; total.paired += b->cts[t2].paired;
            }
            total.total += (long long)n;
            for (size_t k = 0; k < b->used[2]; k++) sjmap[std::make_pair(b->sj[k].g1, b->sj[k].g2)]++;   // UpdateLocal/GlobalSJMap, Mapping.cpp:532-577
            if (!o.silent) { fprintf(stdout, "\r%lld %s tags have been processed in %lld seconds...", total.total, pair_end ? "paired-end" : "singled-end", (long long)(time(NULL) - t0)); fflush(stdout); }
            t_write += now() - t;
            next_out++;
            std::lock_guard<std::mutex> lk(mu);
            in_flight--; cv_space.notify_all();
        }
        reader.join();
        pf1.finish(); pf2.finish();
        for (auto &w : workers) w.join();
        if (bad) { fprintf(stderr, "\nError! GPU mapping failed (%d): %s\n", bad_rc, bad_msg.c_str()); return 1; }
        if (getenv("DART_TIMING")) fprintf(stderr, "[dart timing] read+parse %.3f s, map (sum over workers) %.3f s, format %.3f s, write %.3f s\n", t_read, t_map, t_fmt, t_write);
        if (s1.fp) fclose(s1.fp);
        if (s2.fp) fclose(s2.fp);
        s1.inf.reset(); s2.inf.reset();              // (the inflater threads end before their files are closed)
        if (s1.gz) gzclose(s1.gz);
        if (s2.gz) gzclose(s2.gz);
    }
    if (!o.silent) fprintf(stdout, "\rAll the %lld %s reads have been processed in %lld seconds.\n", total.total, pair_end ? "paired-end" : "single-end", (long long)(time(NULL) - t0));
    if (o.bam) { if (!bam.close()) { fprintf(stderr, "Error while writing %s\n", o.out); return 1; } }
    else fclose(sam);
    for (auto c : clones) dg_destroy(c);
    for (auto c : roots) dg_destroy(c);

    if (total.total > 0) {   // Mapping.cpp:812-822
        const long long T = total.total, U = total.unmapped;
        if (pair_end) fprintf(stdout, "\t# of total mapped reads = %lld (sensitivity = %.2f%%)\n\t# of paired sequences = %lld (%.2f%%)\n", T - U, (int)(10000 * (1.0 * (T - U) / T) + 0.5) / 100.0, total.paired, (int)(10000 * (1.0 * total.paired / T) + 0.5) / 100.0);
        else fprintf(stdout, "\t# of total mapped reads = %lld (sensitivity = %.2f%%)\n", T - U, (int)(10000 * (1.0 * (T - U) / T) + 0.5) / 100.0);
        fprintf(stdout, "\t# of unique mapped reads = %lld (%.2f%%)\n", total.unique, (int)(10000 * (1.0 * total.unique / T) + 0.5) / 100.0);
        if (!o.unique) fprintf(stdout, "\t# of multiple mapped reads = %lld (%.2f%%)\n", T - U - total.unique, (int)(10000 * (1.0 * (T - U - total.unique) / T) + 0.5) / 100.0);
        fprintf(stdout, "\t# of unmapped reads = %lld (%.2f%%)\n", U, (int)(10000 * (1.0 * U / T) + 0.5) / 100.0);
        // OutputSpliceJunctions, Mapping.cpp:683-716
        FILE *jf = fopen(o.sj, "w");
        int nj = 0;
        const int nc = (int)ix.names.size();
        std::vector<int64_t> key(2 * nc); std::vector<int> who(2 * nc);
        for (int i = 0; i < nc; i++) { key[i] = ix.off[i] + ix.len[i] - 1; who[i] = i; key[2 * nc - 1 - i] = 2 * ix.l_pac - ix.off[i] - 1; who[2 * nc - 1 - i] = i; }
        for (auto &kv : sjmap) {
            const int lo = (int)(std::lower_bound(key.begin(), key.end(), kv.first.first) - key.begin());
            if (lo >= 2 * nc) continue;
            const int c = who[lo];
            if (jf) fprintf(jf, "%s\t%lld\t%lld\t%d\n", ix.names[c].c_str(), (long long)(kv.first.first + 1 - ix.off[c]), (long long)(kv.first.second + 1 - ix.off[c]), kv.second);
            nj++;
        }
        if (jf) fclose(jf);
        fprintf(stdout, "\t# of splice junctions = %d (file: %s)\n", nj, o.sj);
        fprintf(stdout, "\tAlignment output: %s\n\n", o.out);
    }
    return 0;
}
