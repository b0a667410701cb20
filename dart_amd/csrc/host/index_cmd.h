// dart_amd/csrc/host/index_cmd.h -- `dart index ref.fa prefix` (main.cpp:125-127 -> bwa_idx_build, BWT_Index/bwtindex.c:77-148).
//
// Host side: the FASTA reader and the packing rules of bns_fasta2bntseq / add1 (bntseq.c:104-156,158-215) -- names and comments as
// kseq.h:175-204 splits them, a run of one ambiguous character = one hole of PREFIX.amb, every ambiguous base replaced by
// lrand48() & 3 after srand48(11) (libc's own generator: the same stream as the reference's) -- and the text files PREFIX.ann /
// PREFIX.amb in bns_dump's format (bntseq.c:59-89).  Device side: di_build_files (include/dartindex.h, libdartindex.so next to this
// program) writes PREFIX.pac / .bwt / .sa.  The files are the reference indexer's bytes (tests/test_gpu_cli.py).
#pragma once
#include "dartindex.h"
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <ctime>
#include <string>
#include <vector>

namespace index_cmd {

struct Ann { std::string name, anno; long long offset; long long len; int n_ambs; };
struct Amb { long long offset; int len; char amb; };

struct LineReader {                                       // lines of a plain or gzipped file, '\n' stripped (and a '\r' in front of it)
    gzFile fp; std::vector<char> buf; size_t at = 0, end = 0; bool eof = false;
    explicit LineReader(gzFile f) : fp(f), buf(1 << 22) {}
    bool next(std::string &line)
    {
        line.clear();
        for (;;) {
            if (at == end) {
                if (eof) return !line.empty();
                const int got = gzread(fp, buf.data(), (unsigned)buf.size());
                if (got <= 0) { eof = true; return !line.empty(); }
                at = 0; end = (size_t)got;
            }
            const char *nl = (const char *)memchr(buf.data() + at, '\n', end - at);
            if (nl) {
                line.append(buf.data() + at, (size_t)(nl - (buf.data() + at)));
                at = (size_t)(nl - buf.data()) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(buf.data() + at, end - at);
            at = end;
        }
    }
};

static inline int nt4(unsigned char c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

static void log_line(const char *line, void *) { fprintf(stdout, "%s\n", line); fflush(stdout); }

static int run(const char *self, const char *fa, const char *prefix)
{
    const clock_t t0 = clock();
    gzFile fp = gzopen(fa, "r");
    if (!fp) { fprintf(stderr, "[dart index] fail to open file '%s'\n", fa); return 1; }
    gzbuffer(fp, 1 << 20);
    fprintf(stdout, "[dart index] Pack FASTA... "); fflush(stdout);
    std::vector<Ann> anns; std::vector<Amb> ambs; std::vector<uint8_t> fwd;
    srand48(11);                                          // bntseq.c:173-174: a fixed seed
    uint8_t tab[256];
    for (int i = 0; i < 256; i++) tab[i] = (uint8_t)nt4((unsigned char)i);
    {
        struct stat sb;
        if (stat(fa, &sb) == 0 && S_ISREG(sb.st_mode)) fwd.reserve((size_t)sb.st_size + (size_t)sb.st_size / 8);      // plain FASTA: about its size; .gz grows as it goes
    }
    LineReader in(fp);
    std::string line;
    bool in_record = false;
    int lasts = 0;
    while (in.next(line)) {
        if (line.empty()) continue;
        if (line[0] == '>' || line[0] == '@') {           // header: the name ends at the first white space, the rest of the line is the comment
            size_t e = 1;
            while (e < line.size() && !isspace((unsigned char)line[e])) e++;
            Ann a; a.name = line.substr(1, e - 1);
            a.anno = e + 1 <= line.size() && e < line.size() ? line.substr(e + 1) : std::string();
            if (a.anno.empty()) a.anno = "(null)";
            a.offset = (long long)fwd.size(); a.len = 0; a.n_ambs = 0;
            anns.push_back(a);
            in_record = true; lasts = 0;
            continue;
        }
        if (!in_record) continue;                         // text in front of the first header
        if (line[0] == '+') {                             // a FASTQ record: the quality block is as long as the sequence
            long long left = anns.back().len;
            while (left > 0 && in.next(line)) left -= (long long)line.size();
            in_record = false;
            continue;
        }
        Ann &a = anns.back();
        const size_t base = fwd.size(), n = line.size();
        if (fwd.capacity() < base + n) fwd.reserve(std::max(base + n, fwd.capacity() + fwd.capacity() / 2 + (size_t)(1 << 20)));
        fwd.resize(base + n);
        uint8_t *dst = fwd.data() + base, bad = 0;
        for (size_t i = 0; i < n; i++) { const uint8_t c = tab[(unsigned char)line[i]]; dst[i] = c; bad |= c; }
        if (bad & 4) {                                    // the line holds ambiguous characters: holes and random bases, in sequence order
            for (size_t i = 0; i < n; i++) {
                const unsigned char ch = (unsigned char)line[i];
                if (dst[i] >= 4) {
                    if (lasts == (int)ch) ambs.back().len++;  // the run of one ambiguous character goes on
                    else { ambs.push_back(Amb{(long long)(base + i), 1, (char)ch}); a.n_ambs++; }
                    dst[i] = (uint8_t)(lrand48() & 3);
                }
                lasts = (int)ch;
            }
        } else lasts = (int)(unsigned char)line[n - 1];
        a.len += (long long)n;
    }
    gzclose(fp);
    const long long L = (long long)fwd.size();
    {
        FILE *f = fopen((std::string(prefix) + ".ann").c_str(), "w");
        if (!f) { fprintf(stderr, "\n[dart index] fail to write '%s.ann'\n", prefix); return 1; }
        fprintf(f, "%lld %d %u\n", L, (int)anns.size(), 11u);
        for (const Ann &a : anns) {
            fprintf(f, "%d %s", 0, a.name.c_str());
            if (!a.anno.empty()) fprintf(f, " %s\n", a.anno.c_str()); else fprintf(f, "\n");
            fprintf(f, "%lld %d %d\n", a.offset, (int)a.len, a.n_ambs);
        }
        fclose(f);
        f = fopen((std::string(prefix) + ".amb").c_str(), "w");
        if (!f) { fprintf(stderr, "\n[dart index] fail to write '%s.amb'\n", prefix); return 1; }
        fprintf(f, "%lld %d %u\n", L, (int)anns.size(), (unsigned)ambs.size());
        for (const Amb &h : ambs) fprintf(f, "%lld %d %c\n", h.offset, h.len, h.amb);
        fclose(f);
    }
    fprintf(stdout, "%.2f sec\n", (float)(clock() - t0) / CLOCKS_PER_SEC);
    if (L < 32) { fprintf(stderr, "[dart index] '%s' holds %lld bases: nothing to index\n", fa, L); return 1; }
    // libdartindex.so sits next to this program
    std::string dir = ".";
    {
        char exe[4096]; const ssize_t k = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (k > 0) { exe[k] = 0; const char *s = strrchr(exe, '/'); if (s) dir.assign(exe, (size_t)(s - exe)); }
        else if (self && strrchr(self, '/')) dir.assign(self, (size_t)(strrchr(self, '/') - self));
    }
    void *h = dlopen((dir + "/libdartindex.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "[dart index] %s\n", dlerror()); return 1; }
    auto build = (int (*)(int, const uint8_t *, uint64_t, const char *, di_log_fn, void *, uint64_t *))dlsym(h, "di_build_files");
    auto last_error = (const char *(*)(void))dlsym(h, "di_last_error");
    if (!build || !last_error) { fprintf(stderr, "[dart index] libdartindex.so lacks di_build_files\n"); return 1; }
    fprintf(stdout, "[dart index] Construct BWT, Occ and SA for %lld bases on the MI355X...\n", L); fflush(stdout);
    const int device = getenv("DART_INDEX_DEVICE") ? atoi(getenv("DART_INDEX_DEVICE")) : 0;
    uint64_t primary = 0;
    const int rc = build(device, fwd.data(), (uint64_t)L, prefix, log_line, nullptr, &primary);
    if (rc != 0) { fprintf(stderr, "[dart index] di_build_files failed (%d): %s\n", rc, last_error()); return 1; }
    fprintf(stdout, "[dart index] done: %s.{pac,ann,amb,bwt,sa}, primary = %llu\n", prefix, (unsigned long long)primary);
    return 0;
}

}  // namespace index_cmd
