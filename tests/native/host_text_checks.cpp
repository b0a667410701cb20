// CPU checks of the host program's parallel FASTQ pipeline pieces (dart_amd/csrc/host/fast_fastq.h), no GPU involved:
//  * the reversed / reverse-complement / double-complement copies (AVX2 and scalar) and the integer printer against their definitions
//  * index_fastq (newline counting + per-share record walk, 1..9 threads, shares of a few bytes) against a sequential line splitter on
//    random FASTQ-like text: empty lines, empty records, lines starting with '@' or '+', with and without a final newline
// Test infrastructure: tests/test_host_text.py builds it with g++ and runs it.  The library entry points the header refers to are
// stubs here (the pipeline itself is exercised on the GPU box by tests/test_gpu_cli.py).
#include "dartgpu.h"
#include <sys/stat.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
extern "C" void *dg_host_alloc(size_t n) { return malloc(n); }
extern "C" void dg_host_free(void *p) { free(p); }
extern "C" int dg_set_params(dg_ctx *, const dg_params *) { return 0; }
extern "C" int dg_map_batch(dg_ctx *, int, const uint32_t *, const uint16_t *, const char *, dg_read_out *, dg_report_out *, uint32_t *, dg_sj_out *, const size_t *, size_t *) { return 0; }
extern "C" const char *dg_last_error(const dg_ctx *) { return ""; }
#include "fast_fastq.h"

static unsigned long long rs = 88172645463325252ull;
static unsigned long long rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

// `host_text_checks gz FILE.gz PLAIN`: the whole-file .gz path (MappedFile::open_gz + FastqIndex::run(gz)): prints whether the library qualifies for the
// parallel pipeline and, if so, whether the inflated bytes are PLAIN's and how many records were indexed
static int gz_mode(const char *gz, const char *plain)
{
    if (!lib_deflate().ok()) { printf("libdeflate missing\n"); return 0; }
    FastqIndex fi;
    fi.run(gz, nullptr, 3, true, (size_t)1 << 30);
    if (!fi.ok) { printf("qualifies=0\n"); return 0; }
    MappedFile pf;
    const bool same = pf.open(plain) && pf.n == fi.m1.n && (pf.n == 0 || memcmp(pf.p, fi.m1.p, pf.n) == 0);
    printf("qualifies=1 same_bytes=%d records=%zu\n", (int)same, fi.r1.size());
    return same ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc == 4 && strcmp(argv[1], "gz") == 0) return gz_mode(argv[2], argv[3]);
    long bad = 0, n_str = 0, n_files = 0, n_recs = 0;
    const char *al = "ACGTacgtNnRYxX-*@!";
    for (int it = 0; it < 100000; it++) {
        const size_t l = rnd() % 300;
        std::string s(l, 'A');
        for (size_t i = 0; i < l; i++) s[i] = (it & 1) ? (char)(rnd() & 255) : al[rnd() % 18];
        TextBuf o; o.need(4 * l + 64);
        o.put_rev(s.data(), l); o.put_revcomp(s.data(), l); o.put_comp2(s.data(), l);
        for (size_t i = 0; i < l; i++) {
            if (o.b[i] != s[l - 1 - i]) bad++;
            if (o.b[l + i] != comp_base_f(s[l - 1 - i])) bad++;
            if (o.b[2 * l + i] != comp_base_f(comp_base_f(s[i]))) bad++;
        }
        const long long v = (long long)(rnd() >> (rnd() & 63)) * ((it % 3) ? 1 : -1);
        TextBuf n2; n2.need(64); n2.num(v); char ref[32]; sprintf(ref, "%lld", v);
        if (n2.n != strlen(ref) || memcmp(n2.b, ref, n2.n)) bad++;
        n_str++;
    }
    for (int it = 0; it < 3000; it++) {
        std::string f;
        const int lines = (int)(rnd() % 60);
        for (int k = 0; k < lines; k++) {
            const int kind = (int)(rnd() % 10), len = kind == 0 ? 0 : (int)(rnd() % 40);
            for (int i = 0; i < len; i++) f += (i == 0 && kind < 4) ? "@+"[rnd() & 1] : "ACGTNI#"[rnd() % 7];
            f += '\n';
        }
        if ((it % 3) == 0 && !f.empty()) f.pop_back();                 // no newline at the end
        // the sequential definition: lines = split at '\n' (a trailing newline ends the last line); records = groups of four lines
        std::vector<FqRec> want; bool want_empty = false;
        {
            std::vector<std::pair<size_t, uint32_t>> ln;                // (start, length incl. newline)
            size_t at = 0;
            while (at < f.size()) { const size_t e = f.find('\n', at); const size_t l = e == std::string::npos ? f.size() - at : e - at + 1; ln.push_back({at, (uint32_t)l}); at += l; }
            for (size_t k = 0; k < ln.size(); k += 4) {
                FqRec r; r.off = ln[k].first; r.l0 = ln[k].second; r.l1 = k + 1 < ln.size() ? ln[k + 1].second : 0; r.l2 = k + 2 < ln.size() ? ln[k + 2].second : 0; r.l3 = k + 3 < ln.size() ? ln[k + 3].second : 0;
                if ((int)r.l1 - 1 <= 0) want_empty = true;
                want.push_back(r);
            }
        }
        MappedFile mf; mf.p = f.data(); mf.n = f.size();               // (not a mapping: the destructor must not unmap it)
        for (int nt : {1, 2, 3, 5, 9}) {
            std::vector<FqRec> got; bool e = false;
            index_fastq(mf, nt, got, &e, 7);
            bool same = got.size() == want.size() && e == want_empty;
            for (size_t k = 0; same && k < got.size(); k++) same = got[k].off == want[k].off && got[k].l0 == want[k].l0 && got[k].l1 == want[k].l1 && got[k].l2 == want[k].l2 && got[k].l3 == want[k].l3;
            if (!same) { if (bad < 5) printf("index differs: file %d, %d threads: %zu vs %zu records\n", it, nt, got.size(), want.size()); bad++; }
        }
        mf.p = nullptr; mf.n = 0;
        n_files++; n_recs += (long)want.size();
    }
    printf("strings %ld, files %ld (%ld records) x 5 thread counts, avx2=%d  bad=%ld\n", n_str, n_files, n_recs, (int)g_avx2, bad);
    return bad != 0;
}
