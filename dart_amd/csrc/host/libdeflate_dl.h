// dart_amd/csrc/host/libdeflate_dl.h -- the system's libdeflate, if it is there (no build dependency: dlopen; zlib is what runs otherwise).
// It inflates a whole gzip member in one call ~3.3x as fast as zlib (fast_fastq.h: .gz FASTQ inflated whole) and deflates a BGZF block ~2x as fast at the
// same level (bam_writer.h; htslib is commonly built on libdeflate for the same reason).  DART_NO_LIBDEFLATE=1 keeps it out altogether.
#pragma once
#include <dlfcn.h>
#include <cstdlib>
#include <cstddef>
#include <initializer_list>

struct LibDeflate {
    void *h = nullptr;
    void *(*alloc)() = nullptr; void (*release)(void *) = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void *(*alloc_c)(int) = nullptr; void (*release_c)(void *) = nullptr;
    size_t (*deflate)(void *, const void *, size_t, void *, size_t) = nullptr;
    LibDeflate() {
        if (getenv("DART_NO_LIBDEFLATE")) return;
        for (const char *nm : {"libdeflate.so.0", "libdeflate.so"}) if ((h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor"); release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        alloc_c = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor"); release_c = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        deflate = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        if (!alloc || !release || !gunzip) alloc = nullptr;
        if (!alloc_c || !release_c || !deflate) alloc_c = nullptr;
    }
    bool ok() const { return alloc != nullptr; }              // whole-member inflate
    bool ok_deflate() const { return alloc_c != nullptr; }    // raw deflate of a block
};
static inline const LibDeflate &lib_deflate() { static LibDeflate l; return l; }
