// dart_amd/csrc/host/bam_writer.h -- `-bo`: the alignment output as BAM.
//
// The reference writes BAM by handing every finished SAM line to htslib: sam_parse1() + sam_write1() on a BGZF stream opened with
// sam_open_format(name, "wb") after sam_hdr_write() of the "@PG / @SQ" text (Mapping.cpp:41-48,655-662,739-755).  This file restates
// that pair from the SAM/BAM specification for the lines this program produces (eleven mandatory fields + NM/AS/XS integer tags):
//   record  = refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen, read_name\0, cigar, 4-bit seq,
//             qual - 33 (0xff when '*'), tags with the smallest integer type that holds the value (htslib's choice);
//   bin     = reg2bin(pos, pos + reference length of the CIGAR, or 1)            [unmapped: pos = -1 -> 4680]
//   an integer tag is read like strtol reads it: "XS:i:7 XS:A:+" (the reference joins its strand tag with a blank, not a tab) gives
//   XS:i:7 and nothing else, exactly as the reference's BAM loses the strand tag;
//   a line htslib would refuse (quality and sequence of different lengths, an unknown reference name) is dropped, as there;
//   BGZF    = blocks of at most 0xff00 input bytes, raw deflate at zlib's default level (by the system's libdeflate when it is there: libdeflate_dl.h), 'BC' extra field, CRC32, ISIZE; the header has
//             its own block(s); the 28-byte end-of-file block closes the file.  Blocks are compressed by `threads` workers, written in order.
// htslib itself cannot be built here (its Makefile generates version.h / config.h), so the COMPRESSED bytes are not pinned against the
// reference's; the decoded content is: tests/test_host_text.py decodes the BAM and compares it with the reference-generated golden SAM.
#pragma once
#include <zlib.h>
#include "libdeflate_dl.h"
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <thread>
#include <unordered_map>

class BamWriter {
public:
    bool open(const char *path, const std::string &header_text, const std::vector<std::string> &names, const std::vector<int64_t> &lens, int threads)
    {
        f_ = fopen(path, "wb");
        if (!f_) return false;
        threads_ = threads < 1 ? 1 : threads;
        for (size_t i = 0; i < names.size(); i++) tid_.emplace(names[i], (int32_t)i);
        std::vector<uint8_t> h;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put32(h, (uint32_t)header_text.size());
        h.insert(h.end(), header_text.begin(), header_text.end());
        put32(h, (uint32_t)names.size());
        for (size_t i = 0; i < names.size(); i++) {
            put32(h, (uint32_t)names[i].size() + 1);
            h.insert(h.end(), names[i].begin(), names[i].end()); h.push_back(0);
            put32(h, (uint32_t)lens[i]);
        }
        append(h.data(), h.size());
        flush_all();                                           // (bam_hdr_write ends with bgzf_flush: the header never shares a block with records)
        return !bad_;
    }
    // whole SAM lines ("...\n"), header lines ('@') skipped
    void add_sam_text(const char *p, size_t n)
    {
        const char *end = p + n;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            if (le > p && *p != '@') {
                rec_.clear();
                if (sam_line_to_bam(p, (size_t)(le - p), rec_)) { append(rec_.data(), rec_.size()); n_records_++; } else n_refused_++;
            }
            p = nl ? nl + 1 : end;
        }
        if (pending_.size() >= (size_t)BLOCK * 64) compress_full_blocks();
    }
    bool close()
    {
        if (!f_) return false;
        flush_all();
        static const uint8_t eof_block[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        if (fwrite(eof_block, 1, 28, f_) != 28) bad_ = true;
        if (fclose(f_) != 0) bad_ = true;
        f_ = nullptr;
        return !bad_;
    }
    // several pieces of SAM text (each whole lines) that follow one another in the output: converted side by side, appended in order
    void add_sam_chunks(const std::vector<std::pair<const char *, size_t>> &chunks)
    {
        const size_t n = chunks.size();
        std::vector<std::vector<uint8_t>> recs(n);
        std::vector<long long> good(n, 0), refused(n, 0);
        const int T = (int)std::min<size_t>((size_t)threads_, n);
        auto work = [&](int t) {
            std::vector<uint8_t> one;
            for (size_t c = (size_t)t; c < n; c += (size_t)T) {
                const char *p = chunks[c].first, *end = p + chunks[c].second;
                // (a chunk's records and counts are gathered in locals: recs[c] / good[c] of neighbouring chunks share cache lines and belong to other threads)
                std::vector<uint8_t> acc; acc.reserve(chunks[c].second);
                long long g = 0, rf = 0;
                while (p < end) {
                    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                    const char *le = nl ? nl : end;
                    if (le > p && *p != '@') {
                        one.clear();
                        if (sam_line_to_bam(p, (size_t)(le - p), one)) { acc.insert(acc.end(), one.begin(), one.end()); g++; } else rf++;
                    }
                    p = nl ? nl + 1 : end;
                }
                recs[c] = std::move(acc); good[c] = g; refused[c] = rf;
            }
        };
        if (T <= 1) work(0);
        else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
        for (size_t c = 0; c < n; c++) { append(recs[c].data(), recs[c].size()); n_records_ += good[c]; n_refused_ += refused[c]; }
        compress_full_blocks();
    }
    long long records() const { return n_records_; }
    long long refused() const { return n_refused_; }

    // one SAM line (no newline) -> one BAM record incl. its block_size; false = htslib's sam_parse1 would have failed
    bool sam_line_to_bam(const char *s, size_t len, std::vector<uint8_t> &out) const
    {
        const char *fld[12]; size_t fl[12];
        size_t nf = 0; const char *p = s, *end = s + len;
        while (nf < 11) {
            const char *t = (const char *)memchr(p, '\t', (size_t)(end - p));
            fld[nf] = p; fl[nf] = (size_t)((t ? t : end) - p); nf++;
            if (!t) { p = end; break; }
            p = t + 1;
        }
        if (nf < 11) return false;
        const char *aux = p;                                  // tags, tab separated (may be empty)
        if (fl[0] == 0 || fl[0] > 254) return false;
        const int32_t flag0 = (int32_t)strtol(fld[1], nullptr, 10);
        int32_t refid = -1;
        if (!(fl[2] == 1 && fld[2][0] == '*')) { auto it = tid_.find(std::string(fld[2], fl[2])); if (it == tid_.end()) return false; refid = it->second; }
        const int64_t pos = strtoll(fld[3], nullptr, 10) - 1;
        const int32_t mapq = (int32_t)strtol(fld[4], nullptr, 10);
        std::vector<uint32_t> cig;
        int64_t rlen = 0;
        if (!(fl[5] == 1 && fld[5][0] == '*')) {
            const char *q = fld[5], *qe = fld[5] + fl[5];
            while (q < qe) {
                char *e2; const unsigned long l = strtoul(q, &e2, 10);
                if (e2 == q || e2 >= qe) return false;
                const char *ops = "MIDNSHP=X"; const char *o = strchr(ops, *e2);
                if (!o) return false;
                const uint32_t op = (uint32_t)(o - ops);
                cig.push_back((uint32_t)(l << 4) | op);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (int64_t)l;
                q = e2 + 1;
            }
        }
        int32_t flag = flag0;
        if (cig.empty() && !(flag & 4)) flag |= 4;            // (sam_parse1: a mapped read without CIGAR is marked unmapped)
        int32_t mrefid = -1;
        if (fl[6] == 1 && fld[6][0] == '=') mrefid = refid;
        else if (!(fl[6] == 1 && fld[6][0] == '*')) { auto it = tid_.find(std::string(fld[6], fl[6])); if (it == tid_.end()) return false; mrefid = it->second; }
        const int64_t mpos = strtoll(fld[7], nullptr, 10) - 1;
        const int64_t tlen = strtoll(fld[8], nullptr, 10);
        const bool no_seq = fl[9] == 1 && fld[9][0] == '*';
        const size_t l_seq = no_seq ? 0 : fl[9];
        const bool no_qual = fl[10] == 1 && fld[10][0] == '*';
        if (!no_qual && fl[10] != l_seq) return false;        // "SEQ and QUAL are of different length"
        const size_t start = out.size();
        put32(out, 0);                                        // block_size, filled in below
        put32(out, (uint32_t)refid); put32(out, (uint32_t)(int32_t)pos);
        const uint32_t bin = reg2bin(pos, pos + (rlen > 0 ? rlen : 1));
        put32(out, (bin << 16) | ((uint32_t)(mapq & 0xff) << 8) | (uint32_t)(fl[0] + 1));
        put32(out, ((uint32_t)flag << 16) | (uint32_t)(cig.size() & 0xffff));
        put32(out, (uint32_t)l_seq); put32(out, (uint32_t)mrefid); put32(out, (uint32_t)(int32_t)mpos); put32(out, (uint32_t)(int32_t)tlen);
        out.insert(out.end(), fld[0], fld[0] + fl[0]); out.push_back(0);
        for (uint32_t c : cig) put32(out, c);
        static const char *nt16 = "=ACMGRSVTWYHKDBN";
        for (size_t i = 0; i < l_seq; i += 2) {
            auto code = [&](char ch) -> uint8_t { const char u = (char)(ch >= 'a' && ch <= 'z' ? ch - 32 : ch); const char *o = strchr(nt16, u); return (uint8_t)(o && u ? (o - nt16) : 15); };
            const uint8_t hi = code(fld[9][i]), lo = i + 1 < l_seq ? code(fld[9][i + 1]) : 0;
            out.push_back((uint8_t)(hi << 4 | lo));
        }
        if (no_qual) out.insert(out.end(), l_seq, 0xff);
        else for (size_t i = 0; i < l_seq; i++) out.push_back((uint8_t)(fld[10][i] - 33));
        // tags
        while (aux < end) {
            const char *t = (const char *)memchr(aux, '\t', (size_t)(end - aux));
            const char *te = t ? t : end;
            const size_t tl = (size_t)(te - aux);
            if (tl < 5 || aux[2] != ':' || aux[4] != ':') return false;
            out.push_back((uint8_t)aux[0]); out.push_back((uint8_t)aux[1]);
            const char ty = aux[3]; const char *v = aux + 5;
            if (ty == 'A' || ty == 'a' || ty == 'c' || ty == 'C') { out.push_back('A'); out.push_back((uint8_t)*v); }
            else if (ty == 'i' || ty == 'I') {
                const std::string tok(v, (size_t)(te - v));
                if (!tok.empty() && tok[0] == '-') {
                    const long x = strtol(tok.c_str(), nullptr, 10);
                    if (x >= INT8_MIN) { out.push_back('c'); out.push_back((uint8_t)(int8_t)x); }
                    else if (x >= INT16_MIN) { out.push_back('s'); put16(out, (uint16_t)(int16_t)x); }
                    else { out.push_back('i'); put32(out, (uint32_t)(int32_t)x); }
                } else {
                    const unsigned long x = strtoul(tok.c_str(), nullptr, 10);
                    if (x <= UINT8_MAX) { out.push_back('C'); out.push_back((uint8_t)x); }
                    else if (x <= UINT16_MAX) { out.push_back('S'); put16(out, (uint16_t)x); }
                    else { out.push_back('I'); put32(out, (uint32_t)x); }
                }
            } else if (ty == 'Z' || ty == 'H') { out.push_back((uint8_t)ty); out.insert(out.end(), v, te); out.push_back(0); }
            else if (ty == 'f') { out.push_back('f'); const float x = strtof(std::string(v, (size_t)(te - v)).c_str(), nullptr); uint32_t u; memcpy(&u, &x, 4); put32(out, u); }
            else return false;
            aux = t ? t + 1 : end;
        }
        const uint32_t bs = (uint32_t)(out.size() - start - 4);
        out[start] = (uint8_t)bs; out[start + 1] = (uint8_t)(bs >> 8); out[start + 2] = (uint8_t)(bs >> 16); out[start + 3] = (uint8_t)(bs >> 24);
        return true;
    }

    static uint32_t reg2bin(int64_t beg, int64_t end)         // SAM specification 5.3 (hts_reg2bin with min_shift 14, 5 levels)
    {
        --end;
        if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
        if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
        if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
        if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
        if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
        return 0;
    }

private:
    static constexpr size_t BLOCK = 0xff00;                    // BGZF_BLOCK_SIZE: input bytes per block
    FILE *f_ = nullptr;
    int threads_ = 1;
    bool bad_ = false;
    long long n_records_ = 0, n_refused_ = 0;
    std::unordered_map<std::string, int32_t> tid_;
    std::vector<uint8_t> pending_;
    mutable std::vector<uint8_t> rec_;

    static void put16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
    static void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 24)); }
    void append(const uint8_t *p, size_t n) { pending_.insert(pending_.end(), p, p + n); }

    static bool compress_block(const uint8_t *src, size_t n, std::vector<uint8_t> &dst)
    {
        dst.resize(18 + compressBound((uLong)n) + 8);
        size_t clen = 0;
        const LibDeflate &ld = lib_deflate();
        if (ld.ok_deflate()) {                        // level 6 = zlib's default; one compressor per worker thread
            struct Own { void *c = nullptr; const LibDeflate *l = nullptr; ~Own() { if (c) l->release_c(c); } };
            static thread_local Own own;
            if (!own.c) { own.c = ld.alloc_c(6); own.l = &ld; }
            if (own.c) clen = ld.deflate(own.c, src, n, dst.data() + 18, dst.size() - 18 - 8);       // 0: did not fit (cannot happen at compressBound) -> zlib below
        }
        if (clen == 0) {
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
            zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n;
            zs.next_out = dst.data() + 18; zs.avail_out = (uInt)(dst.size() - 18 - 8);
            const int rc = deflate(&zs, Z_FINISH);
            clen = zs.total_out;
            deflateEnd(&zs);
            if (rc != Z_STREAM_END) return false;
        }
        if (18 + clen + 8 > 65536) return false;
        static const uint8_t hd[16] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0 };
        memcpy(dst.data(), hd, 16);
        const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
        dst[16] = (uint8_t)bsize; dst[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
        uint8_t *t = dst.data() + 18 + clen;
        for (int k = 0; k < 4; k++) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
        dst.resize(18 + clen + 8);
        return true;
    }
    // compresses the first n_blocks full blocks (or, with `tail`, everything) of pending_ and writes them in order
    void compress_range(size_t n_bytes)
    {
        const size_t nb = (n_bytes + BLOCK - 1) / BLOCK;
        std::vector<std::vector<uint8_t>> outb(nb);
        std::vector<char> ok(nb, 1);
        const int T = (int)std::min<size_t>((size_t)threads_, nb);
        auto work = [&](int t) { for (size_t b = (size_t)t; b < nb; b += (size_t)T) { const size_t o = b * BLOCK, l = std::min(BLOCK, n_bytes - o); ok[b] = compress_block(pending_.data() + o, l, outb[b]); } };
        if (T <= 1) work(0);
        else { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
        for (size_t b = 0; b < nb; b++) { if (!ok[b] || fwrite(outb[b].data(), 1, outb[b].size(), f_) != outb[b].size()) bad_ = true; }
        pending_.erase(pending_.begin(), pending_.begin() + (long)n_bytes);
    }
    void compress_full_blocks() { const size_t full = pending_.size() / BLOCK * BLOCK; if (full) compress_range(full); }
    void flush_all() { if (!pending_.empty()) compress_range(pending_.size()); }
};
