// natac_writer.hpp -- native track writer (host C++17, multi-threaded): the reference's Track.write_track
// (pyatac/tracks.py:37-74) + the bgzip step of run_occ.py:130-136 / run_nuc.py:189-200.
//
// Text format per run of equal values:  chrom \t start \t end \t value \n  with python-2 `str(float)` formatting
// (12 significant digits, ".0" appended to integral values), NaN runs skipped, zero runs skipped when !write_zero.
// Output is plain text or BGZF (blocked gzip, <= 64 KiB members with the 'BC' extra field -- what bgzip / tabix read).
// The reference's writer processes manage 0.31 Mbp/s each; the GPU path produces tracks three orders
// of magnitude faster, so the formatter is the first "next" row of SURVEY.md section 8(f).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace natac_writer {

inline char *fmt_py2_float(char *p, double v) {
    if (std::isinf(v)) {
        const char *s = v > 0 ? "inf" : "-inf";
        size_t n = std::strlen(s);
        std::memcpy(p, s, n);
        return p + n;
    }
    char *b = p;
    auto r = std::to_chars(p, p + 40, v, std::chars_format::general, 12);
    p = r.ptr;
    bool plain = true;  // python-2 str(): digits only -> append ".0"
    for (char *q = b; q < p; ++q)
        if (!((*q >= '0' && *q <= '9') || *q == '-')) { plain = false; break; }
    if (plain) { *p++ = '.'; *p++ = '0'; }
    return p;
}

inline char *fmt_i64(char *p, long long v) {
    auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}

// run-length text of one chunk appended to `out`
inline void format_chunk(std::string &out, const char *chrom, size_t chrom_len, long long start, const double *vals, long long n,
                         bool write_zero) {
    char line[160];
    long long a = 0;
    while (a < n) {
        const double v = vals[a];
        long long b = a + 1;
        if (v != v) {
            while (b < n && vals[b] != vals[b]) ++b;     // NaN run: skipped
            a = b;
            continue;
        }
        while (b < n && vals[b] == v) ++b;
        if (v != 0.0 || write_zero) {
            char *p = line;
            std::memcpy(p, chrom, chrom_len);
            p += chrom_len;
            *p++ = '\t';
            p = fmt_i64(p, start + a);
            *p++ = '\t';
            p = fmt_i64(p, start + b);
            *p++ = '\t';
            p = fmt_py2_float(p, v);
            *p++ = '\n';
            out.append(line, (size_t)(p - line));
        }
        a = b;
    }
}

// one BGZF member for <= 0xff00 input bytes (SAM spec section 4.1)
inline bool bgzf_block(std::string &out, const unsigned char *src, size_t len, int level) {
    unsigned char buf[65536 + 64];
    static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    std::memcpy(buf, hdr, 16);
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(src);
    zs.avail_in = (uInt)len;
    zs.next_out = buf + 18;
    zs.avail_out = 65536 - 18 - 8;
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    const size_t total = 18 + clen + 8;
    buf[16] = (unsigned char)((total - 1) & 0xff);
    buf[17] = (unsigned char)(((total - 1) >> 8) & 0xff);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)len);
    const uint32_t isz = (uint32_t)len;
    for (int i = 0; i < 4; ++i) {
        buf[18 + clen + i] = (unsigned char)((crc >> (8 * i)) & 0xff);
        buf[22 + clen + i] = (unsigned char)((isz >> (8 * i)) & 0xff);
    }
    out.append((const char *)buf, total);
    return true;
}

inline bool bgzf_compress(std::string &out, const std::string &text, int level) {
    const size_t BLK = 0xff00;
    for (size_t o = 0; o < text.size(); o += BLK)
        if (!bgzf_block(out, (const unsigned char *)text.data() + o, std::min(BLK, text.size() - o), level)) return false;
    return true;
}

static const unsigned char BGZF_EOF[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// returns 0 ok, 1 cannot open, 2 write error, 3 deflate error
inline int write_bedgraph(const char *path, bool append, int compress, bool finish, int nc, const char *const *chroms,
                          const int64_t *chunk_start, const int64_t *out_off, const double *vals, bool write_zero, int n_threads,
                          int64_t *bytes_written) {
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 128u);
    n_threads = std::max(1, std::min(n_threads, std::max(1, nc)));
    // contiguous chunk ranges with ~equal numbers of bases
    std::vector<int> cut(n_threads + 1, nc);
    cut[0] = 0;
    const long long total = nc > 0 ? out_off[nc] - out_off[0] : 0;
    for (int t = 1, i = 0; t < n_threads; ++t) {
        const long long target = out_off[0] + total * t / n_threads;
        while (i < nc && out_off[i] < target) ++i;
        cut[t] = i;
    }
    std::vector<std::string> parts(n_threads);
    std::vector<int> err(n_threads, 0);
    auto work = [&](int t) {
        std::string text;
        text.reserve((size_t)1 << 20);
        std::string &dst = compress ? parts[t] : text;
        for (int i = cut[t]; i < cut[t + 1]; ++i) {
            format_chunk(text, chroms[i], std::strlen(chroms[i]), chunk_start[i], vals + out_off[i], out_off[i + 1] - out_off[i],
                         write_zero);
            if (compress && text.size() >= ((size_t)4 << 20)) {   // bound memory: flush whole 0xff00-byte blocks
                const size_t whole = text.size() / 0xff00 * 0xff00;
                if (!bgzf_compress(dst, text.substr(0, whole), compress)) { err[t] = 3; return; }
                text.erase(0, whole);
            }
        }
        if (compress) {
            if (!bgzf_compress(dst, text, compress)) err[t] = 3;
        } else {
            parts[t].swap(text);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (int e : err) if (e) return e;
    FILE *f = std::fopen(path, append ? "ab" : "wb");
    if (!f) return 1;
    int64_t nb = 0;
    for (auto &p : parts) {
        if (!p.empty() && std::fwrite(p.data(), 1, p.size(), f) != p.size()) { std::fclose(f); return 2; }
        nb += (int64_t)p.size();
    }
    if (compress && finish) {
        if (std::fwrite(BGZF_EOF, 1, 28, f) != 28) { std::fclose(f); return 2; }
        nb += 28;
    }
    if (std::fclose(f) != 0) return 2;
    if (bytes_written) *bytes_written = nb;
    return 0;
}

}  // namespace natac_writer
