// natac_writer.hpp -- native track writer (host C++17, multi-threaded): the reference's Track.write_track
// (pyatac/tracks.py:37-74) + the bgzip step of run_occ.py:130-136 / run_nuc.py:189-200.
//
// Text format per run of equal values:  chrom \t start \t end \t value \n  with python-2 `str(float)` formatting
// (12 significant digits, ".0" appended to integral values), NaN runs skipped, zero runs skipped when !write_zero,
// value runs directly followed by a NaN dropped like the reference does.
// Output is plain text or BGZF (blocked gzip, <= 64 KiB members with the 'BC' extra field -- what bgzip / tabix read).
// The reference's writer processes manage 0.31 Mbp/s each; the GPU path produces tracks three orders
// of magnitude faster, so the formatter is the first "next" row of SURVEY.md section 8(f).
#pragma once
#include "natac_cores.hpp"
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace natac_writer {

inline char *fmt_py2_float_slow(char *p, double v) {
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

// python-2 str(float) == '%.12g' (+ ".0" for integral text).  Fast exact path for 1e-4 <= |v| < 1e12 (fixed notation in
// %.12g): |v| = m 2^e exactly, so |v| 10^s = (m 5^s) 2^(e+s) with a 128-bit integer numerator; the 12 significant digits are
// that quotient rounded half-to-even on the exact remainder -- the same correctly rounded result as printf / to_chars, at a
// tenth of the cost.  Everything else (0, tiny, huge, inf) takes the to_chars path.
inline char *fmt_py2_float(char *p, double v) {
    const double a = std::fabs(v);
    if (!(a >= 1e-4 && a < 1e12)) return fmt_py2_float_slow(p, v);
    static const double P10[17] = {1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12};
    static const uint64_t I5[17] = {1ull, 5ull, 25ull, 125ull, 625ull, 3125ull, 15625ull, 78125ull, 390625ull, 1953125ull, 9765625ull,
                                    48828125ull, 244140625ull, 1220703125ull, 6103515625ull, 30517578125ull, 152587890625ull};
    int e10 = -4;
    while (e10 < 11 && a >= P10[e10 + 5]) ++e10;           // provisional floor(log10 a): the inexact 1e-4..1e-1 are fixed below
    int ex;
    const double fr = std::frexp(a, &ex);                   // a = fr 2^ex, fr in [0.5, 1)
    const uint64_t m = (uint64_t)std::ldexp(fr, 53);        // 53-bit integer mantissa
    const int e2 = ex - 53;
    uint64_t D;
    for (;;) {
        const int sft = 11 - e10;                           // 0 .. 15 (16 after a downward correction)
        const unsigned __int128 N = (unsigned __int128)m * I5[sft];
        const int k = -(e2 + sft);                          // a 10^sft = N / 2^k, k > 0 in this range
        const unsigned __int128 q = N >> k, rem = N - (q << k), half = (unsigned __int128)1 << (k - 1);
        D = (uint64_t)q;
        if (rem > half || (rem == half && (D & 1))) ++D;
        if (D < 100000000000ull) { --e10; continue; }       // a was just below the (inexact) table entry
        if (D >= 1000000000000ull) {                         // 999999999999.5+ rounded up to 10^12 (exactly divisible)
            if (e10 >= 11) return fmt_py2_float_slow(p, v);   // ... and 1e+12 is written in scientific notation
            D /= 10;
            ++e10;
        }
        break;
    }
    char dg[12];
    for (int i = 11; i >= 0; --i) { dg[i] = (char)('0' + D % 10); D /= 10; }
    int nd = 12;
    while (nd > 1 && dg[nd - 1] == '0') --nd;               // %g strips trailing zeros
    if (std::signbit(v)) *p++ = '-';
    if (e10 >= 0) {
        const int ni = e10 + 1;                             // integer digits
        for (int i = 0; i < ni; ++i) *p++ = (i < nd) ? dg[i] : '0';
        *p++ = '.';
        if (nd > ni) { for (int i = ni; i < nd; ++i) *p++ = dg[i]; }
        else *p++ = '0';                                    // integral text gets ".0"
    } else {
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -e10 - 1; ++i) *p++ = '0';
        for (int i = 0; i < nd; ++i) *p++ = dg[i];
    }
    return p;
}

inline char *fmt_i64(char *p, long long v) {
    auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}

// run-length text of one chunk appended to `out`
// Reference rule kept on purpose (pyatac/tracks.py:56-66): `prev_value` is overwritten by a NaN BEFORE the open run is
// flushed, so a run of values that is immediately followed by a NaN is never written (for [1, 1, nan, 2, 2] only the last
// run appears).  keep_before_nan = true writes those runs too (deviation, off by default).
inline void format_chunk(std::string &out, const char *chrom, size_t chrom_len, long long start, const double *vals, long long n,
                         bool write_zero, bool keep_before_nan = false) {
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
        const bool dropped = !keep_before_nan && b < n && vals[b] != vals[b];
        if ((v != 0.0 || write_zero) && !dropped) {
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

// one BGZF member for <= 0xff00 input bytes (SAM spec section 4.1).  A Deflater keeps its z_stream (deflateReset per member:
// no 268-KB allocate + clear per 64-KB block).
struct Deflater {
    z_stream zs;
    bool ready = false;
    int level = 0;
    ~Deflater() { if (ready) deflateEnd(&zs); }
    bool init(int lvl) {
        if (ready && lvl == level) return deflateReset(&zs) == Z_OK;
        if (ready) deflateEnd(&zs);
        std::memset(&zs, 0, sizeof zs);
        ready = deflateInit2(&zs, lvl, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
        level = lvl;
        return ready;
    }
};

inline bool bgzf_block(std::string &out, const unsigned char *src, size_t len, int level, Deflater *df = nullptr) {
    unsigned char buf[65536 + 64];
    static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    std::memcpy(buf, hdr, 16);
    Deflater local;
    if (!df) df = &local;
    if (!df->init(level)) return false;
    z_stream &zs = df->zs;
    zs.next_in = const_cast<unsigned char *>(src);
    zs.avail_in = (uInt)len;
    zs.next_out = buf + 18;
    zs.avail_out = 65536 - 18 - 8;
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
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

inline bool bgzf_compress(std::string &out, const char *text, size_t n, int level, Deflater *df = nullptr) {
    const size_t BLK = 0xff00;
    Deflater local;
    if (!df) df = &local;
    for (size_t o = 0; o < n; o += BLK)
        if (!bgzf_block(out, (const unsigned char *)text + o, std::min(BLK, n - o), level, df)) return false;
    return true;
}
inline bool bgzf_compress(std::string &out, const std::string &text, int level) { return bgzf_compress(out, text.data(), text.size(), level); }

static const unsigned char BGZF_EOF[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// returns 0 ok, 1 cannot open, 2 write error, 3 deflate error
inline int write_bedgraph(const char *path, bool append, int compress, bool finish, int nc, const char *const *chroms,
                          const int64_t *chunk_start, const int64_t *out_off, const double *vals, bool write_zero, int n_threads,
                          int64_t *bytes_written, bool keep_before_nan = false) {
    if (n_threads <= 0) n_threads = natac_cores::default_threads(128);
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
    const bool dbg = std::getenv("NATAC_WRITER_DEBUG") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<double> t_fmt(n_threads, 0.0), t_def(n_threads, 0.0);
    std::vector<std::string> parts(n_threads);
    std::vector<int> err(n_threads, 0);
    auto work = [&](int t) {
        const auto w0 = std::chrono::steady_clock::now();
        const long long bases = (cut[t] < nc ? out_off[cut[t + 1]] - out_off[cut[t]] : 0);
        std::string text;
        Deflater df;
        const size_t FLUSH = (size_t)4 << 20;
        if (compress) {
            text.reserve(FLUSH + ((size_t)1 << 16));
            parts[t].reserve((size_t)bases * 14 + 4096);     // ~13 bytes of BGZF per base for float tracks
        } else {
            parts[t].reserve((size_t)bases * 40 + 4096);
        }
        std::string &fmt = compress ? text : parts[t];
        for (int i = cut[t]; i < cut[t + 1]; ++i) {
            format_chunk(fmt, chroms[i], std::strlen(chroms[i]), chunk_start[i], vals + out_off[i], out_off[i + 1] - out_off[i],
                         write_zero, keep_before_nan);
            if (compress && text.size() >= FLUSH) {          // bound memory: flush whole 0xff00-byte blocks
                const size_t whole = text.size() / 0xff00 * 0xff00;
                if (!bgzf_compress(parts[t], text.data(), whole, compress, &df)) { err[t] = 3; return; }
                text.erase(0, whole);
            }
        }
        if (compress && !bgzf_compress(parts[t], text.data(), text.size(), compress, &df)) err[t] = 3;
        t_fmt[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (int e : err) if (e) return e;
    const auto t_mid = std::chrono::steady_clock::now();
    // every part lands at its own offset: parallel pwrite (a serial fwrite of GBs of text is the bottleneck otherwise)
    const int fd = ::open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0644);
    if (fd < 0) return 1;
    int64_t base = 0;
    if (append) {
        const off_t e = ::lseek(fd, 0, SEEK_END);
        if (e < 0) { ::close(fd); return 2; }
        base = (int64_t)e;
    }
    std::vector<int64_t> at(n_threads + 1, base);
    for (int t = 0; t < n_threads; ++t) at[t + 1] = at[t] + (int64_t)parts[t].size();
    std::vector<int> werr(n_threads, 0);
    auto put = [&](int t) {
        const char *p = parts[t].data();
        size_t left = parts[t].size();
        int64_t o = at[t];
        while (left) {
            const ssize_t w = ::pwrite(fd, p, left, (off_t)o);
            if (w <= 0) { werr[t] = 1; return; }
            p += w; left -= (size_t)w; o += w;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) if (!parts[t].empty()) th.emplace_back(put, t);
        put(0);
        for (auto &x : th) x.join();
    }
    int64_t nb = at[n_threads] - base;
    bool ok = true;
    for (int e : werr) ok = ok && !e;
    if (ok && compress && finish) {
        ok = ::pwrite(fd, BGZF_EOF, 28, (off_t)at[n_threads]) == 28;
        nb += 28;
    }
    if (::close(fd) != 0 || !ok) return 2;
    if (dbg) {
        double mx = 0, sm = 0;
        for (double x : t_fmt) { mx = std::max(mx, x); sm += x; }
        std::fprintf(stderr, "[natac writer] threads %d: parallel section %.3f s (worker max %.3f avg %.3f), write %.3f s\n", n_threads,
                     std::chrono::duration<double>(t_mid - t_begin).count(), mx, sm / n_threads,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_mid).count());
    }
    if (bytes_written) *bytes_written = nb;
    return 0;
}

// BED-like rows with python-2 float columns: chrom \t start \t end (\t value)* \n -- what OccPeak.asBed / Nucleosome.asBed
// produce (nucleoatac/Occupancy.py:166-171, NucleosomeCalling.py:195-199), formatted natively for millions of rows.
// returns 0 ok, 1 cannot open, 2 write error
// label_id / labels (may be null): one more text column at the end of every row (the 'occ' / 'nuc' source of the combined map)
inline int write_bed_rows(const char *path, bool append, int64_t n_rows, const int32_t *chrom_id, const char *const *names,
                          const int64_t *start, const int64_t *end, const double *vals, int n_cols, const int32_t *label_id = nullptr,
                          const char *const *labels = nullptr) {
    std::string out;
    out.reserve((size_t)n_rows * (32 + (size_t)n_cols * 16) + 64);
    char line[64 + 40 * 32];
    if (n_cols > 32) return 2;
    for (int64_t r = 0; r < n_rows; ++r) {
        char *p = line;
        const char *nm = names[chrom_id[r]];
        const size_t nl = std::strlen(nm);
        if (nl > 256) { out.append(line, (size_t)(p - line)); out.append(nm, nl); p = line; }
        else { std::memcpy(p, nm, nl); p += nl; }
        *p++ = '\t';
        p = fmt_i64(p, start[r]);
        *p++ = '\t';
        p = fmt_i64(p, end[r]);
        for (int c = 0; c < n_cols; ++c) {
            *p++ = '\t';
            const double v = vals[(size_t)r * n_cols + c];
            if (v != v) { *p++ = 'n'; *p++ = 'a'; *p++ = 'n'; }
            else p = fmt_py2_float(p, v);
        }
        if (label_id) {
            *p++ = '\t';
            out.append(line, (size_t)(p - line));
            out.append(labels[label_id[r]]);
            p = line;
        }
        *p++ = '\n';
        out.append(line, (size_t)(p - line));
    }
    FILE *f = std::fopen(path, append ? "ab" : "wb");
    if (!f) return 1;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0 && ok) ? 0 : 2;
}

}  // namespace natac_writer
