// natac_tabix.hpp -- tabix (.tbi) index of a BGZF-compressed, position-sorted BED / bedGraph file (host side).
//
// Replaces the reference's pysam.tabix_index(..., preset="bed") calls after every track / peak file it writes
// (pyatac/utils.py:135-141 `tabix_bedgraph`, nucleoatac/run_occ.py:136-139, run_nuc.py:204-214).  Format: SAM/tabix
// specification ("The Tabix index file format"): binning index with min_shift 14 / depth 5 + 16-kb linear index over
// BGZF virtual offsets (compressed block start << 16 | offset inside the inflated block), itself BGZF-compressed.
#pragma once
#include "natac_cores.hpp"
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "natac_writer.hpp"

namespace natac_tabix {

struct Block { uint64_t coff; uint32_t csize; uint32_t usize; uint64_t uoff; };

inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> lin;
    uint64_t off_beg = 0, off_end = 0, n_rec = 0;
};

inline void put32(std::string &o, uint32_t v) { for (int i = 0; i < 4; ++i) o.push_back((char)((v >> (8 * i)) & 0xff)); }
inline void put64(std::string &o, uint64_t v) { for (int i = 0; i < 8; ++i) o.push_back((char)((v >> (8 * i)) & 0xff)); }

// Incremental index: records (or runs of records, see push) in file order.  index_bed() feeds it from the parsed text; the
// device-side track writer feeds it runs computed on the GPU (natac_textz.hpp), without ever re-reading the file it writes.
struct Builder {
    std::vector<std::string> names;
    std::vector<RefIndex> refs;
    int cur = -1;
    uint32_t last_bin = 0xffffffffu;
    uint64_t save_off = 0, last_off = 0;
    int64_t last_beg = -1, nrec = 0;
    std::string err;

    void flush_bin() {
        if (cur >= 0 && last_bin != 0xffffffffu && last_off > save_off) refs[cur].bins[last_bin].push_back({save_off, last_off});
    }
    // `count` consecutive records of chromosome `name` that share ONE bin, spanning [beg, end) together, at virtual offsets
    // [v0, v1).  count == 1: an ordinary record.  A run of several records is equivalent to pushing them one by one when all of
    // them lie in the same 16-kb window (a leaf bin): same bin, same linear-index window, the first offset wins.
    bool push(const char *name, size_t nlen, int64_t beg, int64_t end, uint64_t v0, uint64_t v1, int64_t count) {
        if (end <= beg) end = beg + 1;
        if (cur < 0 || names[cur].size() != nlen || std::memcmp(names[cur].data(), name, nlen) != 0) {
            flush_bin();
            if (cur >= 0) refs[cur].off_end = v0;
            for (auto &n : names)
                if (n.size() == nlen && std::memcmp(n.data(), name, nlen) == 0) { err = "chromosome blocks not continuous: " + n; return false; }
            names.emplace_back(name, nlen);
            refs.emplace_back();
            cur = (int)names.size() - 1;
            refs[cur].off_beg = v0;
            last_bin = 0xffffffffu;
            last_beg = -1;
            save_off = v0;
        }
        if (beg < last_beg) { err = "unsorted positions on " + names[cur]; return false; }
        last_beg = beg;
        const uint32_t bin = (uint32_t)reg2bin(beg, end);
        if (bin != last_bin) {
            flush_bin();
            save_off = v0;
            last_bin = bin;
        }
        RefIndex &r = refs[cur];
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (r.lin.size() <= w1) r.lin.resize(w1 + 1, ~0ull);
        for (size_t w = w0; w <= w1; ++w) if (r.lin[w] == ~0ull) r.lin[w] = v0;
        r.n_rec += (uint64_t)count;
        nrec += count;
        last_off = v1;
        return true;
    }
    // returns 0 ok, 3 deflate error, 5 write error
    int write(const char *tbi_path) {
        flush_bin();
        last_bin = 0xffffffffu;
        if (cur >= 0) refs[cur].off_end = last_off;
        std::string out;
        out.append("TBI\1", 4);
        put32(out, (uint32_t)names.size());
        put32(out, 0x10000u);                 // TBX_UCSC: 0-based half-open coordinates (the "bed" preset)
        put32(out, 1); put32(out, 2); put32(out, 3);
        put32(out, (uint32_t)'#');
        put32(out, 0);
        uint32_t l_nm = 0;
        for (auto &n : names) l_nm += (uint32_t)n.size() + 1;
        put32(out, l_nm);
        for (auto &n : names) { out.append(n); out.push_back('\0'); }
        for (auto &r : refs) {
            for (size_t w = r.lin.size(); w-- > 0;)            // windows without a record inherit the next one's offset
                if (r.lin[w] == ~0ull) r.lin[w] = (w + 1 < r.lin.size()) ? r.lin[w + 1] : r.off_end;
            put32(out, (uint32_t)r.bins.size() + 1);
            for (auto &b : r.bins) {
                put32(out, b.first);
                put32(out, (uint32_t)b.second.size());
                for (auto &c : b.second) { put64(out, c.first); put64(out, c.second); }
            }
            put32(out, 37450u);                                 // htslib's pseudo-bin: file range + record counts of the reference
            put32(out, 2);
            put64(out, r.off_beg); put64(out, r.off_end);
            put64(out, r.n_rec); put64(out, 0);
            put32(out, (uint32_t)r.lin.size());
            for (uint64_t v : r.lin) put64(out, v);
        }
        put64(out, 0);                                          // n_no_coor
        std::string comp;
        if (!natac_writer::bgzf_compress(comp, out, 6)) return 3;
        comp.append((const char *)natac_writer::BGZF_EOF, 28);
        FILE *f = std::fopen(tbi_path, "wb");
        if (!f) return 5;
        const bool ok = std::fwrite(comp.data(), 1, comp.size(), f) == comp.size();
        if (std::fclose(f) != 0 || !ok) return 5;
        return 0;
    }
};

// returns 0 ok, 1 cannot open / read, 2 not BGZF, 3 inflate error, 4 unsorted or malformed record, 5 write error
inline int index_bed(const char *path, const char *tbi_path, int n_threads, int64_t *n_records, std::string *errmsg) {
    // ---- read the file and walk the BGZF members
    std::string data;
    {
        FILE *f = std::fopen(path, "rb");
        if (!f) return 1;
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        data.resize((size_t)sz);
        if (sz > 0 && std::fread(&data[0], 1, (size_t)sz, f) != (size_t)sz) { std::fclose(f); return 1; }
        std::fclose(f);
    }
    std::vector<Block> blocks;
    const unsigned char *d = (const unsigned char *)data.data();
    uint64_t p = 0, utotal = 0;
    while (p < data.size()) {
        if (p + 18 > data.size() || d[p] != 0x1f || d[p + 1] != 0x8b || d[p + 2] != 8 || !(d[p + 3] & 4)) return 2;
        const uint32_t xlen = d[p + 10] | (d[p + 11] << 8);
        uint32_t bsize = 0;
        bool found = false;
        for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = d[q + 2] | (d[q + 3] << 8);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2) { bsize = (d[q + 4] | (d[q + 5] << 8)) + 1u; found = true; }
            q += 4 + slen;
        }
        if (!found || p + bsize > data.size()) return 2;
        const uint32_t isize = d[p + bsize - 4] | (d[p + bsize - 3] << 8) | (d[p + bsize - 2] << 16) | ((uint32_t)d[p + bsize - 1] << 24);
        blocks.push_back(Block{p, bsize, isize, utotal});
        utotal += isize;
        p += bsize;
    }
    // ---- inflate all members in parallel into one text buffer
    std::string text(utotal, '\0');
    if (n_threads <= 0) n_threads = natac_cores::default_threads(64);
    n_threads = std::max(1, std::min<int>(n_threads, (int)std::max<size_t>(1, blocks.size() / 16)));
    std::vector<int> err(n_threads, 0);
    auto work = [&](int t) {
        for (size_t b = t; b < blocks.size(); b += n_threads) {
            const Block &k = blocks[b];
            if (k.usize == 0) continue;
            const uint32_t xlen = d[k.coff + 10] | (d[k.coff + 11] << 8);
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) { err[t] = 3; return; }
            zs.next_in = (Bytef *)(d + k.coff + 12 + xlen);
            zs.avail_in = k.csize - 12 - xlen - 8;
            zs.next_out = (Bytef *)&text[k.uoff];
            zs.avail_out = k.usize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { err[t] = 3; return; }
            const unsigned char *tr = d + k.coff + k.csize - 8;          // CRC-32 of the member (RFC 1952)
            const uint32_t want = tr[0] | (tr[1] << 8) | (tr[2] << 16) | ((uint32_t)tr[3] << 24);
            if ((uint32_t)crc32(0L, (const Bytef *)&text[k.uoff], k.usize) != want) { err[t] = 3; return; }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (int e : err) if (e) return e;
    }
    // virtual offset of an uncompressed position (a position at a member's end belongs to the start of the next member)
    size_t bi = 0;
    auto voff = [&](uint64_t u) -> uint64_t {
        while (bi + 1 < blocks.size() && u >= blocks[bi].uoff + blocks[bi].usize) ++bi;
        return (blocks[bi].coff << 16) | (u - blocks[bi].uoff);
    };
    // ---- records: chrom \t beg \t end ...   (0-based half-open: TBX_UCSC)
    Builder bld;
    uint64_t u = 0;
    while (u < utotal) {
        const char *ls = text.data() + u;
        const char *nl = (const char *)std::memchr(ls, '\n', utotal - u);
        const uint64_t len = nl ? (uint64_t)(nl - ls) + 1 : utotal - u;
        if (len > 1 && ls[0] != '#') {
            const char *t1 = (const char *)std::memchr(ls, '\t', len);
            const char *t2 = t1 ? (const char *)std::memchr(t1 + 1, '\t', len - (size_t)(t1 + 1 - ls)) : nullptr;
            if (!t1 || !t2) { if (errmsg) *errmsg = "record without three columns"; return 4; }
            const int64_t beg = std::strtoll(t1 + 1, nullptr, 10);
            const int64_t end = std::strtoll(t2 + 1, nullptr, 10);
            const uint64_t v0 = voff(u);
            if (!bld.push(ls, (size_t)(t1 - ls), beg, end, v0, voff(u + len), 1)) { if (errmsg) *errmsg = bld.err; return 4; }
        }
        u += len;
    }
    const std::string tp = tbi_path ? std::string(tbi_path) : std::string(path) + ".tbi";
    const int wrc = bld.write(tp.c_str());
    if (wrc) return wrc;
    if (n_records) *n_records = bld.nrec;
    return 0;
}

}  // namespace natac_tabix

// ---- reader: values of a tabix-indexed bedGraph over a region (Track.read_track, pyatac/tracks.py:75-87) ----
namespace natac_tabix {

// the parsed .tbi: immutable after open_reader, shared by the cursors of one handle
struct Index {
    std::vector<std::string> names;
    std::vector<std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins;
    std::vector<std::vector<uint64_t>> lin;
    int col_seq = 1, col_beg = 2, col_end = 3;
};

struct Reader {
    FILE *f = nullptr;
    std::shared_ptr<const Index> ix;
    std::string path;
    std::vector<std::unique_ptr<Reader>> workers;     // further cursors on the same file for read_regions (own FILE, own cache)
    std::string buf;
    // The last inflated members with their lines split and their coordinates parsed once: a driver reads neighbouring regions
    // one after the other (three tracks per chunk in `nuc` / `nfr`), a region read starts at the beginning of its 16-kb index
    // window, so consecutive reads walk over the same ~5 members; without the cache each read inflated and parsed them again.
    struct Line { uint32_t off, len; int32_t tid; int32_t nf; int64_t b0, e0; };
    struct Block {
        uint64_t coff = ~0ull, stamp = 0;
        uint32_t bsize = 0;
        uint32_t head_end = 0;          // bytes [0, head_end) finish the line the previous member ended in (or are a line of their own)
        uint32_t tail_off = 0;          // bytes [tail_off, size) start a line that ends in the next member
        std::string text;
        std::vector<Line> lines;        // the complete lines that start after the first newline, in order
    };
    static constexpr int CACHE_BLOCKS = 24;
    std::vector<Block> cache;
    uint64_t clock = 0;
    int last_tid = 0;
    ~Reader() { if (f) std::fclose(f); }
};

// inflate a whole BGZF file into memory (the .tbi itself)
inline int inflate_all(const char *path, std::string &out) {
    std::string data;
    FILE *f = std::fopen(path, "rb");
    if (!f) return 1;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    data.resize((size_t)sz);
    const bool ok = sz == 0 || std::fread(&data[0], 1, (size_t)sz, f) == (size_t)sz;
    std::fclose(f);
    if (!ok) return 1;
    const unsigned char *d = (const unsigned char *)data.data();
    uint64_t p = 0;
    while (p + 18 <= data.size()) {
        if (d[p] != 0x1f || d[p + 1] != 0x8b) return 2;
        const uint32_t xlen = d[p + 10] | (d[p + 11] << 8);
        uint32_t bsize = 0;
        for (uint64_t q = p + 12; q + 4 <= p + 12 + xlen;) {
            const uint32_t slen = d[q + 2] | (d[q + 3] << 8);
            if (d[q] == 'B' && d[q + 1] == 'C' && slen == 2) bsize = (d[q + 4] | (d[q + 5] << 8)) + 1u;
            q += 4 + slen;
        }
        if (!bsize || p + bsize > data.size()) return 2;
        const uint32_t isize = d[p + bsize - 4] | (d[p + bsize - 3] << 8) | (d[p + bsize - 2] << 16) | ((uint32_t)d[p + bsize - 1] << 24);
        const size_t o = out.size();
        out.resize(o + isize);
        if (isize) {
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) return 3;
            zs.next_in = (Bytef *)(d + p + 12 + xlen);
            zs.avail_in = bsize - 12 - xlen - 8;
            zs.next_out = (Bytef *)&out[o];
            zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != isize) return 3;
        }
        {   // CRC-32 of the member (RFC 1952)
            const uint32_t want = d[p + bsize - 8] | (d[p + bsize - 7] << 8) | (d[p + bsize - 6] << 16) | ((uint32_t)d[p + bsize - 5] << 24);
            if ((uint32_t)crc32(0L, isize ? (const Bytef *)&out[o] : (const Bytef *)"", isize) != want) return 3;
        }
        p += bsize;
    }
    return 0;
}

inline uint32_t rd32(const std::string &s, size_t &p) {
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) v |= (uint32_t)(unsigned char)s[p + i] << (8 * i);
    p += 4;
    return v;
}
inline uint64_t rd64(const std::string &s, size_t &p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)(unsigned char)s[p + i] << (8 * i);
    p += 8;
    return v;
}

// 0 ok, 1 cannot open, 2 bad format, 3 inflate error
inline int open_reader(const char *path, Reader **out) {
    std::string raw;
    const int rc = inflate_all((std::string(path) + ".tbi").c_str(), raw);
    if (rc) return rc;
    if (raw.size() < 36 || std::memcmp(raw.data(), "TBI\1", 4) != 0) return 2;
    auto ixp = std::make_shared<Index>();
    Index *r = ixp.get();
    size_t p = 4;
    const uint32_t n_ref = rd32(raw, p);
    rd32(raw, p);
    r->col_seq = (int)rd32(raw, p); r->col_beg = (int)rd32(raw, p); r->col_end = (int)rd32(raw, p);
    rd32(raw, p); rd32(raw, p);
    const uint32_t l_nm = rd32(raw, p);
    if (p + l_nm > raw.size()) return 2;
    for (size_t q = p; q < p + l_nm;) {
        const size_t e = raw.find('\0', q);
        r->names.emplace_back(raw.substr(q, e - q));
        q = e + 1;
    }
    p += l_nm;
    r->bins.resize(n_ref);
    r->lin.resize(n_ref);
    for (uint32_t t = 0; t < n_ref; ++t) {
        if (p + 4 > raw.size()) return 2;
        const uint32_t n_bin = rd32(raw, p);
        for (uint32_t b = 0; b < n_bin; ++b) {
            const uint32_t bin = rd32(raw, p), n_chunk = rd32(raw, p);
            if (p + 16ull * n_chunk > raw.size()) return 2;
            std::vector<std::pair<uint64_t, uint64_t>> cs(n_chunk);
            for (auto &c : cs) { c.first = rd64(raw, p); c.second = rd64(raw, p); }
            if (bin != 37450u) r->bins[t][bin] = std::move(cs);
        }
        const uint32_t n_intv = rd32(raw, p);
        if (p + 8ull * n_intv > raw.size()) return 2;
        r->lin[t].resize(n_intv);
        for (auto &v : r->lin[t]) v = rd64(raw, p);
    }
    Reader *rd = new Reader();
    rd->ix = ixp;
    rd->path = path;
    rd->f = std::fopen(path, "rb");
    if (!rd->f) { delete rd; return 1; }
    *out = rd;
    return 0;
}

inline void close_reader(Reader *r) { delete r; }

// split [p, p + len) at tabs into at most 8 fields; returns the count
inline int split_fields(const char *p, size_t len, const char *fld[8]) {
    int nf = 0;
    fld[nf++] = p;
    for (const char *t = p, *le = p + len; t < le && nf < 8; ++t) if (*t == '\t') fld[nf++] = t + 1;
    return nf;
}

// decimal text -> double, correctly rounded like strtod: up to 15 significant digits and a decimal exponent within +-22 are one
// exact integer and one exactly representable power of ten, so a single multiplication or division rounds correctly (Clinger
// 1990); anything else ("nan", "inf", 17-digit reprs, 1e-300) goes to strtod.  The tracks the writer produces are 12 digits.
inline double parse_double(const char *p) {
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18,
                                   1e19, 1e20, 1e21, 1e22};
    const char *s = p;
    bool neg = false, any = false, ok = true;
    if (*s == '-') { neg = true; ++s; } else if (*s == '+') ++s;
    uint64_t m = 0;
    int nd = 0, frac = 0, ex = 0;
    for (; *s >= '0' && *s <= '9'; ++s) { any = true; if (m || *s != '0') { if (++nd > 15) ok = false; else m = m * 10 + (uint64_t)(*s - '0'); } }
    if (*s == '.') {
        for (++s; *s >= '0' && *s <= '9'; ++s) { any = true; ++frac; if (m || *s != '0') { if (++nd > 15) ok = false; else m = m * 10 + (uint64_t)(*s - '0'); } }
    }
    if (any && (*s == 'e' || *s == 'E')) {
        const char *q = s + 1;
        bool eneg = false;
        if (*q == '-') { eneg = true; ++q; } else if (*q == '+') ++q;
        if (*q >= '0' && *q <= '9') {
            int nde = 0;
            for (; *q >= '0' && *q <= '9'; ++q) { if (++nde > 4) ok = false; else ex = ex * 10 + (*q - '0'); }
            if (eneg) ex = -ex;
            s = q;
        }
    }
    const int e10 = ex - frac;
    if (!any || !ok || e10 < -22 || e10 > 22 || !(*s == '\t' || *s == '\n' || *s == '\0' || *s == '\r' || *s == ' '))
        return std::strtod(p, nullptr);
    double v = (double)m;
    v = e10 < 0 ? v / P10[-e10] : v * P10[e10];
    return neg ? -v : v;
}

// non-negative decimal integer (coordinates); anything else goes to strtoll
inline int64_t parse_int(const char *p) {
    const char *s = p;
    int64_t v = 0;
    int nd = 0;
    for (; *s >= '0' && *s <= '9' && nd < 18; ++s, ++nd) v = v * 10 + (*s - '0');
    if (nd == 0 || (*s >= '0' && *s <= '9')) return std::strtoll(p, nullptr, 10);
    return v;
}

// coordinates of one text line: reference index (-1: comment, too few columns or a name the index does not hold), begin, end
inline void parse_line(Reader *r, const char *ls, size_t ll, Reader::Line &ln) {
    ln.tid = -1; ln.nf = 0; ln.b0 = ln.e0 = 0;
    if (ll == 0 || ls[0] == '#') return;
    const char *fld[8];
    const int nf = split_fields(ls, ll, fld);
    ln.nf = nf;
    if (nf < std::max(r->ix->col_seq, r->ix->col_end)) return;
    const char *sq = fld[r->ix->col_seq - 1];
    const size_t sl = (size_t)((r->ix->col_seq < nf ? fld[r->ix->col_seq] - 1 : ls + ll) - sq);
    auto same = [&](int t) { return r->ix->names[t].size() == sl && std::memcmp(r->ix->names[t].data(), sq, sl) == 0; };
    if (r->last_tid < (int)r->ix->names.size() && same(r->last_tid)) ln.tid = r->last_tid;
    else
        for (int t = 0; t < (int)r->ix->names.size(); ++t) if (same(t)) { ln.tid = r->last_tid = t; break; }
    ln.b0 = parse_int(fld[r->ix->col_beg - 1]);
    ln.e0 = parse_int(fld[r->ix->col_end - 1]);
    if (ln.e0 <= ln.b0) ln.e0 = ln.b0 + 1;
}

// the member at compressed offset coff, inflated and split into lines (from the cache when it was read recently);
// nullptr at EOF / on a read or inflate error
inline Reader::Block *get_block(Reader *r, uint64_t coff) {
    if (r->cache.empty()) r->cache.resize(Reader::CACHE_BLOCKS);
    Reader::Block *slot = &r->cache[0];
    for (auto &b : r->cache) {
        if (b.coff == coff) { b.stamp = ++r->clock; return &b; }
        if (b.stamp < slot->stamp) slot = &b;
    }
    slot->coff = ~0ull;
    slot->stamp = 0;
    unsigned char head[18];
    if (std::fseek(r->f, (long)coff, SEEK_SET) != 0 || std::fread(head, 1, 18, r->f) != 18) return nullptr;
    if (head[0] != 0x1f || head[1] != 0x8b) return nullptr;
    const uint32_t xlen = head[10] | (head[11] << 8);
    r->buf.resize(xlen + 12);
    std::memcpy(&r->buf[0], head, 18);
    if (xlen > 6 && std::fread(&r->buf[18], 1, xlen - 6, r->f) != xlen - 6) return nullptr;
    uint32_t bsize = 0;
    const unsigned char *x = (const unsigned char *)r->buf.data() + 12;
    for (uint32_t q = 0; q + 4 <= xlen;) {
        const uint32_t slen = x[q + 2] | (x[q + 3] << 8);
        if (x[q] == 'B' && x[q + 1] == 'C' && slen == 2) bsize = (x[q + 4] | (x[q + 5] << 8)) + 1u;
        q += 4 + slen;
    }
    if (!bsize) return nullptr;
    const uint32_t clen = bsize - 12 - xlen;
    r->buf.resize(clen);
    if (std::fread(&r->buf[0], 1, clen, r->f) != clen) return nullptr;
    const unsigned char *b = (const unsigned char *)r->buf.data();
    const uint32_t isize = b[clen - 4] | (b[clen - 3] << 8) | (b[clen - 2] << 16) | ((uint32_t)b[clen - 1] << 24);
    slot->text.resize(isize);
    if (isize) {
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) return nullptr;
        zs.next_in = (Bytef *)b;
        zs.avail_in = clen - 8;
        zs.next_out = (Bytef *)&slot->text[0];
        zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != isize) return nullptr;
    }
    {   // CRC-32 of the member (RFC 1952): a damaged payload that still inflates must not be parsed
        const uint32_t want = b[clen - 8] | (b[clen - 7] << 8) | (b[clen - 6] << 16) | ((uint32_t)b[clen - 5] << 24);
        if ((uint32_t)crc32(0L, isize ? (const Bytef *)slot->text.data() : (const Bytef *)"", isize) != want) return nullptr;
    }
    // line table: [0, head_end) belongs to the line the previous member ended in (or is a whole line when that member ended on
    // a newline: the reader decides from its carry), complete lines follow, [tail_off, size) continues in the next member
    slot->lines.clear();
    const char *t0 = slot->text.data(), *te = t0 + isize;
    const char *nl = isize ? (const char *)std::memchr(t0, '\n', isize) : nullptr;
    if (!nl) { slot->head_end = isize + 1; slot->tail_off = isize; }       // no newline at all: everything continues
    else {
        slot->head_end = (uint32_t)(nl + 1 - t0);
        const char *p = nl + 1;
        while (p < te) {
            const char *e = (const char *)std::memchr(p, '\n', (size_t)(te - p));
            if (!e) break;
            Reader::Line ln;
            ln.off = (uint32_t)(p - t0);
            ln.len = (uint32_t)(e - p);
            parse_line(r, p, ln.len, ln);
            slot->lines.push_back(ln);
            p = e + 1;
        }
        slot->tail_off = (uint32_t)(p - t0);
    }
    slot->bsize = bsize;
    slot->coff = coff;
    slot->stamp = ++r->clock;
    return slot;
}

// out[x - start] = value (column value_col) of every record of `chrom` overlapping [start, end), later records overwrite
// earlier ones like the reference's line loop; bases without a record keep `empty`.  Returns #records used, -1 on IO error.
inline int64_t read_values(Reader *r, const char *chrom, int64_t start, int64_t end, int value_col, double empty, double *out) {
    const int64_t n = end - start;
    for (int64_t i = 0; i < n; ++i) out[i] = empty;
    int tid = -1;
    for (size_t i = 0; i < r->ix->names.size(); ++i) if (r->ix->names[i] == chrom) tid = (int)i;
    if (tid < 0 || n <= 0) return 0;
    const int64_t qs = std::max<int64_t>(0, start), qe = end;
    if (qe <= qs) return 0;
    const auto &lin = r->ix->lin[tid];
    const size_t w = (size_t)(qs >> 14);
    const uint64_t min_off = lin.empty() ? 0 : (w < lin.size() ? lin[w] : lin.back());
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    {
        const int64_t b = qs, e = qe - 1;
        auto add = [&](uint32_t bin) {
            auto it = r->ix->bins[tid].find(bin);
            if (it == r->ix->bins[tid].end()) return;
            for (auto &c : it->second) if (c.second > min_off) chunks.push_back({std::max(c.first, min_off), c.second});
        };
        add(0);
        const int shifts[5] = {26, 23, 20, 17, 14};
        const uint32_t offs[5] = {1, 9, 73, 585, 4681};
        for (int l = 0; l < 5; ++l)
            for (int64_t k = b >> shifts[l]; k <= (e >> shifts[l]); ++k) add(offs[l] + (uint32_t)k);
    }
    std::sort(chunks.begin(), chunks.end());
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (auto &c : chunks) {
        if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second);
        else merged.push_back(c);
    }
    int64_t used = 0;
    bool done = false;
    // one record: skipped unless it is on `chrom` and overlaps [qs, qe); the first record at or past qe ends the chunk
    auto record = [&](const Reader::Line &ln, const char *ls) {
        if (ln.tid != tid || ln.nf < value_col) return;
        if (ln.b0 >= qe) { done = true; return; }
        if (ln.e0 > qs) {
            const char *fld[8];
            split_fields(ls, ln.len, fld);
            const double v = parse_double(fld[value_col - 1]);
            const int64_t a = std::max(ln.b0, start) - start, z = std::min(ln.e0, end) - start;
            for (int64_t i = a; i < z; ++i) out[i] = v;
            ++used;
        }
    };
    auto loose = [&](const char *ls, size_t ll) {                  // a line outside a member's table: parsed here
        Reader::Line ln;
        ln.off = 0;
        ln.len = (uint32_t)ll;
        parse_line(r, ls, ll, ln);
        record(ln, ls);
    };
    std::string carry;
    for (auto &c : merged) {
        uint64_t coff = c.first >> 16;
        uint32_t uoff = (uint32_t)(c.first & 0xffff);
        carry.clear();
        done = false;
        while (!done && (coff < (c.second >> 16) || (coff == (c.second >> 16) && uoff < (c.second & 0xffff)))) {
            Reader::Block *blk = get_block(r, coff);
            if (!blk) return -1;
            const uint32_t size = (uint32_t)blk->text.size();
            const uint32_t stop = (coff == (c.second >> 16)) ? std::min<uint32_t>((uint32_t)(c.second & 0xffff), size) : size;
            const char *t0 = blk->text.data();
            uint32_t pos = uoff;
            if (pos < blk->head_end && pos < stop) {
                // up to the first newline: the end of the carried line, or a line that starts at `pos`
                if (blk->head_end > stop) {                            // no newline before `stop`: continues in the next member
                    carry.append(t0 + pos, stop - pos);
                    pos = stop;
                } else {
                    if (!carry.empty()) { carry.append(t0 + pos, blk->head_end - 1 - pos); loose(carry.data(), carry.size()); carry.clear(); }
                    else loose(t0 + pos, blk->head_end - 1 - pos);
                    pos = blk->head_end;
                }
            }
            if (!done && pos < stop) {
                // table lines from `pos` on (index offsets are record starts, i.e. line starts)
                size_t lo = 0, hi = blk->lines.size();
                while (lo < hi) { const size_t mid = (lo + hi) / 2; if (blk->lines[mid].off < pos) lo = mid + 1; else hi = mid; }
                for (size_t k = lo; k < blk->lines.size() && !done; ++k) {
                    const Reader::Line &ln = blk->lines[k];
                    if (ln.off + ln.len >= stop) break;               // its newline is not before `stop`
                    record(ln, t0 + ln.off);
                    pos = ln.off + ln.len + 1;
                }
                if (!done && pos < stop) { carry.append(t0 + pos, stop - pos); pos = stop; }
            }
            coff += blk->bsize;
            uoff = 0;
        }
    }
    return used;
}

// read_values of n regions into out[out_off[i] .. out_off[i] + end[i] - start[i]): contiguous runs of the region list per thread (each
// with its own cursor and member cache, so neighbouring regions share the members they inflate).  Returns #records, -1 on IO error.
inline int64_t read_regions(Reader *r, int64_t n, const int32_t *chrom_id, const char *const *names, int32_t n_names, const int64_t *start,
                            const int64_t *end, int value_col, double empty, double *out, const int64_t *out_off, int n_threads) {
    int T = n_threads > 0 ? n_threads : natac_cores::default_threads(64);     // inflate + parse: ~300 MB/s of text per thread
    T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n / 4));
    while ((int)r->workers.size() < T - 1) {
        std::unique_ptr<Reader> w(new Reader());
        w->ix = r->ix;
        w->path = r->path;
        w->f = std::fopen(r->path.c_str(), "rb");
        if (!w->f) return -1;
        r->workers.push_back(std::move(w));
    }
    std::vector<int64_t> used((size_t)T, 0);
    auto work = [&](int t) {
        Reader *rd = t == 0 ? r : r->workers[(size_t)t - 1].get();
        for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
            const int32_t c = chrom_id[i];
            if (c < 0 || c >= n_names) { used[(size_t)t] = -1; return; }
            const int64_t u = read_values(rd, names[c], start[i], end[i], value_col, empty, out + out_off[i]);
            if (u < 0) { used[(size_t)t] = -1; return; }
            used[(size_t)t] += u;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    int64_t total = 0;
    for (auto u : used) { if (u < 0) return -1; total += u; }
    return total;
}

}  // namespace natac_tabix
