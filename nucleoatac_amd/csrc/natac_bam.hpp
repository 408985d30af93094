// natac_bam.hpp -- native BAM -> fragment arrays extractor (host C++17, multi-threaded BGZF inflate).
//
// The reference opens the BAM inside every per-chunk call and iterates AlignmentFile.fetch through pysam/htslib
// (pyatac/fragments.pyx:21-25), keeping `read.is_proper_pair and not read.is_reverse` and using only read.pos and
// read.template_length.  This extractor decodes the file ONCE into per-reference arrays (pos, |tlen|) of exactly those
// reads; nucleoatac_amd/pyatac/fragments.py slices them per chunk.
// Format: SAM spec section 4 (BGZF blocks = gzip members with the BC extra field; BAM records little-endian).
#pragma once
#include "natac_cores.hpp"
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace natac_bamio {

struct Ref {
    std::string name;
    int64_t length = 0;
    std::vector<int64_t> pos, tlen;
};

struct Bam {
    std::vector<Ref> refs;
    int64_t n_records = 0, n_kept = 0;
    std::string error;
};

inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const unsigned char *p) { return (int32_t)rd32(p); }
inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// returns nullptr + error text on failure
inline Bam *decode(const char *path, int n_threads, std::string &err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return nullptr; }
    std::fseek(f, 0, SEEK_END);
    const long fsz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> raw((size_t)std::max(0L, fsz));
    if (fsz > 0 && std::fread(raw.data(), 1, raw.size(), f) != raw.size()) { std::fclose(f); err = "short read"; return nullptr; }
    std::fclose(f);
    // ---- BGZF block table
    struct Blk { size_t off, csize; uint32_t isize; size_t uoff; };
    std::vector<Blk> blks;
    size_t o = 0, utotal = 0;
    while (o + 18 <= raw.size()) {
        const unsigned char *h = raw.data() + o;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF file (bad block header)"; return nullptr; }
        const unsigned xlen = rd16(h + 10);
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + (size_t)xlen;) {
            const unsigned slen = rd16(h + x + 2);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (size_t)rd16(h + x + 4) + 1;
            x += 4 + slen;
        }
        if (!bsize || o + bsize > raw.size()) { err = "truncated BGZF block"; return nullptr; }
        const uint32_t isize = rd32(h + bsize - 4);
        blks.push_back({o + 12 + xlen, bsize - 12 - xlen - 8, isize, utotal});
        utotal += isize;
        o += bsize;
    }
    if (o != raw.size()) { err = "trailing bytes after the last BGZF block"; return nullptr; }
    std::vector<unsigned char> data(utotal);
    // ---- parallel inflate
    if (n_threads <= 0) n_threads = natac_cores::default_threads(64);
    n_threads = std::max(1, std::min<int>(n_threads, (int)std::max<size_t>(1, blks.size())));
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        z_stream zs;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= blks.size()) break;
            const Blk &b = blks[i];
            if (b.isize == 0) continue;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
            zs.next_in = raw.data() + b.off;
            zs.avail_in = (uInt)b.csize;
            zs.next_out = data.data() + b.uoff;
            zs.avail_out = b.isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != b.isize) { bad = 1; return; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    if (bad) { err = "inflate failed (corrupt BGZF block)"; return nullptr; }
    raw.clear();
    raw.shrink_to_fit();
    // ---- BAM header
    const unsigned char *p = data.data();
    const size_t n = data.size();
    if (n < 12 || std::memcmp(p, "BAM\1", 4) != 0) { err = "not a BAM file (bad magic)"; return nullptr; }
    size_t q = 8 + (size_t)rdi32(p + 4);
    if (q + 4 > n) { err = "truncated BAM header"; return nullptr; }
    const int32_t n_ref = rdi32(p + q);
    q += 4;
    Bam *bam = new Bam();
    bam->refs.resize((size_t)std::max(0, n_ref));
    for (int32_t r = 0; r < n_ref; ++r) {
        if (q + 4 > n) { err = "truncated reference list"; delete bam; return nullptr; }
        const int32_t ln = rdi32(p + q);
        if (ln < 1 || q + 8 + (size_t)ln > n) { err = "truncated reference list"; delete bam; return nullptr; }
        bam->refs[r].name.assign((const char *)p + q + 4, (size_t)ln - 1);
        bam->refs[r].length = rdi32(p + q + 4 + ln);
        q += 8 + (size_t)ln;
    }
    // ---- records: keep FLAG & 0x2 (proper pair) and not FLAG & 0x10 (reverse strand)
    while (q + 4 <= n) {
        const int32_t bs = rdi32(p + q);
        if (bs < 32 || q + 4 + (size_t)bs > n) { err = "truncated alignment record"; delete bam; return nullptr; }
        const unsigned char *rec = p + q + 4;
        const int32_t ref_id = rdi32(rec), pos = rdi32(rec + 4);
        const uint16_t flag = rd16(rec + 14);
        const int32_t tlen = rdi32(rec + 28);
        ++bam->n_records;
        if (ref_id >= 0 && ref_id < n_ref && (flag & 0x2) && !(flag & 0x10)) {
            bam->refs[ref_id].pos.push_back(pos);
            bam->refs[ref_id].tlen.push_back(tlen < 0 ? -(int64_t)tlen : (int64_t)tlen);
            ++bam->n_kept;
        }
        q += 4 + (size_t)bs;
    }
    return bam;
}

}  // namespace natac_bamio
