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

// Streaming decode with bounded memory: the file is read in windows of ~batch_bytes of compressed data; the complete BGZF
// blocks of a window are inflated in parallel into one buffer that is prefixed with the unparsed remainder of the
// previous window (a BAM record -- or the header -- may straddle any number of blocks), the records are walked, and only
// the two numbers per kept read stay.  Peak memory = one compressed window + its inflated form (+ the output arrays),
// independent of the file size.  returns nullptr + error text on failure.
inline Bam *decode(const char *path, int n_threads, std::string &err, size_t batch_bytes = (size_t)48 << 20) {
    FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return nullptr; }
    if (n_threads <= 0) n_threads = natac_cores::default_threads(64);
    batch_bytes = std::max<size_t>(batch_bytes, (size_t)4096);           // + 64 KiB below: always room for one maximal BGZF block
    struct Blk { size_t off, csize; uint32_t isize; size_t uoff; uint32_t crc; };
    std::vector<unsigned char> raw, data;
    std::vector<Blk> blks;
    size_t raw_len = 0;            // valid bytes in raw (starts with the leftover of the previous window)
    size_t pend = 0;               // unparsed bytes at the front of data
    bool eof = false, header_done = false, any_block = false;
    int32_t n_ref = 0;
    Bam *bam = new Bam();
    auto fail = [&](const char *msg) -> Bam * { err = msg; delete bam; std::fclose(f); return nullptr; };
    unsigned long long window_file_off = 0;      // file offset of raw[0]
    raw.resize(batch_bytes + ((size_t)1 << 16));
    while (!eof || raw_len > 0) {
        // ---- refill the compressed window
        if (!eof) {
            const size_t want = raw.size() - raw_len;
            const size_t got = std::fread(raw.data() + raw_len, 1, want, f);
            raw_len += got;
            if (got < want) eof = true;
        }
        // ---- complete BGZF blocks of the window
        blks.clear();
        size_t o = 0, utotal = pend;
        while (o + 18 <= raw_len) {
            const unsigned char *h = raw.data() + o;
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return fail("not a BGZF file (bad block header)");
            const unsigned xlen = rd16(h + 10);
            if (o + 12 + xlen > raw_len) break;
            size_t bsize = 0;
            for (size_t x = 12; x + 4 <= 12 + (size_t)xlen;) {
                const unsigned slen = rd16(h + x + 2);
                if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = (size_t)rd16(h + x + 4) + 1;
                x += 4 + slen;
            }
            if (!bsize || bsize < 12 + (size_t)xlen + 8) return fail("truncated BGZF block");
            if (o + bsize > raw_len) break;                       // the rest of this block comes with the next window
            const uint32_t isize = rd32(h + bsize - 4);
            // a BGZF member inflates to at most 64 KiB (SAM spec 4.1): a larger ISIZE is a corrupt or hostile file and would
            // otherwise size the window buffer (the bounded-memory guarantee of this decoder rests on this check)
            if (isize > 65536) return fail("corrupt BGZF block (ISIZE > 65536)");
            blks.push_back({o + 12 + xlen, bsize - 12 - xlen - 8, isize, utotal, rd32(h + bsize - 8)});
            utotal += isize;
            o += bsize;
            any_block = true;
        }
        if (blks.empty()) {
            if (eof) {
                if (raw_len > 0) return fail(any_block ? "trailing bytes after the last BGZF block" : "not a BGZF file (bad block header)");
                break;
            }
            return fail("BGZF block larger than the read window");
        }
        if (data.size() < utotal) data.resize(utotal);               // pend bytes at the front are kept by resize
        // ---- parallel inflate of the window's blocks
        {
            const int nt = std::max(1, std::min<int>(n_threads, (int)blks.size()));
            std::atomic<size_t> next(0);
            std::atomic<int> bad(0);
            std::atomic<long long> bad_crc(-1);
            auto work = [&]() {
                z_stream zs;
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= blks.size()) break;
                    const Blk &bk = blks[i];
                    if (bk.isize == 0) {
                        if (bk.crc != 0) { long long none = -1; bad_crc.compare_exchange_strong(none, (long long)i); }
                        continue;
                    }
                    std::memset(&zs, 0, sizeof zs);
                    if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
                    zs.next_in = raw.data() + bk.off;
                    zs.avail_in = (uInt)bk.csize;
                    zs.next_out = data.data() + bk.uoff;
                    zs.avail_out = bk.isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    inflateEnd(&zs);
                    if (rc != Z_STREAM_END || zs.total_out != bk.isize) { bad = 1; return; }
                    // the member's CRC-32 (RFC 1952; htslib checks it behind every pysam read of the reference, pyatac/fragments.pyx:21):
                    // a damaged payload can still inflate to ISIZE bytes
                    if ((uint32_t)crc32(0L, data.data() + bk.uoff, bk.isize) != bk.crc) {
                        long long none = -1;
                        bad_crc.compare_exchange_strong(none, (long long)i);
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
            if (bad) return fail("inflate failed (corrupt BGZF block)");
            if (bad_crc >= 0) {
                // the member starts 12 + XLEN bytes before its payload; XLEN is 6 for every BGZF writer, read it back to be exact
                const size_t pay = blks[(size_t)bad_crc.load()].off;
                size_t hdr = pay >= 18 ? pay - 18 : 0;
                for (size_t b = pay >= 18 ? pay - 18 : 0; b + 12 <= pay; ++b)      // (only XLEN = pay - b - 12 is consistent)
                    if (raw[b] == 0x1f && raw[b + 1] == 0x8b && (size_t)rd16(raw.data() + b + 10) == pay - b - 12) { hdr = b; break; }
                static thread_local char msg[128];
                std::snprintf(msg, sizeof msg, "CRC-32 mismatch in the BGZF member at file offset %llu (corrupt file)", window_file_off + hdr);
                return fail(msg);
            }
        }
        std::memmove(raw.data(), raw.data() + o, raw_len - o);      // leftover compressed bytes (a partial block)
        raw_len -= o;
        window_file_off += o;
        // ---- walk the uncompressed bytes [0, utotal)
        const unsigned char *p = data.data();
        const size_t n = utotal;
        size_t q = 0;
        if (!header_done) {
            // the header is parsed in one go once it is complete (it may span several windows)
            bool complete = false;
            do {
                if (n < 12) break;
                if (std::memcmp(p, "BAM\1", 4) != 0) return fail("not a BAM file (bad magic)");
                size_t hq = 8 + (size_t)(uint32_t)rdi32(p + 4);
                if (hq + 4 > n) break;
                n_ref = rdi32(p + hq);
                hq += 4;
                if (n_ref < 0) return fail("truncated reference list");
                std::vector<Ref> refs((size_t)n_ref);
                bool ok = true;
                for (int32_t r = 0; r < n_ref; ++r) {
                    if (hq + 4 > n) { ok = false; break; }
                    const int32_t ln = rdi32(p + hq);
                    if (ln < 1) return fail("truncated reference list");
                    if (hq + 8 + (size_t)ln > n) { ok = false; break; }
                    refs[r].name.assign((const char *)p + hq + 4, (size_t)ln - 1);
                    refs[r].length = rdi32(p + hq + 4 + ln);
                    hq += 8 + (size_t)ln;
                }
                if (!ok) break;
                bam->refs.swap(refs);
                q = hq;
                complete = true;
            } while (false);
            if (!complete) {
                if (eof && raw_len == 0) return fail(n < 12 ? "not a BAM file (bad magic)" : "truncated BAM header");
                pend = n;                                             // wait for more data
                continue;
            }
            header_done = true;
        }
        // ---- records: keep FLAG & 0x2 (proper pair) and not FLAG & 0x10 (reverse strand)
        while (q + 4 <= n) {
            const int32_t bs = rdi32(p + q);
            if (bs < 32) return fail("truncated alignment record");
            if (q + 4 + (size_t)bs > n) break;                        // continues in the next window
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
        pend = n - q;
        std::memmove(data.data(), data.data() + q, pend);
        if (eof && raw_len == 0) break;
    }
    std::fclose(f);
    if (!header_done) { err = any_block ? "truncated BAM header" : "not a BGZF file (bad block header)"; delete bam; return nullptr; }
    if (pend != 0) { err = "truncated alignment record"; delete bam; return nullptr; }
    return bam;
}

}  // namespace natac_bamio
