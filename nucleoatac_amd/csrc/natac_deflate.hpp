// natac_deflate.hpp -- BGZF (blocked gzip) encoder for bedGraph text, shared by a host reference and the device kernels
// (natac_textz.hpp).  The reference bgzips its tracks with pysam.tabix_compress (run_occ.py:130-136, run_nuc.py:195-201); any
// valid BGZF stream of the same text is an equivalent file for tabix / bgzip / gzip readers.
//
// zlib's LZ77 is a serial hash-chain search.  bedGraph text does not need one: almost every match a general matcher would find
// is against the PREVIOUS LINE -- the same column (chromosome name, the leading digits of the coordinates), the previous line's
// end coordinate (== this line's start when runs are adjacent) -- or against this line's own start coordinate (end = start + 1
// differs in the last digits only).  So every line is tokenised independently from three candidate distances that follow from
// the line structure: no hash table, no dependence between lines, one wavefront per line (lane = column).  With a Huffman code built from the
// token histogram of the text at hand this gives SMALLER files than zlib level 4 on float tracks (10.3 vs 12.1 bytes per line on
// an occupancy track) and the same size on integer tracks, and it is embarrassingly parallel: a 0xff00-byte BGZF member is
// one workgroup.
//
// Everything here is NATAC_HD (host + device): the host reference (bgzf_lines_host) and the kernels run the SAME tokeniser and
// produce byte-identical members, so the device path is testable against a CPU restatement and zlib's inflate.
#pragma once
#include <stdint.h>
#include <string.h>

#include "natac_textfmt.hpp"

namespace natac_deflate {

constexpr int BLK = 0xff00;            // input bytes per BGZF member (what bgzip uses)
constexpr int NLL = 286, ND = 30;      // literal/length and distance alphabets (RFC 1951)
constexpr int REGION = 65536 + 64;     // bytes reserved per member in the kernels' scratch output (member starts at byte 2)
constexpr int MAX_MATCH = 258, MIN_MATCH = 3;

NATAC_HD inline void len_symbol(int L, int *sym, int *ebits, int *eval) {      // L in [3, 258]
    if (L == 258) { *sym = 285; *ebits = 0; *eval = 0; return; }
    const int x = L - 3;
    if (x < 8) { *sym = 257 + x; *ebits = 0; *eval = 0; return; }
    const int eb = 29 - __builtin_clz((unsigned)x);   // x in [4 << eb, 8 << eb), eb >= 1
    // groups of four codes per extra-bit count eb >= 1: codes 265 + 4 (eb - 1) + ((x >> eb) & 3)
    *sym = 265 + 4 * (eb - 1) + ((x >> eb) & 3);
    *ebits = eb;
    *eval = x & ((1 << eb) - 1);
}
NATAC_HD inline void dist_symbol(int d, int *sym, int *ebits, int *eval) {     // d in [1, 32768]
    const int x = d - 1;
    if (x < 4) { *sym = x; *ebits = 0; *eval = 0; return; }
    const int eb = 30 - __builtin_clz((unsigned)x);   // x in [2 << eb, 4 << eb), eb >= 1
    *sym = 2 + 2 * eb + ((x >> eb) & 1);
    *ebits = eb;
    *eval = x & ((1 << eb) - 1);
}

// ---- tokeniser ------------------------------------------------------------------------------------------------------
// A line segment is tokenised from three candidate distances given by the line structure; matches are only searched inside the
// first WIN = 64 columns of a segment (a bedGraph line is 30-45 characters; longer lines spill into literals), so that for one
// segment everything a greedy parse needs is three 64-bit equality masks:
//     eq[c] bit i  <=>  byte i of the segment equals the byte d[c] positions earlier (and that byte lies inside the member).
// On the device a wave forms the masks with one ballot per candidate (lane = column) and the parse is scalar bit arithmetic;
// the host restatement forms the same masks byte by byte.
constexpr int WIN = 64;

struct LineGeom {                  // tab positions of a line inside its first WIN columns (-1: none there)
    int tab1, tab2;
};
NATAC_HD inline LineGeom geom_from_tabmask(unsigned long long tabmask) {
    LineGeom g;
    g.tab1 = g.tab2 = -1;
    if (tabmask) {
#if defined(__HIP_DEVICE_COMPILE__)
        g.tab1 = __ffsll((long long)tabmask) - 1;
#else
        g.tab1 = __builtin_ctzll(tabmask);
#endif
        const unsigned long long rest = tabmask & (tabmask - 1);
        if (rest) {
#if defined(__HIP_DEVICE_COMPILE__)
            g.tab2 = __ffsll((long long)rest) - 1;
#else
            g.tab2 = __builtin_ctzll(rest);
#endif
        }
    }
    return g;
}

struct Cands {                     // candidate distances as scalars (0 = unused): arrays indexed in loops would live in scratch memory
    int d0, d1, d2;
};
// candidate distances of the line starting at ls (inside the member), previous line at pls with geometry pg (pls < bs: no previous)
NATAC_HD inline Cands line_candidates(long long bs, long long ls, long long pls, const LineGeom &g, const LineGeom &pg) {
    Cands c;
    c.d0 = c.d1 = c.d2 = 0;
    if (pls >= bs) {
        c.d0 = (int)(ls - pls);                                         // the same column of the previous line
        if (g.tab1 >= 0 && pg.tab2 >= 0) {                               // this start coordinate == the previous end coordinate?
            const long long d = (ls + g.tab1 + 1) - (pls + pg.tab2 + 1);
            if (d > 0 && d != c.d0 && d <= 32768) c.d1 = (int)d;
        }
    }
    if (g.tab1 >= 0 && g.tab2 >= 0) {                                    // this line's own start coordinate, seen from its end coordinate
        const int d = g.tab2 - g.tab1;
        if (d != c.d0 && d != c.d1) c.d2 = d;
    }
    return c;
}

NATAC_HD inline int trailing_ones(unsigned long long x) {                // number of consecutive set bits from bit 0
    const unsigned long long inv = ~x;
    if (!inv) return 64;
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)inv) - 1;
#else
    return __builtin_ctzll(inv);
#endif
}

// greedy parse of one segment of `seglen` bytes from its masks (eq_i belongs to distance c.d_i; a zero mask for an unused
// candidate); ties go to the lower candidate index; byte_at(i) = byte i of the segment
template <class Sink, class ByteAt>
NATAC_HD inline void greedy_tokens(unsigned long long eq0, unsigned long long eq1, unsigned long long eq2, const Cands &c, int seglen,
                                   ByteAt byte_at, Sink &sink) {
    const int n = seglen < WIN ? seglen : WIN;
    // positions where some candidate matches MIN_MATCH (= 3) bytes or more: everywhere else the greedy parse emits a literal, so the
    // loop visits matches only and hands the literal runs between them to the sink byte by byte (a line is ~2 matches and ~17
    // literals: with one test of all three masks per byte the lanes of a wave paid the match path on every step)
    unsigned long long starts = (eq0 & (eq0 >> 1) & (eq0 >> 2)) | (eq1 & (eq1 >> 1) & (eq1 >> 2)) | (eq2 & (eq2 >> 1) & (eq2 >> 2));
    int q = 0;
    while (q < n) {
        const unsigned long long rest = starts >> q;
        if (rest & 1ull) {
            int best = trailing_ones(eq0 >> q), bd = c.d0;
            const int L1 = trailing_ones(eq1 >> q), L2 = trailing_ones(eq2 >> q);
            if (L1 > best) { best = L1; bd = c.d1; }
            if (L2 > best) { best = L2; bd = c.d2; }
            if (best > n - q) best = n - q;
            if (best >= MIN_MATCH) { sink.match(best, bd); q += best; }
            else { sink.literal(byte_at(q)); ++q; }
        } else {
            int run = n - q;
            if (rest) {
                const int z = trailing_ones(~rest);
                if (z < run) run = z;
            }
            for (int i = 0; i < run; ++i) sink.literal(byte_at(q + i));
            q += run;
        }
    }
    for (; q < seglen; ++q) sink.literal(byte_at(q));
}

// host restatement: masks byte by byte.  text indexed by absolute offset; [q0, q1) = the segment, ls = start of its line,
// pls = start of the previous line (or -1)
template <class Sink>
inline void tokenize_segment(const unsigned char *text, long long bs, long long q0, long long q1, long long ls, long long pls,
                             Sink &sink) {
    const int seglen = (int)(q1 - q0);
    const int n = seglen < WIN ? seglen : WIN;
    Cands c;
    c.d0 = c.d1 = c.d2 = 0;
    if (ls >= bs) {                                          // the line starts inside the member: q0 == ls
        auto tabmask = [&](long long a, long long e) {
            unsigned long long m = 0;
            for (int i = 0; i < WIN && a + i < e; ++i)
                if (text[a + i] == '\t') m |= 1ull << i;
            return m;
        };
        const LineGeom g = geom_from_tabmask(tabmask(ls, q1));
        LineGeom pg;
        pg.tab1 = pg.tab2 = -1;
        if (pls >= bs) pg = geom_from_tabmask(tabmask(pls, ls));
        c = line_candidates(bs, ls, pls, g, pg);
    }
    auto mask = [&](int d) {
        unsigned long long m = 0;
        if (d > 0)
            for (int i = 0; i < n; ++i) {
                const long long src = q0 + i - d;
                if (src >= bs && text[src] == text[q0 + i]) m |= 1ull << i;
            }
        return m;
    };
    greedy_tokens(mask(c.d0), mask(c.d1), mask(c.d2), c, seglen, [&](int i) { return text[q0 + i]; }, sink);
}

struct CountSink {                 // token histogram (host: plain increments)
    uint32_t *ll, *d;
    NATAC_HD void literal(unsigned char c) { ll[c] += 1; }
    NATAC_HD void match(int L, int dist) {
        int s, eb, ev;
        len_symbol(L, &s, &eb, &ev);
        ll[s] += 1;
        dist_symbol(dist, &s, &eb, &ev);
        d[s] += 1;
    }
};

struct Codes {                     // Huffman codes, bit-reversed for LSB-first packing; len 0 = unused symbol
    uint16_t ll_code[NLL], d_code[ND];
    uint8_t ll_len[NLL], d_len[ND];
    uint32_t hdr[96];              // BFINAL/BTYPE + the dynamic-Huffman table description, LSB-first, shared by every member
    int hdr_bits;
};

struct BitCountSink {
    const Codes *c;
    long long bits;
    NATAC_HD void literal(unsigned char ch) { bits += c->ll_len[ch]; }
    NATAC_HD void match(int L, int dist) {
        int s, eb, ev;
        len_symbol(L, &s, &eb, &ev);
        bits += c->ll_len[s] + eb;
        dist_symbol(dist, &s, &eb, &ev);
        bits += c->d_len[s] + eb;
    }
};

// ---- CRC-32 (IEEE, reflected) with GF(2) combination of slices (the arithmetic of zlib's crc32_combine) --------------------------
constexpr uint32_t CRC_POLY = 0xedb88320u;
NATAC_HD inline uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(n 2^k) mod p; x2n[j] = x^(2^j) mod p
NATAC_HD inline uint32_t crc_x2nmodp(const uint32_t *x2n, long long n, unsigned k) {
    uint32_t p = 1u << 31;
    while (n) {
        if (n & 1) p = crc_multmodp(x2n[k & 31], p);
        n >>= 1;
        ++k;
    }
    return p;
}
NATAC_HD inline uint32_t crc_combine(const uint32_t *x2n, uint32_t crc1, uint32_t crc2, long long len2) {
    return crc_multmodp(crc_x2nmodp(x2n, len2, 3), crc1) ^ crc2;
}
NATAC_HD inline uint32_t crc_bytes(const uint32_t *table, const unsigned char *p, long long n) {
    uint32_t c = 0xffffffffu;
    for (long long i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

// CRC of a member from the CRCs of its 64-byte slices without a combination tree: crc(A || B) = shift(crc(A), |B|) ^ crc(B) and the
// shift (a multiplication by x^(8 |B|) mod p) is linear, so crc = XOR_i shift(crc_i, bytes after slice i).  All full slices are
// followed by 64 k + (length of the last slice) bytes: one multiplication by slice64[k] per slice, an XOR reduction, one more
// multiplication by bytes[last length] for everything -- instead of ten levels of two multiplications each.
struct CrcTables {
    uint32_t table[256], x2n[32];
    uint32_t slice64[1024];        // x^(8 * 64 * k) mod p
    uint32_t bytes[65];            // x^(8 * r) mod p, r = 0..64
    uint32_t pad[3];
};
inline void crc_init(CrcTables &t) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1;
        t.table[i] = c;
    }
    uint32_t p = 1u << 30;            // x^1
    t.x2n[0] = p;
    for (int n = 1; n < 32; ++n) t.x2n[n] = p = crc_multmodp(p, p);
    for (int r = 0; r <= 64; ++r) t.bytes[r] = crc_x2nmodp(t.x2n, r, 3);
    t.slice64[0] = 1u << 31;          // x^0
    for (int k = 1; k < 1024; ++k) t.slice64[k] = crc_multmodp(t.slice64[k - 1], t.bytes[64]);
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
}

}  // namespace natac_deflate

namespace natac_deflate {
NATAC_HD inline unsigned char bgzf_hdr_byte(int i) {      // fixed 16 bytes of a BGZF member header (SAM spec 4.1); BSIZE follows
    const unsigned char h[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    return h[i];
}
}  // namespace natac_deflate

// ============================================================ host side: Huffman tables, header, reference encoder
#include <algorithm>
#include <queue>
#include <string>
#include <vector>

namespace natac_deflate {

// optimal prefix-code lengths for freq[0..n), at most `limit` bits (frequencies are halved until the tree fits: the result is
// always a complete Huffman code); symbols with freq 0 get length 0
inline void huffman_lengths(const uint32_t *freq, int n, int limit, uint8_t *len) {
    std::vector<uint64_t> f(freq, freq + n);
    for (;;) {
        struct Node { uint64_t w; int l, r; };
        std::vector<Node> nodes;
        typedef std::pair<uint64_t, int> QE;
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
        for (int i = 0; i < n; ++i)
            if (f[i]) { nodes.push_back({f[i], -1 - i, 0}); q.push({f[i], (int)nodes.size() - 1}); }
        for (int i = 0; i < n; ++i) len[i] = 0;
        if (nodes.empty()) return;
        if (nodes.size() == 1) { len[-1 - nodes[0].l] = 1; return; }
        while (q.size() > 1) {
            const QE a = q.top(); q.pop();
            const QE b = q.top(); q.pop();
            nodes.push_back({a.first + b.first, a.second, b.second});
            q.push({a.first + b.first, (int)nodes.size() - 1});
        }
        int maxd = 0;
        std::vector<std::pair<int, int>> st{{q.top().second, 0}};
        while (!st.empty()) {
            const auto [id, d] = st.back();
            st.pop_back();
            if (nodes[id].l < 0) { len[-1 - nodes[id].l] = (uint8_t)std::max(d, 1); maxd = std::max(maxd, d); }
            else { st.push_back({nodes[id].l, d + 1}); st.push_back({nodes[id].r, d + 1}); }
        }
        if (maxd <= limit) return;
        for (auto &x : f) if (x) x = std::max<uint64_t>(1, x >> 1);
    }
}

inline void canonical_codes(const uint8_t *len, int n, uint16_t *code_reversed) {
    int bl_count[16] = {0};
    for (int i = 0; i < n; ++i) bl_count[len[i]]++;
    bl_count[0] = 0;
    int next[16] = {0}, code = 0;
    for (int b = 1; b <= 15; ++b) { code = (code + bl_count[b - 1]) << 1; next[b] = code; }
    for (int i = 0; i < n; ++i) {
        code_reversed[i] = 0;
        if (!len[i]) continue;
        int c = next[len[i]]++, r = 0;
        for (int k = 0; k < len[i]; ++k) r |= ((c >> k) & 1) << (len[i] - 1 - k);
        code_reversed[i] = (uint16_t)r;
    }
}

struct HostBits {
    std::vector<uint32_t> w;
    long long n = 0;
    void put(uint32_t v, int nb) {
        for (int i = 0; i < nb; ++i, ++n) {
            if ((size_t)(n >> 5) >= w.size()) w.push_back(0);
            if ((v >> i) & 1) w[n >> 5] |= 1u << (n & 31);
        }
    }
};

// The Huffman code of a track is built from the tokens of a SAMPLE of its members once there are many: every SAMPLE_STRIDE-th member
// from SAMPLE_MIN_BLK members on (a 2,500-chunk sub-batch has ~3,000).  The histogram pass then costs an eighth, and -- with the emit
// kernel forming its masks itself -- no per-line records travel between the two.  Symbols the sample did not see must still have a
// code: every literal, length and distance symbol gets one count on top (the table description grows to ~190 bytes per 64-KB
// member, +0.2 %).  Batches below the threshold keep the exact histogram and the bytes they always had.
constexpr int SAMPLE_STRIDE = 8, SAMPLE_MIN_BLK = 64;
NATAC_HD inline int sample_stride(long long nblk) { return nblk >= SAMPLE_MIN_BLK ? SAMPLE_STRIDE : 1; }
// hist_ll / hist_d: token counts of the sampled members; `counted` = how many members that were (one end-of-block each)
inline void finish_hist(uint32_t *hist_ll, uint32_t *hist_d, long long counted, int stride) {
    hist_ll[256] += (uint32_t)counted;
    if (stride > 1) {
        for (int i = 0; i < NLL; ++i) hist_ll[i] += 1;
        for (int i = 0; i < ND; ++i) hist_d[i] += 1;
    }
}

// codes + shared member header from the token histogram (hist_ll[256] must already count one end-of-block per member)
inline bool build_codes(const uint32_t *hist_ll, const uint32_t *hist_d, Codes &c) {
    uint32_t fl[NLL], fd[ND];
    for (int i = 0; i < NLL; ++i) fl[i] = hist_ll[i];
    for (int i = 0; i < ND; ++i) fd[i] = hist_d[i];
    if (!fl[256]) fl[256] = 1;
    fd[0] += 1; fd[1] += 1;                                  // at least two distance codes: a complete code for every inflater
    int used = 0;
    for (int i = 0; i < NLL; ++i) used += fl[i] != 0;
    if (used < 2) fl[0] += 1;
    huffman_lengths(fl, NLL, 15, c.ll_len);
    huffman_lengths(fd, ND, 15, c.d_len);
    canonical_codes(c.ll_len, NLL, c.ll_code);
    canonical_codes(c.d_len, ND, c.d_code);
    int nll = NLL, nd = ND;
    while (nll > 257 && !c.ll_len[nll - 1]) --nll;
    while (nd > 1 && !c.d_len[nd - 1]) --nd;
    // code-length sequence with zero runs (17: 3-10 zeros, 18: 11-138 zeros)
    std::vector<uint8_t> seq(c.ll_len, c.ll_len + nll);
    seq.insert(seq.end(), c.d_len, c.d_len + nd);
    struct CL { int sym, ebits, eval; };
    std::vector<CL> cl;
    for (size_t i = 0; i < seq.size();) {
        if (seq[i] == 0) {
            size_t j = i;
            while (j < seq.size() && seq[j] == 0 && j - i < 138) ++j;
            const int run = (int)(j - i);
            if (run >= 11) cl.push_back({18, 7, run - 11});
            else if (run >= 3) cl.push_back({17, 3, run - 3});
            else for (int k = 0; k < run; ++k) cl.push_back({0, 0, 0});
            i = j;
        } else { cl.push_back({seq[i], 0, 0}); ++i; }
    }
    uint32_t fc[19] = {0};
    for (auto &x : cl) fc[x.sym]++;
    {   // a complete code-length code needs two used symbols
        int usedc = 0;
        for (int i = 0; i < 19; ++i) usedc += fc[i] != 0;
        for (int i = 0; usedc < 2 && i < 19; ++i)
            if (!fc[i]) { fc[i] = 1; ++usedc; }
    }
    uint8_t cl_len[19];
    uint16_t cl_code[19];
    huffman_lengths(fc, 19, 7, cl_len);
    canonical_codes(cl_len, 19, cl_code);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int ncl = 19;
    while (ncl > 4 && !cl_len[order[ncl - 1]]) --ncl;
    HostBits hb;
    hb.put(1, 1);                 // BFINAL
    hb.put(2, 2);                 // BTYPE = 10 (dynamic Huffman)
    hb.put(nll - 257, 5);
    hb.put(nd - 1, 5);
    hb.put(ncl - 4, 4);
    for (int i = 0; i < ncl; ++i) hb.put(cl_len[order[i]], 3);
    for (auto &x : cl) {
        hb.put(cl_code[x.sym], cl_len[x.sym]);
        if (x.ebits) hb.put((uint32_t)x.eval, x.ebits);
    }
    if (hb.w.size() > sizeof(c.hdr) / 4) return false;
    memset(c.hdr, 0, sizeof c.hdr);
    for (size_t i = 0; i < hb.w.size(); ++i) c.hdr[i] = hb.w[i];
    c.hdr_bits = (int)hb.n;
    return true;
}

struct EmitSinkHost {
    const Codes *c;
    HostBits *out;
    void literal(unsigned char ch) { out->put(c->ll_code[ch], c->ll_len[ch]); }
    void match(int L, int dist) {
        int s, eb, ev;
        len_symbol(L, &s, &eb, &ev);
        out->put(c->ll_code[s], c->ll_len[s]);
        if (eb) out->put((uint32_t)ev, eb);
        dist_symbol(dist, &s, &eb, &ev);
        out->put(c->d_code[s], c->d_len[s]);
        if (eb) out->put((uint32_t)ev, eb);
    }
};

// segments of member [bs, be): callback(q0, q1, ls, pls) for every line piece inside it, in text order
template <class F>
inline void for_segments(const long long *line_off, long long nlines, long long n_text, long long bs, long long be, F f) {
    long long k = std::upper_bound(line_off, line_off + nlines, bs) - line_off - 1;     // line containing bs
    if (k < 0) k = 0;
    for (; k < nlines && line_off[k] < be; ++k) {
        const long long ls = line_off[k], le = (k + 1 < nlines) ? line_off[k + 1] : n_text;
        const long long q0 = std::max(ls, bs), q1 = std::min(le, be);
        if (q1 > q0) f(q0, q1, ls, k > 0 ? line_off[k - 1] : (long long)-1);
    }
}

// Host reference of the whole encoder (single thread): text + line starts -> BGZF members (no EOF marker).  The device kernels
// must produce exactly these bytes.
inline bool bgzf_lines_host(const unsigned char *text, long long n, const long long *line_off, long long nlines, std::string &out) {
    if (n <= 0) return true;
    const long long nblk = (n + BLK - 1) / BLK;
    uint32_t hl[NLL] = {0}, hd[ND] = {0};
    CountSink cs{hl, hd};
    const int stride = sample_stride(nblk);
    long long counted = 0;
    for (long long b = 0; b < nblk; b += stride, ++counted) {
        const long long bs = b * BLK, be = std::min<long long>(n, bs + BLK);
        for_segments(line_off, nlines, n, bs, be, [&](long long q0, long long q1, long long ls, long long pls) {
            tokenize_segment(text, bs, q0, q1, ls, pls, cs);
        });
    }
    finish_hist(hl, hd, counted, stride);
    Codes c;
    if (!build_codes(hl, hd, c)) return false;
    CrcTables ct;
    crc_init(ct);
    for (long long b = 0; b < nblk; ++b) {
        const long long bs = b * BLK, be = std::min<long long>(n, bs + BLK);
        HostBits hb;
        hb.w.assign(c.hdr, c.hdr + (c.hdr_bits + 31) / 32);
        hb.n = c.hdr_bits;
        EmitSinkHost es{&c, &hb};
        for_segments(line_off, nlines, n, bs, be, [&](long long q0, long long q1, long long ls, long long pls) {
            tokenize_segment(text, bs, q0, q1, ls, pls, es);
        });
        hb.put(c.ll_code[256], c.ll_len[256]);
        std::string data;
        long long nbytes = (hb.n + 7) / 8;
        if (18 + nbytes + 8 > 65536) {              // stored member
            const uint16_t ln = (uint16_t)(be - bs);
            data.push_back(1);
            data.push_back((char)(ln & 0xff)); data.push_back((char)(ln >> 8));
            data.push_back((char)(~ln & 0xff)); data.push_back((char)((~ln >> 8) & 0xff));
            data.append((const char *)text + bs, (size_t)(be - bs));
        } else {
            hb.w.resize((size_t)((nbytes + 3) / 4), 0);
            data.assign((const char *)hb.w.data(), (size_t)nbytes);
        }
        const size_t total = 18 + data.size() + 8;
        for (int i = 0; i < 16; ++i) out.push_back((char)bgzf_hdr_byte(i));
        out.push_back((char)((total - 1) & 0xff));
        out.push_back((char)(((total - 1) >> 8) & 0xff));
        out.append(data);
        const uint32_t crc = crc_bytes(ct.table, text + bs, be - bs), isz = (uint32_t)(be - bs);
        for (int i = 0; i < 4; ++i) out.push_back((char)((crc >> (8 * i)) & 0xff));
        for (int i = 0; i < 4; ++i) out.push_back((char)((isz >> (8 * i)) & 0xff));
    }
    return true;
}

}  // namespace natac_deflate
