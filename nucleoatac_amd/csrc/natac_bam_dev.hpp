// natac_bam_dev.hpp -- BAM -> fragment arrays on the GPU: BGZF members inflated by the device, records walked by the device.
//
// The host extractor (natac_bam.hpp) spends its time in zlib: ~180 MB/s of inflated data per core, 9-18 s for the 100 M
// records behind configs[2] -- several times the whole `occ` run on the accelerated path.  A BGZF file is a sequence of
// INDEPENDENT raw-deflate members of <= 64 KiB (SAM spec 4.1), tens of thousands per gigabyte: here one LANE inflates one
// member (RFC 1951: stored, fixed and dynamic blocks, canonical Huffman decode without tables larger than the code itself), so
// a window of the file keeps every SIMD of the chip busy with serial decoders.
//
// Records chain through their block_size fields and may start anywhere in a member.  Each lane of the walk kernel finds the
// first record start of ITS member by validation (a chain of plausible records: reference ids, name length and terminator,
// field sizes against block_size), walks to the end of the member and reports where it stopped.  The host then follows
// first/exit through the member table from the one start that is known (the end of the header / of the previous window's
// carry): where a member's guess is not where the chain arrives, that member is walked again from the chain's offset (a
// single-lane launch; more than a few thousand of those per window send the file to the host decoder).  The result is exact
// by induction, not heuristic: a walk that starts at a true record start only visits true record starts.
// Kept reads (proper pair, forward strand: pyatac/fragments.pyx:24-38) are written in file order after a scan of the counts.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "natac_bam.hpp"

namespace natac_bamdev {

struct Member { unsigned long long coff; unsigned int csize, isize; unsigned long long uoff; };   // payload offset / sizes / output offset
struct WalkOut { unsigned long long first, exit; unsigned int n_rec, n_kept; };

// ---- raw deflate (RFC 1951), one stream, serial; the same code runs on the host in the CPU test suite ----------------------
// Table storage is a template parameter: on the device the symbol tables of the 64 lanes of a workgroup sit lane-interleaved in
// LDS (element i of a lane at index i * 64 + lane), on the host they are plain arrays.
template <class T, int STRIDE>
struct Strided {
    T *base;
    __host__ __device__ T &operator[](int i) const { return base[i * STRIDE]; }
};

struct BitIn {
    const unsigned char *p;      // at least 8 readable bytes behind p[n - 1]
    unsigned int n, pos;
    unsigned long long buf;
    int cnt;
    __host__ __device__ void init(const unsigned char *src, unsigned int len) { p = src; n = len; pos = 0; buf = 0; cnt = 0; }
    __host__ __device__ void refill() {              // >= 32 valid bits afterwards (bytes past the end are never legitimately used:
        if (cnt <= 32) {                             // consumed() exposes a stream that runs over)
            unsigned int w;
            __builtin_memcpy(&w, p + pos, 4);
            buf |= (unsigned long long)w << cnt;
            pos += 4;
            cnt += 32;
        }
    }
    __host__ __device__ unsigned int peek(int k) { refill(); return (unsigned int)(buf & ((1ull << k) - 1ull)); }
    __host__ __device__ void skip(int k) { buf >>= k; cnt -= k; }
    __host__ __device__ unsigned int bits(int k) { const unsigned int v = peek(k); skip(k); return v; }
    __host__ __device__ unsigned int consumed() const { return pos - (unsigned int)(cnt >> 3); }    // whole bytes taken from the input
};

// Canonical Huffman code in the form the decoder wants: for the next 15 stream bits read as a number X with the first bit on top,
// the code has length l = the smallest l with X < lim[l] (lim is non-decreasing), and the symbol is sym[(X >> (15 - l)) + delta[l]].
// lim lives in registers (static indices only), delta / sym / work in the caller's table storage.
// Returns the number of unused code points (0: complete, > 0: incomplete, < 0: over-subscribed).
template <class L8, class T16>
__host__ __device__ inline int build_code(const L8 &length, int first_sym, int n, unsigned int (&lim)[16], const T16 &delta, const T16 &sym,
                                          const T16 &work) {
#pragma unroll
    for (int l = 0; l < 16; ++l) { work[l] = 0; lim[l] = 0; }
    for (int s = 0; s < n; ++s) work[length[first_sym + s]] = (unsigned short)(work[length[first_sym + s]] + 1);
    if (work[0] == n) return 0;                        // no codes at all: every lim stays 0, decoding any symbol fails
    int left = 1;
    unsigned int code = 0, off = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const unsigned int c = work[l];
        left = (left << 1) - (int)c;
        lim[l] = (code + c) << (15 - l);
        delta[l] = (unsigned short)(off - code);        // modulo 2^16: the index sum below is taken modulo 2^16 too
        work[l] = (unsigned short)off;                  // next free slot of this length
        off += c;
        code = (code + c) << 1;
    }
    if (left < 0) return left;
    for (int s = 0; s < n; ++s) {
        const int l = length[first_sym + s];
        if (l != 0) { const unsigned short at = work[l]; sym[at] = (unsigned short)s; work[l] = (unsigned short)(at + 1); }
    }
    return left;
}

// next symbol; -1: the bits are no code
template <class T16>
__host__ __device__ inline int decode_symbol(BitIn &in, const unsigned int (&lim)[16], const T16 &delta, const T16 &sym) {
    const unsigned int x = __builtin_bitreverse32(in.peek(15)) >> 17;
    int l = 1;
#pragma unroll
    for (int k = 1; k <= 14; ++k) l += x >= lim[k] ? 1 : 0;
    if (x >= lim[15]) return -1;
    in.skip(l);
    return sym[(unsigned short)((x >> (15 - l)) + delta[l])];
}

// inflate one member's payload into out[0, isize); 0 ok, otherwise an error code (1 bad block type, 2 stored length, 3 code
// lengths, 4 bad symbol, 5 distance too far, 6 output overrun / short, 7 input overrun).  src needs 8 readable bytes of slack.
template <class L8, class T16>
__host__ __device__ inline int inflate_member(const unsigned char *src, unsigned int csize, unsigned char *out, unsigned int isize, const L8 &lengths,
                                              const T16 &lsym, const T16 &ldelta, const T16 &lwork, const T16 &dsym, const T16 &ddelta, const T16 &dwork) {
    const unsigned short LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const unsigned char LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const unsigned short DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                      8193, 12289, 16385, 24577};
    const unsigned char DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const unsigned char ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    unsigned int llim[16], dlim[16];
    BitIn in;
    in.init(src, csize);
    unsigned int o = 0;
    for (;;) {
        const unsigned int last = in.bits(1), type = in.bits(2);
        if (type == 0) {
            in.skip(in.cnt & 7);                                  // to the byte boundary
            const unsigned int len = in.bits(16), nlen = in.bits(16);
            if ((len ^ 0xffffu) != nlen) return 2;
            const unsigned int q = in.consumed();
            if (q + len > csize) return 7;
            if (o + len > isize) return 6;
            for (unsigned int i = 0; i < len; ++i) out[o + i] = src[q + i];
            o += len;
            in.pos = q + len;
            in.buf = 0;
            in.cnt = 0;
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                for (int s = 0; s < 144; ++s) lengths[s] = 8;
                for (int s = 144; s < 256; ++s) lengths[s] = 9;
                for (int s = 256; s < 280; ++s) lengths[s] = 7;
                for (int s = 280; s < 288; ++s) lengths[s] = 8;
                for (int s = 288; s < 318; ++s) lengths[s] = 5;
                build_code(lengths, 0, 288, llim, ldelta, lsym, lwork);
                build_code(lengths, 288, 30, dlim, ddelta, dsym, dwork);
            } else {
                const int nlen = (int)in.bits(5) + 257, ndist = (int)in.bits(5) + 1, ncode = (int)in.bits(4) + 4;
                if (nlen > 286 || ndist > 30) return 3;
                for (int i = 0; i < 19; ++i) lengths[i] = 0;
                for (int i = 0; i < ncode; ++i) lengths[ORDER[i]] = (unsigned char)in.bits(3);
                if (build_code(lengths, 0, 19, llim, ldelta, lsym, lwork) != 0) return 3;      // the code-length code must be complete
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = decode_symbol(in, llim, ldelta, lsym);
                    if (sym < 0) return 4;
                    if (sym < 16) lengths[idx++] = (unsigned char)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return 3;
                            val = lengths[idx - 1];
                            rep = 3 + (int)in.bits(2);
                        } else if (sym == 17) rep = 3 + (int)in.bits(3);
                        else rep = 11 + (int)in.bits(7);
                        if (idx + rep > nlen + ndist) return 3;
                        while (rep--) lengths[idx++] = (unsigned char)val;
                    }
                }
                if (lengths[256] == 0) return 3;                  // no end-of-block code
                int left = build_code(lengths, 0, nlen, llim, ldelta, lsym, lwork);
                int zeros = 0;
                for (int s = 0; s < nlen; ++s) zeros += lengths[s] == 0;
                if (left < 0 || (left > 0 && nlen - zeros != 1)) return 3;         // incomplete only with a single code
                left = build_code(lengths, nlen, ndist, dlim, ddelta, dsym, dwork);
                zeros = 0;
                for (int s = 0; s < ndist; ++s) zeros += lengths[nlen + s] == 0;
                if (left < 0 || (left > 0 && ndist - zeros != 1)) return 3;
            }
            for (;;) {
                int sym = decode_symbol(in, llim, ldelta, lsym);
                if (sym < 0) return 4;
                if (sym < 256) {
                    if (o >= isize) return 6;
                    out[o++] = (unsigned char)sym;
                } else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return 4;
                    const unsigned int len = LBASE[sym] + in.bits(LEXT[sym]);
                    const int ds = decode_symbol(in, dlim, ddelta, dsym);
                    if (ds < 0 || ds >= 30) return 4;
                    const unsigned int dist = DBASE[ds] + in.bits(DEXT[ds]);
                    if (dist > o) return 5;
                    if (o + len > isize) return 6;
                    unsigned int i = 0;
                    if (dist >= 4)                                // whole words while source and destination do not overlap within one
                        for (; i + 4 <= len; i += 4) {
                            unsigned int w;
                            __builtin_memcpy(&w, out + o + i - dist, 4);
                            __builtin_memcpy(out + o + i, &w, 4);
                        }
                    for (; i < len; ++i) out[o + i] = out[o + i - dist];
                    o += len;
                }
                if (in.consumed() > csize) return 7;
            }
        } else return 1;
        if (in.consumed() > csize) return 7;
        if (last) break;
    }
    return o == isize ? 0 : 6;
}

// the decoder with plain arrays (host test entry)
inline int inflate_member_host(const unsigned char *src, unsigned int csize, unsigned char *out, unsigned int isize) {
    unsigned char lengths[320];
    unsigned short lsym[288], ldelta[16], lwork[16], dsym[32], ddelta[16], dwork[16];
    typedef Strided<unsigned char, 1> L8;
    typedef Strided<unsigned short, 1> T16;
    return inflate_member(src, csize, out, isize, L8{lengths}, T16{lsym}, T16{ldelta}, T16{lwork}, T16{dsym}, T16{ddelta}, T16{dwork});
}

// ---- kernels -------------------------------------------------------------------------------------------------------------
// one lane per member, 64 lanes per workgroup; the symbol tables of a lane are lane-interleaved in LDS (48 KiB per workgroup), the
// code lengths of a block header -- touched only while a block's tables are built -- stay in private memory.
// status[0] = first error code (0 = none), status[1] = its member
constexpr int INFLATE_LDS_U16 = (288 + 32 + 4 * 16) * 64;
// A launch covers the members [m_begin, n_members) whose bytes have arrived; it fills the chip at most once (3 workgroups per CU: the
// LDS tables) and every lane takes members from the launch's queue until it is empty: a serial decoder needs ~85 ms per 64-KiB member, so with one member per lane a launch of 55 k members took two rounds
// of 49 k lanes, the second nearly empty.
// CRC-32 (IEEE, reflected) of p[0, n) with the slicing-by-8 tables t8[8][256] (t8[0] = the byte table): eight independent lookups
// per 8 bytes, so the dependent chain is one load per 8 bytes -- ~0.5 ms for a 64-KiB member next to the ~85 ms its inflate takes
__device__ __forceinline__ unsigned int crc32_slice8(const unsigned int *__restrict__ t8, const unsigned char *p, unsigned int n) {
    unsigned int c = 0xffffffffu;
    while (n && ((uintptr_t)p & 7)) { c = t8[(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    for (; n >= 8; n -= 8, p += 8) {
        const unsigned long long w = *(const unsigned long long *)p;
        const unsigned int lo = (unsigned int)w ^ c, hi = (unsigned int)(w >> 32);
        c = t8[7 * 256 + (lo & 0xff)] ^ t8[6 * 256 + ((lo >> 8) & 0xff)] ^ t8[5 * 256 + ((lo >> 16) & 0xff)] ^ t8[4 * 256 + (lo >> 24)] ^
            t8[3 * 256 + (hi & 0xff)] ^ t8[2 * 256 + ((hi >> 8) & 0xff)] ^ t8[1 * 256 + ((hi >> 16) & 0xff)] ^ t8[hi >> 24];
    }
    while (n) { c = t8[(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    return c ^ 0xffffffffu;
}
inline void crc32_slice8_tables(unsigned int *t8) {       // host: the tables above
    for (unsigned int i = 0; i < 256; ++i) {
        unsigned int c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        t8[i] = c;
    }
    for (int s = 1; s < 8; ++s)
        for (unsigned int i = 0; i < 256; ++i) t8[s * 256 + i] = t8[(s - 1) * 256 + i] >> 8 ^ t8[t8[(s - 1) * 256 + i] & 0xff];
}

// error code 7: the member inflated to ISIZE bytes but its CRC-32 (the four bytes behind the payload, RFC 1952) does not match
__global__ void __launch_bounds__(64) bamdev_inflate(const unsigned char *__restrict__ raw, const Member *__restrict__ mem, int m_begin, int n_members,
                                                     unsigned char *__restrict__ data, int *__restrict__ status, int *__restrict__ queue,
                                                     const unsigned int *__restrict__ crc_t8) {
    __shared__ unsigned short tab[INFLATE_LDS_U16];
    unsigned char lengths[320];
    typedef Strided<unsigned char, 1> L8;
    typedef Strided<unsigned short, 64> T16;
    unsigned short *t = tab + threadIdx.x;
    for (;;) {
        const int m = m_begin + atomicAdd(queue, 1);
        if (m >= n_members) break;
        const Member mb = mem[m];
        const unsigned char *tr = raw + mb.coff + mb.csize;              // trailer: CRC-32, ISIZE
        const unsigned int want = (unsigned int)tr[0] | ((unsigned int)tr[1] << 8) | ((unsigned int)tr[2] << 16) | ((unsigned int)tr[3] << 24);
        int rc = 0;
        if (mb.isize == 0) rc = want == 0 ? 0 : 7;
        else {
            rc = inflate_member(raw + mb.coff, mb.csize, data + mb.uoff, mb.isize, L8{lengths}, T16{t}, T16{t + 288 * 64}, T16{t + 304 * 64},
                                T16{t + 320 * 64}, T16{t + 352 * 64}, T16{t + 368 * 64});
            if (rc == 0 && crc32_slice8(crc_t8, data + mb.uoff, mb.isize) != want) rc = 7;
        }
        if (rc != 0 && atomicCAS(&status[0], 0, rc) == 0) status[1] = m;
    }
}

__device__ __forceinline__ unsigned int ld32(const unsigned char *p) {
    return (unsigned int)p[0] | ((unsigned int)p[1] << 8) | ((unsigned int)p[2] << 16) | ((unsigned int)p[3] << 24);
}

// is [o, ...) a plausible alignment record (SAM spec 4.2)?  n: valid bytes of data.  *next = start of the following record.
__device__ __forceinline__ bool plausible(const unsigned char *data, unsigned long long o, unsigned long long n, int n_ref, unsigned long long *next) {
    if (o + 36 > n) return false;
    const unsigned char *r = data + o;
    const int bs = (int)ld32(r);
    if (bs < 32 || bs > (1 << 28)) return false;
    const int ref = (int)ld32(r + 4), pos = (int)ld32(r + 8);
    const int lname = r[12];
    const int ncig = r[16] | (r[17] << 8);
    const int lseq = (int)ld32(r + 20);
    const int nref = (int)ld32(r + 24), npos = (int)ld32(r + 28);
    if (ref < -1 || ref >= n_ref || nref < -1 || nref >= n_ref || pos < -1 || npos < -1 || lname < 1 || lseq < 0) return false;
    const long long fixed = 32ll + lname + 4ll * ncig + ((long long)lseq + 1) / 2 + lseq;
    if (fixed > bs) return false;
    const unsigned long long nul = o + 36 + (unsigned long long)lname - 1;
    if (nul < n && data[nul] != 0) return false;
    *next = o + 4 + (unsigned long long)bs;
    return true;
}

// pass 0 (write == 0): first record start of every member (member 0: q0, the others by validation), walk to the member's end:
// counts and the exit offset.  pass 1 (write == 1): the same walk from wo[m].first, kept reads written at kept_base[m] + i.
// write == 2: pass 0 for the single member `only` from the given start q0 (the host found the guess of that member wrong).
// A member the host marked void (first == ~0) owns no record.  The walk stops at the first record that is not complete
// inside [0, n): that offset is the carry into the next window.
__global__ void __launch_bounds__(64) bamdev_walk(const unsigned char *__restrict__ data, unsigned long long n, const Member *__restrict__ mem,
                                                  int n_members, unsigned long long q0, int n_ref, int write, int only,
                                                  const unsigned long long *__restrict__ kept_base, WalkOut *__restrict__ wo, int *__restrict__ o_ref,
                                                  int *__restrict__ o_pos, int *__restrict__ o_tlen) {
    int m = blockIdx.x * 64 + threadIdx.x;
    if (write == 2) {
        if (m != 0) return;
        m = only;
        write = 0;
    } else only = 0;
    if (m >= n_members) return;
    const Member mb = mem[m];
    const unsigned long long ustart = mb.uoff, uend = mb.uoff + mb.isize;
    unsigned long long q;
    if (write) {
        q = wo[m].first;
        if (q == ~0ull) return;
    } else {
        WalkOut w;
        w.first = w.exit = ~0ull;
        w.n_rec = w.n_kept = 0;
        if (m == only) q = q0;
        else {
            // the first offset in the member from which CHAIN records validate, or validate exactly up to the end of the data.
            // (A chain that jumps past the end after fewer records is what a misaligned block_size looks like; the rare true
            // start of that kind -- the window's last, incomplete record -- is settled by the host with a write == 2 launch.)
            const int CHAIN = 8;
            q = ~0ull;
            for (unsigned long long o = ustart; o < uend && q == ~0ull; ++o) {
                unsigned long long c = o, nx = 0;
                int ok = 0;
                while (ok < CHAIN && plausible(data, c, n, n_ref, &nx)) { c = nx; ++ok; }
                if (ok == CHAIN || (ok > 0 && c <= n && c + 36 > n)) q = o;
            }
        }
        w.first = w.exit = q;                                   // no start found: ~0 (the host decides whether that matters)
        wo[m] = w;
        if (q == ~0ull) return;
    }
    unsigned int n_rec = 0, n_kept = 0;
    const unsigned long long base = write ? kept_base[m] : 0ull;
    while (q < uend) {
        if (q + 4 > n) break;
        const int bs = (int)ld32(data + q);
        if (bs < 32) { q = ~0ull - 1; break; }                  // malformed: reported through exit, the host rejects the file
        if (q + 4 + (unsigned long long)bs > n) break;          // incomplete inside this window: the carry starts here
        const unsigned char *r = data + q + 4;
        const int ref = (int)ld32(r);
        const unsigned int flag = r[14] | ((unsigned int)r[15] << 8);
        ++n_rec;
        if (ref >= 0 && ref < n_ref && (flag & 0x2u) && !(flag & 0x10u)) {
            if (write) {
                const int tl = (int)ld32(r + 28);
                o_ref[base + n_kept] = ref;
                o_pos[base + n_kept] = (int)ld32(r + 4);
                o_tlen[base + n_kept] = tl < 0 ? -tl : tl;
            }
            ++n_kept;
        }
        q += 4 + (unsigned long long)bs;
    }
    if (!write) {
        wo[m].exit = q;
        wo[m].n_rec = n_rec;
        wo[m].n_kept = n_kept;
    }
}

// ---- host driver ---------------------------------------------------------------------------------------------------------
// The file goes through the device in windows of ~window_bytes of compressed data (pinned staging buffer); what is not a
// complete record at the end of a window's inflated bytes is carried to the front of the next window's buffer on the device.
// Returns the same object as natac_bamio::decode.  nullptr + err: the file is damaged (same messages as the host decoder);
// nullptr + *undecided = true: the record chain could not be confirmed, the header outgrew a window or a HIP call failed (no
// memory for a window on a shared GPU): the caller uses the host decoder.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // at least n bytes; the first `keep` bytes survive a reallocation
    hipError_t reserve(size_t n, size_t keep = 0, hipStream_t stream = nullptr) {
        if (n <= cap) return hipSuccess;
        const size_t want = n + n / 8 + 4096;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { (void)hipFree(q); return e; }
        }
        if (p) (void)hipFree(p);
        p = q;
        cap = want;
        return hipSuccess;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

// The BGZF chain of a file (where every member starts, its payload and inflated sizes), walked with one small pread per member on a
// thread of its own while the bulk of the file goes to the device; the same checks and messages as the host decoder's window scan.
struct Chain {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Member> members;          // coff = ABSOLUTE file offset of the deflate payload; uoff unused here
    std::vector<unsigned long long> ends;  // absolute end offset of every member
    bool done = false;
    std::string error;
};

inline void walk_chain(int fd, unsigned long long fsize, Chain *ch) {
    using natac_bamio::rd16;
    using natac_bamio::rd32;
    std::vector<Member> part;
    std::vector<unsigned long long> part_end;
    auto publish = [&](bool done, const std::string &error) {
        std::lock_guard<std::mutex> lk(ch->mu);
        ch->members.insert(ch->members.end(), part.begin(), part.end());
        ch->ends.insert(ch->ends.end(), part_end.begin(), part_end.end());
        part.clear(); part_end.clear();
        if (done) { ch->done = true; ch->error = error; }
        ch->cv.notify_all();
    };
    unsigned long long off = 0;
    bool any = false;
    unsigned char h[4 + 1024];
    while (off < fsize) {
        if (off + 18 > fsize) return publish(true, any ? "trailing bytes after the last BGZF block" : "not a BGZF file (bad block header)");
        // the last four bytes of the previous member (its ISIZE) and this member's header in one read
        const unsigned long long from = off >= 4 ? off - 4 : 0;
        const size_t want = (size_t)std::min<unsigned long long>(sizeof h, fsize - from);
        if (pread(fd, h, want, (off_t)from) != (ssize_t)want) return publish(true, "read error");
        const unsigned char *g = h + (off - from);
        const size_t have = want - (size_t)(off - from);
        if (any) {
            const uint32_t isize = rd32(g - 4);
            if (isize > 65536) return publish(true, "corrupt BGZF block (ISIZE > 65536)");
            part.back().isize = isize;
        }
        if (g[0] != 0x1f || g[1] != 0x8b || g[2] != 8 || !(g[3] & 4)) return publish(true, "not a BGZF file (bad block header)");
        const unsigned xlen = rd16(g + 10);
        if (12 + (size_t)xlen > have) {
            if (off + 12 + xlen > fsize) return publish(true, any ? "trailing bytes after the last BGZF block" : "not a BGZF file (bad block header)");
            return publish(true, "BGZF extra field longer than 1 KiB");      // never written by samtools / htslib / this writer
        }
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + (size_t)xlen;) {
            const unsigned slen = rd16(g + x + 2);
            if (g[x] == 'B' && g[x + 1] == 'C' && slen == 2) bsize = (size_t)rd16(g + x + 4) + 1;
            x += 4 + slen;
        }
        if (!bsize || bsize < 12 + (size_t)xlen + 8) return publish(true, "truncated BGZF block");
        if (off + bsize > fsize) return publish(true, any ? "trailing bytes after the last BGZF block" : "not a BGZF file (bad block header)");
        if (part.size() >= 4096) publish(false, "");
        part.push_back({off + 12 + xlen, (unsigned int)(bsize - 12 - xlen - 8), 0u, 0ull});
        part_end.push_back(off + bsize);
        off += bsize;
        any = true;
    }
    if (any) {
        unsigned char t[4];
        if (pread(fd, t, 4, (off_t)(fsize - 4)) != 4) return publish(true, "read error");
        const uint32_t isize = rd32(t);
        if (isize > 65536) return publish(true, "corrupt BGZF block (ISIZE > 65536)");
        part.back().isize = isize;
    }
    publish(true, any ? "" : "not a BGZF file (bad block header)");
}

inline natac_bamio::Bam *decode_device(const char *path, hipStream_t stream, std::string &err, bool *undecided,
                                       size_t window_bytes = (size_t)8 << 30) {
    using natac_bamio::rd16;
    using natac_bamio::rd32;
    using natac_bamio::rdi32;
    *undecided = false;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + path; return nullptr; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); err = "cannot size the file"; return nullptr; }
    const unsigned long long fsize = (unsigned long long)sb.st_size;
    window_bytes = std::max<size_t>(window_bytes, (size_t)4096);
    // staging: the file goes to the device through two small pinned buffers (pinning memory costs ~0.25 s per GiB: a window-sized
    // pinned buffer took longer to allocate than the whole decode), the read of one overlapping the upload of the other
    const size_t STAGE = (size_t)std::min<unsigned long long>((unsigned long long)32 << 20, std::max<unsigned long long>(fsize, 4096));
    unsigned char *stage[2] = {nullptr, nullptr};
    hipEvent_t staged[2] = {nullptr, nullptr};
    natac_bamio::Bam *bam = new natac_bamio::Bam();
    DevBuf d_raw, d_mem, d_data[2], d_wo, d_base, d_ref, d_pos, d_tlen, d_status, d_queue, d_crc;
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    std::vector<Member> mem;
    std::vector<WalkOut> wo;
    std::vector<unsigned long long> base;
    std::vector<int> h_ref, h_pos, h_tlen;
    std::vector<unsigned char> head;
    int cur_buf = 0;
    Chain chain;
    std::thread walker(walk_chain, fd, fsize, &chain);
    auto cleanup = [&]() {
        if (walker.joinable()) walker.join();
        for (int i = 0; i < 2; ++i) { if (stage[i]) (void)hipHostFree(stage[i]); if (staged[i]) (void)hipEventDestroy(staged[i]); }
        for (int i = 0; i < 3; ++i) if (aux[i]) { (void)hipStreamSynchronize(aux[i]); (void)hipStreamDestroy(aux[i]); }
        close(fd);
    };
    auto fail = [&](const std::string &msg) -> natac_bamio::Bam * { err = msg; delete bam; cleanup(); return nullptr; };
    auto give_up = [&]() -> natac_bamio::Bam * { *undecided = true; delete bam; cleanup(); return nullptr; };
// a HIP failure (typically: no memory left for a window on a shared GPU) is not the file's fault: the host decoder answers
#define BAMDEV_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); *undecided = true; return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    for (int i = 0; i < 2; ++i) {
        BAMDEV_HIP(hipHostMalloc((void **)&stage[i], STAGE, hipHostMallocDefault));
        BAMDEV_HIP(hipEventCreateWithFlags(&staged[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 3; ++i) BAMDEV_HIP(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking));
    BAMDEV_HIP(d_status.reserve(4 * sizeof(int)));
    {
        std::vector<unsigned int> t8(8 * 256);
        crc32_slice8_tables(t8.data());
        BAMDEV_HIP(d_crc.reserve(t8.size() * sizeof(unsigned int)));
        BAMDEV_HIP(hipMemcpyAsync(d_crc.p, t8.data(), t8.size() * sizeof(unsigned int), hipMemcpyHostToDevice, stream));
        BAMDEV_HIP(hipStreamSynchronize(stream));      // t8 is a local
    }
    int n_cu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    }
    const bool timing = getenv("NATAC_BAM_DEV_TIMING") != nullptr;
    double t_read = 0, t_scan = 0, t_inflate = 0, t_walk = 0, t_chain = 0, t_out = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](double &acc) { const double t = now(); acc += t - t0; t0 = t; };
    unsigned long long pend = 0;       // carried bytes at the front of d_data[cur_buf]
    bool header_done = false;
    int32_t n_ref = 0;
    unsigned long long win_start = 0;  // file offset of the window = start of its first member
    size_t m0 = 0;                     // its first member in the chain
    for (;;) {
        // ---- the members of this window: as many as fit window_bytes (at least one), once the chain walk has got that far
        // (every member the walker has published carries its ISIZE)
        size_t m1 = m0;
        {
            std::unique_lock<std::mutex> lk(chain.mu);
            chain.cv.wait(lk, [&]() { return chain.done || (!chain.ends.empty() && chain.ends.back() >= win_start + window_bytes); });
            if (!chain.error.empty()) { const std::string e = chain.error; lk.unlock(); return fail(e); }
            while (m1 < chain.ends.size() && (m1 == m0 || chain.ends[m1] <= win_start + window_bytes)) ++m1;
            mem.assign(chain.members.begin() + (long)m0, chain.members.begin() + (long)m1);
        }
        if (mem.empty()) break;                                         // the whole chain is done
        unsigned long long win_end;
        { std::lock_guard<std::mutex> lk(chain.mu); win_end = chain.ends[m1 - 1]; }
        unsigned long long utotal = pend;
        for (auto &mb : mem) { mb.coff -= win_start; mb.uoff = utotal; utotal += mb.isize; }
        const int M = (int)mem.size();
        const unsigned long long n = utotal;
        const size_t o = (size_t)(win_end - win_start);
        lap(t_scan);
        // ---- upload (read of one staging buffer next to the copy of the other) + inflate: the members of every SUBWIN bytes that
        // have arrived are launched on a side stream while the rest of the window is still being read
        BAMDEV_HIP(d_raw.reserve(o + 64));
        BAMDEV_HIP(d_mem.reserve(mem.size() * sizeof(Member)));
        BAMDEV_HIP(d_data[cur_buf].reserve((size_t)n + 64, (size_t)pend, stream));      // the carry at the front stays
        unsigned char *data = (unsigned char *)d_data[cur_buf].p;
        const size_t SUBWIN = (size_t)512 << 20;
        const int max_launches = (int)(o / SUBWIN) + 2;
        BAMDEV_HIP(d_queue.reserve((size_t)max_launches * sizeof(int)));
        BAMDEV_HIP(hipMemcpyAsync(d_mem.p, mem.data(), mem.size() * sizeof(Member), hipMemcpyHostToDevice, stream));
        BAMDEV_HIP(hipMemsetAsync(d_status.p, 0, 4 * sizeof(int), stream));
        BAMDEV_HIP(hipMemsetAsync(d_queue.p, 0, (size_t)max_launches * sizeof(int), stream));
        {
            int which = 0, launched = 0, n_launch = 0;
            size_t since = 0;
            for (size_t at = 0; at < o; which ^= 1) {
                const size_t len = std::min(STAGE, o - at);
                BAMDEV_HIP(hipEventSynchronize(staged[which]));         // the copy that used this buffer last has finished
                {   // four concurrent preads fill the staging buffer (one thread copies out of the page cache at ~5 GB/s)
                    const int RT = len >= ((size_t)4 << 20) ? 4 : 1;
                    bool bad[4] = {false, false, false, false};
                    auto fill = [&](int t) {
                        size_t a = len * (size_t)t / (size_t)RT;
                        const size_t b = len * ((size_t)t + 1) / (size_t)RT;
                        while (a < b) {
                            const ssize_t r = pread(fd, stage[which] + a, b - a, (off_t)(win_start + at + a));
                            if (r <= 0) { bad[t] = true; return; }
                            a += (size_t)r;
                        }
                    };
                    std::thread th[3];
                    for (int t = 1; t < RT; ++t) th[t - 1] = std::thread(fill, t);
                    fill(0);
                    for (int t = 1; t < RT; ++t) th[t - 1].join();
                    if (bad[0] || bad[1] || bad[2] || bad[3]) return fail("read error");
                }
                BAMDEV_HIP(hipMemcpyAsync((unsigned char *)d_raw.p + at, stage[which], len, hipMemcpyHostToDevice, stream));
                BAMDEV_HIP(hipEventRecord(staged[which], stream));
                at += len;
                since += len;
                if (since >= SUBWIN || at == o) {
                    int upto = launched;                                // members that lie completely inside the uploaded bytes
                    while (upto < M && mem[(size_t)upto].coff + mem[(size_t)upto].csize + 8 <= at) ++upto;
                    if (at == o) upto = M;
                    if (upto > launched) {
                        hipStream_t side = aux[n_launch % 3];
                        BAMDEV_HIP(hipStreamWaitEvent(side, staged[which], 0));
                        const int cnt = upto - launched;
                        hipLaunchKernelGGL(bamdev_inflate, dim3((unsigned)std::min<long long>((cnt + 63) / 64, 3ll * n_cu)), dim3(64), 0, side,
                                           (const unsigned char *)d_raw.p, (const Member *)d_mem.p, launched, upto, data, (int *)d_status.p,
                                           (int *)d_queue.p + n_launch, (const unsigned int *)d_crc.p);
                        launched = upto;
                        ++n_launch;
                        since = 0;
                    }
                }
            }
        }
        lap(t_read);
        for (int i = 0; i < 3; ++i) BAMDEV_HIP(hipStreamSynchronize(aux[i]));
        int st[2] = {0, 0};
        BAMDEV_HIP(hipMemcpyAsync(st, d_status.p, sizeof st, hipMemcpyDeviceToHost, stream));
        BAMDEV_HIP(hipStreamSynchronize(stream));
        if (st[0] == 7) {      // the member's file offset: where the previous member of the chain ends
            unsigned long long at = 0;
            { std::lock_guard<std::mutex> lk(chain.mu); if (m0 + (size_t)st[1] > 0) at = chain.ends[m0 + (size_t)st[1] - 1]; }
            return fail("CRC-32 mismatch in the BGZF member at file offset " + std::to_string(at) + " (corrupt file)");
        }
        if (st[0] != 0) return fail("inflate failed (corrupt BGZF block)");
        const bool eof = win_end == fsize;
        const size_t raw_len = 0;          // (no compressed leftover: windows end on member boundaries)
        win_start = win_end;
        m0 = m1;
        lap(t_inflate);
        // ---- header (first window): parsed on the host from the front of the inflated bytes
        unsigned long long q0 = 0;
        if (!header_done) {
            size_t have = (size_t)std::min<unsigned long long>(n, (unsigned long long)8 << 20);
            bool complete = false;
            for (;;) {
                head.resize(have);
                BAMDEV_HIP(hipMemcpy(head.data(), data, have, hipMemcpyDeviceToHost));
                const unsigned char *p = head.data();
                do {
                    if (have < 12) break;
                    if (std::memcmp(p, "BAM\1", 4) != 0) return fail("not a BAM file (bad magic)");
                    size_t hq = 8 + (size_t)(uint32_t)rdi32(p + 4);
                    if (hq + 4 > have) break;
                    n_ref = rdi32(p + hq);
                    hq += 4;
                    if (n_ref < 0) return fail("truncated reference list");
                    std::vector<natac_bamio::Ref> refs((size_t)n_ref);
                    bool ok = true;
                    for (int32_t r = 0; r < n_ref; ++r) {
                        if (hq + 4 > have) { ok = false; break; }
                        const int32_t ln = rdi32(p + hq);
                        if (ln < 1) return fail("truncated reference list");
                        if (hq + 8 + (size_t)ln > have) { ok = false; break; }
                        refs[r].name.assign((const char *)p + hq + 4, (size_t)ln - 1);
                        refs[r].length = rdi32(p + hq + 4 + ln);
                        hq += 8 + (size_t)ln;
                    }
                    if (!ok) break;
                    bam->refs.swap(refs);
                    q0 = hq;
                    complete = true;
                } while (false);
                if (complete || have == n) break;
                have = (size_t)n;
            }
            if (!complete) {
                if (eof && raw_len == 0) return fail(n < 12 ? "not a BAM file (bad magic)" : "truncated BAM header");
                return give_up();          // a header larger than a window: the host decoder streams it
            }
            header_done = true;
        }
        // ---- record walk, pass 0: first / exit / counts per member
        BAMDEV_HIP(d_wo.reserve(mem.size() * sizeof(WalkOut)));
        hipLaunchKernelGGL(bamdev_walk, dim3((M + 63) / 64), dim3(64), 0, stream, (const unsigned char *)data, n, (const Member *)d_mem.p, M, q0, (int)n_ref, 0, 0,
                           (const unsigned long long *)nullptr, (WalkOut *)d_wo.p, (int *)nullptr, (int *)nullptr, (int *)nullptr);
        wo.resize(mem.size());
        BAMDEV_HIP(hipMemcpyAsync(wo.data(), d_wo.p, mem.size() * sizeof(WalkOut), hipMemcpyDeviceToHost, stream));
        BAMDEV_HIP(hipStreamSynchronize(stream));
        lap(t_walk);
        // ---- the chain from the one known start: every member's guess must be where the previous walk stopped
        base.assign(mem.size(), 0);
        unsigned long long cur = q0, kept = 0, nrec = 0;
        bool stop = false;
        int fixups = 0;
        const int MAX_FIXUPS = 4096;       // per window; beyond that the guesses are systematically off: host decoder
        for (int m = 0; m < M; ++m) {
            const unsigned long long uend = mem[m].uoff + mem[m].isize;
            if (stop || cur >= uend) { wo[m].first = ~0ull; continue; }                // owns no record
            if (cur + 36 > n && wo[m].first != cur) { wo[m].first = ~0ull; stop = true; continue; }   // an incomplete record header: the carry
            if (wo[m].first != cur) {
                // the guess of this member is not where the chain arrives (a chance pattern before it, or the window's last,
                // incomplete record): `cur` IS a record start, so the member is walked again from there
                if (++fixups > MAX_FIXUPS) return give_up();
                hipLaunchKernelGGL(bamdev_walk, dim3(1), dim3(64), 0, stream, (const unsigned char *)data, n, (const Member *)d_mem.p, M, cur, (int)n_ref, 2, m,
                                   (const unsigned long long *)nullptr, (WalkOut *)d_wo.p, (int *)nullptr, (int *)nullptr, (int *)nullptr);
                BAMDEV_HIP(hipMemcpyAsync(&wo[m], (const WalkOut *)d_wo.p + m, sizeof(WalkOut), hipMemcpyDeviceToHost, stream));
                BAMDEV_HIP(hipStreamSynchronize(stream));
                if (wo[m].first != cur) return give_up();
            }
            if (wo[m].exit == ~0ull - 1) return fail("truncated alignment record");
            base[m] = kept;
            kept += wo[m].n_kept;
            nrec += wo[m].n_rec;
            cur = wo[m].exit;
        }
        bam->n_records += (int64_t)nrec;
        lap(t_chain);
        // ---- pass 1: the kept reads in file order
        if (kept) {
            BAMDEV_HIP(d_base.reserve(mem.size() * sizeof(unsigned long long)));
            BAMDEV_HIP(d_ref.reserve(kept * sizeof(int)));
            BAMDEV_HIP(d_pos.reserve(kept * sizeof(int)));
            BAMDEV_HIP(d_tlen.reserve(kept * sizeof(int)));
            BAMDEV_HIP(hipMemcpyAsync(d_wo.p, wo.data(), mem.size() * sizeof(WalkOut), hipMemcpyHostToDevice, stream));
            BAMDEV_HIP(hipMemcpyAsync(d_base.p, base.data(), mem.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(bamdev_walk, dim3((M + 63) / 64), dim3(64), 0, stream, (const unsigned char *)data, n, (const Member *)d_mem.p, M, q0, (int)n_ref, 1, 0,
                               (const unsigned long long *)d_base.p, (WalkOut *)d_wo.p, (int *)d_ref.p, (int *)d_pos.p, (int *)d_tlen.p);
            h_ref.resize(kept); h_pos.resize(kept); h_tlen.resize(kept);
            BAMDEV_HIP(hipMemcpyAsync(h_ref.data(), d_ref.p, kept * sizeof(int), hipMemcpyDeviceToHost, stream));
            BAMDEV_HIP(hipMemcpyAsync(h_pos.data(), d_pos.p, kept * sizeof(int), hipMemcpyDeviceToHost, stream));
            BAMDEV_HIP(hipMemcpyAsync(h_tlen.data(), d_tlen.p, kept * sizeof(int), hipMemcpyDeviceToHost, stream));
            BAMDEV_HIP(hipStreamSynchronize(stream));
            for (size_t a = 0; a < kept;) {                   // runs of one reference (a sorted file has one per reference)
                size_t b = a + 1;
                while (b < kept && h_ref[b] == h_ref[a]) ++b;
                natac_bamio::Ref &r = bam->refs[(size_t)h_ref[a]];
                const size_t at = r.pos.size();
                r.pos.resize(at + (b - a));
                r.tlen.resize(at + (b - a));
                for (size_t i = a; i < b; ++i) { r.pos[at + i - a] = h_pos[i]; r.tlen[at + i - a] = h_tlen[i]; }
                a = b;
            }
            bam->n_kept += (int64_t)kept;
        }
        // ---- carry the incomplete tail to the front of the other buffer
        pend = n - cur;
        BAMDEV_HIP(d_data[cur_buf ^ 1].reserve((size_t)pend + 64));
        if (pend) BAMDEV_HIP(hipMemcpyAsync(d_data[cur_buf ^ 1].p, data + cur, (size_t)pend, hipMemcpyDeviceToDevice, stream));
        BAMDEV_HIP(hipStreamSynchronize(stream));
        cur_buf ^= 1;
        lap(t_out);
        if (eof && raw_len == 0) break;
    }
    if (timing)
        std::fprintf(stderr, "[natac_bam_dev] read + upload %.3f s, waiting for the member chain %.3f, inflate %.3f, walk %.3f, chain %.3f, kept reads + carry %.3f\n", t_read,
                     t_scan, t_inflate, t_walk, t_chain, t_out);
#undef BAMDEV_HIP
    cleanup();
    if (!header_done) { err = "truncated BAM header"; delete bam; return nullptr; }
    if (pend != 0) { err = "truncated alignment record"; delete bam; return nullptr; }
    return bam;
}

}  // namespace natac_bamdev
