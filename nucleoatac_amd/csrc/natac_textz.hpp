// natac_textz.hpp -- device kernels of the track writer: per-base float64 track -> run-length bedGraph text
// (natac_textfmt.hpp) -> BGZF members (natac_deflate.hpp).  One pass structure for any batch size:
//
//   tz_flags_count    run starts per 256-base tile                        (a run = equal consecutive values, NaNs together)
//   scan              tile counts -> run index base
//   tz_scatter_runs   R[k] = base index of run k, C[k] = its chunk
//   tz_line_len       emitted? (NaN runs, zero runs, the reference's run-before-NaN rule) + length of the text line
//   scan x2           line index, byte offset
//   tz_write_lines    the text, line_off[]
//   tz_count_tokens   token histogram of a sample of the members (all of a small batch) -> host builds ONE Huffman code per track (natac_deflate.hpp)
//   tz_emit_members   one workgroup per 0xff00-byte member: masks, tokens -> bits, CRC-32, BGZF framing
//   tz_compact        members packed back to back
#pragma once
#include "natac_deflate.hpp"
#include "natac_kernels.hpp"

namespace natac_textz {

using natac_text::P10;
namespace nd = natac_deflate;

struct TextJob {
    const double *vals;            // per-base track of the batch
    const long long *out_off;      // [nc + 1]
    const int *chunk_len;          // [nc]
    const int2 *tiles;             // (chunk, x0), 256-base tiles in chunk order
    int ntiles;
    const int *chrom_id;           // [nc] index into the name table
    const long long *chunk_start;  // [nc] genomic start of the chunk
    const char *names;             // concatenated chromosome names
    const int *name_off;           // [n_names + 1]
    const P10 *p10;
    int write_zero, keep_before_nan;
};

__device__ __forceinline__ bool same_run(double a, double b) { return a == b || (a != a && b != b); }

// ---- run starts ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tz_flags_count(TextJob job, int *__restrict__ tile_count) {
    __shared__ int wsum[4];
    const int2 t = job.tiles[blockIdx.x];
    const int L = job.chunk_len[t.x];
    const int x = t.y + threadIdx.x;
    bool start = false;
    if (x < L) {
        const long long i = job.out_off[t.x] + x;
        start = (x == 0) || !same_run(job.vals[i], job.vals[i - 1]);
    }
    const unsigned long long m = __ballot(start);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(256) tz_scatter_runs(TextJob job, const unsigned long long *__restrict__ tile_base,
                                                        unsigned int *__restrict__ R, int *__restrict__ C) {
    __shared__ int wsum[4];
    const int2 t = job.tiles[blockIdx.x];
    const int L = job.chunk_len[t.x];
    const int x = t.y + threadIdx.x;
    bool start = false;
    long long i = 0;
    if (x < L) {
        i = job.out_off[t.x] + x;
        start = (x == 0) || !same_run(job.vals[i], job.vals[i - 1]);
    }
    const unsigned long long m = __ballot(start);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int k = 0; k < w; ++k) before += wsum[k];
    if (start) {
        const unsigned long long k = tile_base[blockIdx.x] + before + __popcll(m & ((1ull << lane) - 1));
        R[k] = (unsigned int)i;
        C[k] = t.x;
    }
}

// ---- exclusive scans (in: any integer type; out: unsigned long long) -------------------------------------------------------
constexpr int SCAN_PER_BLOCK = 2048;
template <class T>
__global__ void __launch_bounds__(256) tz_scan_block_sums(const T *__restrict__ in, long long n, unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long red[4];
    const long long base = (long long)blockIdx.x * SCAN_PER_BLOCK;
    unsigned long long s = 0;
    for (int j = 0; j < 8; ++j) {
        const long long i = base + threadIdx.x * 8 + j;
        if (i < n) s += (unsigned long long)in[i];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of sums[0..n) in place by one workgroup; total -> sums[n]
__global__ void __launch_bounds__(1024) tz_scan_sums(unsigned long long *__restrict__ sums, long long n) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long long base = 0; base < n; base += 1024) {
        const long long i = base + threadIdx.x;
        const unsigned long long v = i < n ? sums[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned long long add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) sums[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[n] = carry;
}
template <class T>
__global__ void __launch_bounds__(256) tz_scan_final(const T *__restrict__ in, long long n, const unsigned long long *__restrict__ sums,
                                                      unsigned long long *__restrict__ out) {
    __shared__ unsigned long long wtot[4];
    const long long base = (long long)blockIdx.x * SCAN_PER_BLOCK;
    unsigned long long v[8], s = 0;
    for (int j = 0; j < 8; ++j) {
        const long long i = base + threadIdx.x * 8 + j;
        v[j] = i < n ? (unsigned long long)in[i] : 0;
        s += v[j];
    }
    // inclusive scan of the per-thread sums inside the wave, then across the four waves
    unsigned long long inc = s;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned long long before = sums[blockIdx.x] + inc - s;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) before += wtot[k];
    for (int j = 0; j < 8; ++j) {
        const long long i = base + threadIdx.x * 8 + j;
        if (i < n) out[i] = before;
        before += v[j];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = before;     // total
}

// Two exclusive scans of one byte array in one pass: out_sum = scan of in[i] (byte offsets of the lines), out_cnt = scan of (in[i] != 0)
// (indices of the lines: a run that writes no line has length 0).  sums[0..nblk] / sums[nblk + 1 ..] hold the two sets of block sums.
__global__ void __launch_bounds__(256) tz_scan2_block_sums(const unsigned char *__restrict__ in, long long n, long long nblk,
                                                            unsigned long long *__restrict__ sums) {
    __shared__ unsigned long long red[8];
    const long long base = (long long)blockIdx.x * SCAN_PER_BLOCK;
    unsigned long long s = 0, c = 0;
    for (int j = 0; j < 8; ++j) {
        const long long i = base + threadIdx.x * 8 + j;
        if (i < n) { const unsigned int v = in[i]; s += v; c += v ? 1u : 0u; }
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); c += __shfl_xor(c, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) { sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3]; sums[nblk + 1 + blockIdx.x] = red[4] + red[5] + red[6] + red[7]; }
}
__global__ void __launch_bounds__(256) tz_scan2_final(const unsigned char *__restrict__ in, long long n, long long nblk,
                                                       const unsigned long long *__restrict__ sums, unsigned long long *__restrict__ out_sum,
                                                       unsigned long long *__restrict__ out_cnt) {
    __shared__ unsigned long long wtot[8];
    const long long base = (long long)blockIdx.x * SCAN_PER_BLOCK;
    unsigned int v[8];
    unsigned long long s = 0, c = 0;
    for (int j = 0; j < 8; ++j) {
        const long long i = base + threadIdx.x * 8 + j;
        v[j] = i < n ? (unsigned int)in[i] : 0u;
        s += v[j];
        c += v[j] ? 1u : 0u;
    }
    unsigned long long incs = s, incc = c;
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long os = __shfl_up(incs, off), oc = __shfl_up(incc, off);
        if (lane >= off) { incs += os; incc += oc; }
    }
    if (lane == 63) { wtot[threadIdx.x >> 6] = incs; wtot[4 + (threadIdx.x >> 6)] = incc; }
    __syncthreads();
    unsigned long long bs = sums[blockIdx.x] + incs - s, bc = sums[nblk + 1 + blockIdx.x] + incc - c;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) { bs += wtot[k]; bc += wtot[4 + k]; }
    for (int j = 0; j < 8; ++j) {
        const long long i = base + threadIdx.x * 8 + j;
        if (i < n) { out_sum[i] = bs; out_cnt[i] = bc; }
        bs += v[j];
        bc += v[j] ? 1u : 0u;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) { out_sum[n] = bs; out_cnt[n] = bc; }     // totals
}

// ---- lines ------------------------------------------------------------------------------------------------------------
struct RunInfo {
    long long a_rel, b_rel;        // run = [a_rel, b_rel) relative to the chunk start
    double v;
    int chunk;
    bool emitted;
};
__device__ __forceinline__ RunInfo run_info(const TextJob &job, long long k, long long nruns, const unsigned int *R, const int *C) {
    RunInfo r;
    const long long a = R[k];
    r.chunk = C[k];
    const long long cb = job.out_off[r.chunk], ce = job.out_off[r.chunk + 1];
    const long long b = (k + 1 < nruns && C[k + 1] == r.chunk) ? (long long)R[k + 1] : ce;
    r.v = job.vals[a];
    r.a_rel = a - cb;
    r.b_rel = b - cb;
    const bool nan_follows = b < ce && job.vals[b] != job.vals[b];
    // Track.write_track (pyatac/tracks.py:56-66): NaN runs are skipped; a run directly followed by a NaN is lost because prev_value is
    // overwritten before the flush; zero runs are written only with write_zero
    r.emitted = !(r.v != r.v) && (r.v != 0.0 || job.write_zero) && (job.keep_before_nan || !nan_follows);
    return r;
}
// What a reader of the finished file sees at every base: the run's value rounded to its twelve printed digits (natac_text::round12)
// where a line is written, NaN where none is (NaN runs, runs lost before a NaN, zero runs without write_zero) -- the in-process
// stand-in for "write the track, tabix-read it back" (NucChunk.getOcc, NucleosomeCalling.py:284-293).  One thread per run; `out` is
// a second array (a neighbour's look at vals[b] must see the unwritten value).
__global__ void __launch_bounds__(256) tz_as_written(TextJob job, long long nruns, const unsigned int *__restrict__ R, const int *__restrict__ C,
                                                      double *__restrict__ out, int *__restrict__ hard_total) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= nruns) return;
    const RunInfo r = run_info(job, k, nruns, R, C);
    int hard = 0;
    const double w = r.emitted ? natac_text::round12(r.v, job.p10, &hard) : __builtin_nan("");
    const long long cb = job.out_off[r.chunk];
    for (long long i = cb + r.a_rel; i < cb + r.b_rel; ++i) out[i] = w;
    if (hard) atomicAdd(hard_total, hard);
}

// ranges of resident arrays -> one flat array (natac_store_read): region j = src[j][0 .. len[j]) -> dst[dst_off[j] ...]
struct StoreRegion { const double *src; long long dst_off, len; };
__global__ void __launch_bounds__(256) tz_store_gather(const StoreRegion *__restrict__ reg, double *__restrict__ dst) {
    const StoreRegion r = reg[blockIdx.x];
    for (long long i = threadIdx.x; i < r.len; i += 256) dst[r.dst_off + i] = r.src[i];
}

constexpr int MAX_LINE = 160;      // name (<= 64) + 2 coordinates (<= 20 each) + value (<= 24) + 4 separators
constexpr int VTXT = 24;           // bytes kept per run for the text of its value ("-1.23456789012e-308" is 19)

// The value's text is the expensive part of a line (exact '%.12g': ~2,000 instructions): tz_line_len forms it once and leaves it in
// vtxt[VTXT / 8 * k ...] for tz_write_lines, which only adds the name and the two coordinates.
__global__ void __launch_bounds__(256) tz_line_len(TextJob job, long long nruns, const unsigned int *__restrict__ R, const int *__restrict__ C,
                                                    unsigned char *__restrict__ len8, int *__restrict__ hard_total,
                                                    unsigned long long *__restrict__ vtxt) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= nruns) return;
    const RunInfo r = run_info(job, k, nruns, R, C);
    int n = 0, hard = 0;
    if (r.emitted) {
        natac_text::RegSink v;                              // the text in three registers (a char[24] here was 32 bytes of scratch per lane)
        natac_text::fmt_py2_float_to(v, r.v, job.p10, &hard);
        const int nv = v.n;
        static_assert(VTXT == 24, "RegSink holds 24 characters");
        vtxt[3 * k] = v.w0; vtxt[3 * k + 1] = v.w1; vtxt[3 * k + 2] = v.w2;
        const int cid = job.chrom_id[r.chunk];
        const long long s = job.chunk_start[r.chunk];
        n = (job.name_off[cid + 1] - job.name_off[cid]) + natac_text::digits_i64(s + r.a_rel) + natac_text::digits_i64(s + r.b_rel) + nv + 4;
    }
    len8[k] = (unsigned char)n;          // 0 = no line (NaN runs, zero runs without write_zero, the run lost before a NaN)
    if (hard) atomicAdd(hard_total, hard);
}

// The 256 lines of a workgroup are adjacent in the output: they are assembled in LDS (byte writes are cheap there) and leave as one
// contiguous span with 16-byte stores, instead of ~35 scattered byte stores per line.
__global__ void __launch_bounds__(256) tz_write_lines(TextJob job, long long nruns, const unsigned int *__restrict__ R, const int *__restrict__ C,
                                                       const unsigned char *__restrict__ len8, const unsigned long long *__restrict__ byte_off,
                                                       const unsigned long long *__restrict__ line_idx, const unsigned long long *__restrict__ vtxt,
                                                       unsigned char *__restrict__ text, long long *__restrict__ line_off, int *__restrict__ l_cid,
                                                       long long *__restrict__ l_beg, long long *__restrict__ l_end) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];      // 256 x (the longest line of this call) + 32 bytes (host)
    const long long k0 = (long long)blockIdx.x * 256, k = k0 + threadIdx.x;
    const long long k1 = k0 + 256 < nruns ? k0 + 256 : nruns;
    const unsigned long long base = byte_off[k0], end = byte_off[k1];          // byte_off has nruns + 1 entries
    const int shift = (int)(base & 15);                                        // LDS and global positions agree modulo 16
    const int n = k < nruns ? len8[k] : 0;
    if (n) {
        const int chunk = C[k];
        const long long cb = job.out_off[chunk], ce = job.out_off[chunk + 1];
        const long long a = R[k];
        const long long b = (k + 1 < nruns && C[k + 1] == chunk) ? (long long)R[k + 1] : ce;
        const long long s = job.chunk_start[chunk];
        const long long beg = s + (a - cb), stop = s + (b - cb);
        const int cid = job.chrom_id[chunk];
        unsigned long long v[VTXT / 8];
#pragma unroll
        for (int i = 0; i < VTXT / 8; ++i) v[i] = vtxt[(VTXT / 8) * k + i];
        char *dst = (char *)stage + shift + (int)(byte_off[k] - base), *p = dst;
        for (int i = job.name_off[cid]; i < job.name_off[cid + 1]; ++i) *p++ = job.names[i];
        *p++ = '\t';
        p = natac_text::put_i64(p, beg);
        *p++ = '\t';
        p = natac_text::put_i64(p, stop);
        *p++ = '\t';
        const int nv = n - (int)(p - dst) - 1;       // what tz_line_len counted for the value
#pragma unroll
        for (int w = 0; w < VTXT / 8; ++w)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (8 * w + i < nv) *p++ = (char)((v[w] >> (8 * i)) & 0xff);
        *p = '\n';
        const unsigned long long j = line_idx[k];
        line_off[j] = (long long)byte_off[k];
        if (l_cid) {                                 // the record of this line, for the tabix index (tz_group_*)
            l_cid[j] = cid;
            l_beg[j] = beg;
            l_end[j] = stop;
        }
    }
    __syncthreads();
    const long long total = (long long)(end - base);
    const unsigned long long abase = (base + 15) & ~15ull;                      // first 16-byte boundary of the span in `text`
    const long long head = (long long)(abase - base) < total ? (long long)(abase - base) : total;
    for (long long i = threadIdx.x; i < head; i += 256) text[base + i] = stage[shift + i];
    const long long nvec = (total - head) >> 4;
    const uint4 *src = (const uint4 *)(stage + shift + head);
    uint4 *out = (uint4 *)(text + abase);
    for (long long i = threadIdx.x; i < nvec; i += 256) out[i] = src[i];
    for (long long i = head + (nvec << 4) + threadIdx.x; i < total; i += 256) text[base + i] = stage[shift + i];
}

// ---- tabix records without re-reading the file ----------------------------------------------------------------------------
// The index needs, in file order, every record's bin (binning scheme of the tabix specification, min_shift 14, depth 5), its
// 16-kb windows and its virtual offsets.  Consecutive records of one chromosome inside ONE 16-kb window share a leaf bin and a
// window: such a run enters the index exactly like its records one by one (natac_tabix::Builder::push), so the device reduces
// the (tens of millions of) lines to runs -- about one per 16 kb -- and the host index builder only sees those.
__device__ __forceinline__ int tbx_reg2bin(long long beg, long long end) {
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}
__global__ void __launch_bounds__(256) tz_group_flags(long long nlines, const int *__restrict__ l_cid, const long long *__restrict__ l_beg,
                                                       const long long *__restrict__ l_end, unsigned char *__restrict__ flag) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= nlines) return;
    bool start = true;
    if (j > 0) {
        const int b1 = tbx_reg2bin(l_beg[j], l_end[j]), b0 = tbx_reg2bin(l_beg[j - 1], l_end[j - 1]);
        start = l_cid[j] != l_cid[j - 1] || b1 != b0 || b1 < 4681 || l_beg[j] < l_beg[j - 1];    // 4681 = first leaf bin; unsorted input stays visible
    }
    flag[j] = start ? 1 : 0;
}
__global__ void __launch_bounds__(256) tz_group_starts(long long nlines, const unsigned char *__restrict__ flag,
                                                        const unsigned long long *__restrict__ gidx, long long *__restrict__ first) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j < nlines && flag[j]) first[gidx[j]] = j;
}
struct GroupRec { int cid; int pad; long long beg, end, count; unsigned long long t0, t1; };
__global__ void __launch_bounds__(256) tz_group_records(long long ngroups, const long long *__restrict__ first, long long nlines,
                                                         const int *__restrict__ l_cid, const long long *__restrict__ l_beg,
                                                         const long long *__restrict__ l_end, const long long *__restrict__ line_off,
                                                         long long n_text, GroupRec *__restrict__ out) {
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const long long j0 = first[gi], j1 = (gi + 1 < ngroups ? first[gi + 1] : nlines) - 1;
    GroupRec r;
    r.cid = l_cid[j0]; r.pad = 0;
    r.beg = l_beg[j0]; r.end = l_end[j1]; r.count = j1 - j0 + 1;
    r.t0 = (unsigned long long)line_off[j0];
    r.t1 = (unsigned long long)(j1 + 1 < nlines ? line_off[j1 + 1] : n_text);
    out[gi] = r;
}

// python-2 str(float) of arbitrary doubles, MAX_VALUE_CHARS bytes reserved per value (test entry natac_format_doubles)
__global__ void __launch_bounds__(256) tz_format_values(const double *__restrict__ v, long long n, const P10 *__restrict__ p10,
                                                         char *__restrict__ out, int *__restrict__ len, int *__restrict__ hard_total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int hard = 0;
    natac_text::RegSink t;
    natac_text::fmt_py2_float_to(t, v[i], p10, &hard);
    static_assert(natac_text::MAX_VALUE_CHARS == 24, "RegSink holds 24 characters");
    unsigned long long *o = (unsigned long long *)(out + i * natac_text::MAX_VALUE_CHARS);      // 24-byte slots of an aligned buffer
    o[0] = t.w0; o[1] = t.w1; o[2] = t.w2;
    len[i] = t.n;
    if (hard) atomicAdd(hard_total, hard);
}

// ---- deflate ------------------------------------------------------------------------------------------------------------
// One workgroup of TZ_THREADS (16 waves) per member.  A wave takes 64 consecutive lines at a time, in two phases:
//   masks   the wave visits the 64 lines one after the other with lane = column: the tab positions and the three equality masks
//           of natac_deflate.hpp are ballots; lane j keeps the masks / candidate distances of line j;
//   parse   every lane runs the greedy parse of ITS line from its masks (bit arithmetic + one LDS byte per literal), so the
//           token loop -- the bulk of the instructions -- runs 64 lines wide.
constexpr int TZ_THREADS = 1024, TZ_WAVES = TZ_THREADS / 64;


struct DevCountSink {              // LDS histogram
    unsigned int *ll, *d;
    __device__ void literal(unsigned char c) { atomicAdd(&ll[c], 1u); }
    __device__ void match(int L, int dist) {
        int s, eb, ev;
        nd::len_symbol(L, &s, &eb, &ev);
        atomicAdd(&ll[s], 1u);
        nd::dist_symbol(dist, &s, &eb, &ev);
        atomicAdd(&d[s], 1u);
    }
};

// first line index whose segment may touch the member starting at bs: the line containing bs
__device__ __forceinline__ long long line_containing(const long long *line_off, long long nlines, long long pos) {
    long long lo = 0, hi = nlines;              // first index with line_off > pos
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (line_off[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    return lo > 0 ? lo - 1 : 0;
}

struct MemberGeom {
    long long bs, be, k0, nseg;    // member byte range, first line, number of line segments inside
};
__device__ __forceinline__ MemberGeom member_geom(const long long *line_off, long long nlines, long long n_text, long long b) {
    MemberGeom g;
    g.bs = b * nd::BLK;
    g.be = g.bs + nd::BLK < n_text ? g.bs + nd::BLK : n_text;
    g.k0 = line_containing(line_off, nlines, g.bs);
    long long lo = g.k0, hi = nlines;            // first line starting at or after be
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (line_off[mid] < g.be) lo = mid + 1; else hi = mid;
    }
    g.nseg = lo - g.k0;
    return g;
}

struct LaneLine {                  // what a lane needs to parse its line
    bool valid;
    int q0rel, seglen;             // segment start relative to the member start, length
    nd::Cands c;
    unsigned long long eq0, eq1, eq2;
};

__device__ __forceinline__ long long shfl_i64(long long v, int src) {
    const int lo = __shfl((int)(v & 0xffffffffll), src), hi = __shfl((int)(v >> 32), src);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// 8 bytes of the member text in LDS from any byte offset: three aligned words and two byte alignments
__device__ __forceinline__ unsigned long long lds_load8(const unsigned char *lds_text, int off) {
    const unsigned int *w = (const unsigned int *)(lds_text + (off & ~3));
    const unsigned int sh = (unsigned int)off & 3u;
    const unsigned int a0 = w[0], a1 = w[1], a2 = w[2];
    return ((unsigned long long)__builtin_amdgcn_alignbyte(a2, a1, sh) << 32) | __builtin_amdgcn_alignbyte(a1, a0, sh);
}
// bit i set <=> byte i of x equals byte i of y
__device__ __forceinline__ unsigned int eq_bytes(unsigned long long x, unsigned long long y) {
    const unsigned long long t = x ^ y, m = 0x7f7f7f7f7f7f7f7full;
    const unsigned long long z = ~(((t & m) + m) | t | m);          // 0x80 in every byte that is zero in t
    return (unsigned int)(((z >> 7) * 0x0102040810204080ull) >> 56);
}
// equality mask of the first n (<= 64) bytes at `off` against the bytes `d` earlier (positions before the member start never match)
__device__ __forceinline__ unsigned long long eq_mask(const unsigned char *lds_text, int off, int d, int n) {
    unsigned long long m = 0;
    if (d <= 0 || n <= 0) return 0;
    if (off - d >= 0) {
        for (int w = 0; 8 * w < n; ++w)
            m |= (unsigned long long)eq_bytes(lds_load8(lds_text, off + 8 * w), lds_load8(lds_text, off + 8 * w - d)) << (8 * w);
    } else {
        for (int i = 0; i < n; ++i)
            if (off + i - d >= 0 && lds_text[off + i - d] == lds_text[off + i]) m |= 1ull << i;
    }
    return n < 64 ? m & ((1ull << n) - 1ull) : m;
}
__device__ __forceinline__ unsigned long long tab_mask(const unsigned char *lds_text, int off, int n) {
    unsigned long long m = 0;
    for (int w = 0; 8 * w < n; ++w) m |= (unsigned long long)eq_bytes(lds_load8(lds_text, off + 8 * w), 0x0909090909090909ull) << (8 * w);
    return n < 64 ? (n > 0 ? m & ((1ull << n) - 1ull) : 0ull) : m;
}

// masks phase for the segments [base_s, base_s + cnt) (cnt <= 64) of the member: every lane forms the masks of ITS line, eight
// bytes at a time (the masks are what natac_deflate.hpp's tokenize_segment forms byte by byte)
__device__ __forceinline__ LaneLine group_masks(const unsigned char *lds_text, const MemberGeom &g, const long long *line_off, long long nlines,
                                                long long n_text, long long base_s, int cnt, int lane) {
    LaneLine me;
    me.valid = false;
    me.q0rel = me.seglen = 0;
    me.c.d0 = me.c.d1 = me.c.d2 = 0;
    me.eq0 = me.eq1 = me.eq2 = 0;
    const long long kk = g.k0 + base_s + lane;
    const long long ls = lane < cnt ? line_off[kk] : 0;
    const long long le = lane < cnt ? ((kk + 1 < nlines) ? line_off[kk + 1] : n_text) : 0;
    // start of the previous line: the neighbouring lane's, for lane 0 from memory
    long long pls = shfl_i64(ls, lane > 0 ? lane - 1 : 0);
    if (lane == 0) pls = kk > 0 ? line_off[kk - 1] : (long long)-1;
    const long long q0 = ls > g.bs ? ls : g.bs, q1 = le < g.be ? le : g.be;
    const int seglen = lane < cnt ? (int)(q1 - q0) : 0;
    const int n = seglen < nd::WIN ? seglen : nd::WIN;
    const int off = (int)(q0 - g.bs);
    const bool whole = lane < cnt && ls >= g.bs;                               // the line starts inside the member: q0 == ls
    const unsigned long long tabs = whole ? tab_mask(lds_text, off, n) : 0ull;
    // tab positions of the previous line (inside its first WIN columns): the neighbouring lane's mask, for lane 0 from the text
    unsigned long long ptabs;
    {
        const unsigned int lo = __shfl((unsigned int)(tabs & 0xffffffffull), lane > 0 ? lane - 1 : 0);
        const unsigned int hi = __shfl((unsigned int)(tabs >> 32), lane > 0 ? lane - 1 : 0);
        ptabs = ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0 && whole && pls >= g.bs) {
        const int pn = (int)(ls - pls) < nd::WIN ? (int)(ls - pls) : nd::WIN;
        ptabs = tab_mask(lds_text, (int)(pls - g.bs), pn);
    }
    if (whole) {
        const nd::LineGeom lg = nd::geom_from_tabmask(tabs);
        nd::LineGeom pg;
        pg.tab1 = pg.tab2 = -1;
        if (pls >= g.bs) pg = nd::geom_from_tabmask(ptabs);
        me.c = nd::line_candidates(g.bs, ls, pls, lg, pg);
        me.eq0 = eq_mask(lds_text, off, me.c.d0, n);
        me.eq1 = eq_mask(lds_text, off, me.c.d1, n);
        me.eq2 = eq_mask(lds_text, off, me.c.d2, n);
    }
    me.valid = seglen > 0;
    me.q0rel = off;
    me.seglen = seglen;
    if (lane >= cnt) { me.valid = false; me.q0rel = me.seglen = 0; }
    return me;
}

template <class Sink>
__device__ __forceinline__ void parse_lane_line(const unsigned char *lds_text, const LaneLine &me, Sink &sink) {
    if (!me.valid) return;
    const unsigned char *seg = lds_text + me.q0rel;
    nd::greedy_tokens(me.eq0, me.eq1, me.eq2, me.c, me.seglen, [&](int i) -> unsigned char { return seg[i]; }, sink);
}

// What pass A of tz_emit_members leaves per line segment for its pass B (round 6): the line's FINISHED bits, from bit 0 of a private
// range of 32-bit words in a staging buffer, and their count.  Pass B only shifts them to where the prefix sums put the line -- it does
// not parse again (until round 6 pass A left the three masks, 40 bytes per line, and pass B repeated the greedy parse: 2 of the kernel's
// 5 ms).  A token never takes more than 16 bits per character it covers (a literal <= 15; a match of >= 3 characters <= 15 + 5 + 15 + 13),
// so segment k of a member (starting q0rel bytes into it) owns the words [(q0rel + k + 1) >> 1, (q0rel' + k + 2) >> 1) -- at least
// ceil(seglen / 2) of them: the ranges cannot overlap and STAGE_WORDS per member hold them all (a line is >= 8 characters).
constexpr int STAGE_WORDS = (nd::BLK + nd::BLK / 8 + 4) / 2 + 4;
__device__ __forceinline__ int stage_offset(int q0rel, int k) { return (q0rel + k + 1) >> 1; }
struct DevStageWriter {            // LSB-first bit stream from bit 0 of its own words: plain stores
    unsigned int *wp, *w0;
    unsigned long long acc;
    int nacc;
    __device__ void init(unsigned int *w) { wp = w0 = w; acc = 0; nacc = 0; }
    __device__ void put(unsigned int v, int nb) {
        acc |= (unsigned long long)v << nacc;
        nacc += nb;
        if (nacc >= 32) { *wp++ = (unsigned int)acc; acc >>= 32; nacc -= 32; }
    }
    __device__ unsigned int finish() {          // returns the number of bits written
        if (nacc > 0) *wp = (unsigned int)acc;
        return (unsigned int)(wp - w0) * 32u + (unsigned int)nacc;
    }
};
struct DevStageSink {
    const nd::Codes *c;
    const unsigned int *lit;       // LDS: ll_code | ll_len << 16 per literal / length symbol -- one lookup per literal instead of two
    DevStageWriter *w;
    __device__ void literal(unsigned char ch) { const unsigned int e = lit[ch]; w->put(e & 0xffffu, (int)(e >> 16)); }
    __device__ void match(int L, int dist) {
        int s, eb, ev;
        nd::len_symbol(L, &s, &eb, &ev);
        const unsigned int e = lit[s];
        w->put((e & 0xffffu) | ((unsigned int)ev << (e >> 16)), (int)(e >> 16) + eb);
        nd::dist_symbol(dist, &s, &eb, &ev);
        w->put(c->d_code[s] | ((unsigned int)ev << c->d_len[s]), c->d_len[s] + eb);
    }
};
// number of line segments of every member (for the offsets of the records above)
__global__ void __launch_bounds__(256) tz_member_nseg(const long long *__restrict__ line_off, long long nlines, long long n_text,
                                                       long long nblk, unsigned int *__restrict__ nseg, long long *__restrict__ first_line) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b < nblk) {
        const MemberGeom g = member_geom(line_off, nlines, n_text, b);
        nseg[b] = (unsigned int)g.nseg;
        first_line[b] = g.k0;
    }
}
// the geometry of member b from what tz_member_nseg left: the member kernels used to repeat its two binary searches over the line offsets
// -- ~50 dependent loads in front of everything else, by every thread of the workgroup
__device__ __forceinline__ MemberGeom member_geom_stored(const unsigned int *nseg, const long long *first_line, long long n_text, long long b) {
    MemberGeom g;
    g.bs = b * nd::BLK;
    g.be = g.bs + nd::BLK < n_text ? g.bs + nd::BLK : n_text;
    g.k0 = first_line[b];
    g.nseg = nseg[b];
    return g;
}

__device__ __forceinline__ void load_member_text(unsigned char *lds_text, const unsigned char *text, const MemberGeom &g) {
    const int n = (int)(g.be - g.bs);
    const unsigned char *src = text + g.bs;      // bs is a multiple of 0xff00: 256-byte aligned
    const int n16 = n >> 4;
    const uint4 *s4 = (const uint4 *)src;
    uint4 *d4 = (uint4 *)lds_text;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) d4[i] = s4[i];
    for (int i = (n16 << 4) + threadIdx.x; i < n; i += blockDim.x) lds_text[i] = src[i];
}

__device__ __forceinline__ void wave_range(const MemberGeom &g, int wave, long long *s0, long long *s1) {
    const long long per = (g.nseg + TZ_WAVES - 1) / TZ_WAVES;
    *s0 = per * wave < g.nseg ? per * wave : g.nseg;
    *s1 = *s0 + per < g.nseg ? *s0 + per : g.nseg;
}

// token histogram of the sampled members: workgroup j takes member j * stride (natac_deflate.hpp: sample_stride)
__global__ void __launch_bounds__(TZ_THREADS) tz_count_tokens(const unsigned char *__restrict__ text, long long n_text,
                                                               const long long *__restrict__ line_off, long long nlines, int stride,
                                                               const unsigned int *__restrict__ nseg, const long long *__restrict__ first_line,
                                                               unsigned int *__restrict__ hist /* [NLL + ND] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lds_text = smem;
    unsigned int *h = (unsigned int *)(smem + 65536);
    const MemberGeom g = member_geom_stored(nseg, first_line, n_text, (long long)blockIdx.x * stride);
    for (int i = threadIdx.x; i < nd::NLL + nd::ND; i += TZ_THREADS) h[i] = 0;
    load_member_text(lds_text, text, g);
    __syncthreads();
    long long s0, s1;
    wave_range(g, threadIdx.x >> 6, &s0, &s1);
    const int lane = threadIdx.x & 63;
    DevCountSink sink{h, h + nd::NLL};
    for (long long base = s0; base < s1; base += 64) {
        const int cnt = (int)(s1 - base < 64 ? s1 - base : 64);
        const LaneLine me = group_masks(lds_text, g, line_off, nlines, n_text, base, cnt, lane);
        parse_lane_line(lds_text, me, sink);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nd::NLL + nd::ND; i += TZ_THREADS)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

struct DevBitWriter {              // LSB-first bit stream into zero-initialised 32-bit words; words shared with a neighbour are OR-ed
    unsigned int *wp;              // the word the pending bits belong to (a member's stream is < 2^19 bits: 32-bit arithmetic throughout)
    unsigned long long acc;        // pending bits, acc bit 0 = bit 0 of *wp
    int nacc;
    bool first;                    // the next flushed word is the first one of this writer (may be shared with the previous line)
    __device__ void init(unsigned int *w, unsigned int start_bit) {
        wp = w + (start_bit >> 5);
        acc = 0;
        nacc = (int)(start_bit & 31u);  // the bits below the start in the first word are pending zeros: OR-ing zeros is harmless
        first = true;
    }
    __device__ void put(unsigned int v, int nb) {
        acc |= (unsigned long long)v << nacc;
        nacc += nb;
        if (nacc >= 32) {
            if (first) { atomicOr(wp, (unsigned int)acc); first = false; }
            else *wp = (unsigned int)acc;
            ++wp;
            acc >>= 32;
            nacc -= 32;
        }
    }
    __device__ void finish() {
        if (nacc > 0) atomicOr(wp, (unsigned int)acc);
        nacc = 0;
    }
};
__device__ __forceinline__ void or_bytes(unsigned int *words, long long byte_off, unsigned long long value, int nbytes) {
    for (int i = 0; i < nbytes; ++i) {
        const long long o = byte_off + i;
        atomicOr(&words[o >> 2], (unsigned int)((value >> (8 * i)) & 0xff) << (8 * (o & 3)));
    }
}

__device__ __forceinline__ unsigned long long wave_excl_scan(unsigned long long v, int lane, unsigned long long *total) {
    unsigned long long inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    *total = __shfl(inc, 63);
    return inc - v;
}

// One workgroup per member.  out_regions: nd::REGION bytes per member, zeroed; the member occupies bytes [2, 2 + size) of its
// region (so that the deflate payload, 18 bytes later, starts on a 32-bit word).  sizes[b] = member size in bytes.
__global__ void __launch_bounds__(TZ_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) tz_emit_members(const unsigned char *__restrict__ text, long long n_text,
                                                               const long long *__restrict__ line_off, long long nlines,
                                                               const unsigned int *__restrict__ nseg, const long long *__restrict__ first_line,
                                                               const unsigned long long *__restrict__ seg_base, unsigned int *__restrict__ stage, unsigned short *__restrict__ seg_bits,
                                                               const nd::Codes *__restrict__ codes_g, const nd::CrcTables *__restrict__ crc_g,
                                                               unsigned char *__restrict__ out_regions, unsigned int *__restrict__ sizes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *lds_text = smem;
    nd::Codes *codes = (nd::Codes *)(smem + 65536);
    __shared__ unsigned long long wbits[TZ_WAVES];
    __shared__ unsigned int crc_w[TZ_WAVES];
    __shared__ unsigned int crc_tab[256];
    __shared__ unsigned int lit32[nd::NLL];
    const MemberGeom g = member_geom_stored(nseg, first_line, n_text, blockIdx.x);
    {
        const unsigned int *src = (const unsigned int *)codes_g;
        unsigned int *dst = (unsigned int *)codes;
        for (int i = threadIdx.x; i < (int)(sizeof(nd::Codes) / 4); i += TZ_THREADS) dst[i] = src[i];
        for (int i = threadIdx.x; i < 256; i += TZ_THREADS) crc_tab[i] = crc_g->table[i];
        for (int i = threadIdx.x; i < nd::NLL; i += TZ_THREADS) lit32[i] = (unsigned int)codes_g->ll_code[i] | ((unsigned int)codes_g->ll_len[i] << 16);
    }
    load_member_text(lds_text, text, g);
    __syncthreads();
    const int n = (int)(g.be - g.bs);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long s0, s1;
    wave_range(g, wave, &s0, &s1);
    // pass A: masks + ONE greedy parse per line, the line's bits into its staging words, their count into seg_bits
    unsigned int *mystage = stage + (size_t)blockIdx.x * STAGE_WORDS;
    unsigned short *mybits = seg_bits + seg_base[blockIdx.x];
    unsigned long long wave_bits = 0;
#pragma unroll 1
    for (long long base = s0; base < s1; base += 64) {
        const int cnt = (int)(s1 - base < 64 ? s1 - base : 64);
        const LaneLine me = group_masks(lds_text, g, line_off, nlines, n_text, base, cnt, lane);
        unsigned int nbits = 0;
        if (me.valid) {
            DevStageWriter sw;
            sw.init(mystage + stage_offset(me.q0rel, (int)(base + lane)));
            DevStageSink ss{codes, lit32, &sw};
            parse_lane_line(lds_text, me, ss);
            nbits = sw.finish();
        }
        if (lane < cnt) mybits[base + lane] = (unsigned short)nbits;       // a line is <= 160 characters x 15 bits
        unsigned long long tot;
        (void)wave_excl_scan((unsigned long long)nbits, lane, &tot);
        wave_bits += tot;
    }
    if (lane == 0) wbits[wave] = wave_bits;
    // CRC-32: my 64-byte slice, shifted to the start of the last slice (natac_deflate.hpp: CrcTables), XOR-reduced over the workgroup
    const int n_slices = (n + 63) >> 6;
    {
        unsigned int v = 0;
        if ((int)threadIdx.x < n_slices - 1) {
            const unsigned int c = nd::crc_bytes(crc_tab, lds_text + threadIdx.x * 64, 64);
            v = nd::crc_multmodp(crc_g->slice64[n_slices - 2 - (int)threadIdx.x], c);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v ^= __shfl_xor(v, off);
        if (lane == 0) crc_w[wave] = v;
    }
    __syncthreads();
    unsigned long long my_start = 0, total_tok_bits = 0;
    unsigned int crc = 0;
    for (int w = 0; w < TZ_WAVES; ++w) {
        if (w < wave) my_start += wbits[w];
        total_tok_bits += wbits[w];
        crc ^= crc_w[w];
    }
    if (threadIdx.x == 0 && n_slices > 0) {     // (only thread 0 writes the trailer) the last slice follows everything reduced so far
        const int c0 = (n_slices - 1) * 64;
        crc = nd::crc_multmodp(crc_g->bytes[n - c0], crc) ^ nd::crc_bytes(crc_tab, lds_text + c0, n - c0);
    }
    unsigned char *region = out_regions + (size_t)blockIdx.x * nd::REGION;
    unsigned int *words = (unsigned int *)region;
    const int eob_len = codes->ll_len[256];
    const unsigned long long total_bits = (unsigned long long)codes->hdr_bits + total_tok_bits + eob_len;
    long long payload = (long long)((total_bits + 7) >> 3);
    const bool stored = 18 + payload + 8 > 65536;
    if (!stored) {
        // shared header: whole words by the first threads, the ragged last word OR-ed (the token stream continues in it)
        const int hw = codes->hdr_bits >> 5, hr = codes->hdr_bits & 31;
        for (int i = threadIdx.x; i < hw; i += TZ_THREADS) words[5 + i] = codes->hdr[i];
        if (threadIdx.x == 0 && hr) atomicOr(&words[5 + hw], codes->hdr[hw]);
        unsigned long long pos = (unsigned long long)codes->hdr_bits + my_start;      // bit position of this wave's next group
#pragma unroll 1
        for (long long base = s0; base < s1; base += 64) {
            const bool have = base + lane < s1;
            const unsigned int nbits = have ? mybits[base + lane] : 0u;
            unsigned long long tot;
            const unsigned long long mine = wave_excl_scan((unsigned long long)nbits, lane, &tot);
            if (nbits) {
                // this lane's line: its staged words shifted to bit (pos + mine) of the member's stream.  The first and the last word it
                // touches may be shared with the neighbouring lines (OR-ed); the words in between have one owner (plain stores).
                const long long kk = g.k0 + base + lane;
                const long long ls = line_off[kk];
                const int q0rel = (int)((ls > g.bs ? ls : g.bs) - g.bs);
                const unsigned int *src = mystage + stage_offset(q0rel, (int)(base + lane));
                const unsigned int P = (unsigned int)(pos + mine), sh = P & 31u;
                unsigned int *dst = words + 5 + (P >> 5);
                const int nw = (int)((nbits + 31u) >> 5), nout = (int)((sh + nbits + 31u) >> 5);
                unsigned int prev = 0;
                for (int j = 0; j < nout; ++j) {
                    const unsigned int cur = j < nw ? src[j] : 0u;
                    const unsigned int w = sh ? (cur << sh) | (prev >> (32u - sh)) : cur;
                    prev = cur;
                    if (j == 0 || j == nout - 1) atomicOr(dst + j, w);
                    else dst[j] = w;
                }
            }
            pos += tot;
        }
        if (threadIdx.x == 0) {                               // end of block after the last token
            DevBitWriter bw;
            bw.init(words + 5, (unsigned int)((unsigned long long)codes->hdr_bits + total_tok_bits));
            bw.put(codes->ll_code[256], eob_len);
            bw.finish();
        }
    } else {
        payload = 5 + n;
        if (threadIdx.x == 0) {
            region[20] = 1;                                   // BFINAL = 1, BTYPE = 00, padding
            region[21] = (unsigned char)(n & 0xff); region[22] = (unsigned char)(n >> 8);
            region[23] = (unsigned char)(~n & 0xff); region[24] = (unsigned char)((~n >> 8) & 0xff);
        }
        for (int i = threadIdx.x; i < n; i += TZ_THREADS) region[25 + i] = lds_text[i];
    }
    // No fence here: every word two writers share (the ends of a line's bits, the end-of-block code, the trailer) is written with
    // atomicOr, which commutes; the plain stores go to words / bytes with one owner.  (An agent-scope __threadfence() writes the XCD's
    // L2 back on this part: it cost 8 of the kernel's 20 ms.)
    if (threadIdx.x == 0) {
        const long long total = 18 + payload + 8;
        for (int i = 0; i < 16; ++i) region[2 + i] = nd::bgzf_hdr_byte(i);
        region[18] = (unsigned char)((total - 1) & 0xff);
        region[19] = (unsigned char)(((total - 1) >> 8) & 0xff);
        if (stored) {
            for (int i = 0; i < 4; ++i) { region[20 + payload + i] = (unsigned char)(crc >> (8 * i)); region[24 + payload + i] = (unsigned char)((unsigned int)n >> (8 * i)); }
        } else {
            or_bytes(words, 20 + payload, crc, 4);
            or_bytes(words, 24 + payload, (unsigned int)n, 4);
        }
        sizes[blockIdx.x] = (unsigned int)total;
    }
}

// members packed back to back: out[pos[b] ..) = region b bytes [2, 2 + size)
__global__ void __launch_bounds__(256) tz_compact(const unsigned char *__restrict__ regions, const unsigned int *__restrict__ sizes,
                                                   const unsigned long long *__restrict__ pos, unsigned char *__restrict__ out) {
    const unsigned char *src = regions + (size_t)blockIdx.x * nd::REGION + 2;
    unsigned char *dst = out + pos[blockIdx.x];
    const int n = (int)sizes[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

}  // namespace natac_textz
