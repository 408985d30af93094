// natac_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for NucleoATAC's occ + nuc signal path.
//
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fma(); everything else
// rounds like the reference's numpy expressions.  Wave = 64 lanes throughout.
//
// Geometry (reference: SURVEY.md Appendix C).  Inside a chunk all coordinates are relative to the chunk
// start; `bias` is the chunk's log-bias slice, index j <-> coordinate j - bias_left.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace natac {

constexpr int WAVE = 64;

struct ChunkTable {
    int nc;
    const int *chunk_len;       // [nc]
    const long long *frag_off;  // [nc+1]
    const int *lpos;            // [nf]
    const int *ilen;            // [nf]
    const int *centre;          // [nf]  lpos + floor((ilen-1)/2), non-decreasing inside a chunk
    const long long *bias_off;  // [nc+1] or null
    const double *bias;         // log-bias or null
    const double *ebias;        // exp(bias), same layout (natac_exp_bias; filled by natac_run_nuc / natac_run_occ), or null
    int bias_left, bias_right;
    const long long *out_off;   // [nc+1]
    const long long *grid_off;  // [nc+1]
};

constexpr int VPAD = 128;  // zero columns on both sides of every row of VMatDev::matp (natac_frag_gather)

struct VMatDev {
    const double *mat;   // R x W row-major
    const double *matp;  // R x (W + 2 VPAD): the same rows between VPAD zero columns
    const double *srow;  // [R] sizes[lower + r]
    const double *lrt;   // R x W: log(V[r,c] / sizes[lower + r]) (natac_lr_table; models without exact zeros), or null
    int lower, upper, w, R, W;
    int has_zero;        // the template or srow holds an exact 0 (host-computed)
};

__device__ __forceinline__ int floor_half(int x) { return x >> 1; }  // arithmetic shift == python x//2

__device__ __forceinline__ int lower_bound_i32(const int *a, int lo, int hi, int key) {
    // first index in [lo,hi) with a[idx] >= key
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Wave-wide reductions without the LDS crossbar: 4 DPP steps (quad xor 1, quad xor 2, half-row mirror, row mirror)
// leave every 16-lane row holding its row total; the four row totals are combined through v_readlane (SGPRs), so
// every lane ends with the same value.  (__shfl_xor lowers to ds_bpermute: ~100+ cycles per dependent step.)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_min(double v) {
    v = fmin(v, dpp_mov_f64<0xB1>(v));
    v = fmin(v, dpp_mov_f64<0x4E>(v));
    v = fmin(v, dpp_mov_f64<0x141>(v));
    v = fmin(v, dpp_mov_f64<0x140>(v));
    return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ int wave_or(int v) {      // OR of 4-bit flag words: per-bit ballots, no data movement at all
    int r = 0;
#pragma unroll
    for (int bit = 1; bit <= 8; bit <<= 1)
        if (__ballot((v & bit) != 0) != 0ull) r |= bit;
    return r;
}
// reductions inside each 16-lane row (the first four DPP steps of the wave reductions): every lane of a row ends with
// the row's result; the four rows of a wave work on four independent problems at once.
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double row_max(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    return v;
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row_min_i32(int v) {
    v = min(v, dpp_mov_i32<0xB1>(v));
    v = min(v, dpp_mov_i32<0x4E>(v));
    v = min(v, dpp_mov_i32<0x141>(v));
    v = min(v, dpp_mov_i32<0x140>(v));
    return v;
}
__device__ __forceinline__ int row_max_i32(int v) {
    v = max(v, dpp_mov_i32<0xB1>(v));
    v = max(v, dpp_mov_i32<0x4E>(v));
    v = max(v, dpp_mov_i32<0x141>(v));
    v = max(v, dpp_mov_i32<0x140>(v));
    return v;
}

// ------------------------------------------------------------------------------------------------
// fragment centres: c = l + (n-1)//2   (pyatac/fragments.pyx:36)
// ------------------------------------------------------------------------------------------------
__global__ void natac_frag_centres(const int *__restrict__ lpos, const int *__restrict__ ilen, int *__restrict__ centre,
                                   long long nf) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < nf; i += stride) centre[i] = lpos[i] + floor_half(ilen[i] - 1);
}

// E = exp(log-bias) of the whole batch, once per stage: the background, occupancy and candidate kernels all stage windows of
// it (a ~40-instruction exp per element otherwise repeated in every tile's halo)
__global__ void __launch_bounds__(256) natac_exp_bias(const double *__restrict__ b, double *__restrict__ e, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) e[i] = exp(b[i]);
}

// ------------------------------------------------------------------------------------------------
// K0  sparse V-plot gather: nuc_cov, nfr_cov, raw signal.
//   nuc_cov[g] = #{frag: vlower <= n < vupper, |c-g| <= w}          tracks.py:209-222 via NucleosomeCalling.py:257-260
//   nfr_cov[g] = #{frag: 0 <= n < vlower,      |c-g| <= w}          NucleosomeCalling.py:271-273
//   raw[g]     = sum_{frag: vlower<=n<vupper, |c-g|<=w} V[n-vlower, c-g+w]   NucleosomeCalling.py:29-36
// The reference builds the dense (upper x L) count matrix and correlates it with V (17,666 MAC/base);
// only ~F*W entries are non-zero, so one thread per base walks the (centre-sorted) fragments in its window.
// tile = (chunk, x0): 256 consecutive bases of one chunk.
// ------------------------------------------------------------------------------------------------
// number of leading entries of the sorted array a[from, n) that are < key, found 64 at a time with a ballot
__device__ __forceinline__ int advance_while_less(const int *a, int from, int n, int key, int lane) {
    int f = from;
    while (f < n) {
        const int i = f + lane;
        const int v = (i < n) ? a[i] : 0x7fffffff;
        const int cnt = __popcll(__ballot(v < key));
        f += cnt;
        if (cnt < WAVE) break;
    }
    return f;
}

// fragment range [t0, t1) of every 256-base tile: centres within [x0 - w, x0 + 255 + w]; one thread per tile
__global__ void __launch_bounds__(256) natac_tile_ranges256(ChunkTable ct, const int2 *__restrict__ tiles, int ntiles, int w,
                                                              int2 *__restrict__ ranges) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntiles) return;
    const int2 t = tiles[i];
    const int nfr = (int)(ct.frag_off[t.x + 1] - ct.frag_off[t.x]);
    const int *cen = ct.centre + ct.frag_off[t.x];
    const int t0 = lower_bound_i32(cen, 0, nfr, t.y - w);
    const int t1 = lower_bound_i32(cen, t0, nfr, t.y + 255 + w + 1);
    ranges[i] = make_int2(t0, t1);
}

// occ_cov != nullptr: also writes nuc_cov + nfr_cov (OccChunk.getCov, Occupancy.py:221-224, when the occupancy window and size
// range coincide with the V-plot's: exact integers).
//
// One wave = 128 consecutive bases, two adjacent ones per lane (a fragment is visited by (121 + 128) / 128 = 1.9 waves instead
// of 2.9 with 64).  The fragments that can touch them (centre-sorted, ~60 on configs[2]) are taken 64 at a time, one per
// lane, and compacted by size class through a per-wave LDS strip (ballot + mbcnt); their centres / template row offsets then
// reach all lanes as SCALARS (v_readlane with the loop counter), eight nucleosome-sized fragments per trip, so eight
// template reads -- the two adjacent columns of the lane's bases, 16 bytes -- are in flight before the first value is
// added.  The rows come from a copy of the template with VPAD zero columns on both sides: a base outside a fragment's
// window reads a zero instead of being masked out, and list tails are padded with a fragment whose window lies left of the
// wave (zeros again).  The additions per base keep the list (= centre) order; + 0.0 leaves the sum's bits alone.  Round 1's
// loop walked one fragment per trip through generic pointers (flat loads) with a dependent L2 read each: ~700 cycles per
// fragment and wave, 5.9 ms per launch -- latency- and not, as first thought, L2-bandwidth-bound.
template <int NBL>
struct __attribute__((packed, aligned(8))) GatherCols { double v[NBL]; };   // columns d - NBL + 1 .. d of a padded template row

// NBL adjacent bases per lane; tile = 256 bases = 256 / (64 NBL) waves
template <int NBL>
__global__ void __launch_bounds__(256 / NBL) natac_frag_gather(ChunkTable ct, const int2 *__restrict__ tiles,
                                                                 const int2 *__restrict__ ranges, VMatDev vm,
                                                                 double *__restrict__ nuc_cov, double *__restrict__ nfr_cov,
                                                                 double *__restrict__ raw, double *__restrict__ occ_cov) {
    constexpr int WB = 64 * NBL;                               // bases per wave
    static_assert(NBL - 1 <= VPAD, "template padding");          // lanes that straddle a window's edge read NBL - 1 padding columns at most
    __shared__ int strip_s[256 / WB][3][64];
    const int2 t = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunk = t.x, gw0 = t.y + WB * wv, g = gw0 + NBL * lane;     // the lane's bases: g .. g + NBL - 1
    const int L = ct.chunk_len[chunk];
    const int2 tr = ranges[blockIdx.x];
    const int nt = tr.y - tr.x;
    const int *__restrict__ cen = ct.centre + ct.frag_off[chunk] + tr.x;
    const int *__restrict__ iln = ct.ilen + ct.frag_off[chunk] + tr.x;
    if (gw0 >= L) return;                                      // wave-uniform
    const double *__restrict__ matp = vm.matp;
    const int W = vm.W, WP = W + 2 * VPAD, lower = vm.lower, upper = vm.upper;
    const int goff = vm.w - g;                                 // column of base g in a fragment's template row: c + goff (g + b: b less)
    const int clo = gw0 - vm.w, chi = gw0 + WB - 1 + vm.w;     // centres that can touch the wave's bases
    const int cfill = clo - 1;                                 // a centre whose window ends left of the wave: columns -WB .. -1
    int *sc = strip_s[wv][0], *so = strip_s[wv][1], *sf = strip_s[wv][2];
    int cnt_nuc[NBL], cnt_nfr[NBL];
    double acc[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) { cnt_nuc[b] = 0; cnt_nfr[b] = 0; acc[b] = 0.0; }
    // all fragments of the tile's range (those of all its waves; no per-wave search: its dependent loads cost more than
    // filtering the other waves' fragments out of the batch)
    int nxc = cfill, nxn = -1;                                 // the batch after the current one is already requested
    if (lane < nt) { nxc = cen[lane]; nxn = iln[lane]; }
    for (int base = 0; base < nt; base += 64) {
        const int myc = nxc, myn = nxn;
        const int fi = base + 64 + lane;
        nxc = cfill; nxn = -1;
        if (fi < nt) { nxc = cen[fi]; nxn = iln[fi]; }
        const bool mine = myc >= clo && myc <= chi;
        const bool isnuc = mine && myn >= lower && myn < upper, isnfr = mine && myn >= 0 && myn < lower;
        const unsigned long long mnuc = __ballot(isnuc), mnfr = __ballot(isnfr);
        const int nn = __popcll(mnuc), nf = __popcll(mnfr);
        if (nn + nf == 0) continue;                            // wave-uniform
        sc[lane] = cfill; so[lane] = VPAD; sf[lane] = cfill;      // tails: row 0, its left zero columns
        __builtin_amdgcn_wave_barrier();
        if (isnuc) {
            const int p = __builtin_amdgcn_mbcnt_hi((unsigned)(mnuc >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mnuc, 0u));
            sc[p] = myc;
            so[p] = (myn - lower) * WP + VPAD;
        }
        if (isnfr) sf[__builtin_amdgcn_mbcnt_hi((unsigned)(mnfr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mnfr, 0u))] = myc;
        __builtin_amdgcn_wave_barrier();
        const int cc = sc[lane], oo = so[lane], cf = sf[lane];
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < nf; ++i) {                         // short fragments: coverage only
            const int d = __builtin_amdgcn_readlane(cf, i) + goff;
#pragma unroll
            for (int b = 0; b < NBL; ++b) cnt_nfr[b] += ((unsigned)(d - b) < (unsigned)W) ? 1 : 0;
        }
        constexpr int NQ = 16 / NBL;                           // reads in flight per trip (16 columns per lane)
        for (int i = 0; i < nn; i += NQ) {                     // nucleosome-sized: coverage + template values
            GatherCols<NBL> v[NQ];
            int d[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                d[q] = __builtin_amdgcn_readlane(cc, i + q) + goff;
                // byte offset in 32 bits (the padded template is < 4 GB): the load takes the table's scalar base + this offset, no
                // sign extension and 64-bit address arithmetic per fragment
                unsigned off = (unsigned)(__builtin_amdgcn_readlane(oo, i + q) + d[q] - (NBL - 1)) * (unsigned)sizeof(double);
                // a lane whose NBL columns all lie outside the fragment's window reads zeros: from ONE place (the table's first bytes
                // are padding) instead of the row's own padding -- those lanes are half of every visit (121 of ~249 columns matter),
                // and the rows' paddings are 292 KB of zeros that had to come out of L2 like the values
                if ((unsigned)d[q] >= (unsigned)(W + NBL - 1)) off = 0u;
                v[q] = *(const GatherCols<NBL> *)((const char *)matp + off);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int b = 0; b < NBL; ++b) {
                    cnt_nuc[b] += ((unsigned)(d[q] - b) < (unsigned)W) ? 1 : 0;
                    acc[b] += v[q].v[NBL - 1 - b];
                }
            }
        }
    }
    const long long o = ct.out_off[chunk] + g;
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
        if (g + b < L) {
            nuc_cov[o + b] = (double)cnt_nuc[b];
            nfr_cov[o + b] = (double)cnt_nfr[b];
            raw[o + b] = acc[b];
            if (occ_cov) occ_cov[o + b] = (double)(cnt_nuc[b] + cnt_nfr[b]);
        }
    }
}

// occ coverage when the occupancy window / size range coincide with the V-plot's: cov = nuc_cov + nfr_cov (exact integers)
__global__ void __launch_bounds__(256) natac_add_tracks(const double *__restrict__ a, const double *__restrict__ b,
                                                          double *__restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = a[i] + b[i];
}

// ------------------------------------------------------------------------------------------------
// K1  dense background: the dominant kernel.
//   B[i,x]   = sizes[i] * E[x-(i-1)//2] * E[x+i//2],  E = exp(log-bias)      chunkmat2d.py:140-156
//   num[g]   = sum_{r<R} sum_{c<W} B[lower+r, g-w+c] * V[r,c]                 NucleosomeCalling.py:60-63
//   covB[g]  = sum_{r<R} sum_{c<W} B[lower+r, g-w+c]                          NucleosomeCalling.py:56-58
//   bg[g]    = num[g] * nuc_cov[g] / covB[g];  norm[g] = raw[g] - bg[g]       NucleosomeCalling.py:64, 38-43
// 17,666 fp64 FMA per base.  The reference materialises B (4.7 MB / chunk); here B never exists:
//   * one 64-lane workgroup owns a tile of TW = 64*G consecutive bases of one chunk;
//   * E for the tile (+halo) is computed once into LDS;
//   * per V-plot row r the product row P_r[u] = s_r E[.] E[.] (TW+2w values) is built in LDS (1/121 of the work),
//     its running column sum Q[u] = sum_r P_r[u] stays in registers (gives covB for free);
//   * each lane then slides over P_r for its G consecutive outputs: one ds_read_b64 feeds G FMAs whose
//     V[r,c] operand is wave-uniform (scalar loads -> SGPR operand of v_fma_f64);
//   * G is odd so that the lane stride (2G dwords) is conflict-free on the 64-bank LDS without padding.
// fp64 VALU bound (arithmetic intensity ~440 flop/B); MFMA is not applicable (matrix-vector shaped,
// and fp64 MFMA has no rate advantage on gfx950).
// Template W: fast path for the default V-plot width 121; natac_background_generic handles any other width.
// ------------------------------------------------------------------------------------------------
template <int G, int W>
__global__ void __launch_bounds__(64) natac_background(ChunkTable ct, const int2 *__restrict__ tiles, VMatDev vm,
                                                         const double *__restrict__ nuc_cov, const double *__restrict__ raw,
                                                         double *__restrict__ bg, double *__restrict__ norm,
                                                         double *__restrict__ bnum, double *__restrict__ bcov) {
    constexpr int TW = WAVE * G;
    constexpr int HW = W / 2;
    constexpr int PW = TW + W - 1;               // product-row length
    constexpr int NQ = (PW + WAVE - 1) / WAVE;   // product elements per lane
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int A = (vm.upper - 2) >> 1;           // max left half-length  (i-1)//2, i = upper-1
    const int Bh = (vm.upper - 1) >> 1;          // max right half-length i//2
    const int EW = PW + A + Bh;
    double *Et = smem;                           // [EW]
    double *Pb = smem + ((EW + 1) & ~1);         // [PW]

    // --- E tile: coordinate of Et[u] is x0 - HW - A + u
    {
        const double *b = ct.bias ? ct.bias + ct.bias_off[chunk] : nullptr;
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = x0 - HW - A + ct.bias_left;
        for (int u = lane; u < EW; u += WAVE) {
            const int j = j0 + u;
            double e = 1.0;
            if (b) e = (j >= 0 && j < nb) ? exp(b[j]) : 0.0;
            Et[u] = e;
        }
    }
    __syncthreads();

    double acc[G];
    double q[NQ];
#pragma unroll
    for (int k = 0; k < G; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = 0.0;

    const int ub = lane * G;
    for (int r = 0; r < vm.R; ++r) {
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const double s = vm.srow[r];
        const double *el = Et + (A - hl);
        const double *er = Et + (A + hr);
        // phase A: product row (i == 1 would be a single-cell row; rows of a V-plot start far above 1,
        // the host rejects lower < 2 for this kernel)
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int u = lane + WAVE * k;
            if (u < PW) {
                const double p = (s * el[u]) * er[u];
                q[k] += p;
                Pb[u] = p;
            }
        }
        __syncthreads();
        // phase B: sliding FMA; V row operand is wave-uniform
        const double *__restrict__ vr = vm.mat + r * W;
        const double *pl = Pb + ub;
#pragma unroll
        for (int j = 0; j < G + W - 1; ++j) {
            const double p = pl[j];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int c = j - k;
                if (c >= 0 && c < W) acc[k] = fma(p, vr[c], acc[k]);
            }
        }
        __syncthreads();
    }
    // --- covB: box sum of Q over W columns
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int u = lane + WAVE * k;
        if (u < PW) Pb[u] = q[k];
    }
    __syncthreads();
    const long long ob = ct.out_off[chunk];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int g = x0 + ub + k;
        double cv = 0.0;
        for (int c = 0; c < W; ++c) cv += Pb[ub + k + c];
        if (g < L) {
            const long long o = ob + g;
            const double b = (acc[k] * nuc_cov[o]) / cv;
            bg[o] = b;
            norm[o] = raw[o] - b;
            bnum[o] = acc[k];        // sum B V and sum B of the window at this base: reused by the candidate statistics
            bcov[o] = cv;
        }
    }
}

// generic-width fallback (any W, any lower >= 0): one thread per base, E read through exp() each time.
__global__ void __launch_bounds__(256) natac_background_generic(ChunkTable ct, const int2 *__restrict__ tiles, VMatDev vm,
                                                                  const double *__restrict__ nuc_cov,
                                                                  const double *__restrict__ raw, double *__restrict__ bg,
                                                                  double *__restrict__ norm, double *__restrict__ bnum,
                                                                  double *__restrict__ bcov) {
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, g = t.y + threadIdx.x;
    const int L = ct.chunk_len[chunk];
    if (g >= L) return;
    const double *b = ct.bias ? ct.bias + ct.bias_off[chunk] + ct.bias_left : nullptr;
    double num = 0.0, cov = 0.0;
    for (int r = 0; r < vm.R; ++r) {
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const double s = vm.srow[r];
        for (int c = 0; c < vm.W; ++c) {
            const int x = g - vm.w + c;
            double p = s;
            if (b) p = (hl == -hr) ? s * exp(b[x]) : (s * exp(b[x - hl])) * exp(b[x + hr]);
            num = fma(p, vm.mat[r * vm.W + c], num);
            cov += p;
        }
    }
    const long long o = ct.out_off[chunk] + g;
    const double v = (num * nuc_cov[o]) / cov;
    bg[o] = v;
    norm[o] = raw[o] - v;
    bnum[o] = num;
    bcov[o] = cov;
}

// ------------------------------------------------------------------------------------------------
// K2 / K4  NaN-aware Gaussian smoothing, numpy 'same' alignment (pyatac/utils.py:23-52):
//   y[t] = sum_n w[n] x0[t-h+n] / sum_n w[n] ok[t-h+n],  h = (M-1)/2, x0 = x with NaN->0, zero padded;
//   denominator 0 -> NaN.   CLAMP: x<0 -> 0 first (NucleosomeCalling.py:280).
// tile = 256 bases of one chunk; input tile (+halo) staged in LDS.
// ------------------------------------------------------------------------------------------------
template <bool CLAMP>
__global__ void __launch_bounds__(256) natac_smooth_same(ChunkTable ct, const int2 *__restrict__ tiles,
                                                           const double *__restrict__ win, int M, double win_sum,
                                                           const double *__restrict__ x, double *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int any_gap;
    const int h = (M - 1) / 2;
    double *xs = smem;               // [256 + 2h]  value (NaN -> 0)
    double *ok = smem + 256 + 2 * h; // [256 + 2h]  1 / 0
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const long long ob = ct.out_off[chunk];
    if (threadIdx.x == 0) any_gap = 0;
    __syncthreads();
    bool gap = false;
    for (int u = threadIdx.x; u < 256 + 2 * h; u += 256) {
        const int g = x0 - h + u;
        double v = 0.0, o = 0.0;
        if (g >= 0 && g < L) {
            v = x[ob + g];
            if (v != v) { v = 0.0; } else { o = 1.0; if (CLAMP && v < 0) v = 0.0; }
        }
        xs[u] = v;
        ok[u] = o;
        gap |= (o == 0.0);
    }
    if (gap) any_gap = 1;
    __syncthreads();
    const int g = x0 + threadIdx.x;
    if (g >= L) return;
    // the window values are wave-uniform: scalar loads, SGPR operands of the FMAs (no LDS traffic for them)
    double num = 0.0, den;
    if (!any_gap) {
        // no NaN and no chunk edge under any window of the tile: the denominator is the plain sum of the window, formed on
        // the host in the same order (fma(w, 1, den) == den + w), so the quotient has the same bits as the general path
        for (int n = 0; n < M; ++n) num = fma(win[n], xs[threadIdx.x + 2 * h - n], num);
        den = win_sum;
    } else {
        den = 0.0;
        for (int n = 0; n < M; ++n) {
            // np.convolve(w, x)[t+h] = sum_k w[k] x[t+h-k]
            const double wv = win[n];
            const int u = threadIdx.x + 2 * h - n;
            num = fma(wv, xs[u], num);
            den = fma(wv, ok[u], den);
        }
    }
    y[ob + g] = (den == 0.0) ? __builtin_nan("") : num / den;
}

// The same smoothing with four consecutive bases per lane for the window M = 2 H + 1 (compile time): a lane reads its
// 4 + 2 H inputs once (16 LDS reads per base instead of 61; LDS holds the tile as four phase arrays so that lanes read
// consecutive addresses) and every term order is natac_smooth_same's: identical bits.  The NaN-aware denominator is formed
// per WAVE: only waves whose bases see a NaN or a chunk edge under their windows pay for it.
// tile = (chunk, x0), x0 a multiple of 1024.  Dynamic LDS: 2 x 4 x SM4_S(H) doubles.
__host__ __device__ constexpr int SM4_S(int H) { return (1024 + 2 * H) / 4 + 2; }

template <bool CLAMP, int H>
__global__ void __launch_bounds__(256) natac_smooth_same4(ChunkTable ct, const int2 *__restrict__ tiles,
                                                            const double *__restrict__ win, double win_sum,
                                                            const double *__restrict__ x, double *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int gap_s[4];
    constexpr int M = 2 * H + 1, S = SM4_S(H), NU = 1024 + 2 * H;
    double *xs = smem;               // value (NaN -> 0): element u (base x0 - H + u) at (u & 3) S + (u >> 2)
    double *ok = smem + 4 * S;       // 1 / 0, same layout
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const long long ob = ct.out_off[chunk];
    if (threadIdx.x < 4) gap_s[threadIdx.x] = 0;
    __syncthreads();
    for (int u = threadIdx.x; u < NU; u += 256) {
        const int g = x0 - H + u;
        double v = 0.0, o = 0.0;
        if (g >= 0 && g < L) {
            v = x[ob + g];
            if (v != v) { v = 0.0; } else { o = 1.0; if (CLAMP && v < 0) v = 0.0; }
        }
        xs[(u & 3) * S + (u >> 2)] = v;
        ok[(u & 3) * S + (u >> 2)] = o;
        if (o == 0.0) {              // element u lies under the windows of the tile's bases u - 2H .. u (beyond L: never written)
            const int w0 = max(u - 2 * H, 0) >> 8, w1 = min(u, 1023) >> 8;
            if (g < L + H) { gap_s[w0] = 1; gap_s[w1] = 1; }
        }
    }
    __syncthreads();
    const int tid = threadIdx.x, g0 = x0 + 4 * tid;
    if (g0 >= L) return;
    const bool gap = gap_s[tid >> 6] != 0;          // wave-uniform
    double num[4] = {0.0, 0.0, 0.0, 0.0}, den[4] = {win_sum, win_sum, win_sum, win_sum};
    if (!gap) {
#pragma unroll
        for (int kk = 0; kk < 2 * H + 4; ++kk) {
            const int k = 2 * H + 3 - kk;               // inputs in descending order = taps n ascending for every output
            const double r = xs[(k & 3) * S + tid + (k >> 2)];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int n = o + 2 * H - k;
                if (n >= 0 && n < M) num[o] = fma(win[n], r, num[o]);
            }
        }
    } else {
#pragma unroll
        for (int o = 0; o < 4; ++o) den[o] = 0.0;
#pragma unroll
        for (int kk = 0; kk < 2 * H + 4; ++kk) {
            const int k = 2 * H + 3 - kk;
            const double r = xs[(k & 3) * S + tid + (k >> 2)], q = ok[(k & 3) * S + tid + (k >> 2)];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int n = o + 2 * H - k;
                if (n >= 0 && n < M) { num[o] = fma(win[n], r, num[o]); den[o] = fma(win[n], q, den[o]); }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (g0 + o < L) y[ob + g0 + o] = (den[o] == 0.0) ? __builtin_nan("") : num[o] / den[o];
}

// ------------------------------------------------------------------------------------------------
// K3  occupancy grid MLE (nucleoatac/Occupancy.py:104-146).
// For grid point k of a chunk (base g = halfstep + k*step):
//   ins[j]  = #{frag: n == j, |c-g| <= flank}                 j in [0, upper)
//   bias[j] = sum_{|d|<=flank} B0[j, g+d],  B0[j,x] = E[x-(j-1)//2] E[x+j//2]  (j == 1: E[x])
//   pn = nuc_probs*bias / sum,  pf = nfr_probs*bias / sum
//   ll[a] = sum_j ins[j] log(alpha_a pn[j] + (1-alpha_a) pf[j]);  NaN -> -inf
//   occ = alpha[argmax ll] (first max), lower/upper = min/max alpha with 2(max-ll) < cutoff
//   only if sum(ins) > 0, else the three outputs stay NaN.
// Workgroup = 256 threads = tile of T grid points:
//   phase 1: thread j owns insert size j and slides the flank window along the tile (products of LDS-staged E),
//            writing bias[k][j] into LDS;
//   phase 2: one wave per grid point; lanes own alphas (a = lane, lane+64).  The log-likelihood is accumulated as
//            a running PRODUCT with frexp renormalisation (mantissa product + integer exponent), so each
//            (fragment, alpha) costs a handful of fp64 ops instead of a log(); one log per alpha at the end.
//            `0 * log(0) = NaN -> -inf` of the reference is reproduced with the zero-probability flags.
// ------------------------------------------------------------------------------------------------
constexpr int OCC_T = 16;    // grid points per pass (one 16-lane row or one wave each in phase 2)
constexpr int OCC_NP = 4;    // passes per tile
constexpr int OCC_FMAX = 512;    // fragments of a tile staged in LDS (larger tiles read them from global memory)

struct OccModelDev {
    const double *nuc_probs, *nfr_probs, *alphas;
    int upper, n_alpha, step, halfstep, flank;
    double cutoff;
    double ci_factor;   // exp(-cutoff / 2): likelihood-ratio threshold in the product domain
    int zero_flags;     // host-known zero pattern of the model: 1 some nuc_prob == 0, 2 some nfr_prob == 0, 4 some both
    double b_floor;     // window bias sums >= b_floor cannot make a non-zero probability underflow to 0 (else: exact flag loop)
};

// fragment range [t0, t1) of every occupancy tile (centres within the tile's windows); one thread per tile so the
// dependent binary-search loads are hidden by occupancy instead of stalling a whole MLE workgroup.
__global__ void __launch_bounds__(256) natac_occ_tile_ranges(ChunkTable ct, const int2 *__restrict__ tiles, int ntiles,
                                                               int step, int halfstep, int flank, int2 *__restrict__ ranges) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntiles) return;
    const int2 t = tiles[i];
    const int nfr = (int)(ct.frag_off[t.x + 1] - ct.frag_off[t.x]);
    const int *cen = ct.centre + ct.frag_off[t.x];
    const int gfirst = halfstep + t.y * step;
    const int t0 = lower_bound_i32(cen, 0, nfr, gfirst - flank);
    const int t1 = lower_bound_i32(cen, t0, nfr, gfirst + (OCC_T * OCC_NP - 1) * step + flank + 1);
    ranges[i] = make_int2(t0, t1);
}

// Tiles whose cost follows their fragment count (heavy-tailed peak sets: a few windows hold ten times the reads of the rest): the
// HEAVY ones -- more fragments than `thr` (the host: a multiple of the batch's mean) -- are listed so that a launch can visit them
// first; their long waves then run beside the bulk instead of after it.  Everything else keeps the chunk order (neighbouring tiles
// share cache lines of the fragment list and of the block sums: a full sort by count measured 2 % slower on a uniform batch).
// head[0] = number of listed tiles (<= HEAVY_CAP; more stay in chunk order), list[], flag[tile] = 1 for the listed ones.
constexpr int HEAVY_CAP = 65536;
__global__ void __launch_bounds__(256) natac_tile_heavy(const int2 *__restrict__ ranges, int n, int thr, int *__restrict__ head,
                                                          int *__restrict__ list, unsigned char *__restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int2 r = ranges[i];
    unsigned char f = 0;
    if (r.y - r.x > thr) {
        const int k = atomicAdd(head + 1, 1);        // head[1]: claims (may pass the cap), head[0] = min(claims, cap) below
        if (k < HEAVY_CAP) { list[k] = i; f = 1; atomicMax(head, k + 1); }
    }
    flag[i] = f;
}

// ---- phase 2 of natac_occ_mle, row-parallel form: each 16-lane row of a wave owns one grid point (4 per wave, the 16 of
// a tile in one pass) and each lane 7 of the <= 112 alphas (a = l + 16 t).  The per-grid-point overheads of the
// wave-per-grid-point form (normaliser reductions, window search, per-fragment probabilities, the argmax / interval
// decision) are paid once per four grid points; the multiply chain  L(alpha) = prod_f (c_f + alpha (a_f - c_f))  costs the
// same lane work as before.  One product chain per alpha (7 independent chains per lane hide the fp64 latency), rescaled
// by an exact power of two after every 4th factor (every factor when a probability < 2^-200 is present).
constexpr int OCC_RA = 7;     // alphas per lane in the row-parallel form
constexpr int OCC_AC_STRIDE = 34;   // doubles per row list: 16 x 2 + 2, so the four rows' broadcast reads hit different banks
constexpr int OCC_ACL = 16 * OCC_AC_STRIDE;   // >= 4 waves x 2 x 64 of the wave-per-point form
__device__ __forceinline__ void occ_phase2_rows(const ChunkTable &ct, const OccModelDev &om, int chunk, int k0, int nk, int step, int fl,
                                                int gfirst, int U, int UP, const double *bw, const double *nucp, const double *nfrp,
                                                double *acl, const int *cs, const int *is, int nt, double *__restrict__ g_occ,
                                                double *__restrict__ g_lo, double *__restrict__ g_hi, int *__restrict__ status) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15;
    // rows of one 32-lane half take grid points 4 apart: their bw rows (stride UP doubles) then sit 128 B apart modulo the
    // 256-B bank span, so the 16 + 16 lanes of a half read 64 distinct banks
    const int kk = wave + 4 * row, k = k0 + kk;
    const bool live = k < nk;                                   // row-uniform
    const int sh = 16 * row;
    const double *bj = bw + kk * UP;
    double al[OCC_RA];
#pragma unroll
    for (int t = 0; t < OCC_RA; ++t) { const int a = l + 16 * t; al[t] = (a < om.n_alpha) ? om.alphas[a] : 0.0; }
    // normalisers of  nuc_probs * bias / sum(...)  (Occupancy.py:108-111) + zero / NaN flags of the quotients:
    // 1: some pn == 0, 2: some pf == 0, 4: some both == 0, 8: some NaN.  A product nuc_prob * bias is 0 only if a factor is
    // (the zero pattern of the model is known on the host) or if it underflows, which needs a bias sum below om.b_floor;
    // the per-element tests run only for waves that see such a sum.
    double sn = 0.0, sf = 0.0, bmin = __builtin_inf();
    for (int j = l; j < U; j += 16) {
        const double b = bj[j];
        sn += nucp[j] * b;
        sf += nfrp[j] * b;
        bmin = fmin(bmin, b);
    }
    sn = row_sum(sn);
    sf = row_sum(sf);
    int flags = om.zero_flags;
    if (__ballot(!(bmin >= om.b_floor)) != 0ull) {            // wave-uniform, rare: exact per-element flags
        int lf = 0;
        for (int j = l; j < U; j += 16) {
            const double b = bj[j];
            const double pa = nucp[j] * b, pc = nfrp[j] * b;
            if (pa == 0.0) lf |= 1;
            if (pc == 0.0) lf |= 2;
            if (pa == 0.0 && pc == 0.0) lf |= 4;
            if (pa != pa || pc != pc) lf |= 8;
        }
        flags = 0;
#pragma unroll
        for (int bit = 1; bit <= 8; bit <<= 1)
            if (((__ballot((lf & bit) != 0) >> sh) & 0xffffull) != 0ull) flags |= bit;
    }
    if (!(sn > 0.0 && sn < __builtin_inf() && sf > 0.0 && sf < __builtin_inf())) flags |= 8;  // 0/0, x/inf, NaN sums
    // fragments of the row's window [g-fl, g+fl]: ranks of the two keys in the tile's sorted centre list
    int f0 = 0, f1 = 0;
    {
        const int kmax = gfirst + (wave + 12) * step + fl + 1;
        for (int base = 0; base < nt; base += WAVE) {
            const int i = base + lane;
            const int v = (i < nt) ? cs[i] : 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = gfirst + (wave + 4 * r) * step;
                const int c0 = __popcll(__ballot(v < gr - fl)), c1 = __popcll(__ballot(v < gr + fl + 1));
                if (row == r) { f0 += c0; f1 += c1; }
            }
            if (__ballot(v < kmax) != ~0ull) break;             // the rest of the list lies right of every window
        }
    }
    const int cnt_row = f1 - f0;
    const int maxcnt = max(max(__builtin_amdgcn_readlane(cnt_row, 0), __builtin_amdgcn_readlane(cnt_row, 16)),
                           max(__builtin_amdgcn_readlane(cnt_row, 32), __builtin_amdgcn_readlane(cnt_row, 48)));
    double m[OCC_RA];
    int e[OCC_RA];
#pragma unroll
    for (int t = 0; t < OCC_RA; ++t) { m[t] = 1.0; e[t] = 0; }
    int nins = 0;
    double *ac = acl + (wave * 4 + row) * OCC_AC_STRIDE;        // 16 x (a - c, c) of the row's current batch
    for (int b0 = 0; b0 < maxcnt; b0 += 16) {
        int n = -1;
        if (b0 + l < cnt_row) n = is[f0 + b0 + l];
        const bool ok = (n >= 0 && n < U);
        const unsigned long long bal = __ballot(ok);
        const unsigned rowmask = (unsigned)(bal >> sh) & 0xffffu;
        nins += __popc(rowmask);
        ac[2 * l] = 0.0;                                        // neutral factor (x = 1 for every alpha) in unused slots
        ac[2 * l + 1] = 1.0;
        __builtin_amdgcn_wave_barrier();
        bool tiny = false;
        if (ok) {                                               // lane-parallel: one fragment per lane, compacted per row
            const double b = bj[n];
            const int pos = __popc(rowmask & ((1u << l) - 1u));
            const double a = (nucp[n] * b) / sn, c = (nfrp[n] * b) / sf;
            ac[2 * pos] = a - c;                                // mixture alpha a + (1 - alpha) c evaluated as fma(alpha, a - c, c)
            ac[2 * pos + 1] = c;
            tiny = !(a >= 0x1p-200 && c >= 0x1p-200);
        }
        const bool safe = __ballot(tiny) == 0ull;
        __builtin_amdgcn_wave_barrier();
        const int mc = max(max(__popc((unsigned)bal & 0xffffu), __popc((unsigned)(bal >> 16) & 0xffffu)),
                           max(__popc((unsigned)(bal >> 32) & 0xffffu), __popc((unsigned)(bal >> 48) & 0xffffu)));
        if (safe) {
            for (int q = 0; q < mc; q += 4) {                   // slots >= the row's count hold the neutral factor
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double d = ac[2 * (q + v)], c = ac[2 * (q + v) + 1];
#pragma unroll
                    for (int t = 0; t < OCC_RA; ++t) m[t] *= fma(al[t], d, c);
                }
#pragma unroll
                for (int t = 0; t < OCC_RA; ++t) { int ex; m[t] = frexp(m[t], &ex); e[t] += ex; }
            }
        } else {
            for (int q = 0; q < mc; ++q) {
                const double d = ac[2 * q], c = ac[2 * q + 1];
#pragma unroll
                for (int t = 0; t < OCC_RA; ++t) { int ex; m[t] = frexp(m[t] * fma(al[t], d, c), &ex); e[t] += ex; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // decision without logarithms (see natac_occ_mle): L = m * 2^e, m in [0.5, 1), 0 stands for log L = -inf
    const int ENONE = -(1 << 30);
    int lemax = ENONE;
    const bool anyflag = __ballot(flags != 0) != 0ull;
#pragma unroll
    for (int t = 0; t < OCC_RA; ++t) {
        int ex;
        m[t] = frexp(m[t], &ex);
        e[t] += ex;
        if (anyflag) {                                                     // wave-uniform
            const double be = 1 - al[t];
            if (flags & (8 | 4)) m[t] = 0.0;
            if ((flags & 2) && al[t] == 0.0) m[t] = 0.0;
            if ((flags & 1) && be == 0.0) m[t] = 0.0;
        }
        if (!(m[t] > 0.0) || l + 16 * t >= om.n_alpha) m[t] = 0.0;      // NaN likelihood -> -inf as well
        if (!(m[t] > 0.0)) e[t] = ENONE;
        lemax = max(lemax, e[t]);
    }
    const int iemax = row_max_i32(lemax);
    const bool none = (iemax == ENONE);
    double lmmax = 0.0;
#pragma unroll
    for (int t = 0; t < OCC_RA; ++t) lmmax = fmax(lmmax, e[t] == iemax ? m[t] : 0.0);
    const double mmax = row_max(lmmax);
    const double thr = mmax * om.ci_factor;
    int limax = 0x7fffffff, lilo = 0x7fffffff, lihi = -1;
#pragma unroll
    for (int t = OCC_RA - 1; t >= 0; --t) {
        const int a = l + 16 * t;
        if (e[t] == iemax && m[t] == mmax) limax = a;
        const int d = e[t] - iemax;
        const bool in = m[t] > 0.0 && d > -1100 && ldexp(m[t], d) > thr;
        if (in) { lilo = a; lihi = max(lihi, a); }
    }
    const int imax = row_min_i32(limax), ilo = row_min_i32(lilo), ihi = row_max_i32(lihi);
    if (live && l == 0) {
        const long long go = ct.grid_off[chunk] + k;
        const double qn = __builtin_nan("");
        if (nins == 0) {                         // sum(new_inserts) > 0 fails: stay NaN (Occupancy.py:143)
            g_occ[go] = qn; g_lo[go] = qn; g_hi[go] = qn;
        } else if (none) {
            // every likelihood is -inf: the reference raises ValueError (min of empty, Occupancy.py:118)
            g_occ[go] = qn; g_lo[go] = qn; g_hi[go] = qn;
            atomicOr(&status[chunk], 1);
        } else {
            g_occ[go] = om.alphas[imax];
            g_lo[go] = om.alphas[ilo];
            g_hi[go] = om.alphas[ihi];
        }
    }
}

// STEP/FLANK > 0: compile-time fast path (defaults 5 / 60, requires (2*FLANK) % STEP == 0); STEP == 0: runtime values.
// A tile is OCC_T * OCC_NP consecutive grid points of one chunk, processed in OCC_NP passes of OCC_T: the exp(bias) window
// and the fragments are staged once per tile, and in the fast path the block sums of the window products slide on from
// pass to pass in registers (80 new products per insert size and pass instead of 196).
template <int STEP, int FLANK, int ABL = 0, int ROWS = 0>   // ABL != 0: ablation variants for tools/microbench_occ.hip only;
                                                          // ROWS: row-parallel phase 2 (n_alpha <= 16 * OCC_RA)
// tile_list != nullptr: the workgroups walk the first *tile_count entries of tile_list (tiles deferred by the fast path,
// natac_occ_fast.hpp) instead of taking tile blockIdx.x.
// three waves per SIMD with 96-204 B of scratch beat two waves without spills (general path on the configs[2] step: 54.2 vs 64.8 ms)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) natac_occ_mle(ChunkTable ct, const int2 *__restrict__ tiles,
                                                       const int2 *__restrict__ ranges, OccModelDev om,
                                                       double *__restrict__ g_occ, double *__restrict__ g_lo,
                                                       double *__restrict__ g_hi, int *__restrict__ status,
                                                       const int *__restrict__ tile_list, const int *__restrict__ tile_count) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n_listed = tile_list ? *tile_count : 0;
  for (int tile_it = blockIdx.x; tile_list ? tile_it < n_listed : tile_it == (int)blockIdx.x; tile_it += gridDim.x) {
    const int tile_id = tile_list ? tile_list[tile_it] : tile_it;
    __syncthreads();                                   // LDS of the previous tile is free
    const int U = om.upper, UP = (U + 1) & ~1;
    const int step = STEP ? STEP : om.step;
    const int fl = STEP ? FLANK : om.flank, WIN = 2 * fl + 1;
    const int A = (U - 2) >> 1, Bh = (U - 1) >> 1;
    const int span = (OCC_T * OCC_NP - 1) * step + WIN + step;   // centre positions covered by the tile (+ the last block in full)
    const int EW = span + A + Bh + 2;
    double *Et = smem;                                 // [EW]
    double *bw = smem + ((EW + 1) & ~1);               // [OCC_T][UP]
    double *nucp = bw + OCC_T * UP;                    // [UP]
    double *nfrp = nucp + UP;                          // [UP]
    double *acl = nfrp + UP;                           // [4 waves][64 x (pn, pf)] in phase 2
    double *ones = acl + OCC_ACL;                      // [n_ones] of 1.0: the right factor of the single-cell row j == 1
    const int n_ones = ((OCC_T - 1) * step + WIN + step + 3) & ~1;
    int *cen_s = (int *)(ones + n_ones);               // [OCC_FMAX]
    int *iln_s = cen_s + OCC_FMAX;                     // [OCC_FMAX]
    const int2 t = tiles[tile_id];
    const int chunk = t.x, k0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int nk = (L - om.halfstep + step - 1) / step;         // len(range(halfstep, L, step))
    const int gfirst = om.halfstep + k0 * step;                 // base of the tile's first grid point
    const int *cen = ct.centre + ct.frag_off[chunk];
    const int *iln = ct.ilen + ct.frag_off[chunk];
    // ---- phase 0: stage the tile's fragments, the model and exp(bias) in LDS
    const int2 tr = ranges[tile_id];
    const int t0 = tr.x, nt = tr.y - tr.x;
    const bool staged = nt <= OCC_FMAX;
    if (staged)
        for (int i = threadIdx.x; i < nt; i += 256) { cen_s[i] = cen[t0 + i]; iln_s[i] = iln[t0 + i]; }
    for (int j = threadIdx.x; j < U; j += 256) { nucp[j] = om.nuc_probs[j]; nfrp[j] = om.nfr_probs[j]; }
    for (int u = threadIdx.x; u < n_ones; u += 256) ones[u] = 1.0;
    {   // Et[u] <-> coordinate gfirst - fl - A + u
        const double *b = ct.bias ? ct.bias + ct.bias_off[chunk] : nullptr;
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = gfirst - fl - A + ct.bias_left;
        // only the part of the window that the chunk's remaining grid points reach
        const int need = min(EW, (min(OCC_T * OCC_NP, nk - k0) - 1) * step + WIN + A + Bh + 2);
        for (int u = threadIdx.x; u < EW; u += 256) {
            const int j = j0 + u;
            double e = 0.0;
            if (u < need) {
                e = 1.0;
                if (b) e = (j >= 0 && j < nb) ? exp(b[j]) : 0.0;
            }
            Et[u] = e;
        }
    }
    const int *cs = staged ? cen_s : cen + t0;   // window search + gather source (LDS copy, or global for huge tiles)
    const int *is = staged ? iln_s : iln + t0;
    constexpr int SS = STEP ? STEP : 1;
    constexpr int Q = STEP ? (2 * FLANK) / SS : 1;
    double blk[OCC_T + Q];                       // fast path: block sums of thread j's products, carried from pass to pass
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *ac = acl + wave * 2 * WAVE;
    const int a0 = lane, a1 = lane + WAVE;
    const double al0 = (a0 < om.n_alpha) ? om.alphas[a0] : 0.0;
    const double al1 = (a1 < om.n_alpha) ? om.alphas[a1] : 0.0;
    const double be0 = 1 - al0, be1 = 1 - al1;
    int f0 = 0, f1 = 0;
    for (int pass = 0; pass < OCC_NP; ++pass) {
        const int kbase = k0 + pass * OCC_T;
        if (kbase >= nk) break;                                  // block-uniform
        const int uoff = pass * OCC_T * step;                    // Et offset of the pass's first window
        __syncthreads();                                         // staging done / phase 2 of the previous pass is done with bw
        // ---- phase 1: window sums of B0 for every insert size j (thread j), the OCC_T grid points of the pass
        {
            const int j = threadIdx.x;
            if (ABL == 2) { for (int k = 0; k < OCC_T; ++k) if (j < U) bw[k * UP + j] = 1.0; }
            else if (j < U) {
                const int hl = floor_half(j - 1), hr = floor_half(j);
                const bool single = (hl == -hr);   // j == 1: the two pattern ones coincide (chunkmat2d.py:150-151)
                // branch-free: the j == 1 row multiplies by a row of ones (exact) instead of selecting per element
                const double *el = Et + (A - hl) + uoff;
                const double *er = single ? ones : Et + (A + hr) + uoff;
                if (STEP) {
                    // window k = columns [k*STEP, k*STEP + WIN), WIN = Q*STEP + 1: sum of Q aligned STEP-blocks + the first
                    // element of block k+Q.  Every product is formed once; block sums stay in registers (static indices).
                    if (pass > 0) {
#pragma unroll
                        for (int m = 0; m < Q; ++m) blk[m] = blk[m + OCC_T];
                    }
#pragma unroll
                    for (int m = 0; m < OCC_T + Q; ++m) {
                        if (m < Q && pass > 0) continue;               // carried over from the previous pass
                        double sacc = 0.0;
#pragma unroll
                        for (int d = 0; d < SS; ++d) {
                            const int u = m * SS + d;
                            sacc += (ABL == 7) ? (el[0] + (double)u) * er[0] : el[u] * er[u];
                        }
                        blk[m] = sacc;
                    }
                    double T = 0.0;
#pragma unroll
                    for (int m = 0; m < Q; ++m) T += blk[m];
#pragma unroll
                    for (int k = 0; k < OCC_T; ++k) {
                        // + the first element of block k + Q (formed again: cheaper than 16 more live registers)
                        bw[k * UP + j] = T + el[(k + Q) * SS] * er[(k + Q) * SS];
                        T = (T - blk[k]) + blk[k + Q];
                    }
                } else {
                    double S = 0.0;
                    for (int u = 0; u < WIN; ++u) S += el[u] * er[u];
                    bw[j] = S;
                    for (int k = 1; k < OCC_T; ++k) {
                        const int u0 = (k - 1) * step;
                        for (int d = 0; d < step; ++d) {
                            const int ua = u0 + d, ub = u0 + WIN + d;
                            S -= el[ua] * er[ua];
                            S += el[ub] * er[ub];
                        }
                        bw[k * UP + j] = S;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2
        if (ABL == 1) continue;
        if (ROWS) {
            occ_phase2_rows(ct, om, chunk, kbase, nk, step, fl, gfirst + uoff, U, UP, bw, nucp, nfrp, acl, cs, is, nt, g_occ, g_lo, g_hi,
                            status);
            continue;
        }
        for (int kk = wave; kk < OCC_T; kk += 4) {
            const int k = kbase + kk;
            if (k >= nk) break;                      // wave-uniform
            const int g = om.halfstep + k * step;
            const double *bj = bw + kk * UP;
            // normalisers of  nuc_probs * bias / sum(...)  (Occupancy.py:108-111) + zero / NaN flags of the quotients:
            // a quotient is 0 iff its product is 0 (sum finite, > 0) and NaN iff the product or the sum is.
            double sn = 0.0, sf = 0.0;
            int flags = 0;   // 1: some pn == 0, 2: some pf == 0, 4: some both == 0, 8: some NaN
            for (int j = lane; j < U && ABL != 6; j += WAVE) {
                const double b = bj[j];
                const double pa = nucp[j] * b, pc = nfrp[j] * b;
                sn += pa;
                sf += pc;
                if (pa == 0.0) flags |= 1;
                if (pc == 0.0) flags |= 2;
                if (pa == 0.0 && pc == 0.0) flags |= 4;
                if (pa != pa || pc != pc) flags |= 8;
            }
            if (ABL == 6) { sn = 1.0; sf = 1.0; }
            sn = wave_sum(sn);
            sf = wave_sum(sf);
            flags = wave_or(flags);
            if (!(sn > 0.0 && sn < __builtin_inf() && sf > 0.0 && sf < __builtin_inf())) flags |= 8;  // 0/0, x/inf, NaN sums
            // fragments of the window [g-fl, g+fl] (sorted by centre; windows only move right)
            f0 = advance_while_less(cs, f0, nt, g - fl, lane);
            f1 = advance_while_less(cs, (f1 > f0 ? f1 : f0), nt, g + fl + 1, lane);
            // log-likelihood as a running product (mantissa x 2^exponent), 4 independent chains per alpha.  frexp only
            // rescales by an exact power of two, so renormalising every 4th multiply instead of every multiply leaves the
            // mantissa bits unchanged as long as nothing underflows: safe when every factor is >= 2^-200 (then the product
            // of a mantissa in [0.5,1) and 4 factors stays >= 2^-801).  Windows holding a smaller probability take the
            // renormalise-every-multiply path.
            double m0[4] = {1.0, 1.0, 1.0, 1.0}, m1[4] = {1.0, 1.0, 1.0, 1.0};
            int e0 = 0, e1 = 0, nins = 0;
            for (int base = f0; base < f1 && ABL != 3; base += WAVE) {
                const int i = base + lane;
                int n = -1;
                if (i < f1) n = is[i];
                const bool ok = (n >= 0 && n < U);
                const unsigned long long mask = __ballot(ok);
                const int cnt = __popcll(mask);
                nins += cnt;
                bool tiny = false;
                if (ok) {                            // lane-parallel: one fragment per lane, compacted into LDS
                    const double b = bj[n];
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    const double a = (nucp[n] * b) / sn, c = (nfrp[n] * b) / sf;
                    ac[2 * pos] = a - c;                 // mixture alpha a + (1 - alpha) c evaluated as fma(alpha, a - c, c)
                    ac[2 * pos + 1] = c;
                    tiny = !(a >= 0x1p-200 && c >= 0x1p-200);
                }
                const bool safe = __ballot(tiny) == 0ull;
                __builtin_amdgcn_wave_barrier();
                int q = 0;
                if (safe) {
                    for (; q + 16 <= cnt; q += 16) {           // 16 fragments: 4 multiplies on each of the 8 chains, one renorm
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
    #pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const double au = ac[2 * (q + 4 * u + v)], cu = ac[2 * (q + 4 * u + v) + 1];
                                m0[v] *= fma(al0, au, cu);
                                m1[v] *= fma(al1, au, cu);
                            }
                        }
    #pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            int ex0, ex1;
                            m0[v] = frexp(m0[v], &ex0);
                            m1[v] = frexp(m1[v], &ex1);
                            e0 += ex0;
                            e1 += ex1;
                        }
                    }
                    for (; q + 4 <= cnt; q += 4) {             // 4 fragments: one multiply per chain
    #pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const double au = ac[2 * (q + v)], cu = ac[2 * (q + v) + 1];
                            int ex0, ex1;
                            m0[v] = frexp(m0[v] * fma(al0, au, cu), &ex0);
                            m1[v] = frexp(m1[v] * fma(al1, au, cu), &ex1);
                            e0 += ex0;
                            e1 += ex1;
                        }
                    }
                }
                for (; q < cnt; ++q) {                         // tail / unsafe windows: renormalise every multiply
                    const double au = ac[2 * q], cu = ac[2 * q + 1];
                    int ex0, ex1;
                    m0[0] = frexp(m0[0] * fma(al0, au, cu), &ex0);   // static index: keeps the chains in registers
                    m1[0] = frexp(m1[0] * fma(al1, au, cu), &ex1);
                    e0 += ex0;
                    e1 += ex1;
                }
                __builtin_amdgcn_wave_barrier();
            }
            double mm0, mm1;
            {
                int ex0, ex1, ex2, ex3, ex4, ex5;
                const double p0 = frexp(m0[0] * m0[1], &ex0), p1 = frexp(m0[2] * m0[3], &ex1);
                const double r0 = frexp(m1[0] * m1[1], &ex2), r1 = frexp(m1[2] * m1[3], &ex3);
                mm0 = frexp(p0 * p1, &ex4);
                mm1 = frexp(r0 * r1, &ex5);
                e0 += ex0 + ex1 + ex4;
                e1 += ex2 + ex3 + ex5;
            }
            const long long go = ct.grid_off[chunk] + k;
            if (nins == 0) {                         // sum(new_inserts) > 0 fails: stay NaN (Occupancy.py:143)
                if (lane == 0) { g_occ[go] = g_lo[go] = g_hi[go] = __builtin_nan(""); }
                continue;
            }
            if (ABL == 5) { if (lane == 0) g_occ[go] = mm0 + mm1 + e0 + e1; continue; }
            // Decision without logarithms.  The likelihood of alpha is L = mm * 2^e with mm in [0.5, 1) (0 stands for
            // log L = -inf).  argmax ll == argmax (e, mm) lexicographically, and the reference's likelihood-ratio test
            //   2 (max ll - ll) < cutoff   <=>   L > Lmax * exp(-cutoff / 2)
            // is evaluated as  ldexp(mm, e - emax) > mmax * ci_factor  (ci_factor = exp(-cutoff/2) from the host).
            // reference: a zero-probability insert size gives log(0)*ins = -inf (ins>0) or NaN (ins==0) -> -inf
            if (flags & (8 | 4)) { mm0 = 0.0; mm1 = 0.0; }
            if ((flags & 2) && al0 == 0.0) mm0 = 0.0;
            if ((flags & 2) && al1 == 0.0) mm1 = 0.0;
            if ((flags & 1) && be0 == 0.0) mm0 = 0.0;
            if ((flags & 1) && be1 == 0.0) mm1 = 0.0;
            if (!(mm0 > 0.0) || a0 >= om.n_alpha) mm0 = 0.0;      // NaN likelihood -> -inf as well
            if (!(mm1 > 0.0) || a1 >= om.n_alpha) mm1 = 0.0;
            const double NONE = -1e300;
            const double ed0 = mm0 > 0.0 ? (double)e0 : NONE, ed1 = mm1 > 0.0 ? (double)e1 : NONE;
            const double emax = wave_max(fmax(ed0, ed1));
            if (emax == NONE) {
                // every likelihood is -inf: the reference raises ValueError (min of empty, Occupancy.py:118)
                if (lane == 0) { g_occ[go] = g_lo[go] = g_hi[go] = __builtin_nan(""); atomicOr(&status[chunk], 1); }
                continue;
            }
            const double mmax = wave_max(fmax(ed0 == emax ? mm0 : 0.0, ed1 == emax ? mm1 : 0.0));
            const unsigned long long eq0 = __ballot(ed0 == emax && mm0 == mmax);
            const unsigned long long eq1 = __ballot(ed1 == emax && mm1 == mmax);
            const int imax = eq0 ? (__ffsll((long long)eq0) - 1) : (WAVE + __ffsll((long long)eq1) - 1);
            const double thr = mmax * om.ci_factor;
            const int iemax = (int)emax;
            const int d0 = e0 - iemax, d1 = e1 - iemax;
            const bool in0 = mm0 > 0.0 && d0 > -1100 && ldexp(mm0, d0) > thr;
            const bool in1 = mm1 > 0.0 && d1 > -1100 && ldexp(mm1, d1) > thr;
            const unsigned long long c0 = __ballot(in0);
            const unsigned long long c1 = __ballot(in1);
            if (lane == 0) {
                const int ilo = c0 ? (__ffsll((long long)c0) - 1) : (WAVE + __ffsll((long long)c1) - 1);
                const int ihi = c1 ? (WAVE + 63 - __clzll((long long)c1)) : (63 - __clzll((long long)c0));
                g_occ[go] = om.alphas[imax];
                g_lo[go] = om.alphas[ilo];
                g_hi[go] = om.alphas[ihi];
            }
        }

    }
    if (!tile_list) break;
  }
}

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// The un-smoothed tracks are piecewise constant on `step`-bp blocks (block k = bases [k*step, (k+1)*step)), so the
// 2h+1-tap convolution collapses to <= 2*ceil(h/step)+2 block terms per base: y[t] = sum_b W[j][b] v[b] / sum_b W[j][b] ok[b]
// with W[j][b] = the window weights that fall on block b for a base at offset j inside its own block (the same for every
// block, computed once per workgroup).  The last block of a chunk can be cut by the chunk end; its weight is summed directly.
// ------------------------------------------------------------------------------------------------
// K4  occupancy smoothing: expand the per-grid values to bases ([i-halfstep, min(i+halfstep+1, L)),
//     Occupancy.py:144-146; bases past the last grid block stay NaN) and apply the NaN-aware Gaussian
//     (makeSmoothed, Occupancy.py:147-153) to vals / lower / upper; also occ cov = nuc_cov+nfr_cov is
//     produced by natac_occ_cov below.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) natac_occ_smooth(ChunkTable ct, const int2 *__restrict__ tiles, OccModelDev om,
                                                          const double *__restrict__ win, int M,
                                                          const double *__restrict__ g_occ, const double *__restrict__ g_lo,
                                                          const double *__restrict__ g_hi, double *__restrict__ s_occ,
                                                          double *__restrict__ s_lo, double *__restrict__ s_hi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int h = (M - 1) / 2, step = om.step;
    const int NB = 2 * ((h + step - 1) / step) + 2;
    const int NG = (255 + 2 * h) / step + 3;
    double *wl = smem;                  // [M]
    double *wb = wl + ((M + 1) & ~1);   // [step][NB]
    double *gv = wb + step * NB;        // [NG] x 3
    double *gl = gv + NG, *gh = gl + NG;
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int nk = (L - om.halfstep + step - 1) / step;
    const long long gb = ct.grid_off[chunk];
    const int kfirst = floor_div(x0 - h, step);
    const double qn = __builtin_nan("");
    for (int u = threadIdx.x; u < M; u += 256) wl[u] = win[u];
    for (int u = threadIdx.x; u < NG; u += 256) {
        const int kb = kfirst + u;
        double v = qn, lo = qn, hi = qn;
        if (kb >= 0 && kb < nk) { v = g_occ[gb + kb]; lo = g_lo[gb + kb]; hi = g_hi[gb + kb]; }
        gv[u] = v; gl[u] = lo; gh[u] = hi;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < step * NB; idx += 256) {
        const int j = idx / NB, bi = idx - j * NB;
        const int bmin = floor_div(j - h, step);
        double sacc = 0.0;
        for (int d = 0; d < step; ++d) {
            const int n = j + h - (bmin + bi) * step - d;      // tap that lands on position d of block bmin+bi
            if (n >= 0 && n < M) sacc += wl[n];
        }
        wb[idx] = sacc;
    }
    __syncthreads();
    const int g = x0 + threadIdx.x;
    if (g >= L) return;
    const int k0 = g / step, j = g - k0 * step;
    const int bmin = floor_div(j - h, step);
    double nv = 0.0, nl = 0.0, nh = 0.0, den = 0.0;
    for (int bi = 0; bi < NB; ++bi) {
        const int kb = k0 + bmin + bi;
        const int u = kb - kfirst;
        if (u < 0 || u >= NG) continue;
        const double v = gv[u];
        if (v != v) continue;                               // NaN block (no inserts) or outside the chunk
        double w = wb[j * NB + bi];
        if ((kb + 1) * step > L) {                          // block cut by the chunk end: only existing bases count
            w = 0.0;
            for (int p = kb * step; p < L; ++p) {
                const int n = g + h - p;
                if (n >= 0 && n < M) w += wl[n];
            }
        }
        nv = fma(w, v, nv);
        nl = fma(w, gl[u], nl);
        nh = fma(w, gh[u], nh);
        den += w;
    }
    const long long o = ct.out_off[chunk] + g;
    s_occ[o] = den == 0.0 ? qn : nv / den;
    if (s_lo) s_lo[o] = den == 0.0 ? qn : nl / den;
    if (s_hi) s_hi[o] = den == 0.0 ? qn : nh / den;
}

// doubles as unsigned keys with the same order (atomicMin over the finite values of a chunk)
__device__ __forceinline__ unsigned long long double_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_double(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// The same smoothing, one lane per step-block (STEP bases): the STEP bases of a block see the same NB blocks -- ceil(h / STEP) to the
// left of their own, (h + STEP - 1) / STEP to the right (2 h / STEP + 2 with a zero-weight spare for h a multiple of STEP; a block
// that only some of the bases reach has the weight 0 for the others, which adds +0 like the skipped blocks of natac_occ_smooth)
// --, so a lane reads each block's three values once for all of them (16
// LDS reads per base instead of 104) and the block weights wb[bi][j] -- the table natac_occ_smooth builds per workgroup,
// here precomputed by the host in the same summation order -- are wave-uniform scalar loads.  Term order, NaN skipping and
// the cut last block are those of natac_occ_smooth: identical bits.  Also leaves, per chunk, the minimum finite smoothed
// occupancy (as double_key, atomicMin) and whether it holds a NaN, for natac_fill_nan_chunks.
// tile = (chunk, x0), x0 a multiple of 256 STEP.  Dynamic LDS: 3 x (256 + NB) + 256 STEP doubles.
template <int STEP>
__global__ void __launch_bounds__(256) natac_occ_smooth_blk(ChunkTable ct, const int2 *__restrict__ tiles, OccModelDev om,
                                                              const double *__restrict__ win, int M,
                                                              const double *__restrict__ wb /* [NB][STEP] */, int NB,
                                                              const double *__restrict__ g_occ, const double *__restrict__ g_lo,
                                                              const double *__restrict__ g_hi, double *__restrict__ s_occ,
                                                              double *__restrict__ s_lo, double *__restrict__ s_hi,
                                                              unsigned long long *__restrict__ cmin_key, int *__restrict__ cnan) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int h = (M - 1) / 2, nbh = (h + STEP - 1) / STEP;      // blocks to the left of a base's own block that its window can reach
    const int NG = 256 + NB;
    double *gv = smem, *gl = gv + NG, *gh = gl + NG;
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int nk = (L - om.halfstep + STEP - 1) / STEP;
    const long long gb = ct.grid_off[chunk];
    const int kb0 = x0 / STEP, kfirst = kb0 - nbh;
    const double qn = __builtin_nan("");
    for (int u = threadIdx.x; u < NG; u += 256) {
        const int kb = kfirst + u;
        double v = qn, lo = qn, hi = qn;
        if (kb >= 0 && kb < nk) { v = g_occ[gb + kb]; lo = g_lo[gb + kb]; hi = g_hi[gb + kb]; }
        gv[u] = v; gl[u] = lo; gh[u] = hi;
    }
    __syncthreads();
    const int kme = kb0 + threadIdx.x, base0 = kme * STEP;
    double nv[STEP], nl[STEP], nh[STEP], den[STEP];
#pragma unroll
    for (int j = 0; j < STEP; ++j) { nv[j] = 0.0; nl[j] = 0.0; nh[j] = 0.0; den[j] = 0.0; }
    // only the last block of a chunk can be cut by the chunk end, and only when L is not a whole number of blocks past the
    // half step: the per-lane test is compiled in for those chunks only (block-uniform).
    // CLEAN (wave-uniform): no NaN block -- no fragment-free stretch, no chunk border -- and no cut block within reach of any lane of
    // the wave, the ordinary case: every select falls away and the denominator is the table's own sum wb[NB][j], formed by the host
    // with the same fma sequence (the same bits as adding the weights up here).
    auto sweep = [&](auto may_cut, auto clean) {
        constexpr bool CLEAN = decltype(clean)::value;
        for (int bi = 0; bi < NB; ++bi) {
            const int u = threadIdx.x + bi, kb = kme - nbh + bi;
            const double v = gv[u], lo = gl[u], hi = gh[u];
            double w[STEP];
#pragma unroll
            for (int j = 0; j < STEP; ++j) w[j] = wb[bi * STEP + j];    // wave-uniform: one group of scalar loads
            if (CLEAN) {
#pragma unroll
                for (int j = 0; j < STEP; ++j) { nv[j] = fma(w[j], v, nv[j]); nl[j] = fma(w[j], lo, nl[j]); nh[j] = fma(w[j], hi, nh[j]); }
                continue;
            }
            const bool skip = v != v;                           // NaN block (no inserts) or outside the chunk: adds w * 0 = +0 to
            const double vz = skip ? 0.0 : v, lz = skip ? 0.0 : lo, hz = skip ? 0.0 : hi;   // every sum (the table weights are finite)
            const double okf = skip ? 0.0 : 1.0;                // den + w as fma(w, 1, den): the same bits
            if (decltype(may_cut)::value && !skip && (kb + 1) * STEP > L) {   // cut block: only existing bases count
#pragma unroll
                for (int j = 0; j < STEP; ++j) {
                    double wc = 0.0;
                    for (int p = kb * STEP; p < L; ++p) {
                        const int n = base0 + j + h - p;
                        if (n >= 0 && n < M) wc += win[n];
                    }
                    w[j] = wc;
                }
            }
#pragma unroll
            for (int j = 0; j < STEP; ++j) {
                nv[j] = fma(w[j], vz, nv[j]); nl[j] = fma(w[j], lz, nl[j]); nh[j] = fma(w[j], hz, nh[j]);
                den[j] = fma(w[j], okf, den[j]);
            }
        }
        if (CLEAN) {
#pragma unroll
            for (int j = 0; j < STEP; ++j) den[j] = wb[NB * STEP + j];
        }
    };
    bool wave_clean;
    {   // the wave's lanes reach blocks u = w0 .. w0 + 63 + NB - 1 of the staged strip (w0 = first lane's threadIdx)
        const int w0 = threadIdx.x & ~63, ln = threadIdx.x & 63;
        bool bad = false;
        for (int u = ln; u < 64 + NB; u += 64) bad = bad || (gv[w0 + u] != gv[w0 + u]);      // NB may exceed 64 (step 1)
        // ... and the last block any lane reaches lies whole inside the chunk (a cut last block is not NaN, but its weights differ)
        wave_clean = __ballot(bad) == 0ull && (long long)(kb0 + w0 + 63 + (NB - 1 - nbh) + 1) * STEP <= L;
    }
    if (base0 < L) {
        if (wave_clean) sweep(std::false_type{}, std::true_type{});      // a cut block's successor is outside the chunk: NaN, never clean
        else if (nk * STEP > L) sweep(std::true_type{}, std::false_type{});
        else sweep(std::false_type{}, std::false_type{});
    }
    // results -> the wave's LDS strip (lane-major: [lane][j], conflict-free for odd STEP) -> coalesced stores of 64 STEP bases
    double mn = __builtin_inf();
    int anynan = 0;
    const long long ob = ct.out_off[chunk];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *strip = gh + NG + wv * (64 * STEP);
    const int wbase = (kb0 + wv * 64) * STEP;                   // first base of the wave
#pragma unroll
    for (int trk = 0; trk < 3; ++trk) {
#pragma unroll
        for (int j = 0; j < STEP; ++j) {
            const double num = trk == 0 ? nv[j] : (trk == 1 ? nl[j] : nh[j]);
            strip[lane * STEP + j] = den[j] == 0.0 ? qn : num / den[j];
        }
        __builtin_amdgcn_wave_barrier();
        double *dst = trk == 0 ? s_occ : (trk == 1 ? s_lo : s_hi);
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
            const int g = wbase + i * 64 + lane;
            if (g < L) {
                const double o = strip[i * 64 + lane];
                dst[ob + g] = o;
                if (trk == 0) { if (o != o) anynan = 1; else mn = fmin(mn, o); }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    mn = wave_min(mn);
    anynan = wave_or(anynan);
    if ((threadIdx.x & 63) == 0) {
        if (mn != __builtin_inf()) atomicMin(cmin_key + chunk, double_key(mn));
        if (anynan) atomicOr(cnan + chunk, 1);
    }
}

// call_peaks' in-place NaN fill for the track natac_occ_smooth_blk wrote (see natac_fill_nan_min): chunks without a NaN -- or
// without a finite value -- are left alone; one workgroup per chunk.
__global__ void __launch_bounds__(256) natac_fill_nan_chunks(ChunkTable ct, const unsigned long long *__restrict__ cmin_key,
                                                               const int *__restrict__ cnan, double *__restrict__ track) {
    const int chunk = blockIdx.x;
    if (!cnan[chunk]) return;
    const unsigned long long key = cmin_key[chunk];
    if (key == ~0ull) return;
    const double mn = key_double(key);
    const int L = ct.chunk_len[chunk];
    double *p = track + ct.out_off[chunk];
    for (int g = threadIdx.x; g < L; g += 256)
        if (p[g] != p[g]) p[g] = mn;
}

// occ coverage (all insert sizes < upper, window 2*flank+1): Occupancy.py:221-224.  When the occupancy window
// equals the V-plot window and upper == vupper this is nuc_cov + nfr_cov; computed independently so that the occ
// stage does not depend on the nuc stage.
__global__ void __launch_bounds__(256) natac_occ_cov(ChunkTable ct, const int2 *__restrict__ tiles, int upper, int flank,
                                                       double *__restrict__ cov) {
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, g = t.y + threadIdx.x;
    const int L = ct.chunk_len[chunk];
    if (g >= L) return;
    const int nfr = (int)(ct.frag_off[chunk + 1] - ct.frag_off[chunk]);
    const int *cen = ct.centre + ct.frag_off[chunk];
    const int *iln = ct.ilen + ct.frag_off[chunk];
    int f = lower_bound_i32(cen, 0, nfr, g - flank);
    int cnt = 0;
    for (; f < nfr; ++f) {
        if (cen[f] > g + flank) break;
        const int n = iln[f];
        cnt += (n >= 0 && n < upper);
    }
    cov[ct.out_off[chunk] + g] = (double)cnt;
}

// call_peaks' in-place NaN fill (pyatac/utils.py:86-91) applied to smoothed_vals before it is written
// (Occupancy.py:227 -> run_occ.py:47): NaNs become the chunk's minimum finite value; all-NaN chunks stay.
// One workgroup per chunk.
__global__ void __launch_bounds__(256) natac_fill_nan_min(ChunkTable ct, const double *__restrict__ src,
                                                            double *__restrict__ dst) {
    __shared__ double red[4];
    __shared__ int anynan[4];
    const int chunk = blockIdx.x;
    const int L = ct.chunk_len[chunk];
    const long long ob = ct.out_off[chunk];
    double mn = __builtin_inf();
    int nn = 0;
    for (int g = threadIdx.x; g < L; g += 256) {
        const double v = src[ob + g];
        if (v != v) nn = 1; else mn = fmin(mn, v);
    }
    mn = wave_min(mn);
    nn = wave_or(nn);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mn; anynan[threadIdx.x >> 6] = nn; }
    __syncthreads();
    mn = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    nn = anynan[0] | anynan[1] | anynan[2] | anynan[3];
    const bool fill = nn && (mn != __builtin_inf());
    for (int g = threadIdx.x; g < L; g += 256) {
        const double v = src[ob + g];
        dst[ob + g] = (fill && v != v) ? mn : v;
    }
}

// ------------------------------------------------------------------------------------------------
// K6  per-base insertion counts (pyatac/fragments.pyx:43-67): +1 at l and at r = l+n-1 for lower<=n<upper.
// One workgroup walks the fragments of one chunk; integer atomics (order-independent, bit-exact).
// ------------------------------------------------------------------------------------------------
// chunks that fit LDS (the host checks the longest one): counts in LDS, every base written once -- no memset, no global atomics
__global__ void __launch_bounds__(256) natac_insertions_lds(ChunkTable ct, int lower, int upper, int *__restrict__ ins) {
    extern __shared__ int cnt_s[];
    const int chunk = blockIdx.x;
    const int L = ct.chunk_len[chunk];
    const long long fa = ct.frag_off[chunk], fb = ct.frag_off[chunk + 1];
    for (int g = threadIdx.x; g < L; g += 256) cnt_s[g] = 0;
    __syncthreads();
    for (long long f = fa + threadIdx.x; f < fb; f += 256) {
        const int n = ct.ilen[f];
        if (n < lower || n >= upper) continue;
        const int l = ct.lpos[f], r = l + n - 1;
        if (l >= 0 && l < L) atomicAdd(&cnt_s[l], 1);
        if (r >= 0 && r < L) atomicAdd(&cnt_s[r], 1);
    }
    __syncthreads();
    int *out = ins + ct.out_off[chunk];
    for (int g = threadIdx.x; g < L; g += 256) out[g] = cnt_s[g];
}

__global__ void __launch_bounds__(256) natac_insertions(ChunkTable ct, int lower, int upper, int *__restrict__ ins) {
    const int chunk = blockIdx.x;
    const int L = ct.chunk_len[chunk];
    const long long fa = ct.frag_off[chunk], fb = ct.frag_off[chunk + 1];
    int *out = ins + ct.out_off[chunk];
    for (long long f = fa + threadIdx.x; f < fb; f += 256) {
        const int n = ct.ilen[f];
        if (n < lower || n >= upper) continue;
        const int l = ct.lpos[f], r = l + n - 1;
        if (l >= 0 && l < L) atomicAdd(&out[l], 1);
        if (r >= 0 && r < L) atomicAdd(&out[r], 1);
    }
}

// ------------------------------------------------------------------------------------------------
// K7  candidate statistics: one workgroup per candidate position p of a chunk.
//   window cells (r, c): insert size i = vlower + r, centre x = p - w + c
//   B0 = E[x-(i-1)//2] E[x+i//2],  B = sizes[i] B0
//   S_B = sum B, S_BV = sum B V, S_BV2 = sum B V^2, S_B0V = sum V B0
//   var = int(nuc_cov[p]) * (S_BV2/S_B - (S_BV/S_B)^2)                 multinomial_cov.pyx:20-31 (closed form)
//   lr  = sum_frag log(V B0 / S_B0V) - sum_frag log(B / S_B)           NucleosomeCalling.py:110-122
//   z   = norm[p] / sqrt(var)                                          NucleosomeCalling.py:123-127
// A zero cell in either model makes log(0)*0 = NaN in the reference -> lr = NaN here as well.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) natac_candidates(ChunkTable ct, VMatDev vm, const int *__restrict__ cand_chunk,
                                                          const int *__restrict__ cand_pos,
                                                          const double *__restrict__ nuc_cov, const double *__restrict__ norm,
                                                          double *__restrict__ out_lr, double *__restrict__ out_var,
                                                          double *__restrict__ out_z) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[5][4];
    __shared__ int redz[4];
    const int k = blockIdx.x;
    const int chunk = cand_chunk[k], p = cand_pos[k];
    const int L = ct.chunk_len[chunk];
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int EW = vm.W + A + Bh;
    double *Et = smem;  // Et[u] <-> coordinate p - w - A + u
    {
        const double *b = ct.bias ? ct.bias + ct.bias_off[chunk] : nullptr;
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = p - vm.w - A + ct.bias_left;
        for (int u = threadIdx.x; u < EW; u += 256) {
            const int j = j0 + u;
            double e = 1.0;
            if (b) e = (j >= 0 && j < nb) ? exp(b[j]) : 0.0;
            Et[u] = e;
        }
    }
    __syncthreads();
    double sB = 0, sBV = 0, sBV2 = 0, sB0V = 0;
    int zero = 0;
    // 2-D sweep without integer division: 128 lanes across the window columns, two row phases
    {
        const int c = threadIdx.x & 127;
        if (c < vm.W) {
            for (int r = threadIdx.x >> 7; r < vm.R; r += 2) {
                const int i = vm.lower + r;
                const int hl = floor_half(i - 1), hr = floor_half(i);
                const double b0 = (hl == -hr) ? Et[c + A] : Et[c + A - hl] * Et[c + A + hr];
                const double bb = vm.srow[r] * b0;
                const double v = vm.mat[r * vm.W + c];
                const double vb0 = v * b0;
                sB += bb;
                sBV = fma(bb, v, sBV);
                sBV2 = fma(bb * v, v, sBV2);
                sB0V += vb0;
                if (vb0 == 0.0 || bb == 0.0) zero = 1;
            }
        }
    }
    sB = wave_sum(sB); sBV = wave_sum(sBV); sBV2 = wave_sum(sBV2); sB0V = wave_sum(sB0V);
    zero = wave_or(zero);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = sB; red[1][wv] = sBV; red[2][wv] = sBV2; red[3][wv] = sB0V; redz[wv] = zero; }
    __syncthreads();
    sB = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sBV = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    sBV2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    sB0V = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
    zero = redz[0] | redz[1] | redz[2] | redz[3];
    // likelihoods over the window's fragments
    const int nfr = (int)(ct.frag_off[chunk + 1] - ct.frag_off[chunk]);
    const int *cen = ct.centre + ct.frag_off[chunk];
    const int *iln = ct.ilen + ct.frag_off[chunk];
    const int f0 = lower_bound_i32(cen, 0, nfr, p - vm.w);
    const int f1 = lower_bound_i32(cen, f0, nfr, p + vm.w + 1);
    double nl = 0.0, ul = 0.0;
    for (int f = f0 + threadIdx.x; f < f1; f += 256) {
        const int n = iln[f];
        if (n < vm.lower || n >= vm.upper) continue;
        const int r = n - vm.lower, c = cen[f] - p + vm.w;
        const int hl = floor_half(n - 1), hr = floor_half(n);
        const double b0 = (hl == -hr) ? Et[c + A] : Et[c + A - hl] * Et[c + A + hr];
        nl += log((vm.mat[r * vm.W + c] * b0) / sB0V);
        ul += log((vm.srow[r] * b0) / sB);
    }
    nl = wave_sum(nl); ul = wave_sum(ul);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[0][wv] = nl; red[1][wv] = ul; }
    __syncthreads();
    if (threadIdx.x == 0) {
        nl = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        ul = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const long long o = ct.out_off[chunk] + p;
        const double m1 = sBV / sB;
        const int reads = (int)nuc_cov[o];
        const double var = (double)reads * (sBV2 / sB - m1 * m1);
        out_lr[k] = zero ? __builtin_nan("") : (nl - ul);
        out_var[k] = var;
        out_z[k] = norm[o] / sqrt(var);
    }
}

// K7b  the same statistics, four candidates per wave: every template value V[r,c] is loaded once per lane and applied to
// four candidate windows whose exp(bias) slices sit in LDS (4x less L2 template traffic, 4 independent dependency chains,
// no block-level synchronisation: DPP wave reductions only).  Workgroup = 4 waves = 16 candidates.
constexpr int CAND_PER_WAVE = 4;

// first index in the sorted array a[lo, hi) with a[idx] >= key, found by the whole wave: 64 evenly spaced probes per step
// (one coalesced-ish load + a ballot) instead of one dependent load per bisection step.  Every lane returns the result.
__device__ __forceinline__ int wave_lower_bound(const int *__restrict__ a, int lo, int hi, int key, int lane) {
    while (hi - lo > WAVE) {
        const int n = hi - lo, stride = (n + WAVE - 1) / WAVE;
        const int pi = min((lane + 1) * stride - 1, n - 1);
        const int cnt = __popcll(__ballot(a[lo + pi] < key));      // probes are sorted: the predicate is a prefix
        lo = min(lo + cnt * stride, hi);
        hi = min(lo + stride, hi);
    }
    const int i = lo + lane;
    const int v = (i < hi) ? a[i] : 0x7fffffff;
    return lo + __popcll(__ballot(v < key));
}

// LDS: [zeros: A + Bh + 36][ones: W] shared by the workgroup, then [4 waves][CAND_PER_WAVE][EWP] bias windows.
// Lanes without a template column (c >= W) read the zeros block instead of branching: their products are exactly 0.
// USEBG: sum B and sum B V of a candidate's window are the per-base values the background kernel already formed at that
// position (bcov / bnum of natac_run_nuc); the sweep then keeps two accumulators per candidate instead of four.
// USEBG keeps two accumulators per candidate and column and fits three waves per SIMD (168 VGPRs); the four-accumulator form
// needs ~230 VGPRs: it is compiled for two waves per SIMD instead of spilling 196 registers to scratch (round 2's build)
template <bool USEBG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(USEBG ? 3 : 2, USEBG ? 3 : 2))) natac_candidates4(ChunkTable ct, VMatDev vm, const int *__restrict__ cand_chunk,
                                                           const int *__restrict__ cand_pos, int ncand,
                                                           const double *__restrict__ nuc_cov, const double *__restrict__ norm,
                                                           const double *__restrict__ bnum, const double *__restrict__ bcov,
                                                           double *__restrict__ out_lr, double *__restrict__ out_var,
                                                           double *__restrict__ out_z) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int EW = vm.W + A + Bh, EWP = (EW + 1) & ~1;
    const int ZN = (A + Bh + 5 + 32) & ~1, ON = (vm.W + 1) & ~1; // idle lanes index the zeros block from 2..33 (bases run 2 below)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int u = threadIdx.x; u < ZN + ON; u += 256) smem[u] = (u < ZN) ? 0.0 : 1.0;
    const int et0 = ZN + ON + wave * CAND_PER_WAVE * EWP;          // smem[et0 + q * EWP + u] <-> coordinate p_q - w - A + u
    const int k0 = (blockIdx.x * 4 + wave) * CAND_PER_WAVE;
    const bool active = k0 < ncand;                                // wave-uniform
    int chunk[CAND_PER_WAVE], pos[CAND_PER_WAVE];
    bool esmall[CAND_PER_WAVE];
#pragma unroll
    for (int q = 0; q < CAND_PER_WAVE; ++q) {
        esmall[q] = false; chunk[q] = 0; pos[q] = 0;
        if (!active) continue;
        const int k = (k0 + q < ncand) ? k0 + q : k0;          // tail: recompute candidate k0 (result discarded)
        chunk[q] = cand_chunk[k];
        pos[q] = cand_pos[k];
        const int L = ct.chunk_len[chunk[q]];
        const double *b = ct.bias ? ct.ebias + ct.bias_off[chunk[q]] : nullptr;     // exp(bias), natac_exp_bias
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = pos[q] - vm.w - A + ct.bias_left;
        double emin = __builtin_inf();
        for (int u = lane; u < EW; u += WAVE) {
            const int j = j0 + u;
            double e = 1.0;
#ifndef NATAC_ABL_CAND_STAGE
            if (b) e = (j >= 0 && j < nb) ? b[j] : 0.0;
#endif
            smem[et0 + q * EWP + u] = e;
            emin = fmin(emin, e);
        }
        esmall[q] = !(wave_min(emin) > 0x1p-500);     // no product of two window values can underflow to 0 above this
    }
    __syncthreads();
    if (!active) return;
    double sB[CAND_PER_WAVE], sBV[CAND_PER_WAVE], sBV2[CAND_PER_WAVE], sB0V[CAND_PER_WAVE];
    bool zero[CAND_PER_WAVE];
#pragma unroll
    for (int q = 0; q < CAND_PER_WAVE; ++q) { sB[q] = sBV[q] = sBV2[q] = sB0V[q] = 0.0; zero[q] = false; }
    // a model cell is exactly 0 only if the template or the size distribution holds a 0 (host-known flags in vm) or a
    // product of two exp(bias) values underflows; the per-cell test runs only in those (wave-uniform) cases
    const bool check_any = vm.has_zero || esmall[0] || esmall[1] || esmall[2] || esmall[3];
    const int c1 = lane, c2 = lane + WAVE;
    const bool h1 = c1 < vm.W, h2 = c2 < vm.W;
    const int cc1 = h1 ? c1 : 0, cc2 = h2 ? c2 : 0;
    int o1[CAND_PER_WAVE], o2[CAND_PER_WAVE];                   // per-lane window bases (the zeros block for idle lanes)
#pragma unroll
    for (int q = 0; q < CAND_PER_WAVE; ++q) {
        // an idle lane reads the zeros block at the bank its own column would use: no conflict with the busy lanes
        o1[q] = h1 ? et0 + q * EWP + c1 : 2 + ((et0 + q * EWP + c1 - 2) & 31);
        o2[q] = h2 ? et0 + q * EWP + c2 : 2 + ((et0 + q * EWP + c2 - 2) & 31);
    }
    const int R = vm.R, W = vm.W;
    // one template row, any geometry (used for the R % 4 tail rows and for V-plots that include insert size 1)
    auto row_generic = [&](int r, auto CHECK) {
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const bool single = (hl == -hr);     // i == 1 (the two ends coincide, chunkmat2d.py:150-151): right factor from the ones block
        const double sr = vm.srow[r];
        const double v1 = vm.mat[r * W + cc1], v2 = vm.mat[r * W + cc2];
        const double v1sq = v1 * v1, v2sq = v2 * v2;
        const int ol = A - hl, orr = A + hr;
#pragma unroll
        for (int q = 0; q < CAND_PER_WAVE; ++q) {
            const int r1 = single ? ZN + cc1 : o1[q] + orr, r2 = single ? ZN + cc2 : o2[q] + orr;
            {
                const double b0 = smem[o1[q] + ol] * smem[r1];
                const double bb = sr * b0;
                if (!USEBG) { sB[q] += bb; sBV[q] = fma(bb, v1, sBV[q]); }
                        sB0V[q] = fma(v1, b0, sB0V[q]); sBV2[q] = fma(bb, v1sq, sBV2[q]);
                if (decltype(CHECK)::value) zero[q] |= (h1 && (v1 * b0 == 0.0 || bb == 0.0));
            }
            {
                const double b0 = smem[o2[q] + ol] * smem[r2];
                const double bb = sr * b0;
                if (!USEBG) { sB[q] += bb; sBV[q] = fma(bb, v2, sBV[q]); }
                        sB0V[q] = fma(v2, b0, sB0V[q]); sBV2[q] = fma(bb, v2sq, sBV2[q]);
                if (decltype(CHECK)::value) zero[q] |= (h2 && (v2 * b0 == 0.0 || bb == 0.0));
            }
        }
    };
    // four rows at a time (lower >= 2).  rb % 4 == 0, so the parity of i0 = lower + rb is that of `lower` (PAR) and the
    // half-lengths inside a block are compile-time offsets from the block's first row: every LDS address is a running
    // base register plus an immediate.  Template values arrive through two alternating register sets (no copies).
    auto sweep4 = [&](auto CHECK, auto PARITY) {
        constexpr int PAR = decltype(PARITY)::value;
        constexpr int DL[4] = {0, PAR ? 0 : 1, 1, PAR ? 1 : 2};     // hl(i0 + u) - hl(i0)
        constexpr int DR[4] = {0, PAR ? 1 : 0, 1, PAR ? 2 : 1};     // hr(i0 + u) - hr(i0)
        int lb1[CAND_PER_WAVE], lb2[CAND_PER_WAVE], rb1[CAND_PER_WAVE], rb2[CAND_PER_WAVE];
        {
            const int hl0 = floor_half(vm.lower - 1), hr0 = floor_half(vm.lower);
#pragma unroll
            for (int q = 0; q < CAND_PER_WAVE; ++q) {
                lb1[q] = o1[q] + A - hl0 - 2; lb2[q] = o2[q] + A - hl0 - 2;
                rb1[q] = o1[q] + A + hr0;     rb2[q] = o2[q] + A + hr0;
            }
        }
        auto loadv = [&](int rb, double (&x1)[4], double (&x2)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int r = min(rb + u, R - 1); x1[u] = vm.mat[r * W + cc1]; x2[u] = vm.mat[r * W + cc2]; }
        };
        auto block = [&](int rb, const double (&x1)[4], const double (&x2)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double sr = vm.srow[rb + u];
                const double v1 = x1[u], v2 = x2[u];
                const double v1sq = v1 * v1, v2sq = v2 * v2;
#pragma unroll
                for (int q = 0; q < CAND_PER_WAVE; ++q) {
                    {
                        const double b0 = smem[lb1[q] + (2 - DL[u])] * smem[rb1[q] + DR[u]];
                        const double bb = sr * b0;
                        if (!USEBG) { sB[q] += bb; sBV[q] = fma(bb, v1, sBV[q]); }
                        sB0V[q] = fma(v1, b0, sB0V[q]); sBV2[q] = fma(bb, v1sq, sBV2[q]);
                        if (decltype(CHECK)::value) zero[q] |= (h1 && (v1 * b0 == 0.0 || bb == 0.0));
                    }
                    {
                        const double b0 = smem[lb2[q] + (2 - DL[u])] * smem[rb2[q] + DR[u]];
                        const double bb = sr * b0;
                        if (!USEBG) { sB[q] += bb; sBV[q] = fma(bb, v2, sBV[q]); }
                        sB0V[q] = fma(v2, b0, sB0V[q]); sBV2[q] = fma(bb, v2sq, sBV2[q]);
                        if (decltype(CHECK)::value) zero[q] |= (h2 && (v2 * b0 == 0.0 || bb == 0.0));
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < CAND_PER_WAVE; ++q) { lb1[q] -= 2; lb2[q] -= 2; rb1[q] += 2; rb2[q] += 2; }
        };
        double va1[4], va2[4], vb1[4], vb2[4];
        const int nblk = R / 4;
        int bk = 0;
        if (USEBG) {
            // three waves per SIMD hide the template loads: no second register set
#pragma unroll 1
            for (; bk < nblk; ++bk) {
                loadv(4 * bk, va1, va2);
                block(4 * bk, va1, va2);
            }
            (void)vb1; (void)vb2;
        } else {
            loadv(0, va1, va2);
            for (; bk + 2 <= nblk; bk += 2) {
                loadv(4 * (bk + 1), vb1, vb2);
                block(4 * bk, va1, va2);
                loadv(4 * (bk + 2), va1, va2);
                block(4 * (bk + 1), vb1, vb2);
            }
            if (bk < nblk) block(4 * bk, va1, va2);
        }
        for (int r = 4 * nblk; r < R; ++r) row_generic(r, CHECK);
    };
    // wave-uniform dispatch to straight-line bodies
    if (vm.lower >= 2) {
        if (vm.lower & 1) { if (check_any) sweep4(std::true_type{}, std::integral_constant<int, 1>{}); else sweep4(std::false_type{}, std::integral_constant<int, 1>{}); }
        else              { if (check_any) sweep4(std::true_type{}, std::integral_constant<int, 0>{}); else sweep4(std::false_type{}, std::integral_constant<int, 0>{}); }
    } else {
        for (int r = 0; r < R; ++r) { if (check_any) row_generic(r, std::true_type{}); else row_generic(r, std::false_type{}); }
    }
#pragma unroll
    for (int q = 0; q < CAND_PER_WAVE; ++q) {
        const int p = pos[q];
        const long long op = ct.out_off[chunk[q]] + p;
        const double tB = USEBG ? bcov[op] : wave_sum(sB[q]), tBV = USEBG ? bnum[op] : wave_sum(sBV[q]);
        const double tBV2 = wave_sum(sBV2[q]), tB0V = wave_sum(sB0V[q]);
        const bool anyzero = __ballot(zero[q]) != 0ull;
        const int nfr = (int)(ct.frag_off[chunk[q] + 1] - ct.frag_off[chunk[q]]);
        const int *cen = ct.centre + ct.frag_off[chunk[q]];
        const int *iln = ct.ilen + ct.frag_off[chunk[q]];
        const int f0 = wave_lower_bound(cen, 0, nfr, p - vm.w, lane);
        const int f1 = wave_lower_bound(cen, f0, min(nfr, f0 + 4 * WAVE), p + vm.w + 1, lane) ;
#ifdef NATAC_ABL_CAND_TAIL
        const int f1x = f0; (void)f1;
#else
        const int f1x = (f1 == f0 + 4 * WAVE) ? wave_lower_bound(cen, f1, nfr, p + vm.w + 1, lane) : f1;   // very dense windows
#endif
        const double *e = smem + et0 + q * EWP;
        double nl = 0.0, ul = 0.0;
        for (int f = f0 + lane; f < f1x; f += WAVE) {
            const int n = iln[f];
            if (n < vm.lower || n >= vm.upper) continue;
            const int r = n - vm.lower, c = cen[f] - p + vm.w;
            const int hl = floor_half(n - 1), hr = floor_half(n);
            const double b0 = (hl == -hr) ? e[c + A] : e[c + A - hl] * e[c + A + hr];
            nl += log((vm.mat[r * vm.W + c] * b0) / tB0V);
            ul += log((vm.srow[r] * b0) / tB);
        }
        nl = wave_sum(nl);
        ul = wave_sum(ul);
        if (lane == 0 && k0 + q < ncand) {
            const long long o = ct.out_off[chunk[q]] + p;
            const double m1 = tBV / tB;
            const int reads = (int)nuc_cov[o];
            const double var = (double)reads * (tBV2 / tB - m1 * m1);
            out_lr[k0 + q] = anyzero ? __builtin_nan("") : (nl - ul);
            out_var[k0 + q] = var;
            out_z[k0 + q] = norm[o] / sqrt(var);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K8  candidate search on the device: utils.call_peaks (pyatac/utils.py:82-102) applied to norm + smoothed signal as in
// NucChunk.findAllNucs (nucleoatac/NucleosomeCalling.py:297-301).
//   1. NaNs of the combined signal are replaced by the chunk's minimum finite value (all-NaN chunk: no candidates);
//   2. local maxima: y[g] > y[clip(g +- s)] for s = 1..order on the jittered signal y = x * (1 + u[g]); u is the host
//      generated RandomState(25).uniform(0, 1e-12) stream of the reference (same tie-break, bit for bit);
//   3. x[g] >= min_signal, boundary <= g < L - boundary;
//   4. reduce_peaks: visit peaks by decreasing x, keep a peak and drop every peak closer than `sep` (utils.py:56-78).
// ------------------------------------------------------------------------------------------------
constexpr int PEAK_MAX = 2048;   // local maxima per chunk held in LDS.  natac_peaks_chunk (chunks longer than 16,384 bases) keeps the lists of a
                                 // chunk with more in global scratch; the register variants set status bit 1 (host fallback)

// steps 4 + output of the peak search for one chunk (see natac_peaks_chunk): n maxima in (pos, sig, state = 0), ascending
__device__ __forceinline__ void peaks_thin_and_write(int n, int pk_cap, int sep, const double *sig, const int *pos, unsigned char *state,
                                                      int *wave_cnt, int *__restrict__ dst, int *__restrict__ count_out,
                                                      int *__restrict__ status_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, bt = blockDim.x, nw = bt >> 6;   // wave_cnt holds nw <= 16 counts
    const bool overflow = n > pk_cap;
    if (overflow) n = pk_cap;
    // ---- 4. reduce_peaks as parallel rounds.  A kept peak (state 1) has no free peak closer than sep once its round is
    // over, so in later rounds the tests below never meet one: "free or kept" in the first pass only covers peaks that another
    // thread keeps during the same pass
    for (;;) {
        int nfree = 0;
        for (int i = threadIdx.x; i < n; i += bt) {
            if (state[i] != 0) continue;
            const double si = sig[i];
            const int pi = pos[i];
            bool top = true;
            for (int k = i - 1; top && k >= 0 && pi - pos[k] < sep; --k) top = !(state[k] <= 1 && sig[k] > si);
            for (int k = i + 1; top && k < n && pos[k] - pi < sep; ++k) top = !(state[k] <= 1 && sig[k] >= si);
            if (top) state[i] = 1; else ++nfree;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += bt) {
            if (state[i] != 0) continue;
            const int pi = pos[i];
            bool ex = false;
            for (int k = i - 1; !ex && k >= 0 && pi - pos[k] < sep; --k) ex = state[k] == 1;
            for (int k = i + 1; !ex && k < n && pos[k] - pi < sep; ++k) ex = state[k] == 1;
            if (ex) { state[i] = 2; --nfree; }
        }
        if (__syncthreads_or(nfree > 0) == 0) break;
    }
    // ---- kept positions, ascending
    int m_out = 0;
    for (int base = 0; base < n; base += bt) {
        const int i = base + threadIdx.x;
        const bool kp = i < n && state[i] == 1;
        const unsigned long long m = __ballot(kp);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = m_out;
        int tot = 0;
        for (int w = 0; w < nw; ++w) { const int cw = wave_cnt[w]; off += (w < wave) ? cw : 0; tot += cw; }
        if (kp) dst[off + __popcll(m & ((1ull << lane) - 1ull))] = pos[i];
        m_out += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *count_out = m_out;
        if (overflow) atomicOr(status_out, 2);
    }
}

// One workgroup per chunk does the whole search: minimum of the finite values, the jittered local-maximum test on
// segments of `seg` bases staged in LDS, ordered compaction of the maxima, greedy thinning, and writes the kept positions
// (ascending) into the chunk's slot region cand_slot[cap_off[chunk] ...] and their number into count[chunk].
// The thinning runs as parallel rounds: a free peak that out-ranks every free peak closer than `sep` is kept and excludes
// those neighbours -- the same set the reference's sequential visit by decreasing signal produces (a peak is only ever
// excluded by a kept peak of higher rank, and a kept peak is the highest-ranked free one of its neighbourhood at the time it
// is visited).  Rank: larger signal first, ties: the later position first (a stable ascending sort read backwards).
// Dynamic LDS: ys[seg + 2 order (even)] | sig[pk_cap] | pos[pk_cap] (int) | state[pk_cap] (bytes); seg a multiple of 256.
__global__ void __launch_bounds__(256) natac_peaks_chunk(ChunkTable ct, const double *__restrict__ norm,
                                                           const double *__restrict__ smooth, const double *__restrict__ jitter,
                                                           double min_signal, int boundary, int order, int sep, int seg, int pk_cap,
                                                           const long long *__restrict__ cap_off, int *__restrict__ cand_slot,
                                                           int *__restrict__ count, int *__restrict__ status,
                                                           double *__restrict__ big_sig, int *__restrict__ big_pos,
                                                           unsigned char *__restrict__ big_state) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    __shared__ int wave_cnt[16];
    __shared__ int row_cnt[4 * 16];
    const int ysn = (seg + 2 * order + 1) & ~1;
    double *ys = smem;                              // jittered signal, index u <-> base clip(x0 - order + u)
    double *sig = ys + ysn;
    int *pos = (int *)(sig + pk_cap);
    unsigned char *state = (unsigned char *)(pos + pk_cap);   // 0 free, 1 kept, 2 excluded
    const int chunk = blockIdx.x;
    {   // a chunk that can hold more maxima than the LDS lists (L / (order + 1) + 2 > pk_cap: tens of kb and more, what ChunkList.merge
        // makes of adjacent windows) keeps its lists in the global scratch, region cap_off[chunk] ...: no overflow, no host fallback
        const long long c0 = cap_off[chunk];
        const int cap_c = (int)(cap_off[chunk + 1] - c0);
        if (big_sig && cap_c > pk_cap) { sig = big_sig + c0; pos = big_pos + c0; state = big_state + c0; pk_cap = cap_c; }
    }
    const int L = ct.chunk_len[chunk];
    const long long ob = ct.out_off[chunk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // ---- 1. minimum finite value of the combined signal
    double mn = __builtin_inf();
    for (int g0 = threadIdx.x; g0 < L; g0 += 8 * 256) {      // eight independent loads per thread in flight
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int g = g0 + 256 * q;
            v[q] = __builtin_nan("");
            if (g < L) v[q] = smooth ? norm[ob + g] + smooth[ob + g] : norm[ob + g];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (v[q] == v[q]) mn = fmin(mn, v[q]);
    }
    mn = wave_min(mn);
    if (lane == 0) red[wave] = mn;
    __syncthreads();
    const double fillv = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    if (fillv == __builtin_inf()) {                 // all-NaN chunk: nothing
        if (threadIdx.x == 0) count[chunk] = 0;
        return;
    }
    // ---- 2. + 3. local maxima of the jittered signal, thresholds, ordered compaction
    int n = 0;
    for (int x0 = 0; x0 < L; x0 += seg) {
        const int len = min(seg, L - x0);
        __syncthreads();
        for (int u0 = threadIdx.x; u0 < len + 2 * order; u0 += 8 * 256) {
            double v[8], jt[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int u = u0 + 256 * q;
                int g = x0 - order + u;
                g = g < 0 ? 0 : (g > L - 1 ? L - 1 : g);                 // numpy take(..., mode='clip')
                v[q] = smooth ? norm[ob + g] + smooth[ob + g] : norm[ob + g];
                jt[q] = jitter[g];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int u = u0 + 256 * q;
                if (u < len + 2 * order) ys[u] = ((v[q] != v[q]) ? fillv : v[q]) * (1 + jt[q]);
            }
        }
        __syncthreads();
        // the segment's <= 16 rows of 256 bases: local-maximum tests first, then the un-jittered values of the maxima (all
        // loads of a thread in flight), thresholds, and one ordered compaction for all rows
        unsigned pkf = 0;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int t = 256 * it + threadIdx.x;
            bool pk = t < len;
            if (pk) {
                const double y = ys[t + order];
                for (int sft = 1; sft <= order && pk; ++sft) pk = (y > ys[t + order + sft]) && (y > ys[t + order - sft]);
            }
            pkf |= pk ? (1u << it) : 0u;
        }
        double v[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int g = x0 + 256 * it + threadIdx.x;
            v[it] = 0.0;
            if (pkf & (1u << it)) v[it] = smooth ? norm[ob + g] + smooth[ob + g] : norm[ob + g];
        }
        unsigned long long bal[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int g = x0 + 256 * it + threadIdx.x;
            if (v[it] != v[it]) v[it] = fillv;
            const bool pk = (pkf & (1u << it)) && (v[it] >= min_signal) && (g >= boundary) && (g < L - boundary);
            bal[it] = __ballot(pk);
            if (lane == 0) row_cnt[it * 4 + wave] = __popcll(bal[it]);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            int off = n;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int cw = row_cnt[it * 4 + w];
                off += (w < wave) ? cw : 0;
                n += cw;
            }
            if ((bal[it] >> lane) & 1ull) {
                const int i = off + __popcll(bal[it] & ((1ull << lane) - 1ull));
                if (i < pk_cap) { pos[i] = x0 + 256 * it + threadIdx.x; sig[i] = v[it]; state[i] = 0; }
            }
        }
    }
    __syncthreads();
    peaks_thin_and_write(n, pk_cap, sep, sig, pos, state, wave_cnt, cand_slot + cap_off[chunk], count + chunk, status + chunk);
}

// The same search for batches whose chunks all fit one segment (L <= BT NJ; BT = 256 threads, or 1,024 for chunks up to 16,384
// bases: gfx950 gives a workgroup up to 160 KB of LDS): every thread keeps its NJ bases in registers,
// so the signal is read from memory once, with all loads of a thread in flight together (the general kernel above walks
// the chunk three times with one dependent load per iteration).  LDS as above with seg = 256 NJ.
template <int NJ, int BT>
__global__ void __launch_bounds__(BT) natac_peaks_chunk_reg(ChunkTable ct, const double *__restrict__ norm,
                                                               const double *__restrict__ smooth, const double *__restrict__ jitter,
                                                               double min_signal, int boundary, int order, int sep, int pk_cap,
                                                               const long long *__restrict__ cap_off, int *__restrict__ cand_slot,
                                                               int *__restrict__ count, int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = BT / 64;
    __shared__ double red[NW];
    __shared__ int wave_cnt[16];
    __shared__ int row_cnt[NW * NJ];
    const int ysn = (BT * NJ + 2 * order + 1) & ~1;
    double *ys = smem;                              // jittered signal, index u <-> base clip(u - order)
    double *sig = ys + ysn;
    int *pos = (int *)(sig + pk_cap);
    unsigned char *state = (unsigned char *)(pos + pk_cap);
    const int chunk = blockIdx.x;
    const int L = ct.chunk_len[chunk];
    const long long ob = ct.out_off[chunk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v[NJ], jt[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int g = threadIdx.x + BT * j;
        v[j] = __builtin_nan("");
        jt[j] = 0.0;
        if (g < L) {
            v[j] = smooth ? norm[ob + g] + smooth[ob + g] : norm[ob + g];
            jt[j] = jitter[g];
        }
    }
    double mn = __builtin_inf();
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (v[j] == v[j]) mn = fmin(mn, v[j]);
    mn = wave_min(mn);
    if (lane == 0) red[wave] = mn;
    __syncthreads();
    double fillv = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) fillv = fmin(fillv, red[w]);
    if (fillv == __builtin_inf()) {                 // all-NaN chunk: nothing
        if (threadIdx.x == 0) count[chunk] = 0;
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int g = threadIdx.x + BT * j;
        if (v[j] != v[j]) v[j] = fillv;
        if (g < L) {
            const double y = v[j] * (1 + jt[j]);
            ys[order + g] = y;
            if (g == 0)
                for (int u = 0; u < order; ++u) ys[u] = y;                 // numpy take(..., mode='clip')
            if (g == L - 1)
                for (int u = 0; u < order; ++u) ys[order + L + u] = y;
        }
    }
    __syncthreads();
    // maxima of the thread's bases, then one ordered compaction for all NJ rows (base order = row-major over (j, thread)).
    // The test "larger than both neighbours at every distance up to `order`" leaves a lane after 1.5 comparisons on average, but a wave
    // only when its last lane leaves -- and among 64 consecutive bases one nearly always stays to the end, so every row cost `order`
    // trips.  Two phases (round 5): distances 1 and 2 for every base (thresholds first), the survivors -- about a fifth -- go into a
    // per-wave list, and the remaining distances run over that dense list, 64 candidates at a time; the rows' masks are put back
    // together in LDS.  The list lives where the chunk's peak lists go afterwards (not in use yet).
    unsigned long long bal[NJ];
    {
        const int region = pk_cap * (int)(sizeof(double) + sizeof(int) + 1);
        const int stride = (region / NW) & ~7;
        unsigned long long *wmask = (unsigned long long *)((unsigned char *)sig + wave * stride);     // [NJ] survivors of row j
        unsigned short *wlist = (unsigned short *)(wmask + NJ);
        const int cap = (stride - NJ * 8) / 2;
        const bool two_phase = order > 2 && stride >= NJ * 8 + 2 * 128;      // room for a useful list (else: the direct test below)
        if (stride < NJ * 8) {
            // a large `order` shrinks the peak lists (pk_cap = maxL / (order + 1)) below even the NJ masks of a wave (maxL 4,096, order 150:
            // 104 bytes for 128): the masks then stay in registers, every row takes the direct test
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int g = threadIdx.x + BT * j;
                bool pk = g < L && (v[j] >= min_signal) && (g >= boundary) && (g < L - boundary);
                if (pk) {
                    const double y = ys[g + order];
                    for (int sft = 1; sft <= order && pk; ++sft) pk = (y > ys[g + order + sft]) && (y > ys[g + order - sft]);
                }
                bal[j] = __ballot(pk);
                if (lane == 0) row_cnt[j * NW + wave] = __popcll(bal[j]);
            }
        } else {
        if (lane < NJ) wmask[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        int cnt = 0;
        const int o1 = order < 2 ? order : 2;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int g = threadIdx.x + BT * j;
            bool pk = g < L && (v[j] >= min_signal) && (g >= boundary) && (g < L - boundary);
            double y = 0.0;
            if (pk) {
                y = ys[g + order];
                for (int sft = 1; sft <= o1 && pk; ++sft) pk = (y > ys[g + order + sft]) && (y > ys[g + order - sft]);
            }
            const unsigned long long m = __ballot(pk);
            const int alive = __popcll(m);
            if (two_phase && cnt + alive <= cap) {                      // wave-uniform
                if (pk) wlist[cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
                            (unsigned short)(j * 64 + lane);
                cnt += alive;
            } else {                                                    // no list (or full): finish the row here
                if (pk)
                    for (int sft = o1 + 1; sft <= order && pk; ++sft) pk = (y > ys[g + order + sft]) && (y > ys[g + order - sft]);
                const unsigned long long mf = __ballot(pk);
                if (lane == 0) wmask[j] = mf;
            }
        }
        __builtin_amdgcn_wave_barrier();
        unsigned int *wm32 = (unsigned int *)wmask;
        for (int idx = lane; idx - lane < cnt; idx += 64) {             // wave-uniform trip count
            if (idx < cnt) {
                const int e = wlist[idx], jj = e >> 6, ll = e & 63;
                const int g = (wave << 6) + ll + BT * jj;
                const double y = ys[g + order];
                bool pk = true;
                for (int sft = 3; sft <= order && pk; ++sft) pk = (y > ys[g + order + sft]) && (y > ys[g + order - sft]);
                if (pk) atomicOr(&wm32[2 * jj + (ll >> 5)], 1u << (ll & 31));
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bal[j] = wmask[j];
            if (lane == 0) row_cnt[j * NW + wave] = __popcll(bal[j]);
        }
        }
    }
    __syncthreads();
    int n = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int off = n;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int cw = row_cnt[j * NW + w];
            off += (w < wave) ? cw : 0;
            n += cw;
        }
        if ((bal[j] >> lane) & 1ull) {
            const int i = off + __popcll(bal[j] & ((1ull << lane) - 1ull));
            if (i < pk_cap) { pos[i] = threadIdx.x + BT * j; sig[i] = v[j]; state[i] = 0; }
        }
    }
    __syncthreads();
    peaks_thin_and_write(n, pk_cap, sep, sig, pos, state, wave_cnt, cand_slot + cap_off[chunk], count + chunk, status + chunk);
}

// exclusive prefix sum of per-chunk counts (single workgroup) + total: tiles of 4,096 counts, four consecutive ones per thread
// (coalesced), wave scans by shuffles, the 16 wave totals through LDS, a running carry between tiles
__global__ void __launch_bounds__(1024) natac_scan_counts(const int *__restrict__ count, int nc, long long *__restrict__ offs) {
    __shared__ long long wtot[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    long long carry = 0;
    for (int base = 0; base < nc; base += 4096) {
        const int i0 = base + 4 * t;
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = (i0 + q < nc) ? count[i0 + q] : 0;
        const long long mine = (long long)c[0] + c[1] + c[2] + c[3];
        long long incl = mine;                               // inclusive scan over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        long long before = carry, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const long long v = wtot[w];
            before += (w < wave) ? v : 0;
            total += v;
        }
        long long run = before + incl - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (i0 + q < nc) offs[i0 + q] = run;
            run += c[q];
        }
        carry += total;
        __syncthreads();
    }
    if (t == 0) offs[nc] = carry;
}

__global__ void __launch_bounds__(256) natac_compact_candidates(int nc, const int *__restrict__ count,
                                                                  const long long *__restrict__ offs,
                                                                  const long long *__restrict__ cap_off,
                                                                  const int *__restrict__ cand_slot, int *__restrict__ cand_chunk,
                                                                  int *__restrict__ cand_pos) {
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= nc) return;
    const int lane = threadIdx.x & 63;
    const int n = count[chunk];
    const long long o = offs[chunk];
    const int *src = cand_slot + cap_off[chunk];
    for (int i = lane; i < n; i += 64) { cand_chunk[o + i] = chunk; cand_pos[o + i] = src[i]; }
}

// ------------------------------------------------------------------------------------------------
// K9  OccChunk.callPeaks filter + OccChunk.getNucDist on the device (nucleoatac/Occupancy.py:225-240).
// One workgroup per chunk walks the chunk's occupancy peaks (ascending, as call_peaks returns them): a peak is kept if
// smoothed_lower[pos] > min_occ and cov[pos] > 0 (:229-230); every kept peak adds its window's insert-size histogram
// (fragments centred within +-flank, sizes [0, upper)) divided by its total to the chunk's nuc_dist, in peak order like
// the reference's `nuc_dist += sub_sum` (:236-239).  out_vals[4][n] = occ, lower, upper, reads of every peak (OccPeak).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) natac_occ_peak_dist(ChunkTable ct, const long long *__restrict__ pk_offs,
                                                             const int *__restrict__ pk_pos, const double *__restrict__ occ,
                                                             const double *__restrict__ lower, const double *__restrict__ upper_t,
                                                             const double *__restrict__ cov, double min_occ, int flank, int U,
                                                             long long cap, double *__restrict__ out_vals,
                                                             int *__restrict__ keep, double *__restrict__ nuc_dist) {
    extern __shared__ int hist_s[];                 // [U]
    __shared__ int total_s;
    const int chunk = blockIdx.x;
    const long long ob = ct.out_off[chunk];
    const int nfr = (int)(ct.frag_off[chunk + 1] - ct.frag_off[chunk]);
    const int *cen = ct.centre + ct.frag_off[chunk];
    const int *iln = ct.ilen + ct.frag_off[chunk];
    const int nj = (U + 255) / 256;                 // insert sizes per thread (1 for upper <= 256)
    double nd[4] = {0.0, 0.0, 0.0, 0.0};            // thread t owns sizes t, t + 256, ... (U <= 1024)
    for (long long k = pk_offs[chunk]; k < pk_offs[chunk + 1]; ++k) {
        const int p = pk_pos[k];
        const double lo = lower[ob + p], rd = cov[ob + p];
        const bool kp = lo > min_occ && rd > 0;      // block-uniform
        if (threadIdx.x == 0) {
            out_vals[k] = occ[ob + p];
            out_vals[cap + k] = lo;
            out_vals[2 * cap + k] = upper_t[ob + p];
            out_vals[3 * cap + k] = rd;
            keep[k] = kp ? 1 : 0;
        }
        if (!kp) continue;
        for (int j = threadIdx.x; j < U; j += 256) hist_s[j] = 0;
        if (threadIdx.x == 0) total_s = 0;
        __syncthreads();
        const int f0 = lower_bound_i32(cen, 0, nfr, p - flank);
        const int f1 = lower_bound_i32(cen, f0, nfr, p + flank + 1);
        int mine = 0;
        for (int f = f0 + threadIdx.x; f < f1; f += 256) {
            const int n = iln[f];
            if (n >= 0 && n < U) { atomicAdd(&hist_s[n], 1); ++mine; }
        }
        if (mine) atomicAdd(&total_s, mine);
        __syncthreads();
        const double tot = (double)total_s;
        for (int q = 0; q < nj && q < 4; ++q) {
            const int j = threadIdx.x + 256 * q;
            if (j < U) nd[q] = nd[q] + (double)hist_s[j] / tot;
        }
        __syncthreads();
    }
    for (int q = 0; q < nj && q < 4; ++q) {
        const int j = threadIdx.x + 256 * q;
        if (j < U) nuc_dist[(long long)chunk * U + j] = nd[q];
    }
}

// ------------------------------------------------------------------------------------------------
// drop-in kernels for the Cython functions (single region, absolute coordinates)
// ------------------------------------------------------------------------------------------------
// makeFragmentMat, pyatac/fragments.pyx:17-40 (mat pre-zeroed; float64 atomics are exact for integer counts)
__global__ void natac_fragment_mat(const long long *__restrict__ l, const int *__restrict__ n, long long nf,
                                   long long start, int ncol, int lower, int nrow, double *__restrict__ mat) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < nf; i += stride) {
        const int ilen = n[i];
        const long long row = ilen - lower;
        const long long col = (long long)floor_half(ilen - 1) + l[i] - start;
        if (col >= 0 && col < ncol && row >= 0 && row < nrow) atomicAdd(&mat[row * ncol + col], 1.0);
    }
}

// getInsertions, pyatac/fragments.pyx:43-67 (int32 counts; converted to float64 on the way out)
__global__ void natac_insertions_region(const long long *__restrict__ l, const int *__restrict__ n, long long nf,
                                        long long start, int npos, int lower, int upper, int *__restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < nf; i += stride) {
        const int ilen = n[i];
        if (ilen < lower || ilen >= upper) continue;
        const long long lp = l[i] - start, rp = lp + ilen - 1;
        if (lp >= 0 && lp < npos) atomicAdd(&out[lp], 1);
        if (rp >= 0 && rp < npos) atomicAdd(&out[rp], 1);
    }
}

// getStrandedInsertions, pyatac/fragments.pyx:71-97: left ends -> plus, right ends -> minus
__global__ void natac_stranded_insertions_region(const long long *__restrict__ l, const int *__restrict__ n, long long nf,
                                                 long long start, int npos, int lower, int upper, int *__restrict__ plus,
                                                 int *__restrict__ minus) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < nf; i += stride) {
        const int ilen = n[i];
        if (ilen < lower || ilen >= upper) continue;
        const long long lp = l[i] - start, rp = lp + ilen - 1;
        if (lp >= 0 && lp < npos) atomicAdd(&plus[lp], 1);
        if (rp >= 0 && rp < npos) atomicAdd(&minus[rp], 1);
    }
}

__global__ void natac_i32_to_f64(const int *__restrict__ a, double *__restrict__ b, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = (double)a[i];
}

// getFragmentSizesFromChunkList, pyatac/fragments.pyx:123-145: a fragment counts once per chunk containing its
// centre c (chunks may overlap before ChunkList.merge: the reference loops the chunks and re-fetches).  With the chunk
// starts and the chunk ends sorted INDEPENDENTLY (host),  #{k: cs[k] <= c < ce[k]} = #{cs <= c} - #{ce <= c}  for any set of
// intervals with cs[k] <= ce[k], so a fragment costs two searches instead of a scan over all chunks.  A workgroup takes
// SIZE_SEG consecutive fragments; reads come position-sorted from the BAM, so their centres span a few chunks only: the
// slices of cs / ce that can matter for the segment (found by four bisections per segment) are staged in LDS and searched
// there; segments that span more than SIZE_STAGE chunk bounds search global memory.  Per-workgroup LDS histogram, one
// global atomic per non-empty bin at the end (lh == nullptr: bins too many for LDS, global atomics directly).
constexpr int SIZE_SEG = 2048;     // fragments per segment (8 per thread)
constexpr int SIZE_STAGE = 1024;   // staged chunk starts / ends per segment

__device__ __forceinline__ int upper_bound_i64(const long long *a, int lo, int hi, long long key) {
    // number of entries <= key in the sorted array a[0, hi), searching from lo
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) natac_size_hist(const long long *__restrict__ l, const int *__restrict__ n,
                                                         long long nf, const long long *__restrict__ cs,
                                                         const long long *__restrict__ ce, int nchunks, int lower, int upper,
                                                         int use_lds_hist, unsigned long long *__restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char size_smem[];
    __shared__ long long seg_min, seg_max;
    __shared__ int bnd[4];                                  // cntS(cmin - 1), cntS(cmax), cntE(cmin - 1), cntE(cmax)
    long long *ss = (long long *)size_smem;                 // [SIZE_STAGE] staged chunk starts
    long long *es = ss + SIZE_STAGE;                        // [SIZE_STAGE] staged chunk ends
    unsigned int *lh = (unsigned int *)(es + SIZE_STAGE);   // [upper - lower]
    const int nb = upper - lower;
    if (use_lds_hist) for (int b = threadIdx.x; b < nb; b += 256) lh[b] = 0;
    const long long nseg = (nf + SIZE_SEG - 1) / SIZE_SEG;
    for (long long seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        if (threadIdx.x == 0) { seg_min = 0x7fffffffffffffffLL; seg_max = -0x7fffffffffffffffLL - 1; }
        __syncthreads();
        long long c[SIZE_SEG / 256];
        int il[SIZE_SEG / 256];
        long long mn = 0x7fffffffffffffffLL, mx = -0x7fffffffffffffffLL - 1;
#pragma unroll
        for (int q = 0; q < SIZE_SEG / 256; ++q) {
            const long long i = seg * SIZE_SEG + q * 256 + threadIdx.x;
            il[q] = -1;
            c[q] = 0;
            if (i < nf) {
                const int ilen = n[i];
                if (ilen >= lower && ilen < upper) {
                    il[q] = ilen;
                    c[q] = l[i] + floor_half(ilen - 1);
                    mn = c[q] < mn ? c[q] : mn;
                    mx = c[q] > mx ? c[q] : mx;
                }
            }
        }
        if (mn <= mx) { atomicMin(&seg_min, mn); atomicMax(&seg_max, mx); }
        __syncthreads();
        const long long cmin = seg_min, cmax = seg_max;
        if (cmin > cmax) continue;                          // no countable fragment in the segment (block-uniform)
        if (threadIdx.x < 4) {
            const long long *arr = (threadIdx.x < 2) ? cs : ce;
            bnd[threadIdx.x] = upper_bound_i64(arr, 0, nchunks, (threadIdx.x & 1) ? cmax : cmin - 1);
        }
        __syncthreads();
        const int s0 = bnd[0], s1 = bnd[1], e0 = bnd[2], e1 = bnd[3];
        const bool staged = (s1 - s0 <= SIZE_STAGE) && (e1 - e0 <= SIZE_STAGE);
        if (staged) {
            for (int k = threadIdx.x; k < s1 - s0; k += 256) ss[k] = cs[s0 + k];
            for (int k = threadIdx.x; k < e1 - e0; k += 256) es[k] = ce[e0 + k];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SIZE_SEG / 256; ++q) {
            if (il[q] < 0) continue;
            int a, b;
            if (staged) {
                a = s0 + upper_bound_i64(ss, 0, s1 - s0, c[q]);
                b = e0 + upper_bound_i64(es, 0, e1 - e0, c[q]);
            } else {
                a = upper_bound_i64(cs, s0, s1, c[q]);
                b = upper_bound_i64(ce, e0, e1, c[q]);
            }
            const int cnt = a - b;                          // chunks with cs <= c < ce
            if (cnt > 0) {
                if (use_lds_hist) atomicAdd(&lh[il[q] - lower], (unsigned)cnt);
                else atomicAdd(&hist[il[q] - lower], (unsigned long long)cnt);
            }
        }
        __syncthreads();                                    // ss / es / seg_min are rewritten by the next segment
    }
    __syncthreads();
    if (use_lds_hist)
        for (int b = threadIdx.x; b < nb; b += 256)
            if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

// calculateCov, nucleoatac/multinomial_cov.pyx:20-31.
// mode 0: closed form, partial sums {sum p v^2, sum p v, sum p^2 v^2}: value = r*(S1 - S2^2) where the .pyx's
//         diagonal p(1-p)v^2 and off-diagonal -2 p_i p_j v_i v_j terms regroup to S1 - S2^2.
__global__ void __launch_bounds__(256) natac_cov_closed(const double *__restrict__ p, const double *__restrict__ v,
                                                          long long n, double *__restrict__ partial) {
    __shared__ double red[2][4];
    double s1 = 0, s2 = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double pv = p[i] * v[i];
        s1 = fma(pv, v[i], s1);
        s2 += pv;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
// mode 1: literal pair sum.  Workgroup b owns rows i = b, b+gridDim, ...; for each i the 256 threads stride over
// j >= i with the .pyx's own term expressions.  partial[b] = sum of its terms.
__global__ void __launch_bounds__(256) natac_cov_literal(const double *__restrict__ p, const double *__restrict__ v,
                                                           long long n, double *__restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long long i = blockIdx.x; i < n; i += gridDim.x) {
        const double pi = p[i], vi = v[i];
        for (long long j = i + threadIdx.x; j < n; j += 256) {
            if (j == i) acc += pi * (1 - pi) * (vi * vi);
            else acc += pi * p[j] * -2 * vi * v[j];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}


// ------------------------------------------------------------------------------------------------
// operator-level kernels behind the Track / BiasMat2D / InsertionBiasTrack / calculateOccupancy mirrors
// ------------------------------------------------------------------------------------------------
// utils.smooth (pyatac/utils.py:23-52) on one array.  mode 0 = 'valid' (nout = n-M+1), 1 = 'same' (nout = n, n >= M).
// np.convolve(w, x)[m] = sum_k w[k] x[m-k]; NaNs of x count as 0; norm: divide by the same convolution of the
// not-NaN indicator (0 -> NaN).
__global__ void __launch_bounds__(256) natac_smooth1d(const double *__restrict__ x, long long n, const double *__restrict__ w,
                                                        int M, int mode, int norm, double *__restrict__ y, long long nout) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nout) return;
    const long long m = mode == 0 ? t + (M - 1) : t + (M - 1) / 2;   // index into the 'full' convolution
    double num = 0.0, den = 0.0;
    for (int k = 0; k < M; ++k) {
        const long long i = m - k;
        if (i < 0 || i >= n) continue;
        const double v = x[i];
        if (v == v) { num = fma(w[k], v, num); den += w[k]; }
    }
    y[t] = norm ? (den == 0.0 ? __builtin_nan("") : num / den) : num;
}

// BiasMat2D.makeBiasMat (pyatac/chunkmat2d.py:140-153), dense: mat[i-lower, x] = exp(b[g-(i-1)//2] + b[g+i//2]),
// g = start + x; the i == 1 row is exp(b[g]) (both pattern ones on one cell).  b index = g - track_start.
__global__ void __launch_bounds__(256) natac_bias_mat_dense(const double *__restrict__ b, long long nb, long long off0,
                                                              int ncol, int lower, int nrow, double *__restrict__ mat,
                                                              int *__restrict__ oob) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)nrow * ncol) return;
    const int r = (int)(idx / ncol), x = (int)(idx - (long long)r * ncol);
    const int i = lower + r;
    const int hl = floor_half(i - 1), hr = floor_half(i);
    const long long jl = off0 + x - hl, jr = off0 + x + hr;
    if (jl < 0 || jr < 0 || jl >= nb || jr >= nb) { atomicOr(oob, 1); mat[idx] = __builtin_nan(""); return; }
    mat[idx] = (hl == -hr) ? exp(b[jl]) : exp(b[jl] + b[jr]);
}

// InsertionBiasTrack.computeBias (pyatac/bias.py:85-92) + seq_to_mat (pyatac/seq.py:37-45):
// out[x] = sum_k logpwm[row(seq[x+k]), k]; characters that are not one of the PWM's nucleotides add 0.
// the same score for the packed bias array of a batch: chunk k's sequence window starts at seq_off[k] and holds its
// bias_off[k+1] - bias_off[k] + K - 1 bases; one workgroup per 1024 positions of a chunk (grid.y = chunk)
__global__ void __launch_bounds__(256) natac_pwm_score_chunks(const unsigned char *__restrict__ seq, const long long *__restrict__ seq_off,
                                                                const long long *__restrict__ bias_off, const double *__restrict__ logpwm,
                                                                const unsigned char *__restrict__ nucs, int nrow, int K,
                                                                double *__restrict__ bias) {
    const int chunk = blockIdx.y;
    const long long nb = bias_off[chunk + 1] - bias_off[chunk];
    const unsigned char *s = seq + seq_off[chunk];
    double *o = bias + bias_off[chunk];
    for (long long x = (long long)blockIdx.x * 256 + threadIdx.x; x < nb; x += (long long)gridDim.x * 256) {
        double acc = 0.0;
        for (int r = 0; r < nrow; ++r) {
            double rs = 0.0;
            const unsigned char c = nucs[r];
            for (int k = 0; k < K; ++k) if (s[x + k] == c) rs += logpwm[r * K + k];
            acc += rs;
        }
        o[x] = acc;
    }
}

__global__ void __launch_bounds__(256) natac_pwm_score(const unsigned char *__restrict__ seq, long long n,
                                                         const double *__restrict__ logpwm, const unsigned char *__restrict__ nucs,
                                                         int nrow, int K, double *__restrict__ out) {
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= n - K + 1) return;
    double acc = 0.0;
    for (int r = 0; r < nrow; ++r) {          // same association as the reference's row-by-row correlate
        double rs = 0.0;
        const unsigned char c = nucs[r];
        for (int k = 0; k < K; ++k) if (seq[x + k] == c) rs += logpwm[r * K + k];
        acc += rs;
    }
    out[x] = acc;
}

// calculateOccupancy (nucleoatac/Occupancy.py:104-120) for ONE window given dense inserts / bias vectors -- the literal
// formula (sum_j ins[j] * log(alpha pn[j] + (1-alpha) pf[j]), NaN -> -inf); one thread per alpha, n_alpha <= 128.
// out = {occ, lower, upper}; status 1 when no alpha passes the ratio test (the reference raises ValueError).
__global__ void __launch_bounds__(128) natac_occupancy_single(const double *__restrict__ ins, const double *__restrict__ bias,
                                                                OccModelDev om, double *__restrict__ out, int *__restrict__ status) {
    __shared__ double ll[128];
    const int a = threadIdx.x;
    double sn = 0.0, sf = 0.0;
    for (int j = 0; j < om.upper; ++j) { sn += om.nuc_probs[j] * bias[j]; sf += om.nfr_probs[j] * bias[j]; }
    double v = -__builtin_inf();
    if (a < om.n_alpha) {
        const double al = om.alphas[a], be = 1 - al;
        double acc = 0.0;
        for (int j = 0; j < om.upper; ++j) {
            const double pn = (om.nuc_probs[j] * bias[j]) / sn, pf = (om.nfr_probs[j] * bias[j]) / sf;
            acc += log(al * pn + be * pf) * ins[j];
        }
        v = (acc != acc) ? -__builtin_inf() : acc;
    }
    ll[a] = v;
    __syncthreads();
    if (a == 0) {
        double mx = ll[0];
        int im = 0;
        for (int k = 1; k < om.n_alpha; ++k) if (ll[k] > mx) { mx = ll[k]; im = k; }
        int lo = -1, hi = -1;
        for (int k = 0; k < om.n_alpha; ++k) if (2 * (mx - ll[k]) < om.cutoff) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { out[0] = om.alphas[im]; out[1] = out[2] = __builtin_nan(""); *status = 1; }
        else { out[0] = om.alphas[im]; out[1] = om.alphas[lo]; out[2] = om.alphas[hi]; *status = 0; }
    }
}


// signal.correlate(sub, vmat, mode='valid')[0] on dense matrices (nucleoatac/NucleosomeCalling.py:34-36, 60-63):
// out[g] = sum_{r<R} sum_{c<W} sub[r, g+c] * vmat[r, c].  Operator-level entry for SignalTrack / BiasTrack on
// materialised matrices (the batched path never builds them).
__global__ void __launch_bounds__(256) natac_correlate_dense(const double *__restrict__ sub, long long ncol,
                                                               const double *__restrict__ vmat, int R, int W,
                                                               double *__restrict__ out, long long nout) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= nout) return;
    double acc = 0.0;
    for (int r = 0; r < R; ++r) {
        const double *row = sub + (long long)r * ncol + g;
        const double *vr = vmat + r * W;
        double racc = 0.0;
        for (int c = 0; c < W; ++c) racc = fma(row[c], vr[c], racc);
        acc += racc;
    }
    out[g] = acc;
}

}  // namespace natac
