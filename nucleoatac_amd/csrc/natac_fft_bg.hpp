// natac_fft_bg.hpp -- dense background correlation through hand-written fp64 FFTs (gfx950).
//
// Same quantity as natac_background (natac_kernels.hpp): num[g] = sum_r sum_c B[lower+r, g-w+c] V[r,c], i.e. the sum over
// V-plot rows of 1-D cross-correlations of the product row P_r with the template row V_r (NucleosomeCalling.py:60-63;
// the reference's scipy.signal.correlate also picks an FFT for these sizes).  Direct evaluation costs W = 121 FMA per
// base and row; here
//   * two rows are packed into one complex signal  z = P_a + i P_b  and the template pair into  k = V_a + i V_b :
//       sum_c z[g+c] conj(k[c]) = (P_a*V_a + P_b*V_b)[g] + i (...)   -> the real part is the sum of both correlations;
//   * one wave transforms z with a 512-point radix-8 FFT (8 complex values per lane, three in-register 8-point DFTs, two
//     LDS transposes with conflict-free layouts), multiplies by conj(K) (K = FFT(k), precomputed with the SAME transform,
//     so no bit-reversal is ever undone) and ACCUMULATES IN THE FREQUENCY DOMAIN over all row pairs;
//   * one inverse FFT per tile gives 512 - W + 1 = 392 valid outputs (no wrap-around inside the valid range).
// Cost per row pair and tile: ~350 fp64 ops per lane instead of 2 x 121 x 6.1 = 1480 for the same outputs (~4x fewer),
// numerics: forward error ~1e-15 relative to the tile's signal norm (tests: <= 1e-12 vs the direct kernel).
#pragma once
#include "natac_kernels.hpp"

namespace natac {

#ifndef NATAC_FFT_HFOLD
#define NATAC_FFT_HFOLD 1      // dft8: 1/sqrt2 folded into the output butterflies of the odd half (tools/fft_ab.sh "0 1" NATAC_FFT_HFOLD)
#endif
constexpr int FFT_N = 512;
constexpr int FFT_LA = 576;   // layout A: p + 8 * (p >> 6)
constexpr int FFT_LB = 520;   // layout B: (p & 7) * 65 + (p >> 3)
// Extended tiles (round 5).  A 512-point circular correlation holds, besides its TV = 512 - W + 1 exact outputs, FFT_EXT outputs on each
// side that are wrong by a few terms only: output TV - 1 + m (m = 1..16) wraps its last m template columns onto the tile's first
// samples, output -k (circular index 512 - k) its first k columns onto the tile's last samples.  The edge pass at the end of bg_fft_tile
// (bg_edge_side) replaces those m (k) wrapped products per row by the true ones -- 2 x 136 multiply-adds per row and tile, summed
// directly -- so a tile yields TV + 2 FFT_EXT outputs: a 2,120-base chunk takes 5 transforms per row pair instead of 6.  Which tiles of
// a chunk are extended is the host's decision (natac_api.hip, bg_chunk_tiling: the cheapest mix); the flag travels in tiles[].y.
constexpr int FFT_EXT = 16;
constexpr int FFT_EXT_BIT = 1 << 30;
// exp(bias) entries of a tile that feed bases of its chunk (the rest of the staged window is zero): ext = FFT_EXT or 0
__device__ __forceinline__ int bg_tile_need(int TV, int W, int A, int Bh, int L, int x0, int ext) {
    return ext + min(TV + ext, L - x0) + W - 1 + A + Bh;
}

// 8-point DFT in registers, natural order in and out.  INV: conjugate twiddles (unnormalised inverse).
template <bool INV>
__device__ __forceinline__ void dft8(double (&re)[8], double (&im)[8]) {
    const double h = 0.70710678118654752440;
    double ar[8], ai[8];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        ar[n] = re[n] + re[n + 4]; ai[n] = im[n] + im[n + 4];
        ar[n + 4] = re[n] - re[n + 4]; ai[n + 4] = im[n] - im[n + 4];
    }
    // even outputs: DFT4 of a0..a3
    {
        const double b0r = ar[0] + ar[2], b0i = ai[0] + ai[2], b2r = ar[0] - ar[2], b2i = ai[0] - ai[2];
        const double b1r = ar[1] + ar[3], b1i = ai[1] + ai[3];
        const double tr = ar[1] - ar[3], ti = ai[1] - ai[3];
        const double b3r = INV ? -ti : ti, b3i = INV ? tr : -tr;          // * (-i) forward, * (+i) inverse
        re[0] = b0r + b1r; im[0] = b0i + b1i; re[4] = b0r - b1r; im[4] = b0i - b1i;
        re[2] = b2r + b3r; im[2] = b2i + b3i; re[6] = b2r - b3r; im[6] = b2i - b3i;
    }
    // odd outputs: DFT4 of a4, a5 W, a6 W^2, a7 W^3  with W = exp(-+ i pi/4)
    {
        const double c0r = ar[4], c0i = ai[4];
        double c2r, c2i;
        if (!INV) { c2r = ai[6]; c2i = -ar[6]; }                              // -i
        else      { c2r = -ai[6]; c2i = ar[6]; }                              // +i
        const double d0r = c0r + c2r, d0i = c0i + c2i, d2r = c0r - c2r, d2i = c0i - c2i;
#if NATAC_FFT_HFOLD
        // c1 = h p1, c3 = h p3 with p1 = a5 (1 -+ i), p3 = a7 (-1 -+ i): the factor h = 1/sqrt2 is not applied to p1, p3 but folded
        // into the four output butterflies as FMAs (out = d +- h (p1 +- p3)): 4 + 4 additions and 8 FMAs instead of 4 + 4 + 8
        // additions and 4 multiplications
        double p1r, p1i, p3r, p3i;
        if (!INV) { p1r = ar[5] + ai[5]; p1i = ai[5] - ar[5]; p3r = ai[7] - ar[7]; p3i = -(ai[7] + ar[7]); }
        else      { p1r = ar[5] - ai[5]; p1i = ai[5] + ar[5]; p3r = -(ar[7] + ai[7]); p3i = ar[7] - ai[7]; }
        const double e1r = p1r + p3r, e1i = p1i + p3i;       // d1 / h
        const double ur = p1r - p3r, ui = p1i - p3i;         // t / h;  d3 = -+ i t
        re[1] = fma(h, e1r, d0r); im[1] = fma(h, e1i, d0i); re[5] = fma(-h, e1r, d0r); im[5] = fma(-h, e1i, d0i);
        if (!INV) { re[3] = fma(h, ui, d2r); im[3] = fma(-h, ur, d2i); re[7] = fma(-h, ui, d2r); im[7] = fma(h, ur, d2i); }
        else      { re[3] = fma(-h, ui, d2r); im[3] = fma(h, ur, d2i); re[7] = fma(h, ui, d2r); im[7] = fma(-h, ur, d2i); }
#else
        double c1r, c1i, c3r, c3i;
        if (!INV) {
            c1r = (ar[5] + ai[5]) * h; c1i = (ai[5] - ar[5]) * h;            // (1 - i)/sqrt2
            c3r = (ai[7] - ar[7]) * h; c3i = -(ai[7] + ar[7]) * h;           // (-1 - i)/sqrt2
        } else {
            c1r = (ar[5] - ai[5]) * h; c1i = (ai[5] + ar[5]) * h;            // (1 + i)/sqrt2
            c3r = -(ar[7] + ai[7]) * h; c3i = (ar[7] - ai[7]) * h;           // (-1 + i)/sqrt2
        }
        const double d1r = c1r + c3r, d1i = c1i + c3i;
        const double tr = c1r - c3r, ti = c1i - c3i;
        const double d3r = INV ? -ti : ti, d3i = INV ? tr : -tr;
        re[1] = d0r + d1r; im[1] = d0i + d1i; re[5] = d0r - d1r; im[5] = d0i - d1i;
        re[3] = d2r + d3r; im[3] = d2i + d3i; re[7] = d2r - d3r; im[7] = d2i - d3i;
#endif
    }
}

// eight doubles base[64 j], j = 0..7, as eight ds_read_b64 (2 LDS cycles each).  Left to the compiler these become four
// ds_read2st64_b64, which move the same bytes at half the rate (MI355X_MICROARCH.md, LDS table: 8 cycles per 1 KiB against
// 2 x 2) -- and the FFT kernel is bound by the LDS pipe.  The values are only valid after lds_wait16n<N>().
__device__ __forceinline__ void lds_read8_b64(double (&v)[8], const double *base) {
    const unsigned a = (unsigned)(uintptr_t)base;     // LDS offset = low half of the generic pointer
    asm volatile("ds_read_b64 %0, %1" : "=v"(v[0]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(v[1]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(v[2]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:1536" : "=v"(v[3]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(v[4]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:2560" : "=v"(v[5]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(v[6]) : "v"(a));
    asm volatile("ds_read_b64 %0, %1 offset:3584" : "=v"(v[7]) : "v"(a));
}
// ---- the edge pass of extended tiles: a small matrix product (the one place of this library where the matrix pipe fits) --------------
// Side 0 = left edge (outputs x0 - 16 .. x0 - 1), side 1 = right edge (outputs x0 + TV .. x0 + TV + 15).  Sample s of a side (lane
// t = s): the sample outside the tile that the outputs need (`out`: -16 + s on the left, 512 + s on the right) and the sample inside
// that the circular transform used in its place (`in`: 496 + s, s).  D_r[s] = P_r[out] - P_r[in] per row; output number o of the side
// is short of  sum_r sum_s D_r[s] C_r[i]  with i = the distance between output and sample (right: o - s for s <= o; left: s - o for
// s >= o) and C_r[i] = s_r V_r[W - 1 - i] (right), s_r V_r[i] (left) -- the template column that sample meets -- and its window sum
// of  sum_r s_r P_r[out]  over the same samples.  So per tile and side OUT[s][i] = sum_r D_r[s] C_r[i], a 16 x R by R x 16 product,
// and the outputs are sums along its anti-diagonals.  Round 5 built this first as an outer product on the vector pipe (every lane
// needs the 16 C_r[i] of every row as uniform operands: 146 x 2 x 128 bytes through the scalar cache per wave, the counters showed
// the waves waiting on it for half of their time: 0.78 ms per 100 k tiles).  v_mfma_f64_16x16x4_f64 takes BOTH operands distributed
// over the lanes (0.57 ms as a kernel of its own, one wave per tile with the windows staged again; 0.45 ms here, at the end of the wave
// that just transformed the tile and still has its exp(bias) window in LDS):
// A[t = lane & 15][k = lane >> 4] = D of row 4 j + k, B[k][i = lane & 15] = one coalesced 512-byte load of the table per step; a tile
// and side is ceil(R / 4) instructions and every D is formed once.  fp64 MFMA has no rate advantage over fp64 FMAs on gfx950
// (profiles/r1/probe_fp64_mfma_vs_valu.txt); what it buys here is operand delivery.
// bg_edge_side: one side for one tile, by one wave: returns (lanes 0..15: output number lane & 15 of the side) the correction of the
// numerator and of the window sum.  Windows (entry e of the tile's staged exp(bias) window, E):
//   wl_lo[w] = E(w), wl_hi[w] = E(512 + w)                       left factors  (entry 16 + A - fh(i - 1) + u of sample u, row i)
//   wr_lo[w] = E(A + hr0 + w), wr_hi[w] = E(512 + A + hr0 + w)   right factors (entry 16 + A + fh(i) + u)
// left edge:  out = sample -16 + t -> wl_lo[dl + t] wr_lo[dr + t];  in = sample 496 + t -> wl_hi[dl + t] wr_hi[dr + t]
// right edge: out = sample 512 + t -> wl_hi[16 + dl + t] wr_hi[16 + dr + t];  in = sample t -> wl_lo[16 + dl + t] wr_lo[16 + dr + t]
// with dl = A - fh(i - 1), dr = fh(i) - hr0; a lane's rows are 4 j + k: dl falls and dr rises by 2 per step (rows past R in the last
// step would index up to two entries outside a window: their loads are predicated off).  sws = the rows' weights (4 NJ doubles, zero past R), sc =
// EDGE_SCRATCH doubles of LDS, mtab = natac_fft_edge_table_mfma's table.  (The accumulators stay in VGPRs because the kernel's register
// budget is <= 256: with 512 the compiler picks the AGPR form and copies sixteen registers around every pair of instructions.)
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int EDGE_RS = FFT_EXT + 1;       // stride of a sample's row in the reduction scratch
constexpr int EDGE_SCRATCH = FFT_EXT * EDGE_RS + 64 + FFT_EXT + 128;
__device__ __forceinline__ void bg_edge_side(const int side, const double *wl_lo, const double *wl_hi, const double *wr_lo, const double *wr_hi,
                                             const double *sws, double *sc, const double *__restrict__ mtab, const int NJ, const int R,
                                             const int lower, const int A, const int hr0, const int lane, double &corr_out, double &qc_out) {
    const int t = lane & (FFT_EXT - 1), k = lane >> 4;
    double *sq = sc + FFT_EXT * EDGE_RS, *sqt = sq + 64, *sp = sqt + FFT_EXT;
    const int dl0 = A - floor_half(lower + k - 1), dr0 = floor_half(lower + k) - hr0;
    const int NJf = R / 4;                       // steps whose four rows all exist
    {
        const double *pl_o = (side ? wl_hi + FFT_EXT : wl_lo) + dl0 + t, *pr_o = (side ? wr_hi + FFT_EXT : wr_lo) + dr0 + t;
        const double *pl_i = (side ? wl_lo + FFT_EXT : wl_hi) + dl0 + t, *pr_i = (side ? wr_lo + FFT_EXT : wr_hi) + dr0 + t;
        const double *mt = mtab + (size_t)side * NJ * 64 + lane;
        const double *sw = sws + k;
        d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        double qx = 0.0;
        // step u of a trip: rows 4 (j + u) + k; the pointers sit at step j
#define NATAC_EDGE_STEP(u, acc)                                                                                  \
        {                                                                                                        \
            const double po = pl_o[-2 * (u)] * pr_o[2 * (u)], pi = pl_i[-2 * (u)] * pr_i[2 * (u)];             \
            qx = fma(sw[4 * (u)], po, qx);                                                                       \
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(po - pi, mt[64 * (u)], acc, 0, 0, 0);                     \
        }
        int j = 0;
        for (; j + 4 <= NJf; j += 4) {
            NATAC_EDGE_STEP(0, acc0) NATAC_EDGE_STEP(1, acc1) NATAC_EDGE_STEP(2, acc0) NATAC_EDGE_STEP(3, acc1)
            pl_o -= 8; pl_i -= 8; pr_o += 8; pr_i += 8; mt += 256; sw += 16;
        }
        for (; j < NJf; ++j) {
            NATAC_EDGE_STEP(0, acc0)
            pl_o -= 2; pl_i -= 2; pr_o += 2; pr_i += 2; mt += 64; sw += 4;
        }
#undef NATAC_EDGE_STEP
        if (j < NJ) {                                // the last rows: lanes past R contribute zeros (their window entries are not the model's)
            const bool valid = 4 * j + k < R;      // rows past R would index up to two entries below / above the windows: not read at all
            double lo = 0.0, ro = 0.0, li = 0.0, ri = 0.0;
            if (valid) { lo = pl_o[0]; ro = pr_o[0]; li = pl_i[0]; ri = pr_i[0]; }
            const double po = lo * ro, pi = li * ri;
            qx = fma(sw[0], po, qx);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(po - pi, mt[0], acc1, 0, 0, 0);
        }
        const d4_t acc = acc0 + acc1;
        // OUT[t' = k + 4 q][i = lane & 15] = acc[q]; a lane's column sums are partial over its rows
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) sc[(k + 4 * q) * EDGE_RS + t] = acc[q];
        sq[lane] = qx;
        __builtin_amdgcn_wave_barrier();
        if (lane < FFT_EXT) sqt[lane] = ((sq[lane] + sq[16 + lane]) + sq[32 + lane]) + sq[48 + lane];     // the missing column sum of sample `lane`
        __builtin_amdgcn_wave_barrier();
        // output number o = t of the side:
        //   right: m = o + 1, samples 512 + s for s <= o, OUT[s][o - s];  left: output x0 - (16 - o), samples s >= o, OUT[s][s - o]
        // lane (k, o) adds the samples s = k, k + 4, k + 8, k + 12; the four partial sums are added in the order of k
        double corr = 0.0, qc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sidx = k + 4 * q, ix = side ? t - sidx : sidx - t;
            const double a = sc[sidx * EDGE_RS + max(ix, 0)], qq = sqt[sidx];
            corr += ix >= 0 ? a : 0.0;
            qc += ix >= 0 ? qq : 0.0;
        }
        sp[lane] = corr; sp[64 + lane] = qc;
        __builtin_amdgcn_wave_barrier();
        corr_out = ((sp[t] + sp[16 + t]) + sp[32 + t]) + sp[48 + t];
        qc_out = ((sp[64 + t] + sp[80 + t]) + sp[96 + t]) + sp[112 + t];
    }
}

// ---- explicit LDS instructions of the skewed pair loop (bg_fft_tile) -----------------------------------------------------------------
// The skewed loop keeps two transforms of one wave in flight, so its waits must name HOW MANY of the wave's LDS instructions may still
// be outstanding (LDS instructions of a wave execute and return in issue order; lgkmcnt counts them).  The compiler counts only the LDS
// instructions it emitted itself, so every LDS access of that loop is inline asm and every wait is written out.
typedef double d2v __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ void ds_st128(unsigned a, double xr, double xi) {
    const d2v v = {xr, xi};
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(a), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void ds_ld128(d2v &v, unsigned a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
}
// eight complex values a + STRIDE j (bytes)
template <int STRIDE> __device__ __forceinline__ void ds_ld128x8(d2v (&v)[8], unsigned a) {
    ds_ld128<0 * STRIDE>(v[0], a); ds_ld128<1 * STRIDE>(v[1], a); ds_ld128<2 * STRIDE>(v[2], a); ds_ld128<3 * STRIDE>(v[3], a);
    ds_ld128<4 * STRIDE>(v[4], a); ds_ld128<5 * STRIDE>(v[5], a); ds_ld128<6 * STRIDE>(v[6], a); ds_ld128<7 * STRIDE>(v[7], a);
}
// wait until at most N of the wave's LDS instructions are outstanding; the operands tie every later use of the eight values to it
template <int N> __device__ __forceinline__ void lds_wait_c8(d2v (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N)
                 : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait16n(double (&x)[8], double (&y)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%16)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                   "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])
                 : "n"(N)
                 : "memory");
}

struct FftTwiddles {            // per-lane twiddles, loaded once per kernel
    double w1r[8], w1i[8];      // W_512^(lane * m)
    double w2r[8], w2i[8];      // W_64^((lane & 7) * m)
};

__device__ __forceinline__ void fft_load_twiddles(FftTwiddles &t, const double *__restrict__ tw /* [512][2] cos, -sin */, int lane) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = (lane * m) & 511, b = (((lane & 7) * m) * 8) & 511;
        t.w1r[m] = tw[2 * a]; t.w1i[m] = tw[2 * a + 1];
        t.w2r[m] = tw[2 * b]; t.w2i[m] = tw[2 * b + 1];
    }
}

// template spectra as [pair][m][lane][re, im]: a lane's two values of a bin arrive in ONE 16-byte load (8 instead of 16 requests per pair)
#define NATAC_FFT_KIDX(m, lane, c) (((m) * 64 + (lane)) * 2 + (c))
#ifndef NATAC_FFT_TW_EARLY
#define NATAC_FFT_TW_EARLY 1       // the per-lane twiddles requested in front of the window staging instead of behind the conditioning test
#endif                             // (both switches, tools/r5_ext6.sh, 100 k extended tiles, five runs each: 7.51 -> 7.44 -> 7.39 ms)
#ifndef NATAC_FFT_EPI_PREFETCH
#define NATAC_FFT_EPI_PREFETCH 1   // the epilogue's nuc_cov / raw inputs requested before the inverse transform (tools/fft_ab.sh "0 1" NATAC_FFT_EPI_PREFETCH)
#endif
// complex scratch of the transposes: interleaved double2 (two planes of doubles with 8-byte accesses measured 5 % slower)
#define CST(p, i, xr, xi) do { (p)[(i)] = make_double2((xr), (xi)); } while (0)
#define CLD(p, i, xr, xi) do { const double2 v_ = (p)[(i)]; (xr) = v_.x; (xi) = v_.y; } while (0)

// the two halves of fft512_fwd after the first in-register DFT: twiddle + transpose 1 (so that the caller can place loads
// between them), then DFT + twiddle + transpose 2 + DFT
__device__ __forceinline__ void fft512_fwd_t1(double (&re)[8], double (&im)[8], const FftTwiddles &t, double2 *sa, int lane) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {                      // twiddle + transpose 1 (layout A: lane + 72 m)
        const double xr = fma(re[m], t.w1r[m], -(im[m] * t.w1i[m])), xi = fma(re[m], t.w1i[m], im[m] * t.w1r[m]);
        CST(sa, lane + 72 * m, m ? xr : re[0], m ? xi : im[0]);
    }
    __builtin_amdgcn_wave_barrier();
    const int m2 = lane >> 3, n1 = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) CLD(sa, 72 * m2 + n1 + 8 * j, re[j], im[j]);
    __builtin_amdgcn_wave_barrier();   // sb* may alias sa*
}
__device__ __forceinline__ void fft512_fwd_rest(double (&re)[8], double (&im)[8], const FftTwiddles &t, double2 *sb, int lane) {
    const int m2 = lane >> 3, n1 = lane & 7;
    dft8<false>(re, im);
#pragma unroll
    for (int m = 0; m < 8; ++m) {                      // twiddle + transpose 2 (layout B: element n1 of butterfly 8 m2 + m)
        const double xr = fma(re[m], t.w2r[m], -(im[m] * t.w2i[m])), xi = fma(re[m], t.w2i[m], im[m] * t.w2r[m]);
        CST(sb, n1 * 65 + 8 * m2 + m, m ? xr : re[0], m ? xi : im[0]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < 8; ++n) CLD(sb, n * 65 + lane, re[n], im[n]);
    dft8<false>(re, im);
    __builtin_amdgcn_wave_barrier();
}

// forward 512-point FFT of the wave's data (lane n, register j <-> element n + 64 j); result: lane b, register m holds the
// bin of "stage-3 butterfly b, output m" (a fixed permutation of the frequencies, identical for signal and template).
// sa / sb: the wave's LDS scratch, FFT_LA and FFT_LB doubles for the real and for the imaginary parts each.
__device__ __forceinline__ void fft512_fwd(double (&re)[8], double (&im)[8], const FftTwiddles &t, double2 *sa, double2 *sb, int lane) {
    dft8<false>(re, im);
    fft512_fwd_t1(re, im, t, sa, lane);
    fft512_fwd_rest(re, im, t, sb, lane);
}

// exact inverse of fft512_fwd up to the factor 512
__device__ __forceinline__ void fft512_inv(double (&re)[8], double (&im)[8], const FftTwiddles &t, double2 *sa, double2 *sb, int lane) {
    dft8<true>(re, im);
#pragma unroll
    for (int n = 0; n < 8; ++n) CST(sb, n * 65 + lane, re[n], im[n]);
    __builtin_amdgcn_wave_barrier();
    const int m2 = lane >> 3, n1 = lane & 7;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        double xr, xi;
        CLD(sb, n1 * 65 + 8 * m2 + m, xr, xi);
        re[m] = m ? fma(xr, t.w2r[m], xi * t.w2i[m]) : xr;                 // * conj(W)
        im[m] = m ? fma(xi, t.w2r[m], -(xr * t.w2i[m])) : xi;
    }
    __builtin_amdgcn_wave_barrier();
    dft8<true>(re, im);
#pragma unroll
    for (int j = 0; j < 8; ++j) CST(sa, 72 * m2 + n1 + 8 * j, re[j], im[j]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        double xr, xi;
        CLD(sa, lane + 72 * m, xr, xi);
        re[m] = m ? fma(xr, t.w1r[m], xi * t.w1i[m]) : xr;
        im[m] = m ? fma(xi, t.w1r[m], -(xr * t.w1i[m])) : xi;
    }
    dft8<true>(re, im);
    __builtin_amdgcn_wave_barrier();
}

// ---- the skewed pair loop's pieces ------------------------------------------------------------------------------------------------------
// Same transform as fft512_fwd -- the same operations on the same values in the same order, so the same bits -- cut at its two transposes
// so that bg_fft_tile can put ANOTHER row pair's arithmetic between a transpose's store + read-back and the first use of what comes back:
//   S1 = products + column sums + first DFT + twiddle 1        -> W1 (8 stores, layout A), R1 (8 loads)
//   S2 = second DFT + twiddle 2                                 -> W2 (8 stores, layout B), R2 (8 loads)
//   S3 = third DFT + acc += Z conj(K)
// Byte addresses (LDS offsets) of a lane, `c` = the wave's complex scratch:
//   W1: c + 16 (lane + 72 m)          R1: c + 16 (72 m2 + n1 + 8 j)        W2: c + 16 (65 n1 + 8 m2 + m)        R2: c + 16 (65 n + lane)
__device__ __forceinline__ void skew_tw1(double (&re)[8], double (&im)[8], const FftTwiddles &t) {
#pragma unroll
    for (int m = 1; m < 8; ++m) {
        const double xr = fma(re[m], t.w1r[m], -(im[m] * t.w1i[m])), xi = fma(re[m], t.w1i[m], im[m] * t.w1r[m]);
        re[m] = xr; im[m] = xi;
    }
}
__device__ __forceinline__ void skew_w1(const double (&re)[8], const double (&im)[8], unsigned a) {
    ds_st128<0 * 1152>(a, re[0], im[0]); ds_st128<1 * 1152>(a, re[1], im[1]); ds_st128<2 * 1152>(a, re[2], im[2]); ds_st128<3 * 1152>(a, re[3], im[3]);
    ds_st128<4 * 1152>(a, re[4], im[4]); ds_st128<5 * 1152>(a, re[5], im[5]); ds_st128<6 * 1152>(a, re[6], im[6]); ds_st128<7 * 1152>(a, re[7], im[7]);
}
// second DFT of the values R1 brought (skew_s2), then twiddle 2 and W2 (skew_w2)
__device__ __forceinline__ void skew_s2(const d2v (&A)[8], double (&re)[8], double (&im)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { re[j] = A[j].x; im[j] = A[j].y; }
    dft8<false>(re, im);
}
__device__ __forceinline__ void skew_w2(const double (&re)[8], const double (&im)[8], const FftTwiddles &t, unsigned a) {
#define NATAC_SKEW_W2(m) { const double xr = fma(re[m], t.w2r[m], -(im[m] * t.w2i[m])), xi = fma(re[m], t.w2i[m], im[m] * t.w2r[m]); ds_st128<16 * (m)>(a, xr, xi); }
    ds_st128<0>(a, re[0], im[0]);
    NATAC_SKEW_W2(1) NATAC_SKEW_W2(2) NATAC_SKEW_W2(3) NATAC_SKEW_W2(4) NATAC_SKEW_W2(5) NATAC_SKEW_W2(6) NATAC_SKEW_W2(7)
#undef NATAC_SKEW_W2
}
// third DFT of the values R2 brought and acc += Z conj(K)
__device__ __forceinline__ void skew_s3(const d2v (&B)[8], const double (&kr)[8], const double (&ki)[8], double (&accr)[8], double (&acci)[8]) {
    double re[8], im[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { re[j] = B[j].x; im[j] = B[j].y; }
    dft8<false>(re, im);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        accr[m] = fma(re[m], kr[m], fma(im[m], ki[m], accr[m]));
        acci[m] = fma(im[m], kr[m], fma(-re[m], ki[m], acci[m]));
    }
}

// K[pair] = FFT(s_a V_a + i s_b V_b) in the layout fft512_fwd produces; kout[pair][m][lane][re, im] (NATAC_FFT_KIDX).  One wave per pair.
// The insert-size weights s_r (BiasMat2D.normByInsertDist, chunkmat2d.py:154-156) are folded into the template: the
// correlation is linear, so sum_r (s_r B0_r) * V_r = sum_r B0_r * (s_r V_r) and the kernel transforms the unweighted products.
__global__ void __launch_bounds__(64) natac_fft_template(const double *__restrict__ vmat, const double *__restrict__ srow, int R, int W,
                                                           const double *__restrict__ tw, double *__restrict__ kout) {
    __shared__ double2 sa[FFT_LA], sb[FFT_LB];
    const int lane = threadIdx.x, pair = blockIdx.x;
    const int ra = 2 * pair, rb = 2 * pair + 1;
    FftTwiddles t;
    fft_load_twiddles(t, tw, lane);
    double re[8], im[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int u = lane + 64 * j;
        re[j] = (u < W) ? srow[ra] * vmat[ra * W + u] : 0.0;
        im[j] = (u < W && rb < R) ? srow[rb] * vmat[rb * W + u] : 0.0;
    }
    fft512_fwd(re, im, t, sa, sb, lane);
    double *o = kout + (size_t)pair * 2 * FFT_N;
#pragma unroll
    for (int m = 0; m < 8; ++m) { o[NATAC_FFT_KIDX(m, lane, 0)] = re[m]; o[NATAC_FFT_KIDX(m, lane, 1)] = im[m]; }
}

// background + normalised signal for tiles of TV = 512 - W + 1 bases; one tile per wave, one wave per workgroup.
// Tiles whose bias window is not well conditioned for an FFT (non-finite or zero values, or a dynamic range > 3e4, where the
// FFT's error -- relative to the tile's largest product -- could show in the smallest outputs, and where a NaN must stay
// confined to the bases whose window touches it) are evaluated by direct summation in the same order as natac_background.
constexpr double FFT_MAX_RANGE = 3e4;   // real Tn5 PWM log-bias spans <= 8.7 log units genome-wide (e^8.7 = 6e3)

// one tile of the background through FFTs; `smem` = this wave's LDS (EWP + 2 FFT_LA doubles), `t` = (chunk, x0).
// TW_LOADED: the caller holds the per-lane twiddles in `tww` (persistent kernel); otherwise they are loaded here, after the
// conditioning test, exactly where the one-tile-per-workgroup kernel always loaded them.
template <bool TW_LOADED>
__device__ __forceinline__ void bg_fft_tile(const ChunkTable &ct, const int2 t, const VMatDev &vm, const double *__restrict__ tw,
                                            const double *__restrict__ ktab, const double *__restrict__ nuc_cov,
                                            const double *__restrict__ raw, double *__restrict__ bg, double *__restrict__ norm,
                                            double *__restrict__ bnum, double *__restrict__ bcov, double *smem, FftTwiddles &tww,
                                            const int lane, const double *__restrict__ mtab = nullptr, const double *__restrict__ swt = nullptr,
                                            const int NJ = 0) {
    const int W = vm.W, HW = W / 2, TV = FFT_N - W + 1;
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int ext = (t.y & FFT_EXT_BIT) ? FFT_EXT : 0;   // extended tile: FFT_EXT more outputs on each side, finished by the edge pass below
    const int EW = FFT_N + A + Bh + 2 * ext, EWP = (FFT_N + A + Bh + 2 * FFT_EXT + 1) & ~1;
    double *Et = smem;
    double2 *ca = (double2 *)(Et + EWP), *cb = ca;     // complex scratch of the transposes; layouts A and B are never live together
    double *sar = (double *)ca, *sai = sar + FFT_LA;   // the same memory as two real arrays (epilogue)
    const int chunk = t.x, x0 = t.y & (FFT_EXT_BIT - 1);
    const int L = ct.chunk_len[chunk];
    // the per-lane twiddles requested in front of the window staging (one round trip to L2 that nothing waits for); the rare tile that
    // falls back to direct summation has loaded them for nothing
    if (!TW_LOADED && NATAC_FFT_TW_EARLY) fft_load_twiddles(tww, tw, lane);
    bool use_fft;
    {   // Et[u] <-> coordinate x0 - ext - HW - A + u
        const double *b = ct.bias ? ct.ebias + ct.bias_off[chunk] : nullptr;       // exp(bias), natac_exp_bias
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = x0 - ext - HW - A + ct.bias_left;
        // only the first `need` entries feed bases of this chunk; the rest of the tile is padded with zeros
        const int need = bg_tile_need(TV, W, A, Bh, L, x0, ext);
        double emax = 0.0, emin = 1e300;
        bool okl = true;
        for (int u = lane; u < EW; u += WAVE) {
            const int j = j0 + u;
            double e = 0.0;
            if (u < need) {
                e = 1.0;
                if (b) e = (j >= 0 && j < nb) ? b[j] : 0.0;
                okl = okl && (e > 0.0) && (e < 1e300);   // false for NaN, inf, zero
                emax = fmax(emax, e); emin = fmin(emin, e);
            }
            Et[u] = e;
        }
        emax = wave_max(emax); emin = wave_min(emin);
        use_fft = (__ballot(!okl) == 0ull) && (emax <= emin * FFT_MAX_RANGE);
    }
    __builtin_amdgcn_wave_barrier();
    const long long ob = ct.out_off[chunk];
    Et += ext;                   // Et[u] <-> coordinate x0 - HW - A + u from here on, as for a plain tile
    if (!use_fft) {
#pragma unroll 1
        for (int j = 0; j < 9; ++j) {
            const int u = lane + 64 * j - ext, g = x0 + u;      // an extended tile's edge outputs too
            if (u >= TV + ext || g >= L) continue;
            double num = 0.0, cv = 0.0;
#pragma unroll 1
            for (int r = 0; r < vm.R; ++r) {
                const int i = vm.lower + r;
                const double s = vm.srow[r];
                const double *el = Et + (A - floor_half(i - 1)) + u, *er = Et + (A + floor_half(i)) + u;
                const double *vr = vm.mat + r * W;
#pragma unroll 1
                for (int c = 0; c < W; ++c) {
                    const double p = (s * el[c]) * er[c];
                    cv += p;
                    num = fma(p, vr[c], num);
                }
            }
            const long long o = ob + g;
            const double bgv = (num * nuc_cov[o]) / cv;
            if (bg) bg[o] = bgv;
            norm[o] = raw[o] - bgv;
            bnum[o] = num;
            bcov[o] = cv;
        }
        return;
    }
    if (!TW_LOADED && !NATAC_FFT_TW_EARLY) fft_load_twiddles(tww, tw, lane);
    double accr[8], acci[8], q[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { accr[m] = 0.0; acci[m] = 0.0; q[m] = 0.0; }
    const int npair = (vm.R + 1) >> 1;
    // exp(bias) operands of a row pair (a = 2 pair, b = a + 1; insert sizes ia = lower + a, ib = ia + 1).  Consecutive insert
    // sizes share a factor and consecutive pairs another one:
    //   ia odd  (= 2m+1):  P_a = s_a E[u-m]   E[u+m],  P_b = s_b E[u-m] E[u+m+1];  next pair: right factor of a = this pair's of b
    //   ia even (= 2m):    P_a = s_a E[u-m+1] E[u+m],  P_b = s_b E[u-m] E[u+m];    next pair: left factor of a = this pair's of b
    // so a pair needs 16 new LDS values per lane instead of 32 (the carried factor stays in registers), read with
    // lds_read8_b64 (plain ds_read_b64, see there).
    const bool lodd = (vm.lower & 1) != 0;
    const bool pairs_full = (vm.R & 1) == 0;
    double carry[8];
    if (pairs_full) {
        const int i0 = vm.lower;
        const double *c0 = lodd ? Et + (A + floor_half(i0)) : Et + (A - floor_half(i0 - 1));
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = c0[lane + 64 * j];
    }
    if (pairs_full) {
        // The skewed pair loop (round 6).  A transform is S1 -> W1 R1 -> S2 -> W2 R2 -> S3 (skew_* above); a wave that runs them in that
        // order sits out two LDS round trips per row pair (8 stores at 13 cycles, 8 loads, the queue of the CU's seven other waves in
        // front of them), and the SIMD's second wave covered only part of that (round 5: 23.5 ms of fp64 issue + 18.3 ms of LDS array time
        // overlapped by 4.5 ms).  Here the wave has TWO row pairs in flight, p (second half) and p + 1 (first half); one trip is
        //     wait R1(p) | X,Y(p+1) requested | S2(p) | W2(p) | wait X,Y | R2(p) requested | S1(p+1) | K(p) requested | W1(p+1) |
        //     wait R2(p) | R1(p+1) requested | S3(p)
        // so R2(p) travels under S1(p+1) (112 fp64 instructions) and R1(p+1) under S3(p) (84).  One scratch region is enough: a wave's LDS
        // instructions execute in issue order, R2(p) is issued before W1(p+1) overwrites what it reads.  Both waits inside the trip sit
        // where only the eight stores just issued are behind them: lgkmcnt is a 4-bit counter, a wait with 8 stores + 8 loads behind it can
        // name at most 15 and then sits out the first store's trip through the LDS queue (measured: that alone gave the gain back).
        // Registers: while S1(p+1) runs, R2's eight complex targets are the only extra live values (the template spectrum K(p) is requested
        // after it: a phase earlier the loop spills); while S3(p) runs, R1's.  244 VGPRs, no scratch.
        // Every value is computed by the same operations in the same order as in fft512_fwd: the outputs are bit-identical to round 5's
        // loop (tools/test_fft_bg.hip prints a hash of the four output arrays; profiles/r6/fft_skew_session*.txt).  Measured, same box,
        // 100 k extended tiles: 7.65 -> 7.12 ms; in the configs[2] step 40.4 -> 37.4 ms (profiles/r6/bench_ab_*).  The ablation builds
        // behind these numbers (stores / loads / template spectra / operands removed one by one) are commits 0c98ab4 and 1054b69.
        const unsigned cbase = (unsigned)(uintptr_t)ca;
        const int m2 = lane >> 3, n1 = lane & 7;
        const unsigned aw1 = cbase + 16u * lane, ar1 = cbase + 16u * (72 * m2 + n1), aw2 = cbase + 16u * (65 * n1 + 8 * m2), ar2 = aw1;
        // One form for an odd and for an even first insert size: with X = the factor both rows of a pair share and Y = row b's other factor
        // (odd: X = left factor, Y = right factor of b, carried = right factor of a; even: X = right factor, Y = left factor of b, carried =
        // left factor of a) the products are  re = X carried, im = X Y, carried' = Y  either way -- only the ADDRESSES differ (a product does
        // not depend on the order of its factors, so these are the bits of the loop that tested `lodd` per trip).  Left factors move down
        // by one sample per pair, right factors up.
        double x[8], y[8], tr[8], ti[8];
        d2v Av[8], Bv[8];
        const double *px = (lodd ? Et + (A - floor_half(vm.lower - 1)) : Et + (A + floor_half(vm.lower))) + lane;
        const double *py = (lodd ? Et + (A + floor_half(vm.lower + 1)) : Et + (A - floor_half(vm.lower))) + lane;
        const int dx = lodd ? -1 : 1;
        // yn = where the pair's Y goes, yo = the Y of the pair before (the carried factor).  The loop is unrolled by two and the two buffers swap
        // roles from trip to trip (as one buffer + `carry` every trip ended with eight register copies).
        auto issue_xy = [&](int pair, double (&yn)[8]) {
            lds_read8_b64(x, px + dx * pair);
            lds_read8_b64(yn, py - dx * pair);
        };
        auto s1 = [&](const double sa, const double sb, const double (&yn)[8], const double (&yo)[8]) {     // x, Y, carried Y -> products, column sums
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                tr[j] = x[j] * yo[j];
                ti[j] = x[j] * yn[j];
                q[j] = fma(sb, ti[j], fma(sa, tr[j], q[j]));
            }
        };
        auto s1b = [&]() {
            dft8<false>(tr, ti);
            skew_tw1(tr, ti, tww);
        };
        double kr[8], ki[8];
        auto load_k = [&](int pair) {
            const double *k = ktab + (size_t)pair * 2 * FFT_N;
#pragma unroll
            for (int m = 0; m < 8; ++m) { kr[m] = k[NATAC_FFT_KIDX(m, lane, 0)]; ki[m] = k[NATAC_FFT_KIDX(m, lane, 1)]; }
        };
        double ur[8], ui[8];
        // one trip: second half of `pair`, first half of pair + 1 (whose Y goes to yn; yo = pair's Y)
        auto trip = [&](int pair, double (&yn)[8], double (&yo)[8]) {
            lds_wait_c8<0>(Av);                        // R1(pair)
            issue_xy(pair + 1, yn);
            const double sa = vm.srow[2 * pair + 2], sb = vm.srow[2 * pair + 3];     // requested a phase ahead of their use in S1
            __builtin_amdgcn_sched_barrier(0);
            skew_s2(Av, ur, ui);
            __builtin_amdgcn_sched_barrier(0);
            skew_w2(ur, ui, tww, aw2);
            lds_wait16n<8>(x, yn);                    // the operands, with W2's eight stores outstanding but BEFORE R2 is requested: lgkmcnt is a 4-bit
            ds_ld128x8<1040>(Bv, ar2);                // counter, a wait behind R2 could not leave all 16 outstanding and would sit out the first store
            __builtin_amdgcn_sched_barrier(0);
            s1(sa, sb, yn, yo);
            s1b();
            __builtin_amdgcn_sched_barrier(0);
            load_k(pair);                             // K(pair), after S1: requested earlier it does not fit the register file (spills)
            skew_w1(tr, ti, aw1);
            lds_wait_c8<8>(Bv);                       // R2(pair): only W1's eight stores may still be outstanding; R1 is requested behind the wait
            ds_ld128x8<128>(Av, ar1);                  // R1(pair + 1)
            __builtin_amdgcn_sched_barrier(0);
            skew_s3(Bv, kr, ki, accr, acci);
        };
        issue_xy(0, y);
        lds_wait16n<0>(x, y);
        s1(vm.srow[0], vm.srow[1], y, carry);
        s1b();
        skew_w1(tr, ti, aw1);
        ds_ld128x8<128>(Av, ar1);
        {
            int pair = 0;
            for (; pair + 2 < npair; pair += 2) {
                trip(pair, carry, y);
                trip(pair + 1, y, carry);
            }
            if (pair + 1 < npair) trip(pair, carry, y);
        }
        lds_wait_c8<0>(Av);
        skew_s2(Av, ur, ui);
        skew_w2(ur, ui, tww, aw2);
        ds_ld128x8<1040>(Bv, ar2);
        load_k(npair - 1);
        lds_wait_c8<0>(Bv);
        skew_s3(Bv, kr, ki, accr, acci);
    } else   // odd row count: the plain loop, one transform after the other (the missing row reads row a's window with weight 0)
    for (int pair = 0; pair < npair; ++pair) {
        const int ra = 2 * pair, rb = ra + 1;
        const int ia = vm.lower + ra, ib = (rb < vm.R) ? ia + 1 : ia;
        const double sa = vm.srow[ra], sb = (rb < vm.R) ? vm.srow[rb] : 0.0;
        double re[8], im[8];
        const double *k = ktab + (size_t)pair * 2 * FFT_N;
        double kr[8], ki[8];
        const double *ela = Et + (A - floor_half(ia - 1)), *era = Et + (A + floor_half(ia));
        const double *elb = Et + (A - floor_half(ib - 1)), *erb = Et + (A + floor_half(ib));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = lane + 64 * j;
            re[j] = ela[u] * era[u];
            im[j] = elb[u] * erb[u];
            q[j] = fma(sb, im[j], fma(sa, re[j], q[j]));
        }
        fft512_fwd(re, im, tww, ca, cb, lane);
#pragma unroll
        for (int m = 0; m < 8; ++m) { kr[m] = k[NATAC_FFT_KIDX(m, lane, 0)]; ki[m] = k[NATAC_FFT_KIDX(m, lane, 1)]; }
#pragma unroll
        for (int m = 0; m < 8; ++m) {                 // acc += Z * conj(K)
            accr[m] = fma(re[m], kr[m], fma(im[m], ki[m], accr[m]));
            acci[m] = fma(im[m], kr[m], fma(-re[m], ki[m], acci[m]));
        }
    }
#if NATAC_FFT_EPI_PREFETCH
    // the epilogue's two inputs per output, requested before the inverse transform: behind the barriers and asm waits below the compiler
    // cannot move the loads up, and every one of the eight output rows then waited for its own round trip to HBM
    double pf_cov[8], pf_raw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int u = lane + 64 * j, g = x0 + u;
        const bool core = u < TV && g < L;
        pf_cov[j] = core ? nuc_cov[ob + g] : 0.0;
        pf_raw[j] = core ? raw[ob + g] : 0.0;
    }
#endif
    fft512_inv(accr, acci, tww, ca, cb, lane);
    // covB: W-wide box sum of Q (both rows of every pair already added), two levels: T[u] = sum of B consecutive Q,
    // cov[u] = sum of nb strided T + the remainder  (B = 11, nb = 11 for W = 121: 23 LDS reads per base instead of 121)
    int B = 1;
    while ((B + 1) * (B + 1) <= W) ++B;
    const int nbk = W / B;
#pragma unroll
    for (int j = 0; j < 8; ++j) sar[lane + 64 * j] = q[j];
    sar[FFT_N + lane] = 0.0;
    __builtin_amdgcn_wave_barrier();
    // both levels with the eight outputs of a lane inside the loop over the terms: eight independent reads per trip instead of one
    // (every output still adds its terms in the same order: the same bits as with the loops the other way round)
    {
        double ts[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ts[j] = 0.0;
        for (int c = 0; c < B; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ts[j] += sar[lane + 64 * j + c];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sai[lane + 64 * j] = ts[j];
    }
    sai[FFT_N + lane] = 0.0;      // an extended tile's right edge sums past the tile's last sample: the samples out there are the edge pass's
    __builtin_amdgcn_wave_barrier();
    double cvv[8];
    {
        int ub[8];                // outputs past the tile's last one read the last one's terms (in bounds, not used)
#pragma unroll
        for (int j = 0; j < 8; ++j) { ub[j] = min(lane + 64 * j, TV + ext - 1); cvv[j] = 0.0; }
        for (int k = 0; k < nbk; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cvv[j] += sai[ub[j] + B * k];
        }
        for (int c = B * nbk; c < W; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cvv[j] += sar[ub[j] + c];
        }
    }
    // left edge of an extended tile: output x0 - k sits at the circular index 512 - k (register 7 of the last FFT_EXT lanes); of its window
    // the tile holds samples 0 .. W - 1 - k: the window sum of output 0 (lane 0, register 0) less the last k samples of that window
    double cvl = 0.0;
    if (ext) {
        const double full = __shfl(cvv[0], 0);
        const int k = FFT_N - (lane + 64 * 7);
        double suf = 0.0;
#pragma unroll
        for (int c = 0; c < FFT_EXT; ++c) { const double v = sar[W - 1 - c]; suf += c < k ? v : 0.0; }
        cvl = full - suf;
    }
    double *pe = sar;                   // [side][output][numerator, window sum] of the edge outputs, as far as the tile holds them
    __builtin_amdgcn_wave_barrier();    // the sums above are done with sar / sai
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int u = lane + 64 * j;
        const int g = x0 + u;
        const bool core = u < TV && g < L;
        if (core || (u < TV + ext && g < L)) {
            const long long o = ob + g;
            const double num = accr[j] * (1.0 / FFT_N), cv = cvv[j];
            if (core) {
#if NATAC_FFT_EPI_PREFETCH
                const double b = (num * pf_cov[j]) / cv;
                if (bg) bg[o] = b;       // null: the track is formed from bnum / bcov on request (natac_bg_from_factors)
                norm[o] = pf_raw[j] - b;
#else
                const double b = (num * nuc_cov[o]) / cv;
                if (bg) bg[o] = b;
                norm[o] = raw[o] - b;
#endif
            }
            if (core) {
                bnum[o] = num;       // sum B V and sum B of the window at this base: reused by the candidate statistics
                bcov[o] = cv;
            } else {                 // right edge of an extended tile: both without the samples past the tile, finished below
                pe[2 * (FFT_EXT + u - TV)] = num; pe[2 * (FFT_EXT + u - TV) + 1] = cv;
            }
        } else if (j == 7 && u >= FFT_N - ext && x0 - (FFT_N - u) < L) {      // left edge
            pe[2 * (u - (FFT_N - FFT_EXT))] = accr[j] * (1.0 / FFT_N); pe[2 * (u - (FFT_N - FFT_EXT)) + 1] = cvl;
        }
    }
    if (ext) {
        // The edge pass (see bg_edge_side): the windows are ranges of the exp(bias) window this wave staged.  Entry e of that window
        // (coordinate x0 - 16 - HW - A + e; zero from `need` on and outside the chunk's bias slice):
        //   wl_lo[w] = E(w), wl_hi[w] = E(512 + w)                       left factors  (entry 16 + A - fh(i - 1) + u of sample u, row i)
        //   wr_lo[w] = E(A + hr0 + w), wr_hi[w] = E(512 + A + hr0 + w)   right factors (entry 16 + A + fh(i) + u)
        const double *Ea = Et - ext;
        const int hr0 = floor_half(vm.lower);
        double *sc = pe + 4 * FFT_EXT, *sws = sc + EDGE_SCRATCH;
        for (int i = lane; i < 4 * NJ; i += WAVE) sws[i] = swt[i];
        __builtin_amdgcn_wave_barrier();
        const int t = lane & (FFT_EXT - 1);
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            double corr, qc;
            bg_edge_side(side, Ea, Ea + FFT_N, Ea + A + hr0, Ea + FFT_N + A + hr0, sws, sc, mtab, NJ, vm.R, vm.lower, A, hr0, lane, corr, qc);
            const int g = side ? x0 + TV + t : x0 - FFT_EXT + t;
            if (lane < FFT_EXT && g >= 0 && g < L) {
                const long long o = ob + g;
                const double num = pe[2 * (FFT_EXT * side + t)] + corr, cv = pe[2 * (FFT_EXT * side + t) + 1] + qc;
                const double bb = (num * nuc_cov[o]) / cv;
                if (bg) bg[o] = bb;
                norm[o] = raw[o] - bb;
                bnum[o] = num;
                bcov[o] = cv;
            }
        }
    }
}

// background + normalised signal for tiles of TV = 512 - W + 1 bases; one tile per wave, one wave per workgroup.
// two waves per SIMD on purpose: a third one (reachable with the stage-2 twiddles in LDS, 145 VGPRs) only adds LDS contention
// (measured 11.4 vs 8.7 ms per 20 k chunks).
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) natac_background_fft(ChunkTable ct, const int2 *__restrict__ tiles, VMatDev vm,
                                                             const double *__restrict__ tw, const double *__restrict__ ktab,
                                                             const double *__restrict__ nuc_cov, const double *__restrict__ raw,
                                                             double *__restrict__ bg, double *__restrict__ norm,
                                                             double *__restrict__ bnum, double *__restrict__ bcov,
                                                             unsigned n_tiles, const double *__restrict__ mtab,
                                                             const double *__restrict__ swt, int NJ) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const unsigned ti = blockIdx.x;
    FftTwiddles tww;
    bg_fft_tile<false>(ct, tiles[ti], vm, tw, ktab, nuc_cov, raw, bg, norm, bnum, bcov, smem, tww, (int)threadIdx.x, mtab, swt, NJ);
}

// T_BACKGROUND on request (natac_api.hip: materialise_bg): the epilogue's own expression from the two factors it leaves per base
__global__ void __launch_bounds__(256) natac_bg_from_factors(const double *__restrict__ bnum, const double *__restrict__ nuc_cov,
                                                             const double *__restrict__ bcov, double *__restrict__ bg, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) bg[i] = (bnum[i] * nuc_cov[i]) / bcov[i];
}

// LDS of one wave of natac_background_fft, in bytes (host): the exp(bias) window of an extended tile + the transposes' scratch
__host__ __device__ inline size_t bg_fft_lds_bytes(int upper) {
    const int EWX = FFT_N + ((upper - 2) >> 1) + ((upper - 1) >> 1) + 2 * FFT_EXT;
    return ((size_t)((EWX + 1) & ~1) + 2 * FFT_LA) * sizeof(double);
}

// the edge pass's table (bg_edge_side):
// mtab[side][j][lane] = C of row 4 j + (lane >> 4), column index lane & 15 (0 for rows past R); swt[4 j + k] = s_r (0 past R).
__global__ void natac_fft_edge_table_mfma(const double *__restrict__ vmat, const double *__restrict__ srow, int R, int W, int NJ,
                                          double *__restrict__ mtab, double *__restrict__ swt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NJ * 4) swt[i] = i < R ? srow[i] : 0.0;
    if (i >= 2 * NJ * 64) return;
    const int side = i / (NJ * 64), j = (i / 64) % NJ, lane = i % 64;
    const int r = 4 * j + (lane >> 4), c = lane & 15;
    mtab[i] = r < R ? srow[r] * vmat[r * W + (side ? W - 1 - c : c)] : 0.0;
}
}  // namespace natac
