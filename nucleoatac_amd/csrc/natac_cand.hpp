// natac_cand.hpp -- per-candidate statistics (Nucleosome.getLR / getZScore, NucleosomeCalling.py:110-127), paired-row form.
//
// Same quantities as natac_candidates4 (natac_kernels.hpp): for candidate position p and the V-plot window around it
//   S_B0V = sum_{r,c} V[r,c] B0[r,x_c],   S_BV2 = sum_{r,c} s_r V[r,c]^2 B0[r,x_c],   B0[i,x] = E[x-(i-1)//2] E[x+i//2]
// (S_B and S_BV are the per-base values the background kernel left at p), then lr from the window's fragments, var, z.
// Consecutive insert sizes share one exp(bias) factor and consecutive PAIRS of sizes the other one:
//   i = 2m+1, 2m+2:   B0 = L_m R_m,  L_m R_{m+1}        (L_m[x] = E[x-m], R_m[x] = E[x+m])
//   i = 2m,   2m+1:   B0 = L_{m-1} R_m,  L_m R_m
// so a row pair costs  shared * (v_a * other_a + v_b * other_b)  = 3 flops per output and column instead of 4, and 2 LDS
// reads per column instead of 4 (the carried factor stays in registers).  The per-fragment likelihood terms of the four
// candidates of a wave are evaluated together, one 16-lane row per candidate (row-wide 16-ary window search, two
// fragments per lane, DPP row sums).  V-plots or size distributions with exact zeros, and windows whose exp(bias) values
// could make a product underflow, take the exact per-cell zero test of natac_candidates4 (launched instead by the host / by
// the wave-uniform fallback below).
#pragma once
#include "natac_kernels.hpp"

namespace natac {

// first index in a[lo, hi) with a[idx] >= key, searched by one 16-lane row (every lane of the row returns it); the four
// rows of a wave search four different ranges at once.  l = lane & 15, sh = 16 * (lane >> 4).
__device__ __forceinline__ int row_lower_bound(const int *__restrict__ a, int lo, int hi, int key, int l, int sh) {
    // all rows iterate until every row is done (ballots are wave-wide)
    while (__ballot(hi - lo > 16) != 0ull) {
        const int n = hi - lo;
        const bool big = n > 16;
        const int stride = big ? (n + 15) / 16 : 1;
        const int pi = min((l + 1) * stride - 1, max(n - 1, 0));
        const bool less = big && a[lo + pi] < key;
        const int cnt = __popc((unsigned)(__ballot(less) >> sh) & 0xffffu);
        if (big) {
            lo = min(lo + cnt * stride, hi);
            hi = min(lo + stride, hi);
        }
    }
    const int i = lo + l;
    const bool less = (i < hi) && a[i] < key;
    return lo + __popc((unsigned)(__ballot(less) >> sh) & 0xffffu);
}

// Nucleosome.getLR (NucleosomeCalling.py:110-122) over the fragments f of a candidate's window:
//     lr = sum_f log(V[r_f,c_f] B0_f / S_B0V) - sum_f log(s_{r_f} B0_f / S_B)  =  sum_f log(V[r_f,c_f] / s_{r_f})  -  n_f log(S_B0V / S_B):
// the fragment's bias product B0_f is in both terms and drops out, what is left per fragment is a constant of the model.  This table
// holds it (R x W doubles, formed once per V-plot / size distribution), so a fragment costs one lookup instead of two exp(bias) reads, two
// divisions and two logarithms.  Models with exact zeros never get here (natac_candidates4 keeps the per-cell form); windows with an
// exp(bias) of 0 or NaN give NaN as before (the zero-cell test below / S_B0V itself).
__global__ void __launch_bounds__(256) natac_lr_table(const double *__restrict__ vmat, const double *__restrict__ srow, int R, int W,
                                                        double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < R * W) out[i] = log(vmat[i] / srow[i / W]);
}

constexpr int CANDP_STRIDE = 376;   // doubles per candidate window in LDS (compile-time: every LDS address of the sweep is a
                                    // running base register + an immediate); V-plots with W + upper - 2 > 376 use natac_candidates4

// LODD: parity of vm.lower.  Requires vm.lower >= 2, R even, W >= 64, EW <= CANDP_STRIDE, bnum / bcov of the current model
// (host-checked).
template <bool LODD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) natac_candidates_paired(
    ChunkTable ct, VMatDev vm, const int *__restrict__ cand_chunk, const int *__restrict__ cand_pos, int ncand,
    const double *__restrict__ nuc_cov, const double *__restrict__ norm, const double *__restrict__ bnum,
    const double *__restrict__ bcov, const long long *__restrict__ tile_first, const int2 *__restrict__ ranges256,
    double *__restrict__ out_lr, double *__restrict__ out_var, double *__restrict__ out_z) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int Q = CAND_PER_WAVE;
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int W = vm.W, R = vm.R;
    const int EW = W + A + Bh;
    constexpr int EWP = CANDP_STRIDE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *Ew = smem + wave * Q * EWP;                       // Ew[q * EWP + u] <-> coordinate p_q - w - A + u
    const int k0 = (blockIdx.x * 4 + wave) * Q;
    if (k0 >= ncand) return;                                   // wave-uniform; no block-level synchronisation below
    int chunk[Q], pos[Q];
    bool esmall = false;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
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
            if (b) e = (j >= 0 && j < nb) ? b[j] : 0.0;
            Ew[q * EWP + u] = e;
            emin = fmin(emin, e);
        }
        esmall |= !(wave_min(emin) > 0x1p-500);                // no product of two window values can underflow to 0 above this
    }
    __builtin_amdgcn_wave_barrier();
    // ---- sweep: lanes own template columns c1 = lane, c2 = lane + 64 (idle second columns: clamped address, zero template)
    const int c1 = lane, c2 = min(lane + WAVE, W - 1);
    const bool h2 = lane + WAVE < W;
    double a1[Q][2], a2[Q][2];                                 // S_B0V / S_BV2 partial sums per candidate and column
    double carry[Q][2];
#pragma unroll
    for (int q = 0; q < Q; ++q) { a1[q][0] = a1[q][1] = a2[q][0] = a2[q][1] = 0.0; }
    // pair k: rows 2k, 2k+1, insert sizes i = lower + 2k, i + 1
    const int m0 = LODD ? (vm.lower - 1) >> 1 : vm.lower >> 1;   // m of the first pair
    {   // carried factor of the first pair: LODD: R_m0 (index c + A + m0);  else: L_{m0-1} (index c + A - m0 + 1)
        const int off = LODD ? A + m0 : A - m0 + 1;
#pragma unroll
        for (int q = 0; q < Q; ++q) { carry[q][0] = Ew[q * EWP + c1 + off]; carry[q][1] = Ew[q * EWP + c2 + off]; }
    }
    const double *__restrict__ vmat = vm.mat;
    const int npair = R >> 1;
    // running LDS pointers of the two columns: shared factor (moves by -1 / +1 per pair) and new other factor (+1 / -1)
    const double *psh1 = Ew + c1 + (LODD ? A - m0 : A + m0), *psh2 = Ew + c2 + (LODD ? A - m0 : A + m0);
    const double *pnw1 = Ew + c1 + (LODD ? A + m0 + 1 : A - m0), *pnw2 = Ew + c2 + (LODD ? A + m0 + 1 : A - m0);
    constexpr int DSH = LODD ? -1 : 1, DNW = LODD ? 1 : -1;
    // template values and size weights of pair k (rows 2k, 2k+1), requested one trip ahead of their use: the L2 round trip of
    // these loads is as long as the arithmetic of a trip and three waves per SIMD do not hide it
    struct PairT { double va0, va1, vb0, vb1, sa, sb; };
    const double hm = h2 ? 1.0 : 0.0;                          // idle second column: clamped (valid) address, zero template
    const unsigned o1 = (unsigned)c1 * 8u, o2 = (unsigned)c2 * 8u;      // byte offsets of the lane's two template columns
    auto load_pair = [&](int k) {
        PairT t;
        t.sa = vm.srow[2 * k]; t.sb = vm.srow[2 * k + 1];
        // uniform row base + the lane's byte offset (forcing the scalar-base form of the load -- an opaque 32-bit offset next to it --
        // costs more in moves than the 64-bit address adds it saves: measured 2.25 against 2.19 ms per 20 k chunks)
        const char *ra = (const char *)(vmat + (size_t)(2 * k) * (size_t)W), *rb = ra + (size_t)W * sizeof(double);
        t.va0 = *(const double *)(ra + o1); t.va1 = *(const double *)(ra + o2);
        t.vb0 = *(const double *)(rb + o1); t.vb1 = *(const double *)(rb + o2);
        return t;
    };
    // one pair: row a multiplies the carried factor `cin`, row b the newly read one, which is handed on in `cout`
    auto pair_step = [&](const PairT &t, const double (&cin)[Q][2], double (&cout)[Q][2]) {
        const double va0 = t.va0, va1 = hm * t.va1, vb0 = t.vb0, vb1 = hm * t.vb1;
        const double wa0 = t.sa * (va0 * va0), wa1 = t.sa * (va1 * va1);     // weights of S_BV2: s_r v^2
        const double wb0 = t.sb * (vb0 * vb0), wb1 = t.sb * (vb1 * vb1);
        double s0[Q], s1[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {                          // all 16 LDS operands of the step are requested before the arithmetic
            s0[q] = psh1[q * EWP]; s1[q] = psh2[q * EWP];
            cout[q][0] = pnw1[q * EWP]; cout[q][1] = pnw2[q * EWP];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const double n0 = cout[q][0], n1 = cout[q][1];
            a1[q][0] = fma(s0[q], fma(vb0, n0, va0 * cin[q][0]), a1[q][0]);
            a1[q][1] = fma(s1[q], fma(vb1, n1, va1 * cin[q][1]), a1[q][1]);
            a2[q][0] = fma(s0[q], fma(wb0, n0, wa0 * cin[q][0]), a2[q][0]);
            a2[q][1] = fma(s1[q], fma(wb1, n1, wa1 * cin[q][1]), a2[q][1]);
        }
        psh1 += DSH; psh2 += DSH; pnw1 += DNW; pnw2 += DNW;
    };
    double carry2[Q][2];
    int k = 0;
    PairT t0 = load_pair(0), t1 = load_pair(min(1, npair - 1));
    // four pairs per trip: two sets of template registers take turns and are loaded directly -- with two pairs per trip the prefetched
    // values were copied into place, 8 v_mov_b64 per trip (9.5 -> 9.0 ms per configs[2] step)
    for (; k + 4 <= npair; k += 4) {
        const PairT u0 = load_pair(min(k + 2, npair - 1));
        pair_step(t0, carry, carry2);
        const PairT u1 = load_pair(min(k + 3, npair - 1));
        pair_step(t1, carry2, carry);
        t0 = load_pair(min(k + 4, npair - 1));
        pair_step(u0, carry, carry2);
        t1 = load_pair(min(k + 5, npair - 1));
        pair_step(u1, carry2, carry);
    }
    for (; k + 2 <= npair; k += 2) {                          // two pairs per trip: the carried factor ping-pongs, no copies
        const PairT u0 = load_pair(min(k + 2, npair - 1));
        pair_step(t0, carry, carry2);
        t0 = u0;
        const PairT u1 = load_pair(min(k + 3, npair - 1));
        pair_step(t1, carry2, carry);
        t1 = u1;
    }
    if (k < npair) pair_step(t0, carry, carry2);
    // ---- exact zero-cell test, only for windows whose exp(bias) values are small enough for a product to underflow
    // (the host launches natac_candidates4 instead when the template / size distribution hold exact zeros)
    bool zero[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) zero[q] = false;
    if (esmall) {                                              // wave-uniform, rare
        for (int r = 0; r < R; ++r) {
            const int i = vm.lower + r;
            const int hl = floor_half(i - 1), hr = floor_half(i);
            const double sr = vm.srow[r];
            const double v1 = vmat[r * W + c1], v2 = vmat[r * W + c2];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const double *e = Ew + q * EWP;
                const double b1 = e[c1 + A - hl] * e[c1 + A + hr], b2 = e[c2 + A - hl] * e[c2 + A + hr];
                zero[q] |= (v1 * b1 == 0.0 || sr * b1 == 0.0) || (h2 && (v2 * b2 == 0.0 || sr * b2 == 0.0));
            }
        }
    }
    double tBV2[Q], tB0V[Q];
    bool anyzero[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        tB0V[q] = wave_sum(a1[q][0] + a1[q][1]);
        tBV2[q] = wave_sum(a2[q][0] + a2[q][1]);
        anyzero[q] = __ballot(zero[q]) != 0ull;
    }
    // ---- likelihood terms: row `row` of the wave handles candidate `row`
    const int row = lane >> 4, l = lane & 15, sh = 16 * row;
    int mych = chunk[0], myp = pos[0];
    double myB0V = tB0V[0], myBV2 = tBV2[0];
    bool myzero = anyzero[0];
#pragma unroll
    for (int q = 1; q < Q; ++q)
        if (row == q) { mych = chunk[q]; myp = pos[q]; myB0V = tB0V[q]; myBV2 = tBV2[q]; myzero = anyzero[q]; }
    const long long op = ct.out_off[mych] + myp;
    const double tB = bcov[op], tBV = bnum[op];
    const int nfr = (int)(ct.frag_off[mych + 1] - ct.frag_off[mych]);
    const int *cen = ct.centre + ct.frag_off[mych];
    const int *iln = ct.ilen + ct.frag_off[mych];
    int f0, f1;
    if (ranges256) {
        // the gather's index over the fragment list (natac_tile_ranges256): the fragments whose centres lie within w of the candidate's
        // 256-base tile.  The row counts the centres below the window's two ends among them -- one round of independent loads; the
        // 16-ary searches over the whole chunk are three dependent rounds each (and were most of this kernel's tail)
        const int2 tr = ranges256[tile_first[mych] + (myp >> 8)];
        const int klo = myp - vm.w, khi = myp + vm.w + 1;
        int below_lo = 0, below_hi = 0;
        int span = tr.y - tr.x;                                 // rows iterate together
        span = max(max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 16)),
                   max(__builtin_amdgcn_readlane(span, 32), __builtin_amdgcn_readlane(span, 48)));
        for (int base = 0; base < span; base += 64) {
            int cv[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int i = tr.x + base + 16 * it + l;
                cv[it] = 0x7fffffff;
                if (i < tr.y) cv[it] = cen[i];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                below_lo += __popc((unsigned)(__ballot(cv[it] < klo) >> sh) & 0xffffu);
                below_hi += __popc((unsigned)(__ballot(cv[it] < khi) >> sh) & 0xffffu);
            }
        }
        f0 = tr.x + below_lo;
        f1 = tr.x + below_hi;
    } else {
        f0 = row_lower_bound(cen, 0, nfr, myp - vm.w, l, sh);
        f1 = row_lower_bound(cen, f0, nfr, myp + vm.w + 1, l, sh);
    }
    // sum_f log(V / s) over the window's nucleosome-sized fragments (see natac_lr_table) and their number.  The sizes / centres of up to
    // 64 fragments of the row's window (4 per lane) are requested together, then their table values: two memory round trips for the
    // whole window instead of two per 16 fragments
    const double *__restrict__ lrt = vm.lrt;
    double acc = 0.0, cntf = 0.0;
    int fmax = f1 - f0;                                         // rows iterate together: longest window of the wave
    fmax = max(max(__builtin_amdgcn_readlane(fmax, 0), __builtin_amdgcn_readlane(fmax, 16)),
               max(__builtin_amdgcn_readlane(fmax, 32), __builtin_amdgcn_readlane(fmax, 48)));
    {
        int nn[4], cc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int f = f0 + 16 * it + l;
            nn[it] = -1; cc[it] = 0;
            if (f < f1) { nn[it] = iln[f]; cc[it] = cen[f]; }
        }
        double tv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            tv[it] = 0.0;
            if (nn[it] >= vm.lower && nn[it] < vm.upper) tv[it] = lrt[(nn[it] - vm.lower) * W + (cc[it] - myp + vm.w)];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (nn[it] >= vm.lower && nn[it] < vm.upper) { acc += tv[it]; cntf += 1.0; }
    }
    for (int base = 64; base < fmax; base += 16) {             // windows with more than 64 fragments
        const int f = f0 + base + l;
        if (f < f1) {
            const int n = iln[f];
            if (n >= vm.lower && n < vm.upper) { acc += lrt[(n - vm.lower) * W + (cen[f] - myp + vm.w)]; cntf += 1.0; }
        }
    }
    acc = row_sum(acc);
    cntf = row_sum(cntf);                                       // small integers: exact
    // LR = sum_f log(V / s)[r_f, c_f] - n_f log(S_B0V / S_B): Nucleosome.getLR's sum of log(V b0 / (b ...)) with the per-fragment bias
    // product cancelled.  Algebraically equal; it DEVIATES from the reference only where a single fragment's V * b0 under- or overflows
    // in the reference (it returns -inf / NaN there, this form stays finite; `myzero` keeps the reference's NaN for exact zeros), and
    // it rounds differently at ~1e-13 relative (tests hold lr to the oracle at 1e-10).
    const double nl = acc, ul = cntf * log(myB0V / tB);
    if (l == 0 && k0 + row < ncand) {
        const double m1 = tBV / tB;
        const int reads = (int)nuc_cov[op];
        const double var = (double)reads * (myBV2 / tB - m1 * m1);
        out_lr[k0 + row] = myzero ? __builtin_nan("") : (nl - ul);
        out_var[k0 + row] = var;
        out_z[k0 + row] = norm[op] / sqrt(var);
    }
}

}  // namespace natac
