// natac_fuzzfit.hpp -- objective + finite-difference gradient of many Nucleosome.getFuzz fits in one call (host C++).
//
// nucleoatac/fuzzfit.py advances the L-BFGS-B runs of a task in lockstep and needs, per round, the objective
// (reference NucleosomeCalling.py:176-184: sum of squared differences between the window and up to three height-scaled Gaussians,
// `norm()` at :92-97) at n + 1 parameter sets per fit.  The numpy formulation costs ~40 array operations per round; this is the
// same arithmetic in one pass over the data.  "The same" is literal: scipy's optimiser takes different steps for inputs that
// differ in the last bit, and the fits are ill-conditioned, so every value must be the double numpy produces:
//   * +, -, *, /, sqrt are IEEE operations in numpy and here (the library is built with -ffp-contract=off);
//   * exp is NOT reimplemented: the caller passes numpy's own inner loop for float64 exp (the function pointer stored in the
//     np.exp ufunc object) and it runs on one contiguous buffer -- elementwise, independent of position and length;
//   * the row sums follow numpy's pairwise summation (8 accumulators up to 128 elements, halving above) over each row's own length;
//   * maxima are exact in any order.
// fuzzfit.available() compares this path with scipy.optimize.minimize + numpy on a fit before any use; tests compare every fit.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace natac_fuzzfit {

typedef void (*ufunc_loop)(char **args, const intptr_t *dimensions, const intptr_t *steps, void *data);

// numpy's DOUBLE_pairwise_sum for a contiguous array (numpy/_core/src/umath/loops_utils.h.src)
inline double pairwise_sum(const double *a, intptr_t n) {
    if (n < 8) {
        double res = 0.0;
        for (intptr_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        intptr_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    intptr_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}
// np.sum of a contiguous float64 array: the reduction starts from the identity 0.0
inline double np_sum(const double *a, intptr_t n) { return 0.0 + pairwise_sum(a, n); }

// K fits with n = 3 G parameters each (variance, weight, mean per Gaussian); X0, lb, ub: [K][n]; sig, xs: [K][M] (sig padded with
// 0, xs with a large abscissa); lens[k] = true window length; work: (n + G + n + 1) * K * M doubles.  f: [K], g: [K][n].
inline void evaluate(int K, int n, int M, const double *X0, const double *lb, const double *ub, const double *sig, const double *xs,
                     const int64_t *lens, ufunc_loop exp_loop, void *exp_data, double *work, double *f, double *g) {
    const int G = n / 3, R = n + G;
    const double FD_STEP = 1e-8;
    std::vector<double> xh((size_t)K * n), dx((size_t)K * n), scale((size_t)K * R);
    double *y = work;                                   // [K][R][M]
    double *fit = work + (size_t)K * R * M;             // [K][n + 1][M]
    // ---- finite-difference points (scipy.optimize._numdiff._adjust_scheme_to_bounds, 1-sided) and the exp arguments
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < n; ++i) {
            const double x0 = X0[(size_t)k * n + i], lo = lb[(size_t)k * n + i], hi = ub[(size_t)k * n + i];
            double h = FD_STEP;
            const double lower_dist = x0 - lo, upper_dist = hi - x0;
            const double x = x0 + h;
            const bool violated = (x < lo) || (x > hi);
            const bool fitting = std::fabs(h) <= std::fmax(lower_dist, upper_dist);
            if (violated && fitting) h *= -1.0;
            if (upper_dist >= lower_dist && !fitting) h = upper_dist;
            if (upper_dist < lower_dist && !fitting) h = -lower_dist;
            xh[(size_t)k * n + i] = x0 + h;
            dx[(size_t)k * n + i] = xh[(size_t)k * n + i] - x0;
        }
        for (int r = 0; r < R; ++r) {
            const int a = r < n ? r / 3 : r - n, c = r < n ? r % 3 : -1;
            const double *p0 = X0 + (size_t)k * n + 3 * a, *ph = xh.data() + (size_t)k * n + 3 * a;
            const double v = c == 0 ? ph[0] : p0[0], w = c == 1 ? ph[1] : p0[1], mean = c == 2 ? ph[2] : p0[2];
            const double two_v = 2.0 * v;
            double *yr = y + ((size_t)k * R + r) * M;
            const double *x = xs + (size_t)k * M;
            for (int i = 0; i < M; ++i) {
                double t = x[i] - mean;
                t = t * t;
                t = -t;
                yr[i] = t / two_v;
            }
            scale[(size_t)k * R + r] = w;               // the weight; divided by the row maximum below
            // 1.0 / np.sqrt(2 * np.pi * v): kept in fit's first element slot of this row? no -- recomputed after exp (cheap)
        }
    }
    // ---- numpy's exp on the whole buffer, in place
    {
        char *args[2] = {(char *)y, (char *)y};
        const intptr_t dims[1] = {(intptr_t)K * R * M};
        const intptr_t steps[2] = {8, 8};
        exp_loop(args, dims, steps, exp_data);
    }
    const double two_pi = 2 * 3.141592653589793;        // python's `2 * np.pi`
    for (int k = 0; k < K; ++k) {
        const int64_t m = lens[k];
        for (int r = 0; r < R; ++r) {
            const int a = r < n ? r / 3 : r - n, c = r < n ? r % 3 : -1;
            const double v = c == 0 ? xh[(size_t)k * n + 3 * a] : X0[(size_t)k * n + 3 * a];
            const double cst = 1.0 / std::sqrt(two_pi * v);
            double *yr = y + ((size_t)k * R + r) * M;
            double mx = 0.0;
            bool first = true;
            for (int i = 0; i < M; ++i) {
                const double t = cst * yr[i];
                yr[i] = t;
                if (first) { mx = t; first = false; }
                else if (!(mx >= t) && !(mx != mx)) mx = t;       // np.maximum semantics: NaN propagates
            }
            const double s = scale[(size_t)k * R + r] / mx;
            for (int i = 0; i < M; ++i) yr[i] = yr[i] * s;
        }
        // ---- fit of parameter set i = ((0 + y_0) + y_1) + y_2 with its own version of every Gaussian; squared residuals; sums
        double s_base = 0.0;
        double sums[10];
        for (int i = n; i >= 0; --i) {                  // the point itself (i == n) first: f0 is needed for the differences
            double *fr = fit + ((size_t)k * (n + 1) + i) * M;
            for (int q = 0; q < M; ++q) fr[q] = 0.0;
            for (int j = 0; j < G; ++j) {
                const double *src = y + ((size_t)k * R + ((i < n && i / 3 == j) ? i : n + j)) * M;
                for (int q = 0; q < M; ++q) fr[q] = fr[q] + src[q];
            }
            const double *sg = sig + (size_t)k * M;
            for (int q = 0; q < M; ++q) { const double d = fr[q] - sg[q]; fr[q] = d * d; }
            sums[i] = np_sum(fr, (intptr_t)m);
            if (i == n) s_base = sums[i];
        }
        f[k] = s_base;
        for (int i = 0; i < n; ++i) g[(size_t)k * n + i] = (sums[i] - s_base) / dx[(size_t)k * n + i];
    }
}

}  // namespace natac_fuzzfit
