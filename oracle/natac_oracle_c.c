/* C part of the CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/natac_oracle.py).
 *
 * oracle_calculate_cov restates calculateCov, nucleoatac/multinomial_cov.pyx:20-31:
 * the literal O(N^2) double loop, same visiting order (i outer, j from i), same
 * expression shapes, accumulator starting at 0 (the .pyx leaves it uninitialised at
 * :23; 0 is the value the reference's tests/test_var.py:34-43 pins), result * r with
 * r a C int (:20).
 */
#include <stddef.h>

double oracle_calculate_cov(const double *p, const double *v, long n, int r) {
    double value = 0.0;
    for (long i = 0; i < n; i++) {
        for (long j = i; j < n; j++) {
            if (i == j)
                value += p[i] * (1 - p[i]) * (v[i] * v[i]);
            else
                value += p[i] * p[j] * -2 * v[i] * v[j];
        }
    }
    return value * r;
}

/* dense 'valid' cross-correlation of a (R x ncol) matrix with a (R x W) template, first
 * (only) output row -- signal.correlate(sub, vmat, mode='valid')[0] as used at
 * nucleoatac/NucleosomeCalling.py:34-36 and :60-63, by direct summation. */
void oracle_correlate_valid(const double *sub, long ncol, const double *vmat, int R, int W, double *out) {
    long nout = ncol - W + 1;
    for (long g = 0; g < nout; g++) out[g] = 0.0;
    for (int r = 0; r < R; r++) {
        const double *row = sub + (size_t)r * ncol;
        const double *vr = vmat + (size_t)r * W;
        for (long g = 0; g < nout; g++) {
            double acc = 0.0;
            for (int c = 0; c < W; c++) acc += row[g + c] * vr[c];
            out[g] += acc;
        }
    }
}
