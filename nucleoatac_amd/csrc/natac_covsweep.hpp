// natac_covsweep.hpp -- calculateCov (nucleoatac/multinomial_cov.pyx:20-31) for MANY candidate windows at once, in three
// arithmetic variants: the literal O(N^2) pair sum of the .pyx, the closed form r (sum p v^2 - (sum p v)^2) in fp64 and the
// same closed form in fp32.  BASELINE.json configs[4] / SURVEY.md section 8(d) cfg 5: "fp64 multinomial_cov path, tolerance
// sweep" at the candidates of a sample.  The probability vector of a candidate is the one SignalDistribution builds
// (NucleosomeCalling.py:70-76): p = B window / sum(B window), B = sizes[i] exp(b[x-(i-1)//2]) exp(b[x+i//2]).
#pragma once
#include "natac_kernels.hpp"

namespace natac {

// p[k][r*W + c] for candidate k: one workgroup per candidate.  LDS: exp(bias) window (EW doubles).
__global__ void __launch_bounds__(256) natac_cand_window_probs(ChunkTable ct, VMatDev vm, const int *__restrict__ cand_chunk,
                                                                 const int *__restrict__ cand_pos, double *__restrict__ p_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4];
    const int k = blockIdx.x;
    const int chunk = cand_chunk[k], p = cand_pos[k];
    const int L = ct.chunk_len[chunk];
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int EW = vm.W + A + Bh;
    double *Et = smem;   // Et[u] <-> coordinate p - w - A + u
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
    const int N = vm.R * vm.W;
    double *dst = p_out + (size_t)k * N;
    double sB = 0;
    for (int idx = threadIdx.x; idx < N; idx += 256) {
        const int r = idx / vm.W, c = idx - r * vm.W;
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const double b0 = (hl == -hr) ? Et[c + A] : Et[c + A - hl] * Et[c + A + hr];
        const double bb = vm.srow[r] * b0;
        dst[idx] = bb;
        sB += bb;
    }
    sB = wave_sum(sB);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sB;
    __syncthreads();
    sB = (red[0] + red[1]) + (red[2] + red[3]);
    for (int idx = threadIdx.x; idx < N; idx += 256) dst[idx] = dst[idx] / sB;
}

// literal pair sum for candidate blockIdx.y: workgroup (x, y) owns rows i = x, x + gridDim.x, ... of candidate y, the 256
// threads stride over j >= i with the .pyx's own term expressions (same code as natac_cov_literal).
__global__ void __launch_bounds__(256) natac_cov_literal_many(const double *__restrict__ p_all, const double *__restrict__ v,
                                                                int n, double *__restrict__ partial) {
    __shared__ double red[4];
    const double *p = p_all + (size_t)blockIdx.y * n;
    double acc = 0.0;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double pi = p[i], vi = v[i];
        for (int j = i + threadIdx.x; j < n; j += 256) {
            if (j == i) acc += pi * (1 - pi) * (vi * vi);
            else acc += pi * p[j] * -2 * vi * v[j];
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// closed form, one workgroup per candidate: out[k] = S1 - S2^2 with S1 = sum p v^2, S2 = sum p v.  T = double or float
// (the fp32 variant converts p and v to float first and keeps every sum in float).
template <class T>
__global__ void __launch_bounds__(256) natac_cov_closed_many(const double *__restrict__ p_all, const double *__restrict__ v, int n,
                                                               double *__restrict__ out) {
    __shared__ double red[2][4];
    const double *p = p_all + (size_t)blockIdx.x * n;
    T s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const T pi = (T)p[i], vi = (T)v[i];
        const T pv = pi * vi;
        s1 += pv * vi;
        s2 += pv;
    }
    // wave / block reduction in the variant's own precision
    if constexpr (std::is_same<T, float>::value) {
        float a = s1, b = s2;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float S1 = ((float)red[0][0] + (float)red[0][1]) + ((float)red[0][2] + (float)red[0][3]);
            const float S2 = ((float)red[1][0] + (float)red[1][1]) + ((float)red[1][2] + (float)red[1][3]);
            out[blockIdx.x] = (double)(S1 - S2 * S2);
        }
    } else {
        double a = wave_sum(s1), b = wave_sum(s2);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double S1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            const double S2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
            out[blockIdx.x] = S1 - S2 * S2;
        }
    }
}

// out[k] = src[idx[k]]
__global__ void __launch_bounds__(256) natac_gather_f64(const double *__restrict__ src, const long long *__restrict__ idx, long long n,
                                                         double *__restrict__ out) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = src[idx[k]];
}

}  // namespace natac
