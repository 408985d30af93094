// probe_mfma64.hip -- issue-rate probe: v_mfma_f64_16x16x4_f64 alone, v_fma_f64 alone, and both pipes from co-resident waves
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double a0, double b0) {
    d4 acc[4];
    for (int t = 0; t < 4; ++t) acc[t] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_valu(double *out, int iters, double a0, double b0) {
    double acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = t;
    double a = a0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = fma(acc[t], a, b0);
    }
    double s = 0;
    for (int t = 0; t < 16; ++t) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// waves 0,1 of each block run MFMA, waves 2,3 run VALU FMAs (one of each per SIMD pair)
__global__ void __launch_bounds__(256) k_both(double *out, int iters, double a0, double b0) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave & 1) {
        d4 acc[4];
        for (int t = 0; t < 4; ++t) acc[t] = d4{0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    } else {
        double acc[16];
        for (int t = 0; t < 16; ++t) acc[t] = t;
        double a = a0 + threadIdx.x * 1e-9;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = fma(acc[t], a, b0);
        }
        for (int t = 0; t < 16; ++t) s += acc[t];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    double *d; CK(hipMalloc(&d, 4096 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 2048, iters = 20000;
    for (int which = 0; which < 3; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
            if (which == 1) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999, 1e-3);
            if (which == 2) hipLaunchKernelGGL(k_both, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999, 1e-3);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double waves = (double)blocks * 4;
        double flop;
        if (which == 0) flop = waves * iters * 4.0 * 2048;             // 16x16x4x2 per MFMA
        else if (which == 1) flop = waves * iters * 16.0 * 128;        // 64 lanes x 2 per FMA
        else flop = waves / 2 * iters * (4.0 * 2048 + 16.0 * 128);
        printf("%s: %.3f ms  %.2f TFLOP/s\n", which == 0 ? "mfma_f64_16x16x4 only" : which == 1 ? "v_fma_f64 only" : "half the waves each", best, flop / best / 1e9);
    }
    return 0;
}
