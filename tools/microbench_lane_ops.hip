// microbench_lane_ops.hip -- development: issue cost of cross-lane register moves against v_fma_f64 on gfx950 (one wave per SIMD and
// four waves per SIMD), to price a register-level 8x8 transpose against an LDS round trip.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_lane_ops.hip -o tools/mb_lane_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int REP = 64, ITER = 2000;

template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters) {
    double a[8];
    unsigned u[16];
    for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x + 256 * i];
    for (int i = 0; i < 16; ++i) u[i] = (unsigned)threadIdx.x * (i + 3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (OP == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = fma(a[i], 1.0000001, 1e-9);
            } else if (OP == 1) {      // v_mov_b32 dpp row_ror:8, masked to half the banks
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0x3" : "+v"(u[i]) : "v"(u[i + 8]));
            } else if (OP == 2) {      // v_permlane32_swap
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[i + 8]));
            } else if (OP == 3) {      // v_permlane16_swap
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[i + 8]));
            } else if (OP == 4) {      // plain v_mov_b32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[i + 8]));
            } else if (OP == 5) {      // v_add_f64
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = a[i] + 1e-9;
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    unsigned x = 0;
    for (int i = 0; i < 16; ++i) x ^= u[i];
    out[threadIdx.x] = s + x;
}

template <int OP>
void run(const char *name, double *d, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, ITER);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    // one block = 4 waves = one wave per SIMD of a CU (if blocks == #CUs); wave-instructions per SIMD = ITER * REP * (blocks / 256)
    const double inst = (double)ITER * REP * (blocks / 256.0);
    printf("%-22s blocks=%5d  %.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, blocks, best, best * 1e6 / inst);
}

int main() {
    double *d; CK(hipMalloc(&d, 256 * 8 * 8)); CK(hipMemset(d, 0, 256 * 8 * 8));
    for (int blocks : {256, 1024}) {
        run<0>("v_fma_f64", d, blocks);
        run<5>("v_add_f64", d, blocks);
        run<4>("v_mov_b32", d, blocks);
        run<1>("v_mov_b32_dpp ror:8", d, blocks);
        run<2>("v_permlane32_swap", d, blocks);
        run<3>("v_permlane16_swap", d, blocks);
    }
    return 0;
}
