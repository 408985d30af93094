// microbench_bg.hip -- standalone timing of background-kernel variants (development tool, not shipped in the .so)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/microbench_bg.hip -o /tmp/mb_bg
#include "../nucleoatac_amd/csrc/natac_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace natac;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- variant FMA-only ceiling: no V loads (V operand = constant derived from r), same LDS P reads
template <int G, int W, int MODE>
__global__ void __launch_bounds__(64) bg_variant(ChunkTable ct, const int2 *__restrict__ tiles, VMatDev vm,
                                                   const double *__restrict__ nuc_cov, const double *__restrict__ raw,
                                                   double *__restrict__ bg, double *__restrict__ norm) {
    constexpr int TW = WAVE * G, HW = W / 2, PW = TW + W - 1, NQ = (PW + WAVE - 1) / WAVE;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int EW = PW + A + Bh;
    double *Et = smem;
    double *Pb = smem + ((EW + 1) & ~1);
    double *Vl = Pb + ((PW + 1) & ~1);   // [128]
    {
        const double *b = ct.bias + ct.bias_off[chunk];
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = x0 - HW - A + ct.bias_left;
        for (int u = lane; u < EW; u += WAVE) { const int j = j0 + u; Et[u] = (j >= 0 && j < nb) ? exp(b[j]) : 0.0; }
    }
    __syncthreads();
    double acc[G]; double q[NQ];
#pragma unroll
    for (int k = 0; k < G; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = 0.0;
    const int ub = lane * G;
    double vn0 = vm.mat[lane], vn1 = (lane + 64 < W) ? vm.mat[lane + 64] : 0.0;
    for (int r = 0; r < vm.R; ++r) {
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const double s = vm.srow[r];
        const double *el = Et + (A - hl); const double *er = Et + (A + hr);
        if (MODE == 1) { Vl[lane] = vn0; Vl[lane + 64] = vn1; }
#pragma unroll
        for (int k = 0; k < NQ; ++k) { const int u = lane + WAVE * k; if (u < PW) { const double p = (s * el[u]) * er[u]; q[k] += p; Pb[u] = p; } }
        if (MODE == 1 && r + 1 < vm.R) { vn0 = vm.mat[(r + 1) * W + lane]; vn1 = (lane + 64 < W) ? vm.mat[(r + 1) * W + lane + 64] : 0.0; }
        if (MODE == 4) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        const double *pl = Pb + ub;
        if (MODE == 0 || MODE == 4) {
            const double *__restrict__ vr = vm.mat + r * W;
#pragma unroll
            for (int j = 0; j < G + W - 1; ++j) { const double p = pl[j];
#pragma unroll
                for (int k = 0; k < G; ++k) { const int c = j - k; if (c >= 0 && c < W) acc[k] = fma(p, vr[c], acc[k]); } }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < G + W - 1; ++j) { const double p = pl[j];
#pragma unroll
                for (int k = 0; k < G; ++k) { const int c = j - k; if (c >= 0 && c < W) acc[k] = fma(p, Vl[c], acc[k]); } }
        } else {  // MODE 2: no V loads at all (ceiling)
            const double vconst = s;
#pragma unroll
            for (int j = 0; j < G + W - 1; ++j) { const double p = pl[j];
#pragma unroll
                for (int k = 0; k < G; ++k) { const int c = j - k; if (c >= 0 && c < W) acc[k] = fma(p, vconst, acc[k]); } }
        }
        if (MODE == 4) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    }
    const long long ob = ct.out_off[chunk];
#pragma unroll
    for (int k = 0; k < G; ++k) { const int g = x0 + ub + k; if (g < L) { bg[ob + g] = acc[k] + q[k % NQ]; norm[ob + g] = acc[k]; } }
}

template <int G, int MODE>
float run(const ChunkTable &ct, const VMatDev &vm, int nc, int L, double *d_a, double *d_b, double *d_o1, double *d_o2, int reps) {
    const int TW = 64 * G;
    std::vector<int2> tiles;
    for (int i = 0; i < nc; ++i) for (int x = 0; x < L; x += TW) tiles.push_back(make_int2(i, x));
    int2 *d_t; CK(hipMalloc(&d_t, tiles.size() * sizeof(int2)));
    CK(hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    const int PW = TW + 120, EW = PW + 249;
    size_t lds = ((size_t)((EW + 1) & ~1) + ((PW + 1) & ~1) + 128) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((bg_variant<G, 121, MODE>), dim3(tiles.size()), dim3(64), lds, 0, ct, d_t, vm, d_a, d_b, d_o1, d_o2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    double flop = 2.0 * 146 * 121 * (double)nc * L;
    printf("G=%2d MODE=%d tiles=%zu lds=%zu  %.3f ms  %.2f TFLOP/s (useful)  %.1f Mbp/s\n", G, MODE, tiles.size(), lds, best, flop / best / 1e9, (double)nc * L / best / 1e3);
    CK(hipFree(d_t));
    return best;
}

// ---- variant 3: product row of r+1 built while row r is swept (double buffer, single-wave workgroup => no s_barrier)
template <int G, int W>
__global__ void __launch_bounds__(64) bg_pipelined(ChunkTable ct, const int2 *__restrict__ tiles, VMatDev vm,
                                                     const double *__restrict__ nuc_cov, const double *__restrict__ raw,
                                                     double *__restrict__ bg, double *__restrict__ norm) {
    constexpr int TW = WAVE * G, HW = W / 2, PW = TW + W - 1, NQ = (PW + WAVE - 1) / WAVE, PWP = (PW + 1) & ~1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int EW = PW + A + Bh;
    double *Et = smem;
    double *P0 = smem + ((EW + 1) & ~1);
    double *P1 = P0 + PWP;
    {
        const double *b = ct.bias + ct.bias_off[chunk];
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = x0 - HW - A + ct.bias_left;
        for (int u = lane; u < EW; u += WAVE) { const int j = j0 + u; Et[u] = (j >= 0 && j < nb) ? exp(b[j]) : 0.0; }
    }
    __syncthreads();
    double acc[G]; double q[NQ];
#pragma unroll
    for (int k = 0; k < G; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = 0.0;
    const int ub = lane * G;
    auto build = [&](int r, double *__restrict__ Pb) {
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const double s = vm.srow[r];
        const double *el = Et + (A - hl); const double *er = Et + (A + hr);
#pragma unroll
        for (int k = 0; k < NQ; ++k) { const int u = lane + WAVE * k; if (u < PW) { const double p = (s * el[u]) * er[u]; q[k] += p; Pb[u] = p; } }
    };
    auto sweep = [&](int r, const double *__restrict__ Pb) {
        const double *__restrict__ vr = vm.mat + r * W;
        const double *pl = Pb + ub;
#pragma unroll
        for (int j = 0; j < G + W - 1; ++j) { const double p = pl[j];
#pragma unroll
            for (int k = 0; k < G; ++k) { const int c = j - k; if (c >= 0 && c < W) acc[k] = fma(p, vr[c], acc[k]); } }
    };
    build(0, P0);
    int r = 0;
    for (; r + 1 < vm.R; r += 2) {
        build(r + 1, P1);
        sweep(r, P0);
        if (r + 2 < vm.R) build(r + 2, P0);
        sweep(r + 1, P1);
    }
    if (r < vm.R) sweep(r, P0);
    const long long ob = ct.out_off[chunk];
#pragma unroll
    for (int k = 0; k < G; ++k) { const int g = x0 + ub + k; if (g < L) { bg[ob + g] = acc[k] + q[k % NQ]; norm[ob + g] = acc[k]; } }
}

template <int G>
float run3(const ChunkTable &ct, const VMatDev &vm, int nc, int L, double *d_a, double *d_b, double *d_o1, double *d_o2, int reps) {
    const int TW = 64 * G;
    std::vector<int2> tiles;
    for (int i = 0; i < nc; ++i) for (int x = 0; x < L; x += TW) tiles.push_back(make_int2(i, x));
    int2 *d_t; CK(hipMalloc(&d_t, tiles.size() * sizeof(int2)));
    CK(hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    const int PW = TW + 120, EW = PW + 249;
    size_t lds = ((size_t)((EW + 1) & ~1) + 2 * ((PW + 1) & ~1)) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((bg_pipelined<G, 121>), dim3(tiles.size()), dim3(64), lds, 0, ct, d_t, vm, d_a, d_b, d_o1, d_o2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    double flop = 2.0 * 146 * 121 * (double)nc * L;
    printf("G=%2d PIPELINED tiles=%zu lds=%zu  %.3f ms  %.2f TFLOP/s (useful)  %.1f Mbp/s\n", G, tiles.size(), lds, best, flop / best / 1e9, (double)nc * L / best / 1e3);
    CK(hipFree(d_t));
    return best;
}

// ---- variant 6: explicit 4-deep register prefetch of the product row in the sweep (scalar-loaded template as MODE 0)
// ---- variant 7: two rows per barrier (two product buffers built, then both swept)
template <int G, int W, int MODE>
__global__ void __launch_bounds__(64) bg_variant2(ChunkTable ct, const int2 *__restrict__ tiles, VMatDev vm,
                                                    const double *__restrict__ nuc_cov, const double *__restrict__ raw,
                                                    double *__restrict__ bg, double *__restrict__ norm) {
    constexpr int TW = WAVE * G, HW = W / 2, PW = TW + W - 1, NQ = (PW + WAVE - 1) / WAVE, PWP = (PW + 1) & ~1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, x0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int A = (vm.upper - 2) >> 1, Bh = (vm.upper - 1) >> 1;
    const int EW = PW + A + Bh;
    double *Et = smem;
    double *P0 = smem + ((EW + 1) & ~1);
    double *P1 = P0 + PWP;
    {
        const double *b = ct.bias + ct.bias_off[chunk];
        const int nb = L + ct.bias_left + ct.bias_right;
        const int j0 = x0 - HW - A + ct.bias_left;
        for (int u = lane; u < EW; u += WAVE) { const int j = j0 + u; Et[u] = (j >= 0 && j < nb) ? exp(b[j]) : 0.0; }
    }
    __syncthreads();
    double acc[G]; double q[NQ];
#pragma unroll
    for (int k = 0; k < G; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = 0.0;
    const int ub = lane * G;
    auto build = [&](int r, double *__restrict__ Pb) {
        const int i = vm.lower + r;
        const int hl = floor_half(i - 1), hr = floor_half(i);
        const double s = vm.srow[r];
        const double *el = Et + (A - hl); const double *er = Et + (A + hr);
#pragma unroll
        for (int k = 0; k < NQ; ++k) { const int u = lane + WAVE * k; if (u < PW) { const double p = (s * el[u]) * er[u]; q[k] += p; Pb[u] = p; } }
    };
    auto sweep = [&](int r, const double *__restrict__ Pb) {
        const double *__restrict__ vr = vm.mat + r * W;
        const double *pl = Pb + ub;
        if (MODE == 6) {
            constexpr int D = 4;
            double pw[D];
#pragma unroll
            for (int d = 0; d < D; ++d) pw[d] = pl[d];
#pragma unroll
            for (int j = 0; j < G + W - 1; ++j) {
                const double p = pw[j % D];
                if (j + D < G + W - 1) pw[j % D] = pl[j + D];
#pragma unroll
                for (int k = 0; k < G; ++k) { const int c = j - k; if (c >= 0 && c < W) acc[k] = fma(p, vr[c], acc[k]); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < G + W - 1; ++j) { const double p = pl[j];
#pragma unroll
                for (int k = 0; k < G; ++k) { const int c = j - k; if (c >= 0 && c < W) acc[k] = fma(p, vr[c], acc[k]); } }
        }
    };
    if (MODE == 6) {
        for (int r = 0; r < vm.R; ++r) { build(r, P0); __syncthreads(); sweep(r, P0); __syncthreads(); }
    } else {
        int r = 0;
        for (; r + 1 < vm.R; r += 2) { build(r, P0); build(r + 1, P1); __syncthreads(); sweep(r, P0); sweep(r + 1, P1); __syncthreads(); }
        if (r < vm.R) { build(r, P0); __syncthreads(); sweep(r, P0); }
    }
    const long long ob = ct.out_off[chunk];
#pragma unroll
    for (int k = 0; k < G; ++k) { const int g = x0 + ub + k; if (g < L) { bg[ob + g] = acc[k] + q[k % NQ]; norm[ob + g] = acc[k]; } }
}

template <int G, int MODE>
float run4(const ChunkTable &ct, const VMatDev &vm, int nc, int L, double *d_a, double *d_b, double *d_o1, double *d_o2, int reps) {
    const int TW = 64 * G;
    std::vector<int2> tiles;
    for (int i = 0; i < nc; ++i) for (int x = 0; x < L; x += TW) tiles.push_back(make_int2(i, x));
    int2 *d_t; CK(hipMalloc(&d_t, tiles.size() * sizeof(int2)));
    CK(hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    const int PW = TW + 120, EW = PW + 249;
    size_t lds = ((size_t)((EW + 1) & ~1) + 2 * ((PW + 1) & ~1)) * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((bg_variant2<G, 121, MODE>), dim3(tiles.size()), dim3(64), lds, 0, ct, d_t, vm, d_a, d_b, d_o1, d_o2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    double flop = 2.0 * 146 * 121 * (double)nc * L;
    printf("G=%2d VARIANT %d tiles=%zu lds=%zu  %.3f ms  %.2f TFLOP/s (useful)\n", G, MODE, tiles.size(), lds, best, flop / best / 1e9);
    CK(hipFree(d_t));
    return best;
}

int main(int argc, char **argv) {
    int nc = argc > 1 ? atoi(argv[1]) : 20000, L = 2120, F = 0;
    const int R = 146, W = 121, lo = 105, up = 251, bl = 246, br = 247;
    std::vector<int> len(nc, L); std::vector<long long> foff(nc + 1, 0), boff(nc + 1), ooff(nc + 1);
    for (int i = 0; i <= nc; ++i) { boff[i] = (long long)i * (L + bl + br); ooff[i] = (long long)i * L; }
    std::vector<double> bias((size_t)nc * (L + bl + br)); for (auto &x : bias) x = (rand() / (double)RAND_MAX - 0.5) * 2.0;
    std::vector<double> vm(R * W), srow(R, 0.01); for (auto &x : vm) x = rand() / (double)RAND_MAX * 0.01;
    ChunkTable ct{}; VMatDev v{};
    int *d_len; long long *d_foff, *d_boff, *d_ooff; double *d_bias, *d_vm, *d_srow, *d_a, *d_b, *d_o1, *d_o2;
    CK(hipMalloc(&d_len, nc * 4)); CK(hipMalloc(&d_foff, (nc + 1) * 8)); CK(hipMalloc(&d_boff, (nc + 1) * 8)); CK(hipMalloc(&d_ooff, (nc + 1) * 8));
    CK(hipMalloc(&d_bias, bias.size() * 8)); CK(hipMalloc(&d_vm, vm.size() * 8)); CK(hipMalloc(&d_srow, R * 8));
    size_t nbp = (size_t)nc * L;
    CK(hipMalloc(&d_a, nbp * 8)); CK(hipMalloc(&d_b, nbp * 8)); CK(hipMalloc(&d_o1, nbp * 8)); CK(hipMalloc(&d_o2, nbp * 8));
    CK(hipMemset(d_a, 0, nbp * 8)); CK(hipMemset(d_b, 0, nbp * 8));
    CK(hipMemcpy(d_len, len.data(), nc * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_foff, foff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_boff, boff.data(), (nc + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ooff, ooff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, bias.data(), bias.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vm, vm.data(), vm.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_srow, srow.data(), R * 8, hipMemcpyHostToDevice));
    ct.nc = nc; ct.chunk_len = d_len; ct.frag_off = d_foff; ct.bias_off = d_boff; ct.bias = d_bias; ct.bias_left = bl; ct.bias_right = br; ct.out_off = d_ooff;
    v.mat = d_vm; v.srow = d_srow; v.lower = lo; v.upper = up; v.w = 60; v.R = R; v.W = W;
    (void)F;
    run4<9, 6>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run4<17, 6>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run4<9, 7>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run4<17, 7>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<9, 4>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<17, 4>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<13, 4>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<13, 0>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<9, 2>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<17, 2>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<9, 0>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<17, 0>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<9, 1>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<17, 1>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<7, 1>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    run<13, 1>(ct, v, nc, L, d_a, d_b, d_o1, d_o2, 2);
    return 0;
}
