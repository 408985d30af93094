// test_fft_bg.hip -- development harness: FFT background kernel vs the direct kernel (accuracy + timing)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 tools/test_fft_bg.hip -o tools/mb_fft
#include "../nucleoatac_amd/csrc/natac_fft_bg.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace natac;
static double *g_x1 = nullptr, *g_x2 = nullptr;   // the kernels' bnum / bcov outputs
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// variant: 0 = the product kernel (one wave per workgroup), plain tiles; 1 = the same with extended tiles (edge pass at the end of the tile's wave)
// where the library would use them.  (Round 5's multi-wave workgroup variants natac_background_fft_wg<NW, SYNC> -- measured slower,
// profiles/r5/fft_hfold_and_multiwave_workgroups.txt -- left the sources with commit 8c075d2's successor.)
float run_fft(const ChunkTable &ct, const VMatDev &v, int nc, int L, const double *d_tw, const double *d_k, double *d_a, double *d_b,
              double *d_o1, double *d_o2, int reps, int variant = 0) {
    const int TV = FFT_N - v.W + 1;
    std::vector<int2> tiles;
    std::vector<int> ext_list;
    const int TVX = TV + 2 * FFT_EXT;
    // the library's rule (natac_api.hip, bg_chunk_tiling): n tiles of which the first k are extended, 100 n + 11 k smallest
    int bn = (L + TV - 1) / TV, bk = 0;
    if (variant == 1) {
        long long best = 100LL * bn;
        for (int n = bn - 1; n >= 1 && (long long)n * TVX >= L; --n) {
            const int k = (int)((L - (long long)n * TV + 2 * FFT_EXT - 1) / (2 * FFT_EXT));
            if (100LL * n + 11LL * k < best) { best = 100LL * n + 11LL * k; bn = n; bk = k; }
        }
    }
    const bool ext = bk > 0;
    for (int i = 0; i < nc; ++i) {
        int x = 0;
        for (int t = 0; t < bn; ++t) {
            if (t < bk) { ext_list.push_back((int)tiles.size()); tiles.push_back(make_int2(i, (x + FFT_EXT) | FFT_EXT_BIT)); x += TVX; }
            else { tiles.push_back(make_int2(i, x)); x += TV; }
        }
    }
    int2 *d_t; CK(hipMalloc(&d_t, tiles.size() * sizeof(int2)));
    CK(hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    double *d_mtab = nullptr, *d_swt = nullptr;
    const int NJ = (v.R + 3) / 4;
    if (ext) {
        CK(hipMalloc(&d_mtab, 2 * NJ * 64 * 8)); CK(hipMalloc(&d_swt, 4 * NJ * 8));
        hipLaunchKernelGGL(natac_fft_edge_table_mfma, dim3((2 * NJ * 64 + 255) / 256), dim3(256), 0, 0, v.mat, v.srow, v.R, v.W, NJ, d_mtab, d_swt);
    }
    const size_t lds = bg_fft_lds_bytes(v.upper);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    const int nt = (int)tiles.size();
    if (getenv("NATAC_HARNESS_REPS")) reps = atoi(getenv("NATAC_HARNESS_REPS"));     // many back-to-back launches (tools/r6_power.sh samples power and clock under them)
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipEventRecord(e0));
        {
                hipLaunchKernelGGL(natac_background_fft, dim3(nt), dim3(64), lds, 0, ct, d_t, v, d_tw, d_k, d_a, d_b, d_o1, d_o2, g_x1, g_x2, (unsigned)nt, d_mtab, d_swt, NJ);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("FFT variant=%d tiles=%zu (extended %zu) lds=%zu  %.3f ms  %.1f Mbp/s\n", variant, tiles.size(), ext_list.size(), lds, best, (double)nc * L / best / 1e3);
    CK(hipFree(d_t));
    return best;
}

int main(int argc, char **argv) {
    int nc = argc > 1 ? atoi(argv[1]) : 20000, L = argc > 2 ? atoi(argv[2]) : 2120;
    const int R = getenv("NATAC_HARNESS_R") ? atoi(getenv("NATAC_HARNESS_R")) : 146, W = 121;      // NATAC_HARNESS_R=2: one row pair -- what a tile costs besides its pair loop
    const int lo = getenv("NATAC_HARNESS_LO") ? atoi(getenv("NATAC_HARNESS_LO")) : 105, up = lo + R, bl = 246, br = 247;   // NATAC_HARNESS_LO=104: an even first insert size
    std::vector<int> len(nc, L); std::vector<long long> foff(nc + 1, 0), boff(nc + 1), ooff(nc + 1);
    for (int i = 0; i <= nc; ++i) { boff[i] = (long long)i * (L + bl + br); ooff[i] = (long long)i * L; }
    std::vector<double> bias((size_t)nc * (L + bl + br)); for (auto &x : bias) x = (rand() / (double)RAND_MAX - 0.5) * 3.0 - 4.0;
    std::vector<double> vm(R * W), srow(R); for (auto &x : vm) x = rand() / (double)RAND_MAX * 0.01;
    for (int r = 0; r < R; ++r) srow[r] = 0.002 + 0.01 * rand() / (double)RAND_MAX;
    size_t nbp = (size_t)nc * L;
    std::vector<double> ncov(nbp), raw(nbp);
    for (auto &x : ncov) x = 1.0 + rand() % 50; for (auto &x : raw) x = rand() / (double)RAND_MAX;
    ChunkTable ct{}; VMatDev v{};
    int *d_len; long long *d_foff, *d_boff, *d_ooff; double *d_bias, *d_vm, *d_srow, *d_a, *d_b, *d_o1, *d_o2, *d_p1, *d_p2;
    CK(hipMalloc(&d_len, nc * 4)); CK(hipMalloc(&d_foff, (nc + 1) * 8)); CK(hipMalloc(&d_boff, (nc + 1) * 8)); CK(hipMalloc(&d_ooff, (nc + 1) * 8));
    CK(hipMalloc(&d_bias, bias.size() * 8)); CK(hipMalloc(&d_vm, vm.size() * 8)); CK(hipMalloc(&d_srow, R * 8));
    CK(hipMalloc(&d_a, nbp * 8)); CK(hipMalloc(&d_b, nbp * 8)); CK(hipMalloc(&d_o1, nbp * 8)); CK(hipMalloc(&d_o2, nbp * 8));
    CK(hipMalloc(&d_p1, nbp * 8)); CK(hipMalloc(&d_p2, nbp * 8));
    double *d_x1, *d_x2; CK(hipMalloc(&d_x1, nbp * 8)); CK(hipMalloc(&d_x2, nbp * 8));
    CK(hipMalloc(&g_x1, nbp * 8)); CK(hipMalloc(&g_x2, nbp * 8));     // the FFT run's bnum / bcov (the direct kernel's stay in d_x1 / d_x2)
    CK(hipMemcpy(d_a, ncov.data(), nbp * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, raw.data(), nbp * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, len.data(), nc * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_foff, foff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_boff, boff.data(), (nc + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ooff, ooff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, bias.data(), bias.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vm, vm.data(), vm.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_srow, srow.data(), R * 8, hipMemcpyHostToDevice));
    {   // exp(bias) array the kernels stage their windows from (natac_exp_bias in the library)
        std::vector<double> eb(bias.size());
        for (size_t i = 0; i < bias.size(); ++i) eb[i] = exp(bias[i]);
        double *d_eb; CK(hipMalloc(&d_eb, eb.size() * 8)); CK(hipMemcpy(d_eb, eb.data(), eb.size() * 8, hipMemcpyHostToDevice));
        ct.ebias = d_eb;
    }
    ct.nc = nc; ct.chunk_len = d_len; ct.frag_off = d_foff; ct.bias_off = d_boff; ct.bias = d_bias; ct.bias_left = bl; ct.bias_right = br; ct.out_off = d_ooff;
    v.mat = d_vm; v.srow = d_srow; v.lower = lo; v.upper = up; v.w = 60; v.R = R; v.W = W;
    // twiddles + template spectra
    std::vector<double> tw(2 * FFT_N);
    for (int k = 0; k < FFT_N; ++k) { tw[2 * k] = cos(2 * M_PI * k / FFT_N); tw[2 * k + 1] = -sin(2 * M_PI * k / FFT_N); }
    double *d_tw, *d_k; const int npair = (R + 1) / 2;
    CK(hipMalloc(&d_tw, tw.size() * 8)); CK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_k, (size_t)npair * 2 * FFT_N * 8));
    hipLaunchKernelGGL(natac_fft_template, dim3(npair), dim3(64), 0, 0, d_vm, d_srow, R, W, d_tw, d_k);
    CK(hipDeviceSynchronize());
    // direct kernel (G=17)
    if (!(argc > 4)) {
        constexpr int G = 17; const int TW = 64 * G;
        std::vector<int2> tiles;
        for (int i = 0; i < nc; ++i) for (int x = 0; x < L; x += TW) tiles.push_back(make_int2(i, x));
        int2 *d_t; CK(hipMalloc(&d_t, tiles.size() * sizeof(int2)));
        CK(hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
        const int PW = TW + 120, EW = PW + 249;
        size_t lds = ((size_t)((EW + 1) & ~1) + PW) * 8;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((natac_background<G, 121>), dim3(tiles.size()), dim3(64), lds, 0, ct, d_t, v, d_a, d_b, d_p1, d_p2, d_x1, d_x2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("direct G=17  %.3f ms  %.1f Mbp/s\n", best, (double)nc * L / best / 1e3);
    }
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    run_fft(ct, v, nc, L, d_tw, d_k, d_a, d_b, d_o1, d_o2, 2, variant);
    {
        std::vector<double> x(nbp), y(nbp);
        CK(hipMemcpy(x.data(), d_p1, nbp * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), d_o1, nbp * 8, hipMemcpyDeviceToHost));
        double maxrel = 0, maxabs = 0; size_t bad = 0, worst = 0;
        for (size_t i = 0; i < nbp; ++i) {
            const double d = fabs(x[i] - y[i]), r = d / (fabs(x[i]) + 1e-300);
            if (!(d == d)) { if (!(x[i] != x[i] && y[i] != y[i])) ++bad; continue; }
            if (r > maxrel) { maxrel = r; worst = i; }
            if (d > maxabs) maxabs = d;
        }
        printf("bg: max rel err %.3e (at %zu: %.17g vs %.17g)  max abs %.3e  nan-mismatch %zu\n", maxrel, worst, x[worst], y[worst], maxabs, bad);
        printf("sample: %.6g %.6g | %.6g %.6g | %.6g %.6g\n", x[0], y[0], x[391], y[391], x[392], y[392]);
        {   // the FFT run's four outputs as one 64-bit FNV-1a hash: equal between two builds <=> bit-identical outputs
            const double *pp[4] = {d_o1, d_o2, g_x1, g_x2};
            unsigned long long h = 1469598103934665603ull;
            std::vector<double> z(nbp);
            for (int k = 0; k < 4; ++k) {
                CK(hipMemcpy(z.data(), pp[k], nbp * 8, hipMemcpyDeviceToHost));
                const unsigned long long *u = (const unsigned long long *)z.data();
                for (size_t i = 0; i < nbp; ++i) { h ^= u[i]; h *= 1099511628211ull; }
            }
            printf("outputs fnv1a64 %016llx\n", h);
        }
        const double *pa[3] = {d_p2, d_x1, d_x2}, *pb[3] = {d_o2, g_x1, g_x2};
        const char *nm[3] = {"norm", "bnum", "bcov"};
        for (int k = 0; k < 3 && !(argc > 4); ++k) {
            CK(hipMemcpy(x.data(), pa[k], nbp * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), pb[k], nbp * 8, hipMemcpyDeviceToHost));
            double mr = 0; size_t w = 0;
            for (size_t i = 0; i < nbp; ++i) { const double r = fabs(x[i] - y[i]) / (fabs(x[i]) + (k ? 1e-300 : 1e-3)); if (!(r <= mr)) { mr = r; w = i; } }
            printf("%s: max rel err %.3e (at %zu = base %zu: %.17g vs %.17g)\n", nm[k], mr, w, w % L, x[w], y[w]);
        }
    }
    return 0;
}
