// microbench_occ.hip -- ablation timing of the occupancy MLE kernel (development tool)
#include "../nucleoatac_amd/csrc/natac_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
using namespace natac;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ABL, int ROWS = 0>
void run(const ChunkTable &ct, const OccModelDev &om, int2 *d_t, int2 *d_r, int ntiles, double *g0, double *g1, double *g2, int *st, size_t lds, long long bp) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((natac_occ_mle<5, 60, ABL, ROWS>), dim3(ntiles), dim3(256), lds, 0, ct, d_t, d_r, om, g0, g1, g2, st, nullptr, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("ABL=%d ROWS=%d  %.3f ms  %.1f Mbp/s\n", ABL, ROWS, best, bp / best / 1e3);
}

int main(int argc, char **argv) {
    int nc = argc > 1 ? atoi(argv[1]) : 20000, L = 2120, F = 500;
    const int U = 251, bl = 246, br = 247, step = 5, half = 2, fl = 60;
    std::vector<int> len(nc, L); std::vector<long long> foff(nc + 1), boff(nc + 1), ooff(nc + 1), goff(nc + 1);
    const int nk = (L - half + step - 1) / step;
    for (int i = 0; i <= nc; ++i) { foff[i] = (long long)i * F; boff[i] = (long long)i * (L + bl + br); ooff[i] = (long long)i * L; goff[i] = (long long)i * nk; }
    std::vector<double> bias((size_t)nc * (L + bl + br)); for (auto &x : bias) x = (rand() / (double)RAND_MAX - 0.5) * 2.0;
    std::vector<int> cen((size_t)nc * F), iln((size_t)nc * F), lp((size_t)nc * F);
    for (int i = 0; i < nc; ++i) { std::vector<int> c(F); for (auto &x : c) x = rand() % (L + 252) - 126; std::sort(c.begin(), c.end());
        for (int f = 0; f < F; ++f) { cen[(size_t)i * F + f] = c[f]; iln[(size_t)i * F + f] = 20 + rand() % 330; lp[(size_t)i*F+f] = c[f] - (iln[(size_t)i*F+f]-1)/2; } }
    std::vector<double> nucp(U), nfrp(U), al(101);
    double s1 = 0, s2 = 0; for (int j = 0; j < U; ++j) { nucp[j] = exp(-0.5 * pow((j - 185.0) / 18.0, 2)) + 1e-9; nfrp[j] = exp(-j / 60.0) + 1e-9; s1 += nucp[j]; s2 += nfrp[j]; }
    for (int j = 0; j < U; ++j) { nucp[j] /= s1; nfrp[j] /= s2; }
    for (int a = 0; a < 101; ++a) al[a] = a * 0.01;
    int *d_len, *d_cen, *d_iln, *d_lp, *d_st; long long *d_foff, *d_boff, *d_ooff, *d_goff; double *d_bias, *d_nucp, *d_nfrp, *d_al, *g0, *g1, *g2;
    CK(hipMalloc(&d_len, nc * 4)); CK(hipMalloc(&d_foff, (nc + 1) * 8)); CK(hipMalloc(&d_boff, (nc + 1) * 8)); CK(hipMalloc(&d_ooff, (nc + 1) * 8)); CK(hipMalloc(&d_goff, (nc + 1) * 8));
    CK(hipMalloc(&d_bias, bias.size() * 8)); CK(hipMalloc(&d_cen, cen.size() * 4)); CK(hipMalloc(&d_iln, cen.size() * 4)); CK(hipMalloc(&d_lp, cen.size() * 4)); CK(hipMalloc(&d_st, nc * 4));
    CK(hipMalloc(&d_nucp, U * 8)); CK(hipMalloc(&d_nfrp, U * 8)); CK(hipMalloc(&d_al, 101 * 8));
    size_t ng = (size_t)nc * nk; CK(hipMalloc(&g0, ng * 8)); CK(hipMalloc(&g1, ng * 8)); CK(hipMalloc(&g2, ng * 8));
    CK(hipMemset(d_st, 0, nc * 4));
    CK(hipMemcpy(d_len, len.data(), nc * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_foff, foff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_boff, boff.data(), (nc + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ooff, ooff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_goff, goff.data(), (nc + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, bias.data(), bias.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cen, cen.data(), cen.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_iln, iln.data(), cen.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_lp, lp.data(), cen.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nucp, nucp.data(), U * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_nfrp, nfrp.data(), U * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_al, al.data(), 101 * 8, hipMemcpyHostToDevice));
    ChunkTable ct{}; ct.nc = nc; ct.chunk_len = d_len; ct.frag_off = d_foff; ct.lpos = d_lp; ct.ilen = d_iln; ct.centre = d_cen; ct.bias_off = d_boff; ct.bias = d_bias; ct.bias_left = bl; ct.bias_right = br; ct.out_off = d_ooff; ct.grid_off = d_goff;
    OccModelDev om{}; om.nuc_probs = d_nucp; om.nfr_probs = d_nfrp; om.alphas = d_al; om.upper = U; om.n_alpha = 101; om.step = step; om.halfstep = half; om.flank = fl; om.cutoff = 2.705543454095404; om.zero_flags = 0; om.b_floor = 1e-290;
    std::vector<int2> tiles; for (int i = 0; i < nc; ++i) for (int k = 0; k < nk; k += OCC_T * OCC_NP) tiles.push_back(make_int2(i, k));
    int2 *d_t; CK(hipMalloc(&d_t, tiles.size() * sizeof(int2))); CK(hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    const int UP = (U + 1) & ~1, span = (OCC_T * OCC_NP - 1) * step + 2 * fl + 1 + step, EW = span + ((U - 2) >> 1) + ((U - 1) >> 1) + 2;
    size_t lds = ((size_t)((EW + 1) & ~1) + (size_t)OCC_T * UP + 2 * (size_t)UP + OCC_ACL + (((OCC_T - 1) * step + 2 * fl + 1 + step + 3) & ~1)) * 8 + 2 * OCC_FMAX * 4;
    int2 *d_r; CK(hipMalloc(&d_r, tiles.size() * sizeof(int2)));
    { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventRecord(e0));
      hipLaunchKernelGGL(natac_occ_tile_ranges, dim3((tiles.size() + 255) / 256), dim3(256), 0, 0, ct, d_t, (int)tiles.size(), step, half, fl, d_r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("tile ranges %.3f ms\n", ms); }
    printf("tiles=%zu lds=%zu grid points=%zu\n", tiles.size(), lds, ng);
    long long bp = (long long)nc * L;
    std::vector<double> ref(3 * ng), got(3 * ng);
    run<0>(ct, om, d_t, d_r, tiles.size(), g0, g1, g2, d_st, lds, bp);
    CK(hipMemcpy(ref.data(), g0, ng * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref.data() + ng, g1, ng * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref.data() + 2 * ng, g2, ng * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(g0, 0, ng * 8)); CK(hipMemset(g1, 0, ng * 8)); CK(hipMemset(g2, 0, ng * 8));
    run<0, 1>(ct, om, d_t, d_r, tiles.size(), g0, g1, g2, d_st, lds, bp);
    CK(hipMemcpy(got.data(), g0, ng * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(got.data() + ng, g1, ng * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(got.data() + 2 * ng, g2, ng * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, nan = 0;
    for (size_t i = 0; i < 3 * ng; ++i) {
        const bool n1 = ref[i] != ref[i], n2 = got[i] != got[i];
        if (n1) ++nan;
        if (n1 != n2 || (!n1 && ref[i] != got[i])) { if (bad < 5) printf("mismatch at %zu (track %zu, idx %zu): %g vs %g\n", i, i / ng, i % ng, ref[i], got[i]); ++bad; }
    }
    printf("rows vs wave-per-point: %zu mismatches of %zu (%zu NaN in ref)\n", bad, 3 * ng, nan);
    run<7, 1>(ct, om, d_t, d_r, tiles.size(), g0, g1, g2, d_st, lds, bp);
    run<2, 1>(ct, om, d_t, d_r, tiles.size(), g0, g1, g2, d_st, lds, bp);
    run<1>(ct, om, d_t, d_r, tiles.size(), g0, g1, g2, d_st, lds, bp);
    run<3>(ct, om, d_t, d_r, tiles.size(), g0, g1, g2, d_st, lds, bp);
    return 0;
}
