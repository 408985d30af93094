// natac_occ_fast.hpp -- the occupancy grid MLE (nucleoatac/Occupancy.py:104-146) restructured around two facts.
//
// (1) The window bias of an insert size cancels out of the decision.  With pn[j] = nuc_probs[j] b[j] / sn and
//     pf[j] = nfr_probs[j] b[j] / sf (Occupancy.py:108-111), the likelihood of alpha over the window's fragments is
//         L(alpha) = prod_f (alpha pn[n_f] + (1 - alpha) pf[n_f]) = (prod_f pf[n_f]) * prod_f (1 + alpha (rho[n_f] kappa - 1)),
//     rho[n] = nuc_probs[n] / nfr_probs[n] (model constant), kappa = sf / sn (one number per grid point).  The first factor
//     is the same for every alpha, so argmax and the likelihood-ratio interval 2 (max ll - ll) < cutoff only need kappa:
//     the 251 window sums b[j] are needed only through  sn = sum_j nuc_probs[j] b[j]  and  sf = sum_j nfr_probs[j] b[j].
// (2) sn is a box sum over the window's bases of  g_n(x) = sum_j nuc_probs[j] B0[j, x],  B0[j, x] = E[x-(j-1)//2] E[x+j//2]
//     (chunkmat2d.py:140-153; j == 1: E[x]).  With j = 2m, 2m+1:
//         g(x) = E[x] (p0 E[x+1] + p1) + sum_{m>=1} E[x+m] (p_{2m} E[x-m+1] + p_{2m+1} E[x-m]).
//     The window of grid point k (2 flank + 1 bases, starting on a block border) is Q = (2 flank + 1) / step aligned step-blocks
//     plus the first REM = (2 flank + 1) % step bases of block k + Q (the defaults: 24 blocks + 1 base), so natac_occ_gsum forms
//     per block the sums of g_n, g_f over its bases (the base loop merged into the m loop: 18 fp64 ops per m and block instead of
//     ~30 per base) and over its first REM bases, ONCE per block; natac_occ_decide adds Q + 1 of them per grid point.
// (3) log L(alpha) is concave in alpha, so the first maximum over the 101-point alpha grid and the two ends of the
//     likelihood-ratio interval are found by a fixed-schedule multi-section search (31 likelihood evaluations per grid
//     point in five rounds instead of 101), one LANE per grid point: no cross-lane reductions at all.
// The arithmetic differs from the reference's sum of logs at the 1e-15 relative level (as did the product-domain kernel it
// replaces); the alpha indices are identical on every golden vector and on 848,000 grid points of the configs[2] batch
// (tests/test_gpu_properties.py).  Tiles the fast path cannot treat exactly like the reference -- exp(bias) values that
// are not finite or could make a probability under- / overflow -- are handed to the general kernel natac_occ_mle through a
// device-side list; models with an insert size of probability zero under BOTH distributions, alpha grids that are not increasing or
// longer than 101 values, or steps beyond 9 use natac_occ_mle for everything.  Any odd step up to 9 (the CLI's --step) and any flank (--flank) stay here.
#pragma once
#include "natac_kernels.hpp"

namespace natac {

constexpr int GS_BLOCKS = 64;    // step-blocks per natac_occ_gsum workgroup (one per lane; the 4 waves split the m range)
constexpr int GS_MG = 8;         // m values per fully unrolled register window
constexpr int OD_FM = 512;       // valid fragments of a 64-grid-point tile staged in LDS (denser tiles read global memory)
constexpr int OD_NA = 101;       // longest alpha grid of the fast path (the schedule's coarse points are 0, 10, ..., 100, clipped)

struct OccFastDev {
    const double *q4;       // [nm][4] = {nuc_probs[2m], nuc_probs[2m+1], nfr_probs[2m], nfr_probs[2m+1]} (0 beyond upper)
    const double *rho;      // [upper] nuc_probs / nfr_probs
    const double *alphas;   // [na]
    int na;                 // number of alphas, <= OD_NA
    int nm;                 // (upper + 1) / 2
    int upper, step, halfstep, flank, Q;
    int flags;              // bit 0: some nuc_prob == 0 (alpha with 1 - alpha == 0 is excluded, Occupancy.py:112-114);
                            // bit 1: some nfr_prob == 0 (alpha == 0 is excluded likewise; such a bin has rho = +inf, see occ_eval)
    double ci_factor;       // exp(-cutoff / 2)
    double e_lo, e_hi;      // exp(bias) values outside [e_lo, e_hi] = 2^-+190 (or NaN) send a tile to the general kernel: inside, no
                            // probability nuc_probs[j] b[j] / sn of the reference can under- or overflow (b ratio >= 2^-760 / 121,
                            // model probabilities >= 2^-200), so its zero pattern is the model's and the ratio form is equivalent
};

// ---- per-block sums of g_n, g_f --------------------------------------------------------------------------------
// tile = (chunk, b0): blocks b0 .. b0 + 63 of the chunk; block b covers bases xs + b step .. + step - 1, xs = halfstep - flank.
// out[0..3][blk_off[chunk] + b] = {sum g_n, sum g_f, g_n over the first REM bases, g_f over the first REM bases}.
template <int STEP, int REM>
__global__ void __launch_bounds__(256) natac_occ_gsum(ChunkTable ct, const int2 *__restrict__ tiles, OccFastDev om,
                                                        const long long *__restrict__ blk_off, long long total_blocks,
                                                        double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int bad_s;
    const int R = om.nm - 1;                                  // reach: E[x - R] .. E[x + R]
    const int EW = GS_BLOCKS * STEP + 2 * R + 2 * GS_MG;      // + GS_MG on both sides: the register windows of the last m group
    double *Et = smem;                                        // [EWP]  Et[u] <-> coordinate xt0 + u
    double *red = smem + ((EW + 1) & ~1);                     // [4 values][4 parts][64]
    const int2 t = tiles[blockIdx.x];
    const int chunk = t.x, b0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int nk = (L - om.halfstep + STEP - 1) / STEP;
    const int nblk = nk + om.Q;
    const int nb_here = min(GS_BLOCKS, nblk - b0);
    const int xs = om.halfstep - om.flank + b0 * STEP;        // first base of the tile
    const int xt0 = xs - R - GS_MG;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    {
        const double *b = ct.bias ? ct.ebias + ct.bias_off[chunk] : nullptr;       // exp(bias), natac_exp_bias
        const int nbias = L + ct.bias_left + ct.bias_right;
        const int need = nb_here * STEP + 2 * R;              // entries [GS_MG, GS_MG + need) feed blocks of this chunk
        bool bad = false;
        for (int u = threadIdx.x; u < EW; u += 256) {
            double e = 1.0;
            if (u >= GS_MG && u < GS_MG + need && b) {
                const int j = xt0 + u + ct.bias_left;
                e = (j >= 0 && j < nbias) ? b[j] : 0.0;
                bad |= !(e >= om.e_lo && e <= om.e_hi);       // NaN, inf, 0, tiny, huge
            }
            Et[u] = e;
        }
        if (bad) bad_s = 1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // m range of the part: [m_lo, m_hi) out of [0, nm)
    const int per = (om.nm + 3) >> 2;
    const int m_lo = part * per, m_hi = min(om.nm, m_lo + per);
    const double *e0 = Et + GS_MG + R + lane * STEP;          // e0[i] = E[x0 + i], x0 = first base of the lane's block
    double Gn = 0.0, Gf = 0.0, gn0 = 0.0, gf0 = 0.0;
    int m = m_lo;
    if (m == 0 && m < m_hi) {
        // m = 0: j = 0 -> E[x] E[x+1], j = 1 -> E[x] alone (the two pattern ones coincide, chunkmat2d.py:150-151)
        const double p0n = om.q4[0], p1n = om.q4[1], p0f = om.q4[2], p1f = om.q4[3];
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
            const double ex = e0[i];
            const double pa = ex * e0[i + 1];
            sa += pa;
            sb += ex;
            if (i == REM - 1) { gn0 = fma(p0n, sa, p1n * sb); gf0 = fma(p0f, sa, p1f * sb); }     // the first REM bases
        }
        Gn = fma(p0n, sa, p1n * sb);
        Gf = fma(p0f, sa, p1f * sb);
        m = 1;
    }
    for (; m < m_hi; m += GS_MG) {
        // register window for m .. m + MG - 1:  ep[i] = E[x0 + m + i] (i < STEP + MG - 1),  em[i] = E[x0 - (m + MG - 1) + i]
        double ep[STEP + GS_MG - 1], em[STEP + GS_MG];
#pragma unroll
        for (int i = 0; i < STEP + GS_MG - 1; ++i) ep[i] = e0[m + i];
#pragma unroll
        for (int i = 0; i < STEP + GS_MG; ++i) em[i] = e0[i - (m + GS_MG - 1)];
#pragma unroll
        for (int d = 0; d < GS_MG; ++d) {
            if (m + d < m_hi) {                               // wave-uniform
                const double *q = om.q4 + 4 * (m + d);        // scalar loads: the index is wave-uniform
                const double pn0 = q[0], pn1 = q[1], pf0 = q[2], pf1 = q[3];
                // base i: E[x+m'] = ep[d + i];  E[x-m'+1] = em[MG - 1 - d + i + 1];  E[x-m'] = em[MG - 1 - d + i]   (m' = m + d)
                double sa = ep[d] * em[GS_MG - d], sb = ep[d] * em[GS_MG - 1 - d];
                if (REM == 1) { gn0 = fma(pn0, sa, fma(pn1, sb, gn0)); gf0 = fma(pf0, sa, fma(pf1, sb, gf0)); }
#pragma unroll
                for (int i = 1; i < STEP; ++i) {
                    sa = fma(ep[d + i], em[GS_MG - d + i], sa);
                    sb = fma(ep[d + i], em[GS_MG - 1 - d + i], sb);
                    if (i == REM - 1) { gn0 = fma(pn0, sa, fma(pn1, sb, gn0)); gf0 = fma(pf0, sa, fma(pf1, sb, gf0)); }   // prefix of REM bases
                }
                Gn = fma(pn0, sa, fma(pn1, sb, Gn));
                Gf = fma(pf0, sa, fma(pf1, sb, Gf));
            }
        }
    }
    red[(0 * 4 + part) * 64 + lane] = Gn;
    red[(1 * 4 + part) * 64 + lane] = Gf;
    red[(2 * 4 + part) * 64 + lane] = gn0;
    red[(3 * 4 + part) * 64 + lane] = gf0;
    __syncthreads();
    {   // wave v adds the four parts of value v (fixed order) and writes the tile's 64 blocks
        const int v = part;
        double s = (red[(v * 4 + 0) * 64 + lane] + red[(v * 4 + 1) * 64 + lane]) +
                   (red[(v * 4 + 2) * 64 + lane] + red[(v * 4 + 3) * 64 + lane]);
        if (bad_s) s = __builtin_nan("");                     // natac_occ_decide sends the tiles that touch these blocks to natac_occ_mle
        if (lane < nb_here) out[(long long)v * total_blocks + blk_off[chunk] + b0 + lane] = s;
    }
}

// ---- decision kernel ---------------------------------------------------------------------------------------------
struct OccLik {           // likelihood as mantissa x 2^exponent, m in [0.5, 1) or 0 (= log-likelihood -inf)
    double m;
    int e;
};
// m -> mantissa in [0.5, 1), e += exponent.  The two instructions frexp() is made of, without its special cases (a likelihood
// here is 0 or a finite positive number far inside the fp64 range: v_frexp_mant / v_frexp_exp give (0, 0) for 0 like frexp does):
// frexp()'s zero / infinity tests were five of the eight instructions of a renormalisation.
__device__ __forceinline__ void lik_renorm(double &m, int &e) {
    e += __builtin_amdgcn_frexp_exp(m);
    m = __builtin_amdgcn_frexp_mant(m);
}
__device__ __forceinline__ bool lik_gt(double m1, int e1, double m2, int e2) {   // L1 > L2
    return m1 > 0.0 && (!(m2 > 0.0) || e1 > e2 || (e1 == e2 && m1 > m2));
}

// K likelihoods per lane: L_k = prod_i (1 + al[k] t_i) over the lane's window fragments [f0, f0 + cnt); every lane of the
// wave runs `trips` (multiple of 4) iterations, fragments past its own window contribute the factor 1.
// GLOBAL: the tile's fragments are read from global memory (cen / iln not compacted: invalid sizes give the factor 1).
// RN: factors between two frexp renormalisations (4 or 16; the host picks 16 when 16 factors cannot leave the fp64 range)
// ZF: the model has insert sizes with nfr_prob == 0 < nuc_prob (rho = +inf).  A fragment of such a size has the probability
//     alpha pn[n] -- the constant pn[n] drops out of every comparison like the prod pf of the others --, i.e. the factor alpha
//     instead of 1 + alpha t: factor = u + alpha t with (u, t) = (0, 1).
template <int K, bool GLOBAL, int RN, bool ZF>
__device__ __forceinline__ void occ_eval(const double (&al)[K], double kappa, int f0, int cnt, int trips, const double *rho_s,
                                         const int *iln_g, const double *rho_g, int U, int flags, double (&m)[K], int (&e)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) { m[k] = 1.0; e[k] = 0; }
    for (int i = 0; i < trips; i += 4) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            // lanes past their own window sit this fragment out under the exec mask (their factor is exactly 1): a predicated block
            // costs one compare, the select form (index, t) cost four VALU instructions per fragment and pass
            if (i + v < cnt) {
                double t, u = 1.0;
                if (GLOBAL) {
                    const int n = iln_g[f0 + i + v];
                    t = 0.0;
                    if (n >= 0 && n < U) {
                        const double r = rho_g[n];
                        t = fma(r, kappa, -1.0);
                        if (ZF && r == __builtin_inf()) { t = 1.0; u = 0.0; }
                    }
                } else {
                    const double r = rho_s[f0 + i + v];
                    t = fma(r, kappa, -1.0);
                    if (ZF && r == __builtin_inf()) { t = 1.0; u = 0.0; }
                }
#pragma unroll
                for (int k = 0; k < K; ++k) m[k] *= ZF ? fma(al[k], t, u) : fma(al[k], t, 1.0);
            }
        }
        if (RN == 4 || ((i >> 2) & (RN / 4 - 1)) == RN / 4 - 1) {      // wave-uniform
#pragma unroll
            for (int k = 0; k < K; ++k) lik_renorm(m[k], e[k]);
        }
    }
    if (RN != 4) {
#pragma unroll
        for (int k = 0; k < K; ++k) lik_renorm(m[k], e[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // alpha with 1 - alpha == 0 while some nuc_prob is 0: 0 * log(0) = NaN -> -inf in the reference (Occupancy.py:112-114)
        if ((flags & 1) && (1 - al[k]) == 0.0) m[k] = 0.0;
        if (ZF && (flags & 2) && al[k] == 0.0) m[k] = 0.0;        // ... and alpha == 0 while some nfr_prob is 0
        if (!(m[k] > 0.0)) { m[k] = 0.0; e[k] = 0; }
    }
}

// One wave = one tile of 64 consecutive grid points of one chunk (the tile table of natac_occ_mle), one lane per grid
// point; a workgroup holds 4 independent waves.  `heavy` (or null; natac_tile_heavy): every lane of a wave runs its tile's longest
// window, so a dense tile is a long wave -- the launch has HEAVY_CAP leading slots that visit the listed heavy tiles first, the slots
// behind them visit all tiles in chunk order and skip the listed ones.  sums[0..3] = the natac_occ_gsum arrays.  Tiles that cannot be decided
// here (non-finite / non-positive normalisers: poisoned blocks) are appended to `defer` for natac_occ_mle.
template <int STEP, int RN, bool ZF>
__global__ void __launch_bounds__(256) natac_occ_decide(ChunkTable ct, const int2 *__restrict__ tiles, int ntiles,
                                                          const int2 *__restrict__ ranges, const int *__restrict__ heavy,
                                                          OccFastDev om, const long long *__restrict__ blk_off, long long total_blocks,
                                                          const double *__restrict__ sums, double *__restrict__ g_occ,
                                                          double *__restrict__ g_lo, double *__restrict__ g_hi,
                                                          int *__restrict__ defer_count, int *__restrict__ defer_list) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Q = om.Q, NG = 64 + Q;                          // blocks a tile's windows touch
    const int NGP = (NG + 1) & ~1;
    // per-wave LDS: rho_s[OD_FM + 4] | gs[4][NGP] | cen_s[OD_FM] (int)
    const int per_wave = (OD_FM + 4) + 4 * NGP + OD_FM / 2;
    double *rho_s = smem + wave * per_wave;
    double *gs = rho_s + OD_FM + 4;
    int *cen_s = (int *)(gs + 4 * NGP);
    int tile = blockIdx.x * 4 + wave;
    if (heavy) {        // heavy = {count, claims, list[HEAVY_CAP], flag bytes[ntiles]}
        const unsigned char *flag = (const unsigned char *)(heavy + 2 + HEAVY_CAP);
        if (tile < HEAVY_CAP) {
            if (tile >= heavy[0]) return;
            tile = heavy[2 + tile];
        } else {
            tile -= HEAVY_CAP;
            if (tile >= ntiles || flag[tile]) return;
        }
    }
    if (tile >= ntiles) return;
    const int2 t = tiles[tile];
    const int chunk = t.x, k0 = t.y;
    const int L = ct.chunk_len[chunk];
    const int nk = (L - om.halfstep + STEP - 1) / STEP;
    const int U = om.upper, fl = om.flank;
    // ---- stage the block sums of blocks k0 .. k0 + 63 + Q
    {
        const long long bo = blk_off[chunk] + k0;
        const int nblk = nk + Q;
        for (int u = lane; u < NG; u += 64) {
            const bool in = k0 + u < nblk;
#pragma unroll
            for (int v = 0; v < 4; ++v) gs[v * NGP + u] = in ? sums[(long long)v * total_blocks + bo + u] : 0.0;
        }
    }
    // ---- stage the tile's valid fragments (0 <= n < U), compacted, with rho[n]
    const int2 tr = ranges[tile];
    const int t0 = tr.x, nt = tr.y - tr.x;
    const int *cen_g = ct.centre + ct.frag_off[chunk] + t0;
    const int *iln_g = ct.ilen + ct.frag_off[chunk] + t0;
    int nv = 0;
    bool staged = true;
    for (int base = 0; base < nt; base += 64) {
        const int i = base + lane;
        int n = -1, c = 0;
        if (i < nt) { n = iln_g[i]; c = cen_g[i]; }
        const bool ok = n >= 0 && n < U;
        const unsigned long long bal = __ballot(ok);
        const int pos = nv + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        const int tot = nv + __popcll(bal);
        if (tot > OD_FM) { staged = false; break; }            // wave-uniform
        if (ok) { cen_s[pos] = c; rho_s[pos] = om.rho[n]; }
        nv = tot;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- this lane's grid point
    const int k = k0 + lane;
    const bool live = k < nk;
    const int g = om.halfstep + k * STEP;
    double sn = 0.0, sf = 0.0;
    for (int b = 0; b < Q; ++b) { sn += gs[lane + b]; sf += gs[NGP + lane + b]; }
    sn += gs[2 * NGP + lane + Q];
    sf += gs[3 * NGP + lane + Q];
    const bool okn = sn > 0.0 && sn < __builtin_inf() && sf > 0.0 && sf < __builtin_inf();
    if (__ballot(live && !okn) != 0ull) {                     // poisoned blocks: the whole tile goes to natac_occ_mle
        if (lane == 0) defer_list[atomicAdd(defer_count, 1)] = tile;
        return;
    }
    const double kappa = live ? sf / sn : 1.0;
    // window [g - fl, g + fl] in the (staged: compacted / global: full) centre list
    int f0, cnt, nins;
    if (staged) {
        int lo = 0, hi = nv;
        while (__ballot(lo < hi) != 0ull) { if (lo < hi) { const int mid = (lo + hi) >> 1; if (cen_s[mid] < g - fl) lo = mid + 1; else hi = mid; } }
        f0 = lo;
        hi = nv;
        while (__ballot(lo < hi) != 0ull) { if (lo < hi) { const int mid = (lo + hi) >> 1; if (cen_s[mid] < g + fl + 1) lo = mid + 1; else hi = mid; } }
        cnt = live ? lo - f0 : 0;
        nins = cnt;
    } else {
        f0 = lower_bound_i32(cen_g, 0, nt, g - fl);
        const int f1 = lower_bound_i32(cen_g, f0, nt, g + fl + 1);
        cnt = live ? f1 - f0 : 0;
        nins = 0;
        for (int i = 0; i < cnt; ++i) { const int n = iln_g[f0 + i]; nins += (n >= 0 && n < U); }
    }
    int trips = row_max_i32(cnt);                             // wave maximum: every lane runs the same number of iterations
    trips = max(max(__builtin_amdgcn_readlane(trips, 0), __builtin_amdgcn_readlane(trips, 16)),
                max(__builtin_amdgcn_readlane(trips, 32), __builtin_amdgcn_readlane(trips, 48)));
    trips = (trips + 3) & ~3;
    const double *al_g = om.alphas;
    const int NA = om.na;                                      // <= OD_NA; coarse point c is index min(10 c, NA - 1)
    auto eval11 = [&](const double (&al)[11], double (&m)[11], int (&e)[11]) {
        if (staged) occ_eval<11, false, RN, ZF>(al, kappa, f0, cnt, trips, rho_s, nullptr, nullptr, U, om.flags, m, e);
        else occ_eval<11, true, RN, ZF>(al, kappa, f0, cnt, trips, nullptr, iln_g, om.rho, U, om.flags, m, e);
    };
    auto eval8 = [&](const double (&al)[8], double (&m)[8], int (&e)[8]) {
        if (staged) occ_eval<8, false, RN, ZF>(al, kappa, f0, cnt, trips, rho_s, nullptr, nullptr, U, om.flags, m, e);
        else occ_eval<8, true, RN, ZF>(al, kappa, f0, cnt, trips, nullptr, iln_g, om.rho, U, om.flags, m, e);
    };
    auto eval2 = [&](const double (&al)[2], double (&m)[2], int (&e)[2]) {
        if (staged) occ_eval<2, false, RN, ZF>(al, kappa, f0, cnt, trips, rho_s, nullptr, nullptr, U, om.flags, m, e);
        else occ_eval<2, true, RN, ZF>(al, kappa, f0, cnt, trips, nullptr, iln_g, om.rho, U, om.flags, m, e);
    };
    // ---- round A: the coarse grid 0, 10, ..., 100
    double cm[11];
    int ce[11];
    {
        double al[11];
#pragma unroll
        for (int c = 0; c < 11; ++c) al[c] = al_g[min(10 * c, NA - 1)];
        eval11(al, cm, ce);
    }
    int cbest = 0;
    double bm = cm[0];
    int be = ce[0];
#pragma unroll
    for (int c = 1; c < 11; ++c)                                // a clipped duplicate of the last point is never strictly larger
        if (lik_gt(cm[c], ce[c], bm, be)) { bm = cm[c]; be = ce[c]; cbest = c; }
    // ---- round B: 10 cbest +- 2, 4, 6, 8 (clipped); the maximum over the alpha grid lies within 10 of the coarse one (concavity)
    int amax = min(10 * cbest, NA - 1);
    {
        double al[8], m[8];
        int e[8], idx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = (j < 4) ? 2 * j - 8 : 2 * j - 6;     // -8 -6 -4 -2 +2 +4 +6 +8
            idx[j] = min(max(10 * cbest + off, 0), NA - 1);
            al[j] = al_g[idx[j]];
        }
        eval8(al, m, e);
        // first maximum among the nine known points around the coarse one, in ascending index order
#pragma unroll
        for (int j = 0; j < 4; ++j)                              // points below: a strictly larger value, or an equal one (lower index wins)
            if (lik_gt(m[j], e[j], bm, be) || (idx[j] < amax && m[j] == bm && e[j] == be)) { bm = m[j]; be = e[j]; amax = idx[j]; }
#pragma unroll
        for (int j = 4; j < 8; ++j)
            if (lik_gt(m[j], e[j], bm, be)) { bm = m[j]; be = e[j]; amax = idx[j]; }
    }
    // ---- round C: amax +- 1
    {
        double al[2], m[2];
        int e[2];
        const int i0 = max(amax - 1, 0), i1 = min(amax + 1, NA - 1);
        al[0] = al_g[i0];
        al[1] = al_g[i1];
        eval2(al, m, e);
        if (lik_gt(m[0], e[0], bm, be) || (i0 < amax && m[0] == bm && e[0] == be)) { bm = m[0]; be = e[0]; amax = i0; }
        if (lik_gt(m[1], e[1], bm, be)) { bm = m[1]; be = e[1]; amax = i1; }
    }
    // ---- likelihood-ratio interval: 2 (max ll - ll) < cutoff  <=>  L > Lmax exp(-cutoff / 2)
    const double thr = bm * om.ci_factor;
    auto inset = [&](double m, int e) {
        const int d = e - be;
        return m > 0.0 && d > -1100 && ldexp(m, d) > thr;
    };
    // brackets from the coarse points: lower end in (pl, sl], upper end in [sh, ph)
    int pl = -1, ph = NA;
#pragma unroll
    for (int c = 0; c < 11; ++c) {
        const bool in = inset(cm[c], ce[c]);
        const int ic = min(10 * c, NA - 1);
        if (!in && ic < amax) pl = ic;                          // the largest such c (ascending loop)
    }
#pragma unroll
    for (int c = 10; c >= 0; --c) {
        const bool in = inset(cm[c], ce[c]);
        const int ic = min(10 * c, NA - 1);
        if (!in && ic > amax) ph = ic;                          // the smallest such c
    }
    const int sl = min(pl + 10, amax), sh = max(ph - 10, amax);
    int ilo, ihi;
    {   // round D: four interior points of each bracket (clipped to the known member of the set)
        double al[8], m[8];
        int e[8], idx[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            idx[j] = max(min(pl + 2 * (j + 1), sl), 0);
            idx[4 + j] = min(max(ph - 2 * (j + 1), sh), NA - 1);
            al[j] = al_g[idx[j]];
            al[4 + j] = al_g[idx[4 + j]];
        }
        eval8(al, m, e);
        // smallest member among the lower points / largest among the upper ones
        ilo = sl;
#pragma unroll
        for (int j = 3; j >= 0; --j) if (inset(m[j], e[j])) ilo = min(ilo, idx[j]);
        ihi = sh;
#pragma unroll
        for (int j = 3; j >= 0; --j) if (inset(m[4 + j], e[4 + j])) ihi = max(ihi, idx[4 + j]);
    }
    {   // round E: the one unknown neighbour on each side
        double al[2], m[2];
        int e[2];
        const int i0 = max(ilo - 1, 0), i1 = min(ihi + 1, NA - 1);
        al[0] = al_g[i0];
        al[1] = al_g[i1];
        eval2(al, m, e);
        if (i0 > pl && inset(m[0], e[0])) ilo = i0;
        if (i1 < ph && inset(m[1], e[1])) ihi = i1;
    }
    if (live) {
        const long long go = ct.grid_off[chunk] + k;
        const double qn = __builtin_nan("");
        if (nins == 0) {                                        // sum(new_inserts) > 0 fails: stay NaN (Occupancy.py:143)
            g_occ[go] = qn; g_lo[go] = qn; g_hi[go] = qn;
        } else {
            g_occ[go] = al_g[amax];
            g_lo[go] = al_g[ilo];
            g_hi[go] = al_g[ihi];
        }
    }
}

}  // namespace natac
