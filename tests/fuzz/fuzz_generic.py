"""Randomised parity sweep over MODEL geometries (development / release check): per round a random V-plot (lower bound, even /
odd row count, width), smoothing width, occupancy window / step / size range / alpha grid -- so the fallback kernels run
(generic smoothing, natac_occ_mle, natac_candidates4, the FFT kernel's odd-row loop, natac_background_generic) -- on a
few ragged chunks, against the CPU oracle.   usage: python tests/fuzz/fuzz_generic.py [n_rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_track, cancel_scale, expand_grid, golden  # noqa: E402
from nucleoatac_amd import _lib as L  # noqa: E402
from nucleoatac_amd.device import Context  # noqa: E402
from nucleoatac_amd.packing import PackedChunks  # noqa: E402
from nucleoatac_amd.synth import synth_occ_distributions, synth_size_distribution  # noqa: E402
from oracle import natac_oracle as O  # noqa: E402


def one_round(rng, par):
    w = int(rng.choice([30, 50, 60]))
    vlo = int(rng.integers(80, 131))
    R = int(rng.integers(40, 147))
    vup = min(vlo + R, 276)
    R = vup - vlo
    if rng.random() < 0.5 and vlo >= 105 and vup <= 251:
        vm = np.ascontiguousarray(par["vmat"][vlo - 105:vup - 105, 60 - w:60 + w + 1])
    else:
        vm = rng.random((R, 2 * w + 1)) * 0.01 + 1e-4
    occ_up = int(rng.choice([200, 251, min(vup, 256)]))      # the occupancy kernels hold size ranges up to 256
    flank = int(rng.choice([33, 45, 60, 61, 62, 75]))        # round 5: any flank / odd step <= 9 / alpha grid <= 101 stays on the block-sum
    step = int(rng.choice([1, 3, 5, 7, 9, 11]))              # kernels; 11 is the general kernel
    n_alpha = int(rng.choice([101, 65, 37, 3]))
    sd = int(rng.choice([7, 10]))
    sizes = synth_size_distribution(max(vup, occ_up, 251))
    nucp, nfrp = synth_occ_distributions(251)
    if occ_up <= 251:
        nucp, nfrp = nucp[:occ_up] / nucp[:occ_up].sum(), nfrp[:occ_up] / nfrp[:occ_up].sum()
    else:
        nucp = np.concatenate((nucp, np.full(occ_up - 251, nucp[-1])))
        nfrp = np.concatenate((nfrp, np.full(occ_up - 251, nfrp[-1])))
        nucp, nfrp = nucp / nucp.sum(), nfrp / nfrp.sum()
    cutoff = 2.705543454095404
    lens = [int(rng.integers(2 * max(flank, 3 * sd, w) + 30, 1500)) for _ in range(int(rng.integers(1, 4)))]
    fr = []
    for Lc in lens:
        nf = int(float(rng.choice([0.05, 0.3, 1.0, 3.0])) * Lc)
        n = rng.integers(1, 330, size=nf)
        c = np.sort(rng.integers(-150, Lc + 150, size=nf))
        fr.append((c - (n - 1) // 2, n))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    BL, BR = 330, 331                 # bias flanks: enough for a V-plot up to 276 with w = 60 (the default 246 / 247 is for 251)
    nb = [Lc + BL + BR for Lc in lens]
    bias = rng.normal(0, 0.7, size=sum(nb))
    pk = PackedChunks(np.arange(len(lens)) * 20000, lens, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]),
                      np.concatenate(([0], np.cumsum(nb))), bias, bias_left=BL, bias_right=BR)
    desc = dict(w=w, vlo=vlo, vup=vup, occ_up=occ_up, flank=flank, step=step, n_alpha=n_alpha, sd=sd, lens=lens)
    with Context(0) as c:
        c.set_vmat(vm, vlo, vup)
        c.set_sizes(sizes[:max(vup, occ_up)])
        c.set_occ_model(nucp, nfrp, alphas=np.linspace(0, 1, n_alpha), cutoff=cutoff, step=step, flank=flank)
        b = c.upload(pk)
        b.run_nuc(sd)
        b.run_occ()
        st = b.status()
        tr = {t: b.split(b.track(t)) for t in (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH, L.T_OCC_PREFILL,
                                               L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)}
        kw = dict(min_signal=0, sep=25, boundary=w, order=10)
        cc, cp, lr, var, z = b.run_peaks(**kw)
        for k, Lc in enumerate(lens):
            l, n = fr[k]
            try:
                nt = O.nuc_chunk_tracks(l, n, 0, Lc, pk.chunk_bias(k), -BL, vm, vlo, vup, sizes[:vup], smooth_sd=sd)
                assert_track(tr[L.T_NUC_COV][k], nt["nuc_cov"], "nuc_cov", exact=True)
                assert_track(tr[L.T_NFR_COV][k], nt["nfr_cov"], "nfr_cov", exact=True)
                assert_track(tr[L.T_RAW][k], nt["raw"], "raw")
                assert_track(tr[L.T_BACKGROUND][k], nt["bg"], "bg")
                assert_track(tr[L.T_NORM][k], nt["norm"], "norm", scale=cancel_scale(nt["raw"], nt["bg"]))
                assert_track(tr[L.T_SMOOTH][k], nt["smoothed"], "smoothed", scale=cancel_scale(nt["raw"], nt["bg"]))
                if not (st[k] & 1):
                    oc = O.occ_chunk_tracks(l, n, 0, Lc, pk.chunk_bias(k), -BL, nucp, nfrp, upper=occ_up, flank=flank, step=step,
                                            cutoff=cutoff, n_alpha=n_alpha)
                    assert_track(tr[L.T_OCC_PREFILL][k], oc["smoothed_vals"], "occ")
                    assert_track(tr[L.T_OCC_LOWER][k], oc["smoothed_lower"], "occ lower")
                    assert_track(tr[L.T_OCC_UPPER][k], oc["smoothed_upper"], "occ upper")
                    assert_track(tr[L.T_OCC_COV][k], oc["cov"], "occ cov", exact=True)
                if not (st[k] & 2):
                    hp = np.asarray(O.call_peaks((tr[L.T_NORM][k] + tr[L.T_SMOOTH][k]).copy(), **kw), np.int64)
                    mine = cp[cc == k]
                    assert np.array_equal(mine, hp), "peaks"
                    for pos, lrv, varv in list(zip(mine, lr[cc == k], var[cc == k]))[:3]:
                        ref_lr = O.get_lr(nt["mat"], nt["mat_start"], nt["bmat"], nt["b0"], nt["b_start"], vm, vlo, vup, int(pos))
                        if np.isnan(ref_lr) or np.isinf(ref_lr):
                            assert np.isnan(lrv) or np.isinf(lrv)
                        else:
                            assert abs(lrv - ref_lr) <= 1e-10 * max(1.0, abs(ref_lr)), ("lr", lrv, ref_lr)
                        pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], vlo, vup, w, int(pos))
                        ref_var = O.calculate_cov_closed(pr, np.ravel(vm), nt["nuc_cov"][pos])
                        assert abs(varv - ref_var) <= 1e-9 * max(1e-12, abs(ref_var)), ("var", varv, ref_var)
            except AssertionError as e:
                raise AssertionError("%s | chunk %d | %r" % (e, k, desc))
        b.free()
    return sum(lens)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    par = golden("params_example")
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bp = 0
    done = 0
    for r in range(rounds):
        bp += one_round(rng, par)
        done += 1
        if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "1e9")):
            break
    print("generic fuzz ok: %d rounds, %d bases, %.0f s" % (done, bp, time.time() - t0))


if __name__ == "__main__":
    main()
