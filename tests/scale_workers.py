"""worker functions of the scale-parity tests (tests only): one chunk through the CPU oracle.  Kept in an importable
module so that a `spawn` multiprocessing pool can run them (the pytest process already holds a HIP context)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def occ_grid_worker(args):
    """raw occupancy grid (vals / lower_bound / upper_bound at the grid points, Occupancy.py:128-146) of one chunk"""
    from oracle import natac_oracle as O
    l, n, L, bias, bias_left, nucp, nfrp = args
    oc = O.occ_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, L, bias, -bias_left, nucp, nfrp)
    return oc["occ"][2::5], oc["occ_lower"][2::5], oc["occ_upper"][2::5]


def cand_worker(args):
    """oracle candidates + (lr, var, z) at the given positions of one chunk (NucleosomeCalling.py:110-127, 294-315)"""
    from oracle import natac_oracle as O
    l, n, L, bias, bias_left, vmat, vlo, vup, sizes, positions = args
    nt = O.nuc_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, L, bias, -bias_left, vmat, vlo, vup, sizes)
    comb = nt["norm"] + nt["smoothed"]
    ref_c = O.call_peaks(comb.copy(), min_signal=0, sep=25, boundary=60, order=12)
    ref_c = np.array([int(i) for i in ref_c], dtype=np.int64)
    out = []
    w = vmat.shape[1] // 2
    for p in positions:
        p = int(p)
        lr = O.get_lr(nt["mat"], nt["mat_start"], nt["bmat"], nt["b0"], nt["b_start"], vmat, vlo, vup, p)
        pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], vlo, vup, w, p)
        z, var = O.z_score(nt["norm"][p], nt["nuc_cov"][p], pr, vmat)
        out.append((lr, var, z, nt["nuc_cov"][p], comb[p]))
    return ref_c, comb[ref_c] if len(ref_c) else np.zeros(0), np.array(out, dtype=np.float64).reshape(-1, 5)


def tracks_worker(args):
    """every per-base track the oracle produces for one chunk (nuc + occ + insertions)"""
    from oracle import natac_oracle as O
    l, n, L, bias, bias_left, vmat, vlo, vup, sizes, nucp, nfrp = args
    l, n = l.astype(np.int64), n.astype(np.int64)
    nt = O.nuc_chunk_tracks(l, n, 0, L, bias, -bias_left, vmat, vlo, vup, sizes)
    oc = O.occ_chunk_tracks(l, n, 0, L, bias, -bias_left, nucp, nfrp)
    return dict(nuc_cov=nt["nuc_cov"], nfr_cov=nt["nfr_cov"], raw=nt["raw"], bg=nt["bg"], norm=nt["norm"], smoothed=nt["smoothed"],
                occ=oc["smoothed_vals"], occ_lower=oc["smoothed_lower"], occ_upper=oc["smoothed_upper"], occ_cov=oc["cov"],
                ins=O.get_insertions(l, n, 0, L).astype(np.int32))


def cov_literal_worker(args):
    """the oracle's literal calculateCov (C restatement of multinomial_cov.pyx:20-31) + closed form at given candidates of a chunk"""
    from oracle import natac_oracle as O
    l, n, L, bias, bias_left, vmat, vlo, vup, sizes, positions = args
    nt = O.nuc_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, L, bias, -bias_left, vmat, vlo, vup, sizes)
    w = vmat.shape[1] // 2
    v = np.ravel(vmat)
    out = []
    for p in positions:
        pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], vlo, vup, w, int(p))
        r = nt["nuc_cov"][int(p)]
        out.append((O.calculate_cov_literal(pr, v, r), O.calculate_cov_closed(pr, v, r)))
    return np.array(out, dtype=np.float64).reshape(-1, 2)
