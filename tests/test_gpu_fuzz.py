"""GPU: a short run of tests/fuzz/fuzz_parity.py -- random ragged chunk sets (lengths 121..9000, densities 0..12 fragments per base,
fragment-free stretches, with / without bias, sizes at the model's edges): every track, insertion counts and the candidate
search against the CPU oracle.  (The release check ran ~2,400 rounds / 10 M bases of the same generator.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_chunk_sets_match_the_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_parity as F
    from helpers import golden
    from nucleoatac_amd.device import Context
    from nucleoatac_amd.synth import synth_occ_distributions, synth_size_distribution
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    ctx = Context(0)
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(sizes)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    rng = np.random.default_rng(2024)
    bp = 0
    for r in range(30):
        bp += F.one_round(ctx, rng, par, sizes, nucp, nfrp, r)[0]
    ctx.close()
    assert bp > 30000


def test_random_model_geometries_match_the_oracle():
    """a short run of tests/fuzz/fuzz_generic.py: random V-plot bounds / row counts / widths (incl. narrower than a wave: the paired
    candidate kernel must hand over), smoothing widths, occupancy window / step / size range / alpha grids"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_generic as G
    from helpers import golden
    par = golden("params_example")
    rng = np.random.default_rng(99)
    bp = sum(G.one_round(rng, par) for _ in range(25))
    assert bp > 10000


def test_random_dropin_calls_match_the_oracle():
    """a short run of tests/fuzz/fuzz_dropins.py: the Cython functions' replacements and the operator-level helpers on random
    regions / fragment sets / chunk lists / windows / PWMs"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_dropins as D
    from nucleoatac_amd.device import Context
    from nucleoatac_amd.synth import synth_occ_distributions
    D.one_round.model = synth_occ_distributions(251)
    rng = np.random.default_rng(5)
    with Context(0) as c:
        c.set_occ_model(*D.one_round.model, step=5, flank=60)
        for _ in range(150):
            D.one_round(c, rng)


def test_writer_fuzz_slice():
    """a short run of tests/fuzz/fuzz_writer.py: the device track writer on random geometries, names and tracks (runs, zero runs,
    NaN stretches, values of every scale) in every flag combination -- text, members and index against the host side"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_writer
    done, lines = fuzz_writer.run(12, 21)
    assert done == 12 and lines > 100000


def test_bam_fuzz_slice():
    """a short run of tests/fuzz/fuzz_bam.py: the device BAM decoder against the host decoder on random files"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_bam
    done, on_device = fuzz_bam.run(15, 8)
    assert done == 15 and on_device == 15


def test_round4_paths_fuzz_slice():
    """a short run of tests/fuzz/fuzz_round4.py: co-scheduled stages == serial stages, resident store == the parse of the device's own
    text, on random ragged batches (incl. non-finite bias stretches)"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_round4.py"), "12", "7"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "fuzz_round4 ok: 12 rounds" in out.stdout
