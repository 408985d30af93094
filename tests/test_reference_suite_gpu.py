"""GPU: the reference's own unit tests (tests/test_*.py of GreenleafLab/NucleoATAC), re-expressed against the
nucleoatac_amd host API.  Every numeric call goes through libnatac_hip.so.  Fixtures are the reference's data
files (tests/golden/ref_*); where the reference needs the absent example.bam a seeded synthetic store is used."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, golden, synth_stores

pytestmark = pytest.mark.gpu

BED = os.path.join(GOLDEN, "ref_example.bed")
SCORES = os.path.join(GOLDEN, "ref_example.Scores.bedgraph.gz")


@pytest.fixture(scope="module")
def first_chunk():
    from nucleoatac_amd.pyatac.chunk import ChunkList
    return ChunkList.read(BED)[0]


# ---- tests/test_var.py ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def signaldist(first_chunk):
    import nucleoatac_amd.nucleoatac.NucleosomeCalling as Nuc
    from nucleoatac_amd.pyatac.bias import InsertionBiasTrack
    from nucleoatac_amd.pyatac.chunkmat2d import BiasMat2D
    from nucleoatac_amd.pyatac.VMat import VMat
    chunk = first_chunk
    vmat = VMat.open(os.path.join(GOLDEN, "ref_example.VMat"))
    biastrack = InsertionBiasTrack(chunk.chrom, chunk.start, chunk.end)
    biastrack.read_track(SCORES)
    biasmat = BiasMat2D(chunk.chrom, chunk.start + 200, chunk.end - 200, 100, 250)
    biasmat.makeBiasMat(biastrack)
    return Nuc.SignalDistribution(chunk.start + 300, vmat, biasmat, 35)


def test_sd1(signaldist):
    """variance calculation is close to what is obtained by simulation"""
    np.random.seed(1)
    signaldist.simulateDist(5000)
    sd1 = np.std(signaldist.scores)
    sd2 = signaldist.analStd()
    assert abs(sd1 - sd2) < 0.05 * sd1


def test_sd2(signaldist):
    """variance calculation equals the alternate (outer product) calculation"""
    var_term = np.sum(signaldist.prob_mat * (1 - signaldist.prob_mat) * signaldist.vmat.mat ** 2)
    tmp = signaldist.prob_mat * signaldist.vmat.mat
    cov_term = np.sum(np.outer(tmp, tmp)) - np.sum(tmp ** 2)
    sd1 = np.sqrt(signaldist.reads * (var_term - cov_term))
    sd2 = signaldist.analStd()
    assert abs(sd1 - sd2) < 0.001 * sd1
    g = golden("cov_var_example")
    assert abs(sd2 ** 2 - float(g["var"])) < 1e-5 * float(g["var"])   # the reference's own calculateCov value


# ---- tests/test_chunkmat2d.py --------------------------------------------------------------------
@pytest.fixture(scope="module")
def biasmat(first_chunk):
    from nucleoatac_amd.pyatac.bias import InsertionBiasTrack
    from nucleoatac_amd.pyatac.chunkmat2d import BiasMat2D
    chunk = first_chunk
    bt = InsertionBiasTrack(chunk.chrom, chunk.start, chunk.end)
    bt.read_track(SCORES)
    bm = BiasMat2D(chunk.chrom, chunk.start + 100, chunk.end - 100, 100, 200)
    bm.makeBiasMat(bt)
    return bt, bm


def test_biasmat1(biasmat):
    bt, bm = biasmat
    correct = np.exp(bt.get(pos=bm.start - 49) + bt.get(pos=bm.start + 50))
    assert abs(correct - bm.mat[0, 0]) < 0.01 * correct


def test_biasmat2(biasmat):
    bt, bm = biasmat
    correct = np.exp(bt.get(pos=bm.start + 145) + bt.get(pos=bm.start + 295))
    assert abs(correct - bm.mat[51, 220]) < 0.01 * correct


def test_normByInsertDist(biasmat):
    from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
    bt, bm = biasmat
    isizes = FragmentSizes(lower=100, upper=200, vals=np.array(range(100, 200)))
    bm.normByInsertDist(isizes)
    correct = np.exp(bt.get(pos=bm.start - 50) + bt.get(pos=bm.start + 50)) * isizes.get(size=101)
    assert abs(correct - bm.mat[1, 0]) < 0.01 * correct


def test_biasmat_matches_reference_values():
    """rows of the reference's BiasMat2D on the example scores (golden cov_var_example)"""
    from nucleoatac_amd import get_context
    g = golden("cov_var_example")
    bm = get_context().make_bias_mat(g["bias_track"], int(g["bias_track_start"]), int(g["biasmat_start"]),
                                     int(g["biasmat_end"]), int(g["biasmat_lower"]), int(g["biasmat_upper"]))
    np.testing.assert_allclose(bm[g["biasmat_sample_rows"]], g["biasmat_samples"], rtol=1e-12)
    with pytest.raises(Exception):
        get_context().make_bias_mat(g["bias_track"][:100], int(g["bias_track_start"]), int(g["biasmat_start"]),
                                    int(g["biasmat_end"]), 100, 250)


# ---- tests/test_occupancy.py ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def toy_params():
    from nucleoatac_amd.nucleoatac.Occupancy import FragmentMixDistribution, OccupancyCalcParams
    from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
    fd = FragmentMixDistribution(0, 3)
    fd.nfr_fit = FragmentSizes(0, 3, vals=np.array([0.5, 0.49, 0.01]))
    fd.nuc_fit = FragmentSizes(0, 3, vals=np.array([0.01, 0.49, 0.5]))
    return fd, OccupancyCalcParams(0, 3, fd)


def test_occupancy_calc1(toy_params):
    from nucleoatac_amd.nucleoatac.Occupancy import calculateOccupancy
    assert calculateOccupancy(np.array([1, 0, 0]), np.array([1, 1, 1]), toy_params[1])[0] == 0


def test_occupancy_calc2(toy_params):
    from nucleoatac_amd.nucleoatac.Occupancy import calculateOccupancy
    assert calculateOccupancy(np.array([1, 1, 1]), np.array([1, 1, 1]), toy_params[1])[0] == 0.5


def test_occupancy_calc3_and_4(toy_params):
    from nucleoatac_amd.nucleoatac.Occupancy import calculateOccupancy
    fd, params = toy_params
    rng = np.random.RandomState(3)
    for bias in (np.array([1, 1, 1]), np.array([3, 2, 1])):
        nfrprob = fd.nfr_fit.get() * bias
        nucprob = fd.nuc_fit.get() * bias
        nfrprob, nucprob = nfrprob / nfrprob.sum(), nucprob / nucprob.sum()
        res = np.array([calculateOccupancy(rng.multinomial(10, nfrprob) + rng.multinomial(30, nucprob), bias, params)
                        for _ in range(100)])
        assert abs(np.mean(res[:, 0]) - 0.75) < 0.1
        assert np.sum(res[:, 2] < 0.75) < 85 and np.sum(res[:, 1] > 0.75) < 85


def test_occupancy_golden_draws(toy_params):
    """bit-exact against the reference on the 42 stored windows"""
    from nucleoatac_amd.nucleoatac.Occupancy import calculateOccupancy
    g = golden("toy_occupancy")
    for ins, bias, ref in zip(g["ins"], g["bias"], g["result"]):
        assert tuple(calculateOccupancy(ins, bias, toy_params[1])) == tuple(ref)


# ---- tests/test_tracks.py ------------------------------------------------------------------------
def test_ins_methods(first_chunk):
    """two methods for getting the insertion track give the same result (reference fixture single_read.bam)"""
    from nucleoatac_amd.pyatac.chunkmat2d import FragmentMat2D
    from nucleoatac_amd.pyatac.tracks import InsertionTrack
    chunk = first_chunk
    bam = os.path.join(GOLDEN, "ref_single_read.bam")
    ins1 = InsertionTrack(chunk.chrom, chunk.start, chunk.end)
    ins1.calculateInsertions(bam)
    mat = FragmentMat2D(chunk.chrom, chunk.start, chunk.end, 0, 100)
    mat.makeFragmentMat(bam)
    ins2 = mat.getIns()
    a = ins1.get(chunk.start + 100, chunk.start + 300)
    assert np.array_equal(a, ins2.get(chunk.start + 100, chunk.start + 300)) and a.sum() == 1
    sr = golden("single_read")
    assert np.array_equal(ins1.vals, sr["ins"])


# ---- tests/test_xcor.py (example.bam is absent from the reference checkout: synthetic fragments) -----------------
def test_signal_calc():
    import nucleoatac_amd.nucleoatac.NucleosomeCalling as Nuc
    from nucleoatac_amd.pyatac.chunk import Chunk
    from nucleoatac_amd.pyatac.chunkmat2d import FragmentMat2D
    from nucleoatac_amd.pyatac.VMat import VMat
    frags, _ = synth_stores(11)
    chunk = Chunk("chrS", 3000, 4100)
    vmat = VMat.open(os.path.join(GOLDEN, "ref_example.VMat"))
    mat = FragmentMat2D(chunk.chrom, chunk.start - vmat.w, chunk.end + vmat.w, vmat.lower, vmat.upper)
    mat.makeFragmentMat(frags)
    assert mat.mat.sum() > 100
    sig = Nuc.SignalTrack(chunk.chrom, chunk.start, chunk.end)
    sig.calculateSignal(mat, vmat)
    for off in (0, 100):
        a = np.sum(mat.get(start=chunk.start + off, end=chunk.start + off + vmat.w * 2 + 1) * vmat.mat)
        b = sig.get(pos=chunk.start + off + vmat.w)
        assert abs(a - b) < 0.0001


# ---- utils.smooth on the GPU vs numpy semantics --------------------------------------------------------------
def test_smooth_matches_reference_semantics():
    from nucleoatac_amd.pyatac.utils import smooth
    from oracle import natac_oracle as O
    rng = np.random.default_rng(2)
    x = rng.normal(size=500)
    x[100:180] = np.nan
    for kw in (dict(window_len=61, window="gaussian", sd=10, mode="same", norm=True),
               dict(window_len=121, window="flat", mode="valid", norm=False),
               dict(window_len=121, window="gaussian", sd=20, mode="same", norm=True),
               dict(window_len=15, window="flat", mode="valid", norm=True)):
        got, ref = smooth(x.copy(), **kw), O.smooth(x.copy(), **kw)
        assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_allclose(got[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-12, atol=1e-13)
    with pytest.warns(UserWarning):
        assert smooth(np.ones(20), 4, mode="valid", norm=False).shape == (16,)
    with pytest.raises(Exception):
        smooth(np.ones(20), 5, window="hann")
