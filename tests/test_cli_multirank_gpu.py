"""GPU: the sharded `occ` / `nuc` drivers (chunk list split over ranks, per-rank part files concatenated in chunk order,
nuc_dist summed in chunk order) give byte-identical outputs with 1 and 2 ranks.  2 ranks = gloo + both on GPU 0."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import GOLDEN, golden, read_bed3, synth_saccer3, synth_stores

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(tmp_path):
    from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
    par = golden("params_example")
    frags, fasta = synth_stores(11)
    bam = str(tmp_path / "synth.npz")
    frags.save_npz(bam)
    fa = str(tmp_path / "synth.fa")
    with open(fa, "w") as f:
        f.write(">chrS\n" + fasta.seqs["chrS"].tobytes().decode() + "\n")
    bed = str(tmp_path / "r.bed")
    with open(bed, "w") as f:
        for s in range(1200, 12000, 1500):
            f.write("chrS\t%d\t%d\n" % (s, s + 1100))
    sizes = str(tmp_path / "sizes.txt")
    FragmentSizes(0, 251, vals=par["sizes"]).save(sizes)
    vm = str(tmp_path / "v.npz")
    np.savez(vm, vmat=par["vmat"], vlower=par["vlower"], vupper=par["vupper"])
    return bed, bam, fa, sizes, vm


def _run(world, out, bed, bam, fa, sizes, vm):
    common = ["--bed", bed, "--bam", bam, "--fasta", fa, "--sizes", sizes, "--out", out]
    for sub in (["occ"] + common, ["nuc"] + common + ["--vmat", vm, "--write_all"]):
        if world == 1:
            cmd = [sys.executable, "-m", "nucleoatac_amd.nucleoatac.cli"] + sub
            env = dict(os.environ)
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                   "127.0.0.1", "--master-port", "29541", "-m", "nucleoatac_amd.nucleoatac.cli"] + sub
            env = dict(os.environ, NATAC_DIST_BACKEND="gloo", NATAC_DEVICE="0")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]


def test_two_ranks_equal_one_rank(tmp_path):
    bed, bam, fa, sizes, vm = _inputs(tmp_path)
    _run(1, str(tmp_path / "one"), bed, bam, fa, sizes, vm)
    _run(2, str(tmp_path / "two"), bed, bam, fa, sizes, vm)
    names = ["occ.bedgraph.gz", "occ.lower_bound.bedgraph.gz", "occ.upper_bound.bedgraph.gz", "occpeaks.bed.gz",
             "nucleoatac_signal.bedgraph.gz", "nucleoatac_signal.smooth.bedgraph.gz", "nucleoatac_raw.bedgraph.gz",
             "nucleoatac_background.bedgraph.gz", "nucpos.bed.gz", "nucpos.redundant.bed.gz"]
    for n in names:
        a = gzip.open(str(tmp_path / "one") + "." + n, "rt").read()
        b = gzip.open(str(tmp_path / "two") + "." + n, "rt").read()
        assert a == b and (len(a) > 0 or "redundant" in n), n
    assert open(str(tmp_path / "one") + ".nuc_dist.txt").read() == open(str(tmp_path / "two") + ".nuc_dist.txt").read()
    left = [f for f in os.listdir(str(tmp_path)) if ".rank" in f]
    assert not left, left


def test_two_ranks_from_a_real_bam_equal_one_rank_from_the_stand_in(tmp_path):
    """`occ` + `nuc` under two ranks from a REAL .bam: rank 0 decodes it on the GPU (natac_bam_open_device) and shares the arrays
    through /dev/shm, both ranks take their part of the chunk list -- the tracks equal the one-rank run from the .npz stand-in"""
    from nucleoatac_amd.synth import cli_dataset_as_real_files
    bed, bam, fa, sizes, vm = _inputs(tmp_path)
    fa_npz = str(tmp_path / "fa.npz")
    seq = open(fa).read().split("\n")[1]
    np.savez(fa_npz, chrom_names=np.array(["chrS"]), **{"seq_chrS": np.frombuffer(seq.encode(), dtype=np.uint8)})
    real_bam, _ = cli_dataset_as_real_files(bam, fa_npz, str(tmp_path))
    _run(1, str(tmp_path / "one"), bed, bam, fa, sizes, vm)
    _run(2, str(tmp_path / "two"), bed, real_bam, fa, sizes, vm)
    for n in ("occ.bedgraph.gz", "occ.upper_bound.bedgraph.gz", "nucleoatac_signal.bedgraph.gz", "nucpos.bed.gz"):
        a = gzip.open(str(tmp_path / "one") + "." + n, "rt").read()
        b = gzip.open(str(tmp_path / "two") + "." + n, "rt").read()
        assert a == b and len(a) > 0, n
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("natac_frags_")]


def test_nuc_cores_pool_equals_serial(tmp_path):
    """--cores N farms the per-nucleosome L-BFGS fits out to spawned host processes: outputs identical to --cores 1"""
    bed, bam, fa, sizes, vm = _inputs(tmp_path)
    outs = []
    for cores in (1, 3):
        out = str(tmp_path / ("c%d" % cores))
        cmd = [sys.executable, "-m", "nucleoatac_amd.nucleoatac.cli", "nuc", "--bed", bed, "--bam", bam, "--fasta", fa, "--sizes",
               sizes, "--out", out, "--vmat", vm, "--cores", str(cores)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(gzip.open(out + ".nucpos.bed.gz", "rt").read())
    assert outs[0] == outs[1] and len(outs[0]) > 0


@pytest.mark.parametrize("ranks", [2, 8])
def test_run_n_ranks_equal_one_rank(tmp_path, ranks):
    """`nucleoatac run` (all five steps chained through their files) under torchrun with 2 and with 8 ranks (all on GPU 0): occ,
    nuc and nfr shard the chunk list, the BAM is decoded once and shared, vprocess / merge run on rank 0 behind an ok / failed
    flag; every output equals the single-process run byte for byte"""
    bed = os.path.join(GOLDEN, "ref_example.bed")
    bam, fa = synth_saccer3(str(tmp_path), read_bed3(bed), seed=3)
    outs = {}
    for world in (1, ranks):
        out = str(tmp_path / ("run%d" % world))
        sub = ["run", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--write_all", "--cores", "4"]
        if world == 1:
            cmd = [sys.executable, "-m", "nucleoatac_amd.nucleoatac.cli"] + sub
            env = dict(os.environ)
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                   "127.0.0.1", "--master-port", str(29543 + world), "-m", "nucleoatac_amd.nucleoatac.cli"] + sub
            env = dict(os.environ, NATAC_DIST_BACKEND="gloo", NATAC_DEVICE="0")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[world] = out
    for n in ("occ.bedgraph.gz", "occ.lower_bound.bedgraph.gz", "occ.upper_bound.bedgraph.gz", "occpeaks.bed.gz", "nucpos.bed.gz",
              "nucpos.redundant.bed.gz", "nucleoatac_signal.bedgraph.gz", "nucleoatac_signal.smooth.bedgraph.gz",
              "nucleoatac_raw.bedgraph.gz", "nucleoatac_background.bedgraph.gz", "nucmap_combined.bed.gz", "nfrpos.bed.gz",
              "ins.bedgraph.gz"):
        a = gzip.open(outs[1] + "." + n, "rt").read()
        b = gzip.open(outs[ranks] + "." + n, "rt").read()
        assert a == b, n
    for n in ("nuc_dist.txt", "fragmentsizes.txt", "VMat"):
        assert open(outs[1] + "." + n).read() == open(outs[ranks] + "." + n).read(), n
    assert len(gzip.open(outs[1] + ".nucpos.bed.gz", "rt").read()) > 0
    assert not [f for f in os.listdir(str(tmp_path)) if ".rank" in f]
    if ranks == 2:
        # one rank writes nfr's insertion track on the device and builds its .tbi from the device's records: the same index
        # natac_tabix_index builds by reading the finished file
        import shutil
        from nucleoatac_amd.writer import tabix_index
        copy = str(tmp_path / "ins_copy.bedgraph.gz")
        shutil.copy(outs[1] + ".ins.bedgraph.gz", copy)
        tabix_index(copy)
        assert open(copy + ".tbi", "rb").read() == open(outs[1] + ".ins.bedgraph.gz.tbi", "rb").read()


def test_rank0_failure_ends_every_rank(tmp_path):
    """a single-rank step that fails on rank 0 (merge of a missing file) must end ALL ranks promptly -- the others get rank 0's
    failed flag instead of waiting in a barrier until the process group times out"""
    import time
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", "-m", "nucleoatac_amd.nucleoatac.cli", "merge", "--occpeaks", str(tmp_path / "missing.bed.gz"),
           "--nucpos", str(tmp_path / "missing2.bed.gz"), "--out", str(tmp_path / "m")]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, NATAC_DIST_BACKEND="gloo", NATAC_DEVICE="0"),
                       cwd=ROOT)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "rank 0 failed" in r.stderr or "No such file" in r.stderr or "Error" in r.stderr


def test_device_writer_and_host_writer_give_the_same_files(tmp_path):
    """`occ` + `nuc` with the tracks formatted / compressed / indexed on the GPU (default) and with the native host writer
    (NATAC_DEVICE_WRITER=0): the text inside every .bedgraph.gz is identical, both sets of .tbi answer region reads identically"""
    from nucleoatac_amd.pyatac.tracks import Track
    bed, bam, fa, sizes, vm = _inputs(tmp_path)
    outs = {}
    for mode in ("1", "0"):
        out = str(tmp_path / ("w" + mode))
        common = ["--bed", bed, "--bam", bam, "--fasta", fa, "--sizes", sizes, "--out", out]
        for sub in (["occ"] + common, ["nuc"] + common + ["--vmat", vm, "--write_all"]):
            r = subprocess.run([sys.executable, "-m", "nucleoatac_amd.nucleoatac.cli"] + sub, capture_output=True, text=True, timeout=900,
                               env=dict(os.environ, NATAC_DEVICE_WRITER=mode, NATAC_BATCH_CHUNKS="3"), cwd=ROOT)   # several sub-batches
            assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = out
    for n in ("occ", "occ.lower_bound", "occ.upper_bound", "nucleoatac_signal", "nucleoatac_signal.smooth", "nucleoatac_raw",
              "nucleoatac_background"):
        a, b = outs["1"] + "." + n + ".bedgraph.gz", outs["0"] + "." + n + ".bedgraph.gz"
        ta, tb = gzip.open(a, "rt").read(), gzip.open(b, "rt").read()
        assert ta == tb and len(ta) > 1000, n
        assert os.path.getsize(a) < 1.05 * os.path.getsize(b), n          # the device's members are not larger than zlib level 4's
        for s in (1300, 5600, 10300):
            x, y = Track("chrS", s, s + 700), Track("chrS", s, s + 700)
            x.read_track(a)
            y.read_track(b)
            assert np.array_equal(x.vals, y.vals, equal_nan=True), (n, s)
    for n in ("occpeaks.bed.gz", "nucpos.bed.gz"):
        assert gzip.open(outs["1"] + "." + n, "rt").read() == gzip.open(outs["0"] + "." + n, "rt").read()
