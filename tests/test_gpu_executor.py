"""GPU: nucleoatac_amd.executor -- the pipelined multi-context executor the CLI drivers and bench.py's host_to_host run on.
Results in input order and identical to a plain single-context pass, errors of a worker / of the packing iterator surface on
the consumer's thread, an abandoned map() leaves the executor usable."""
import numpy as np
import pytest

from helpers import golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.executor import PipelinedExecutor, Stages
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu


def _configure(ctx):
    par = golden("params_example")
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(synth_size_distribution(251))
    nucp, nfrp = synth_occ_distributions(251)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)


def _subs(n, seed=0):
    out = []
    for i in range(n):
        pk = make_synthetic_chunks(40 + 7 * (i % 3), 900 + 100 * (i % 4), 200, seed=seed + i, first_chunk=100 * i)
        pk.chroms = ["chr%d" % (1 + i % 3)] * pk.n_chunks
        out.append(pk)
    return out


def test_results_in_order_and_equal_to_a_single_context():
    from nucleoatac_amd.device import Context
    subs = _subs(11)
    stages = Stages(nuc_sd=10, occ=True, ins=(0, 2000), peaks=dict(min_signal=0, sep=25, boundary=60, order=12),
                    occ_peaks=dict(min_occ=0.1, sep=120), tracks=(L.T_NORM, L.T_OCC, L.T_INS), text_tracks=(L.T_OCC_UPPER,))
    ref = []
    with Context(0) as ctx:
        _configure(ctx)
        for pk in subs:
            b = ctx.upload(pk)
            n = stages.run(b)
            ref.append(dict(norm=b.track(L.T_NORM), occ=b.track(L.T_OCC), ins=b.track(L.T_INS), peaks=b.download_peaks(n),
                            z=b.format_track(L.T_OCC_UPPER, pk.chroms, pk.chunk_start, compress=True)[0].tobytes(),
                            occ_peaks=b.run_occ_peaks(min_occ=0.1, sep=120)))
            b.free()
    with PipelinedExecutor(0, _configure, stages, n_contexts=4, slots_per_context=2) as ex:
        for rep in range(2):                                   # contexts and slots are reused by the second map()
            seen = []
            for r in ex.map((pk, i) for i, pk in enumerate(subs)):
                assert r.tag == r.seq == len(seen)
                w = ref[r.seq]
                assert np.array_equal(r.tracks[L.T_NORM], w["norm"]) and np.array_equal(r.tracks[L.T_OCC], w["occ"], equal_nan=True)
                assert np.array_equal(r.tracks[L.T_INS], w["ins"])
                for a, b_ in zip(r.peaks, w["peaks"]):
                    assert np.array_equal(a, b_, equal_nan=True)
                for a, b_ in zip(r.occ_peaks, w["occ_peaks"]):
                    assert np.array_equal(a, b_, equal_nan=True)
                assert r.text[L.T_OCC_UPPER].tobytes() == w["z"] and len(r.text_index[L.T_OCC_UPPER]["cid"]) > 0
                seen.append(r.seq)
                r.release()
            assert seen == list(range(len(subs)))
        assert ex.bytes_down > 0 and ex.bytes_up > 0


def test_errors_surface_on_the_consumer_thread():
    subs = _subs(6)
    stages = Stages(nuc_sd=10, occ=True, ins=None)
    bad = make_synthetic_chunks(3, 100, 20, seed=1)              # shorter than the 121-base occupancy window: natac_run_occ refuses it
    with PipelinedExecutor(0, _configure, stages, n_contexts=3) as ex:
        with pytest.raises(L.NatacError):
            for r in ex.map((pk, i) for i, pk in enumerate(subs[:3] + [bad] + subs[3:])):
                r.release()

        def packing():                                            # an exception while producing the next sub-batch
            yield subs[0], 0
            raise KeyError("chunk on an unknown chromosome")
        with pytest.raises(KeyError):
            for r in ex.map(packing()):
                r.release()
        # a consumer that stops early; the executor stays usable
        for r in ex.map((pk, i) for i, pk in enumerate(subs)):
            r.release()
            break
        n = 0
        for r in ex.map((pk, i) for i, pk in enumerate(subs)):
            assert r.seq == n
            n += 1
            r.release()
        assert n == len(subs)


def test_writer_mixes_device_members_and_host_written_batches(tmp_path):
    """a sub-batch whose track holds a value the device formatter cannot decide (n_hard > 0) is formatted by the native host
    writer instead; its members are appended between the device's, the file stays one valid BGZF stream with the same text, and
    the index falls back to the file-based indexer (the device's record log no longer covers the whole file)"""
    import gzip
    import types
    from nucleoatac_amd.nucleoatac.run_occ import _Writer, finish_indexes
    from nucleoatac_amd.pyatac.chunk import Chunk
    from nucleoatac_amd.pyatac.tracks import Track
    from nucleoatac_amd.writer import tabix_index, write_bedgraph
    subs = _subs(5)
    for i, pk in enumerate(subs):                  # one sorted file: chromosome blocks contiguous, positions ascending
        pk.chroms = ["chr%d" % (1 + i // 2)] * pk.n_chunks
    stages = Stages(nuc_sd=None, occ=True, ins=None, tracks=(L.T_OCC,), text_tracks=(L.T_OCC,))
    path = str(tmp_path / "mixed.bedgraph.gz")
    ref_path = str(tmp_path / "host.bedgraph.gz")
    parts = [[Chunk(c, int(s), int(s) + int(n)) for c, s, n in zip(pk.chroms, pk.chunk_start, pk.chunk_len)] for pk in subs]
    w = _Writer({"occ": path}, {"occ": L.T_OCC}, lambda r: None, len(subs), True)
    w.start()
    with PipelinedExecutor(0, _configure, stages, n_contexts=2) as ex:
        for r in ex.map((pk, part) for pk, part in zip(subs, parts)):
            write_bedgraph(ref_path, r.packed.chroms, r.packed.chunk_start, r.packed.out_off, r.tracks[L.T_OCC].copy(),
                           append=r.seq > 0, compress=4, finish=r.seq == len(subs) - 1)
            if r.seq in (1, 3):                    # what executor._process does when info["hard"] > 0
                r.text[L.T_OCC] = None
            w.put(r)
    w.finish()
    assert not w.index_ok["occ"] and len(w.index_log["occ"]) == 3
    ws = types.SimpleNamespace(index_log=w.index_log, index_ok=w.index_ok, offset=w.offset)
    assert finish_indexes(ws, ["occ"], lambda n: path) == [path]
    tabix_index(path)
    tabix_index(ref_path)
    assert gzip.open(path, "rt").read() == gzip.open(ref_path, "rt").read()
    pk = subs[3]
    a, b = Track(pk.chroms[5], int(pk.chunk_start[5]), int(pk.chunk_start[5]) + 300), Track(pk.chroms[5], int(pk.chunk_start[5]), int(pk.chunk_start[5]) + 300)
    a.read_track(path)
    b.read_track(ref_path)
    assert np.array_equal(a.vals, b.vals, equal_nan=True) and np.isfinite(a.vals).any()
