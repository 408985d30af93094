"""nucleoatac_amd -- NucleoATAC's occ + nuc signal path on AMD MI355X (gfx950).

Sub-packages mirror the reference's module names for the hot path:
  nucleoatac_amd.pyatac      chunk, tracks, chunkmat2d, fragments, fragmentsizes, bias, seq, VMat, utils
  nucleoatac_amd.nucleoatac  Occupancy, NucleosomeCalling, run_occ, run_nuc, cli
  nucleoatac_amd.executor    how sub-batches move through one GPU: ResidentShard, PipelinedExecutor (the CLI drivers and bench.py)
  nucleoatac_amd.shard       chunk-list sharding over the GPUs of a node, control-plane group, shared BAM decode
All numerics run in libnatac_hip.so (include/natac.h); there is no CPU fallback.
"""
import os
import threading

__version__ = "0.1.0"

_default_ctx = None
# The process-wide context is ONE stream with per-context model state (V-plot, sizes, occupancy model): work on it from more than
# one thread (a driver's writer thread re-doing an overflow chunk while the main thread packs, a reader pulling resident tracks)
# takes this lock for the whole install-constants + run + download sequence.  The executor's worker contexts are private to their
# threads and need none.
context_lock = threading.RLock()


def get_context():
    """process-wide natac context on GPU `LOCAL_RANK` (0 when unset); created on first use"""
    global _default_ctx
    with context_lock:
        if _default_ctx is None:
            _default_ctx = _make_default()
    return _default_ctx


def _make_default():
    from .device import Context
    # one context per process: GPU = LOCAL_RANK (torchrun); NATAC_DEVICE overrides it (e.g. several ranks on one GPU in tests)
    return Context(default_device())


def default_device():
    """ordinal of the GPU this process computes on: NATAC_DEVICE, else LOCAL_RANK (torchrun) wrapped into the visible devices -- a
    launcher that shows every rank only its own GPU leaves ordinal 0"""
    if "NATAC_DEVICE" in os.environ:
        return int(os.environ["NATAC_DEVICE"])
    from .device import Context
    dev, n = int(os.environ.get("LOCAL_RANK", "0")), Context.device_count()
    return dev % n if n > 0 else dev


def set_context(ctx):
    global _default_ctx
    _default_ctx = ctx
