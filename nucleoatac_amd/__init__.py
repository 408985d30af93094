"""nucleoatac_amd -- NucleoATAC's occ + nuc signal path on AMD MI355X (gfx950).

Sub-packages mirror the reference's module names for the hot path:
  nucleoatac_amd.pyatac      chunk, tracks, chunkmat2d, fragments, fragmentsizes, bias, seq, VMat, utils
  nucleoatac_amd.nucleoatac  Occupancy, NucleosomeCalling, run_occ, run_nuc, cli
  nucleoatac_amd.executor    how sub-batches move through one GPU: ResidentShard, PipelinedExecutor (the CLI drivers and bench.py)
  nucleoatac_amd.shard       chunk-list sharding over the GPUs of a node, control-plane group, shared BAM decode
All numerics run in libnatac_hip.so (include/natac.h); there is no CPU fallback.
"""
import os

__version__ = "0.1.0"

_default_ctx = None


def get_context():
    """process-wide natac context on GPU `LOCAL_RANK` (0 when unset); created on first use"""
    global _default_ctx
    if _default_ctx is None:
        from .device import Context
        # one context per process: GPU = LOCAL_RANK (torchrun); NATAC_DEVICE overrides it (e.g. several ranks on one GPU in tests)
        _default_ctx = Context(int(os.environ.get("NATAC_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default_ctx


def set_context(ctx):
    global _default_ctx
    _default_ctx = ctx
