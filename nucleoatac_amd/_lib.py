"""ctypes binding of libnatac_hip.so (C-ABI declared in include/natac.h).

There is no CPU fallback: if the HIP library is missing, or a compute entry point is
called without a GPU, this raises -- the product path never routes through oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The pipelined executor runs six contexts with two streams each (stages + the copies of finished results).  The HIP runtime maps streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise: with the default the bedGraph.gz
# host-to-host leg measured 550 Mbp/s, with 8 queues 567-592 (profiles/r6/README.md).  Read once when the runtime initialises, so it is set
# here, before the library is loaded; an explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ABI_VERSION = 3      # NATAC_ABI_VERSION of include/natac.h (tests/test_abi.py compares the two)
LIB_PATH = os.environ.get("NATAC_LIB") or os.path.join(_HERE, "libnatac_hip.so")   # NATAC_LIB: A/B builds of the same ABI

# enums of include/natac.h
T_NUC_COV, T_NFR_COV, T_RAW, T_BACKGROUND, T_NORM, T_SMOOTH = 0, 1, 2, 3, 4, 5
T_OCC, T_OCC_LOWER, T_OCC_UPPER, T_OCC_COV, T_INS, T_OCC_PREFILL = 6, 7, 8, 9, 10, 11
G_OCC, G_LOWER, G_UPPER = 0, 1, 2
K_FRAG_GATHER, K_BACKGROUND, K_SMOOTH_NUC, K_OCC_MLE, K_OCC_SMOOTH, K_OCC_FILL, K_INS, K_CAND, K_SIZE_HIST = range(9)
KERNEL_NAMES = ["frag_gather", "background", "smooth_nuc", "occ_mle", "occ_smooth", "occ_fill", "insertions",
                "candidates", "size_hist"]

_vp, _i32, _i64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
_pp = C.POINTER(C.c_void_p)

# every exported symbol of include/natac.h with its signature (tests check this list against the header)
SIGNATURES = {
    "natac_abi_version": (C.c_int, []),
    "natac_last_error": (C.c_char_p, []),
    "natac_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "natac_ctx_create": (C.c_int, [C.c_int, _pp]),
    "natac_ctx_destroy": (None, [_vp]),
    "natac_ctx_sync": (C.c_int, [_vp]),
    "natac_ctx_device_info": (C.c_int, [_vp, C.c_char_p, _sz, C.POINTER(C.c_int), C.POINTER(_sz)]),
    "natac_ctx_device_ids": (C.c_int, [_vp, C.POINTER(C.c_int), C.c_char_p, _sz]),
    "natac_set_vmat": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "natac_set_sizes": (C.c_int, [_vp, _vp, C.c_int]),
    "natac_set_occ_model": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int, _f64, C.c_int, C.c_int]),
    "natac_batch_create": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _pp]),
    "natac_batch_create_from_seq": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _i32, _i32, _pp]),
    "natac_batch_free": (None, [_vp]),
    "natac_batch_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "natac_bg_tiling": (C.c_int, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "natac_batch_release_outputs": (C.c_int, [_vp]),
    "natac_run_nuc": (C.c_int, [_vp, _f64]),
    "natac_run_occ": (C.c_int, [_vp]),
    "natac_run_ins": (C.c_int, [_vp, C.c_int, C.c_int]),
    "natac_store_create": (C.c_int, [_pp]),
    "natac_store_free": (None, [_vp]),
    "natac_store_adopt": (C.c_int, [_vp, _vp, _i32, _vp, C.c_int, C.POINTER(_i64), C.POINTER(_i32)]),
    "natac_store_read": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _sz]),
    "natac_store_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "natac_store_set_budget": (C.c_int, [_vp, _i64, _i64]),
    "natac_store_declined": (C.c_int, [_vp, C.POINTER(_i64)]),
    "natac_clock_trace_start": (C.c_int, [_vp, C.c_int, C.c_int]),
    "natac_clock_trace_stop": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "natac_clock_trace_fetch": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "natac_run_candidates": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "natac_run_candidates_cov": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, _vp]),
    "natac_run_peaks": (C.c_int, [_vp, _f64, C.c_int, C.c_int, C.c_int, _vp, _i64, C.POINTER(_i64)]),
    "natac_download_peaks": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "natac_run_track_peaks": (C.c_int, [_vp, C.c_int, _f64, C.c_int, C.c_int, C.c_int, _vp, _i64, C.POINTER(_i64)]),
    "natac_run_occ_peaks": (C.c_int, [_vp, _f64, C.c_int, _vp, _i64, C.POINTER(_i64)]),
    "natac_download_occ_peaks": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "natac_download_nuc_dist": (C.c_int, [_vp, _vp, _sz]),
    "natac_batch_download": (C.c_int, [_vp, C.c_int, _vp, _sz]),
    "natac_batch_download_grid": (C.c_int, [_vp, C.c_int, _vp, _sz]),
    "natac_batch_set_track": (C.c_int, [_vp, C.c_int, _vp, _sz]),
    "natac_batch_status": (C.c_int, [_vp, _vp, _sz]),
    "natac_batch_track_ptr": (C.c_int, [_vp, C.c_int, _pp]),
    "natac_make_fragment_mat": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp]),
    "natac_get_insertions": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp]),
    "natac_get_stranded_insertions": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp, _vp]),
    "natac_fragment_sizes": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _vp, _vp, C.c_int, C.c_int, _vp]),
    "natac_calculate_cov": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.POINTER(_f64)]),
    "natac_smooth": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "natac_make_bias_mat": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, C.c_int, C.c_int, _vp]),
    "natac_pwm_bias": (C.c_int, [_vp, _vp, _i64, _vp, _vp, C.c_int, C.c_int, _vp]),
    "natac_correlate_valid": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, C.c_int, _vp]),
    "natac_calculate_occupancy": (C.c_int, [_vp, _vp, _vp, _vp]),
    "natac_write_bedgraph": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, _i32, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                             C.POINTER(_i64)]),
    "natac_batch_format_track": (C.c_int, [_vp, C.c_int, _vp, _vp, _i32, _vp, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64),
                                 C.POINTER(_i64), C.POINTER(_i32)]),
    "natac_batch_format_fetch": (C.c_int, [_vp, _vp, _sz]),
    "natac_batch_format_fetch_begin": (C.c_int, [_vp, _vp, _sz]),
    "natac_batch_format_fetch_wait": (C.c_int, [_vp]),
    "natac_tbi_create": (C.c_int, [_pp]),
    "natac_tbi_free": (None, [_vp]),
    "natac_batch_format_index_size": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "natac_batch_format_index_fetch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "natac_tbi_push": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64]),
    "natac_tbi_write": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "natac_format_doubles": (C.c_int, [_vp, _vp, _i64, _vp, _sz, _vp, C.POINTER(_i32)]),
    "natac_bgzf_lines_host": (C.c_int, [C.c_char_p, _i64, _vp, _i64, _vp, _sz, C.POINTER(_i64)]),
    "natac_write_bed_rows": (C.c_int, [C.c_char_p, C.c_int, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i32]),
    "natac_write_bed_rows_labeled": (C.c_int, [C.c_char_p, C.c_int, _i64, _vp, _vp, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, C.c_int32]),
    "natac_bgzip_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "natac_tabix_index": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_i64)]),
    "natac_tbx_open": (C.c_int, [C.c_char_p, _pp]),
    "natac_tbx_close": (None, [_vp]),
    "natac_tbx_read_values": (C.c_int, [_vp, C.c_char_p, _i64, _i64, C.c_int, _f64, _vp, C.POINTER(_i64)]),
    "natac_tbx_read_regions": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int32, _vp, _vp, C.c_int, _f64, _vp, _vp, C.c_int, C.POINTER(_i64)]),
    "natac_pack_chunks": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "natac_bam_open": (C.c_int, [C.c_char_p, C.c_int, _pp]),
    "natac_bam_close": (None, [_vp]),
    "natac_bam_counts": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64)]),
    "natac_bam_ref_info": (C.c_int, [_vp, _i32, C.c_char_p, _sz, C.POINTER(_i64), C.POINTER(_i64)]),
    "natac_bam_ref_reads": (C.c_int, [_vp, _i32, _vp, _vp, _i64]),
    "natac_bam_open_device": (C.c_int, [_vp, C.c_char_p, _pp, C.POINTER(C.c_int)]),
    "natac_inflate_raw_host": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t]),
    "natac_fuzz_evaluate": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "natac_bedtab_open": (C.c_int, [C.c_char_p, _vp, C.c_int32, _pp]),
    "natac_bedtab_close": (None, [_vp]),
    "natac_bedtab_dims": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "natac_bedtab_name": (C.c_int, [_vp, _i32, C.c_char_p, C.c_size_t]),
    "natac_bedtab_fetch": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "natac_fasta_open": (C.c_int, [C.c_char_p, C.c_int, _pp]),
    "natac_fasta_close": (None, [_vp]),
    "natac_fasta_count": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "natac_fasta_info": (C.c_int, [_vp, _i32, C.c_char_p, C.c_size_t, C.POINTER(_i64)]),
    "natac_fasta_read": (C.c_int, [_vp, _i32, _vp, _i64]),
    "natac_host_alloc": (C.c_int, [_sz, _pp]),
    "natac_host_free": (C.c_int, [_vp]),
    "natac_pool_trim": (C.c_int, []),
    "natac_profile_enable": (C.c_int, [_vp, C.c_int]),
    "natac_profile_get": (C.c_int, [_vp, C.c_int, C.POINTER(_f64), C.POINTER(_i64)]),
    "natac_profile_reset": (C.c_int, [_vp]),
    "natac_timer_start": (C.c_int, [_vp]),
    "natac_timer_stop": (C.c_int, [_vp, C.POINTER(_f64)]),
}


def csrc_sha16():
    """first 16 hex digits of the sha256 over the library's sources (csrc/* and include/natac.h, sorted by name): stamps
    profiles so that a counter summary collected from an older build is recognisable"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".inc")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "natac.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profile_sha16(synth_path=None, bench_path=None):
    """stamp of a committed counter summary (profiles/*/pmc_summary.csv): the library's sources (csrc_sha16) AND the input the counters
    were collected on -- nucleoatac_amd/synth.py (the generator) and bench.py's workload builder (make_workload and the defaults of the
    arguments it reads).  A change of either makes bench.py report `traffic_source.stale: true`.  The two paths are parameters for
    tests/test_host_logic.py."""
    import hashlib
    import re
    root = os.path.dirname(_HERE)
    h = hashlib.sha256(csrc_sha16().encode())
    with open(synth_path or os.path.join(_HERE, "synth.py"), "rb") as fh:
        h.update(fh.read())
    with open(bench_path or os.path.join(root, "bench.py")) as fh:
        src = fh.read()
    m = re.search(r"^def make_workload\(.*?(?=^def |\Z)", src, flags=re.S | re.M)
    h.update((m.group(0) if m else "").encode())
    for arg in ("--workload", "--chunks", "--chunk-len", "--frags-per-chunk", "--seed", "--sub-chunks"):
        m = re.search(r"add_argument\(\"%s\".*" % re.escape(arg), src)
        h.update((m.group(0) if m else "").encode())
    return h.hexdigest()[:16]


class NatacError(RuntimeError):
    """error reported by libnatac_hip.so (negative return code of the C-ABI)"""

    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libnatac_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """dlopen libnatac_hip.so and bind every symbol; raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("nucleoatac_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.natac_abi_version() != ABI_VERSION:
        raise ImportError("libnatac_hip.so ABI version %d, this binding is for %d: rebuild (python __graft_entry__.py)" % (lib.natac_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise NatacError(rc, load().natac_last_error().decode("utf-8", "replace"))
