"""Thin object layer over the C-ABI: one Context per GPU, DeviceBatch = packed chunks resident in HBM.

All numerics run in libnatac_hip.so (HIP, gfx950).  numpy is only used for the host buffers
that cross the boundary.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib as L
from .packing import PackedChunks


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class _PinnedOwner(object):
    """keeps a hipHostMalloc block alive for the numpy arrays that view it"""

    def __init__(self, nbytes):
        self._lib = L.load()
        p = C.c_void_p()
        L.check(self._lib.natac_host_alloc(int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                self._lib.natac_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (natac_host_alloc): PCIe transfers to / from it run at full rate.  The block
    is released when the last array viewing it dies: numpy collapses `.base` chains to the buffer exporter, so the owner is
    attached to that exporter (the ctypes array), not to an ndarray subclass."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if not np.isscalar(shape) else int(shape)
    owner = _PinnedOwner(max(1, n * dt.itemsize))
    buf = (C.c_char * owner.nbytes).from_address(owner.ptr)
    buf._natac_owner = owner
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


def pinned_copy(a):
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


def pool_trim():
    L.check(L.load().natac_pool_trim())


class TrackStore(object):
    """device-resident copies of batch tracks as their .bedgraph file shows them (natac_store_*; see occstore.py)"""

    def __init__(self):
        self._lib = L.load()
        self._h = C.c_void_p()
        L.check(self._lib.natac_store_create(C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.natac_store_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def set_budget(self, max_bytes=-1, min_free_bytes=-1):
        """HBM budget (natac_store_set_budget): the store stays below max_bytes and leaves min_free_bytes of the device to the
        pipeline's batches (defaults: no cap / a quarter of the device); the first refusal closes the store"""
        L.check(self._lib.natac_store_set_budget(self._h, int(max_bytes), int(min_free_bytes)))

    def adopt(self, batch, tracks, write_zero=True, keep_runs_before_nan=False):
        """copy `tracks` of `batch` into the store; returns the segment id, or None when a value cannot be rounded exactly on the
        device or the store's HBM budget declines the segment (nothing is kept then; the caller reads the files)"""
        t = np.ascontiguousarray(tracks, dtype=np.int32)
        seg, hard = C.c_int64(-1), C.c_int32(0)
        try:
            L.check(self._lib.natac_store_adopt(self._h, batch._h, len(t), _ptr(t), (1 if write_zero else 0) | (2 if keep_runs_before_nan else 0),
                                                C.byref(seg), C.byref(hard)))
        except L.NatacError as e:
            if e.code == -4:        # NATAC_E_NOMEM: HBM is full (24 bytes per base add up on a large genome) -- the files are there
                return None
            raise
        return None if seg.value < 0 else int(seg.value)

    def read(self, ctx, segment, offset, length, slot):
        sg, of, ln = (np.ascontiguousarray(a, dtype=np.int64) for a in (segment, offset, length))
        out = np.empty(int(ln.sum()), dtype=np.float64)
        L.check(self._lib.natac_store_read(self._h, ctx._h, len(sg), _ptr(sg), _ptr(of), _ptr(ln), int(slot), _ptr(out), out.size))
        return out

    def info(self):
        n, by = C.c_int64(0), C.c_int64(0)
        L.check(self._lib.natac_store_info(self._h, C.byref(n), C.byref(by)))
        d = C.c_int64(0)
        L.check(self._lib.natac_store_declined(self._h, C.byref(d)))
        return dict(segments=n.value, bytes=by.value, declined=d.value)


class Context(object):
    """one HIP device + stream (natac_ctx)"""

    def __init__(self, device_id=0):
        self._lib = L.load()
        h = C.c_void_p()
        L.check(self._lib.natac_ctx_create(int(device_id), C.byref(h)))
        self._h = h
        self.device_id = int(device_id)
        self.vmat_shape = None
        self.occ_step = None
        self._batches = weakref.WeakSet()

    def close(self):
        """destroy the context; batches that are still alive are freed first (they hold a pointer to it)"""
        if getattr(self, "_h", None):
            for b in list(self._batches):
                b.free()
            self._lib.natac_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def device_count():
        n = C.c_int(0)
        L.load().natac_device_count(C.byref(n))
        return n.value

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int(0)
        mem = C.c_size_t(0)
        L.check(self._lib.natac_ctx_device_info(self._h, name, 256, C.byref(ncu), C.byref(mem)))
        return dict(name=name.value.decode(), n_cu=ncu.value, mem_bytes=mem.value)

    def device_ids(self):
        """(HIP device ordinal, PCI bus id) of the GPU this context computes on"""
        dev = C.c_int(-1)
        buf = C.create_string_buffer(64)
        L.check(self._lib.natac_ctx_device_ids(self._h, C.byref(dev), buf, 64))
        return dev.value, buf.value.decode()

    def sync(self):
        L.check(self._lib.natac_ctx_sync(self._h))

    # ---- constants -------------------------------------------------------------------------
    def set_vmat(self, mat, lower, upper):
        """VMat template: mat (upper-lower, 2w+1) for insert sizes [lower, upper) (pyatac/VMat.py:24-37)."""
        mat = _f64(mat)
        if mat.ndim != 2 or mat.shape[0] != upper - lower or mat.shape[1] % 2 != 1:
            raise ValueError("mat shape is not consistent with insert limits")
        L.check(self._lib.natac_set_vmat(self._h, _ptr(mat), int(lower), int(upper), mat.shape[1] // 2))
        self.vmat_shape = mat.shape

    def bg_tiling(self, chunk_len):
        """(tiles, extended tiles among them) of the background stage for a chunk of this length with the V-plot that is set
        (natac_bg_tiling)."""
        nt, ex = C.c_int32(0), C.c_int32(0)
        L.check(self._lib.natac_bg_tiling(self._h, int(chunk_len), C.byref(nt), C.byref(ex)))
        return nt.value, ex.value

    def set_sizes(self, sizes):
        """global insert-size distribution over [0, len(sizes)) (pyatac/chunkmat2d.py:154-156)."""
        sizes = _f64(sizes)
        L.check(self._lib.natac_set_sizes(self._h, _ptr(sizes), sizes.shape[0]))

    def set_occ_model(self, nuc_probs, nfr_probs, alphas=None, cutoff=None, step=5, flank=60):
        """OccupancyCalcParams / OccupancyParameters (nucleoatac/Occupancy.py:89-102, 175-193)."""
        nuc_probs, nfr_probs = _f64(nuc_probs), _f64(nfr_probs)
        if nuc_probs.shape != nfr_probs.shape:
            raise ValueError("nuc_probs / nfr_probs shape mismatch")
        if alphas is None:
            alphas = np.linspace(0, 1, 101)
        alphas = _f64(alphas)
        if cutoff is None:
            from scipy import stats
            cutoff = float(stats.chi2.ppf(0.9, 1))
        L.check(self._lib.natac_set_occ_model(self._h, _ptr(nuc_probs), _ptr(nfr_probs), nuc_probs.shape[0],
                                              _ptr(alphas), alphas.shape[0], float(cutoff), int(step), int(flank)))
        step = int(step)
        self.occ_step = step - 1 if step % 2 == 0 else step
        self.occ_upper = nuc_probs.shape[0]

    # ---- Cython-function drop-ins ------------------------------------------------------------
    def make_fragment_mat(self, l, n, start, end, lower, upper):
        """makeFragmentMat (pyatac/fragments.pyx:17-40) on packed fragments (l = pos+4, n = |tlen|-8)."""
        l = np.ascontiguousarray(l, dtype=np.int64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        mat = np.empty((upper - lower, end - start), dtype=np.float64)
        L.check(self._lib.natac_make_fragment_mat(self._h, l.shape[0], _ptr(l), _ptr(n), int(start), int(end),
                                                  int(lower), int(upper), _ptr(mat)))
        return mat

    def get_insertions(self, l, n, start, end, lower=0, upper=2000):
        """getInsertions (pyatac/fragments.pyx:43-67)."""
        l = np.ascontiguousarray(l, dtype=np.int64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        out = np.empty(end - start, dtype=np.float64)
        L.check(self._lib.natac_get_insertions(self._h, l.shape[0], _ptr(l), _ptr(n), int(start), int(end),
                                               int(lower), int(upper), _ptr(out)))
        return out

    def get_stranded_insertions(self, l, n, start, end, lower=0, upper=2000):
        """getStrandedInsertions (pyatac/fragments.pyx:71-97): (plus, minus)."""
        l = np.ascontiguousarray(l, dtype=np.int64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        plus, minus = np.empty(end - start, dtype=np.float64), np.empty(end - start, dtype=np.float64)
        L.check(self._lib.natac_get_stranded_insertions(self._h, l.shape[0], _ptr(l), _ptr(n), int(start), int(end),
                                                        int(lower), int(upper), _ptr(plus), _ptr(minus)))
        return plus, minus

    def fragment_sizes(self, l, n, chunk_starts, chunk_ends, lower, upper):
        """getFragmentSizesFromChunkList (pyatac/fragments.pyx:123-145), one chromosome."""
        l = np.ascontiguousarray(l, dtype=np.int64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        cs = np.ascontiguousarray(chunk_starts, dtype=np.int64)
        ce = np.ascontiguousarray(chunk_ends, dtype=np.int64)
        out = np.empty(upper - lower, dtype=np.float64)
        L.check(self._lib.natac_fragment_sizes(self._h, l.shape[0], _ptr(l), _ptr(n), cs.shape[0], _ptr(cs), _ptr(ce),
                                               int(lower), int(upper), _ptr(out)))
        return out

    def calculate_cov(self, p, v, r, literal=False):
        """calculateCov (nucleoatac/multinomial_cov.pyx:20-31); raises ValueError on a shape mismatch (:21-22)."""
        p, v = _f64(p), _f64(v)
        if p.ndim != 1 or v.ndim != 1 or p.shape[0] != v.shape[0]:
            raise ValueError("p and v must be same shape")
        out = C.c_double(0)
        L.check(self._lib.natac_calculate_cov(self._h, _ptr(p), _ptr(v), p.shape[0], int(r), 1 if literal else 0,
                                              C.byref(out)))
        return out.value

    # ---- operator-level entry points ------------------------------------------------------------
    def smooth(self, sig, window_len, window="flat", sd=None, mode="valid", norm=True):
        """utils.smooth (pyatac/utils.py:23-52) on the GPU."""
        import warnings
        if window not in ("flat", "gaussian"):
            raise Exception("Incorrect window input for smooth. Options are flat, gaussian")
        if mode not in ("valid", "same"):
            raise Exception("mode must be 'valid' or 'same'")
        if window_len % 2 != 1:
            warnings.warn("Window length is even number.  Needs to be odd so adding 1.")
            window_len += 1
        if window == "gaussian":
            if sd is None:
                sd = (window_len - 1) / 6.0
            nn = np.arange(window_len) - (window_len - 1) / 2.0
            w = np.exp(-0.5 * (nn / sd) ** 2)     # scipy.signal.gaussian
        else:
            w = np.ones(window_len)
        sig = _f64(sig)
        w = _f64(w)
        out = np.empty(sig.shape[0] - window_len + 1 if mode == "valid" else sig.shape[0], dtype=np.float64)
        L.check(self._lib.natac_smooth(self._h, _ptr(sig), sig.shape[0], _ptr(w), int(window_len),
                                       0 if mode == "valid" else 1, 1 if norm else 0, _ptr(out)))
        return out

    def make_bias_mat(self, bias_log, track_start, start, end, lower, upper):
        """BiasMat2D.makeBiasMat (pyatac/chunkmat2d.py:140-153) for a log-scale bias track."""
        b = _f64(bias_log)
        mat = np.empty((upper - lower, end - start), dtype=np.float64)
        L.check(self._lib.natac_make_bias_mat(self._h, _ptr(b), b.shape[0], int(track_start), int(start), int(end),
                                              int(lower), int(upper), _ptr(mat)))
        return mat

    def pwm_bias(self, sequence, pwm_mat, nucleotides):
        """InsertionBiasTrack.computeBias (pyatac/bias.py:85-92): log-bias of every position of `sequence`."""
        if isinstance(sequence, np.ndarray) and sequence.dtype == np.uint8:
            seq = np.ascontiguousarray(sequence)              # already upper-case ASCII codes (FastaStore)
        else:
            seq = np.frombuffer(sequence.encode("ascii") if isinstance(sequence, str) else bytes(sequence), dtype=np.uint8)
        logp = _f64(np.log(np.asarray(pwm_mat, dtype=np.float64)))
        lens = set(len(x) for x in nucleotides)
        if len(lens) != 1:     # the reference's seq_to_mat check (pyatac/seq.py:39-41)
            raise Exception("Usage Error! Nucleotides must all be of same length! No mixing single nucleotides with dinucleotides, etc")
        if lens != {1}:
            raise NotImplementedError("k-mer PWMs (words of %d letters) are not supported by natac_pwm_bias: single-"
                                      "nucleotide PWMs only (every PWM shipped with the reference is one)" % lens.pop())
        nucs = np.frombuffer("".join(nucleotides).encode("ascii"), dtype=np.uint8)
        out = np.empty(len(seq) - logp.shape[1] + 1, dtype=np.float64)
        L.check(self._lib.natac_pwm_bias(self._h, _ptr(seq), len(seq), _ptr(logp), _ptr(nucs), logp.shape[0],
                                         logp.shape[1], _ptr(out)))
        return out

    def correlate_valid(self, sub, vmat):
        """signal.correlate(sub, vmat, mode='valid')[0] (nucleoatac/NucleosomeCalling.py:34-36)."""
        sub, vmat = _f64(sub), _f64(vmat)
        if sub.ndim != 2 or vmat.ndim != 2 or sub.shape[0] != vmat.shape[0]:
            raise ValueError("sub and vmat must have the same number of rows")
        out = np.empty(sub.shape[1] - vmat.shape[1] + 1, dtype=np.float64)
        L.check(self._lib.natac_correlate_valid(self._h, _ptr(sub), sub.shape[1], _ptr(vmat), vmat.shape[0],
                                                vmat.shape[1], _ptr(out)))
        return out

    def calculate_occupancy(self, inserts, bias):
        """calculateOccupancy (nucleoatac/Occupancy.py:104-120) with the model of set_occ_model."""
        ins, b = _f64(inserts), _f64(bias)
        out = np.empty(3, dtype=np.float64)
        rc = self._lib.natac_calculate_occupancy(self._h, _ptr(ins), _ptr(b), _ptr(out))
        if rc == -1 and b"likelihood-ratio" in self._lib.natac_last_error():
            raise ValueError("min() arg is an empty sequence")   # what the reference raises (Occupancy.py:118)
        L.check(rc)
        return float(out[0]), float(out[1]), float(out[2])

    def format_doubles(self, vals):
        """python-2 str(float) of every value, formatted ON THE DEVICE (natac_format_doubles; validation of the track writer's
        '%.12g' arithmetic).  Returns (list of str, number of undecidable values)."""
        v = _f64(np.ravel(vals))
        n = v.shape[0]
        out = np.empty(24 * n, dtype=np.uint8)
        off = np.empty(n + 1, dtype=np.int64)
        hard = C.c_int32(0)
        L.check(self._lib.natac_format_doubles(self._h, _ptr(v), n, _ptr(out), out.nbytes, _ptr(off), C.byref(hard)))
        raw = out.tobytes()
        return [raw[off[i]:off[i + 1]].decode("ascii") for i in range(n)], hard.value

    # ---- profiling ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        L.check(self._lib.natac_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        L.check(self._lib.natac_profile_reset(self._h))

    def profile(self):
        """{kernel name: (total ms, launches)} from HIP events on the context's stream"""
        out = {}
        for k, name in enumerate(L.KERNEL_NAMES):
            ms, n = C.c_double(0), C.c_int64(0)
            L.check(self._lib.natac_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def clock_trace_start(self, max_samples=20000, interval_us=100):
        """start the one-wave shader-clock sampler (natac_clock_trace_start); profile_enable(True) makes the launches that follow
        appear as intervals on its time axis"""
        L.check(self._lib.natac_clock_trace_start(self._h, int(max_samples), int(interval_us)))

    def clock_trace_stop(self):
        """-> dict(t_ms, ghz: clock between consecutive samples; per_kernel: {kernel name: mean GHz over its launches' intervals,
        weighted by time}; intervals: [(kernel name, t0_ms, t1_ms)])"""
        ns, ni = C.c_int64(0), C.c_int64(0)
        L.check(self._lib.natac_clock_trace_stop(self._h, C.byref(ns), C.byref(ni)))
        t, cyc = np.empty(ns.value, np.float64), np.empty(ns.value, np.float64)
        ik, i0, i1 = np.empty(ni.value, np.int32), np.empty(ni.value, np.float64), np.empty(ni.value, np.float64)
        L.check(self._lib.natac_clock_trace_fetch(self._h, ns.value, _ptr(t), _ptr(cyc), ni.value, _ptr(ik), _ptr(i0), _ptr(i1)))
        dt = np.diff(t)
        ghz = np.where(dt > 0, np.diff(cyc) / np.maximum(dt, 1e-12) / 1e6, np.nan)     # cycles per ms / 1e6 = GHz
        if len(dt):      # the two counters are read a few cycles apart: a slope over a very short interval (or a wrapped counter) is noise
            ghz[(dt < 0.25 * np.median(dt)) | ~(ghz > 0) | (ghz > 10.0)] = np.nan
        mid = 0.5 * (t[1:] + t[:-1])
        per, w = {}, {}
        for k, a, b in zip(ik, i0, i1):
            m = (mid >= a) & (mid <= b) & np.isfinite(ghz)
            if m.any():
                name = L.KERNEL_NAMES[int(k)]
                per[name] = per.get(name, 0.0) + float(np.sum(ghz[m] * dt[m]))
                w[name] = w.get(name, 0.0) + float(np.sum(dt[m]))
        return dict(t_ms=t, ghz=ghz, per_kernel={k: per[k] / w[k] for k in per},
                    intervals=[(L.KERNEL_NAMES[int(k)], float(a), float(b)) for k, a, b in zip(ik, i0, i1)])

    def timer_start(self):
        L.check(self._lib.natac_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_double(0)
        L.check(self._lib.natac_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def upload(self, packed):
        return DeviceBatch(self, packed)


class DeviceBatch(object):
    """a PackedChunks batch resident in HBM (natac_batch) and its per-base output tracks"""

    def __init__(self, ctx, packed):
        if not isinstance(packed, PackedChunks):
            raise TypeError("expected PackedChunks")
        self.ctx = ctx
        self._lib = ctx._lib
        self.packed = packed
        h = C.c_void_p()
        if getattr(packed, "seq", None) is not None:        # Tn5 bias scored on the device from the sequence windows
            L.check(self._lib.natac_batch_create_from_seq(
                ctx._h, packed.n_chunks, _ptr(packed.chunk_len), _ptr(packed.frag_off), _ptr(packed.frag_lpos), _ptr(packed.frag_ilen),
                _ptr(packed.seq_off), _ptr(packed.seq), _ptr(packed.pwm_log), _ptr(packed.pwm_nucs), packed.pwm_log.shape[0],
                packed.pwm_log.shape[1], int(packed.bias_left), int(packed.bias_right), C.byref(h)))
        else:
            L.check(self._lib.natac_batch_create(
                ctx._h, packed.n_chunks, _ptr(packed.chunk_len), _ptr(packed.frag_off), _ptr(packed.frag_lpos),
                _ptr(packed.frag_ilen), _ptr(packed.bias_off) if packed.bias_log is not None else None,
                _ptr(packed.bias_log), int(packed.bias_left), int(packed.bias_right), C.byref(h)))
        self._h = h
        self.total_bp = packed.total_bp
        ctx._batches.add(self)

    def free(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):       # never touch a batch whose context is gone
                self._lib.natac_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def release_outputs(self):
        """free every output array of the batch, keep the packed inputs in HBM (natac_batch_release_outputs)"""
        L.check(self._lib.natac_batch_release_outputs(self._h))

    def run_nuc(self, smooth_sd=10):
        L.check(self._lib.natac_run_nuc(self._h, float(smooth_sd)))

    def run_occ(self):
        L.check(self._lib.natac_run_occ(self._h))

    def run_ins(self, lower=0, upper=2000):
        L.check(self._lib.natac_run_ins(self._h, int(lower), int(upper)))

    def run_candidates(self, cand_chunk, cand_pos):
        cc = np.ascontiguousarray(cand_chunk, dtype=np.int32)
        cp = np.ascontiguousarray(cand_pos, dtype=np.int32)
        if cc.shape != cp.shape:
            raise ValueError("cand_chunk / cand_pos shape mismatch")
        n = cc.shape[0]
        lr, var, z = (np.empty(n, dtype=np.float64) for _ in range(3))
        L.check(self._lib.natac_run_candidates(self._h, n, _ptr(cc), _ptr(cp), _ptr(lr), _ptr(var), _ptr(z)))
        return lr, var, z

    def run_candidates_cov(self, cand_chunk, cand_pos, mode="closed"):
        """calculateCov at many candidates in one arithmetic variant (natac_run_candidates_cov): "closed" (fp64 closed form),
        "literal" (the .pyx's O(N^2) pair sum, fp64) or "fp32" (closed form in float)"""
        cc = np.ascontiguousarray(cand_chunk, dtype=np.int32)
        cp = np.ascontiguousarray(cand_pos, dtype=np.int32)
        if cc.shape != cp.shape:
            raise ValueError("cand_chunk / cand_pos shape mismatch")
        var = np.empty(cc.shape[0], dtype=np.float64)
        L.check(self._lib.natac_run_candidates_cov(self._h, cc.shape[0], _ptr(cc), _ptr(cp),
                                                   {"closed": 0, "literal": 1, "fp32": 2}[mode], _ptr(var)))
        return var

    def run_peaks(self, min_signal=0.0, sep=25, boundary=60, order=12, download=True):
        """candidate search + LR / var / z on the device (natac_run_peaks): call_peaks(norm + smoothed, ...) of every chunk
        as NucChunk.findAllNucs does it (nucleoatac/NucleosomeCalling.py:297-301).  Returns (chunk, pos, lr, var, z), or with
        download=False only the number of candidates (the arrays stay in HBM for download_peaks)."""
        maxL = int(self.packed.chunk_len.max())
        jitter = np.ascontiguousarray(np.random.RandomState(seed=25).uniform(0, 10 ** -12, maxL))   # utils.py:94-97
        n = C.c_int64(0)
        L.check(self._lib.natac_run_peaks(self._h, float(min_signal), int(sep), int(boundary), int(order), _ptr(jitter), maxL,
                                          C.byref(n)))
        n = n.value
        if not download:
            return n
        return self.download_peaks(n)

    def download_peaks(self, n):
        """(chunk, pos, lr, var, z) of the last run_peaks"""
        cc, cp = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        lr, var, z = (np.empty(n, dtype=np.float64) for _ in range(3))
        L.check(self._lib.natac_download_peaks(self._h, n, _ptr(cc), _ptr(cp), _ptr(lr), _ptr(var), _ptr(z)))
        return cc, cp, lr, var, z

    def run_track_peaks(self, track, min_signal=0.0, sep=120, boundary=None, order=1):
        """utils.call_peaks on one per-base track of every chunk, on the device (natac_run_track_peaks).
        Returns (chunk, pos) arrays in chunk order."""
        if boundary is None:
            boundary = sep // 2
        maxL = int(self.packed.chunk_len.max())
        jitter = np.ascontiguousarray(np.random.RandomState(seed=25).uniform(0, 10 ** -12, maxL))
        n = C.c_int64(0)
        L.check(self._lib.natac_run_track_peaks(self._h, int(track), float(min_signal), int(sep), int(boundary), int(order),
                                                _ptr(jitter), maxL, C.byref(n)))
        n = n.value
        cc, cp = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        L.check(self._lib.natac_download_peaks(self._h, n, _ptr(cc), _ptr(cp), None, None, None))
        return cc, cp

    def run_occ_peaks(self, min_occ=0.1, sep=120):
        """OccChunk.callPeaks + getNucDist of every chunk on the device (natac_run_occ_peaks).  Returns
        (chunk, pos, occ, lower, upper, reads, keep) of every call_peaks peak and nuc_dist[n_chunks, upper]."""
        maxL = int(self.packed.chunk_len.max())
        jitter = np.ascontiguousarray(np.random.RandomState(seed=25).uniform(0, 10 ** -12, maxL))
        n = C.c_int64(0)
        L.check(self._lib.natac_run_occ_peaks(self._h, float(min_occ), int(sep), _ptr(jitter), maxL, C.byref(n)))
        n = n.value
        cc, cp, keep = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
        occ, lo, up, rd = (np.empty(n, np.float64) for _ in range(4))
        L.check(self._lib.natac_download_occ_peaks(self._h, n, _ptr(cc), _ptr(cp), _ptr(occ), _ptr(lo), _ptr(up), _ptr(rd), _ptr(keep)))
        upper = self.ctx.occ_upper
        nd = np.empty((self.packed.n_chunks, upper), dtype=np.float64)
        L.check(self._lib.natac_download_nuc_dist(self._h, _ptr(nd), nd.nbytes))
        return cc, cp, occ, lo, up, rd, keep, nd

    def format_track(self, t, chroms, chunk_start, write_zero=True, compress=True, keep_runs_before_nan=False, out=None, wait=True):
        """Track.write_track of per-base track `t` for every chunk ON THE DEVICE (natac_batch_format_track): returns
        (bytes as a uint8 array -- bedGraph text, or BGZF members when `compress` --, info dict).  `chroms`: one chromosome name
        per chunk; `out`: a function n_bytes -> uint8 buffer (e.g. a pinned slot) that receives the result.  `wait=False`: the copy into
        `out`'s (pinned) buffer is only STARTED (natac_batch_format_fetch_begin) so that the next track can be formatted meanwhile; the
        bytes are valid after `format_wait()`."""
        nc = self.packed.n_chunks
        if len(chroms) != nc or len(chunk_start) != nc:
            raise ValueError("one chromosome name and start per chunk")
        # the name table of a batch is formed once, not once per track: a per-chunk Python loop here (str(), a dict lookup) holds the GIL for
        # ~2 ms per 2,500 chunks -- with five tracks per sub-batch and six executor threads that serialised the whole bedGraph.gz
        # pipeline on the interpreter (round 6: kernels busy 55 % of the leg's wall time)
        key = (id(chroms), id(chunk_start))
        cached = getattr(self, "_fmt_names", None)
        if cached is None or cached[0] != key:
            uniq, inv = np.unique(np.asarray(chroms, dtype=str), return_inverse=True)      # sorted, like sorted(set(...))
            names = [str(c) for c in uniq]
            cached = (key, names, np.ascontiguousarray(inv, dtype=np.int32), np.ascontiguousarray(chunk_start, dtype=np.int64),
                      (C.c_char_p * len(names))(*[c.encode("ascii") for c in names]), chroms, chunk_start)   # (the two objects are kept alive: id() stays theirs)
            self._fmt_names = cached
        _, names, cid, cs, arr = cached[:5]
        nb, nt, nl, hard = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int32(0)
        L.check(self._lib.natac_batch_format_track(self._h, int(t), _ptr(cid), arr, len(names), _ptr(cs),
                                                   (1 if write_zero else 0) | (2 if keep_runs_before_nan else 0), 1 if compress else 0,
                                                   C.byref(nb), C.byref(nt), C.byref(nl), C.byref(hard)))
        buf = out(nb.value) if out is not None else np.empty(nb.value, dtype=np.uint8)
        if buf.dtype != np.uint8 or buf.size < nb.value:
            raise ValueError("out must give a uint8 buffer of at least %d bytes" % nb.value)
        if wait:
            L.check(self._lib.natac_batch_format_fetch(self._h, _ptr(buf), buf.nbytes))
        else:
            L.check(self._lib.natac_batch_format_fetch_begin(self._h, _ptr(buf), buf.nbytes))
        info = dict(bytes=nb.value, text_bytes=nt.value, lines=nl.value, hard=hard.value)
        if compress:
            # tabix records of this result (runs of lines per 16-kb leaf bin) for writer.TbiBuilder.push
            ng, nm, ntx = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            L.check(self._lib.natac_batch_format_index_size(self._h, C.byref(ng), C.byref(nm), C.byref(ntx)))
            ng, nm = ng.value, nm.value
            rec = dict(names=names, n_text=ntx.value, cid=np.empty(ng, np.int32), beg=np.empty(ng, np.int64), end=np.empty(ng, np.int64),
                       count=np.empty(ng, np.int64), t0=np.empty(ng, np.uint64), t1=np.empty(ng, np.uint64),
                       member_pos=np.zeros(nm + 1, np.uint64))
            L.check(self._lib.natac_batch_format_index_fetch(self._h, _ptr(rec["cid"]), _ptr(rec["beg"]), _ptr(rec["end"]), _ptr(rec["count"]),
                                                             _ptr(rec["t0"]), _ptr(rec["t1"]), _ptr(rec["member_pos"])))
            info["index"] = rec
        return buf[:nb.value], info

    def format_wait(self):
        """wait for every result whose copy `format_track(..., wait=False)` started"""
        L.check(self._lib.natac_batch_format_fetch_wait(self._h))

    def set_track(self, t, vals):
        """overwrite float64 per-base track `t` with host values (natac_batch_set_track), e.g. to send an externally computed
        track through the device-side writer"""
        v = _f64(vals)
        L.check(self._lib.natac_batch_set_track(self._h, int(t), _ptr(v), v.shape[0]))

    def track(self, t, out=None):
        """download one per-base track (concatenated over chunks); `out`: destination array (e.g. pinned_empty)"""
        dt = np.int32 if t == L.T_INS else np.float64
        if out is None:
            out = np.empty(self.total_bp, dtype=dt)
        elif out.dtype != dt or out.size != self.total_bp or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a contiguous %s array of %d elements" % (dt.__name__, self.total_bp))
        L.check(self._lib.natac_batch_download(self._h, int(t), _ptr(out), out.nbytes))
        return out

    def grid_info(self):
        bp, grid, nf = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(self._lib.natac_batch_info(self._h, C.byref(bp), C.byref(grid), C.byref(nf)))
        return bp.value, grid.value, nf.value

    def grid(self, which):
        _, total_grid, _ = self.grid_info()
        out = np.empty(total_grid, dtype=np.float64)
        L.check(self._lib.natac_batch_download_grid(self._h, int(which), _ptr(out), out.nbytes))
        return out

    def status(self):
        out = np.empty(self.packed.n_chunks, dtype=np.int32)
        L.check(self._lib.natac_batch_status(self._h, _ptr(out), out.nbytes))
        return out

    def split(self, flat):
        """split a concatenated per-base array into per-chunk views"""
        off = self.packed.out_off
        return [flat[int(off[i]):int(off[i + 1])] for i in range(self.packed.n_chunks)]
