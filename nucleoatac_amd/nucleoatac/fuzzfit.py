"""Many `Nucleosome.getFuzz` fits (reference NucleosomeCalling.py:137-194) advanced in lockstep.

The reference fits up to three Gaussians per call with scipy's L-BFGS-B and lets scipy difference the objective numerically:
~70 objective + gradient evaluations per call, every one a few dozen numpy calls on arrays of ~100 elements, i.e. the
interpreter, not the arithmetic.  94 % of `nucleoatac nuc` on the accelerated path was this loop.

scipy's optimiser is a reverse-communication routine (`scipy.optimize._lbfgsb.setulb`: call, it asks for f and g at x, call
again), so K independent fits can be advanced together: each round collects the points the K optimisers ask for, evaluates
all objectives and finite-difference gradients in ONE set of numpy calls on (K, n + 1, M) arrays, and hands the values back.
Every element goes through the same numpy operations as in the per-call path (`fit_fuzz_one`): the elementwise ones do not
depend on the position in an array, the row maximum is exact, and the row sums are taken over arrays of each row's own length
(numpy's pairwise summation depends on it; padding is cut off before the sum).  The optimisers therefore see bit-identical
values and take the same steps: tests/test_host_logic.py compares the two paths on every fit of a few hundred chunks.

The driver loop below is `scipy.optimize._lbfgsb_py._minimize_lbfgsb` (scipy 1.15) with the bookkeeping this fit does not use
removed.  Attribution for that restated loop: SciPy, Copyright (c) 2001-2002 Enthought, Inc. 2003-, SciPy Developers, BSD 3-Clause
License (the L-BFGS-B Fortran code behind `setulb` -- C. Zhu, R. Byrd, P. Lu, J. Nocedal, J. L. Morales -- is called, not copied).  `available()` runs one small fit through both paths at import time of the caller; a scipy whose private routine has
another signature, or answers differently, makes the callers fall back to `fit_fuzz_one`."""
from bisect import bisect_left

import numpy as np

M_CORR = 10                      # L-BFGS-B defaults of scipy.optimize.minimize(method="L-BFGS-B")
FTOL = 2.2204460492503131e-09
GTOL = 1e-5
MAXFUN = 15000
MAXITER = 15000
MAXLS = 20
FD_STEP = 1e-8                   # absolute 2-point step of approx_derivative as L-BFGS-B calls it
import os
GROUP = int(os.environ.get("NATAC_FIT_GROUP", "192"))   # fits advanced together (the (GROUP, n + 1, M) temporaries stay cache-sized)
PAD_X = 1e9                      # abscissa of the padding columns: exp(-(1e9 - mean)^2 / 2v) == 0 < every row maximum


def problem(vals, allnucs, index, nonredundant_sep, smooth_sd):
    """window, bounds and start of the fit for the call at chunk-relative `index` (NucleosomeCalling.py:139-173); clamps
    vals[left:right] at 0 in place like the reference.  Returns (sig, lb, ub, guess, left)."""
    third = nonredundant_sep // 3
    x = bisect_left(allnucs, index)
    if x > 0 and index - allnucs[x - 1] < nonredundant_sep:
        left = allnucs[x - 1]
        means = (index - allnucs[x - 1], 0)
    else:
        left = index - third
        means = (third,)
    if x < len(allnucs) - 1 and allnucs[x + 1] - index < nonredundant_sep:
        right = allnucs[x + 1]
        means += (allnucs[x + 1] - left,)
    else:
        right = index + third + 1
    sig = vals[left:right]
    sig[sig < 0] = 0
    top = max(sig)
    lb, ub, guess = [], [], []
    for m in means:
        lb += [2 ** 2, 0.001, m - 10]
        ub += [50 ** 2, top * 1.1, m + 10]
        guess += [smooth_sd ** 2, top * 0.9, m]
    return sig, np.array(lb, dtype=np.float64), np.array(ub, dtype=np.float64), np.array(guess, dtype=np.float64), left


class _State(object):
    """workspace of one L-BFGS-B run (the arrays _minimize_lbfgsb allocates)"""
    __slots__ = ("x", "task", "args", "nit", "nreq", "redo")

    def __init__(self, guess, lb, ub):
        n, m = len(guess), M_CORR
        self.x = np.array(np.clip(guess, lb, ub), dtype=np.float64)
        self.task = np.zeros(2, dtype=np.int32)
        # setulb's argument list, built once: (m, x, l, u, nbd [both bounds finite], f, g, factr, pgtol, wa, iwa, task, lsave, isave,
        # dsave, maxls, ln_task); slots 5 and 6 take the objective and gradient of every round
        self.args = [m, self.x, lb, ub, np.full(n, 2, dtype=np.int32), 0.0, np.zeros(n, dtype=np.float64), FTOL / np.finfo(float).eps, GTOL,
                     np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, dtype=np.float64), np.zeros(3 * n, dtype=np.int32), self.task,
                     np.zeros(4, dtype=np.int32), np.zeros(44, dtype=np.int32), np.zeros(29, dtype=np.float64), MAXLS,
                     np.zeros(2, dtype=np.int32)]
        self.nit = 0
        self.nreq = 0            # f, g requests so far: an upper bound of scipy's nfev (ScalarFunction does not count a repeated point)
        self.redo = False        # the evaluation / iteration limits came into reach: this fit is repeated through scipy.optimize.minimize


def _advance(st, setulb, factr=None):
    """run the optimiser until it asks for f, g (True) or stops (False) -- the `while True` of _minimize_lbfgsb"""
    args, task = st.args, st.task
    while True:
        setulb(*args)
        t = task[0]
        if t == 3:
            st.nreq += 1
            return True
        if t == 1:
            st.nit += 1
            if st.nit >= MAXITER or st.nreq > MAXFUN:    # scipy would stop here or soon (its nfev <= nreq): let scipy decide
                st.redo = True
                return False
        else:
            return False


def _evaluate(X0, lb, ub, sig, xs, lens, work):
    """objective and 2-point finite-difference gradient of K fits with the same number of Gaussians.
    X0, lb, ub: (K, n); sig, xs: (K, M) padded (sig with 0, xs with PAD_X); lens: (K,) true window lengths, ascending;
    work: two float64 buffers of at least (n + G) K M and (n + 1) K M elements.  Returns f (K,), g (K, n)."""
    K, n = X0.shape
    M = xs.shape[1]
    G = n // 3
    # step of every parameter: +1e-8, flipped at an upper bound (scipy.optimize._numdiff._adjust_scheme_to_bounds, 1-sided)
    h = np.full((K, n), FD_STEP)
    lower_dist, upper_dist = X0 - lb, ub - X0
    x = X0 + h
    violated = (x < lb) | (x > ub)
    fitting = np.abs(h) <= np.maximum(lower_dist, upper_dist)
    h[violated & fitting] *= -1
    forward = (upper_dist >= lower_dist) & ~fitting
    h[forward] = upper_dist[forward]
    backward = (upper_dist < lower_dist) & ~fitting
    h[backward] = -lower_dist[backward]
    xh = X0 + h
    dx = xh - X0
    # `norm()` (NucleosomeCalling.py:92-97) of the Gaussians that differ between the n + 1 parameter sets of a fit: set i < n shifts
    # one parameter of Gaussian i // 3 and shares the other Gaussians with the point itself (set n), so 4 G evaluations (rows
    # 0..n-1: Gaussian i // 3 with parameter i shifted; rows n..n+G-1: the unshifted ones) stand for (n + 1) G.  In place in one
    # (4 G, K, M) buffer, the same operations in the same order as `err` in fit_fuzz_one.
    par = np.empty((3, n + G, K))                          # variance, weight, mean of every row
    for c in range(3):
        par[c, :n] = X0.T[3 * (np.arange(n) // 3) + c]
        par[c, c:n:3] = xh.T[c::3]
        par[c, n:] = X0.T[c::3]
    v, w, mean = par[0][:, :, None], par[1], par[2][:, :, None]
    y = work[0][:(n + G) * K * M].reshape(n + G, K, M)
    np.subtract(xs, mean, out=y)
    np.square(y, out=y)                                    # x ** 2 is numpy's square
    np.negative(y, out=y)
    np.divide(y, 2 * v, out=y)
    np.exp(y, out=y)
    np.multiply(1.0 / np.sqrt(2 * np.pi * v), y, out=y)
    np.multiply(y, (w / y.max(axis=2))[:, :, None], out=y)
    # fit of set i = ((0 + y_0) + y_1) + y_2 with its own version of every Gaussian, then the squared residuals
    fit = work[1][:(n + 1) * K * M].reshape(n + 1, K, M)
    fit[:] = 0.0
    for j in range(G):
        for i in range(n + 1):
            np.add(fit[i], y[i if i < n and i // 3 == j else n + j], out=fit[i])
    r = np.subtract(fit, sig, out=fit)
    np.square(r, out=r)
    s = np.empty((n + 1, K))
    cut = np.concatenate(([0], np.flatnonzero(np.diff(lens)) + 1, [K]))   # runs of equal window length
    for a, b in zip(cut[:-1], cut[1:]):                    # sums over each row's own length (pairwise summation order)
        np.sum(r[:, a:b, :lens[a]], axis=2, out=s[:, a:b])
    f0 = s[n]
    return f0, ((s[:n] - f0) / dx.T).T


NATIVE_EVAL = True      # one native call per round (natac_fuzz_evaluate) instead of ~40 numpy calls, when it gives numpy's bits
_NATIVE = None          # (lib, exp loop pointer, exp loop data) once checked, False when unavailable


def _numpy_exp_loop():
    """numpy's inner loop for float64 exp: the function pointer and its data stored in the np.exp ufunc object (PyUFuncObject,
    numpy/ufuncobject.h: PyObject_HEAD, four ints, functions, data, ntypes, reserved1, name, types).  The native evaluator calls
    it on its own buffer, so the exponentials are numpy's to the last bit."""
    import ctypes as C
    import sys
    import sysconfig
    # the struct below is the layout of a regular CPython build with numpy 1.x / 2.x; anywhere else (PyPy, a free-threaded or
    # debug interpreter with a larger object header, a numpy whose major version this was not checked against) reading it would
    # dereference garbage -- which no try / except catches -- so those keep the numpy evaluation
    if (sys.implementation.name != "cpython" or sysconfig.get_config_var("Py_GIL_DISABLED") or hasattr(sys, "gettotalrefcount")
            or int(np.__version__.split(".")[0]) not in (1, 2) or C.sizeof(C.c_void_p) != 8 or type(np.exp) is not np.ufunc
            or object.__basicsize__ != 16):
        raise RuntimeError("interpreter / numpy layout not known to this reader")

    class UFunc(C.Structure):
        _fields_ = [("ob_refcnt", C.c_ssize_t), ("ob_type", C.c_void_p), ("nin", C.c_int), ("nout", C.c_int), ("nargs", C.c_int),
                    ("identity", C.c_int), ("functions", C.POINTER(C.c_void_p)), ("data", C.POINTER(C.c_void_p)), ("ntypes", C.c_int),
                    ("reserved1", C.c_int), ("name", C.c_char_p), ("types", C.POINTER(C.c_char))]
    u = UFunc.from_address(id(np.exp))
    if (u.nin, u.nout, u.nargs, u.name) != (1, 1, 2, b"exp") or not 0 < u.ntypes < 64:
        raise RuntimeError("np.exp does not look like the ufunc object this was written for")
    NPY_DOUBLE = 12
    for i in range(u.ntypes):
        if ord(u.types[2 * i]) == NPY_DOUBLE and ord(u.types[2 * i + 1]) == NPY_DOUBLE:
            return u.functions[i], u.data[i]
    raise RuntimeError("np.exp has no float64 loop")


def _native():
    """(lib, loop, data) of the native evaluator after it reproduced the numpy evaluation bit for bit on random fits, else False"""
    global _NATIVE
    if _NATIVE is None:
        try:
            from .. import _lib as L
            lib = L.load()
            loop, data = _numpy_exp_loop()
            cand = (lib, loop, data)
            rng = np.random.default_rng(0)
            ok = True
            for n in (3, 6, 9):
                K, M = 7, 150
                lens = np.sort(rng.integers(M - 60, M + 1, K)).astype(np.int64)
                lens[-1] = M
                lb = np.tile([4.0, 0.001, 30.0] * (n // 3), (K, 1))
                ub = np.tile([2500.0, 3.0, 50.0] * (n // 3), (K, 1))
                X0 = np.clip(np.tile([100.0, 1.5, 40.0] * (n // 3), (K, 1)) * rng.uniform(0.3, 1.2, (K, n)), lb, ub)
                X0[0, 0], X0[1, 1] = ub[0, 0], lb[1, 1]               # on a bound: the step flips
                cols = np.arange(M, dtype=np.float64)
                xs = np.where(cols[None, :] < lens[:, None], cols[None, :], PAD_X)
                sig = rng.uniform(0, 2, (K, M)) * (cols[None, :] < lens[:, None])
                work = (np.empty((n + n // 3) * K * M), np.empty((n + 1) * K * M))
                f0, g0 = _evaluate(X0, lb, ub, sig, xs, lens, work)
                f1, g1 = _evaluate_native(cand, X0, lb, ub, sig, xs, lens)
                ok = ok and np.array_equal(f0, f1) and np.array_equal(g0, g1)
            _NATIVE = cand if ok else False
        except Exception:      # noqa: BLE001 -- no library, another numpy: the numpy evaluation stays
            _NATIVE = False
    return _NATIVE


def _evaluate_native(nat, X0, lb, ub, sig, xs, lens):
    """_evaluate through natac_fuzz_evaluate (csrc/natac_fuzzfit.hpp): the same doubles, one call"""
    import ctypes as C
    from .. import _lib as L
    lib, loop, data = nat
    K, n = X0.shape
    M = xs.shape[1]
    X0, lb, ub, sig, xs = (np.ascontiguousarray(a, dtype=np.float64) for a in (X0, lb, ub, sig, xs))
    lens = np.ascontiguousarray(lens, dtype=np.int64)
    work = np.empty((2 * n + n // 3 + 1) * K * M)
    f, g = np.empty(K), np.empty((K, n))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    L.check(lib.natac_fuzz_evaluate(K, n, M, vp(X0), vp(lb), vp(ub), vp(sig), vp(xs), vp(lens), C.c_void_p(loop), C.c_void_p(data), vp(work),
                                    vp(f), vp(g)))
    return f, g


def _run_group(probs, setulb):
    """K fits with the same parameter count to the end; returns x (K, n)"""
    K = len(probs)
    factr = FTOL / np.finfo(float).eps
    lens = np.array([len(p[0]) for p in probs], dtype=np.int64)
    M = int(lens.max())
    cols = np.arange(M, dtype=np.float64)
    sig = np.zeros((K, M))
    for k, p in enumerate(probs):
        sig[k, :lens[k]] = p[0]
    xs = np.where(cols[None, :] < lens[:, None], cols[None, :], PAD_X)
    lb = np.array([p[1] for p in probs])
    ub = np.array([p[2] for p in probs])
    states = [_State(p[3], p[1], p[2]) for p in probs]
    n = lb.shape[1]
    nat = _native() if NATIVE_EVAL else False
    work = None if nat else (np.empty((n + n // 3) * K * M), np.empty((n + 1) * K * M))
    active = [k for k in range(K) if _advance(states[k], setulb, factr)]
    while active:
        a = np.array(active)
        X0 = np.array([states[k].x for k in active])
        if nat:
            f, g = _evaluate_native(nat, X0, lb[a], ub[a], sig[a], xs[a], lens[a])
        else:
            f, g = _evaluate(X0, lb[a], ub[a], sig[a], xs[a], lens[a], work)
        nxt = []
        for i, k in enumerate(active):
            st = states[k]
            st.args[5] = f[i]
            st.args[6] = np.ascontiguousarray(g[i])
            if _advance(st, setulb):
                nxt.append(k)
        active = nxt
    x = np.array([st.x for st in states])
    for k, st in enumerate(states):
        if st.redo:
            x[k] = _public_fit(probs[k])
    return x


def _public_fit(prob):
    """one fit through scipy.optimize.minimize with the batched finite differences of fit_fuzz_one (same values)"""
    from scipy import optimize
    sig, lb, ub, guess, _left = prob
    vals = np.array(sig, dtype=np.float64)
    xs = np.linspace(0, len(vals) - 1, len(vals))

    def fun_and_grad(pars):
        f, g = _evaluate(np.asarray(pars, dtype=np.float64)[None, :], lb[None, :], ub[None, :], vals[None, :], xs[None, :],
                         np.array([len(vals)]), (np.empty(12 * len(vals)), np.empty(10 * len(vals))))
        return f[0], g[0]
    res = optimize.minimize(fun_and_grad, guess, jac=True, bounds=list(zip(lb, ub)), method="L-BFGS-B")
    return res["x"]


def fit_many(probs):
    """[(sig, lb, ub, guess, left)] -> (fuzz, weight, fit_pos) arrays, the values `fit_fuzz_one` returns per call"""
    from scipy.optimize import _lbfgsb
    for p in probs:
        if np.any(p[2] < p[1]):          # an all-zero window: scipy.optimize.minimize refuses it the same way
            raise ValueError("An upper bound is less than the corresponding lower bound.")
    out = np.empty((len(probs), 3))
    order = sorted(range(len(probs)), key=lambda k: (len(probs[k][3]), len(probs[k][0])))
    a = 0
    while a < len(order):
        n = len(probs[order[a]][3])
        e = a
        while e < len(order) and e - a < GROUP and len(probs[order[e]][3]) == n:
            e += 1
        ks = order[a:e]
        x = _run_group([probs[k] for k in ks], _lbfgsb.setulb)
        out[ks, 0] = np.sqrt(x[:, 0])
        out[ks, 1] = x[:, 1]
        out[ks, 2] = x[:, 2] + np.array([probs[k][4] for k in ks])
        a = e
    return out


_AVAILABLE = None


def available():
    """True when this scipy's private setulb drives a fit to the same bits as scipy.optimize.minimize (checked once)"""
    global _AVAILABLE
    if _AVAILABLE is None:
        try:
            from . import NucleosomeCalling as N
            x = np.arange(400, dtype=np.float64)
            vals = 1.3 * np.exp(-0.5 * ((x - 150) / 22.0) ** 2) + 0.8 * np.exp(-0.5 * ((x - 230) / 15.0) ** 2) - 0.01
            keys = [150, 230]
            ref = [N.fit_fuzz_one(vals.copy(), keys, k, 120, 10) for k in keys]
            got = fit_many([problem(vals.copy(), keys, k, 120, 10) for k in keys])
            _AVAILABLE = bool(np.array_equal(np.array(ref, dtype=np.float64), got))
        except Exception:      # noqa: BLE001 -- another scipy: the public path stays
            _AVAILABLE = False
    return _AVAILABLE
