"""ctypes wrapper of oracle/libkgcn_ref.so (the C restatement).  TEST INFRASTRUCTURE: only
tests/, smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libkgcn_ref.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("%s missing: run `make -C oracle` (or __graft_entry__.build())" % _PATH)
        _lib = ctypes.CDLL(_PATH)
        _lib.kgcn_ref_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return load().kgcn_ref_max_threads()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def flatten_coo(adjs_ch):
    """adjs_ch: list over graphs of (idx, val, shape) -> (off int64[T+1], idx int32[nnz,2], val f32)."""
    off = np.zeros(len(adjs_ch) + 1, np.int64)
    idx, val = [], []
    for t, (i, v, _) in enumerate(adjs_ch):
        i = np.asarray(i).reshape(-1, 2)
        off[t + 1] = off[t] + i.shape[0]
        idx.append(i.astype(np.int32))
        val.append(np.asarray(v, np.float32))
    return (off, np.ascontiguousarray(np.concatenate(idx) if idx else np.zeros((0, 2), np.int32)),
            np.ascontiguousarray(np.concatenate(val) if val else np.zeros(0, np.float32)))


def graphconv_fwd(off, idx, val, x, w, bias, nthreads=0):
    T, n, din = x.shape
    dout = w.shape[1]
    x, w = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32).reshape(-1)
    out = np.empty((T, n, dout), np.float32)
    load().kgcn_ref_graphconv_fwd(T, n, din, dout, _p(off), _p(idx), _p(val), _p(x), _p(w), _p(b),
                                  _p(out), 0, int(nthreads))
    return out


def graphconv_bwd(off, idx, val, x, w, g, nthreads=0, want_dx=True):
    T, n, din = x.shape
    dout = w.shape[1]
    x, w, g = (np.ascontiguousarray(a, np.float32) for a in (x, w, g))
    dx = np.empty_like(x) if want_dx else None
    dw = np.empty((din, dout), np.float32)
    db = np.empty((1, dout), np.float32)
    load().kgcn_ref_graphconv_bwd(T, n, din, dout, _p(off), _p(idx), _p(val), _p(x), _p(w), _p(g),
                                  _p(dx), _p(dw), _p(db), 0, int(nthreads))
    return dx, dw, db


def bspmm(off, idx, val, rhs, m, k, adjoint_a=False, nthreads=0):
    T, _, d = rhs.shape
    rhs = np.ascontiguousarray(rhs, np.float32)
    out = np.empty((T, k if adjoint_a else m, d), np.float32)
    load().kgcn_ref_bspmm(T, m, k, d, _p(off), _p(idx), _p(val), _p(rhs), _p(out), int(adjoint_a),
                          int(nthreads))
    return out


def dense_fwd(x, w, bias=None, act=0, nthreads=0):
    """y = act(x @ w + bias) on [m, din] rows (act: 0 none, 1 sigmoid, 2 relu, 3 tanh)."""
    x, w = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(w, np.float32)
    m, din = x.shape
    dout = w.shape[1]
    b = None if bias is None else np.ascontiguousarray(bias, np.float32).reshape(-1)
    y = np.empty((m, dout), np.float32)
    load().kgcn_ref_dense_fwd(ctypes.c_int64(m), din, dout, _p(x), _p(w), _p(b), int(act), _p(y), int(nthreads))
    return y


def dense_bwd(x, w, y, g, act=0, want_dx=True, nthreads=0):
    """-> dx [m, din] f32 (or None), dw [din, dout] f64, db [dout] f64 (sums over the m rows accumulated in fp64)."""
    x, w, y, g = (np.ascontiguousarray(a, np.float32) for a in (x, w, y, g))
    m, din = x.shape
    dout = w.shape[1]
    dx = np.empty_like(x) if want_dx else None
    dw = np.empty((din, dout), np.float64)
    db = np.empty((dout,), np.float64)
    load().kgcn_ref_dense_bwd(ctypes.c_int64(m), din, dout, _p(x), _p(w), _p(y), _p(g), int(act), _p(dx), _p(dw), _p(db),
                              int(nthreads))
    return dx, dw, db


def gin_aggregate(off, idx, val, x, eps, adjoint=False, dot_with=None, nthreads=0):
    """out[t] = eps x[t] + op(A[t]) x[t]; with dot_with=g also returns <x, g> in fp64 (d eps)."""
    x = np.ascontiguousarray(x, np.float32)
    T, n, d = x.shape
    out = np.empty_like(x)
    g2 = None if dot_with is None else np.ascontiguousarray(dot_with, np.float32)
    dot = ctypes.c_double(0.0)
    load().kgcn_ref_gin_aggregate(T, n, d, _p(off), _p(idx), _p(val), _p(x), ctypes.c_float(eps), int(adjoint), _p(out),
                                  _p(g2), ctypes.byref(dot) if g2 is not None else None, int(nthreads))
    return (out, dot.value) if g2 is not None else out
