"""ctypes wrapper over the SNN-construction restatement in oracle/snn_oracle.c.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import sa_oracle as _so


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def knn(X, k):
    """(n, k) int32: column 0 = the point itself, then its k-1 nearest others by (fp32 distance, index)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, dim = X.shape
    nn = np.empty((n, k), dtype=np.int32)
    rc = _so.lib().orc_knn_f32(_p(X, C.c_float), n, dim, int(k), _p(nn, C.c_int32))
    if rc:
        raise ValueError("orc_knn_f32 failed (k out of range?)")
    return nn


def snn_rows(nn, prune=0.0):
    """CSR (rowptr int64, col int32, shared int32) of s_ij = |N(i) & N(j)|, j != i, s/(2k-s) >= prune."""
    nn = np.ascontiguousarray(nn, dtype=np.int32)
    n, k = nn.shape
    rowptr = np.zeros(n + 1, dtype=np.int64)
    lib = _so.lib()
    lib.orc_snn_rows.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.orc_snn_rows(_p(nn, C.c_int32), n, k, float(prune), _p(rowptr, C.c_int64), None, None)
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    shared = np.empty(int(rowptr[-1]), dtype=np.int32)
    lib.orc_snn_rows(_p(nn, C.c_int32), n, k, float(prune), _p(rowptr, C.c_int64), _p(col, C.c_int32),
                     _p(shared, C.c_int32))
    return rowptr, col, shared


def trim(rowptr, col, shared, ord, alive=None):
    """alive mask (uint8 per stored entry) after the sequential symmetric top-`ord` trim (ranked by `shared`, any
    integer key), starting from `alive` (default: everything)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    alive = np.ones(len(col), dtype=np.uint8) if alive is None else np.ascontiguousarray(alive, dtype=np.uint8).copy()
    lib = _so.lib()
    lib.orc_snn_trim.argtypes = [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                                 C.POINTER(C.c_uint8)]
    lib.orc_snn_trim(len(rowptr) - 1, _p(rowptr, C.c_int64), _p(np.ascontiguousarray(col, dtype=np.int32), C.c_int32),
                     _p(np.ascontiguousarray(shared, dtype=np.int32), C.c_int32), int(ord or 0), _p(alive, C.c_uint8))
    return alive


def snn_graph(X, k, prune=0.0, ord=None):
    """Whole pipeline: returns (nn, rowptr, col, shared) of the trimmed graph (rows ascending by column)."""
    nn = knn(X, k)
    rowptr, col, shared = snn_rows(nn, prune)
    alive = trim(rowptr, col, shared, ord).astype(bool)
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    deg = np.bincount(rows[alive], minlength=len(rowptr) - 1)
    out_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    return nn, out_ptr, col[alive], shared[alive]


def snn_graph_variant(X, k, prune=0.0, ord=None, symmetric=True, enhance=None, bonus=2.0, ord2=None):
    """The notebooks' optional chunks on top of the base construction: unsymmetric first trim, "enhance shared edges"
    (mutual bonus / A + t(A)), second trim.  Returns (nn, rowptr, col, shared, code): the stored rows are the COLUMNS
    of the result (entry e of row i with col[e] = r is A[r, i]); code 0 = w, 1 = w + bonus, 2 = w + w."""
    lib = _so.lib()
    nn = knn(X, k)
    rowptr, col, shared = snn_rows(nn, prune)
    n = len(rowptr) - 1
    alive = np.ones(len(col), dtype=np.uint8)
    if ord:
        if symmetric:
            alive = trim(rowptr, col, shared, ord)
        else:
            lib.orc_snn_trim_cols.argtypes = [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_uint8)]
            lib.orc_snn_trim_cols(n, _p(rowptr, C.c_int64), _p(shared, C.c_int32), int(ord), _p(alive, C.c_uint8))
    code = np.zeros(len(col), dtype=np.uint8)
    if enhance is not None:
        out = np.zeros_like(alive)
        lib.orc_snn_enhance.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                        C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        lib.orc_snn_enhance(n, 1 if enhance == "mutual" else 2, _p(rowptr, C.c_int64), _p(col, C.c_int32),
                            _p(alive, C.c_uint8), _p(out, C.c_uint8), _p(code, C.c_uint8))
        alive = out
    if ord2:
        w = shared / (2.0 * k - shared)
        val = np.where(code == 1, w + bonus, np.where(code == 2, w + w, w))
        key = np.searchsorted(np.unique(val), val).astype(np.int32)       # ranks: the trim only compares
        alive = trim(rowptr, col, key, ord2, alive=alive)
    alive = alive.astype(bool)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    deg = np.bincount(rows[alive], minlength=n)
    out_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    return nn, out_ptr, col[alive], shared[alive], code[alive]


def snn_graph_rounded(X, k, prune=0.0, ord=None, round_digits=2, negative_below=None):
    """The rounding chunk of Pbmc3k_normalization_simulated_data.Rmd:597-616 on the CSR form: the trim ranks by
    round(s / (2k - s), digits) (dense ranks; ties by row index as R's stable order()); entries whose rounded weight is
    below `negative_below` (the notebook's negative edges) sort below the zeros of their column in R, so a trim deletes
    them -- they leave before it -- provided every column has at least `ord` non-negative positions (asserted); without
    a trim they stay, code 3.  Returns (nn, rowptr, col, shared, code)."""
    nn = knn(X, k)
    rowptr, col, shared = snn_rows(nn, prune)
    n = len(rowptr) - 1
    wr = np.round(shared / (2.0 * k - shared), round_digits)
    neg = (wr < negative_below) & (wr != 0) if negative_below else np.zeros(len(col), dtype=bool)
    key = np.searchsorted(np.unique(np.round(np.arange(k + 1) / (2.0 * k - np.arange(k + 1)), round_digits)), wr).astype(np.int32)
    code = np.where(neg, 3, 0).astype(np.uint8)
    alive = np.ones(len(col), dtype=np.uint8)
    if ord:
        if neg.any():
            assert n - int(np.diff(rowptr).max()) >= ord, "negative entries could survive the trim on a graph this small"
            alive = (~neg).astype(np.uint8)
        alive = trim(rowptr, col, key, ord, alive=alive)
    alive = alive.astype(bool)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    deg = np.bincount(rows[alive], minlength=n)
    out_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    return nn, out_ptr, col[alive], shared[alive], code[alive]
