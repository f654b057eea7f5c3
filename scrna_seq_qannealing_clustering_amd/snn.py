"""SNN-graph construction on the GPU: points -> exact kNN -> shared-neighbour (Jaccard) graph -> prune ->
zero diagonal -> the reference's sequential symmetric top-``ord`` trim.

This is the step the reference performs in R before its Python package ever runs
(`/root/reference/R/pbmc3k/Pbmc3k_prepare_data_for_QA_clustering.Rmd:67-79`: Seurat ``FindNeighbors`` +
the trimming loop; the result reaches Python as a GEXF file, `create_graphs.py:5-8`).  Everything numeric
runs in libmi_sa.so (csrc/snn_kernels.hip, C ABI include/mi_snn.h); there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib


class SnnGraph:
    """Result of :func:`build_snn`: CSR of shared-neighbour counts plus the kNN table."""

    def __init__(self, n, k, nn, rowptr, col, shared, timing, code=None, bonus=2.0, symmetric=True,
                 round_digits=None, negative_value=-0.3):
        self.n, self.k = int(n), int(k)
        self.nn, self.rowptr, self.col, self.shared = nn, rowptr, col, shared
        self.timing = timing
        self.code = code if code is not None else np.zeros(len(col), dtype=np.uint8)
        self.bonus = float(bonus)
        self.symmetric = bool(symmetric)     # False: the stored rows are the COLUMNS of an asymmetric matrix
        self.round_digits = round_digits     # the rounding variant: weights are round(w, digits); code 3 = a negative edge
        self.negative_value = float(negative_value)

    @property
    def weights(self) -> np.ndarray:
        """Jaccard weights ``s / (2k - s)`` in fp64 -- the values Seurat's SNN matrix holds -- with the notebooks'
        enhancement applied where the build asked for one (``w + bonus`` on mutual entries, ``w + w`` on doubled ones)."""
        s = self.shared.astype(np.float64)
        w = s / (2.0 * self.k - s)
        if self.round_digits is not None:                      # Pbmc3k_normalization_simulated_data.Rmd:599-605
            w = np.round(w, self.round_digits)
            return np.where(self.code == 3, self.negative_value, w)
        return np.where(self.code == 1, w + self.bonus, np.where(self.code == 2, w + w, w))

    @property
    def max_degree(self) -> int:
        return int(np.diff(self.rowptr).max()) if self.n else 0

    def edge_list(self):
        """``(nodes, eu, ev, w)``: string node ids '0'..'n-1' and the upper-triangular edges in row-major
        order -- what ``nx.from_numpy_matrix`` + the GEXF round trip of the notebooks produce."""
        rows = np.repeat(np.arange(self.n, dtype=np.int32), np.diff(self.rowptr))
        nodes = [str(i) for i in range(self.n)]
        w = self.weights
        if self.symmetric:
            up = self.col > rows
            return nodes, rows[up].astype(np.int32), self.col[up].astype(np.int32), w[up]
        # asymmetric result: entry e of stored row i with col r is A[r, i].  nx.from_numpy_matrix semantics
        # (graphs.edges_from_matrix): edge {u < v} carries A[v, u] when present, else A[u, v]; node u lists first the
        # v it met in its own row (A[u, v] != 0), then those it only learnt from their rows
        r, c = self.col.astype(np.int64), rows.astype(np.int64)              # A[r, c]
        n = self.n
        upper = r < c                                                          # entries A[u, v], u < v
        lower = ~upper                                                         # entries A[v, u], stored as (r = v, c = u)
        key_up = r[upper] * n + c[upper]
        key_lo = c[lower] * n + r[lower]                                       # as (u, v) with u < v
        both_keys = np.union1d(key_up, key_lo)
        in_up = np.isin(both_keys, key_up)
        w_up = dict(zip(key_up.tolist(), w[upper].tolist()))
        w_lo = dict(zip(key_lo.tolist(), w[lower].tolist()))
        u_all, v_all = both_keys // n, both_keys % n
        order = np.lexsort((v_all, ~in_up, u_all))                             # by u, own-row entries first, then v
        eu, ev = u_all[order], v_all[order]
        ww = np.array([w_lo[k] if k in w_lo else w_up[k] for k in both_keys[order].tolist()], dtype=np.float64)
        return nodes, eu.astype(np.int32), ev.astype(np.int32), ww

    def to_graph(self):
        from .graphs import EdgeListGraph
        return EdgeListGraph(*self.edge_list())


def build_snn(X: np.ndarray, k: int, prune: float = 0.0, ord: Optional[int] = None, device: int = 0,
              symmetric: bool = True, enhance: Optional[str] = None, mutual_bonus: float = 2.0,
              ord2: Optional[int] = None, round_digits: Optional[int] = None, negative_below: Optional[float] = None,
              negative_value: float = -0.3) -> SnnGraph:
    """``X``: (n, dim) coordinates (fp32 on the device), ``k`` = Seurat's ``k.param`` (self included),
    ``prune`` = ``prune.SNN``, ``ord`` = degree cap of the trim loop (None: no trim).

    The notebooks' optional chunks (`Pbmc3k_general_data_preparation.Rmd:77-123`, `Kidney_data.Rmd:235-266` -- the
    reference's ``..._trimmed_15enh.gexf`` inputs, `main.py:105-110`): ``symmetric=False`` = the UNSYMMETRIC first
    trim (columns only), ``enhance="mutual"`` adds ``mutual_bonus`` (2 in the PBMC notebook, 1 in the kidney one) to
    entries present in both directions, ``enhance="sum"`` forms ``A + t(A)``, ``ord2`` trims a second time.

    The rounding variant (`Pbmc3k_normalization_simulated_data.Rmd:597-616`): ``round_digits=2`` rounds the weights
    (``round(snn, digits=2)``) before the trim, which then ranks by the rounded values; ``negative_below=0.16`` with
    ``negative_value=-0.3`` is the notebook's "also negative edges" branch (``snn[snn < 0.16 & snn != 0] <- -0.3``):
    R's ``order(decreasing=TRUE)`` ranks those entries below the zeros, so a trim deletes them all (include/mi_snn.h);
    without a trim they stay in the graph with the negative weight."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    if X.ndim != 2:
        raise ValueError("X must be (n, dim)")
    n, dim = X.shape
    lib = _lib.load()
    h = C.c_void_p()
    if enhance not in (None, "mutual", "sum"):
        raise ValueError("enhance must be None, 'mutual' or 'sum'")
    flags = (0 if symmetric else 1) | (2 if enhance == "mutual" else 0) | (4 if enhance == "sum" else 0)
    if round_digits is None and negative_below is not None:
        raise ValueError("negative_below applies to the rounded weights: give round_digits")
    if round_digits is not None:
        if flags or ord2:
            raise ValueError("the rounding variant is the notebook's plain pipeline: symmetric trim, no enhancement")
        _lib.check(lib.mi_snn_build_rounded_f32(X.ctypes.data_as(C.POINTER(C.c_float)), n, dim, int(k), float(prune),
                                                int(ord or 0), int(round_digits), float(negative_below or 0.0),
                                                int(device), C.byref(h)))
    else:
        _lib.check(lib.mi_snn_build_ex_f32(X.ctypes.data_as(C.POINTER(C.c_float)), n, dim, int(k), float(prune),
                                           int(ord or 0), C.c_uint32(flags), float(mutual_bonus), int(ord2 or 0),
                                           int(device), C.byref(h)))
    try:
        nnz = C.c_int64(0)
        _lib.check(lib.mi_snn_info(h, None, None, C.byref(nnz), None))
        nn = np.empty((n, int(k)), dtype=np.int32)
        rowptr = np.empty(n + 1, dtype=np.int64)
        col = np.empty(int(nnz.value), dtype=np.int32)
        shared = np.empty(int(nnz.value), dtype=np.int32)
        code = np.zeros(int(nnz.value), dtype=np.uint8)
        i32p = C.POINTER(C.c_int32)
        _lib.check(lib.mi_snn_fetch(h, nn.ctypes.data_as(i32p), rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                    col.ctypes.data_as(i32p), shared.ctypes.data_as(i32p)))
        _lib.check(lib.mi_snn_fetch_codes(h, code.ctypes.data_as(C.POINTER(C.c_uint8))))
        t = [C.c_float(0.0), C.c_float(0.0), C.c_float(0.0)]
        _lib.check(lib.mi_snn_kernel_ms(h, C.byref(t[0]), C.byref(t[1]), C.byref(t[2])))
    finally:
        lib.mi_snn_destroy(h)
    return SnnGraph(n, k, nn, rowptr, col, shared,
                    {"knn_ms": t[0].value, "snn_ms": t[1].value, "trim_ms": t[2].value}, code=code, bonus=mutual_bonus,
                    symmetric=bool(symmetric) or enhance == "sum", round_digits=round_digits, negative_value=negative_value)
