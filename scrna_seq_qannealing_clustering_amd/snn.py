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

    def __init__(self, n, k, nn, rowptr, col, shared, timing):
        self.n, self.k = int(n), int(k)
        self.nn, self.rowptr, self.col, self.shared = nn, rowptr, col, shared
        self.timing = timing

    @property
    def weights(self) -> np.ndarray:
        """Jaccard weights ``s / (2k - s)`` in fp64 -- the values Seurat's SNN matrix holds."""
        s = self.shared.astype(np.float64)
        return s / (2.0 * self.k - s)

    @property
    def max_degree(self) -> int:
        return int(np.diff(self.rowptr).max()) if self.n else 0

    def edge_list(self):
        """``(nodes, eu, ev, w)``: string node ids '0'..'n-1' and the upper-triangular edges in row-major
        order -- what ``nx.from_numpy_matrix`` + the GEXF round trip of the notebooks produce."""
        rows = np.repeat(np.arange(self.n, dtype=np.int32), np.diff(self.rowptr))
        up = self.col > rows
        return ([str(i) for i in range(self.n)], rows[up].astype(np.int32), self.col[up].astype(np.int32),
                self.weights[up])

    def to_graph(self):
        from .graphs import EdgeListGraph
        return EdgeListGraph(*self.edge_list())


def build_snn(X: np.ndarray, k: int, prune: float = 0.0, ord: Optional[int] = None, device: int = 0) -> SnnGraph:
    """``X``: (n, dim) coordinates (fp32 on the device), ``k`` = Seurat's ``k.param`` (self included),
    ``prune`` = ``prune.SNN``, ``ord`` = degree cap of the trim loop (None: no trim)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    if X.ndim != 2:
        raise ValueError("X must be (n, dim)")
    n, dim = X.shape
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.mi_snn_build_f32(X.ctypes.data_as(C.POINTER(C.c_float)), n, dim, int(k), float(prune),
                                    int(ord or 0), int(device), C.byref(h)))
    try:
        nnz = C.c_int64(0)
        _lib.check(lib.mi_snn_info(h, None, None, C.byref(nnz), None))
        nn = np.empty((n, int(k)), dtype=np.int32)
        rowptr = np.empty(n + 1, dtype=np.int64)
        col = np.empty(int(nnz.value), dtype=np.int32)
        shared = np.empty(int(nnz.value), dtype=np.int32)
        i32p = C.POINTER(C.c_int32)
        _lib.check(lib.mi_snn_fetch(h, nn.ctypes.data_as(i32p), rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                    col.ctypes.data_as(i32p), shared.ctypes.data_as(i32p)))
        t = [C.c_float(0.0), C.c_float(0.0), C.c_float(0.0)]
        _lib.check(lib.mi_snn_kernel_ms(h, C.byref(t[0]), C.byref(t[1]), C.byref(t[2])))
    finally:
        lib.mi_snn_destroy(h)
    return SnnGraph(n, k, nn, rowptr, col, shared,
                    {"knn_ms": t[0].value, "snn_ms": t[1].value, "trim_ms": t[2].value})
