"""Graph input: the reference's loaders plus a synthetic SNN generator.

* ``create_graph`` / ``create_graph_csv`` mirror `/root/reference/Python_Functions/create_graphs.py:5-18`
  (same return shapes; ``spring_layout`` is optional because it is O(n^2) per iteration and only
  feeds the plots).
* ``synthetic_snn`` builds a surrogate for the R pipeline that produces the reference's inputs
  (`R/pbmc3k/Pbmc3k_prepare_data_for_QA_clustering.Rmd:67-79`): Gaussian clusters -> exact kNN
  (k incl. self, as Seurat ``FindNeighbors(k.param=k)``) -> SNN Jaccard ``s/(2k-s)`` (``prune.SNN=0``)
  -> minus identity -> the sequential, symmetric, in-place top-``ord`` trim of :75-79.  No PBMC or
  kidney data ships with the reference, so the BASELINE configs run on these surrogates.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def create_graph(dir, layout: bool = True):
    """`create_graphs.py:5-8`: ``G = nx.read_gexf(dir); pos = nx.spring_layout(G)``."""
    import networkx as nx
    G = nx.read_gexf(dir)
    pos = nx.spring_layout(G) if layout else None
    return G, pos


def create_graph_csv(dirs, layout: bool = True):
    """`create_graphs.py:10-18`: CSV with columns 1,2,3 = u, v, weight."""
    import networkx as nx
    import pandas as pd
    input_data = pd.read_csv(dirs["graph_in_csv"], header=0, usecols=[1, 2, 3])
    records = input_data.to_records(index=False)
    G = nx.Graph()
    G.add_weighted_edges_from(list(records))
    pos = nx.spring_layout(G) if layout else None
    return G, pos


def graph_from_edges(nodes, eu, ev, w):
    """networkx graph with the given node order and edge order (fixtures, synthetic graphs)."""
    import networkx as nx
    G = nx.Graph()
    G.add_nodes_from(nodes)
    for a, b, ww in zip(eu, ev, w):
        G.add_edge(nodes[int(a)], nodes[int(b)], weight=float(ww))
    return G


def _knn_exact(X: np.ndarray, k: int, block: int = 2048) -> np.ndarray:
    """Indices of the k nearest points (self included, first) for every row; ties by index."""
    n = X.shape[0]
    sq = np.einsum("ij,ij->i", X, X)
    out = np.empty((n, k), dtype=np.int64)
    for s in range(0, n, block):
        e = min(n, s + block)
        d = sq[s:e, None] + sq[None, :] - 2.0 * (X[s:e] @ X.T)
        d[np.arange(e - s), np.arange(s, e)] = -1.0            # self is always nearest
        idx = np.argpartition(d, k - 1, axis=1)[:, :k]
        dd = np.take_along_axis(d, idx, axis=1)
        order = np.lexsort((idx, dd), axis=1)
        out[s:e] = np.take_along_axis(idx, order, axis=1)
    return out


def snn_from_points(X: np.ndarray, k: int, ord: Optional[int], symmetric: bool = True,
                    enhance: Optional[str] = None, bonus: float = 2.0, ord2: Optional[int] = None,
                    round_digits: Optional[int] = None, negative_below: Optional[float] = None,
                    negative_value: float = -0.3) -> np.ndarray:
    """Dense SNN weight matrix (zero diagonal), trimmed to degree <= ``ord`` if given.

    A LITERAL dense numpy restatement of the R lines (O(n^2) memory, Python loop over columns): it generates
    the small synthetic workloads of bench.py / the tests and pins oracle/snn_oracle.c.  It is not a fallback
    of the product path: graphs are BUILT by ``snn.build_snn`` (GPU, csrc/snn_kernels.hip), which raises when
    the HIP library is missing.  The optional arguments are the notebooks' optional chunks
    (`Pbmc3k_general_data_preparation.Rmd:77-123`): ``symmetric=False`` the UNSYMMETRIC first trim (:77-83),
    ``enhance="mutual"`` Method 2 (:85-101, ``+ bonus * mutual``), ``enhance="sum"`` ``A + t(A)`` (:103-113),
    ``ord2`` the second trim (:116-123); ``round_digits`` / ``negative_below`` / ``negative_value`` the rounding chunk
    of `Pbmc3k_normalization_simulated_data.Rmd:597-606` (``round(snn, digits=2)``; ``snn[snn < 0.16 & snn != 0] <-
    -0.3``) ahead of the trim."""
    n = X.shape[0]
    nn = _knn_exact(X, k)
    M = np.zeros((n, n), dtype=np.float32)
    M[np.repeat(np.arange(n), k), nn.reshape(-1)] = 1.0
    shared = (M @ M.T).astype(np.int64)                        # |N(i) & N(j)|
    snn = np.where(shared > 0, shared / (2.0 * k - shared), 0.0)
    np.fill_diagonal(snn, 0.0)                                 # snn - diag(n)  (Rmd :72)
    if round_digits is not None:                               # Pbmc3k_normalization_simulated_data.Rmd:599 / :602
        snn = np.round(snn, round_digits)
        if negative_below is not None:                         # :603-605
            snn[(snn < negative_below) & (snn != 0)] = negative_value

    def trim_symmetric(cap):
        # Rmd :75-79 -- for i in 1..n: to_delete = order(snn[,i], decreasing=TRUE)[(ord+1):n];
        # zero column i and row i there.  Sequential and in place; R's order() is stable.
        for i in range(n):
            colv = snn[:, i]
            order = np.argsort(-colv, kind="stable")
            to_delete = order[cap:]
            snn[to_delete, i] = 0.0
            snn[i, to_delete] = 0.0

    if ord is not None and symmetric:
        trim_symmetric(ord)
    elif ord is not None:
        for i in range(n):                                     # :77-83 -- the column only
            to_delete = np.argsort(-snn[:, i], kind="stable")[ord:]
            snn[to_delete, i] = 0.0
    if enhance == "mutual":                                    # :85-101
        mutual = (snn != 0) & (snn.T != 0)
        snn = snn + bonus * mutual
    elif enhance == "sum":                                     # :103-113
        snn = snn + snn.T
    elif enhance is not None:
        raise ValueError("enhance must be None, 'mutual' or 'sum'")
    if ord2 is not None:                                       # :116-123
        trim_symmetric(ord2)
    return snn


def edges_from_matrix(A: np.ndarray):
    """``(eu, ev, w)`` of ``nx.from_numpy_matrix(A)`` for a possibly ASYMMETRIC matrix, in ``G.edges`` order: the
    undirected edge {u, v}, u < v, exists when either entry is non-zero; it carries ``A[v, u]`` when that is non-zero
    (row v is visited after row u and overwrites the attribute), else ``A[u, v]``; node u reports first the
    neighbours v > u it met in its own row, then those it only learnt from their rows."""
    n = A.shape[0]
    up = np.triu(A, 1)
    lo = np.tril(A, -1).T                                      # lo[u, v] = A[v, u], u < v
    own = up != 0
    other = (~own) & (lo != 0)
    eu, ev, w = [], [], []
    for u in range(n):
        for mask in (own[u], other[u]):
            vs = np.nonzero(mask)[0]
            eu.extend([u] * len(vs))
            ev.extend(vs.tolist())
            w.extend(np.where(lo[u, vs] != 0, lo[u, vs], up[u, vs]).tolist())
    return np.asarray(eu, dtype=np.int32), np.asarray(ev, dtype=np.int32), np.asarray(w, dtype=np.float64)


def synthetic_snn(n: int = 2638, k: int = 5, dim: int = 15, ord: Optional[int] = 15,
                  n_clusters: int = 9, seed: int = 0, spread: float = 1.0,
                  proportions: Optional[np.ndarray] = None):
    """Surrogate "PBMC-like" SNN graph.  Returns ``(nodes, eu, ev, w, truth)`` with string node ids
    '0'..'n-1' (what the GEXF round trip of the R notebooks yields), edges in
    upper-triangular row-major order (``nx.from_numpy_matrix`` order) and the planted labels."""
    rng = np.random.RandomState(seed)
    if proportions is None:
        # PBMC3k-like cluster sizes (Seurat tutorial: 9 clusters from ~700 down to ~15 cells)
        base = np.array([0.26, 0.18, 0.17, 0.13, 0.10, 0.06, 0.06, 0.03, 0.01])
        proportions = np.resize(base, n_clusters)
        proportions = proportions / proportions.sum()
    sizes = np.floor(proportions * n).astype(int)
    sizes[0] += n - sizes.sum()
    centers = rng.normal(scale=4.0, size=(n_clusters, dim))
    truth = np.repeat(np.arange(n_clusters), sizes)
    X = centers[truth] + rng.normal(scale=spread, size=(n, dim))
    perm = rng.permutation(n)                                  # cells are not sorted by type
    X, truth = X[perm], truth[perm]
    snn = snn_from_points(X, k, ord)
    iu, ju = np.nonzero(np.triu(snn, 1))
    w = snn[iu, ju].astype(np.float64)
    nodes = [str(i) for i in range(n)]
    return nodes, iu.astype(np.int32), ju.astype(np.int32), w, truth


class EdgeListGraph:
    """Tiny stand-in for the part of ``nx.Graph`` the model builders use (``nodes``, ``edges(data=True)``,
    ``number_of_edges``, ``size(weight=)``) so that large synthetic graphs need no networkx object."""

    def __init__(self, nodes, eu, ev, w):
        self._nodes = list(nodes)
        self._eu, self._ev, self._w = np.asarray(eu), np.asarray(ev), np.asarray(w, dtype=np.float64)

    @property
    def nodes(self):
        return self._nodes

    def number_of_edges(self):
        return len(self._w)

    def edges(self, data=False):
        nd = self._nodes
        if data:
            return ((nd[int(a)], nd[int(b)], {"weight": float(c)})
                    for a, b, c in zip(self._eu, self._ev, self._w))
        return ((nd[int(a)], nd[int(b)]) for a, b in zip(self._eu, self._ev))

    def size(self, weight=None):
        # networkx: sum of weighted degrees / 2
        deg = np.zeros(len(self._nodes))
        np.add.at(deg, self._eu, self._w)
        np.add.at(deg, self._ev, self._w)
        return float(deg.sum() / 2.0) if weight else float(len(self._w))
