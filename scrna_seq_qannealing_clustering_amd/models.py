"""Model builders: SNN graph -> graph-partition QUBO / Potts-DQM in array form.

These are the host-side counterparts of the reference's in-line Python model construction
(SURVEY.md section 8a rows A1-A4).  The reference accumulates a ``defaultdict`` of n(n+1)/2 Python
floats (`/root/reference/Python_Functions/BQM_clustering.py:36-47`) or O(n^2 K) ``set_quadratic``
dicts (`DQM_clustering.py:36-43`); here the same coefficients are produced directly as numpy arrays
in the layouts the HIP kernels consume (dense symmetric ``Qs`` and/or CSR sparse part + uniform pair
coefficient), so nothing quadratic in n is ever built in interpreted Python.

Array convention (used everywhere): ``Qs`` is the symmetrised n x n matrix, ``Qs[i,i] = Q[i,i]``,
``Qs[i,j] = Qs[j,i] = (Q[i,j] + Q[j,i]) / 2``; then ``E(x) = x^T Qs x + offset`` for binary x.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Hashable, List, Optional, Sequence, Tuple

import numpy as np


# --------------------------------------------------------------------------------------------
# graph -> arrays
# --------------------------------------------------------------------------------------------
def _is_simple_networkx_graph(G) -> bool:
    return (hasattr(G, "adj") and hasattr(G, "is_directed") and hasattr(G, "is_multigraph")
            and not G.is_directed() and not G.is_multigraph())


class RootGraphArrays:
    """The adjacency of a simple networkx graph as arrays, in adjacency order -- built by ONE walk, then reused for every
    ``G.subgraph(part)`` view a recursive bisection takes of it (BQM_clustering.py:121-122; networkx collapses a
    subgraph of a subgraph view into a view of the root graph with a new node set).  ``arrays_for(view)`` returns what
    ``graph_arrays_and_weight(view)`` returns, entry for entry and bit for bit, without walking the view in Python:

    * nodes: ``list(view.nodes)`` -- networkx's own iteration (it depends on the size of the node set);
    * a neighbour dict of a view iterates the ROOT's dict in its order, filtered by the node set, unless the set has
      less than half as many nodes as the dict has entries -- then (tiny subgraphs only) this class steps aside;
    * an edge is reported at the endpoint that comes first in node order (the ``seen`` set of ``G.edges``);
    * ``W = G.size(weight)``: per-node weighted degree added up in adjacency order (``np.bincount`` adds its weights
      one after the other, as the Python loop does; a self-loop twice), then Python's ``sum`` over the nodes, halved.

    The edge weights of the root must not change while the object is in use (a clustering run does not change them)."""

    def __init__(self, root):
        self.graph = root
        nodes = list(root._adj)
        self.index = {v: i for i, v in enumerate(nodes)}
        index = self.index
        counts = np.fromiter((len(nbrs) for nbrs in root._adj.values()), dtype=np.int64, count=len(nodes))
        self.rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        col: List[int] = []
        w: List[float] = []
        self.ok = True
        for nbrs in root._adj.values():
            for v, data in nbrs.items():
                col.append(index[v])
                wt = data.get("weight")
                if wt is None or isinstance(wt, bool) or not isinstance(wt, (int, float)):
                    self.ok = False                      # (unweighted / exotic weights: the plain walk decides what happens)
                    wt = 0.0
                w.append(wt)
        self.col = np.asarray(col, dtype=np.int64)
        self.w = np.asarray(w, dtype=np.float64)
        self.max_degree = int(counts.max()) if len(counts) else 0

    @staticmethod
    def of(G):
        """For a simple networkx graph or a node-induced view of one; None for anything else."""
        if not _is_simple_networkx_graph(G):
            return None
        root = G._graph if hasattr(G, "_NODE_OK") else G
        if hasattr(root, "_NODE_OK") or not hasattr(root, "_adj") or not _is_simple_networkx_graph(root):
            return None
        arrays = RootGraphArrays(root)
        return arrays if arrays.ok else None

    def arrays_for(self, G):
        if G is self.graph:
            nodes = list(G.nodes)
        else:
            try:
                import networkx as nx
                ok = (G._graph is self.graph and G._EDGE_OK is nx.filters.no_filter and hasattr(G._NODE_OK, "nodes"))
            except Exception:
                ok = False
            if not ok or 2 * len(G._NODE_OK.nodes) < self.max_degree:
                return None
            nodes = list(G.nodes)
        n = len(nodes)
        if n == 0:
            return None
        root_idx = np.fromiter((self.index[v] for v in nodes), dtype=np.int64, count=n)
        pos = np.full(len(self.index), -1, dtype=np.int64)
        pos[root_idx] = np.arange(n)
        start, cnt = self.rowptr[root_idx], self.rowptr[root_idx + 1] - self.rowptr[root_idx]
        total = int(cnt.sum())
        row = np.repeat(np.arange(n), cnt)
        flat = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt) + np.repeat(start, cnt)
        nb = pos[self.col[flat]]
        keep = nb >= 0
        row, nb, ww = row[keep], nb[keep], self.w[flat][keep]
        twice = np.where(nb == row, 2, 1)                                   # a self-loop enters its degree twice, back to back
        deg = np.bincount(np.repeat(row, twice), weights=np.repeat(ww, twice), minlength=n)
        W = float(sum(deg.tolist()) / 2)
        first = nb >= row                                                   # the other endpoint has not been visited yet
        return (nodes, row[first].astype(np.int32), nb[first].astype(np.int32), ww[first], W)


def graph_arrays_and_weight(G, arrays: Optional["RootGraphArrays"] = None):
    """``(nodes, eu, ev, w, W)``: the arrays of ``graph_arrays`` and ``W = G.size(weight="weight")`` from ONE walk over the
    adjacency.  On a networkx graph the three calls the reference makes (``number_of_edges``, ``edges``, ``size``) are
    three walks, and on the subgraph VIEWS the recursive bisection hands down (``G.subgraph(part)``,
    BQM_clustering.py:121-122) every step of a walk goes through the view's node filter: 2/3 of the model-build time
    of a 4-level bisection.  Same edge order as ``G.edges`` (an edge is reported at its first endpoint in node order,
    neighbours in adjacency order) and the same summation as ``Graph.size`` (per-node weighted degree in adjacency
    order, self-loops twice, summed over the nodes with ``sum``, halved), so gamma stays bit-identical.
    ``arrays``: the root graph's ``RootGraphArrays`` -- the same result from array operations (a clustering run builds
    it once and hands it down its recursion)."""
    if arrays is not None:
        got = arrays.arrays_for(G)
        if got is not None:
            return got
    if not _is_simple_networkx_graph(G):
        nodes, eu, ev, w = _graph_arrays_by_calls(G)
        return nodes, eu, ev, w, _graph_total_weight(G, w)
    nodes = list(G.nodes)
    index = {v: i for i, v in enumerate(nodes)}
    eu: List[int] = []
    ev: List[int] = []
    w: List[float] = []
    degs = []
    seen = set()
    for u, nbrs in G.adj.items():
        iu = index[u]
        deg = 0
        for v, data in nbrs.items():
            deg += data.get("weight", 1)
            if v not in seen:
                eu.append(iu)
                ev.append(index[v])
                w.append(data["weight"])
            if v == u:
                deg += data.get("weight", 1)
        degs.append(deg)
        seen.add(u)
    W = float(sum(degs) / 2)
    return (nodes, np.asarray(eu, dtype=np.int32), np.asarray(ev, dtype=np.int32),
            np.asarray(w, dtype=np.float64), W)


def _graph_arrays_by_calls(G):
    nodes = list(G.nodes)
    index = {v: i for i, v in enumerate(nodes)}
    m = G.number_of_edges()
    eu = np.empty(m, dtype=np.int32)
    ev = np.empty(m, dtype=np.int32)
    w = np.empty(m, dtype=np.float64)
    for k, (u, v, data) in enumerate(G.edges(data=True)):
        eu[k] = index[u]
        ev[k] = index[v]
        w[k] = data["weight"]
    return nodes, eu, ev, w


def graph_arrays(G) -> Tuple[List[Hashable], np.ndarray, np.ndarray, np.ndarray]:
    """``(nodes, eu, ev, w)`` from a networkx graph, in ``G.nodes`` / ``G.edges`` order -- the orders
    the reference's loops iterate in (BQM_clustering.py:38,43,46; DQM_clustering.py:30,36,40)."""
    return graph_arrays_and_weight(G)[:4] if _is_simple_networkx_graph(G) else _graph_arrays_by_calls(G)


def _graph_total_weight(G, w: np.ndarray) -> float:
    """``G.size(weight="weight")`` (BQM_clustering.py:29).  networkx computes it as
    ``sum(degree(weight)) / 2``; call it when available so gamma is bit-identical."""
    try:
        return float(G.size(weight="weight"))
    except Exception:  # plain edge-list objects
        return float(np.sum(w))


def _csr_from_edges(n: int, eu: np.ndarray, ev: np.ndarray, val: np.ndarray):
    """Symmetric CSR (both directions stored), neighbours in ascending column order, duplicate
    (u,v) pairs summed.  Returns (rowptr int32[n+1], col int32[nnz], val float64[nnz])."""
    rows = np.concatenate([eu, ev]).astype(np.int64)
    cols = np.concatenate([ev, eu]).astype(np.int64)
    vals = np.concatenate([val, val]).astype(np.float64)
    keep = rows != cols
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    key = rows * n + cols
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    if len(key):
        start = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))       # first entry of every (row, col) run
        uniq = key[start]
        summed = np.add.reduceat(vals, start)
    else:
        uniq, summed = key, vals
    r = (uniq // n).astype(np.int32)
    c = (uniq % n).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
    return rowptr, c, summed


@dataclass
class QuboModel:
    """Binary quadratic model in array form.  ``E(x) = sum_i lin_i x_i + sum_{i<j} (c_pair + S_ij) x_i x_j
    + offset`` with S sparse (CSR, both directions, holding the *upper-triangular coefficient* of the
    pair, i.e. ``Q[i,j] + Q[j,i] - c_pair``).  ``dense_Qs()`` expands to the symmetric n x n form."""
    variables: List[Hashable]
    lin: np.ndarray                      # float64[n]   Q[i,i]
    rowptr: np.ndarray                   # int32[n+1]
    col: np.ndarray                      # int32[nnz]
    val: np.ndarray                      # float64[nnz] pair coefficient minus c_pair
    c_pair: float = 0.0                  # uniform coefficient on every pair i<j
    offset: float = 0.0
    info: Dict[str, Any] = field(default_factory=dict)
    _dense: Optional[np.ndarray] = None  # set when the model was given densely (general Q)
    # positive integer weights a_i of the pair term: the coefficient of pair i<j is c_pair a_i a_j + S_ij (None: all 1).
    # A squared linear constraint with slack bits has this shape (add_size_window_penalty); the structured kernels take
    # it (include/mi_sa.h: mi_sa_problem_set_pair_weights) when the weights other than 1 are few and have no edges.
    weights: Optional[np.ndarray] = None

    @property
    def num_variables(self) -> int:
        return len(self.variables)

    def dense_Qs(self, dtype=np.float64) -> np.ndarray:
        if self._dense is not None:
            return self._dense.astype(dtype, copy=False)
        n = self.num_variables
        if self.weights is None:
            Qs = np.full((n, n), 0.5 * self.c_pair, dtype=np.float64)
        else:
            a = np.asarray(self.weights, dtype=np.float64)
            Qs = 0.5 * self.c_pair * np.outer(a, a)
        rows = np.repeat(np.arange(n), np.diff(self.rowptr))
        Qs[rows, self.col] += 0.5 * self.val
        Qs[np.arange(n), np.arange(n)] = self.lin
        return Qs.astype(dtype, copy=False)

    def to_qubo_dict(self) -> Dict[Tuple[Hashable, Hashable], float]:
        """Upper-triangular dict form (small n only) -- the type the reference hands to
        ``sampler.sample_qubo`` (BQM_clustering.py:57)."""
        Qs = self.dense_Qs()
        v = self.variables
        n = len(v)
        out = {}
        for i in range(n):
            out[(v[i], v[i])] = float(Qs[i, i])
            for j in range(i + 1, n):
                c = 2.0 * float(Qs[i, j])
                if c != 0.0:
                    out[(v[i], v[j])] = c
        return out

    def energies(self, X: np.ndarray) -> np.ndarray:
        """fp64 energies of states X (R x n, 0/1) using the structured form (no n^2 work)."""
        X = np.asarray(X)
        Xf = X.astype(np.float64)
        if self._dense is not None:
            return np.einsum("ri,ri->r", Xf @ self._dense, Xf) + self.offset
        if self.weights is None:
            s = Xf.sum(axis=1)
            e = Xf @ self.lin + self.c_pair * 0.5 * s * (s - 1.0) + self.offset
        else:                                                   # c sum_{i<j} a_i a_j x_i x_j = c/2 ((a.x)^2 - a^2.x)
            a = np.asarray(self.weights, dtype=np.float64)
            e = Xf @ self.lin + self.c_pair * 0.5 * ((Xf @ a) ** 2 - Xf @ (a * a)) + self.offset
        rows = np.repeat(np.arange(self.num_variables), np.diff(self.rowptr))
        e += 0.5 * np.einsum("re,re,e->r", Xf[:, rows], Xf[:, self.col], self.val)
        return e


def _cut_qubo_parts(n, eu, ev, w, k):
    """Per-edge ``Q[u,u]+=k w; Q[v,v]+=k w; Q[u,v]+=-2 k w`` (BQM_clustering.py:38-41, :230-233,
    :366-369) as (lin, pair-coefficient per edge)."""
    lin = np.zeros(n, dtype=np.float64)
    np.add.at(lin, eu, k * w)
    np.add.at(lin, ev, k * w)
    return lin, k * -2 * w


def build_bqm_qubo(G, gamma_factor: float, k: float = 8, arrays=None) -> QuboModel:
    """A1 -- `clustering_bqm` model, BQM_clustering.py:29-47:
    ``gamma = gamma_factor * W / n``; cut term with ``k = 8`` (:33); ``Q[i,i] += gamma (1 - n)``
    (:43-44); ``Q[i,j] += 2 gamma`` for every pair (:46-47).  Closed form
    ``E = k cut_w + gamma (s^2 - n s)``.  ``arrays``: see ``graph_arrays_and_weight``."""
    nodes, eu, ev, w, W = graph_arrays_and_weight(G, arrays)
    n = len(nodes)
    gamma = gamma_factor * W / n
    lin, pair = _cut_qubo_parts(n, eu, ev, w, k)
    lin = lin + gamma * (1 - n)
    rowptr, col, val = _csr_from_edges(n, eu, ev, pair)
    return QuboModel(nodes, lin, rowptr, col, val, c_pair=2 * gamma, offset=0.0,
                     info={"gamma": gamma, "k": k, "W": W, "kind": "bqm"})


def build_bqm2_qubo(G, gamma_factor: float, k: float, arrays=None) -> QuboModel:
    """A2 -- `clustering_bqm_2` model, BQM_clustering.py:210-236: ``gamma = (W / n) gamma_factor``
    (:222), cut term with caller's k (:230-233), linear-only penalty ``Q[i,i] += gamma`` (:235-236);
    also the QPU-only ``chain_strength = mean(w) mean(deg) 2`` (:212-220) for reporting."""
    nodes, eu, ev, w, W = graph_arrays_and_weight(G, arrays)
    n = len(nodes)
    gamma = (W / n) * gamma_factor
    lin, pair = _cut_qubo_parts(n, eu, ev, w, k)
    lin = lin + gamma
    rowptr, col, val = _csr_from_edges(n, eu, ev, pair)
    deg = np.zeros(n, dtype=np.int64)
    np.add.at(deg, eu, 1)
    np.add.at(deg, ev, 1)
    chain_strength = float(np.mean(w)) * float(np.mean(deg)) * 2 if len(w) else 0.0
    return QuboModel(nodes, lin, rowptr, col, val, c_pair=0.0, offset=0.0,
                     info={"gamma": gamma, "k": k, "W": W, "chain_strength": chain_strength,
                           "kind": "bqm_2"})


def build_bqm3_cut_qubo(G, k: float = 8) -> QuboModel:
    """A3 (QUBO part) -- `clustering_bqm_3`, BQM_clustering.py:363-369: cut term only.  The slack
    inequality of :373-380 is added by ``add_size_window_penalty``."""
    nodes, eu, ev, w = graph_arrays(G)
    n = len(nodes)
    lin, pair = _cut_qubo_parts(n, eu, ev, w, k)
    rowptr, col, val = _csr_from_edges(n, eu, ev, pair)
    return QuboModel(nodes, lin, rowptr, col, val, c_pair=0.0, offset=0.0,
                     info={"k": k, "kind": "bqm_3"})


# --------------------------------------------------------------------------------------------
# Potts / DQM
# --------------------------------------------------------------------------------------------
@dataclass
class PottsModel:
    """k-way model ``E(l) = lin_offset + sum_{u<v, l_u == l_v} (c_pair + S_uv)`` -- the form
    `clustering_dqm` (DQM_clustering.py:29-43) reduces to after its ``set_`` overwrites: the linear
    biases are case-independent (a constant under one-hot) and every pair couples equal cases only."""
    variables: List[Hashable]
    num_cases: int
    rowptr: np.ndarray
    col: np.ndarray
    val: np.ndarray                      # float64[nnz]  B_uv - c_pair on stored pairs
    c_pair: float
    lin: np.ndarray                      # float64[n]    case-independent linear bias per variable
    info: Dict[str, Any] = field(default_factory=dict)

    @property
    def num_variables(self) -> int:
        return len(self.variables)

    @property
    def lin_offset(self) -> float:
        return float(np.sum(self.lin))

    def energies(self, L: np.ndarray) -> np.ndarray:
        L = np.asarray(L).astype(np.int64)
        R, n = L.shape
        rows = np.repeat(np.arange(n), np.diff(self.rowptr))
        same = (L[:, rows] == L[:, self.col])
        e = 0.5 * (same * self.val[None, :]).sum(axis=1)
        for r in range(R):
            cnt = np.bincount(L[r], minlength=self.num_cases).astype(np.float64)
            e[r] += self.c_pair * 0.5 * float(np.sum(cnt * (cnt - 1.0)))
        return e + self.lin_offset


def build_dqm_potts(G, num_of_clusters: int, gamma: float) -> PottsModel:
    """A4 -- `clustering_dqm` model, DQM_clustering.py:29-43, in Potts form.

    After the reference's overwrites: ``lin[v][c]`` = weight of the LAST edge in ``G.edges`` order
    touching v (:42-43), or ``gamma (1 - n/K)`` for isolated nodes (:33-34); pair bias on equal cases
    ``-2 w_uv`` on edges (:41, overwriting the ``2 gamma`` of :36-37), ``2 gamma`` elsewhere."""
    nodes, eu, ev, w = graph_arrays(G)
    n = len(nodes)
    K = int(num_of_clusters)
    lin = np.full(n, gamma * (1 - n / K), dtype=np.float64)
    for a, b, ww in zip(eu.tolist(), ev.tolist(), w.tolist()):   # order matters: last write wins
        lin[a] = ww
        lin[b] = ww
    # pair bias on an edge is SET to -2w (last edge wins if a pair repeats; nx.Graph has no repeats)
    pair = -2.0 * w - 2.0 * gamma
    rowptr, col, val = _csr_from_edges(n, eu, ev, pair)
    return PottsModel(nodes, K, rowptr, col, val, c_pair=2.0 * gamma, lin=lin,
                      info={"gamma": gamma, "kind": "dqm"})


def build_cqm_potts(G, num_of_clusters: int, min_cluster_size: int = 20) -> PottsModel:
    """`clustering_cqm` -- CQM_clustering.py:30-48 -- in Potts form.  The reference's objective is
    ``sum_{(i,j) in E} sum_p [v_ip + v_jp - 2 w_ij v_ip v_jp]`` (:40-44) under one-hot ``add_discrete``
    constraints per node (:36-38): with exactly one case set per node the linear part is the constant
    ``2 |E|`` and the rest is ``-2 w_ij`` for every edge inside a cluster.  The ``>= 20`` members per cluster
    constraints (:46-48) are carried as ``info["min_cluster_size"]`` and enforced by the sampler as a hard
    constraint on the moves (mi_sa.h "min_cluster_size")."""
    nodes, eu, ev, w = graph_arrays(G)
    n = len(nodes)
    rowptr, col, val = _csr_from_edges(n, eu, ev, -2.0 * w)
    lin = np.zeros(n, dtype=np.float64)
    if n:
        lin[0] = 2.0 * len(w)                                   # the constant, kept where lin_offset sums it
    return PottsModel(nodes, int(num_of_clusters), rowptr, col, val, c_pair=0.0, lin=lin,
                      info={"kind": "cqm", "min_cluster_size": int(min_cluster_size)})


def build_subsampling_qubo(G, gamma: float, P: float = 1.0) -> QuboModel:
    """`graph_subsampling` -- QA_subsampling.py:28-35: ``Q[u,u] += -P (1 - w)``, ``Q[v,v] += -P (1 - w)``,
    ``Q[u,v] += P (1 - w)`` per edge, ``Q[i,i] += gamma`` per node.  Sparse, no uniform pair term."""
    nodes, eu, ev, w = graph_arrays(G)
    n = len(nodes)
    lin = np.full(n, float(gamma), dtype=np.float64)
    np.add.at(lin, eu, -P * (1.0 - w))
    np.add.at(lin, ev, -P * (1.0 - w))
    rowptr, col, val = _csr_from_edges(n, eu, ev, P * (1.0 - w))
    return QuboModel(nodes, lin, rowptr, col, val, c_pair=0.0, offset=0.0, info={"kind": "subsampling", "gamma": gamma})


def build_mis_qubo(G, lagrange: float = 2.0) -> QuboModel:
    """The QUBO ``dwave_networkx.maximum_independent_set`` hands to its sampler (QA_subsampling.py:102 calls
    it; dwave_networkx is an absent third-party package, its published formulation is restated): ``-1`` on
    every node, ``+lagrange`` on every edge."""
    nodes, eu, ev, w = graph_arrays(G)
    n = len(nodes)
    rowptr, col, val = _csr_from_edges(n, eu, ev, np.full(len(w), float(lagrange)))
    return QuboModel(nodes, np.full(n, -1.0), rowptr, col, val, c_pair=0.0, offset=0.0,
                     info={"kind": "mis", "lagrange": lagrange})


# --------------------------------------------------------------------------------------------
# generic Q dict -> arrays (any caller of sample_qubo: QA_subsampling.py:28-35, other_tools.py:62 ...)
# --------------------------------------------------------------------------------------------
def qubo_dict_to_model(Q: Dict[Tuple[Hashable, Hashable], float], offset: float = 0.0,
                       detect_uniform: bool = True) -> QuboModel:
    """Lift a dimod-style QUBO dict into array form.  Variable order = first appearance in the
    dict's iteration order (u before v), as ``dimod.BinaryQuadraticModel.from_qubo`` would add them.
    If at least half of all pairs carry the same non-zero coefficient it is split off as the
    uniform pair term ``c_pair`` (the ``2 gamma`` of BQM_clustering.py:46-47), leaving S sparse."""
    # the dict has n(n+1)/2 entries in the reference's models (3.5 M at n = 2638): everything per entry runs
    # inside C iterators (dict.fromkeys keeps first-appearance order; map/fromiter do the label lookups)
    from itertools import chain
    m = len(Q)
    flat = None
    try:
        # one pass in C: the 2m labels as an object array, numbered in order of first appearance by pandas' hash table
        # (0.4 s instead of 1.0 s for the two Python-level passes below at m = 3.5 M)
        import ctypes
        import pandas as pd
        labels = np.fromiter(chain.from_iterable(Q.keys()), dtype=object, count=2 * m)
        # the 2m labels are references to a few thousand objects (the graph's node ids): numbered by OBJECT first -- the
        # array's PyObject pointers seen as integers, no hashing of 7 M strings -- then the few distinct objects by value
        ptr = np.ctypeslib.as_array((ctypes.c_ssize_t * (2 * m)).from_address(labels.ctypes.data)) if m else np.zeros(0, dtype=np.intp)
        codes, uptr = pd.factorize(ptr, use_na_sentinel=False)
        first = np.full(len(uptr), -1, dtype=np.int64)
        first[codes[::-1]] = np.arange(2 * m - 1, -1, -1)        # position of every object's first appearance
        objs = labels[first]
        index: Dict[Hashable, int] = {}
        remap = np.empty(len(objs), dtype=np.int64)
        for k, v in enumerate(objs):                             # (equal labels held by distinct objects fold together here)
            remap[k] = index.setdefault(v, len(index))
        variables = list(index)
        flat = remap[codes] if len(index) != len(objs) else codes.astype(np.int64, copy=False)
        if any(v is None or v != v for v in variables):          # (None / NaN labels: the plain walk)
            flat = None
    except (ImportError, TypeError, ValueError):
        flat = None
    if flat is None:
        variables = list(dict.fromkeys(chain.from_iterable(Q.keys())))
        index: Dict[Hashable, int] = {v: i for i, v in enumerate(variables)}
        flat = np.fromiter(map(index.__getitem__, chain.from_iterable(Q.keys())), dtype=np.int64, count=2 * m)
    n = len(variables)
    us, vs = flat[0::2].copy(), flat[1::2].copy()
    bs = np.fromiter(Q.values(), dtype=np.float64, count=m)
    lin = np.zeros(n, dtype=np.float64)
    diag = us == vs
    np.add.at(lin, us[diag], bs[diag])
    off = ~diag
    lo = np.minimum(us, vs)[off]
    hi = np.maximum(us, vs)[off]
    pb = bs[off]
    c_pair = 0.0
    npairs = n * (n - 1) // 2
    if detect_uniform and len(pb) and len(pb) * 2 >= npairs and npairs > 0:
        vals, counts = np.unique(pb, return_counts=True)
        top = int(np.argmax(counts))
        # merged (u,v)/(v,u) duplicates would break the "one entry per pair" assumption
        seen = np.zeros(n * n, dtype=bool)               # (len(pb) >= npairs / 2, so this is O(len(pb)) bytes)
        seen[lo * n + hi] = True
        if counts[top] * 2 >= npairs and vals[top] != 0.0 and int(np.count_nonzero(seen)) == len(pb):
            c_pair = float(vals[top])
    if c_pair != 0.0:
        # sparse part = what is left of each pair after the uniform term (the entries are unique pairs here)
        sel = pb != c_pair
        su, sv, sr = lo[sel], hi[sel], pb[sel] - c_pair
        if len(pb) < npairs:
            # pairs absent from the dict have coefficient 0 => sparse part -c_pair there
            present = np.zeros((n, n), dtype=bool)
            present[lo, hi] = True
            au, av = np.nonzero(~present & np.triu(np.ones((n, n), dtype=bool), 1))
            su, sv = np.concatenate([su, au]), np.concatenate([sv, av])
            sr = np.concatenate([sr, np.full(len(au), -c_pair)])
            order = np.lexsort((sv, su))                      # upper-triangular row-major, as before
            su, sv, sr = su[order], sv[order], sr[order]
        else:
            order = np.lexsort((sv, su))
            su, sv, sr = su[order], sv[order], sr[order]
        su, sv, sr, weights = _split_off_pair_weights(n, su, sv, sr, c_pair)
        rowptr, col, val = _csr_from_edges(n, su.astype(np.int32), sv.astype(np.int32), sr)
    else:
        weights = None
        rowptr, col, val = _csr_from_edges(n, lo.astype(np.int32), hi.astype(np.int32), pb)
    return QuboModel(variables, lin, rowptr, col, val, c_pair=c_pair, offset=float(offset),
                     info={"kind": "dict"}, weights=weights)


def _split_off_pair_weights(n, su, sv, sr, c_pair, max_heavy: int = 64):
    """What is left of a QUBO after its uniform pair term ``c_pair`` (upper-triangular entries ``sr`` on pairs ``su < sv``)
    may still hold a few variables coupled to EVERYONE: the slack bits of a squared linear constraint,
    ``lam (sum x_i + sum c_j t_j - ub)^2`` (bqm.add_linear_inequality_constraint, BQM_clustering.py:373-380), whose pairs
    carry ``c_pair c_j`` with the cells and ``c_pair c_j c_k`` among themselves.  If the residual of such variables is
    exactly ``c_pair (a_h - 1)`` on every pair with an ordinary variable and ``c_pair (a_g a_h - 1)`` among them, for
    integers ``a_h >= 2``, those entries are dropped and returned as pair-term WEIGHTS (QuboModel.weights): the model
    stays sparse + rank one and runs on the structured kernels.  Otherwise everything is returned unchanged, weights None."""
    if len(sr) == 0 or n < 4:
        return su, sv, sr, None
    deg = np.bincount(su, minlength=n) + np.bincount(sv, minlength=n)
    heavy = np.flatnonzero(deg >= 0.9 * (n - 1))
    if len(heavy) == 0 or len(heavy) > max_heavy or n - len(heavy) < 2:
        return su, sv, sr, None
    is_h = np.zeros(n, dtype=bool)
    is_h[heavy] = True
    n_light = n - len(heavy)
    hl = is_h[su] != is_h[sv]
    hh = is_h[su] & is_h[sv]
    hvar = np.where(is_h[su], su, sv)[hl]
    r = sr[hl]
    a = np.ones(n, dtype=np.int64)
    for h in heavy:
        rh = r[hvar == h]
        if len(rh) != n_light:
            return su, sv, sr, None
        ah = 1.0 + float(np.median(rh)) / c_pair
        ai = int(round(ah))
        if ai < 2 or abs(ah - ai) > 1e-9 * max(1.0, abs(ah)) or np.max(np.abs(rh - c_pair * (ai - 1))) > 1e-9 * abs(c_pair) * ai:
            return su, sv, sr, None
        a[h] = ai
    # among the heavy variables: every pair present with residual c (a_g a_h - 1)
    if int(np.count_nonzero(hh)) != len(heavy) * (len(heavy) - 1) // 2:
        return su, sv, sr, None
    want = c_pair * (a[su[hh]] * a[sv[hh]] - 1)
    if len(want) and np.max(np.abs(sr[hh] - want)) > 1e-9 * np.max(np.abs(want)):
        return su, sv, sr, None
    keep = ~(hl | hh)
    return su[keep], sv[keep], sr[keep], a


def slot_independent_order(rowptr: np.ndarray, col: np.ndarray, slot: int = 64) -> np.ndarray:
    """Sweep order for the structured kernels: a permutation ``perm`` (``perm[new] = old``) that packs the
    variables into consecutive blocks of ``slot`` (the wavefront width) such that, as far as a greedy
    balanced colouring manages, no two variables of a block are neighbours.  Inside such a block the
    decisions of a sweep interact only through the global sum, which the kernels exploit (accept masks by
    fixed-point rounds); blocks that keep an internal edge simply take the general path.  Any visiting order is a
    valid Metropolis sweep; this one is deterministic (degree-descending greedy, ties by index).

    The greedy pass runs in the native library (``mi_sa_plan_slot_order``, host code: O(n * slots) -- 40 ms at
    n = 50 000 where the numpy loop it replaces took seconds); oracle/model_oracle.py keeps the restatement the
    tests compare it with."""
    import ctypes as C
    from . import _lib
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = len(rowptr) - 1
    perm = np.empty(n, dtype=np.int64)
    _lib.check(_lib.load().mi_sa_plan_slot_order(
        rowptr.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)), int(n), int(slot),
        perm.ctypes.data_as(C.POINTER(C.c_int64))))
    return perm


def padded_slot_layout(rowptr: np.ndarray, col: np.ndarray, slot: int = 64, max_slots: Optional[int] = None):
    """``(pos, slots, clashes)``: seat ``pos[i] = slot_index * slot + rank`` of every variable in a layout of ``slots``
    blocks of ``slot`` seats, the fewest (from ``ceil(n / slot)`` up to ``max_slots``, default ``3 * ceil(n / slot) + 4``)
    for which the greedy colouring of ``slot_independent_order`` leaves NO edge inside a block; the seats left over
    are holes.  ``clashes`` > 0: even ``max_slots`` did not suffice and the packed layout is returned.  Native
    (``mi_sa_plan_slot_layout``); ``oracle/model_oracle.py`` keeps the restatement the tests compare it with."""
    import ctypes as C
    from . import _lib
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = len(rowptr) - 1
    s0 = (n + slot - 1) // slot
    if max_slots is None:
        max_slots = 3 * s0 + 4
    pos = np.empty(n, dtype=np.int64)
    slots, clashes = C.c_int(0), C.c_int(0)
    _lib.check(_lib.load().mi_sa_plan_slot_layout(
        rowptr.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)), int(n), int(slot),
        int(max_slots), pos.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(slots), C.byref(clashes)))
    return pos, int(slots.value), int(clashes.value)


def _reseat_order(rowptr, col, new_of_old, n_dev: int):
    """``(rowptr_new, col_new, order)`` of a CSR whose variable ``i`` moves to index ``new_of_old[i]`` of ``n_dev``: rows in
    new index order, neighbours ascending by new index; ``order`` gathers any per-entry array into the new entry order.
    One sort of a combined 64-bit key (row * n_dev + column)."""
    new_of_old = np.asarray(new_of_old, dtype=np.int64)
    deg = np.diff(rowptr)
    rows_old = np.repeat(np.arange(len(new_of_old)), deg)
    key = new_of_old[rows_old] * np.int64(n_dev) + new_of_old[np.asarray(col)]
    order = np.argsort(key, kind="stable")
    cnt = np.zeros(n_dev, dtype=np.int64)
    cnt[new_of_old] = deg
    rp = np.zeros(n_dev + 1, dtype=np.int32)
    rp[1:] = np.cumsum(cnt)
    return rp, (key[order] % n_dev).astype(np.int32), order


def pad_csr(rowptr, col, val, pos, n_dev: int, also=None):
    """CSR of the same symmetric matrix with variable ``i`` moved to seat ``pos[i]`` of ``n_dev`` seats; the other seats are
    holes (empty rows).  Rows keep their neighbours in ascending NEW index order (as ``permute_csr``).  ``also``: a second
    per-entry array (the fp64 coefficients of the energy model) carried through the same reordering -- returned fourth."""
    rp, c, order = _reseat_order(rowptr, col, pos, n_dev)
    if also is None:
        return rp, c, np.asarray(val)[order]
    return rp, c, np.asarray(val)[order], np.asarray(also)[order]


def permute_csr(rowptr, col, val, perm, also=None):
    """CSR of the same symmetric matrix with variables renumbered by ``perm`` (``perm[new] = old``); rows keep
    their neighbours in ascending NEW index order.  ``also``: as in :func:`pad_csr`."""
    n = len(perm)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    rp, c, order = _reseat_order(rowptr, col, inv, n)
    if also is None:
        return rp, c, np.asarray(val)[order]
    return rp, c, np.asarray(val)[order], np.asarray(also)[order]


def add_size_window_penalty(model: QuboModel, lb: float, ub: float, lagrange_multiplier: float,
                            slack_prefix: str = "slack_c1_constraint_") -> QuboModel:
    """Penalty form of ``bqm.add_linear_inequality_constraint([(x_i, 1)...], lb, ub, lagrange)``
    (BQM_clustering.py:373-380) in dimod's construction (:func:`bqm.inequality_slack`, shared with the
    ``BinaryQuadraticModel`` look-alike): binary slack variables ``t_j`` with positive coefficients ``c_j`` spanning
    ``[0, int(ub - lb)]`` and the energy ``lagrange * (sum_i x_i + sum_j c_j t_j - ub)^2`` with ``ub`` kept fractional
    (the reference passes ``n / 6``: for n = 256, lb = 40 the sizes 41-43 cost the minimum ``lagrange / 9``, not zero).
    ``ValueError`` when ``ub < lb``.  Returned over ``variables + slack``: in STRUCTURED form (the model's sparse part + a
    uniform pair term ``2 lagrange`` with integer weights 1 / ``c_j``) when the model itself is sparse without a pair
    term -- the reference's `clustering_bqm_3` -- else dense-backed."""
    from .bqm import inequality_slack
    n = model.num_variables
    coeffs, ub_c = inequality_slack([1] * n, lb, ub, 0, slack_prefix.rstrip("_"))
    if coeffs is None:
        coeffs, lam = [], 0.0                                   # feasible for every state: nothing is added
    else:
        lam = float(lagrange_multiplier)
    lb_c = max(0.0, float(lb))
    ns = len(coeffs)
    a = np.concatenate([np.ones(n), np.asarray(coeffs, dtype=np.float64)])     # sum a_i z_i - ub_c
    N = n + ns
    variables = list(model.variables) + [slack_prefix + str(i) for i in range(ns)]
    if (model._dense is None and model.c_pair == 0.0 and model.weights is None and lam != 0.0 and ns <= 64
            and all(float(c) == int(c) and int(c) >= 1 for c in coeffs)):
        # STRUCTURED form: lam (a.z - ub_c)^2 = lam sum_i (a_i^2 - 2 ub_c a_i) z_i + 2 lam sum_{i<j} a_i a_j z_i z_j + lam ub_c^2
        # -- the sparse cut term of the model plus a WEIGHTED uniform pair term (c_pair = 2 lam, weights a): slack bits have
        # no sparse couplings, so the structured kernels run it (mi_sa_problem_set_pair_weights) instead of the dense ones
        ai = np.concatenate([np.ones(n, dtype=np.int64), np.asarray(coeffs, dtype=np.int64)])
        lin = np.concatenate([model.lin, np.zeros(ns)]) + lam * (a * a - 2.0 * ub_c * a)
        rowptr = np.concatenate([model.rowptr, np.full(ns, model.rowptr[-1], dtype=model.rowptr.dtype)])
        return QuboModel(variables, lin, rowptr, model.col, model.val, c_pair=2.0 * lam,
                         offset=model.offset + lam * ub_c * ub_c,
                         info=dict(model.info, slack=ns, lb=lb_c, ub=ub_c),
                         weights=None if np.all(ai == 1) else ai)
    Qs = np.zeros((N, N), dtype=np.float64)
    Qs[:n, :n] = model.dense_Qs()
    # lam (a.z - ub_c)^2 = lam [ sum_i a_i^2 z_i + 2 sum_{i<j} a_i a_j z_i z_j - 2 ub_c a.z + ub_c^2 ]
    outer = lam * np.outer(a, a)
    Qs += outer - np.diag(np.diag(outer))
    Qs[np.arange(N), np.arange(N)] += lam * (a * a - 2.0 * ub_c * a)
    out = QuboModel(variables, np.diag(Qs).copy(), np.zeros(N + 1, dtype=np.int32),
                    np.zeros(0, dtype=np.int32), np.zeros(0), c_pair=0.0,
                    offset=model.offset + lam * ub_c * ub_c,
                    info=dict(model.info, slack=ns, lb=lb_c, ub=ub_c))
    out._dense = Qs
    return out


# --------------------------------------------------------------------------------------------
# schedules (neal-compatible defaults with the degenerate-range guard of SURVEY.md 8c)
# --------------------------------------------------------------------------------------------
def default_beta_range(model: QuboModel, rel_zero: float = 1e-9) -> Tuple[float, float]:
    """neal's rule ``beta_hot = ln2 / (2 max_i(|h_i| + sum_j |J_ij|))``, ``beta_cold = ln100 /
    (2 min nonzero |bias|)`` on the Ising form (h_i = Q_ii/2 + sum_j Q_ij/4, J_ij = Q_ij/4), with
    biases below ``rel_zero * max|J|`` treated as exactly zero: for A1 the h_i vanish analytically
    but evaluate to ~1e-14, which would otherwise give beta_cold ~ 5e15."""
    n = model.num_variables
    if model._dense is not None:
        Qs = model._dense
        off = Qs - np.diag(np.diag(Qs))
        J_abs_rowsum = np.abs(off).sum(axis=1) / 2.0
        h = np.diag(Qs) / 2.0 + off.sum(axis=1) / 2.0
        nzJ = np.abs(off[off != 0.0]) / 2.0
    else:
        rows = np.repeat(np.arange(n), np.diff(model.rowptr))
        # pair (i, j) carries c_pair a_i a_j + S_ij (a = 1 without weights: the expressions below reduce to c_pair (n - 1) etc.)
        a = np.ones(n) if model.weights is None else np.asarray(model.weights, dtype=np.float64)
        others = a * (a.sum() - a)                                  # a_i * sum_{j != i} a_j
        uni = model.c_pair * a[rows] * a[model.col]                 # the uniform part on the stored pairs
        pair_sum = np.zeros(n)
        np.add.at(pair_sum, rows, model.val)
        pair_sum += model.c_pair * others
        h = model.lin / 2.0 + pair_sum / 4.0
        full = model.val + uni
        abs_sum = np.zeros(n)
        np.add.at(abs_sum, rows, np.abs(full) - np.abs(uni))
        abs_sum += abs(model.c_pair) * others
        J_abs_rowsum = abs_sum / 4.0
        cand = [np.abs(full[full != 0.0]) / 4.0]
        if model.c_pair != 0.0 and len(model.val) < n * (n - 1) and n >= 2:
            two = np.partition(a, 1)[:2]                            # the smallest product of two weights
            cand.append(np.array([abs(model.c_pair) * two[0] * two[1] / 4.0]))
        nzJ = np.concatenate(cand) if cand else np.zeros(0)
    maxJ = float(nzJ.max()) if len(nzJ) else 0.0
    scale = max(maxJ, float(np.max(np.abs(h))) if n else 0.0)
    tiny = rel_zero * scale
    h_eff = np.where(np.abs(h) > tiny, np.abs(h), 0.0)
    max_field = float(np.max(h_eff + J_abs_rowsum)) if n else 1.0
    biases = np.concatenate([h_eff[h_eff > 0.0], nzJ[nzJ > tiny]])
    min_bias = float(biases.min()) if len(biases) else 1.0
    if max_field <= 0.0:
        max_field = 1.0
    beta_hot = np.log(2.0) / (2.0 * max_field)
    beta_cold = np.log(100.0) / (2.0 * min_bias)
    return float(beta_hot), float(beta_cold)


def make_beta_schedule(num_sweeps: int, beta_range: Sequence[float],
                       beta_schedule_type: str = "geometric", num_sweeps_per_beta: int = 1,
                       beta_schedule: Optional[Sequence[float]] = None) -> np.ndarray:
    """neal-style schedule expanded to one beta per sweep (float64[num_sweeps])."""
    if beta_schedule_type == "custom":
        if beta_schedule is None:
            raise ValueError("'beta_schedule' must be provided for beta_schedule_type = 'custom'")
        b = np.asarray(beta_schedule, dtype=np.float64)
        if np.any(b < 0) or not np.all(np.isfinite(b)):
            raise ValueError("'beta_schedule' cannot include negative or non-finite values")
        return np.repeat(b, num_sweeps_per_beta)
    if num_sweeps_per_beta < 1:
        raise ValueError("'num_sweeps_per_beta' must be a positive integer")
    if num_sweeps % num_sweeps_per_beta != 0:
        raise ValueError("'num_sweeps' must be divisible by 'num_sweeps_per_beta'")
    num_betas = num_sweeps // num_sweeps_per_beta
    hot, cold = float(beta_range[0]), float(beta_range[1])
    if hot < 0 or cold < 0:
        raise ValueError("beta_range values must be non-negative")
    if num_betas == 0:
        return np.zeros(0, dtype=np.float64)
    if beta_schedule_type == "linear":
        b = np.linspace(hot, cold, num=num_betas)
    elif beta_schedule_type == "geometric":
        if hot <= 0 or cold <= 0:
            raise ValueError("'beta_range' must contain non-zero values for a geometric schedule")
        b = np.geomspace(hot, cold, num=num_betas)
    else:
        raise ValueError("Beta schedule type {} not implemented".format(beta_schedule_type))
    return np.repeat(b, num_sweeps_per_beta)
