"""Pins oracle/snn_oracle.c (CPU restatement of the SNN construction) against a LITERAL dense numpy
restatement of the reference's R lines (Pbmc3k_prepare_data_for_QA_clustering.Rmd:67-79) that lives in
graphs.snn_from_points, on seeded point clouds.  CPU only."""
import numpy as np
import pytest

from oracle import snn_oracle as sn
from scrna_seq_qannealing_clustering_amd import graphs


def cloud(n, dim, seed, clusters=4):
    rng = np.random.RandomState(seed)
    return (rng.normal(size=(n, dim)) + 3.0 * rng.randint(0, clusters, size=(n, 1))).astype(np.float32)


@pytest.mark.parametrize("n,k,ord_,dim", [(300, 5, 15, 15), (700, 10, 15, 30), (500, 16, 16, 15), (400, 5, None, 8),
                                          (257, 5, 3, 2)])
def test_oracle_equals_the_dense_restatement_of_the_r_loop(n, k, ord_, dim):
    X = cloud(n, dim, seed=n + k)
    nn, rowptr, col, shared = sn.snn_graph(X, k, 0.0, ord_)
    assert np.array_equal(nn, graphs._knn_exact(X.astype(np.float64), k))       # same neighbours, same order
    dense = graphs.snn_from_points(X.astype(np.float64), k, ord_)
    W = np.zeros((n, n))
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    W[rows, col] = shared / (2.0 * k - shared)
    assert np.array_equal(W, dense)                                              # bit-equal fp64 weights
    assert np.array_equal(W, W.T)
    if ord_ is not None:
        assert np.diff(rowptr).max() <= ord_


def test_prune_drops_light_edges_and_ties_keep_index_order():
    X = cloud(200, 6, seed=1)
    nn = sn.knn(X, 6)
    rp0, col0, s0 = sn.snn_rows(nn, 0.0)
    rp1, col1, s1 = sn.snn_rows(nn, 0.2)             # s/(12-s) >= 0.2  <=>  s >= 2
    assert s1.min() >= 2 and (s0 >= 2).sum() == len(s1)
    # duplicate points: equal distances are ordered by index
    Y = np.zeros((5, 2), dtype=np.float32)
    assert sn.knn(Y, 3).tolist() == [[0, 1, 2], [1, 0, 2], [2, 0, 1], [3, 0, 1], [4, 0, 1]]


VARIANTS = [dict(symmetric=False), dict(enhance="mutual"), dict(enhance="mutual", bonus=1.0, symmetric=False),
            dict(enhance="sum"), dict(enhance="sum", symmetric=False), dict(enhance="mutual", ord2=9),
            dict(enhance="sum", symmetric=False, ord2=7), dict(ord2=5), dict(enhance="sum", ord2=12)]


def _dense_from_variant(n, k, rowptr, col, shared, code, bonus):
    """A[r, i] from the stored rows (= columns) of a variant graph, weights in fp64 as R computes them."""
    w = shared / (2.0 * k - shared)
    w = np.where(code == 1, w + bonus, np.where(code == 2, w + w, w))
    A = np.zeros((n, n))
    A[col, np.repeat(np.arange(n), np.diff(rowptr))] = w
    return A


@pytest.mark.parametrize("kw", VARIANTS)
@pytest.mark.parametrize("n,k,ord_,dim", [(300, 5, 15, 15), (257, 8, 6, 4)])
def test_variant_chunks_equal_their_dense_restatement(n, k, ord_, dim, kw):
    """The notebooks' optional chunks (unsymmetric trim, the two enhancements, the second trim) in the oracle against
    the literal dense restatement of Pbmc3k_general_data_preparation.Rmd:77-123 -- bit-equal fp64 matrices."""
    X = cloud(n, dim, seed=3 * n + k)
    bonus = kw.get("bonus", 2.0)
    nn, rowptr, col, shared, code = sn.snn_graph_variant(X, k, 0.0, ord_, kw.get("symmetric", True), kw.get("enhance"),
                                                        bonus, kw.get("ord2"))
    dense = graphs.snn_from_points(X.astype(np.float64), k, ord_, symmetric=kw.get("symmetric", True),
                                   enhance=kw.get("enhance"), bonus=bonus, ord2=kw.get("ord2"))
    A = _dense_from_variant(n, k, rowptr, col, shared, code, bonus)
    assert np.array_equal(A, dense)
    if kw.get("symmetric", True) or kw.get("enhance") == "sum":
        assert np.array_equal(A, A.T)
    else:
        assert not np.array_equal(A, A.T)                       # the unsymmetric trim really breaks the symmetry
    if kw.get("ord2"):
        assert (A != 0).sum(axis=0).max() <= kw["ord2"]


def test_edge_list_of_an_asymmetric_matrix_is_what_networkx_builds():
    """graphs.edges_from_matrix == nx.from_numpy_array(A) + G.edges(data=True) (the notebooks' export:
    Pbmc3k_general_data_preparation.Rmd:145 `nx.from_numpy_matrix`), order and weights, for an asymmetric matrix."""
    import networkx as nx
    X = cloud(120, 6, seed=9)
    A = graphs.snn_from_points(X.astype(np.float64), 6, 5, symmetric=False, enhance="mutual")
    assert not np.array_equal(A, A.T)
    G = nx.from_numpy_array(A)
    want = [(u, v, d["weight"]) for u, v, d in G.edges(data=True)]
    eu, ev, w = graphs.edges_from_matrix(A)
    assert [(int(a), int(b), float(c)) for a, b, c in zip(eu, ev, w)] == want


@pytest.mark.parametrize("n,k,ord_,dim,digits,neg", [(300, 5, 8, 15, 2, None), (300, 5, 8, 15, 2, 0.16), (500, 10, 8, 20, 2, 0.16),
                                                    (400, 30, 12, 10, 2, 0.05), (350, 5, None, 8, 2, 0.16), (257, 8, 6, 4, 1, 0.25)])
def test_rounding_and_negative_edge_chunk_equals_its_dense_restatement(n, k, ord_, dim, digits, neg):
    """Pbmc3k_normalization_simulated_data.Rmd:597-616: round(snn, 2), optionally snn[snn < 0.16 & snn != 0] <- -0.3, then
    the trim -- ranked by the ROUNDED values (k = 30, one digit ... : different counts tie), the negative entries below
    every zero of their column, so the trim deletes them.  Oracle (CSR) == the literal dense restatement, bit for bit."""
    X = cloud(n, dim, seed=5 * n + k)
    nn, rowptr, col, shared, code = sn.snn_graph_rounded(X, k, 0.0, ord_, digits, neg)
    dense = graphs.snn_from_points(X.astype(np.float64), k, ord_, round_digits=digits, negative_below=neg, negative_value=-0.3)
    W = np.zeros((n, n))
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    W[rows, col] = np.where(code == 3, -0.3, np.round(shared / (2.0 * k - shared), digits))
    assert np.array_equal(W, dense) and np.array_equal(W, W.T)
    if ord_ is not None:
        assert np.diff(rowptr).max() <= ord_ and not (code == 3).any()     # the trim removed every negative edge
    elif neg:
        assert (code == 3).any() and (dense < 0).any()
    if neg and ord_ is not None and k <= 10:                   # (at k = 30 the light edges never reach a top-12 anyway)
        # not the positive-only graph with the light edges merely rounded: the support differs
        plain = graphs.snn_from_points(X.astype(np.float64), k, ord_, round_digits=digits)
        assert not np.array_equal(plain != 0, dense != 0)


def test_negative_edges_can_survive_the_trim_of_a_tiny_dense_graph():
    """Why the build refuses negative edges on a graph with n - (densest column) < ord: R keeps the first `ord` positions
    of [positives, zeros, negatives], and with too few non-negative positions some negatives are among them."""
    X = cloud(12, 2, seed=3, clusters=1)
    dense = graphs.snn_from_points(X.astype(np.float64), 6, 10, round_digits=2, negative_below=0.5, negative_value=-0.3)
    assert (dense < 0).any()
    with pytest.raises(AssertionError):
        sn.snn_graph_rounded(X, 6, 0.0, 10, 2, 0.5)
