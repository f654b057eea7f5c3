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
