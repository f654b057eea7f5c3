"""GPU SNN construction (csrc/snn_kernels.hip through include/mi_snn.h) against oracle/snn_oracle.c:
neighbour tables, shared counts, trimmed graphs -- all integer, bit-exact.  GPU only."""
import numpy as np
import pytest

from oracle import snn_oracle as sn
from scrna_seq_qannealing_clustering_amd import _lib, models, snn

pytestmark = pytest.mark.gpu


def cloud(n, dim, seed, clusters=6):
    rng = np.random.RandomState(seed)
    return (rng.normal(size=(n, dim)) + 3.0 * rng.randint(0, clusters, size=(n, 1))).astype(np.float32)


@pytest.mark.parametrize("n,k,ord_,dim,prune", [(300, 5, 15, 15, 0.0), (1000, 10, 15, 30, 0.0), (777, 16, 16, 15, 1 / 15),
                                                (513, 5, None, 3, 0.0), (64, 2, 1, 1, 0.0), (2638, 5, 15, 15, 0.0),
                                                (1500, 33, 20, 50, 0.0)])
def test_graph_equals_oracle(n, k, ord_, dim, prune):
    X = cloud(n, dim, seed=7 * n + k)
    g = snn.build_snn(X, k, prune, ord_)
    nn, rowptr, col, shared = sn.snn_graph(X, k, prune, ord_)
    assert np.array_equal(g.nn, nn)
    assert np.array_equal(g.rowptr, rowptr)
    assert np.array_equal(g.col, col) and np.array_equal(g.shared, shared)
    assert g.max_degree == int(np.diff(rowptr).max())


def test_duplicate_points_and_size_independent_properties():
    """Ties (duplicated points) resolve by index as in the oracle; at n = 20000 the graph is checked through
    properties: symmetric, zero diagonal, degree <= ord, s in [1, k], every row ascending."""
    X = cloud(400, 4, seed=3)
    X[100:200] = X[0:100]                                   # exact duplicates
    g = snn.build_snn(X, 8, 0.0, 10)
    nn, rowptr, col, shared = sn.snn_graph(X, 8, 0.0, 10)
    assert np.array_equal(g.nn, nn) and np.array_equal(g.col, col) and np.array_equal(g.shared, shared)
    n, k, ord_ = 20000, 5, 15
    g = snn.build_snn(cloud(n, 15, seed=11, clusters=30), k, 0.0, ord_)
    rows = np.repeat(np.arange(n), np.diff(g.rowptr))
    assert np.diff(g.rowptr).max() <= ord_ and (g.col != rows).all()
    assert g.shared.min() >= 1 and g.shared.max() <= k
    key = rows.astype(np.int64) * n + g.col
    assert np.all(np.diff(key) > 0)                        # rows ascending, no duplicates
    assert np.array_equal(np.sort(g.col.astype(np.int64) * n + rows), key)      # symmetric pattern
    assert np.array_equal(g.nn[:, 0], np.arange(n))


def test_built_graph_feeds_the_clustering_model():
    g = snn.build_snn(cloud(600, 10, seed=5, clusters=3), 5, 0.0, 15)
    m = models.build_bqm_qubo(g.to_graph(), 0.05)
    assert m.num_variables == 600 and m._dense is None and len(m.col) == len(g.col)


def test_argument_validation():
    X = cloud(50, 3, seed=0)
    for bad in (dict(k=1), dict(k=65), dict(k=51)):
        with pytest.raises(_lib.MiSaError):
            snn.build_snn(X, bad["k"])
    with pytest.raises(_lib.MiSaError):
        snn.build_snn(np.zeros((10, 65), dtype=np.float32), 3)
