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


@pytest.mark.parametrize("kw", [dict(symmetric=False), dict(enhance="mutual"), dict(enhance="mutual", mutual_bonus=1.0, symmetric=False),
                                dict(enhance="sum"), dict(enhance="sum", symmetric=False), dict(enhance="mutual", ord2=9),
                                dict(enhance="sum", symmetric=False, ord2=7), dict(ord2=5)])
@pytest.mark.parametrize("n,k,ord_,dim", [(300, 5, 15, 15), (2638, 5, 15, 15), (900, 10, 12, 30)])
def test_variant_graphs_equal_oracle(n, k, ord_, dim, kw):
    """The notebooks' optional chunks on the GPU (unsymmetric trim, mutual bonus / A + t(A), second trim:
    Pbmc3k_general_data_preparation.Rmd:77-123, Kidney_data.Rmd:235-266) against oracle/snn_oracle.c: structure, shared
    counts and weight codes bit for bit; the exported edge list equals graphs.edges_from_matrix of the dense matrix."""
    from scrna_seq_qannealing_clustering_amd import graphs
    X = cloud(n, dim, seed=5 * n + k)
    g = snn.build_snn(X, k, 0.0, ord_, **kw)
    nn, rowptr, col, shared, code = sn.snn_graph_variant(X, k, 0.0, ord_, kw.get("symmetric", True), kw.get("enhance"),
                                                        kw.get("mutual_bonus", 2.0), kw.get("ord2"))
    assert np.array_equal(g.rowptr, rowptr) and np.array_equal(g.col, col)
    assert np.array_equal(g.shared, shared) and np.array_equal(g.code, code)
    if n <= 300:
        A = np.zeros((n, n))
        A[g.col, np.repeat(np.arange(n), np.diff(g.rowptr))] = g.weights
        eu, ev, w = graphs.edges_from_matrix(A)
        _, gu, gv, gw = g.edge_list()
        assert np.array_equal(gu, eu) and np.array_equal(gv, ev) and np.array_equal(gw, w)
        m = models.build_bqm_qubo(g.to_graph(), 0.05)              # the variant graph feeds the clustering model
        assert m.num_variables == n


def test_variant_argument_validation():
    X = cloud(100, 3, seed=0)
    with pytest.raises(_lib.MiSaError):
        snn.build_snn(X, 5, 0.0, 10, symmetric=False, enhance="mutual", ord2=5)      # second trim needs a symmetric matrix
    with pytest.raises(_lib.MiSaError):
        snn.build_snn(X, 5, 0.0, None, symmetric=False)                              # unsymmetric trim needs ord
    with pytest.raises(ValueError):
        snn.build_snn(X, 5, 0.0, 10, enhance="both")


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


@pytest.mark.parametrize("n,k,ord_,dim,digits,neg", [(3000, 5, 8, 15, 2, None), (3000, 5, 8, 15, 2, 0.16), (1200, 30, 12, 10, 2, 0.05),
                                                    (900, 5, None, 8, 2, 0.16), (700, 64, 20, 6, 2, None)])
def test_rounding_and_negative_edge_variant_equals_the_oracle(n, k, ord_, dim, digits, neg):
    """snn.build_snn(round_digits=, negative_below=): the rounding chunk of Pbmc3k_normalization_simulated_data.Rmd:597-616
    (the notebook's own parameters first: k = 5, ord = 8, two digits, negative edges below 0.16) against the oracle, which
    the literal dense restatement of the R lines pins (tests/test_snn_oracle.py): graph, counts, codes, fp64 weights."""
    X = cloud(n, dim, seed=7 * n + k, clusters=5)
    g = snn.build_snn(X, k, 0.0, ord_, round_digits=digits, negative_below=neg)
    nn, rowptr, col, shared, code = sn.snn_graph_rounded(X, k, 0.0, ord_, digits, neg)
    assert np.array_equal(g.nn, nn) and np.array_equal(g.rowptr, rowptr)
    assert np.array_equal(g.col, col) and np.array_equal(g.shared, shared) and np.array_equal(g.code, code)
    want = np.where(code == 3, -0.3, np.round(shared / (2.0 * k - shared), digits))
    assert np.array_equal(g.weights, want)
    if ord_ is not None:
        assert g.max_degree <= ord_ and (g.weights > 0).all()
    elif neg:
        assert (g.weights == -0.3).any()
    nodes, eu, ev, w = g.edge_list()
    assert len(w) * 2 == len(g.col)


def test_negative_edges_on_a_tiny_dense_graph_are_refused():
    with pytest.raises(_lib.MiSaError):
        snn.build_snn(cloud(12, 2, seed=3, clusters=1), 6, 0.0, 10, round_digits=2, negative_below=0.5)
    with pytest.raises(ValueError):
        snn.build_snn(cloud(100, 3, seed=0), 5, 0.0, 8, negative_below=0.16)            # needs round_digits


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


def test_untrimmed_graph_through_both_structured_kernels():
    """The reference's graph type 0 (no trim, main.py:88): rows of up to several hundred neighbours, built on the
    GPU and annealed by the runtime-width forms of K2 (balanced cut) and K3 (k-way) behind the sampler."""
    from scrna_seq_qannealing_clustering_amd import MI355XSampler, build_bqm_qubo, build_dqm_potts
    rs = np.random.RandomState(4)
    truth = np.repeat([0, 1, 2], 200)
    X = (rs.normal(scale=6.0, size=(3, 12))[truth] + rs.normal(size=(600, 12))).astype(np.float32)
    g = snn.build_snn(X, 12, 0.0, None)
    assert g.max_degree > 64                                     # wider than the register-resident layout
    G = g.to_graph()
    sampler = MI355XSampler()
    ss = sampler.sample_dqm(build_dqm_potts(G, 3, 0.005), num_reads=32, num_sweeps=300, seed=1)
    assert ss.info["kernel"] == "potts_csr"
    lab = np.array([ss.first.sample[v] for v in G.nodes])
    # the three blobs are disconnected in the SNN graph: one label each
    assert all(len(set(lab[truth == c])) == 1 for c in range(3)) and len(set(lab)) == 3
    m = build_bqm_qubo(G, 0.05)
    sb = sampler.sample_qubo(m, num_reads=64, num_sweeps=1000, seed=1)
    assert sb.info["kernel"] == "csr_rank1"                      # no n x n matrix needed
    x = np.array([sb.first.sample[v] for v in G.nodes])
    assert sb.first.energy == pytest.approx(float(m.energies(x[None, :])[0]), rel=1e-12)
    planted = (truth == 0).astype(np.int8)                       # one blob against the other two: cut 0
    assert sb.first.energy <= float(m.energies(planted[None, :])[0]) + 1e-6    # at least as good as the planted cut
