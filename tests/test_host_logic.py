"""Host-side logic that needs no GPU: sweep ordering, CSR permutation, the f4 model builders, bit packing
for the metrics pass, launch-independent pieces of the SNN wrapper."""
import numpy as np
import pytest

from conftest import load_fixture
from oracle import model_oracle as mo
from scrna_seq_qannealing_clustering_amd import graphs, metrics, models


def test_slot_independent_order_is_a_permutation_that_removes_in_slot_edges():
    nodes, eu, ev, w, _ = graphs.synthetic_snn(1500, 5, 15, 15, 6, seed=1)
    m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
    perm = models.slot_independent_order(m.rowptr, m.col)
    assert sorted(perm.tolist()) == list(range(1500))
    assert np.array_equal(perm, models.slot_independent_order(m.rowptr, m.col))      # deterministic
    rp, cc, vv = models.permute_csr(m.rowptr, m.col, m.val, perm)
    rows = np.repeat(np.arange(1500), np.diff(rp))
    before = (np.repeat(np.arange(1500), np.diff(m.rowptr)) >> 6) == (m.col >> 6)
    after = (rows >> 6) == (cc >> 6)
    assert before.sum() > 100 and after.sum() <= before.sum() // 20
    # same matrix, renumbered: energies agree for x given in the new order
    x = np.random.RandomState(0).randint(0, 2, size=1500)
    W = np.zeros((1500, 1500))
    W[np.repeat(np.arange(1500), np.diff(m.rowptr)), m.col] = m.val
    W2 = np.zeros_like(W)
    W2[rows, cc] = vv
    assert np.isclose(x @ W @ x, x[perm] @ W2 @ x[perm], rtol=1e-12)
    assert all(np.all(np.diff(cc[rp[i]:rp[i + 1]]) > 0) for i in range(0, 1500, 97))  # rows ascending


def test_order_of_tiny_and_dense_graphs():
    assert models.slot_independent_order(np.array([0, 1, 2]), np.array([1, 0])).tolist() == [0, 1]
    n = 130                                                         # complete graph: conflicts are unavoidable
    col = np.array([j for i in range(n) for j in range(n) if j != i])
    rowptr = np.arange(0, n * (n - 1) + 1, n - 1)
    perm = models.slot_independent_order(rowptr, col)
    assert sorted(perm.tolist()) == list(range(n))


def test_cqm_subsampling_and_mis_builders_match_their_literal_forms():
    fx = load_fixture("varied")
    G = fx.graph()
    pm = models.build_cqm_potts(G, 4, 20)
    assert pm.info["min_cluster_size"] == 20 and pm.c_pair == 0.0
    lab = np.random.RandomState(1).randint(0, 4, size=(3, 256))
    lit = []                                                        # CQM_clustering.py:40-44 on the one-hot expansion
    for r in range(3):
        e = 0.0
        for a, b, w in zip(fx.eu, fx.ev, fx.w):
            e += 2.0 - (2.0 * w if lab[r, a] == lab[r, b] else 0.0)
        lit.append(e)
    assert np.allclose(pm.energies(lab), lit, rtol=1e-12)
    mis = models.build_mis_qubo(G, 2.0)
    x = np.random.RandomState(2).randint(0, 2, size=(4, 256))
    lit = [-xr.sum() + 2.0 * sum(1 for a, b in zip(fx.eu, fx.ev) if xr[a] and xr[b]) for xr in x]
    assert np.allclose(mis.energies(x), lit, rtol=1e-12)
    sub = models.build_subsampling_qubo(G, 0.25)
    lit = [sum((1 - w) * (xr[a] * xr[b] - xr[a] - xr[b]) for a, b, w in zip(fx.eu, fx.ev, fx.w)) + 0.25 * xr.sum() for xr in x]
    assert np.allclose(sub.energies(x), lit, rtol=1e-12)


def test_pack_expression_bit_layout():
    X = (np.random.RandomState(0).rand(37, 131) < 0.3) * 2.5
    bits = metrics.pack_expression(X)
    assert bits.shape == (37, 3) and bits.dtype == np.uint64
    for i in (0, 5, 36):
        for g in (0, 1, 63, 64, 65, 127, 128, 130):
            assert ((int(bits[i, g // 64]) >> (g % 64)) & 1) == int(X[i, g] != 0)
    assert int(bits[0, 2]) >> 3 == 0                                # padding bits are clear
