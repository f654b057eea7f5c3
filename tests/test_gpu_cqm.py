"""Row f4: the constrained / auxiliary models of the reference on the GPU -- CQM clustering with a minimum
cluster size (CQM_clustering.py:26-55), the sub-sampling QUBO (QA_subsampling.py:25-35) and the maximum
independent set QUBO behind dnx.maximum_independent_set (QA_subsampling.py:102).  GPU only."""
import numpy as np
import pytest

from conftest import load_fixture
from oracle import sa_oracle as so
from scrna_seq_qannealing_clustering_amd import MI355XSampler, clustering, models
from scrna_seq_qannealing_clustering_amd.engine import Problem
from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range

pytestmark = pytest.mark.gpu


def f32(x):
    return np.asarray(x, dtype=np.float32)


@pytest.mark.parametrize("min_size", [0, 20, 60])
def test_min_cluster_size_chain_equals_oracle(min_size):
    """K3 with the hard size constraint follows the oracle flip for flip; no cluster ever ends below the bound
    although the unconstrained chain (min_size = 0) empties clusters on this model."""
    fx = load_fixture("blobs")
    pm = models.build_cqm_potts(fx.graph(), 4, min_size)
    betas = models.make_beta_schedule(60, default_potts_beta_range(pm))
    init = (np.arange(256)[None, :] % 4).repeat(12, axis=0).astype(np.uint16)          # 64 per cluster: feasible
    args = (pm.rowptr, pm.col, f32(pm.val), 0.0, 256, 4)
    olab, oen, ostats = so.potts_csr_philox(*args, 12, betas, 5, lin_offset=pm.lin_offset, init=init, min_size=min_size)
    with Problem.potts_csr(*args, lin_offset=pm.lin_offset) as p:
        p.set_option("min_cluster_size", min_size)
        p.anneal(12, betas, 5, initial_states=init)
        lab, en, info = p.fetch()
    assert np.array_equal(lab, olab) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)
    sizes = np.array([np.bincount(row, minlength=4) for row in lab])
    assert sizes.min() >= min_size
    if min_size == 0:
        assert sizes.min() < 20            # what the constraint is there to prevent


def test_clustering_cqm_driver_returns_feasible_one_hot_solutions():
    fx = load_fixture("noisy_circles")
    G = fx.graph()
    ss = clustering.clustering_cqm(G, 3, sampler=MI355XSampler(), sampler_kwargs=dict(num_reads=64, num_sweeps=300, seed=3))
    best = ss.first.sample
    sizes = np.bincount(list(best.values()), minlength=3)
    assert sizes.min() >= 20 and sizes.sum() == 256
    # the reference's objective on the one-hot expansion: sum over edges and cases of v_i + v_j - 2 w v_i v_j
    v = clustering.one_hot_sample(best, 3)
    obj = 0.0
    for a, b, d in G.edges(data=True):
        for k in range(3):
            obj += v["v_%s,%d" % (a, k)] + v["v_%s,%d" % (b, k)] - 2 * d["weight"] * v["v_%s,%d" % (a, k)] * v["v_%s,%d" % (b, k)]
    assert ss.first.energy == pytest.approx(obj, rel=1e-12)
    assert sum(v.values()) == 256                                   # one case per node
    # heavy edges stay inside clusters: the two circles are separated (cut 0 between components is optimal)
    comp = fx.components()
    lab = np.array([best[v] for v in fx.nodes])
    cut_w = sum(w for a, b, w in zip(fx.eu, fx.ev, fx.w) if lab[a] != lab[b])
    assert cut_w < 0.15 * fx.W                                      # most of the edge weight is inside clusters
    assert all(len(set(lab[comp == c])) <= 2 for c in np.unique(comp))


def test_subsampling_and_independent_set_models():
    fx = load_fixture("aniso")
    G = fx.graph()
    m = models.build_subsampling_qubo(G, 0.3)
    Q = {}                                                           # literal QA_subsampling.py:28-35
    for u, v, d in G.edges(data=True):
        Q[(u, u)] = Q.get((u, u), 0) - (1 - d["weight"])
        Q[(v, v)] = Q.get((v, v), 0) - (1 - d["weight"])
        Q[(u, v)] = Q.get((u, v), 0) + (1 - d["weight"])
    for i in G.nodes:
        Q[(i, i)] = Q.get((i, i), 0) + 0.3
    x = np.random.RandomState(0).randint(0, 2, size=(5, 256))
    ref = models.qubo_dict_to_model(Q)
    assert np.allclose(m.energies(x), ref.energies(x[:, [m.variables.index(v) for v in ref.variables]]), rtol=1e-12)
    resp = clustering.graph_subsampling(G, 0.3, sampler=MI355XSampler(), sampler_kwargs=dict(num_reads=64, num_sweeps=200, seed=1))
    assert set(d["label1"] for _, d in G.nodes(data=True)) <= {0, 1}
    assert resp.first.energy == pytest.approx(float(m.energies(np.array([[resp.first.sample[v] for v in m.variables]]))[0]))
    S = clustering.graph_subsampling_2(G, sampler=MI355XSampler(), sampler_kwargs=dict(num_reads=128, num_sweeps=400, seed=2))
    inside = set(S)
    assert len(S) > 20 and not any(a in inside and b in inside for a, b in G.edges)   # an independent set
