"""Host-side model layer (array-form builders) against the literal restatement of the reference's
builders in oracle/model_oracle.py.  CPU only."""
import numpy as np
import pytest

from conftest import GRAPH_NAMES, load_fixture
from oracle import model_oracle as mo
from scrna_seq_qannealing_clustering_amd import models
from scrna_seq_qannealing_clustering_amd.bqm import BinaryQuadraticModel, DiscreteQuadraticModel
from scrna_seq_qannealing_clustering_amd.sampler import dqm_to_potts


def dense_from_dict(Q, variables):
    idx = {v: i for i, v in enumerate(variables)}
    n = len(variables)
    Qs = np.zeros((n, n))
    for (u, v), b in Q.items():
        i, j = idx[u], idx[v]
        if i == j:
            Qs[i, i] += b
        else:
            Qs[i, j] += b / 2
            Qs[j, i] += b / 2
    return Qs


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_build_bqm_matches_reference_dict(name):
    fx = load_fixture(name)
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    Q, gamma = mo.q_bqm(fx.nodes, fx.edges, 0.05, edges_weights=fx.W)
    assert m.info["gamma"] == gamma and m.info["W"] == fx.W
    assert m.variables == fx.nodes
    ref = dense_from_dict(Q, fx.nodes)
    got = m.dense_Qs()
    assert np.array_equal(got, got.T)
    assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)
    # energies through the structured (CSR + uniform pair) form == dict sum
    rng = np.random.RandomState(0)
    X = rng.randint(0, 2, size=(5, 256))
    want = [mo.qubo_energy(Q, dict(zip(fx.nodes, x.tolist()))) for x in X]
    assert np.allclose(m.energies(X), want, rtol=1e-10, atol=1e-8)


def test_build_bqm2_bqm3_match_reference_dict():
    fx = load_fixture("varied")
    G = fx.graph()
    m2 = models.build_bqm2_qubo(G, 0.01, 1)
    Q2, gamma, chain = mo.q_bqm_2(fx.nodes, fx.edges, 0.01, 1, weights_sum=fx.W)
    assert m2.info["gamma"] == gamma
    assert m2.info["chain_strength"] == pytest.approx(chain, rel=1e-13)
    assert np.allclose(m2.dense_Qs(), dense_from_dict(Q2, fx.nodes), rtol=1e-13, atol=1e-13)
    m3 = models.build_bqm3_cut_qubo(G)
    assert np.allclose(m3.dense_Qs(), dense_from_dict(mo.q_bqm_3_cut_only(fx.nodes, fx.edges), fx.nodes),
                       rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", ["noisy_circles", "blobs"])
def test_build_dqm_matches_reference_overwrite_semantics(name):
    fx = load_fixture(name)
    K, gamma = 3, 0.005
    pm = models.build_dqm_potts(fx.graph(), K, gamma)
    lin, quad = mo.dqm_model(fx.nodes, fx.edges, K, gamma)
    assert np.array_equal(pm.lin, np.array([lin[v][0] for v in fx.nodes]))   # last-edge-wins
    rng = np.random.RandomState(1)
    L = rng.randint(0, K, size=(4, 256))
    want = [mo.dqm_energy(lin, quad, dict(zip(fx.nodes, l.tolist()))) for l in L]
    assert np.allclose(pm.energies(L), want, rtol=1e-10)


def test_dqm_lookalike_reduces_to_potts():
    fx = load_fixture("noisy_moons")
    keep = fx.nodes[:24]
    ks = set(keep)
    edges = [(u, v, w) for u, v, w in fx.edges if u in ks and v in ks]
    K, gamma = 4, 0.3
    # build exactly as DQM_clustering.py:29-43 does, against the look-alike class
    from itertools import combinations
    dqm = DiscreteQuadraticModel()
    for node in keep:
        dqm.add_variable(K, label=node)
    for node in keep:
        dqm.set_linear(node, [gamma * (1 - len(keep) / K) for _ in range(K)])
    for i, j in combinations(keep, 2):
        dqm.set_quadratic(i, j, {(c, c): 2 * gamma for c in range(K)})
    for u, v, w in edges:
        dqm.set_quadratic(u, v, {(c, c): -2 * w for c in range(K)})
        dqm.set_linear(u, [w for _ in range(K)])
        dqm.set_linear(v, [w for _ in range(K)])
    pm = dqm_to_potts(dqm)
    assert pm.c_pair == 2 * gamma and pm.num_cases == K
    lin, quad = mo.dqm_model(keep, edges, K, gamma)
    rng = np.random.RandomState(2)
    for _ in range(5):
        lab = dict(zip(keep, rng.randint(0, K, size=len(keep)).tolist()))
        e_ref = mo.dqm_energy(lin, quad, lab)
        assert dqm.energy(lab) == pytest.approx(e_ref, rel=1e-12)
        assert pm.energies(np.array([[lab[v] for v in keep]]))[0] == pytest.approx(e_ref, rel=1e-10)
    # a model that is not in Potts form is refused, not silently mangled
    dqm.set_quadratic(keep[0], keep[1], {(0, 1): 1.0})
    with pytest.raises(NotImplementedError):
        dqm_to_potts(dqm)


def test_qubo_dict_roundtrip_and_uniform_detection():
    fx = load_fixture("aniso")
    Q, gamma = mo.q_bqm(fx.nodes, fx.edges, 0.05, edges_weights=fx.W)
    m = models.qubo_dict_to_model(Q)
    assert m.c_pair == 2 * gamma                           # the 2*gamma pair term is split off
    assert len(m.val) == 2 * len(fx.edges)                 # sparse part = edges only
    # variable order = first appearance (u of the first edge, then v, ...)
    assert m.variables[0] == fx.edges[0][0]
    assert np.allclose(m.dense_Qs(), dense_from_dict(Q, m.variables), rtol=1e-12, atol=1e-12)
    # sparse Q (QA_subsampling.py:28-35 shape): no uniform term
    Qs = {}
    for u, v, w in fx.edges[:50]:
        Qs[(u, u)] = Qs.get((u, u), 0) - (1 - w)
        Qs[(v, v)] = Qs.get((v, v), 0) - (1 - w)
        Qs[(u, v)] = Qs.get((u, v), 0) + (1 - w)
    ms = models.qubo_dict_to_model(Qs)
    assert ms.c_pair == 0.0
    x = {v: 1 for v in ms.variables}
    assert ms.energies(np.ones((1, ms.num_variables)))[0] == pytest.approx(mo.qubo_energy(Qs, x), rel=1e-12)
    # both orientations of a pair are merged
    m2 = models.qubo_dict_to_model({("a", "b"): 1.0, ("b", "a"): 2.0, ("a", "a"): -1.0})
    assert m2.energies(np.array([[1, 1]]))[0] == pytest.approx(2.0)


def test_default_beta_range_guard_and_schedule():
    fx = load_fixture("noisy_circles")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    hot, cold = models.default_beta_range(m)
    gamma = m.info["gamma"]
    assert cold == pytest.approx(np.log(100) / (2 * (gamma / 2)), rel=1e-9)    # min |J| = gamma/2
    # dense-backed copy gives the same range
    md = models.qubo_dict_to_model(m.to_qubo_dict(), detect_uniform=False)
    h2, c2 = models.default_beta_range(md)
    assert h2 == pytest.approx(hot, rel=1e-9) and c2 == pytest.approx(cold, rel=1e-9)
    b = models.make_beta_schedule(10, (0.1, 10.0))
    assert b[0] == pytest.approx(0.1) and b[-1] == pytest.approx(10.0) and len(b) == 10
    assert np.allclose(b[1:] / b[:-1], (10.0 / 0.1) ** (1 / 9))
    b2 = models.make_beta_schedule(12, (0.1, 1.0), "linear", num_sweeps_per_beta=3)
    assert len(b2) == 12 and len(set(b2.tolist())) == 4
    assert np.array_equal(models.make_beta_schedule(0, (1, 2), "custom", 2, [1.0, 2.0]), [1, 1, 2, 2])
    with pytest.raises(ValueError):
        models.make_beta_schedule(10, (0.1, 1.0), num_sweeps_per_beta=3)
    with pytest.raises(ValueError):
        models.make_beta_schedule(10, (0.0, 1.0))
    with pytest.raises(ValueError):
        models.make_beta_schedule(10, (0.1, 1.0), "exotic")


def test_size_window_penalty_matches_bqm_lookalike():
    """BQM_clustering.py:371-380: from_qubo + add_linear_inequality_constraint (slack bits)."""
    fx = load_fixture("blobs")
    keep = fx.nodes[:30]
    ks = set(keep)
    eu = [(u, v, w) for u, v, w in fx.edges if u in ks and v in ks]
    Q = mo.q_bqm_3_cut_only(keep, eu)
    bqm = BinaryQuadraticModel.from_qubo(Q)
    slack = bqm.add_linear_inequality_constraint([(v, 1) for v in bqm.variables], lb=3, ub=30 / 6,
                                                 lagrange_multiplier=0.7, label="c1_constraint")
    assert [c for _, c in slack] == [1, 1]                    # int(5 - 3) = 2 -> floor(log2 2) = 1 bit + remainder 1
    base = models.qubo_dict_to_model(Q)
    pen = models.add_size_window_penalty(base, lb=3, ub=30 / 6, lagrange_multiplier=0.7)
    assert pen.num_variables == base.num_variables + 2
    rng = np.random.RandomState(4)
    for _ in range(10):
        z = rng.randint(0, 2, size=pen.num_variables)
        sample = dict(zip(pen.variables, z.tolist()))
        assert pen.energies(z[None, :])[0] == pytest.approx(bqm.energy(sample), rel=1e-10, abs=1e-9)
    # feasible point: 4 ones + one slack bit = 5 = ub: zero penalty
    z = np.zeros(pen.num_variables, dtype=int)
    z[:4] = 1
    z[base.num_variables] = 1
    assert pen.energies(z[None, :])[0] == pytest.approx(base.energies(z[None, :base.num_variables])[0], abs=1e-9)


def test_size_window_known_answers_n256_lb40():
    """dimod's construction with the reference's own arguments (BQM_clustering.py:376-380: lb = size_limit, ub = n / 6,
    FRACTIONAL): n = 256, lb = 40 -> ub_c = 42.666..., slack bound int(2.666) = 2 -> coefficients [1, 1]; the penalty
    lagrange * (s + t0 + t1 - 128/3)^2 has its minimum lagrange / 9 at sizes 41..43 (never zero), 4/9 lagrange at 40
    and 44.  Hand-computed; dimod itself is not importable here (parity unpinned)."""
    from scrna_seq_qannealing_clustering_amd.bqm import inequality_slack
    coeffs, ub_c = inequality_slack([1] * 256, 40, 256 / 6)
    assert coeffs == [1, 1] and ub_c == 256 / 6
    lam = 0.3
    base = models.QuboModel(list(range(256)), np.zeros(256), np.zeros(257, dtype=np.int32), np.zeros(0, dtype=np.int32),
                            np.zeros(0))
    pen = models.add_size_window_penalty(base, lb=40, ub=256 / 6, lagrange_multiplier=lam)
    assert pen.num_variables == 258

    def best(size):
        out = []
        for t0 in (0, 1):
            for t1 in (0, 1):
                z = np.zeros(258, dtype=int)
                z[:size] = 1
                z[256], z[257] = t0, t1
                out.append(pen.energies(z[None, :])[0])
        return min(out)
    for size, want in ((39, lam * (128 / 3 - 41) ** 2), (40, lam * 4 / 9), (41, lam / 9), (42, lam / 9), (43, lam / 9),
                       (44, lam * 16 / 9), (45, lam * (45 - 128 / 3) ** 2)):
        assert best(size) == pytest.approx(want, rel=1e-9)
    # infeasible window raises, an always-feasible one adds nothing (dimod's behaviour)
    with pytest.raises(ValueError):
        models.add_size_window_penalty(base, lb=50, ub=256 / 6, lagrange_multiplier=lam)
    assert models.add_size_window_penalty(base, lb=-5, ub=300, lagrange_multiplier=lam).num_variables == 256



def test_synthetic_snn_shape():
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, truth = graphs.synthetic_snn(300, 5, 15, 15, 4, seed=1)
    assert len(nodes) == 300 and nodes[0] == "0"
    deg = np.bincount(np.concatenate([eu, ev]), minlength=300)
    assert deg.max() <= 15                                       # trim honoured
    allowed = {s / (10 - s) for s in range(1, 6)}
    assert all(min(abs(x - a) for a in allowed) < 1e-12 for x in set(w.tolist()))
    assert np.all(eu < ev)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    Gn = graphs.graph_from_edges(nodes, eu, ev, w)
    assert G.size(weight="weight") == pytest.approx(Gn.size(weight="weight"), rel=1e-13)
    a = models.build_bqm_qubo(G, 0.05)
    b = models.build_bqm_qubo(Gn, 0.05)
    assert np.allclose(a.dense_Qs(), b.dense_Qs(), rtol=1e-13)


def test_one_walk_graph_arrays_equal_the_networkx_calls_on_subgraph_views():
    """`graph_arrays_and_weight` replaces the reference's three graph walks (number_of_edges, edges, size) by one.  On the
    objects the recursive bisection passes down -- nested `G.subgraph(part)` views, whose node order may be the order
    of the filter's node SET -- node list, edge order, weights and `size(weight=)` must come out bit-identical, self-loops
    and weightless degree entries included; the model built from a view equals the one built from its copy."""
    import networkx as nx
    from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn
    nodes, eu, ev, w, _ = synthetic_snn(700, 5, 15, 15, 4, seed=3)
    G = graph_from_edges(nodes, eu, ev, w)
    G.add_edge(nodes[5], nodes[5], weight=0.37)
    G.add_edge(nodes[9], nodes[9], weight=1.25)
    part = [v for i, v in enumerate(G.nodes) if (i * 7) % 3 != 0]
    small = part[::5]                                             # < half of the nodes: the view iterates the SET
    for H in (G, G.subgraph(part), G.subgraph(part).subgraph(part[::2]), G.subgraph(small), nx.Graph(G.subgraph(small))):
        got = models.graph_arrays_and_weight(H)
        want = models._graph_arrays_by_calls(H)
        assert got[0] == want[0] == list(H.nodes)
        assert all(np.array_equal(a, b) for a, b in zip(got[1:4], want[1:]))
        assert got[4] == float(H.size(weight="weight"))
        assert [(got[0][a], got[0][b]) for a, b in zip(got[1], got[2])] == [(u, v) for u, v in H.edges]
    view, copy = G.subgraph(part), nx.Graph(G.subgraph(part))
    mv, mc = models.build_bqm_qubo(view, 0.05), models.build_bqm_qubo(copy, 0.05)
    assert mv.info["gamma"] == mc.info["gamma"] and np.array_equal(mv.lin, mc.lin) and np.array_equal(mv.val, mc.val)


def test_root_graph_arrays_give_every_subgraph_view_without_a_python_walk():
    """`RootGraphArrays` (one walk over the root graph, then array operations per `G.subgraph(part)` view -- what a
    recursive bisection hands down): node list, edge order, weights and total weight bit-identical to the plain walk
    and to networkx's own calls, on the root, on views of either iteration order, on views of views (networkx
    collapses them onto the root), with self-loops and integer weights; views smaller than half a neighbour list,
    copies and foreign graphs are declined (None) and take the plain walk."""
    import networkx as nx
    from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn
    nodes, eu, ev, w, _ = synthetic_snn(700, 5, 15, 15, 4, seed=3)
    G = graph_from_edges(nodes, eu, ev, w)
    G.add_edge(nodes[5], nodes[5], weight=0.37)
    G.add_edge(nodes[9], nodes[9], weight=3)
    G.add_edge(nodes[9], nodes[11], weight=2)
    arrays = models.RootGraphArrays.of(G)
    assert arrays is not None and arrays.graph is G
    part = [v for i, v in enumerate(G.nodes) if (i * 7) % 3 != 0]
    rs = np.random.RandomState(4)
    shuffled = [part[i] for i in rs.permutation(len(part))]
    views = [G, G.subgraph(part), G.subgraph(shuffled), G.subgraph(part).subgraph(part[::2]), G.subgraph(part[::5]),
             G.subgraph(part).subgraph(part[::2]).subgraph(part[::4]), G.subgraph([nodes[5], nodes[9], nodes[11]] + part[:40])]
    for H in views:
        assert models.RootGraphArrays.of(H).graph is G            # found from any of its views
        got = arrays.arrays_for(H)
        want = models.graph_arrays_and_weight(H)
        assert got is not None and got[0] == want[0] == list(H.nodes)
        assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(got[1:4], want[1:4]))
        assert got[4] == want[4] == float(H.size(weight="weight"))
        assert [(got[0][a], got[0][b]) for a, b in zip(got[1], got[2])] == [(u, v) for u, v in H.edges]
        ma, mp = models.build_bqm_qubo(H, 0.05, arrays=arrays), models.build_bqm_qubo(H, 0.05)
        assert ma.info["gamma"] == mp.info["gamma"] and np.array_equal(ma.lin, mp.lin)
        assert np.array_equal(ma.rowptr, mp.rowptr) and np.array_equal(ma.col, mp.col) and np.array_equal(ma.val, mp.val)
    hub = nx.Graph()                                              # a neighbour list more than twice the size of the view:
    hub.add_weighted_edges_from((0, i, 1.0 + i) for i in range(1, 40))   # networkx walks the node SET there -- declined
    hub.add_weighted_edges_from((i, i + 1, 0.5) for i in range(1, 39))
    ha = models.RootGraphArrays.of(hub)
    tiny = hub.subgraph([0, 3, 4, 17, 18])
    assert ha.arrays_for(tiny) is None
    got, want = models.graph_arrays_and_weight(tiny, ha), models.graph_arrays_and_weight(tiny)
    assert got[0] == want[0] and all(np.array_equal(a, b) for a, b in zip(got[1:4], want[1:4])) and got[4] == want[4]
    assert arrays.arrays_for(nx.Graph(G.subgraph(part))) is None      # a copy is a graph of its own
    assert arrays.arrays_for(hub.subgraph(range(30))) is None         # a view of another graph
    unweighted = nx.path_graph(5)
    assert models.RootGraphArrays.of(unweighted) is None              # no weights: the plain walk decides (it raises)
    assert models.RootGraphArrays.of(nx.DiGraph()) is None


def test_qubo_dict_with_exotic_labels_keeps_them():
    """Labels are arbitrary hashables: None, tuples, ints and strings side by side come back as given, in
    order of first appearance, whichever of the two lifting paths handles the dict."""
    Q = {(None, None): 1.0, (None, 3): 2.0, (3, "a"): -1.0, (("t", 1), "a"): 0.5, (("t", 1), ("t", 1)): 0.25}
    m = models.qubo_dict_to_model(Q)
    assert m.variables == [None, 3, "a", ("t", 1)]
    assert m.lin.tolist() == [1.0, 0.0, 0.0, 0.25]
    x = np.array([[1, 1, 0, 1], [0, 1, 1, 1]], dtype=np.uint8)
    assert np.allclose(m.energies(x), [1.0 + 2.0 + 0.25, -1.0 + 0.5 + 0.25])
    Q2 = {("b", "b"): 1.0, ("a", "b"): 2.0, ("a", "a"): -3.0}
    assert models.qubo_dict_to_model(Q2).variables == ["b", "a"]


def test_qubo_dict_with_a_squared_constraint_is_recognised_as_weighted_rank_one():
    """`clustering_bqm_3` hands its sampler a BQM whose quadratic part is dense after add_linear_inequality_constraint
    (BQM_clustering.py:371-386).  `sampler.sample(bqm)` sees only the dict: qubo_dict_to_model splits off the uniform pair
    term AND the slack bits' weights, and ends at the structured model models.add_size_window_penalty builds directly."""
    fx = load_fixture("blobs")
    keep = fx.nodes[:60]
    ks = set(keep)
    eu = [(u, v, w) for u, v, w in fx.edges if u in ks and v in ks]
    Q = mo.q_bqm_3_cut_only(keep, eu)
    bqm = BinaryQuadraticModel.from_qubo(Q)
    slack = bqm.add_linear_inequality_constraint([(v, 1) for v in bqm.variables], lb=4, ub=60 / 2.5,
                                                 lagrange_multiplier=0.7, label="c1_constraint")
    assert sorted(c for _, c in slack) == [1, 2, 4, 5, 8]                    # 20 slack units
    Qd = {(v, v): b for v, b in bqm.linear.items()}
    Qd.update(bqm.quadratic)
    got = models.qubo_dict_to_model(Qd, offset=bqm.offset)
    assert got._dense is None and got.c_pair == pytest.approx(1.4) and got.weights is not None
    assert sorted(got.weights[got.weights != 1].tolist()) == [2, 4, 5, 8]
    assert int(np.diff(got.rowptr).max()) < 30                               # the slack bits kept no sparse couplings
    assert np.all(np.diff(got.rowptr)[got.weights != 1] == 0)
    want = models.add_size_window_penalty(models.qubo_dict_to_model(Q), lb=4, ub=60 / 2.5, lagrange_multiplier=0.7)
    rng = np.random.RandomState(2)
    order = [got.variables.index(v) for v in want.variables]
    for _ in range(10):
        z = rng.randint(0, 2, size=want.num_variables)
        zg = np.zeros_like(z)
        zg[order] = z
        e = bqm.energy(dict(zip(want.variables, z.tolist())))
        assert got.energies(zg[None, :])[0] == pytest.approx(e, rel=1e-10, abs=1e-9)
        assert want.energies(z[None, :])[0] == pytest.approx(e, rel=1e-10, abs=1e-9)
    # a dict that only looks similar (one pair off) stays as it was
    Qd2 = dict(Qd)
    k0 = next(k for k in Qd2 if k[0] != k[1] and "slack" in str(k[0]) + str(k[1]) and abs(Qd2[k] - 1.4) > 1.0)   # a weighted pair
    Qd2[k0] += 0.01
    assert models.qubo_dict_to_model(Qd2, offset=bqm.offset).weights is None
