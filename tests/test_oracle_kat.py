"""Pins the CPU oracle: Philox known-answer vectors, the -ln(u) polynomial, and the model-side
known-answer values of SURVEY.md section 8c on the reference's bundled graphs (CPU only)."""
import math

import numpy as np
import pytest

from conftest import GRAPH_NAMES, load_fixture
from oracle import model_oracle as mo
from oracle import sa_oracle as so

# constants copied from SURVEY.md section 8c (computed there from BQM_clustering.py:29-47 literally)
SURVEY = {
    "noisy_circles": dict(m=2382, W=922.4408936436804, gamma=0.18016423703978135, lenQ=32896,
                          half_E=792.3621025689054, half_cut_w=468.0216202785854, half_cut=1202,
                          comp0_E=-2951.8108596597776, comp0_E_dict=-2951.8108596599504),
    "blobs": dict(m=3417, W=921.076680715226, gamma=0.17989778920219257, half_E=866.3881964612224,
                  half_cut=1755, comp0_E=-2630.1056781360553, bound=-2947.445378288723),
    "noisy_moons": dict(m=2354, W=953.2210584393245, gamma=0.18617598797643056,
                        half_E=743.0374009890934, half_cut=1151),
    "aniso": dict(m=2865, W=956.0196099909722, gamma=0.18672258007636178,
                  half_E=800.1594371228853, half_cut=1442),
    "varied": dict(m=3362, W=900.958180794094, gamma=0.1759683946863465,
                   half_E=766.8495259813803, half_cut=1674),
    "no_structure": dict(m=3098, W=890.9001333204122, gamma=0.17400393228914302,
                         half_E=846.4577460378978, half_cut=1582),
}


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert so.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert so.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                            [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    out = so.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)
    assert out[0] == 0x408f276d and out[1] == 0x41c83b0e and out[3] == 0x6d5451fd


def test_chain_word_addressing():
    # word(i, s, g, tag) = philox(ctr=((i>>8)<<6 | i&63, s, g, tag), key=seed)[(i>>6)&3]
    seed = 0x0123456789abcdef
    for i in (0, 63, 64, 255, 256, 300, 2637):
        blk = ((i >> 8) << 6) | (i & 63)
        ref = so.philox4x32_10([blk, 7, 11, 0], [seed & 0xffffffff, seed >> 32])[(i >> 6) & 3]
        assert so.chain_word(seed, i, 7, 11, 0) == ref


def test_neglog_reduction_forms_agree():
    """The device's compare-free range reduction (csrc/mi_sa_device.h neglog_u) == the oracle's, on all 2^23 inputs."""
    lib = so.lib()
    lib.orc_neglog_forms_differ.restype = __import__("ctypes").c_long
    assert lib.orc_neglog_forms_differ() == 0


def test_neglog_matches_log():
    rng = np.random.RandomState(0)
    for r in list(rng.randint(0, 2 ** 32, size=2000, dtype=np.uint64)) + [0, 1, 511, 512, 2 ** 32 - 1]:
        r = int(r)
        u = 2.0 - np.float32(np.uint32(0x3f800000 | (r >> 9)).view(np.float32))
        want = -math.log(float(u))
        got = so.neglog_u(r)
        assert abs(got - want) <= 2e-7 + 2e-7 * want
    assert so.neglog_u(0) == 0.0          # u = 1


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_bqm_model_kat(name, kat):
    fx = load_fixture(name)
    s = SURVEY[name]
    assert len(fx.nodes) == 256 and len(fx.edges) == s["m"]
    assert fx.W == s["W"]
    Q, gamma = mo.q_bqm(fx.nodes, fx.edges, 0.05, edges_weights=fx.W, k=8)
    assert gamma == s["gamma"]
    assert len(Q) == 256 * 257 // 2
    assert all((v, u) not in Q for (u, v) in Q if u != v)      # never both orientations
    half = {v: int(int(v) < 128) for v in fx.nodes}
    assert mo.cut_edges(fx.edges, half) == s["half_cut"]
    assert mo.bqm_energy_closed_form(fx.nodes, fx.edges, gamma, half) == pytest.approx(s["half_E"], rel=1e-13)
    assert mo.qubo_energy(Q, half) == pytest.approx(s["half_E"], rel=1e-9)
    # committed golden file agrees with the live restatement
    g = kat[name]
    assert g["gamma"] == gamma and g["half_cut_edges"] == s["half_cut"]
    assert g["half_E_dict"] == mo.qubo_energy(Q, half)
    # all-ones / all-zeros have zero energy (Z2-symmetric balance term)
    assert abs(mo.qubo_energy(Q, {v: 1 for v in fx.nodes})) < 1e-7
    assert mo.qubo_energy(Q, {v: 0 for v in fx.nodes}) == 0


def test_circles_component_is_global_optimum(kat):
    fx = load_fixture("noisy_circles")
    s = SURVEY["noisy_circles"]
    Q, gamma = mo.q_bqm(fx.nodes, fx.edges, 0.05, edges_weights=fx.W)
    comp = fx.components()
    x = {v: int(comp[i] == comp[0]) for i, v in enumerate(fx.nodes)}
    assert sum(x.values()) == 128 and mo.cut_edges(fx.edges, x) == 0
    assert mo.bqm_energy_closed_form(fx.nodes, fx.edges, gamma, x) == s["comp0_E"]
    assert mo.qubo_energy(Q, x) == s["comp0_E_dict"]
    assert s["comp0_E"] == -gamma * 256 ** 2 / 4                # the analytic lower bound
    # symmetry E(x) = E(1 - x)
    xc = {v: 1 - b for v, b in x.items()}
    assert mo.qubo_energy(Q, xc) == pytest.approx(mo.qubo_energy(Q, x), rel=1e-12)


def test_blobs_component(kat):
    fx = load_fixture("blobs")
    s = SURVEY["blobs"]
    _, gamma = mo.q_bqm(fx.nodes, fx.edges, 0.05, edges_weights=fx.W)
    comp = fx.components()
    sizes = sorted(np.bincount(np.unique(comp, return_inverse=True)[1]).tolist(), reverse=True)
    assert sizes == [86, 85, 85]
    x = {v: int(comp[i] == comp[0]) for i, v in enumerate(fx.nodes)}
    assert mo.bqm_energy_closed_form(fx.nodes, fx.edges, gamma, x) == pytest.approx(s["comp0_E"], rel=1e-14)
    assert -gamma * 256 ** 2 / 4 == pytest.approx(s["bound"], rel=1e-14)


def test_dqm_model_kat(kat):
    fx = load_fixture("noisy_circles")
    lin, quad = mo.dqm_model(fx.nodes, fx.edges, 3, 0.005)
    assert len(quad) == 256 * 255 // 2
    comp = fx.components()
    ids = {c: k for k, c in enumerate(sorted(set(comp.tolist())))}
    labels = {v: ids[int(comp[i])] for i, v in enumerate(fx.nodes)}
    assert sum(lin[v][0] for v in fx.nodes) == pytest.approx(107.29148216764321, rel=1e-14)
    e = mo.dqm_energy(lin, quad, labels)
    assert e == pytest.approx(-1598.8503051197176, rel=1e-12)   # SURVEY 8c (closed form)
    assert e == kat["dqm_circles"]["E_pairwise"]
    # invariance under a permutation of the case ids
    perm = {0: 2, 1: 0, 2: 1}
    assert mo.dqm_energy(lin, quad, {v: perm[c] for v, c in labels.items()}) == pytest.approx(e, rel=1e-13)


def test_bqm2_and_bqm3_models():
    fx = load_fixture("aniso")
    Q2, gamma, chain = mo.q_bqm_2(fx.nodes, fx.edges, 0.01, 1, weights_sum=fx.W)
    assert gamma == (fx.W / 256) * 0.01
    assert len(Q2) == 256 + len(fx.edges)
    x = {v: int(int(v) % 3 == 0) for v in fx.nodes}
    cut_w = sum(w for u, v, w in fx.edges if x[u] != x[v])
    assert mo.qubo_energy(Q2, x) == pytest.approx(1 * cut_w + gamma * sum(x.values()), rel=1e-12)
    Q3 = mo.q_bqm_3_cut_only(fx.nodes, fx.edges)
    assert mo.qubo_energy(Q3, x) == pytest.approx(8 * cut_w, rel=1e-12)
    assert chain == pytest.approx(np.mean(fx.w) * (2 * len(fx.edges) / 256) * 2, rel=1e-12)


@pytest.mark.parametrize("key,min_E,argmin,cut,second", [
    ("brute_noisy_moons_gf1p0", -224.62017262063736, 312072, 28, -220.383314669),
    ("brute_noisy_moons_gf0p05", 0.0, 0, 0, 21.1778),
    ("brute_aniso_gf0p05", -4.085552102696067, 32832, 2, 0.0),
])
def test_bruteforce_kat(kat, key, min_E, argmin, cut, second):
    g = kat[key]
    assert g["min_E"] == pytest.approx(min_E, abs=1e-9)
    assert g["argmin"] == argmin and g["cut_edges"] == cut and g["num_min"] == 2
    assert g["second_E"] == pytest.approx(second, abs=2e-4)


def test_bruteforce_live_small():
    rng = np.random.RandomState(3)
    for n in (1, 2, 7, 12):
        A = rng.normal(size=(n, n))
        Qs = (A + A.T) / 2
        mn, am, nm, se = so.bruteforce_qubo(Qs, offset=0.5)
        best = min(float(np.array([(k >> i) & 1 for i in range(n)]) @ Qs @
                         np.array([(k >> i) & 1 for i in range(n)])) for k in range(2 ** n))
        assert mn == pytest.approx(best + 0.5, abs=1e-9)
        x = np.array([(am >> i) & 1 for i in range(n)])
        assert float(x @ Qs @ x) + 0.5 == pytest.approx(mn, abs=1e-9)


def _dense_qs(fx, gf=0.05):
    from scrna_seq_qannealing_clustering_amd import models
    m = models.build_bqm_qubo(fx.graph(), gf)
    return m, m.dense_Qs()


def test_oracle_philox_chain_reaches_circles_optimum():
    """P3 at the reference's own scale: explicit beta range (SURVEY 8c trap), 64 reads x 1000 sweeps."""
    from scrna_seq_qannealing_clustering_amd import models
    fx = load_fixture("noisy_circles")
    m, Qs = _dense_qs(fx)
    hot, cold = models.default_beta_range(m)
    assert hot == pytest.approx(6.264e-3, rel=2e-3) and cold == pytest.approx(25.56, rel=2e-3)
    betas = models.make_beta_schedule(1000, (hot, cold))
    st, en, stats = so.sa_dense_philox(Qs.astype(np.float32), 64, betas, 1234)
    assert en.min() == pytest.approx(-2951.8108596597776, rel=1e-6)
    best = st[int(np.argmin(en))]
    assert int(so.cut_edges(fx.eu, fx.ev, best[None, :])[0]) == 0 and best.sum() == 128
    assert 0.05 < stats[1] / stats[0] < 0.6
    # energies are a faithful fp64 re-evaluation
    assert np.allclose(en, so.energy_dense_f64(Qs.astype(np.float32), st), rtol=0, atol=1e-9)


def test_oracle_neal_chain_reaches_circles_optimum():
    from scrna_seq_qannealing_clustering_amd import models
    fx = load_fixture("noisy_circles")
    m, Qs = _dense_qs(fx)
    h, J, off = so.qubo_to_ising_dense(Qs)
    # Ising form is equivalent: E_qubo(x) = E_ising(2x - 1) + off
    rng = np.random.RandomState(1)
    x = rng.randint(0, 2, size=256)
    s = 2 * x - 1
    e_is = float(h @ s + 0.5 * s @ J @ s)
    assert e_is + off == pytest.approx(float(x @ Qs @ x), rel=1e-10, abs=1e-8)
    assert np.max(np.abs(h)) < 1e-9 * np.max(np.abs(J))        # the h ~ 0 trap of SURVEY 8c
    betas = models.make_beta_schedule(1000, models.default_beta_range(m))
    st, en, stats = so.sa_ising_neal_dense(h, J, 32, betas, seed=1234)
    assert (en + off).min() == pytest.approx(-2951.8108596597776, rel=1e-9)
    # CSR form of the same model follows the same trajectory (same RNG stream, same order)
    nbr_ptr = np.arange(0, 256 * 255 + 1, 255)
    nbr = np.array([j for i in range(256) for j in range(256) if j != i])
    nJ = np.array([J[i, j] for i in range(256) for j in range(256) if j != i])
    init = 2 * np.random.RandomState(5).randint(0, 2, size=(3, 256)) - 1
    a, ea, _ = so.sa_ising_neal_dense(h, J, 3, betas[:50], seed=77, init_spins=init)
    b, eb, _ = so.sa_ising_neal_csr(h, nbr_ptr, nbr, nJ, 3, betas[:50], seed=77, init_spins=init)
    assert np.array_equal(a, b) and np.allclose(ea, eb, rtol=1e-12)


def test_oracle_matches_bruteforce_small(kat):
    """P4: the Philox chain finds the exact optimum of the 20-node KAT model."""
    from scrna_seq_qannealing_clustering_amd import models
    g = kat["brute_noisy_moons_gf1p0"]
    fx = load_fixture("noisy_moons")
    keep = set(g["nodes"])
    edges = [(u, v, w) for u, v, w in fx.edges if u in keep and v in keep]
    Q, gamma = mo.q_bqm(g["nodes"], edges, 1.0, edges_weights=g["W"])
    m = models.qubo_dict_to_model(Q)
    assert m.variables[0] == edges[0][0]        # first-appearance order
    Qs = m.dense_Qs()
    betas = models.make_beta_schedule(200, models.default_beta_range(m))
    st, en, _ = so.sa_dense_philox(Qs.astype(np.float32), 32, betas, 7)
    assert en.min() == pytest.approx(g["min_E"], abs=1e-4)


def test_oracle_weighted_pair_term():
    """Chain (2b) with pair-term weights (oracle/sa_oracle.c: orc_sa_csr_rank1_philox_w; the structured form of
    BQM_clustering.py:373-380's squared size window): with every weight 1 it IS chain (2b), bit for bit; with weights its
    energies are those of the dense matrix c a_i a_j + S_ij, and on a 14-variable instance a long cold run ends in the
    brute-force optimum."""
    from scrna_seq_qannealing_clustering_amd import models
    rs = np.random.RandomState(11)
    n = 14
    edges = [(i, j) for i in range(10) for j in range(i + 1, 10) if rs.rand() < 0.35]
    w = rs.uniform(0.2, 1.0, size=len(edges))
    base = models.QuboModel(list(range(10)), *models._cut_qubo_parts(10, np.array([e[0] for e in edges]), np.array([e[1] for e in edges]), w, 8)[:1],
                            *models._csr_from_edges(10, np.array([e[0] for e in edges]), np.array([e[1] for e in edges]),
                                                    models._cut_qubo_parts(10, np.array([e[0] for e in edges]), np.array([e[1] for e in edges]), w, 8)[1]))
    pen = models.add_size_window_penalty(base, lb=2, ub=10.0, lagrange_multiplier=1.5)      # slack range 8: weights 1, 2, 4, 1
    assert pen.weights is not None and pen.num_variables == n and sorted(pen.weights[10:].tolist()) == [1, 1, 2, 4]
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    betas = np.geomspace(0.05, 30.0, 200)
    args = (pen.rowptr, pen.col, f32(pen.val), f32(pen.lin), float(np.float32(pen.c_pair)), 16, betas, 5)
    st, en, stats = so.sa_csr_rank1_philox(*args, offset=pen.offset, weights=pen.weights)
    assert np.allclose(en, pen.energies(st), rtol=1e-5, atol=1e-5)
    Qs = pen.dense_Qs()
    allx = ((np.arange(1 << n)[:, None] >> np.arange(n)) & 1).astype(np.float64)
    brute = np.einsum("ri,ij,rj->r", allx, Qs, allx) + pen.offset
    assert en.min() == pytest.approx(brute.min(), rel=1e-5, abs=1e-5)
    assert np.allclose(pen.energies(allx[:64]), brute[:64], rtol=1e-12)
    # unit weights: the unweighted chain, bit for bit
    s1 = so.sa_csr_rank1_philox(*args, weights=np.ones(n, dtype=np.int32))
    s0 = so.sa_csr_rank1_philox(*args)
    assert np.array_equal(s1[0], s0[0]) and np.array_equal(s1[1], s0[1]) and np.array_equal(s1[2], s0[2])
