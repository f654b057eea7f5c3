"""Parity of the structured kernels (K2: CSR + uniform pair term, K3: Potts / DQM) against the oracle,
through the C ABI.  GPU only.  States / labels / accepted-move counts / edge cuts are bit-exact."""
import numpy as np
import pytest

from conftest import GRAPH_NAMES, load_fixture
from oracle import model_oracle as mo
from oracle import sa_oracle as so
from scrna_seq_qannealing_clustering_amd import MI355XSampler, _lib, models
from scrna_seq_qannealing_clustering_amd.bqm import DiscreteQuadraticModel
from scrna_seq_qannealing_clustering_amd.engine import Problem

pytestmark = pytest.mark.gpu


def f32(x):
    return np.asarray(x, dtype=np.float32)


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_csr_rank1_trajectory_parity(name):
    fx = load_fixture(name)
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    betas = models.make_beta_schedule(40, models.default_beta_range(m))
    args = (m.rowptr, m.col, f32(m.val), f32(m.lin), float(np.float32(m.c_pair)))
    init = np.random.RandomState(2).randint(0, 2, size=(9, 256)).astype(np.uint8)
    for kw in (dict(), dict(init=init, resync_interval=5)):
        ost, oen, ostats = so.sa_csr_rank1_philox(*args, 9, betas, 1234, replica_offset=3, **kw)
        with Problem.csr_rank1(*args) as p:
            p.anneal(9, betas, 1234, replica_offset=3, initial_states=kw.get("init"),
                     resync_interval=kw.get("resync_interval", 0))
            st, en, info = p.fetch()
            idx, e_best, key, s_best = p.best()
        assert np.array_equal(st, ost)
        assert info["accepted"] == int(ostats[1])
        assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)
        assert np.array_equal(so.cut_edges(fx.eu, fx.ev, st), so.cut_edges(fx.eu, fx.ev, ost))
        assert en[idx] == en.min() and np.array_equal(s_best, st[idx])
        # same model as the dense form: energies agree with the fp64 coefficients
        assert np.allclose(en, m.energies(st), rtol=1e-5)


def test_csr_rank1_reaches_circles_optimum(kat):
    fx = load_fixture("noisy_circles")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    betas = models.make_beta_schedule(1000, models.default_beta_range(m))
    with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), float(np.float32(m.c_pair))) as p:
        p.anneal(64, betas, 1234)
        st, en, _ = p.fetch()
    best = st[int(np.argmin(en))]
    assert m.energies(best[None, :])[0] == pytest.approx(kat["noisy_circles"]["comp0_E_closed"], rel=1e-12)
    assert int(so.cut_edges(fx.eu, fx.ev, best[None, :])[0]) == 0


def test_csr_rank1_sparse_only_model_and_ragged_n():
    """A2-shaped model (no uniform pair term) on an induced subgraph with n not a multiple of 64."""
    fx = load_fixture("varied")
    keep = fx.nodes[:150]
    G = fx.graph().subgraph(keep)
    m = models.build_bqm2_qubo(G, 0.01, 1)
    assert m.c_pair == 0.0 and m.num_variables == 150
    betas = np.geomspace(0.1, 20.0, 25)
    args = (m.rowptr, m.col, f32(m.val), f32(m.lin), 0.0)
    ost, oen, ostats = so.sa_csr_rank1_philox(*args, 5, betas, 9)
    with Problem.csr_rank1(*args) as p:
        p.anneal(5, betas, 9)
        st, en, info = p.fetch()
    assert np.array_equal(st, ost) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)


def test_structured_kernels_beyond_4096_variables():
    """n = 5000 (> 64 slots: several state masks in K2, > 4096 labels in K3): 3 replicas vs the oracle."""
    from scrna_seq_qannealing_clustering_amd import graphs
    from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
    nodes, eu, ev, w, _ = graphs.synthetic_snn(5000, 5, 15, 15, 12, seed=3)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    m = models.build_bqm_qubo(G, 0.05)
    betas = models.make_beta_schedule(6, models.default_beta_range(m))
    args = (m.rowptr, m.col, f32(m.val), f32(m.lin), float(np.float32(m.c_pair)))
    ost, oen, ostats = so.sa_csr_rank1_philox(*args, 3, betas, 21, resync_interval=4)
    with Problem.csr_rank1(*args) as p:
        p.anneal(3, betas, 21, resync_interval=4)
        st, en, info = p.fetch()
    assert np.array_equal(st, ost) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)
    pm = models.build_dqm_potts(G, 15, 0.005)
    pb = models.make_beta_schedule(6, default_potts_beta_range(pm))
    pargs = (pm.rowptr, pm.col, f32(pm.val), float(np.float32(pm.c_pair)), 5000, 15)
    olab, oen, ostats = so.potts_csr_philox(*pargs, 3, pb, 22, lin_offset=pm.lin_offset)
    with Problem.potts_csr(*pargs, lin_offset=pm.lin_offset) as p:
        p.anneal(3, pb, 22)
        lab, en, info = p.fetch()
    assert np.array_equal(lab, olab) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)


def _sparse_ring_model(n, seed):
    """A large sparse symmetric model built directly in CSR (the dense SNN surrogate is O(n^2)): ring
    neighbours i+-1, i+-2, i+-64, i+-65 (so every variable has neighbours inside AND outside its own
    64-variable slot) plus random chords, SNN-like weights s/(2k-s), the uniform pair term of A1."""
    rng = np.random.RandomState(seed)
    eu = np.concatenate([np.arange(n)] * 4 + [rng.randint(0, n, size=3 * n)])
    ev = np.concatenate([(np.arange(n) + d) % n for d in (1, 2, 64, 65)] + [rng.randint(0, n, size=3 * n)])
    keep = eu != ev
    lo, hi = np.minimum(eu[keep], ev[keep]), np.maximum(eu[keep], ev[keep])
    key = np.unique(lo.astype(np.int64) * n + hi)
    lo, hi = (key // n).astype(np.int32), (key % n).astype(np.int32)
    w = rng.choice(np.array([1 / 9, 2 / 8, 3 / 7, 4 / 6, 1.0]), size=len(lo))
    rowptr, col, val = models._csr_from_edges(n, lo, hi, -16.0 * w)
    deg = np.zeros(n)
    np.add.at(deg, lo, w)
    np.add.at(deg, hi, w)
    gamma = 0.05 * w.sum() / n
    lin = 8.0 * deg + gamma * (1 - n)
    return rowptr, col, f32(val), f32(lin), float(np.float32(2 * gamma))


@pytest.mark.parametrize("n,R,sweeps", [(20000, 3, 3), (45003, 2, 2)])
def test_csr_rank1_large_models(n, R, sweeps):
    """K2 at large n (BASELINE config 4 is n = 50000): the only per-replica memory is one state bit per
    variable in LDS, so the size is bounded by nothing else.  Bit-exact against the oracle (rows of up to 28
    neighbours: the 32-wide adjacency layout)."""
    args = _sparse_ring_model(n, seed=n)
    assert int(np.diff(args[0]).max()) <= 64
    betas = np.geomspace(0.002, 0.5, sweeps)
    ost, oen, ostats = so.sa_csr_rank1_philox(*args, R, betas, 5, replica_offset=2, resync_interval=2)
    with Problem.csr_rank1(*args) as p:
        p.anneal(R, betas, 5, replica_offset=2, resync_interval=2)
        st, en, info = p.fetch()
    assert info["accepted"] == int(ostats[1]) and info["accepted"] > n // 4
    assert np.array_equal(st, ost)
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)


def test_csr_rank1_slot_independent_order_and_integer_fast_path():
    """order="slots" renumbers the variables so that a wavefront's 64 variables are mutually non-adjacent;
    such slots take the kernel's integer fast path (the accept rule of every lane as a bound on sum(x), found
    by bisecting the exact fp32 predicate).  The run equals the oracle on the SAME renumbered model, flip for
    flip, and states come back in the caller's order.  Hot and cold schedules, both signs of c_pair."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, _ = graphs.synthetic_snn(900, 5, 15, 15, 5, seed=4)
    m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
    perm = models.slot_independent_order(m.rowptr, m.col)
    rp, cc, vv = models.permute_csr(m.rowptr, m.col, m.val, perm)
    rows = np.repeat(np.arange(900), np.diff(rp))
    inside = (rows >> 6) == (cc >> 6)
    assert inside.mean() < 0.02 and len(np.unique(rows[inside] >> 6)) <= 3       # (nearly) no edge inside a 64-block
    init = np.random.RandomState(1).randint(0, 2, size=(40, 900)).astype(np.uint8)
    for c_pair, betas in ((float(np.float32(m.c_pair)), np.geomspace(1e-3, 30.0, 25)),
                          (-float(np.float32(m.c_pair)), np.geomspace(1e-4, 0.5, 10)),
                          (0.0, np.geomspace(1e-2, 5.0, 10))):
        ost, oen, ostats = so.sa_csr_rank1_philox(rp, cc, f32(vv), f32(m.lin[perm]), c_pair, 40, betas, 8,
                                                  init=np.ascontiguousarray(init[:, perm]))
        with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), c_pair, order="slots") as p:
            p.anneal(40, betas, 8, initial_states=init)
            st, en, info = p.fetch()
            idx, e_best, key, s_best = p.best()
        assert info["accepted"] == int(ostats[1]) and info["accepted"] > 0
        assert np.array_equal(st[:, perm], ost)
        assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)
        assert np.array_equal(s_best, st[idx])


def test_csr_rank1_padded_layout_of_a_clustered_subgraph():
    """order="padded": seats with holes, as many 64-blocks as it takes to keep every edge between blocks (what one planted
    cluster -- a leaf of the reference's recursive bisection -- needs: no packed order of it is edge-free).  A hole is a
    position with lin = +inf: it starts at 0, is never proposed, never flips.  The run equals the oracle on the SAME
    padded model, for random and for given initial states, with one and with two replicas per wavefront; states,
    energies (fp64 model) and the best state come back for the caller's n variables; proposals count n, not seats."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, lab = graphs.synthetic_snn(1500, 5, 15, 15, 6, seed=1)
    idx = np.flatnonzero(lab == 0)
    renum = -np.ones(1500, dtype=np.int64)
    renum[idx] = np.arange(len(idx))
    sel = np.isin(eu, idx) & np.isin(ev, idx)
    m = models.build_bqm_qubo(graphs.EdgeListGraph([nodes[i] for i in idx], renum[eu[sel]].astype(np.int32),
                                                   renum[ev[sel]].astype(np.int32), w[sel]), 0.05)
    n = m.num_variables
    c_pair = float(np.float32(m.c_pair))
    pos, nslots, clashes = models.padded_slot_layout(m.rowptr, m.col)
    assert clashes == 0 and nslots > (n + 63) // 64                            # the packed order would not do
    N = nslots * 64
    rp, cc, vv = models.pad_csr(m.rowptr, m.col, f32(m.val), pos, N)
    lin = np.full(N, np.inf, dtype=np.float32)
    lin[pos] = f32(m.lin)
    betas = np.geomspace(2e-3, 40.0, 30)
    init = np.random.RandomState(2).randint(0, 2, size=(9, n)).astype(np.uint8)
    init_dev = np.zeros((9, N), dtype=np.uint8)
    init_dev[:, pos] = init
    o_rand = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, 9, betas, 8, replica_offset=3)
    o_init = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, 9, betas, 8, init=init_dev)
    holes = np.setdiff1d(np.arange(N), pos)
    assert not o_rand[0][:, holes].any() and not o_init[0][:, holes].any()
    with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), c_pair, order="padded",
                           energy_model=(m.val, m.lin, m.c_pair)) as p:
        assert p.n == n and p.n_dev == N
        for mode, tw in ((2, 0), (1, 2), (1, 1)):         # a few-replica kernel; K2p alone; K2p with its threshold wavefront
            p.set_option("k2_pair", mode)
            p.set_option("k2_tw", tw)
            p.anneal(9, betas, 8, replica_offset=3)
            st, en, info = p.fetch()
            assert ("pair" in p.kernel_name()) == (mode == 1) and (mode != 1 or ("tw>" in p.kernel_name()) == (tw == 1))
            assert st.shape == (9, n) and np.array_equal(st, o_rand[0][:, pos])
            assert info["accepted"] == int(o_rand[2][1]) and info["proposals"] == 9 * 30 * n == int(o_rand[2][0])
            assert np.allclose(en, m.energies(st), rtol=1e-12)
            i_best, e_best, _, s_best = p.best()
            assert np.array_equal(s_best, st[i_best]) and e_best == pytest.approx(en.min(), rel=1e-12)
            p.anneal(9, betas, 8, initial_states=init)
            st2, en2, info2 = p.fetch()
            assert np.array_equal(st2, o_init[0][:, pos]) and info2["accepted"] == int(o_init[2][1])
    # the sampler takes this layout for structured models: same energies as the model says, every replica a valid state
    from scrna_seq_qannealing_clustering_amd import MI355XSampler
    ss = MI355XSampler().sample_qubo(m, num_reads=64, num_sweeps=200, seed=5)
    assert ss.record.sample.shape == (len(ss.record.energy), n)
    assert np.allclose(ss.record.energy, m.energies(ss.record.sample.astype(np.uint8)), rtol=1e-12)


@pytest.mark.parametrize("case", ["cluster_128", "whole_256", "whole_128", "wide_128", "whole_64", "cluster_64", "wide_64"])
def test_few_replica_kernel_workgroup_per_replica(case):
    """K2s (csrc/sparse_split_kernels.hip): a workgroup of 1 / 2 / 4 wavefronts sweeps ONE replica over a model laid out
    in edge-free blocks of 64 / 128 / 256 seats (the one-wavefront form is what runs of up to 512 replicas get by
    default; the wider ones on request).  Same chain as the oracle on the same padded model: states, accepted counts, fp64 energies, for random and given initial states, a
    replica offset, a continued run (states + sweep offset), one temperature per replica; and == K2 / K2p on the handle."""
    from scrna_seq_qannealing_clustering_amd import graphs
    block = int(case.split("_")[1])
    if case.startswith("cluster"):
        nodes, eu, ev, w, lab = graphs.synthetic_snn(1500, 5, 15, 15, 6, seed=1)
        idx = np.flatnonzero(lab == 0)
        renum = -np.ones(1500, dtype=np.int64)
        renum[idx] = np.arange(len(idx))
        sel = np.isin(eu, idx) & np.isin(ev, idx)
        G = graphs.EdgeListGraph([nodes[i] for i in idx], renum[eu[sel]].astype(np.int32), renum[ev[sel]].astype(np.int32), w[sel])
    elif case.startswith("wide"):
        nodes, eu, ev, w, _ = graphs.synthetic_snn(700, 8, 15, 30, 5, seed=3, spread=2.5)       # degree cap 30: the 32-wide layout
        G = graphs.EdgeListGraph(nodes, eu, ev, w)
    else:
        nodes, eu, ev, w, _ = graphs.synthetic_snn(1100, 5, 15, 15, 7, seed=4, spread=3.0)
        G = graphs.EdgeListGraph(nodes, eu, ev, w)
    m = models.build_bqm_qubo(G, 0.05)
    n = m.num_variables
    c_pair = float(np.float32(m.c_pair))
    pos, nblocks, clashes = models.padded_slot_layout(m.rowptr, m.col, slot=block)
    assert clashes == 0
    if block == 128 and nblocks % 2:
        nblocks += 1                                   # (whole groups of four 64-seat slots: one more block of holes)
    N = nblocks * block
    rp, cc, vv = models.pad_csr(m.rowptr, m.col, f32(m.val), pos, N)
    lin = np.full(N, np.inf, dtype=np.float32)
    lin[pos] = f32(m.lin)
    betas = np.geomspace(2e-3, 40.0, 24)
    R = 5
    init = np.random.RandomState(2).randint(0, 2, size=(R, n)).astype(np.uint8)
    init_dev = np.zeros((R, N), dtype=np.uint8)
    init_dev[:, pos] = init
    o_rand = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, R, betas, 8, replica_offset=3)
    o_init = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, R, betas, 8, init=init_dev)
    with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), c_pair, order="padded", block=block,
                           energy_model=(m.val, m.lin, m.c_pair)) as p:
        assert p.n == n and p.n_dev == N
        p.anneal(R, betas, 8, replica_offset=3)
        name = p.kernel_name()
        # few replicas: ONE wavefront sweeps a block of 64 / 128 / 256 seats per step (K2w: 1 / 2 / 4 slots), a second
        # wavefront of its workgroup computes the thresholds ahead of it ...
        wide_built = block <= 128 or "wide" not in case                          # (32 entries per variable: up to two slots per step)
        assert name == ("k_anneal_csr_rank1_wide<%d, %d, tw>" % (32 if "wide" in case else 16, block // 64) if wide_built
                        else "k_anneal_csr_rank1_split<32, 4>"), name
        st, en, info = p.fetch()
        p.set_option("k2_tw", 2)                       # ... or the sweeping wavefront does (round-3 forms: K2s with one wavefront, K2w)
        p.anneal(R, betas, 8, replica_offset=3)
        name = p.kernel_name()
        assert name.startswith("k_anneal_csr_rank1_split<" if block == 64 else "k_anneal_csr_rank1_wide<") and "tw" not in name, name
        assert name.endswith(", %d>" % (block // 64)), name
        sn_, en_, in_ = p.fetch()
        assert np.array_equal(sn_, st) and np.allclose(en_, en, rtol=1e-13) and in_["accepted"] == info["accepted"]
        p.set_option("k2_tw", 0)
        if block > 64:                                 # ... and on request K2s with 2 / 4 wavefronts per replica: the same run
            p.set_option("k2_wide", 2)
            p.anneal(R, betas, 8, replica_offset=3)
            assert p.kernel_name().startswith("k_anneal_csr_rank1_split<") and p.kernel_name().endswith(", %d>" % (block // 64))
            sw_, ew_, iw_ = p.fetch()
            assert np.array_equal(sw_, st) and np.allclose(ew_, en, rtol=1e-13) and iw_["accepted"] == info["accepted"]
            p.set_option("k2_wide", 0)
        assert np.array_equal(st, o_rand[0][:, pos]) and info["accepted"] == int(o_rand[2][1])
        assert info["proposals"] == R * len(betas) * n and np.allclose(en, m.energies(st), rtol=1e-12)
        i_best, e_best, _, s_best = p.best()
        assert np.array_equal(s_best, st[i_best]) and e_best == pytest.approx(en.min(), rel=1e-12)
        p.anneal(R, betas, 8, initial_states=init)
        st2, en2, info2 = p.fetch()
        assert np.array_equal(st2, o_init[0][:, pos]) and info2["accepted"] == int(o_init[2][1])
        # a run continued in two pieces == the run in one; then one constant temperature per replica
        p.anneal(R, betas[:9], 8, replica_offset=3)
        p.anneal(R, betas[9:], 8, replica_offset=3, continue_run=True, sweep_offset=9)
        st3, en3, _ = p.fetch()
        assert np.array_equal(st3, st) and np.array_equal(en3, en)
        per = np.geomspace(0.02, 8.0, R)
        p.anneal(R, per, 9, num_sweeps=7, sweep_offset=100)
        st4, _, info4 = p.fetch()
        o_per = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, R, per, 9, sweep_offset=100, num_sweeps=7)
        assert np.array_equal(st4, o_per[0][:, pos]) and info4["accepted"] == int(o_per[2][1])
        # the other kernels of the handle run the same chain
        for key, val, tag in (("k2_split", 2, "k_anneal_csr_rank1<"), ("k2_pair", 1, "pair")):
            p.set_option("k2_split", 2)
            p.set_option(key, val)
            p.anneal(R, betas, 8, replica_offset=3)
            assert tag in p.kernel_name() and "split" not in p.kernel_name()
            sx, ex, ix = p.fetch()
            assert np.array_equal(sx, st) and np.allclose(ex, en, rtol=1e-13) and ix["accepted"] == info["accepted"]
        # zero sweeps: the initial states come back with their energies
        p.set_option("k2_split", 1)
        p.anneal(R, betas[:0], 8, initial_states=init)
        s0, e0, _ = p.fetch()
        assert ("split" in p.kernel_name() or "wide" in p.kernel_name()) and np.array_equal(s0, init)
        assert np.allclose(e0, m.energies(init), rtol=1e-12)


@pytest.mark.parametrize("seed", range(24))
def test_few_replica_kernels_random_models(seed):
    """Random sparse models (size, degree, weights, schedule, replica count and offset drawn per seed) laid out in
    edge-free blocks of 64 / 128 / 256 seats and run on the few-replica kernels -- K2s with 1, 2 or 4 wavefronts per
    replica, K2w with 1, 2 or 4 slots per step, with and without its threshold wavefront -- against the oracle on the same padded model: states, accepted counts,
    energies; random initial states from the replica's own stream and given ones."""
    rs = np.random.RandomState(5000 + seed)
    n = int(rs.choice([70, 130, 257, 700, 1300]))
    max_deg = int(rs.choice([3, 9, 16, 17, 30]))
    m_edges = min(n * max_deg // 3, n * (n - 1) // 2)
    pairs = set()
    deg = np.zeros(n, dtype=int)
    for _ in range(4 * m_edges):
        if len(pairs) >= m_edges:
            break
        a_, b_ = (int(x) for x in rs.randint(0, n, 2))
        if a_ == b_ or (min(a_, b_), max(a_, b_)) in pairs or deg[a_] >= max_deg or deg[b_] >= max_deg:
            continue
        pairs.add((min(a_, b_), max(a_, b_)))
        deg[a_] += 1
        deg[b_] += 1
    edges = sorted(pairs)
    w = rs.choice(np.array([1 / 9, 0.25, 3 / 7, 2 / 3, 1.0, -0.5]), size=len(edges)).astype(np.float32)
    rowptr, col, val = _csr_from_edges(n, edges, w)
    lin = rs.normal(scale=0.7, size=n).astype(np.float32)
    c_pair = float(np.float32(rs.choice([0.0, 0.03, 0.4, -0.02])))
    sweeps = int(rs.choice([1, 4, 13, 30]))
    betas = np.geomspace(float(rs.choice([0.05, 0.5])), float(rs.choice([2.0, 40.0])), sweeps)
    R = int(rs.choice([1, 2, 7]))
    off = int(rs.choice([0, 5, 2 ** 31 - 3]))
    block = int(rs.choice([64, 128, 256]))
    pos, nblocks, clashes = models.padded_slot_layout(rowptr, col, slot=block)
    if clashes:
        pytest.skip("no edge-free layout in blocks of %d for this graph" % block)
    if block == 128 and nblocks % 2:
        nblocks += 1
    N = nblocks * block
    rp, cc, vv = models.pad_csr(rowptr, col, val, pos, N)
    lpad = np.full(N, np.inf, dtype=np.float32)
    lpad[pos] = lin
    init = rs.randint(0, 2, size=(R, n)).astype(np.uint8)
    init_dev = np.zeros((R, N), dtype=np.uint8)
    init_dev[:, pos] = init
    o_rand = so.sa_csr_rank1_philox(rp, cc, vv, lpad, c_pair, R, betas, 7 + seed, replica_offset=off)
    o_init = so.sa_csr_rank1_philox(rp, cc, vv, lpad, c_pair, R, betas, 7 + seed, replica_offset=off, init=init_dev)
    with Problem.csr_rank1(rowptr, col, val, lin, c_pair, order="padded", block=block) as p:
        for wide, tw in (((0, 0), (0, 2), (2, 0)) if block > 64 else ((0, 0), (0, 2))):
            p.set_option("k2_split", 1)
            p.set_option("k2_wide", wide)
            p.set_option("k2_tw", tw)
            p.anneal(R, betas, 7 + seed, replica_offset=off)
            name = p.kernel_name()
            # K2w exists with two slots per step at either width and with four at 16 entries per variable; with one slot
            # per step only beside a threshold wavefront (without one, 64-seat layouts run on K2s' one-wavefront form)
            is_wide = wide == 0 and (block == 64 and tw == 0 or block > 64 and (int(deg.max()) <= 16 or block == 128))
            assert ("wide<" in name) == is_wide and ("split<" in name) == (not is_wide), name
            assert ("tw>" in name) == (is_wide and tw == 0), name
            st, en, info = p.fetch()
            assert np.array_equal(st, o_rand[0][:, pos]) and info["accepted"] == int(o_rand[2][1])
            assert np.allclose(en, o_rand[1], rtol=1e-9, atol=1e-9)
            p.anneal(R, betas, 7 + seed, replica_offset=off, initial_states=init)
            st2, en2, info2 = p.fetch()
            assert np.array_equal(st2, o_init[0][:, pos]) and info2["accepted"] == int(o_init[2][1])
            assert np.allclose(en2, o_init[1], rtol=1e-9, atol=1e-9)


def test_potts_padded_layout_of_a_clustered_subgraph():
    """K3 under order="padded": the holes (mi_sa_problem_set_absent) keep label 0, sit in no cluster -- the size penalty
    does not see them -- and are never proposed.  Equal to the oracle on the same padded model with the same positions
    absent: random and given initial labels, K = 4 (fixed-point rounds) and K = 20 (serial moves), the minimum-size
    constraint; labels and fp64 energies come back for the caller's n variables."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, lab = graphs.synthetic_snn(1500, 5, 15, 15, 6, seed=1)
    idx = np.flatnonzero(lab == 0)
    renum = -np.ones(1500, dtype=np.int64)
    renum[idx] = np.arange(len(idx))
    sel = np.isin(eu, idx) & np.isin(ev, idx)
    G = graphs.EdgeListGraph([nodes[i] for i in idx], renum[eu[sel]].astype(np.int32), renum[ev[sel]].astype(np.int32), w[sel])
    for K, min_size in ((4, 0), (20, 0), (4, 30)):
        pm = models.build_dqm_potts(G, K, 0.005)
        n = pm.num_variables
        c_pair = float(np.float32(pm.c_pair))
        pos, nslots, clashes = models.padded_slot_layout(pm.rowptr, pm.col)
        assert clashes == 0 and nslots > (n + 63) // 64
        N = nslots * 64
        rp, cc, vv = models.pad_csr(pm.rowptr, pm.col, f32(pm.val), pos, N)
        absent = np.ones(N, dtype=np.uint8)
        absent[pos] = 0
        betas = np.geomspace(0.05, 60.0, 25)
        init = (np.arange(9 * n).reshape(9, n) % K).astype(np.uint16)           # every cluster well above min_size
        init_dev = np.zeros((9, N), dtype=np.uint16)
        init_dev[:, pos] = init
        o_init = so.potts_csr_philox(rp, cc, vv, c_pair, N, K, 9, betas, 8, lin_offset=pm.lin_offset, init=init_dev,
                                     min_size=min_size, absent=absent)
        with Problem.potts_csr(pm.rowptr, pm.col, f32(pm.val), c_pair, n, K, lin_offset=pm.lin_offset, order="padded",
                               energy_model=(pm.val, pm.c_pair)) as p:
            assert p.n == n and p.n_dev == N
            if min_size:
                p.set_option("min_cluster_size", min_size)
            p.anneal(9, betas, 8, initial_states=init)
            st, en, info = p.fetch()
            assert st.shape == (9, n) and np.array_equal(st, o_init[0][:, pos])
            assert info["accepted"] == int(o_init[2][1]) and info["proposals"] == 9 * 25 * n == int(o_init[2][0])
            assert np.allclose(en, pm.energies(st), rtol=1e-12)
            if min_size:
                assert min(np.bincount(row, minlength=K).min() for row in st) >= min_size
            else:
                o_rand = so.potts_csr_philox(rp, cc, vv, c_pair, N, K, 9, betas, 8, lin_offset=pm.lin_offset,
                                             replica_offset=3, absent=absent)
                p.anneal(9, betas, 8, replica_offset=3)
                st2, en2, info2 = p.fetch()
                assert np.array_equal(st2, o_rand[0][:, pos]) and info2["accepted"] == int(o_rand[2][1])
                assert not o_rand[0][:, np.flatnonzero(absent)].any()
    from scrna_seq_qannealing_clustering_amd import MI355XSampler
    ss = MI355XSampler().sample_dqm(models.build_dqm_potts(G, 4, 0.005), num_reads=64, num_sweeps=200, seed=5)
    assert ss.record.sample.shape[1] == len(idx)


@pytest.mark.parametrize("lb,ub_div", [(20, 2.5), (40, 6.0)])
def test_weighted_pair_term_size_window_model(lb, ub_div):
    """`clustering_bqm_3`'s model (BQM_clustering.py:363-380: cut term + a squared size window with slack bits) in its
    STRUCTURED form: sparse couplings + a uniform pair term with integer weights (1 on the cells, the slack coefficients
    on the slack bits; models.add_size_window_penalty, mi_sa_problem_set_pair_weights).  The slack bits sit in a slot of
    their own that the kernels sweep serially.  K2w beside its threshold wavefront, K2 and K2p all equal the oracle's
    weighted chain on the same padded model (states, accepted counts, fp64 energies); the sampler takes this path."""
    from scrna_seq_qannealing_clustering_amd import graphs, MI355XSampler
    nodes, eu, ev, w, _ = graphs.synthetic_snn(300, 5, 15, 15, 3, seed=5, spread=3.0)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    base = models.build_bqm3_cut_qubo(G, k=8)
    n0 = base.num_variables
    pen = models.add_size_window_penalty(base, lb=lb, ub=n0 / ub_div, lagrange_multiplier=0.05 * float(np.sum(w)) / n0)
    assert pen._dense is None and pen.c_pair != 0.0
    if ub_div == 2.5:
        assert pen.weights is not None and pen.weights.max() > 1               # 100 slack units: 1, 2, 4, ... and a remainder
    n = pen.num_variables
    X = np.random.RandomState(1).randint(0, 2, size=(5, n))
    assert np.allclose(pen.energies(X), np.einsum("ri,ij,rj->r", X, pen.dense_Qs(), X) + pen.offset, rtol=1e-12)
    c_pair = float(np.float32(pen.c_pair))
    betas = models.make_beta_schedule(12, models.default_beta_range(pen))
    with Problem.csr_rank1(pen.rowptr, pen.col, f32(pen.val), f32(pen.lin), c_pair, offset=pen.offset, order="padded",
                           energy_model=(pen.val, pen.lin, pen.c_pair), weights=pen.weights) as p:
        seats, N = p._inv, p.n_dev
        rp, cc, vv = models.pad_csr(pen.rowptr, pen.col, f32(pen.val), seats, N)
        lin = np.full(N, np.inf, dtype=np.float32)
        lin[seats] = f32(pen.lin)
        wdev = np.ones(N, dtype=np.int32)
        if pen.weights is not None:
            wdev[seats] = pen.weights
            assert len(set(seats[pen.weights != 1] // 64)) == 1                  # one slot of their own
        R = 6
        init = np.random.RandomState(3).randint(0, 2, size=(R, n)).astype(np.uint8)
        init_dev = np.zeros((R, N), dtype=np.uint8)
        init_dev[:, seats] = init
        o_rand = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, R, betas, 21, offset=pen.offset, replica_offset=2, weights=wdev)
        o_init = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, R, betas, 21, offset=pen.offset, init=init_dev, weights=wdev)
        for opts, tag in (({}, "k_anneal_csr_rank1_wide<16, 1, tw>"), ({"k2_split": 2}, "k_anneal_csr_rank1<16, ")):
            for k, v in opts.items():
                p.set_option(k, v)
            p.anneal(R, betas, 21, replica_offset=2)
            assert p.kernel_name().startswith(tag), p.kernel_name()
            st, en, info = p.fetch()
            assert np.array_equal(st, o_rand[0][:, seats]) and info["accepted"] == int(o_rand[2][1])
            assert info["proposals"] == R * len(betas) * n and np.allclose(en, pen.energies(st), rtol=1e-12)
            assert np.allclose(en, o_rand[1], rtol=1e-5)                       # (the oracle sums the fp32 coefficients)
            p.anneal(R, betas, 21, initial_states=init)
            st2, en2, info2 = p.fetch()
            assert np.array_equal(st2, o_init[0][:, seats]) and info2["accepted"] == int(o_init[2][1])
        p.set_option("k2_split", 0)
        # many replicas: two per wavefront (the first six compared; their ids are the same)
        o6 = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, R, betas, 21, offset=pen.offset, weights=wdev)
        for tw in (0, 2):
            p.set_option("k2_tw", tw)
            p.anneal(1100, betas, 21)
            assert p.kernel_name() == ("k_anneal_csr_rank1_pair<16, tw>" if tw == 0 else "k_anneal_csr_rank1_pair<16>")
            stp, enp, _ = p.fetch()
            assert np.array_equal(stp[:R], o6[0][:, seats]) and np.allclose(enp, pen.energies(stp), rtol=1e-12)
    ss = MI355XSampler().sample_qubo(pen, num_reads=64, num_sweeps=300, seed=9)
    assert ss.info["kernel"] == "csr_rank1" and list(ss.variables) == list(pen.variables)
    assert np.allclose(ss.record.energy, pen.energies(ss.record.sample.astype(np.uint8)), rtol=1e-12)


def test_mid_size_model_on_the_cell_state_kernels():
    """Models beyond 4608 variables (the kidney graph of the reference has 10 605 cells) keep 4 bytes of LDS per seat on
    the pair / few-replica kernels, so the library takes those only when the workgroups of a run are resident in one
    round: 500 reads -> K2w with its threshold wavefront, 1100 reads -> K2p, 4096 reads -> K2 (bit state).  All equal to
    the oracle on the padded model (states, accepted counts of the replicas compared)."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, _ = graphs.synthetic_snn(6000, 5, 15, 15, 9, seed=7, spread=3.0)
    m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
    n = m.num_variables
    c_pair = float(np.float32(m.c_pair))
    pos, nslots, clashes = models.padded_slot_layout(m.rowptr, m.col)
    assert clashes == 0
    N = nslots * 64
    rp, cc, vv = models.pad_csr(m.rowptr, m.col, f32(m.val), pos, N)
    lin = np.full(N, np.inf, dtype=np.float32)
    lin[pos] = f32(m.lin)
    betas = np.geomspace(2e-3, 20.0, 6)
    o = so.sa_csr_rank1_philox(rp, cc, vv, lin, c_pair, 6, betas, 11, replica_offset=0)
    with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), c_pair, order="padded") as p:
        for R, tag in ((500, "k_anneal_csr_rank1_wide<16, 1, tw>"), (1100, "k_anneal_csr_rank1_pair<16, tw>"), (4096, "k_anneal_csr_rank1<16, ")):
            p.anneal(R, betas, 11)
            assert p.kernel_name().startswith(tag), (R, p.kernel_name())
            st, en, info = p.fetch()
            assert np.array_equal(st[:6], o[0][:, pos]) and np.allclose(en[:6], o[1], rtol=1e-9)
            assert info["proposals"] == R * len(betas) * n
            if R == 500:
                acc500 = info["accepted"]
                p.set_option("k2_split", 2)                       # the same run on K2: the same accepted count
                p.anneal(R, betas, 11)
                assert p.kernel_name().startswith("k_anneal_csr_rank1<16, ") and p.fetch()[2]["accepted"] == acc500
                p.set_option("k2_split", 0)


@pytest.mark.parametrize("K,wide", [(2, False), (3, False), (4, True), (5, False), (8, False), (8, True), (9, False), (15, False), (16, True)])
def test_potts_fast_kernel(K, wide):
    """K3f (csrc/potts_fast_kernels.hip): the lean Potts kernel for models whose every slot is free of internal edges --
    the field difference as ONE table lookup and one fma per neighbour (K <= 8: a byte table through v_perm_b32, fp16
    +-2.0 straight into v_fma_mix_f32; K <= 16: 2-bit fields through v_bfe_i32).  Equal to the oracle on the same padded
    model (labels, accepted counts, fp64 energies) and to k_anneal_potts on the same handle: random and given initial
    labels, a replica offset, holes, a run continued in two pieces, one constant temperature per replica, a minimum
    cluster size; 16 and 32 adjacency entries per variable."""
    from scrna_seq_qannealing_clustering_amd import graphs
    if wide:
        nodes, eu, ev, w, _ = graphs.synthetic_snn(700, 8, 15, 30, 5, seed=3, spread=2.5)          # degree cap 30: the 32-wide layout
    else:
        nodes, eu, ev, w, _ = graphs.synthetic_snn(900, 5, 15, 15, 6, seed=2, spread=3.0)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    pm = models.build_dqm_potts(G, K, 0.005)
    n = pm.num_variables
    c_pair = float(np.float32(pm.c_pair))
    pos, nslots, clashes = models.padded_slot_layout(pm.rowptr, pm.col)
    assert clashes == 0
    N = nslots * 64
    rp, cc, vv = models.pad_csr(pm.rowptr, pm.col, f32(pm.val), pos, N)
    absent = np.ones(N, dtype=np.uint8)
    absent[pos] = 0
    R = 6
    betas = np.geomspace(0.05, 60.0, 21)
    init = np.random.RandomState(K).randint(0, K, size=(R, n)).astype(np.uint16)
    init_dev = np.zeros((R, N), dtype=np.uint16)
    init_dev[:, pos] = init
    o_rand = so.potts_csr_philox(rp, cc, vv, c_pair, N, K, R, betas, 8, lin_offset=pm.lin_offset, replica_offset=3, absent=absent)
    o_init = so.potts_csr_philox(rp, cc, vv, c_pair, N, K, R, betas, 8, lin_offset=pm.lin_offset, init=init_dev, absent=absent)
    with Problem.potts_csr(pm.rowptr, pm.col, f32(pm.val), c_pair, n, K, lin_offset=pm.lin_offset, order="padded",
                           energy_model=(pm.val, pm.c_pair)) as p:
        assert p.n_dev == N
        for fast, tw in ((0, 0), (0, 2), (2, 0)):          # K3f beside its threshold wavefront (few replicas), K3f alone, K3
            p.set_option("k3_fast", fast)
            p.set_option("k2_tw", tw)
            p.anneal(R, betas, 8, replica_offset=3)
            assert p.kernel_name() == ("k_anneal_potts_fast<%d, %d%s>" % (32 if wide else 16, 8 if K <= 8 else 16, ", tw" if tw == 0 else "")
                                       if fast == 0 else "k_anneal_potts<%d>" % (32 if wide else 16)), p.kernel_name()
            st, en, info = p.fetch()
            assert np.array_equal(st, o_rand[0][:, pos]) and info["accepted"] == int(o_rand[2][1])
            assert info["proposals"] == R * len(betas) * n and np.allclose(en, pm.energies(st), rtol=1e-12)
            p.anneal(R, betas, 8, initial_states=init)
            st2, en2, info2 = p.fetch()
            assert np.array_equal(st2, o_init[0][:, pos]) and info2["accepted"] == int(o_init[2][1])
            assert np.allclose(en2, pm.energies(st2), rtol=1e-12)
            # a run continued in two pieces == the run in one; then one constant temperature per replica
            p.anneal(R, betas[:8], 8, replica_offset=3)
            p.anneal(R, betas[8:], 8, replica_offset=3, continue_run=True, sweep_offset=8)
            st3, en3, _ = p.fetch()
            assert np.array_equal(st3, st) and np.array_equal(en3, en)
            per = np.geomspace(0.05, 30.0, R)
            p.anneal(R, per, 9, num_sweeps=5, sweep_offset=100)
            st4, _, info4 = p.fetch()
            o_per = so.potts_csr_philox(rp, cc, vv, c_pair, N, K, R, per, 9, lin_offset=pm.lin_offset, sweep_offset=100,
                                        num_sweeps=5, absent=absent)
            assert np.array_equal(st4, o_per[0][:, pos]) and info4["accepted"] == int(o_per[2][1])
        # a minimum cluster size (CQM_clustering.py:46-48 as a hard constraint): the same two kernels against the oracle
        ms = max(2, n // (3 * K))
        init_ms = (np.arange(R * n).reshape(R, n) % K).astype(np.uint16)          # every cluster well above the minimum
        init_ms_dev = np.zeros((R, N), dtype=np.uint16)
        init_ms_dev[:, pos] = init_ms
        o_ms = so.potts_csr_philox(rp, cc, vv, c_pair, N, K, R, betas, 8, lin_offset=pm.lin_offset, init=init_ms_dev,
                                   min_size=ms, absent=absent)
        p.set_option("min_cluster_size", ms)
        for fast, tw in ((0, 0), (0, 2), (2, 0)):
            p.set_option("k3_fast", fast)
            p.set_option("k2_tw", tw)
            p.anneal(R, betas, 8, initial_states=init_ms)
            assert p.kernel_name().startswith("k_anneal_potts_fast<" if fast == 0 else "k_anneal_potts<")
            st5, _, info5 = p.fetch()
            assert np.array_equal(st5, o_ms[0][:, pos]) and info5["accepted"] == int(o_ms[2][1])
            assert min(np.bincount(row, minlength=K).min() for row in st5) >= ms


def test_csr_rank1_two_replicas_per_wavefront():
    """K2p (csrc/sparse_pair_kernels.hip): two replicas share one wavefront and one set of adjacency registers.  Same
    chain as K2: equal to the oracle on the renumbered model and to the one-replica kernel, for an odd replica count
    (one idle seat), given initial states, a run cut in two (MI_F_CONTINUE + sweep_offset), one constant temperature
    per replica, a replica offset, the fp64 energy model; the wider (D = 32) adjacency as well."""
    from scrna_seq_qannealing_clustering_amd import graphs
    for (n, k, ordv, ncl, seed) in ((900, 5, 15, 5, 4), (2000, 8, 30, 6, 6)):
        nodes, eu, ev, w, _ = graphs.synthetic_snn(n, k, 15, ordv, ncl, seed=seed, spread=2.5)
        m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
        c_pair = float(np.float32(m.c_pair))
        betas = np.geomspace(2e-3, 40.0, 30)
        init = np.random.RandomState(1).randint(0, 2, size=(7, n)).astype(np.uint8)
        with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), c_pair, order="slots",
                               energy_model=(m.val, m.lin, m.c_pair)) as p:
            perm = p.perm
            rp, cc, vv = models.permute_csr(m.rowptr, m.col, f32(m.val), perm)
            o_args = (rp, cc, vv, f32(m.lin)[perm], c_pair)
            runs = {}
            for mode in (1, 2):                                        # 1 = pair kernel, 2 = one replica per wavefront
                p.set_option("k2_pair", mode)
                p.anneal(7, betas, 8, replica_offset=5)
                a = p.fetch()
                assert ("pair" in p.kernel_name()) == (mode == 1)          # the model is eligible: the option decides
                p.anneal(7, betas, 8, initial_states=init)
                b = p.fetch()
                p.anneal(7, betas[:11], 8, initial_states=init)
                p.anneal(7, betas[11:], 8, sweep_offset=11, continue_run=True)
                c = p.fetch()
                p.anneal(7, np.geomspace(0.01, 20.0, 7), 8, num_sweeps=9)
                d = p.fetch()
                runs[mode] = (a, b, c, d)
                best = p.best()
                assert best[1] == pytest.approx(d[1].min(), rel=1e-6)
            for x, y in zip(runs[1], runs[2]):
                assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2]["accepted"] == y[2]["accepted"]
            a, b, c, d = runs[1]
            assert np.array_equal(b[0], c[0]) and np.array_equal(b[1], c[1])
        ost, _, ostats = so.sa_csr_rank1_philox(*o_args, 7, betas, 8, replica_offset=5)
        assert np.array_equal(a[0][:, perm], ost) and a[2]["accepted"] == int(ostats[1])
        assert np.allclose(a[1], m.energies(a[0]), rtol=1e-12, atol=1e-9)
        ost, _, ostats = so.sa_csr_rank1_philox(*o_args, 7, betas, 8, init=np.ascontiguousarray(init[:, perm]))
        assert np.array_equal(b[0][:, perm], ost) and b[2]["accepted"] == int(ostats[1])
        ost, _, ostats = so.sa_csr_rank1_philox(*o_args, 7, np.geomspace(0.01, 20.0, 7), 8, num_sweeps=9)
        assert np.array_equal(d[0][:, perm], ost) and d[2]["accepted"] == int(ostats[1])


def test_full_size_properties_config2_csr_headline():
    """BASELINE config 2 exactly as bench.py times it: bench.build_workload()'s model (n = 2638, one connected
    component), Problem.csr_rank1(order="slots"), 4096 replicas x 1000 sweeps, seed 1234 -- the two-replicas-per-
    wavefront kernel.  Beyond what the oracle runs in seconds, so: the oracle on three slices of the replicas
    (states, accepted counts, fp64 energies, integer cut counts), the upper half-shard alone == the upper half of the
    full run, the run cut in two == the run, energies == fp64 host evaluation in the caller's model."""
    import bench
    m, Qs, betas, (eu, ev, w), G = bench.build_workload()
    n, R = m.num_variables, bench.REPLICAS_PER_GPU
    assert n == 2638 and len(betas) == 1000
    # the workload is ONE graph component: the optimum has to cut edges
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    ncomp, lab = connected_components(sp.coo_matrix((np.ones(len(eu)), (eu, ev)), shape=(n, n)), directed=False)
    assert np.bincount(lab).max() >= 0.9 * n
    c_pair = float(np.float32(m.c_pair))
    with Problem.csr_rank1(m.rowptr, m.col, f32(m.val), f32(m.lin), c_pair, order="slots",
                           energy_model=(m.val, m.lin, m.c_pair)) as p:
        perm = p.perm
        p.anneal(R, betas, bench.SEED)
        st, en, info = p.fetch()
        idx, e_best, key, s_best = p.best()
        p.anneal(R // 2, betas, bench.SEED, replica_offset=R // 2)
        st_hi, en_hi, _ = p.fetch()
        p.anneal(R, betas[:400], bench.SEED)
        p.anneal(R, betas[400:], bench.SEED, sweep_offset=400, continue_run=True)
        st_2, en_2, info_2 = p.fetch()
    assert st.shape == (R, n) and info["proposals"] == R * 1000 * n
    assert np.array_equal(st[R // 2:], st_hi) and np.array_equal(en[R // 2:], en_hi)
    assert np.array_equal(st, st_2) and np.array_equal(en, en_2)
    assert np.allclose(en, m.energies(st), rtol=1e-12, atol=1e-8)
    assert en[idx] == en.min() and np.array_equal(s_best, st[idx])
    cut = so.cut_edges(eu, ev, st)
    assert int(cut[idx]) > 0                                           # a connected graph: the best state cuts edges
    rp, cc, vv = models.permute_csr(m.rowptr, m.col, f32(m.val), perm)
    acc = 0
    for lo in (0, 2046, 4092):
        ost, oen, ostats = so.sa_csr_rank1_philox(rp, cc, vv, f32(m.lin)[perm], c_pair, 4, betas, bench.SEED,
                                                  replica_offset=lo)
        back = ost[:, np.argsort(perm)]
        assert np.array_equal(st[lo:lo + 4], back)
        assert np.array_equal(cut[lo:lo + 4], so.cut_edges(eu, ev, back))
        assert np.allclose(en[lo:lo + 4], m.energies(back), rtol=1e-12, atol=1e-8)
        assert np.allclose(oen, en[lo:lo + 4], rtol=1e-5)              # the oracle's energies are the fp32 model's
        acc += int(ostats[1])
    assert 0 < acc < 12 * 1000 * n and 0 < info["accepted"] < info["proposals"]


def test_potts_slot_independent_order():
    """K3 under order="slots": equal to the oracle on the renumbered model, labels returned in the caller's order."""
    from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
    fx = load_fixture("varied")
    pm = models.build_dqm_potts(fx.graph(), 5, 0.005)
    perm = models.slot_independent_order(pm.rowptr, pm.col)
    rp, cc, vv = models.permute_csr(pm.rowptr, pm.col, pm.val, perm)
    betas = models.make_beta_schedule(30, default_potts_beta_range(pm))
    init = np.random.RandomState(3).randint(0, 5, size=(6, 256)).astype(np.uint16)
    olab, oen, ostats = so.potts_csr_philox(rp, cc, f32(vv), float(np.float32(pm.c_pair)), 256, 5, 6, betas, 9,
                                            lin_offset=pm.lin_offset, init=np.ascontiguousarray(init[:, perm]))
    with Problem.potts_csr(pm.rowptr, pm.col, f32(pm.val), float(np.float32(pm.c_pair)), 256, 5,
                           lin_offset=pm.lin_offset, order="slots") as p:
        p.anneal(6, betas, 9, initial_states=init)
        lab, en, info = p.fetch()
    assert np.array_equal(lab[:, perm], olab) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)


def test_csr_rank1_hot_schedule_and_launch_shapes():
    """Many flips per slot (a schedule that starts at ~100 % acceptance) and in-slot neighbour updates, for
    every workgroup shape: the chain does not depend on how replicas are packed into workgroups."""
    fx = load_fixture("blobs")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    args = (m.rowptr, m.col, f32(m.val), f32(m.lin), float(np.float32(m.c_pair)))
    betas = np.geomspace(1e-4, 1.0, 12)
    ost, oen, ostats = so.sa_csr_rank1_philox(*args, 70, betas, 3)
    for waves in (0, 1, 3):
        with Problem.csr_rank1(*args) as p:
            p.set_option("k2_waves", waves)
            p.anneal(70, betas, 3)
            st, en, info = p.fetch()
        assert np.array_equal(st, ost) and info["accepted"] == int(ostats[1])
        assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,K", [("noisy_circles", 3), ("blobs", 3), ("aniso", 8), ("no_structure", 15),
                                    ("noisy_moons", 2), ("varied", 33), ("noisy_circles", 64)])
def test_potts_trajectory_parity(name, K):
    fx = load_fixture(name)
    pm = models.build_dqm_potts(fx.graph(), K, 0.005)
    from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
    betas = models.make_beta_schedule(40, default_potts_beta_range(pm))
    args = (pm.rowptr, pm.col, f32(pm.val), float(np.float32(pm.c_pair)), 256, K)
    init = np.random.RandomState(4).randint(0, K, size=(7, 256)).astype(np.uint16)
    for kw in (dict(), dict(init=init)):
        olab, oen, ostats = so.potts_csr_philox(*args, 7, betas, 77, lin_offset=pm.lin_offset,
                                                replica_offset=11, **kw)
        with Problem.potts_csr(*args, lin_offset=pm.lin_offset) as p:
            p.anneal(7, betas, 77, replica_offset=11, initial_states=kw.get("init"))
            lab, en, info = p.fetch()
        assert lab.dtype == np.uint16 and np.array_equal(lab, olab)
        assert info["accepted"] == int(ostats[1])
        assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)
        assert np.allclose(en, pm.energies(lab), rtol=1e-5)
        assert np.array_equal(so.cut_edges(fx.eu, fx.ev, lab), so.cut_edges(fx.eu, fx.ev, olab))


def test_sample_dqm_like_clustering_dqm(kat):
    """DQM_clustering.py:29-47 on the reference's circles graph: K = 3, gamma = 0.005."""
    fx = load_fixture("noisy_circles")
    pm = models.build_dqm_potts(fx.graph(), 3, 0.005)
    sampler = MI355XSampler()
    ss = sampler.sample_dqm(pm, label="DQM - scRAN-seq", num_reads=64, num_sweeps=500, seed=5)
    assert ss.vartype == "DISCRETE"
    lut = ss.first.sample
    assert list(lut.keys()) == fx.nodes                      # variable order = G.nodes order
    lab = np.array(list(lut.values()))
    # literal reference model evaluates the returned labels to the returned energy
    lin, quad = mo.dqm_model(fx.nodes, fx.edges, 3, 0.005)
    assert ss.first.energy == pytest.approx(mo.dqm_energy(lin, quad, dict(lut)), rel=1e-10)
    # integer edge cut of the best labelling, recomputed two ways
    assert int(so.cut_edges(fx.eu, fx.ev, lab[None, :].astype(np.uint16))[0]) == mo.cut_edges(fx.edges, dict(lut))
    # P3: the same best energy as the CPU oracle's chain at equal (reads, sweeps, schedule, seed), and far
    # below a random labelling (the known labels-=-components value is kat["dqm_circles"]; single-site
    # Potts moves at 500 sweeps do not always merge the domains, on the oracle either)
    # (the sampler sweeps in its slot-independent order: the oracle runs on the same renumbered model)
    betas = models.make_beta_schedule(500, ss.info["beta_range"])
    perm = models.slot_independent_order(pm.rowptr, pm.col)
    rp, cc, vv = models.permute_csr(pm.rowptr, pm.col, pm.val, perm)
    olab, oen, _ = so.potts_csr_philox(rp, cc, f32(vv), float(np.float32(pm.c_pair)), 256, 3,
                                       64, betas, 5, lin_offset=pm.lin_offset)
    unperm = np.empty_like(olab)
    unperm[:, perm] = olab
    assert pm.energies(unperm).min() == pytest.approx(ss.first.energy, rel=1e-12)
    rnd = np.random.RandomState(0).randint(0, 3, size=(16, 256))
    assert ss.first.energy < pm.energies(rnd).min() - 100.0
    assert ss.first.energy < kat["dqm_circles"]["E_pairwise"] + 60.0
    assert ss.info["kernel"] == "potts_csr"


def test_dqm_lookalike_through_sampler():
    fx = load_fixture("noisy_moons")
    keep = fx.nodes[:40]
    ks = set(keep)
    edges = [(u, v, w) for u, v, w in fx.edges if u in ks and v in ks]
    from itertools import combinations
    K, gamma = 3, 0.05
    dqm = DiscreteQuadraticModel()
    for node in keep:
        dqm.add_variable(K, label=node)
    for node in keep:
        dqm.set_linear(node, [gamma * (1 - len(keep) / K)] * K)
    for i, j in combinations(keep, 2):
        dqm.set_quadratic(i, j, {(c, c): 2 * gamma for c in range(K)})
    for u, v, w in edges:
        dqm.set_quadratic(u, v, {(c, c): -2 * w for c in range(K)})
        dqm.set_linear(u, [w] * K)
        dqm.set_linear(v, [w] * K)
    ss = MI355XSampler().sample_dqm(dqm, num_reads=32, num_sweeps=300, seed=1)
    assert ss.first.energy == pytest.approx(dqm.energy(dict(ss.first.sample)), rel=1e-10)
    assert len(set(ss.first.sample.values())) >= 1


def test_structured_error_behaviour():
    rowptr = np.array([0, 1, 2], dtype=np.int32)
    col = np.array([1, 0], dtype=np.int32)
    val = np.array([1.0, 1.0], dtype=np.float32)
    with pytest.raises(_lib.MiSaError) as ei:
        Problem.potts_csr(rowptr, col, val, 0.0, 2, 65)           # K too large
    assert ei.value.code == -5
    with pytest.raises(_lib.MiSaError):
        Problem.potts_csr(rowptr, np.array([5, 0], dtype=np.int32), val, 0.0, 2, 3)   # bad column
    p = Problem.potts_csr(rowptr, col, val, 0.0, 2, 3)
    with pytest.raises(_lib.MiSaError):
        p.anneal(1, [1.0], 1, initial_states=np.array([[0, 7]], dtype=np.uint16))      # label >= K
    p.close()
    # rows wider than 64 run on the runtime-width kernels (test_rows_wider_than_64); self-loops are refused
    with pytest.raises(_lib.MiSaError):
        Problem.csr_rank1(np.array([0, 1, 2], dtype=np.int32), np.array([0, 0], dtype=np.int32), val,
                          np.zeros(2, dtype=np.float32), 0.0)


def _csr_from_edges(n, edges, w):
    """symmetric CSR (rows ascending by column) from an undirected edge list"""
    rows = [[] for _ in range(n)]
    for (u, v), x in zip(edges, w):
        rows[u].append((v, x))
        rows[v].append((u, x))
    rowptr = np.zeros(n + 1, dtype=np.int32)
    col, val = [], []
    for i, r in enumerate(rows):
        r.sort()
        rowptr[i + 1] = rowptr[i] + len(r)
        col += [c for c, _ in r]
        val += [x for _, x in r]
    return rowptr, np.asarray(col, dtype=np.int32), np.asarray(val, dtype=np.float32)


def _edge_shapes():
    rs = np.random.RandomState(12)
    ring = lambda n: [(i, (i + 1) % n) for i in range(n)] if n > 2 else []
    star = lambda n, hub, k: [(hub, j) for j in range(n) if j != hub][:k]
    dense64 = [(i, j) for i in range(64) for j in range(i + 1, 64)]           # one full slot, degree 63
    return {
        "single_variable": (1, []),
        "two_isolated": (2, []),
        "one_full_slot_complete_graph": (64, dense64),
        "slot_plus_one_no_edges": (65, []),
        "ring_129": (129, ring(129)),
        "hub_of_degree_64_at_a_slot_boundary": (200, star(200, 64, 64) + ring(200)[:50]),
        "isolated_tail": (300, [(int(a), int(b)) for a, b in rs.randint(0, 100, size=(150, 2)) if a != b]),
    }


@pytest.mark.parametrize("shape", list(_edge_shapes()))
@pytest.mark.parametrize("order", [None, "slots"])
def test_structured_kernels_on_degenerate_graphs(shape, order):
    """Empty / single-variable / one-slot / maximum-degree / isolated-node graphs through K2 and K3, natural
    and slot-independent order, against the oracle (run on the same renumbered model)."""
    n, edges = _edge_shapes()[shape]
    edges = sorted(set((min(e), max(e)) for e in edges))
    w = (np.random.RandomState(3).randint(1, 6, size=len(edges)) / 8.0).astype(np.float32)
    rowptr, col, val = _csr_from_edges(n, edges, w)
    lin = (np.random.RandomState(4).randint(-4, 5, size=n) / 4.0).astype(np.float32)
    betas = np.geomspace(0.2, 8.0, 12)
    R = 5
    with Problem.csr_rank1(rowptr, col, -2.0 * val, lin, 0.25, order=order) as p:
        p.anneal(R, betas, 21)
        st, en, info = p.fetch()
        perm = p.perm
    if perm is None:
        o_args = (rowptr, col, -2.0 * val, lin, 0.25)
        back = slice(None)
    else:
        rp2, c2, v2 = models.permute_csr(rowptr, col, -2.0 * val, perm)
        o_args = (rp2, c2, v2, lin[perm], 0.25)
        back = np.argsort(perm)
    ost, oen, ostats = so.sa_csr_rank1_philox(*o_args, R, betas, 21)
    assert st.shape == (R, n) and np.array_equal(st, ost[:, back])
    assert info["accepted"] == int(ostats[1]) and np.allclose(en, oen, rtol=1e-9, atol=1e-9)

    for K in (1, 2, 5):
        with Problem.potts_csr(rowptr, col, -2.0 * val, 0.01, n, K, order=order) as p:
            p.anneal(R, betas, 22)
            lab, en, info = p.fetch()
            perm = p.perm
        if perm is None:
            o_args, back = (rowptr, col, -2.0 * val, 0.01, n, K), slice(None)
        else:
            rp2, c2, v2 = models.permute_csr(rowptr, col, -2.0 * val, perm)
            o_args, back = (rp2, c2, v2, 0.01, n, K), np.argsort(perm)
        olab, oen, ostats = so.potts_csr_philox(*o_args, R, betas, 22)
        assert lab.shape == (R, n) and np.array_equal(lab, olab[:, back])
        assert info["accepted"] == int(ostats[1]) and np.allclose(en, oen, rtol=1e-9, atol=1e-9)


def test_reported_energies_in_the_callers_fp64_model():
    """mi_sa_problem_set_energy_model_f64: the chain runs on the fp32 model, the energies come back evaluated
    on the device in the fp64 coefficients -- equal to the host's fp64 evaluation of the same states to
    rounding (1e-12 relative), in both orders; the states themselves do not change."""
    fx = load_fixture("aniso")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    betas = models.make_beta_schedule(60, models.default_beta_range(m))
    f32_args = (m.rowptr, m.col, f32(m.val), f32(m.lin), float(np.float32(m.c_pair)))
    for order in (None, "slots"):
        with Problem.csr_rank1(*f32_args, offset=m.offset, order=order) as p:
            p.anneal(9, betas, 5)
            st0, en0, _ = p.fetch()
        with Problem.csr_rank1(*f32_args, offset=m.offset, order=order,
                               energy_model=(m.val, m.lin, m.c_pair)) as p:
            p.anneal(9, betas, 5)
            st, en, _ = p.fetch()
            best = p.best()
        assert np.array_equal(st, st0)
        want = m.energies(st)
        assert np.allclose(en, want, rtol=1e-12, atol=1e-9)
        assert not np.array_equal(en, en0) and np.allclose(en, en0, rtol=1e-5)      # fp32-model energies differ
        assert best[1] == pytest.approx(want.min(), rel=1e-12)
    pm = models.build_dqm_potts(fx.graph(), 4, 0.005)
    pb = models.make_beta_schedule(40, (0.5, 30.0))
    for order in (None, "slots"):
        with Problem.potts_csr(pm.rowptr, pm.col, f32(pm.val), float(np.float32(pm.c_pair)), 256, 4,
                               lin_offset=pm.lin_offset, order=order, energy_model=(pm.val, pm.c_pair)) as p:
            p.anneal(6, pb, 8)
            lab, en, _ = p.fetch()
        assert np.allclose(en, pm.energies(lab), rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        Problem.csr_rank1(*f32_args, energy_model=(m.val[:-1], m.lin, m.c_pair))


def _potts_host_energy(rowptr, col, val64, c_pair, lin_offset, lab, K):
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    e = 0.5 * ((lab[:, rows] == lab[:, col]) * val64[None, :]).sum(axis=1)
    for r in range(lab.shape[0]):
        cnt = np.bincount(lab[r], minlength=K).astype(np.float64)
        e[r] += c_pair * 0.5 * float(np.sum(cnt * (cnt - 1.0)))
    return e + lin_offset


def test_full_size_properties_config3_potts_k8():
    """BASELINE config 3 at full size (n = 2638, K = 8, 4096 replicas) -- beyond what the oracle runs in
    seconds, so checked through size-independent properties: the oracle on a slice of the replicas
    (any replica's chain depends only on its global id), two half-shards == one run, a run cut into two
    calls == one run, every label < K, reported energies == fp64 host evaluation of the returned labels,
    cluster sizes sum to n."""
    from scrna_seq_qannealing_clustering_amd import graphs
    from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
    nodes, eu, ev, w, _ = graphs.synthetic_snn(2638, 5, 15, 15, 9, seed=0)
    pm = models.build_dqm_potts(graphs.EdgeListGraph(nodes, eu, ev, w), 8, 0.005)
    n, K, R = 2638, 8, 4096
    betas = models.make_beta_schedule(24, default_potts_beta_range(pm))
    args = (pm.rowptr, pm.col, f32(pm.val), float(np.float32(pm.c_pair)), n, K)
    with Problem.potts_csr(*args, lin_offset=pm.lin_offset, order="slots", energy_model=(pm.val, pm.c_pair)) as p:
        p.anneal(R, betas, 1234)
        lab, en, info = p.fetch()
        perm = p.perm
        p.anneal(R // 2, betas, 1234, replica_offset=R // 2)               # the upper half-shard alone
        lab_hi, en_hi, _ = p.fetch()
        p.anneal(R, betas[:10], 1234)                                      # the same run in two pieces
        p.anneal(R, betas[10:], 1234, sweep_offset=10, continue_run=True)
        lab_2, en_2, info_2 = p.fetch()
    assert lab.shape == (R, n) and int(lab.max()) < K
    assert np.array_equal(lab[R // 2:], lab_hi) and np.array_equal(en[R // 2:], en_hi)
    assert np.array_equal(lab, lab_2) and np.array_equal(en, en_2)
    assert info["proposals"] == R * len(betas) * n and 0 < info["accepted"] < info["proposals"]
    pick = np.r_[0:8, R // 2 - 4:R // 2 + 4, R - 8:R]
    want = _potts_host_energy(pm.rowptr, pm.col, pm.val, pm.c_pair, pm.lin_offset, lab[pick].astype(np.int64), K)
    assert np.allclose(en[pick], want, rtol=1e-12, atol=1e-9)
    # oracle on replicas 2044..2051 of the renumbered model
    rp2, c2, v2 = models.permute_csr(pm.rowptr, pm.col, f32(pm.val), perm)
    olab, oen, _ = so.potts_csr_philox(rp2, c2, v2, args[3], n, K, 8, betas, 1234, lin_offset=pm.lin_offset,
                                       replica_offset=2044)
    assert np.array_equal(lab[2044:2052], olab[:, np.argsort(perm)])


def test_full_size_properties_config4_50k_cells():
    """BASELINE config 4 at full size (n = 50 000 cells, 1024 replicas = one GPU's share of 8192) on the CSR
    kernel, slot-independent order: oracle on two of the replicas, shard invariance, reported energies ==
    fp64 host evaluation."""
    n, R = 50000, 1024
    rowptr, col, val, lin, c_pair = _sparse_ring_model(n, seed=7)
    betas = np.geomspace(0.002, 0.5, 4)
    val64, lin64 = val.astype(np.float64), lin.astype(np.float64)
    with Problem.csr_rank1(rowptr, col, val, lin, c_pair, order="slots", energy_model=(val64, lin64, c_pair)) as p:
        p.anneal(R, betas, 99)
        st, en, info = p.fetch()
        perm = p.perm
        p.anneal(16, betas, 99, replica_offset=1000)
        st_hi, en_hi, _ = p.fetch()
    assert st.shape == (R, n) and info["proposals"] == R * len(betas) * n
    assert np.array_equal(st[1000:1016], st_hi) and np.array_equal(en[1000:1016], en_hi)
    pick = np.array([0, 511, 1023])
    X = st[pick].astype(np.float64)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    s = X.sum(axis=1)
    want = X @ lin64 + c_pair * 0.5 * s * (s - 1.0) + 0.5 * ((X[:, rows] * X[:, col]) * val64[None, :]).sum(axis=1)
    assert np.allclose(en[pick], want, rtol=1e-12, atol=1e-6)
    if perm is None:
        o_args, back = (rowptr, col, val, lin, c_pair), slice(None)
    else:
        rp2, c2, v2 = models.permute_csr(rowptr, col, val, perm)
        o_args, back = (rp2, c2, v2, lin[perm], c_pair), np.argsort(perm)
    ost, _, _ = so.sa_csr_rank1_philox(*o_args, 2, betas, 99, replica_offset=700)
    assert np.array_equal(st[700:702], ost[:, back])


def test_full_size_properties_config5_kidney_k15():
    """BASELINE config 5's model size (n = 10 605 cells, K = 15): one rung per replica (the tempering launch
    shape), oracle on one replica, labels < K, energies == fp64 host evaluation."""
    n, K, R = 10605, 15, 128
    rowptr, col, val, _, _ = _sparse_ring_model(n, seed=11)
    val = f32(val / 8.0)                                                  # -2 w, the DQM's edge bias
    c_pair = float(np.float32(0.01))
    betas = np.geomspace(0.5, 30.0, R)                                    # one constant beta per replica
    with Problem.potts_csr(rowptr, col, val, c_pair, n, K, order="slots",
                           energy_model=(val.astype(np.float64), c_pair)) as p:
        p.anneal(R, betas, 31, num_sweeps=5)
        lab, en, info = p.fetch()
        perm = p.perm
    assert lab.shape == (R, n) and int(lab.max()) < K and info["proposals"] == R * 5 * n
    pick = np.array([0, 64, 127])
    want = _potts_host_energy(rowptr, col, val.astype(np.float64), c_pair, 0.0, lab[pick].astype(np.int64), K)
    assert np.allclose(en[pick], want, rtol=1e-12, atol=1e-9)
    rp2, c2, v2 = (rowptr, col, val) if perm is None else models.permute_csr(rowptr, col, val, perm)
    back = slice(None) if perm is None else np.argsort(perm)
    olab, _, _ = so.potts_csr_philox(rp2, c2, v2, c_pair, n, K, 1, np.full(5, betas[77]), 31, replica_offset=77)
    assert np.array_equal(lab[77:78], olab[:, back])


def test_full_size_config5_tempering_rounds_and_exchanges():
    """BASELINE config 5 with its defining feature at full size: n = 10 605, K = 15, a ladder of 8 temperatures x 16
    chains, 4 rounds x 5 sweeps through tempering.parallel_tempering on the GPU (anneal rounds at the temperatures
    resident in HBM + the exchange kernel K6).  Every exchange == oracle/pt_oracle.py applied to the energies the round
    left; one replica that changed rungs == the oracle's Potts chain replayed round by round with that replica's rung
    history (initial labels from its own stream, then continued: states + sweep offset); the run without the per-round
    energy reads is the same run."""
    from oracle import pt_oracle
    from scrna_seq_qannealing_clustering_amd import tempering
    n, K, T, chains, rounds, sw, seed = 10605, 15, 8, 16, 4, 5, 31
    R = T * chains
    rowptr, col, val, _, _ = _sparse_ring_model(n, seed=11)
    val = f32(val / 8.0)
    c_pair = float(np.float32(0.01))
    ladder = tempering.geometric_ladder(0.5, 30.0, T)

    class Recording(tempering.ProblemEngine):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.en_hist, self.rung_hist = [], []

        def exchange(self, rnd, seed_, all_energies=None):
            self.en_hist.append(self.energies())
            super().exchange(rnd, seed_, all_energies)
            self.rung_hist.append(self.rungs()[0])

    with Problem.potts_csr(rowptr, col, val, c_pair, n, K, order="slots",
                           energy_model=(val.astype(np.float64), c_pair)) as p:
        eng = Recording(p, seed=seed)
        out = tempering.parallel_tempering(eng, ladder, chains, rounds, sw, seed)
        perm = p.perm
        quiet = tempering.parallel_tempering(tempering.ProblemEngine(p, seed=seed), ladder, chains, rounds, sw, seed,
                                             history=False)
    assert len(eng.en_hist) == rounds - 1 and out["local_states"].shape == (R, n) and int(out["local_states"].max()) < K
    # the exchanges: same decisions as the restatement, from the energies each round left on the device
    rung = np.arange(R, dtype=np.int64) % T
    proposed = accepted = 0
    for rnd in range(rounds - 1):
        rung, pp, aa = pt_oracle.exchange_step(eng.en_hist[rnd], rung, ladder, T, rnd, seed)
        proposed, accepted = proposed + pp, accepted + aa
        assert np.array_equal(eng.rung_hist[rnd], rung)
    assert np.array_equal(out["rung"], rung) and accepted > 0
    assert out["swap_rate"] == pytest.approx(accepted / proposed)
    assert np.array_equal(np.sort(rung.reshape(chains, T), axis=1), np.tile(np.arange(T), (chains, 1)))
    # energies of the final states in the caller's fp64 model
    pick = np.array([0, 77, R - 1])
    want = _potts_host_energy(rowptr, col, val.astype(np.float64), c_pair, 0.0, out["local_states"][pick].astype(np.int64), K)
    assert np.allclose(out["energies"][pick], want, rtol=1e-12, atol=1e-9)
    # one replica whose rung changed, replayed on the CPU: round r at the temperature of the rung it held then
    hist = np.stack([np.arange(R) % T] + eng.rung_hist)                       # [round][replica]
    moved = np.flatnonzero((hist != hist[0]).any(axis=0))
    assert len(moved) > 0
    g = int(moved[len(moved) // 2])
    rp2, c2, v2 = (rowptr, col, val) if perm is None else models.permute_csr(rowptr, col, val, perm)
    back = slice(None) if perm is None else np.argsort(perm)
    lab = None
    for rnd in range(rounds):
        lab, oen, _ = so.potts_csr_philox(rp2, c2, v2, c_pair, n, K, 1, np.full(sw, ladder[hist[rnd][g]]), seed,
                                          replica_offset=g, init=lab, sweep_offset=rnd * sw)
    assert np.array_equal(out["local_states"][g:g + 1], lab[:, back])
    assert np.allclose(out["energies"][g], oen[0], rtol=1e-9)
    # no per-round reads of the energies (nothing leaves HBM between rounds): the same run
    assert np.array_equal(quiet["local_states"], out["local_states"]) and np.array_equal(quiet["rung"], out["rung"])
    assert np.array_equal(quiet["energies"], out["energies"]) and quiet["history"] == []


@pytest.mark.parametrize("seed", range(64))
def test_structured_kernels_random_models(seed):
    """Random sparse models (size, degree, weights, K, schedule, order, replica offset drawn per seed) through K2
    and K3 against the oracle: states / labels, accepted counts and energies."""
    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([3, 17, 63, 64, 65, 130, 257, 700]))
    max_deg = int(rs.choice([1, 3, 9, 16, 17, 33, 60]))
    m = min(n * max_deg // 3, n * (n - 1) // 2)
    pairs = set()
    deg = np.zeros(n, dtype=int)
    for _ in range(4 * m):
        if len(pairs) >= m:
            break
        a, b = (int(x) for x in rs.randint(0, n, 2))
        if a == b or (min(a, b), max(a, b)) in pairs or deg[a] >= max_deg or deg[b] >= max_deg:
            continue
        pairs.add((min(a, b), max(a, b)))
        deg[a] += 1
        deg[b] += 1
    edges = sorted(pairs)
    w = rs.choice(np.array([1 / 9, 0.25, 3 / 7, 2 / 3, 1.0, -0.5]), size=len(edges)).astype(np.float32)
    rowptr, col, val = _csr_from_edges(n, edges, w)
    lin = rs.normal(scale=0.7, size=n).astype(np.float32)
    c_pair = float(np.float32(rs.choice([0.0, 0.03, 0.4, -0.02])))
    sweeps = int(rs.choice([1, 4, 13]))
    betas = np.geomspace(float(rs.choice([0.05, 0.5])), float(rs.choice([2.0, 40.0])), sweeps)
    R = int(rs.choice([1, 2, 7]))
    off = int(rs.choice([0, 5, 2 ** 31 - 3]))
    order = [None, "slots"][seed & 1]
    with Problem.csr_rank1(rowptr, col, val, lin, c_pair, order=order) as p:
        p.anneal(R, betas, 7 + seed, replica_offset=off)
        st, en, info = p.fetch()
        perm = p.perm
        # the two-replicas-per-wavefront kernel wherever the model is eligible for it (else this is K2 again)
        p.set_option("k2_pair", 1)
        p.anneal(R, betas, 7 + seed, replica_offset=off)
        st_p, en_p, info_p = p.fetch()
        assert np.array_equal(st_p, st) and np.array_equal(en_p, en) and info_p["accepted"] == info["accepted"]
    if perm is None:
        o_args, back = (rowptr, col, val, lin, c_pair), slice(None)
    else:
        rp2, c2, v2 = models.permute_csr(rowptr, col, val, perm)
        o_args, back = (rp2, c2, v2, lin[perm], c_pair), np.argsort(perm)
    ost, oen, ostats = so.sa_csr_rank1_philox(*o_args, R, betas, 7 + seed, replica_offset=off)
    assert np.array_equal(st, ost[:, back]) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)

    K = int(rs.choice([2, 3, 8, 15, 16, 17, 32, 33, 64]))
    min_size = int(rs.choice([0, 0, 1, 2])) if n >= 4 * K else 0
    init = None
    if min_size:
        init = (np.arange(n)[None, :] % K).repeat(R, axis=0).astype(np.uint16)       # every cluster >= n // K members
    with Problem.potts_csr(rowptr, col, val, c_pair, n, K, order=order) as p:
        if min_size:
            p.set_option("min_cluster_size", min_size)
        p.anneal(R, betas, 70 + seed, replica_offset=off, initial_states=init)
        lab, en, info = p.fetch()
        perm = p.perm
    if perm is None:
        o_args, back, oinit = (rowptr, col, val, c_pair, n, K), slice(None), init
    else:
        rp2, c2, v2 = models.permute_csr(rowptr, col, val, perm)
        o_args, back, oinit = (rp2, c2, v2, c_pair, n, K), np.argsort(perm), (None if init is None else init[:, perm])
    olab, oen, ostats = so.potts_csr_philox(*o_args, R, betas, 70 + seed, replica_offset=off, init=oinit,
                                            min_size=min_size)
    assert np.array_equal(lab, olab[:, back]) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-9, atol=1e-9)
    if min_size:
        assert all(np.bincount(row, minlength=K).min() >= min_size for row in lab)


@pytest.mark.parametrize("n,max_deg,K", [(200, 100, 3), (700, 300, 8), (130, 129, 15)])
@pytest.mark.parametrize("order", [None, "slots"])
def test_rows_wider_than_64(n, max_deg, K, order):
    """Models whose rows exceed the 64-entry register layout (the reference's UNTRIMMED SNN graphs reach degrees
    of order k^2; (130, 129) is a complete graph): the runtime-width forms of K3 and K2 against the oracle."""
    rs = np.random.RandomState(n + max_deg)
    if max_deg >= n - 1:
        edges = [(i, j) for i in range(n) for j in range(i + 1, n)]
    else:
        pairs, deg = set(), np.zeros(n, dtype=int)
        hubs = rs.choice(n, size=5, replace=False)
        for h in hubs:                                           # a few hubs at the cap, a sparse rest
            for j in rs.permutation(n):
                if deg[h] >= max_deg:
                    break
                j = int(j)
                if j != h and (min(h, j), max(h, j)) not in pairs and deg[j] < max_deg:
                    pairs.add((min(int(h), j), max(int(h), j)))
                    deg[h] += 1
                    deg[j] += 1
        for _ in range(3 * n):
            a, b = (int(x) for x in rs.randint(0, n, 2))
            if a != b and (min(a, b), max(a, b)) not in pairs and deg[a] < max_deg and deg[b] < max_deg:
                pairs.add((min(a, b), max(a, b)))
                deg[a] += 1
                deg[b] += 1
        edges = sorted(pairs)
    w = rs.choice(np.array([1 / 9, 0.25, 3 / 7, 2 / 3, 1.0]), size=len(edges)).astype(np.float32)
    rowptr, col, val = _csr_from_edges(n, edges, -2.0 * w)
    assert int(np.diff(rowptr).max()) > 64
    betas = np.geomspace(0.05, 6.0, 9)
    R = 4
    with Problem.potts_csr(rowptr, col, val, 0.01, n, K, order=order, energy_model=(val.astype(np.float64), 0.01)) as p:
        p.anneal(R, betas, 17, replica_offset=3)
        lab, en, info = p.fetch()
        perm = p.perm
    if perm is None:
        o_args, back = (rowptr, col, val, 0.01, n, K), slice(None)
    else:
        rp2, c2, v2 = models.permute_csr(rowptr, col, val, perm)
        o_args, back = (rp2, c2, v2, 0.01, n, K), np.argsort(perm)
    olab, oen, ostats = so.potts_csr_philox(*o_args, R, betas, 17, replica_offset=3)
    assert np.array_equal(lab, olab[:, back]) and info["accepted"] == int(ostats[1]) > 0
    assert np.allclose(en, _potts_host_energy(rowptr, col, val.astype(np.float64), 0.01, 0.0, lab.astype(np.int64), K),
                       rtol=1e-12, atol=1e-9)
    # the binary kernel on the same wide rows
    lin = rs.normal(scale=0.5, size=n).astype(np.float32)
    with Problem.csr_rank1(rowptr, col, val, lin, 0.02, order=order,
                           energy_model=(val.astype(np.float64), lin.astype(np.float64), 0.02)) as p:
        p.anneal(R, betas, 18, replica_offset=3)
        st, en, info = p.fetch()
        perm = p.perm
    if perm is None:
        o_args, back = (rowptr, col, val, lin, 0.02), slice(None)
    else:
        rp2, c2, v2 = models.permute_csr(rowptr, col, val, perm)
        o_args, back = (rp2, c2, v2, lin[perm], 0.02), np.argsort(perm)
    ost, oen, ostats = so.sa_csr_rank1_philox(*o_args, R, betas, 18, replica_offset=3)
    assert np.array_equal(st, ost[:, back]) and info["accepted"] == int(ostats[1]) > 0
    assert np.allclose(en, oen, rtol=1e-6, atol=1e-6)
