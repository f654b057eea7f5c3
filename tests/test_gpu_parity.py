"""Parity of the HIP path against the CPU oracle, through the C ABI (libmi_sa.so).  GPU only.

Bars: binary states / accepted-move counts / integer edge cuts are BIT-EXACT (the chain computes in
fp32 with a fully specified operation order); device energies are fp64 sums of the fp32 matrix entries
and must match the oracle's fp64 re-evaluation to rel 1e-9; against the caller's fp64 coefficients the
bar is rel 1e-5 (fp32 storage of Q)."""
import ctypes as C

import numpy as np
import pytest

from conftest import GRAPH_NAMES, load_fixture
from oracle import sa_oracle as so
from scrna_seq_qannealing_clustering_amd import _lib, engine, models
from scrna_seq_qannealing_clustering_amd.engine import Problem

pytestmark = pytest.mark.gpu

E_RTOL = 1e-9      # device fp64 energy vs oracle fp64 energy, same fp32 matrix
E_RTOL_F64Q = 1e-5  # vs energies from the caller's fp64 coefficients


def fixture_model(name, gf=0.05):
    fx = load_fixture(name)
    m = models.build_bqm_qubo(fx.graph(), gf)
    return fx, m, np.ascontiguousarray(m.dense_Qs().astype(np.float32))


def random_sym(n, seed, scale=1.0):
    rng = np.random.RandomState(seed)
    A = rng.normal(scale=scale, size=(n, n)).astype(np.float32)
    Q = ((A + A.T) * np.float32(0.5)).astype(np.float32)
    return np.ascontiguousarray((Q + Q.T) * np.float32(0.5))


def run_gpu(Qs, R, betas, seed, **kw):
    with Problem.dense(Qs, offset=kw.pop("offset", 0.0)) as p:
        p.anneal(R, betas, seed, **kw)
        st, en, info = p.fetch()
    return st, en, info


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_trajectory_parity_on_reference_graphs(name):
    fx, m, Qs = fixture_model(name)
    betas = models.make_beta_schedule(60, models.default_beta_range(m))
    st, en, info = run_gpu(Qs, 16, betas, 1234)
    ost, oen, ostats = so.sa_dense_philox(Qs, 16, betas, 1234)
    assert np.array_equal(st, ost)                                   # flip for flip
    assert info["accepted"] == int(ostats[1]) and info["proposals"] == int(ostats[0])
    assert np.allclose(en, oen, rtol=E_RTOL, atol=1e-9)
    assert np.array_equal(so.cut_edges(fx.eu, fx.ev, st), so.cut_edges(fx.eu, fx.ev, ost))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 257, 300, 777, 1000])
def test_trajectory_parity_ragged_sizes(n):
    Qs = random_sym(n, seed=n)
    betas = np.geomspace(0.05, 5.0, 12)
    st, en, info = run_gpu(Qs, 5, betas, 42 + n, offset=1.25)
    ost, oen, ostats = so.sa_dense_philox(Qs, 5, betas, 42 + n, offset=1.25)
    assert np.array_equal(st, ost)
    assert info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=E_RTOL, atol=1e-9)


@pytest.mark.parametrize("variant,unit_rows", [(1, 0), (2, 2), (2, 4), (3, 0)])
@pytest.mark.parametrize("n,R", [(1, 3), (64, 16), (65, 37), (300, 33), (1000, 20), (2638, 18)])
def test_both_kernels_follow_the_same_chain(variant, unit_rows, n, R):
    """K1 (wave per replica), K1w (16-replica workgroup, LDS ring; ragged last workgroup) and K1m (fields
    as MFMA accumulators, row updates on the matrix cores) are the same Markov chain: identical states to
    the oracle for every (size, replica count), with initial states, field re-synchronisation and a
    replica offset in play."""
    Qs = random_sym(n, seed=1000 + n)
    betas = np.geomspace(0.05, 4.0, 9)
    init = np.random.RandomState(n).randint(0, 2, size=(R, n)).astype(np.uint8)
    ost, oen, ostats = so.sa_dense_philox(Qs, R, betas, 7, replica_offset=5, init=init, resync_interval=4)
    with Problem.dense(Qs) as p:
        p.set_option("variant", variant)
        p.set_option("unit_rows", unit_rows)
        p.anneal(R, betas, 7, replica_offset=5, initial_states=init, resync_interval=4)
        st, en, info = p.fetch()
        p.set_option("pace", 0)
        p.anneal(R, betas, 7, replica_offset=5, initial_states=init, resync_interval=4)
        st2, _, _ = p.fetch()
    assert np.array_equal(st, ost) and np.array_equal(st2, ost)
    assert info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=E_RTOL, atol=1e-9)


@pytest.mark.parametrize("variant", [2, 3, 4])
@pytest.mark.parametrize("chunk", [3, 5])
def test_chunked_and_scheduled_runs_equal_one_launch(variant, chunk):
    """A run cut into launches of `chunk` sweeps -- served by K1w, by K1m, or by whichever the device-side
    scheduler picks per chunk (variant 4) -- leaves bits AND cached fields in HBM between launches, so it is
    the same chain as one launch: equal to the oracle, including mid-run re-synchronisation points that
    do not coincide with the cuts."""
    n, R = 700, 50
    Qs = random_sym(n, seed=77)
    betas = np.geomspace(0.02, 30.0, 23)                      # hot -> cold: the scheduler switches kernels
    ost, oen, ostats = so.sa_dense_philox(Qs, R, betas, 3, replica_offset=1, resync_interval=7)
    with Problem.dense(Qs) as p:
        p.set_option("variant", variant)
        p.set_option("chunk_sweeps", chunk)
        p.set_option("mfma_permille", 150)
        p.anneal(R, betas, 3, replica_offset=1, resync_interval=7)
        st, en, info = p.fetch()
    assert np.array_equal(st, ost)
    assert info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=E_RTOL, atol=1e-9)


def test_initial_states_resync_and_zero_sweeps():
    fx, m, Qs = fixture_model("noisy_moons")
    rng = np.random.RandomState(9)
    init = rng.randint(0, 2, size=(6, 256)).astype(np.uint8)
    betas = models.make_beta_schedule(40, models.default_beta_range(m))
    for resync in (0, 1, 7):
        st, en, info = run_gpu(Qs, 6, betas, 5, initial_states=init, resync_interval=resync)
        ost, oen, ostats = so.sa_dense_philox(Qs, 6, betas, 5, init=init, resync_interval=resync)
        assert np.array_equal(st, ost) and info["accepted"] == int(ostats[1])
    # zero sweeps: the initial states come back, with their energies
    st, en, _ = run_gpu(Qs, 6, betas[:0], 5, initial_states=init)
    assert np.array_equal(st, init)
    assert np.allclose(en, so.energy_dense_f64(Qs, init), rtol=E_RTOL, atol=1e-9)
    # zero sweeps without initial states: the replica's own random initial state
    st, en, _ = run_gpu(Qs, 6, betas[:0], 5)
    ost, _, _ = so.sa_dense_philox(Qs, 6, betas[:0], 5)
    assert np.array_equal(st, ost) and 0.3 < st.mean() < 0.7


def test_replica_sharding_is_invariant():
    """Global replica ids key the RNG: [0,8) in one run == [0,4) + [4,8) in two (multi-GPU rule)."""
    _, m, Qs = fixture_model("varied")
    betas = models.make_beta_schedule(30, models.default_beta_range(m))
    st, en, _ = run_gpu(Qs, 8, betas, 77)
    a, ea, _ = run_gpu(Qs, 4, betas, 77, replica_offset=0)
    b, eb, _ = run_gpu(Qs, 4, betas, 77, replica_offset=4)
    assert np.array_equal(st, np.concatenate([a, b])) and np.array_equal(en, np.concatenate([ea, eb]))
    ob, _, _ = so.sa_dense_philox(Qs, 4, betas, 77, replica_offset=4)
    assert np.array_equal(b, ob)


def test_circles_reaches_proven_optimum(kat):
    """P3 on the reference's own graph: best of 64 x 1000 sweeps == known global optimum, cut = 0,
    and equal to the oracle's best at the same (R, sweeps, schedule, seed)."""
    fx, m, Qs = fixture_model("noisy_circles")
    betas = models.make_beta_schedule(1000, models.default_beta_range(m))
    with Problem.dense(Qs) as p:
        p.anneal(64, betas, 1234)
        st, en, info = p.fetch()
        idx, e_best, key, s_best = p.best()
    ost, oen, _ = so.sa_dense_philox(Qs, 64, betas, 1234)
    assert np.array_equal(st, ost)
    assert idx == int(np.argmin(en.astype(np.float32))) or en[idx] == en.min()
    assert e_best == pytest.approx(kat["noisy_circles"]["comp0_E_closed"], rel=1e-6)
    assert int(so.cut_edges(fx.eu, fx.ev, s_best[None, :])[0]) == 0 and int(s_best.sum()) == 128
    assert m.energies(s_best[None, :])[0] == pytest.approx(-2951.8108596597776, rel=1e-12)
    assert np.allclose(en, oen, rtol=E_RTOL, atol=1e-9)
    from scrna_seq_qannealing_clustering_amd.distributed import unpack_key
    e32, gid = unpack_key(key)
    assert gid == idx and e32 == pytest.approx(e_best, rel=1e-6)


def test_energy_kernel_parity():
    """K4, both forms: exact-fp64 VALU (rel 1e-12) and the f32-input MFMA contraction (rel 1e-6, stated in
    include/mi_sa.h: fp32 partial sums of <= 32 terms folded into fp64)."""
    fx, m, Qs = fixture_model("aniso")
    rng = np.random.RandomState(3)
    X = rng.randint(0, 2, size=(37, 256)).astype(np.uint8)
    X[0] = 0
    X[1] = 1
    want = so.energy_dense_f64(Qs, X, offset=-2.0)
    got_valu = engine.energy_dense(Qs, X, offset=-2.0, path=1)
    got_mfma = engine.energy_dense(Qs, X, offset=-2.0, path=2)
    got_auto = engine.energy_dense(Qs, X, offset=-2.0)
    assert np.allclose(got_valu, want, rtol=1e-12, atol=1e-9)
    assert np.allclose(got_mfma, want, rtol=1e-6, atol=1e-3)
    assert np.array_equal(got_auto, got_mfma) or np.allclose(got_auto, got_mfma, rtol=1e-12)   # R >= 32 -> MFMA
    assert got_valu[0] == -2.0 and got_mfma[0] == -2.0
    # Z2 symmetry of the balanced-partition model, E(x) == E(1 - x), up to the fp32 rounding of Q
    assert np.allclose(engine.energy_dense(Qs, 1 - X, path=1), engine.energy_dense(Qs, X, path=1), rtol=1e-6, atol=2e-3)


@pytest.mark.parametrize("n,R", [(1, 1), (2, 33), (31, 64), (33, 65), (130, 3), (257, 100), (1000, 130)])
def test_energy_kernel_ragged_shapes(n, R):
    """A = I-style check with an ASYMMETRIC-looking operand: random symmetric Q, random states, every
    tile edge (n, R not multiples of 32 / 64) exercised on the MFMA path."""
    Qr = random_sym(n, 8 + n)
    Xr = np.random.RandomState(R).randint(0, 2, size=(R, n)).astype(np.uint8)
    want = so.energy_dense_f64(Qr, Xr, offset=0.5)
    assert np.allclose(engine.energy_dense(Qr, Xr, offset=0.5, path=1), want, rtol=1e-12, atol=1e-9)
    scale = np.abs(Qr).sum() if n > 1 else 1.0
    assert np.allclose(engine.energy_dense(Qr, Xr, offset=0.5, path=2), want, rtol=1e-6, atol=1e-6 * scale / max(n, 1))
    # single-entry operands: E must pick exactly Q[i][j] + Q[j][i] + Q[i][i] + Q[j][j]
    if n >= 3:
        Xe = np.zeros((32, n), dtype=np.uint8)
        for r in range(32):
            Xe[r, r % n] = 1
            Xe[r, (3 * r + 1) % n] = 1
        assert np.allclose(engine.energy_dense(Qr, Xe, path=2), so.energy_dense_f64(Qr, Xe), rtol=1e-6, atol=1e-6)


def test_energy_kernel_asymmetric_matrix_is_not_sent_to_the_symmetric_mfma_path():
    """The MFMA energy kernel multiplies only the blocks on and above the diagonal (Qs symmetric).  An upper-triangular
    QUBO matrix at a batch the library would send there (R >= 32) is evaluated on the exact path instead; asking for
    the MFMA path by name with such a matrix is an error, not a wrong number."""
    n, R = 300, 40
    rs = np.random.RandomState(12)
    U = np.triu(rs.normal(size=(n, n))).astype(np.float32)
    X = (rs.rand(R, n) < 0.5).astype(np.uint8)
    Xf = X.astype(np.float64)
    want = np.einsum("ri,ij,rj->r", Xf, U.astype(np.float64), Xf)
    assert np.allclose(engine.energy_dense(U, X), want, rtol=1e-12, atol=1e-9)             # auto: exact path
    assert np.allclose(engine.energy_dense(U, X, path=1), want, rtol=1e-12, atol=1e-9)
    with pytest.raises(_lib.MiSaError):
        engine.energy_dense(U, X, path=2)
    S = ((U + U.T) / 2).astype(np.float32)
    assert np.allclose(engine.energy_dense(S, X, path=2), np.einsum("ri,ij,rj->r", Xf, S.astype(np.float64), Xf), rtol=1e-6, atol=1e-3)


def test_energy_mfma_full_size_batch():
    """BASELINE config 2 shape: 4096 states x n = 2638 on the matrix cores vs the exact VALU form."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, _ = graphs.synthetic_snn(2638, 5, 15, 15, 9, seed=0)
    m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
    Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
    X = np.random.RandomState(0).randint(0, 2, size=(4096, 2638)).astype(np.uint8)
    e_mfma, ms_mfma = engine.energy_dense(Qs, X, path=2, return_ms=True)
    e_valu, ms_valu = engine.energy_dense(Qs[:, :], X[:128], path=1, return_ms=True)
    assert np.allclose(e_mfma[:128], e_valu, rtol=2e-6)
    assert np.allclose(e_mfma, m.energies(X), rtol=1e-5)
    flops = 2.0 * 2638 * 2638 * 4096
    print("K4 MFMA: %.3f ms for 4096 x 2638 (%.1f TFLOP/s f32)" % (ms_mfma, flops / ms_mfma / 1e9))


def test_one_shot_c_entry_point():
    _, m, Qs = fixture_model("blobs")
    betas = np.ascontiguousarray(models.make_beta_schedule(25, models.default_beta_range(m)))
    lib = _lib.load()
    R, n = 7, 256
    st = np.zeros((R, n), dtype=np.uint8)
    en = np.zeros(R)
    stats = np.zeros(3, dtype=np.uint64)
    rc = lib.mi_sa_qubo_dense_f32(
        Qs.ctypes.data_as(C.POINTER(C.c_float)), n, 0.0, R, len(betas),
        betas.ctypes.data_as(C.POINTER(C.c_double)), 31337, None,
        st.ctypes.data_as(C.POINTER(C.c_uint8)), en.ctypes.data_as(C.POINTER(C.c_double)),
        stats.ctypes.data_as(C.POINTER(C.c_uint64)), 0)
    assert rc == 0, lib.mi_last_error()
    ost, oen, ostats = so.sa_dense_philox(Qs, R, betas, 31337)
    assert np.array_equal(st, ost) and int(stats[0]) == R * 25 * n and int(stats[1]) == int(ostats[1])


def test_error_behaviour():
    lib = _lib.load()
    with pytest.raises(ValueError):
        Problem.dense(np.array([[0.0, 1.0], [2.0, 0.0]], dtype=np.float32))       # not symmetric
    with pytest.raises(_lib.MiSaError) as ei:
        # the size check precedes every access to the matrix: a tiny buffer is enough to ask for n = 65537
        tiny = np.zeros(4, dtype=np.float32)
        _lib.check(lib.mi_sa_problem_create_dense_f32(tiny.ctypes.data_as(C.POINTER(C.c_float)), 65537, 0.0, 0,
                                                      C.byref(C.c_void_p())))
    assert ei.value.code == -5
    with pytest.raises(_lib.MiSaError):
        Problem.dense(np.zeros((4, 4), dtype=np.float32), device=99)
    p = Problem.dense(np.zeros((4, 4), dtype=np.float32))
    with pytest.raises(RuntimeError):
        p.fetch()
    with pytest.raises(_lib.MiSaError):
        p.anneal(0, [1.0], 1)
    with pytest.raises(_lib.MiSaError):
        p.anneal(2, [1.0, -1.0], 1)
    with pytest.raises(_lib.MiSaError):
        p.anneal(2, [1.0, float("nan")], 1)
    with pytest.raises(ValueError):
        p.anneal(2, [1.0], 1, initial_states=np.zeros((3, 4)))
    # options: unknown keys and values outside a key's range are refused, the setting stays what it was
    for key, value in (("no_such_option", 1), ("xl_batched", 3), ("xl_chain", 3), ("xl_chain", -1), ("xl_chunk", 0),
                       ("variant", 9)):
        with pytest.raises(_lib.MiSaError):
            p.set_option(key, value)
    for key, value in (("xl_batched", 2), ("xl_chain", 2), ("xl_chain", 0), ("xl_chunk", 3), ("xl_cold_permille", 0), ("xl_async", 0)):
        p.set_option(key, value)
    p.close()
    assert b"" == b"" and lib.mi_last_error() is not None
    # holes of a padded layout: only positions without couplings, only Potts problems (a binary CSR model marks its
    # holes by lin = +inf); planner arguments are checked before anything is touched
    rowptr, col = np.array([0, 1, 2, 2], dtype=np.int32), np.array([1, 0], dtype=np.int32)
    val = np.ones(2, dtype=np.float32)
    with Problem.potts_csr(rowptr, col, val, 0.1, 3, 2) as pp:
        bad = np.array([1, 0, 0], dtype=np.uint8)                                  # variable 0 has a coupling
        assert lib.mi_sa_problem_set_absent(pp._h, bad.ctypes.data_as(C.POINTER(C.c_uint8))) == -1
        ok = np.array([0, 0, 1], dtype=np.uint8)
        _lib.check(lib.mi_sa_problem_set_absent(pp._h, ok.ctypes.data_as(C.POINTER(C.c_uint8))))
        pp.anneal(4, np.geomspace(0.1, 5.0, 6), 3)
        lab, _, info = pp.fetch()
        assert not lab[:, 2].any() and info["proposals"] == 4 * 6 * 3              # (the engine counts the caller's n)
    with Problem.csr_rank1(rowptr, col, val, np.zeros(3, dtype=np.float32), 0.0) as pb:
        assert lib.mi_sa_problem_set_absent(pb._h, ok.ctypes.data_as(C.POINTER(C.c_uint8))) == -5
        # pair-term weights: positive integers; the weighted variables have no sparse couplings and share one slot
        w32 = lambda *a: np.array(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))
        assert lib.mi_sa_problem_set_pair_weights(pb._h, w32(3, 1, 1)) == -1       # variable 0 has a coupling
        assert lib.mi_sa_problem_set_pair_weights(pb._h, w32(1, 1, 0)) == -1       # weights are >= 1
        assert lib.mi_sa_problem_set_pair_weights(pb._h, None) == -1
        assert lib.mi_sa_problem_set_pair_weights(pb._h, w32(1, 1, 5)) == -1       # its slot holds variables with couplings
        _lib.check(lib.mi_sa_problem_set_pair_weights(pb._h, w32(1, 1, 1)))        # all ones: an unweighted model
    with Problem.potts_csr(rowptr, col, val, 0.1, 3, 2) as pp:
        assert lib.mi_sa_problem_set_pair_weights(pp._h, np.ones(3, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))) == -5
    lin130 = np.zeros(130, dtype=np.float32)
    with Problem.csr_rank1(np.zeros(131, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), lin130, 0.1) as pw:
        w = np.ones(130, dtype=np.int32)
        w[3], w[100] = 2, 4                                                        # two slots hold weighted variables
        assert lib.mi_sa_problem_set_pair_weights(pw._h, w.ctypes.data_as(C.POINTER(C.c_int32))) == -1
        w[3] = 1
        w[70] = 7                                                                  # slot 1 alone
        _lib.check(lib.mi_sa_problem_set_pair_weights(pw._h, w.ctypes.data_as(C.POINTER(C.c_int32))))
        pw.anneal(3, np.geomspace(0.1, 5.0, 6), 3)
        st, en, _ = pw.fetch()
        a = w.astype(np.float64)
        assert np.allclose(en, float(np.float32(0.1)) * 0.5 * ((st @ a) ** 2 - st @ (a * a)), rtol=1e-12, atol=1e-12)   # (the fp32 coefficient: no fp64 energy model was set)
    pos = np.zeros(3, dtype=np.int64)
    ns = C.c_int(0)
    assert lib.mi_sa_plan_slot_layout(rowptr.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                                      0, 64, 8, pos.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(ns), None) == -1
    badcol = np.array([7, 0], dtype=np.int32)
    assert lib.mi_sa_plan_slot_layout(rowptr.ctypes.data_as(C.POINTER(C.c_int32)), badcol.ctypes.data_as(C.POINTER(C.c_int32)),
                                      3, 64, 8, pos.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(ns), None) == -1


def test_full_size_properties_pbmc3k_surrogate():
    """BASELINE config 2 shape (n = 2638): size-independent properties + a 2-replica bit-exact spot check."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, _ = graphs.synthetic_snn(2638, 5, 15, 15, 9, seed=0)
    m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
    Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
    betas = models.make_beta_schedule(20, models.default_beta_range(m))
    with Problem.dense(Qs) as p:
        p.anneal(256, betas, 1234)
        st, en, info = p.fetch()
        p.anneal(256, betas, 1234)
        st2, en2, _ = p.fetch()
        p.anneal(2, betas[:8], 5, replica_offset=100)
        s2, e2, i2 = p.fetch()
    assert np.array_equal(st, st2) and np.array_equal(en, en2)                 # deterministic
    assert set(np.unique(st).tolist()) <= {0, 1}
    host = m.energies(st)                                                      # fp64, structured form
    assert np.allclose(en, host, rtol=E_RTOL_F64Q)
    cut = so.cut_edges(eu, ev, st)
    s = st.sum(axis=1).astype(np.float64)
    cut_w = np.array([w[(x[eu] != x[ev])].sum() for x in st])
    closed = 8 * cut_w + m.info["gamma"] * (s * s - 2638 * s)                   # E = k cut + gamma (s^2 - n s)
    assert np.allclose(host, closed, rtol=1e-9)
    assert cut.min() >= 0 and info["proposals"] == 256 * 20 * 2638
    assert 0.05 < info["accepted"] / info["proposals"] < 0.7
    o2, oe2, os2 = so.sa_dense_philox(Qs, 2, betas[:8], 5, replica_offset=100)
    assert np.array_equal(s2, o2) and i2["accepted"] == int(os2[1])


def test_full_size_scheduled_dense_run_equals_the_workgroup_kernel_alone():
    """BASELINE config 2 shape, a whole hot-to-cold schedule of 160 sweeps (5 chunks of 32): by default the hot chunks go to
    K1m (fields as MFMA accumulators) and the rest to K1w; with `mfma_permille = 0` K1w serves all of them.  Same chain:
    identical states, energies and acceptance counts -- and the kernel name says that both kernels ran."""
    from scrna_seq_qannealing_clustering_amd import graphs
    nodes, eu, ev, w, _ = graphs.synthetic_snn(2638, 5, 15, 15, 9, seed=0)
    m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
    Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
    betas = models.make_beta_schedule(160, models.default_beta_range(m))
    with Problem.dense(Qs) as p:
        p.anneal(2048, betas, 99, resync_interval=50)
        name = p.kernel_name()
        st, en, info = p.fetch()
        p.set_option("mfma_permille", 0)
        p.anneal(2048, betas, 99, resync_interval=50)
        name0 = p.kernel_name()
        st0, en0, info0 = p.fetch()
        # 128 .. 1024 replicas (at most a wavefront per SIMD): the wave-per-replica kernel by the library's own choice --
        # the same chain, so its 512 replicas are the first 512 of the run above
        p.set_option("mfma_permille", 600)
        p.anneal(512, betas, 99, resync_interval=50)
        assert p.kernel_name().startswith("k_anneal_dense<")
        st5, en5, _ = p.fetch()
        assert np.array_equal(st5, st[:512]) and np.array_equal(en5, en[:512])
    assert "dense_mfma" in name and "dense_wg" in name and "dense_mfma" not in name0
    assert np.array_equal(st, st0) and np.array_equal(en, en0)
    assert info["accepted"] == info0["accepted"]
    assert np.allclose(en, m.energies(st), rtol=E_RTOL_F64Q)
    o2, _, os2 = so.sa_dense_philox(Qs, 2, betas[:40], 99, resync_interval=50)       # spot check against the oracle
    with Problem.dense(Qs) as p:
        p.set_option("chunk_sweeps", 8)
        p.anneal(2, betas[:40], 99, resync_interval=50)                                # (R < 32: K1, one wave per replica)
        s2, _, _ = p.fetch()
    assert np.array_equal(s2, o2)


@pytest.mark.parametrize("n,R,sweeps", [(4097, 3, 3), (9000, 2, 2), (17000, 2, 2)])
def test_dense_beyond_4096_variables_workgroup_per_replica(n, R, sweeps):
    """K1x (one 512-thread workgroup per replica, fields spread over its registers, Q rows streamed from HBM)
    is the same chain as K1: identical states and flip counts to the oracle at n = 4097 (2 chunks, ragged),
    9000 (3 chunks) and 17000 (5 chunks), with initial states, a re-synchronisation and a replica offset.
    Energies come from the cached fp32 fields here (1e-5 relative)."""
    rng = np.random.RandomState(n)
    Qs = np.zeros((n, n), dtype=np.float32)                      # ~2 % of the pairs coupled, symmetric
    iu, ju = rng.randint(0, n, size=n * n // 100), rng.randint(0, n, size=n * n // 100)
    vals = (rng.rand(len(iu)).astype(np.float32) - 0.5)
    Qs[iu, ju] = vals
    Qs = np.triu(Qs, 1)
    Qs = np.ascontiguousarray(Qs + Qs.T)
    Qs[np.arange(n), np.arange(n)] = rng.randn(n).astype(np.float32)
    assert np.array_equal(Qs, Qs.T)
    betas = np.geomspace(0.3, 3.0, sweeps)
    init = rng.randint(0, 2, size=(R, n)).astype(np.uint8)
    for kw in (dict(), dict(init=init, resync_interval=2)):
        ost, oen, ostats = so.sa_dense_philox(Qs, R, betas, 11, replica_offset=5, **kw)
        with Problem.dense(Qs) as p:
            # 2 = K1x (a workgroup per replica); 1 = K1g (all replicas together, 64 rows per GEMM-shaped pass on the matrix
            # cores -- what runs of >= 256 replicas, or of n >= 16384, take by default)
            # K1g's chain of a group of eight blocks: one fused launch (2; the default up to 512 replicas), or a DIAG + a
            # small pass per block (1)
            for mode, chain, name in ((2, 0, "k_anneal_dense_xl"), (1, 2, "k_xg_chain + k_xg_panel"), (1, 1, "k_xg_diag + k_xg_panel")):
                p.set_option("xl_batched", mode)
                p.set_option("xl_chain", chain)
                p.anneal(R, betas, 11, replica_offset=5, initial_states=kw.get("init"),
                         resync_interval=kw.get("resync_interval", 0))
                st, en, info = p.fetch()
                assert p.kernel_name().startswith(name)
                assert info["accepted"] == int(ostats[1]) and info["accepted"] > n // 10
                assert np.array_equal(st, ost)
                assert np.allclose(en, oen, rtol=1e-5, atol=1e-3)


def test_batched_dense_kernel_many_replicas_equal_workgroup_per_replica():
    """K1g at the replica counts it is meant for (300 = 5 flag words, a ragged last DIAG workgroup) against K1x on the same
    model: identical states, accepted counts and (cached-field) energies over a hot-to-cold schedule; zero sweeps return
    the initial states with their energies."""
    n, R = 4500, 300
    rng = np.random.RandomState(5)
    A = (rng.rand(n, n).astype(np.float32) - 0.5) * (rng.rand(n, n) < 0.02)
    Qs = np.triu(A, 1)
    Qs = np.ascontiguousarray(Qs + Qs.T)
    Qs[np.arange(n), np.arange(n)] = rng.randn(n).astype(np.float32)
    betas = np.geomspace(0.05, 20.0, 6)
    with Problem.dense(Qs) as p:
        p.anneal(R, betas, 3, resync_interval=4)
        assert p.kernel_name().startswith("k_xg_chain")                     # the default for >= 256 replicas (fused chain up to 512)
        st, en, info = p.fetch()
        p.set_option("xl_batched", 2)
        p.anneal(R, betas, 3, resync_interval=4)
        assert p.kernel_name().startswith("k_anneal_dense_xl")
        st2, en2, info2 = p.fetch()
        assert np.array_equal(st, st2) and info["accepted"] == info2["accepted"] and info["accepted"] > R * n
        assert np.allclose(en, en2, rtol=1e-6)
        p.set_option("xl_batched", 1)
        p.anneal(R, betas[:0], 3, initial_states=st)
        st3, en3, _ = p.fetch()
        assert np.array_equal(st3, st) and np.allclose(en3, en, rtol=1e-5)
        # a cooling run: K1g in chunks of two sweeps while they accept > 30 %, then K1x for the rest, continuing from K1g's
        # states and cached fields -- the same chain as either kernel alone
        cool = np.geomspace(0.02, 40.0, 12)
        p.set_option("xl_batched", 2)
        p.anneal(R, cool, 9)
        ref = p.fetch()
        p.set_option("xl_batched", 0)
        p.set_option("xl_chunk", 2)
        p.set_option("xl_cold_permille", 300)
        p.anneal(R, cool, 9)
        name = p.kernel_name()
        got = p.fetch()
        assert "k_xg_chain" in name and "k_anneal_dense_xl" in name
        assert np.array_equal(got[0], ref[0]) and got[2]["accepted"] == ref[2]["accepted"]
        assert np.allclose(got[1], ref[1], rtol=1e-6)
        # the hand-over is decided per chunk on the host -- by a worker thread of the problem, so the call returns at once
        # (as every other anneal does); "xl_async" = 0 keeps it in the caller: same result, and the call takes the run's time
        import time
        p.set_option("xl_async", 0)
        t0 = time.perf_counter()
        p.anneal(R, cool, 9)
        t_in_caller = time.perf_counter() - t0
        in_caller = p.fetch()
        p.set_option("xl_async", 1)
        t0 = time.perf_counter()
        p.anneal(R, cool, 9)
        t_call = time.perf_counter() - t0
        assert p.kernel_name() == name                        # (joins the worker)
        again = p.fetch()
        for other in (in_caller, again):
            assert np.array_equal(other[0], ref[0]) and np.array_equal(other[1], got[1]) and other[2]["accepted"] == ref[2]["accepted"]
        assert t_call < 0.5 * t_in_caller, (t_call, t_in_caller)
        p.set_option("xl_cold_permille", 0)                   # never hand over: K1g alone, same result
        p.anneal(R, cool, 9)
        assert "dense_xl" not in p.kernel_name()
        alone = p.fetch()
        assert np.array_equal(alone[0], ref[0]) and alone[2]["accepted"] == ref[2]["accepted"]
        p.set_option("xl_chain", 1)                           # a DIAG and a small pass per block instead of the fused chain
        p.anneal(R, cool, 9)
        assert p.kernel_name().startswith("k_xg_diag")
        per_block = p.fetch()
        p.set_option("xl_chain", 0)
        assert np.array_equal(per_block[0], ref[0]) and per_block[2]["accepted"] == ref[2]["accepted"]
        assert np.array_equal(per_block[1], alone[1])         # the same cached fields, bit for bit
        import os
        os.environ["MI_XG_ONE_STREAM"] = "1"                  # without CU-masked streams: the same kernels in one stream
        try:
            p.set_option("xl_batched", 1)
            p.anneal(R, cool, 9)
            one = p.fetch()
        finally:
            del os.environ["MI_XG_ONE_STREAM"]
        assert np.array_equal(one[0], ref[0]) and one[2]["accepted"] == ref[2]["accepted"]
        # one constant temperature per replica (a tempering round), continued from the states on the device with the
        # random stream of sweep 12 onwards: both kernels again
        rung = np.geomspace(0.05, 5.0, R)
        outs = []
        for mode in (2, 1):
            p.set_option("xl_batched", mode)
            p.anneal(R, cool[:3], 9)
            p.anneal(R, rung, 9, num_sweeps=2, sweep_offset=3, continue_run=True)
            outs.append(p.fetch())
        assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][2]["accepted"] == outs[1][2]["accepted"]


def test_full_size_properties_config4_dense_50k():
    """BASELINE config 4 in its stated form: a 50 000-cell synthetic SNN graph (built on the GPU by snn.build_snn),
    the clustering_bqm QUBO as a DENSE fp32 matrix resident in HBM (10.6 GB), K1x.  Two replicas x two sweeps at
    temperatures from the middle of the schedule against the oracle's dense chain: states, accepted counts, energies
    (from the cached fp32 fields: 1e-3 at this size), and the integer edge cut of both."""
    from scrna_seq_qannealing_clustering_amd import snn
    n = 50000
    rng = np.random.RandomState(1)
    centers = rng.normal(scale=4.0, size=(30, 15))
    X = (centers[rng.randint(0, 30, size=n)] + rng.normal(size=(n, 15))).astype(np.float32)
    g = snn.build_snn(X, 5, 0.0, 15)
    nodes, eu, ev, w = g.edge_list()
    m = models.build_bqm_qubo(g.to_graph(), 0.05)
    Qs = np.full((n, n), np.float32(m.c_pair / 2.0), dtype=np.float32)          # Qs_ij = (c_pair + S_ij) / 2
    rows = np.repeat(np.arange(n), np.diff(m.rowptr))
    Qs[rows, m.col] += (m.val / 2.0).astype(np.float32)
    Qs[np.arange(n), np.arange(n)] = m.lin.astype(np.float32)
    betas = models.make_beta_schedule(1000, models.default_beta_range(m))[620:622]
    with Problem.dense(Qs) as p:
        p.set_option("xl_batched", 2)                         # K1x: a workgroup per replica
        p.anneal(2, betas, 77, replica_offset=4094)
        st, en, info = p.fetch()
        assert p.kernel_name().startswith("k_anneal_dense_xl<13>")
        p.set_option("xl_batched", 1)                         # K1g: the same two replicas through the batched passes
        p.anneal(2, betas, 77, replica_offset=4094)
        stg, eng, infog = p.fetch()
        assert p.kernel_name().startswith("k_xg_chain")
        assert np.array_equal(stg, st) and infog["accepted"] == info["accepted"] and np.allclose(eng, en, rtol=1e-6)
        # and a full first wave of workgroups: 256 replicas x 1 sweep at the hot end, K1g (its default) against K1x
        hot = models.make_beta_schedule(1000, models.default_beta_range(m))[:1]
        p.set_option("xl_batched", 0)
        p.anneal(256, hot, 5)
        assert p.kernel_name().startswith("k_xg_chain")
        sg, eg, ig = p.fetch()
        p.set_option("xl_batched", 2)
        p.anneal(256, hot, 5)
        sx, ex, ix = p.fetch()
        assert np.array_equal(sg, sx) and ig["accepted"] == ix["accepted"] and ig["accepted"] > 0.99 * 256 * n
        assert np.allclose(eg, ex, rtol=1e-6)
        # bench.py's own shape (`other_kernels.dense_xl_50k`: one GPU's share of config 4, 1024 replicas): from 1024
        # replicas up K1g runs its NON-fused chain -- a DIAG and a small pass per block on 8 reserved CUs, beside the
        # full pass of the previous group on a second CU-masked stream, an event ring between them.  One sweep at the
        # hot end + one from the middle of the schedule against K1x, and again with both parts in one stream.
        sched = models.make_beta_schedule(1000, models.default_beta_range(m))
        two = np.ascontiguousarray(sched[[0, 620]])
        p.set_option("xl_batched", 0)
        p.anneal(1024, two, 31)
        name_k = p.kernel_name()
        assert "k_xg_diag" in name_k and "k_xg_panel" in name_k and "k_xg_chain" not in name_k, name_k
        sk, ek, ik = p.fetch()
        p.set_option("xl_batched", 2)
        p.anneal(1024, two, 31)
        assert p.kernel_name().startswith("k_anneal_dense_xl")
        sy, ey, iy = p.fetch()
        assert np.array_equal(sk, sy) and ik["accepted"] == iy["accepted"] and ik["accepted"] > 1024 * n
        assert np.allclose(ek, ey, rtol=1e-6)
        import os
        os.environ["MI_XG_ONE_STREAM"] = "1"
        try:
            p.set_option("xl_batched", 0)
            p.anneal(1024, two, 31)
            s1, e1, i1 = p.fetch()
        finally:
            del os.environ["MI_XG_ONE_STREAM"]
        assert np.array_equal(s1, sy) and i1["accepted"] == iy["accepted"] and np.array_equal(e1, ek)
    # replicas 63 and 64 of that run (the last of the first range of 64 and the first of the second) against the oracle
    ost2, oen2, _ = so.sa_dense_philox(Qs, 2, two, 31, replica_offset=63)
    assert np.array_equal(sk[63:65], ost2) and np.allclose(ek[63:65], oen2, rtol=1e-3)
    ost, oen, ostats = so.sa_dense_philox(Qs, 2, betas, 77, replica_offset=4094)
    del Qs
    assert info["proposals"] == 2 * 2 * n and info["accepted"] == int(ostats[1]) and info["accepted"] > 100
    assert np.array_equal(st, ost)
    # K1x reports the energy its cached fp32 fields imply (no second pass over 5 GB of rows per replica): 25 000
    # fields of magnitude ~1e3 each carry fp32 rounding, 1e-3 relative at this size; the oracle's is an fp64
    # re-evaluation, equal to the caller's model to fp32 storage of Q
    assert np.allclose(en, oen, rtol=1e-3)
    assert np.allclose(oen, m.energies(st), rtol=1e-6)
    assert np.array_equal(so.cut_edges(eu, ev, st), so.cut_edges(eu, ev, ost))


def test_energy_kernel_fp64_matrix():
    """mi_energy_dense_f64: the caller-model energies of a dense problem's samples, against numpy fp64."""
    from scrna_seq_qannealing_clustering_amd.engine import energy_dense_f64
    rs = np.random.RandomState(9)
    for n, R in ((1, 1), (65, 3), (300, 70), (1030, 9)):
        A = rs.normal(size=(n, n))
        Qs = (A + A.T) / 2
        X = (rs.rand(R, n) < 0.4).astype(np.uint8)
        got = energy_dense_f64(Qs, X, offset=1.25)
        Xf = X.astype(np.float64)
        want = np.einsum("ri,ij,rj->r", Xf, Qs, Xf) + 1.25
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        energy_dense_f64(np.zeros((3, 4)), np.zeros((1, 3), dtype=np.uint8))
