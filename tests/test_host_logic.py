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


def test_native_slot_order_equals_its_restatement():
    """mi_sa_plan_slot_order (host code of the native library) == the numpy restatement in oracle/model_oracle.py on
    SNN-like, ring, complete and empty graphs; a 50 000-variable graph is ordered in well under a second."""
    import time
    rs = np.random.RandomState(3)
    cases = []
    for n, seed in ((1500, 1), (700, 2), (65, 3)):
        nodes, eu, ev, w, _ = graphs.synthetic_snn(n, 5, 15, 15, 4, seed=seed, spread=2.5)
        m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
        cases.append((m.rowptr, m.col))
    n = 300
    cases.append((np.arange(0, 2 * n + 1, 2), np.stack([(np.arange(n) - 1) % n, (np.arange(n) + 1) % n], axis=1).ravel()))
    cases.append((np.zeros(200, dtype=np.int64), np.zeros(0, dtype=np.int64)))          # 199 isolated variables
    for rowptr, col in cases:
        assert np.array_equal(models.slot_independent_order(rowptr, col), mo.slot_independent_order(rowptr, col))
        assert np.array_equal(models.slot_independent_order(rowptr, col, slot=16), mo.slot_independent_order(rowptr, col, slot=16))
    n = 50000
    eu = np.concatenate([np.arange(n)] * 3)
    ev = np.concatenate([(np.arange(n) + d) % n for d in (1, 7, 131)])
    rowptr, col, _ = models._csr_from_edges(n, np.minimum(eu, ev).astype(np.int32), np.maximum(eu, ev).astype(np.int32),
                                            np.ones(len(eu)))
    t0 = time.perf_counter()
    perm = models.slot_independent_order(rowptr, col)
    assert time.perf_counter() - t0 < 2.0
    assert np.array_equal(np.sort(perm), np.arange(n))
    inv = np.argsort(perm)
    assert not np.any((inv[np.repeat(np.arange(n), np.diff(rowptr))] >> 6) == (inv[col] >> 6))   # no edge inside a block
    with pytest.raises(Exception):                                                   # column out of range
        models.slot_independent_order(np.arange(0, 131), np.r_[np.arange(1, 130), 500])


def test_padded_slot_layout_keeps_every_edge_between_blocks():
    """mi_sa_plan_slot_layout == its restatement; the layout of a strongly clustered graph (what a recursive bisection
    leaves: here the planted clusters of the generator, one at a time and in pairs) needs more blocks than ceil(n / 64)
    and then holds no edge inside a block; seats are unique; a complete graph falls back to the packed layout."""
    nodes, eu, ev, w, lab = graphs.synthetic_snn(1500, 5, 15, 15, 6, seed=1)
    padded_somewhere = False
    for keep in ((0,), (1,), (2, 3), (0, 1, 2, 3, 4, 5)):
        idx = np.flatnonzero(np.isin(lab, keep))
        renum = -np.ones(1500, dtype=np.int64)
        renum[idx] = np.arange(len(idx))
        sel = np.isin(eu, idx) & np.isin(ev, idx)
        rowptr, col, _ = models._csr_from_edges(len(idx), renum[eu[sel]].astype(np.int32), renum[ev[sel]].astype(np.int32), w[sel])
        n = len(idx)
        pos, nslots, clashes = models.padded_slot_layout(rowptr, col)
        opos, onslots, oclashes = mo.padded_slot_layout(rowptr, col)
        assert np.array_equal(pos, opos) and (nslots, clashes) == (onslots, oclashes)
        assert clashes == 0 and nslots >= (n + 63) // 64 and len(set(pos.tolist())) == n and pos.max() < nslots * 64
        rows = np.repeat(np.arange(n), np.diff(rowptr))
        assert not np.any((pos[rows] >> 6) == (pos[col] >> 6))                       # no edge inside a block
        assert np.bincount(pos >> 6, minlength=nslots).max() <= 64
        padded_somewhere |= nslots > (n + 63) // 64
        rp2, c2, v2 = models.pad_csr(rowptr, col, np.arange(len(col), dtype=np.float64), pos, nslots * 64)
        assert rp2[-1] == len(col) and np.array_equal(np.diff(rp2)[pos], np.diff(rowptr))
        assert all(np.all(np.diff(c2[rp2[i]:rp2[i + 1]]) > 0) for i in pos[::37])     # rows ascending
    assert padded_somewhere
    n = 70                                                          # complete graph: no number of blocks <= max helps
    col = np.array([j for i in range(n) for j in range(n) if j != i])
    rowptr = np.arange(0, n * (n - 1) + 1, n - 1)
    pos, nslots, clashes = models.padded_slot_layout(rowptr, col, max_slots=8)
    assert nslots == 2 and clashes > 0 and sorted(pos.tolist()) == sorted(set(pos.tolist()))
    assert (np.array_equal(pos, mo.padded_slot_layout(rowptr, col, max_slots=8)[0]))
    pos, nslots, clashes = models.padded_slot_layout(rowptr, col, max_slots=100)     # 70 blocks of one variable each
    assert clashes == 0 and nslots >= 70


def test_padded_slot_layout_of_a_very_large_graph_is_the_packed_identity():
    """Beyond 262144 variables the planner does not run its greedy passes (O(n x slots), repeated while the layout
    grows): seats = indices, as mi_sa_plan_slot_order returns the identity there; clashes are counted, not removed."""
    import time
    n = (1 << 18) + 70
    i = np.arange(n)
    col = np.stack([(i - 1) % n, (i + 1) % n], axis=1).ravel().astype(np.int32)     # a ring: every variable clashes
    rowptr = np.arange(0, 2 * n + 1, 2).astype(np.int32)
    t0 = time.perf_counter()
    pos, nslots, clashes = models.padded_slot_layout(rowptr, col)
    assert time.perf_counter() - t0 < 2.0
    assert np.array_equal(pos, i) and nslots == (n + 63) // 64 and clashes == n


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


def test_the_isa_guard_for_inline_asm_lds_reads_sees_a_premature_use(tmp_path):
    """scripts/check_asm_lds.py (run by __graft_entry__.build()): a register written by an asm ds_read is pending until a
    lgkmcnt wait retires it -- in issue order, so lgkmcnt(1) retires all but the youngest -- and naming it before is flagged."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("check_asm_lds", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "scripts", "check_asm_lds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    good = """_Zkernel_a:
\t;;#ASMSTART
\tds_read_b32 v10, v3
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b32 v11, v4
\t;;#ASMEND
\tv_add_u32_e32 v5, v6, v7
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(1)
\t;;#ASMEND
\tv_add_f32_e32 v12, v10, v12
\ts_waitcnt lgkmcnt(0)
\tv_add_f32_e32 v12, v11, v12
\ts_endpgm
"""
    p = tmp_path / "good.s"
    p.write_text(good)
    assert mod.check_asm(str(p)) == (1, 2, [])
    bad = good.replace("v_add_f32_e32 v12, v10, v12", "v_add_f32_e32 v12, v11, v12")      # v11 is still in flight there
    q = tmp_path / "bad.s"
    q.write_text(bad)
    kernels, reads, viol = mod.check_asm(str(q))
    assert (kernels, reads) == (1, 2) and len(viol) == 1 and viol[0][3][0][0] == 11
    ranged = good.replace("v_add_u32_e32 v5, v6, v7", "v_pk_mul_f32 v[20:21], v[10:11], v[22:23]")   # a register pair names it too
    r = tmp_path / "ranged.s"
    r.write_text(ranged)
    assert len(mod.check_asm(str(r))[2]) == 1
