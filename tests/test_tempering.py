"""Parallel tempering: exchange rule (CPU restatement), 2-rank gloo equivalence with the oracle as the engine
(CPU), and -- on the GPU -- continuation / per-replica-beta parity of the kernels, the exchange kernel K6 against
its restatement, and an end-to-end PT run."""
import os

import numpy as np
import pytest

from conftest import load_fixture
from oracle import pt_oracle
from oracle import sa_oracle as so
from scrna_seq_qannealing_clustering_amd import distributed as D
from scrna_seq_qannealing_clustering_amd import models, tempering


class OracleEngine:
    """The CPU oracle behind the tempering driver's engine interface (tests only): the anneal chains of
    oracle/sa_oracle.c and the exchange rule of oracle/pt_oracle.py."""

    def __init__(self, kind, args, seed, **kw):
        self.kind, self.args, self.seed, self.kw = kind, args, seed, kw
        self.st = self.en = None

    def begin(self, ladder, chains, lo, hi):
        self.ladder, self.T, self.lo, self.hi = np.asarray(ladder, dtype=np.float64), len(ladder), lo, hi
        self.rung = np.arange(len(ladder) * chains, dtype=np.int64) % len(ladder)
        self.proposed = self.accepted = 0

    def round(self, num_sweeps, sweep_offset, first, initial_states=None):
        init = initial_states if first else self.st
        betas_local = self.ladder[self.rung[self.lo:self.hi]]
        fn = {"dense": so.sa_dense_philox, "csr": so.sa_csr_rank1_philox, "potts": so.potts_csr_philox}[self.kind]
        self.st, self.en, _ = fn(*self.args, len(betas_local), betas_local, self.seed,
                                 replica_offset=self.lo, init=init, sweep_offset=sweep_offset,
                                 num_sweeps=num_sweeps, **self.kw)

    def exchange(self, rnd, seed, all_energies=None):
        en = self.en if all_energies is None else all_energies
        self.rung, p, a = pt_oracle.exchange_step(en, self.rung, self.ladder, self.T, rnd, seed)
        self.proposed += p
        self.accepted += a

    def energies(self):
        return self.en

    def states(self):
        return self.st

    def rungs(self):
        return self.rung, self.proposed, self.accepted


def potts_case(name="blobs", K=3, gamma=0.05):
    fx = load_fixture(name)
    pm = models.build_dqm_potts(fx.graph(), K, gamma)
    args = (pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), 256, K)
    return fx, pm, args


def test_exchange_rule_is_metropolis_and_moves_rungs_only():
    ladder = tempering.geometric_ladder(0.1, 10.0, 4)
    rung = np.arange(8) % 4
    en = np.array([5.0, 1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 0.0])
    # pair (0,1) of chain 0: hotter replica has HIGHER energy -> arg = (b0-b1)(E0-E1) < 0 -> random;
    # with energies reversed the swap is certain
    new, p, a = pt_oracle.exchange_step(np.array([1.0, 5.0, 2.0, 3.0, 0, 0, 0, 0]), rung, ladder, 4, 0, 7)
    assert p == 4 and new[0] == 1 and new[1] == 0                    # certain swap: colder rung gets lower E
    assert sorted(new[:4].tolist()) == [0, 1, 2, 3] and sorted(new[4:].tolist()) == [0, 1, 2, 3]
    n2, _, _ = pt_oracle.exchange_step(en, rung, ladder, 4, 1, 7)   # odd round: pairs (1,2) only
    assert n2[0] == 0 and n2[3] == 3
    # deterministic in (seed, round)
    x1 = pt_oracle.exchange_step(en, rung, ladder, 4, 0, 99)[0]
    x2 = pt_oracle.exchange_step(en, rung, ladder, 4, 0, 99)[0]
    assert np.array_equal(x1, x2)
    with pytest.raises(ValueError):
        tempering.geometric_ladder(1, 2, 1)


def test_exchange_acceptance_rate_is_metropolis():
    """Over many independent pairs with a fixed arg < 0 the accepted share is exp(arg) (the log form
    -arg < -ln u of the rule, with the chain's own logarithm and random words)."""
    chains, T = 4000, 2
    ladder = np.array([1.0, 2.0])
    for gap, rnd in ((0.25, 0), (1.5, 2)):
        en = np.tile(np.array([gap, 0.0]), chains)            # arg = (1 - 2) * (gap - 0) = -gap
        _, p, a = pt_oracle.exchange_step(en, np.arange(2 * chains) % 2, ladder, T, rnd, 1234)
        assert p == chains
        assert abs(a / p - np.exp(-gap)) < 4.0 * np.sqrt(np.exp(-gap) * (1 - np.exp(-gap)) / chains)


def test_oracle_continuation_equals_one_long_run():
    """sweep_offset + initial states = resumable runs (the checkpoint/resume of SURVEY.md section 5)."""
    fx, pm, args = potts_case()
    betas = np.geomspace(0.5, 20.0, 12)
    full, efull, _ = so.potts_csr_philox(*args, 4, betas, 3, lin_offset=pm.lin_offset)
    a, _, _ = so.potts_csr_philox(*args, 4, betas[:5], 3, lin_offset=pm.lin_offset)
    b, eb, _ = so.potts_csr_philox(*args, 4, betas[5:], 3, lin_offset=pm.lin_offset, init=a, sweep_offset=5)
    assert np.array_equal(full, b) and np.allclose(efull, eb)


def test_zero_rounds_returns_the_initial_ladder_without_touching_results():
    fx, pm, args = potts_case()
    eng = OracleEngine("potts", args, 11, lin_offset=pm.lin_offset)
    out = tempering.parallel_tempering(eng, tempering.geometric_ladder(0.5, 30.0, 4), chains=3, rounds=0,
                                       sweeps_per_round=5, seed=11)
    assert out["rung"].tolist() == (np.arange(12) % 4).tolist() and out["local_states"] is None
    assert out["history"] == [] and out["swap_rate"] == 0.0 and len(out["energies"]) == 0


def _pt_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    fx, pm, args = potts_case()
    eng = OracleEngine("potts", args, 11, lin_offset=pm.lin_offset)
    out = tempering.parallel_tempering(eng, tempering.geometric_ladder(0.5, 30.0, 4), chains=3, rounds=6,
                                       sweeps_per_round=5, seed=11, rank=rank, world=world)
    np.savez(os.path.join(out_dir, "pt%d.npz" % rank), energies=out["energies"], rung=out["rung"],
             states=out["local_states"], lo=out["local_range"][0], hi=out["local_range"][1],
             swap=out["swap_rate"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tempering_equals_single_process(tmp_path):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_pt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    fx, pm, args = potts_case()
    eng = OracleEngine("potts", args, 11, lin_offset=pm.lin_offset)
    ref = tempering.parallel_tempering(eng, tempering.geometric_ladder(0.5, 30.0, 4), chains=3, rounds=6,
                                       sweeps_per_round=5, seed=11)
    outs = [np.load(os.path.join(str(tmp_path), "pt%d.npz" % r)) for r in range(2)]
    for o in outs:
        assert np.array_equal(o["energies"], ref["energies"])          # all-gathered, global order
        assert np.array_equal(o["rung"], ref["rung"])                  # same exchange decisions everywhere
    states = np.concatenate([outs[0]["states"], outs[1]["states"]])
    assert np.array_equal(states, ref["local_states"])                # shards == single process
    assert 0.0 < float(outs[0]["swap"]) <= 1.0
    assert sorted(ref["rung"][:4].tolist()) == [0, 1, 2, 3]


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "csr", "potts"])
def test_gpu_continuation_and_per_replica_beta_parity(kind):
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    fx = load_fixture("aniso")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    if kind == "dense":
        Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
        mk, oargs, ofn, okw = (lambda: Problem.dense(Qs)), (Qs,), so.sa_dense_philox, {}
    elif kind == "csr":
        a = (m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32), float(np.float32(m.c_pair)))
        mk, oargs, ofn, okw = (lambda: Problem.csr_rank1(*a)), a, so.sa_csr_rank1_philox, {}
    else:
        _, pm, a = potts_case("aniso", 8, 0.005)
        mk, oargs, ofn, okw = (lambda: Problem.potts_csr(*a, lin_offset=pm.lin_offset)), a, so.potts_csr_philox, \
            {"lin_offset": pm.lin_offset}
    R = 40                                                    # dense: the workgroup kernel, ragged last group
    betas = np.geomspace(0.01, 5.0, 14)
    full, efull, _ = ofn(*oargs, R, betas, 5, replica_offset=2, **okw)
    with mk() as p:
        p.anneal(R, betas[:6], 5, replica_offset=2)
        p.anneal(R, betas[6:], 5, replica_offset=2, continue_run=True, sweep_offset=6)
        st, en, _ = p.fetch()
        assert np.array_equal(st, full) and np.allclose(en, efull, rtol=1e-9, atol=1e-9)
        # one constant beta per replica (a tempering rung), 7 sweeps
        per = np.geomspace(0.02, 8.0, R)
        p.anneal(R, per, 9, replica_offset=2, num_sweeps=7, sweep_offset=100)
        st2, en2, info2 = p.fetch()
    ost, oen, ostats = ofn(*oargs, R, per, 9, replica_offset=2, sweep_offset=100, num_sweeps=7, **okw)
    assert np.array_equal(st2, ost) and info2["accepted"] == int(ostats[1])
    assert np.allclose(en2, oen, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_gpu_per_replica_beta_across_launch_chunks_of_the_wave_kernel():
    """K1 (variant 1) serves more replicas than are resident at once in chunks; with one temperature PER REPLICA every
    chunk has to read its own slice of the temperatures (round 1 read the first chunk's for all of them)."""
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    rs = np.random.RandomState(12)
    n, R = 40, 40000
    A = rs.normal(size=(n, n)).astype(np.float32)
    Qs = np.ascontiguousarray((A + A.T) / 2)
    per = np.geomspace(0.02, 8.0, R)
    with Problem.dense(Qs) as p:
        p.set_option("variant", 1)
        p.anneal(R, per, 9, num_sweeps=3)
        st, en, info = p.fetch()
        assert p.kernel_name().startswith("k_anneal_dense<") and p.launch_count() >= 2
    ost, oen, ostats = so.sa_dense_philox(Qs, R, per, 9, num_sweeps=3)
    assert np.array_equal(st, ost) and info["accepted"] == int(ostats[1])
    assert np.allclose(en, oen, rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_gpu_parallel_tempering_matches_oracle_engine_and_improves_on_sa():
    """BASELINE config 5 in miniature: DQM K=3 on the reference's blobs graph, 8 rungs x 8 chains."""
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    fx, pm, args = potts_case("blobs", 3, 0.05)
    ladder = tempering.geometric_ladder(0.3, 40.0, 8)
    with Problem.potts_csr(*args, lin_offset=pm.lin_offset) as p:
        out = tempering.parallel_tempering(tempering.ProblemEngine(p, seed=4), ladder, chains=8, rounds=20,
                                           sweeps_per_round=10, seed=4)
    ref = tempering.parallel_tempering(OracleEngine("potts", args, 4, lin_offset=pm.lin_offset), ladder,
                                       chains=8, rounds=20, sweeps_per_round=10, seed=4)
    assert np.array_equal(out["local_states"], ref["local_states"])     # same chain, same exchanges
    assert np.array_equal(out["rung"], ref["rung"])
    assert np.allclose(out["energies"], ref["energies"], rtol=1e-9)
    assert 0.05 < out["swap_rate"] < 1.0
    assert out["history"][-1] <= out["history"][0]
    # the three blobs are disconnected components: labels = components is the natural optimum
    comp = fx.components()
    ids = {c: k for k, c in enumerate(sorted(set(comp.tolist())))}
    e_comp = pm.energies(np.array([[ids[int(c)] for c in comp]]))[0]
    assert out["best_energy"] <= e_comp * (1 - 1e-6)          # device energies use the fp32-stored coefficients


@pytest.mark.gpu
def test_gpu_exchange_kernel_equals_the_restatement():
    """K6 (mi_sa_tempering_exchange) against oracle/pt_oracle.py on arbitrary energies: same rungs, same counts,
    even and odd rounds, ladders of 2 .. 300 temperatures, several rounds chained (rungs no longer in order), and
    the temperatures the next round anneals at follow the rungs."""
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    fx, pm, args = potts_case("blobs", 3, 0.05)
    rs = np.random.RandomState(5)
    with Problem.potts_csr(*args, lin_offset=pm.lin_offset) as p:
        for T, chains in ((2, 1), (5, 7), (8, 16), (3, 4), (300, 2)):
            R = T * chains
            ladder = np.geomspace(0.3, 9.0, T)
            p.tempering_begin(ladder, chains, 0, R)
            p.anneal(R, None, 3, num_sweeps=1)                              # a run must exist before an exchange
            rung = np.arange(R, dtype=np.int64) % T
            tot_p = tot_a = 0
            for rnd in range(5):
                energies = rs.normal(scale=3.0, size=R)
                p.tempering_exchange(rnd, 77, energies)
                rung, pp, aa = pt_oracle.exchange_step(energies, rung, ladder, T, rnd, 77)
                tot_p, tot_a = tot_p + pp, tot_a + aa
                got, gp, ga = p.tempering_state()
                assert np.array_equal(got, rung) and (gp, ga) == (tot_p, tot_a)
                assert np.array_equal(np.sort(got.reshape(chains, T), axis=1), np.tile(np.arange(T), (chains, 1)))
            assert 0 < tot_a < tot_p or T == 2
            # the next round runs every replica at its rung's temperature: equal to per-replica betas given by hand
            p.anneal(R, None, 3, num_sweeps=4, sweep_offset=50, continue_run=True)
            lab, en, _ = p.fetch()
            p.anneal(R, ladder[np.arange(R) % T], 3, num_sweeps=1)          # the state the exchanges started from
            p.anneal(R, ladder[rung], 3, num_sweeps=4, sweep_offset=50, continue_run=True)
            lab2, en2, _ = p.fetch()
            assert np.array_equal(lab, lab2) and np.array_equal(en, en2)
        with pytest.raises(_lib_error()):
            p.tempering_exchange(0, 1, np.zeros(3))


@pytest.mark.gpu
def test_gpu_exchange_from_a_device_buffer_and_ordinary_anneals_in_between():
    """The multi-GPU round without host staging, on one GPU: the energies as a torch tensor ALIASING the library's HBM
    buffer (the all-gather's send buffer), the exchange reading a device tensor (the all-gather's output) == the host
    form == oracle/pt_oracle.py.  And the temperatures of the next round live in a buffer of their own: an ordinary
    anneal with a long schedule on the same handle in between does not change what the next resident round runs at."""
    import torch
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    fx, pm, args = potts_case("blobs", 3, 0.05)
    T, chains = 6, 5
    R = T * chains
    ladder = np.geomspace(0.3, 9.0, T)
    with Problem.potts_csr(*args, lin_offset=pm.lin_offset) as p:
        p.tempering_begin(ladder, chains, 0, R)
        p.anneal(R, None, 3, num_sweeps=2)
        en_dev = p.device_energies()
        lab, en, _ = p.fetch()
        assert en_dev.dtype == torch.float64 and en_dev.is_cuda and np.array_equal(en_dev.cpu().numpy(), en)
        p.tempering_exchange_device(0, 77, en_dev.clone())
        rung, pp, aa = pt_oracle.exchange_step(en, np.arange(R) % T, ladder, T, 0, 77)
        got, gp, ga = p.tempering_state()
        assert np.array_equal(got, rung) and (gp, ga) == (pp, aa)
        with pytest.raises(ValueError):
            p.tempering_exchange_device(1, 77, en_dev[:3].clone())
        # reference: the next resident round straight away
        p.anneal(R, None, 3, num_sweeps=3, sweep_offset=2, continue_run=True)
        want = p.fetch()
        # again, with an ordinary 40-sweep anneal on the handle between the exchange and the resident round
        p.tempering_begin(ladder, chains, 0, R)
        p.anneal(R, None, 3, num_sweeps=2)
        p.tempering_exchange(0, 77, en)
        p.anneal(R, np.geomspace(0.1, 50.0, 40), 123)                       # rewrites the per-sweep temperature buffer
        p.anneal(R, None, 3, num_sweeps=3, sweep_offset=2, initial_states=lab)
        have = p.fetch()
        assert np.array_equal(have[0], want[0]) and np.array_equal(have[1], want[1])
        ost, _, _ = so.potts_csr_philox(*args, R, ladder[rung], 3, lin_offset=pm.lin_offset, init=lab, sweep_offset=2,
                                        num_sweeps=3)
        assert np.array_equal(want[0], ost)


def _lib_error():
    return (ValueError, __import__("scrna_seq_qannealing_clustering_amd")._lib.MiSaError)
