"""Multi-GPU path on CPU: world_size-2 gloo.  Each rank anneals its shard of the global replica ids
(the oracle stands in for the GPU engine here -- the exchange logic is what is under test), then the
ONE exchange of the path runs: packed-key MIN all-reduce + winner broadcast."""
import os
import socket

import numpy as np
import pytest

from conftest import load_fixture
from scrna_seq_qannealing_clustering_amd import distributed as D
from scrna_seq_qannealing_clustering_amd import models


def test_key_packing_orders_like_energy_then_id():
    rng = np.random.RandomState(0)
    es = np.concatenate([rng.normal(scale=1e3, size=200), [0.0, -0.0, 1e-30, -1e-30, 3.5, 3.5]]).astype(np.float32)
    ids = rng.randint(0, 2 ** 32, size=len(es), dtype=np.uint64)
    keys = [D.pack_key(float(e), int(i)) for e, i in zip(es, ids)]
    order = sorted(range(len(es)), key=lambda k: keys[k])
    got = [(float(es[k]), int(ids[k])) for k in order]
    want = sorted(((float(e), int(i)) for e, i in zip(es, ids)))
    assert [e for e, _ in got] == sorted(e for e, _ in want)      # non-decreasing in energy
    ties = [k for k in range(len(got) - 1) if got[k][0] == got[k + 1][0] and np.signbit(got[k][0]) == np.signbit(got[k + 1][0])]
    assert all(got[k][1] < got[k + 1][1] for k in ties)            # equal energies: lower replica id wins
    for e, i in zip(es, ids):
        e2, i2 = D.unpack_key(D.pack_key(float(e), int(i)))
        assert e2 == float(e) and i2 == int(i)
    # signed transport form keeps the order
    s = [D._to_signed(k) for k in keys]
    assert sorted(range(len(s)), key=lambda k: s[k]) == order
    assert all(D._from_signed(D._to_signed(k)) == k for k in keys)


def test_shard_range_covers_everything():
    for R, W in [(4096, 8), (10, 3), (2, 4), (7, 1)]:
        spans = [D.shard_range(R, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == R
        assert all(spans[k][1] == spans[k + 1][0] for k in range(W - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_owner_follows_from_the_global_id():
    """The winner's rank is derived from its global replica id (no second collective): equal to a search over
    shard_range for even, uneven and more-ranks-than-replicas shardings, with and without a base offset."""
    for R, W, base in [(4096, 8, 0), (10, 3, 0), (11, 2, 100), (2, 4, 0), (7, 1, 5), (32768, 8, 4096), (13, 5, 0)]:
        spans = [D.shard_range(R, r, W) for r in range(W)]
        for g in range(R):
            want = next(r for r, (lo, hi) in enumerate(spans) if lo <= g < hi)
            assert D.owner_of(base + g, R, W, base) == want
        with pytest.raises(ValueError):
            D.owner_of(base + R, R, W, base)


def test_fp64_key_orders_like_energy_then_id_and_keeps_what_fp32_loses():
    rng = np.random.RandomState(1)
    bits = D.id_bits_for(32768)
    assert bits == 15 and D.id_bits_for(1) == 1 and D.id_bits_for(2) == 1 and D.id_bits_for(3) == 2
    es = np.concatenate([rng.normal(scale=1e5, size=300), [0.0, -0.0, 1e-300, -1e-300, -105555.06, -105559.73]])
    ids = rng.randint(0, 32768, size=len(es))
    keys = [D.pack_key64(float(e), int(i), bits) for e, i in zip(es, ids)]
    order = sorted(range(len(es)), key=lambda k: keys[k])
    got = es[order]
    assert np.all(np.diff(got) >= -np.abs(got[1:]) * 2.0 ** -(52 - bits) - 1e-290)          # non-decreasing at the key's resolution
    for e, i, k in zip(es, ids, keys):
        e2, i2 = D.unpack_key64(k, bits)
        assert i2 == int(i) and abs(e2 - e) <= abs(e) * 2.0 ** -(52 - bits) + 1e-290
    # two energies that share an fp32 bucket at |E| ~ 1e5 (resolution 0.0078) are told apart
    a, b = -105559.7291, -105559.7301
    assert np.float32(a) == np.float32(b)
    assert D.pack_key64(b, 9, bits) < D.pack_key64(a, 3, bits)
    assert D.pack_key64(a, 3, bits) < D.pack_key64(a, 9, bits)                      # exact tie: lower id wins
    s = [D._to_signed(k) for k in keys]
    assert sorted(range(len(s)), key=lambda k: s[k]) == order
    with pytest.raises(ValueError):
        D.pack_key64(1.0, 1 << bits, bits)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, R, out_dir):
    import torch.distributed as dist
    from oracle import sa_oracle as so
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    fx = load_fixture("noisy_moons")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
    betas = models.make_beta_schedule(30, models.default_beta_range(m))
    lo, hi = D.shard_range(R, rank, world)
    st, en, _ = so.sa_dense_philox(Qs, hi - lo, betas, 99, replica_offset=lo)     # this rank's shard
    k = int(np.argmin(en.astype(np.float32)))
    key = D.pack_key(float(en[k]), lo + k)
    e, gid, owner, state = D.global_best(key, st[k], num_reads=R)
    k64 = int(np.argmin(en))
    e64, gid64, owner64, state64 = D.global_best_f64(float(en[k64]), lo + k64, st[k64], num_reads=R)
    allen = D.gather_energies(en, num_reads=R)                  # ONE collective: the counts follow from shard_range
    allen2 = D.gather_energies(en)                              # counts gathered first
    with pytest.raises(ValueError):
        D.global_best(key, st[k])                               # num_reads missing: raised before any collective is issued
    with pytest.raises(ValueError):
        D.global_best_f64(float(en[k64]), R + 5, st[k64], num_reads=R)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), e=e, gid=gid, owner=owner, state=state, allen=allen,
             allen2=allen2, lo=lo, hi=hi, e64=e64, gid64=gid64, owner64=owner64, state64=state64)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_global_best_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle import sa_oracle as so
    R, world = 11, 2                                           # uneven shards: ranks get 6 + 5
    mp.spawn(_worker, args=(world, _free_port(), R, str(tmp_path)), nprocs=world, join=True)
    fx = load_fixture("noisy_moons")
    m = models.build_bqm_qubo(fx.graph(), 0.05)
    Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
    betas = models.make_beta_schedule(30, models.default_beta_range(m))
    st, en, _ = so.sa_dense_philox(Qs, R, betas, 99)            # all replicas in one process
    best = int(np.argmin(en.astype(np.float32)))
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for o in outs:
        assert int(o["gid"]) == best                            # global replica id of the winner
        assert float(o["e"]) == float(np.float32(en[best]))
        assert int(o["owner"]) == (0 if best < 6 else 1)
        assert np.array_equal(o["state"], st[best])             # winner's labels on every rank
        assert np.array_equal(o["allen"], en)                   # all-gather in global replica order
        assert np.array_equal(o["allen2"], en)
        b64 = int(np.argmin(en))                                # fp64 key: the exact minimum, its exact energy
        assert int(o["gid64"]) == b64 and float(o["e64"]) == float(en[b64])
        assert int(o["owner64"]) == (0 if b64 < 6 else 1) and np.array_equal(o["state64"], st[b64])
    assert np.array_equal(outs[0]["state"], outs[1]["state"])


def test_global_best_without_process_group_is_identity():
    e, gid, owner, state = D.global_best(D.pack_key(-3.25, 17), np.array([1, 0, 1], dtype=np.uint8))
    assert (e, gid, owner) == (-3.25, 17, 0) and state.tolist() == [1, 0, 1]
    assert D.gather_energies(np.array([1.0, 2.0])).tolist() == [1.0, 2.0]
    e, gid, owner, state = D.global_best_f64(-3.25, 17, np.array([1, 0, 1], dtype=np.uint8), num_reads=32)
    assert (e, gid, owner) == (-3.25, 17, 0) and state.tolist() == [1, 0, 1]


def test_bench_starts_its_own_ranks_as_children_when_typed_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` (what a driver types): N > 1 without WORLD_SIZE in the environment builds the
    torch.distributed.run command for CHILD processes (the parent has made no GPU call); under a launcher, or with
    one GPU, the process is a rank itself."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    argv = ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    cmd = bench.rank_launch_command(bench.parse_args(argv), argv, port=29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    k = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[k + 1:] == argv                                   # the ranks get the arguments as typed
    assert bench.rank_launch_command(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"]) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.rank_launch_command(bench.parse_args(argv), argv) is None
    # the launcher relays the children's output and exit code
    rc = bench.launch_ranks([sys.executable, "-c", "print('{\"ok\": 1}'); raise SystemExit(3)"])
    assert rc == 3


def _rccl_worker(out_path):
    """One rank, backend "nccl" (= RCCL) on cuda:0: the three collectives of the path on device tensors."""
    import json
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    key = D.pack_key64(-105559.7291, 26, D.id_bits_for(4096))
    k = torch.tensor([D._to_signed(key)], dtype=torch.int64, device=dev)
    dist.all_reduce(k, op=dist.ReduceOp.MIN)                                     # (C1)
    st = torch.arange(300, dtype=torch.uint8, device=dev)
    dist.broadcast(st, src=0)                                                    # (C2)
    en = torch.linspace(-3.0, 5.0, 128, dtype=torch.float64, device=dev)
    out = torch.empty(128, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, en)                                         # (C3)
    torch.cuda.synchronize()
    res = {"key_ok": D._from_signed(int(k.item())) == key, "bcast_ok": bool((st.cpu() == torch.arange(300, dtype=torch.uint8)).all()),
           "gather_ok": bool(torch.equal(out, en)), "backend": dist.get_backend()}
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_executes_the_paths_collectives_on_this_gpu(tmp_path):
    """RCCL itself (torch.distributed backend "nccl"), one rank on the box's one GPU, in a child process with
    HSA_ENABLE_IPC_MODE_LEGACY=0 in place before its first HIP call: the 8-byte MIN all-reduce of the packed key, the
    broadcast of n label bytes and the all-gather of energies into a device tensor run and return what they were given.
    (Several ranks need several GPUs: the driver's scaling run.)"""
    import json
    import torch.multiprocessing as mp
    out = str(tmp_path / "rccl.json")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(out,))
    p.start()
    p.join(240)
    assert p.exitcode == 0
    res = json.load(open(out))
    assert res == {"key_ok": True, "bcast_ok": True, "gather_ok": True, "backend": "nccl"}
