#!/usr/bin/env python3
"""bench.py -- spin-flip updates/s of the anneal on the PBMC3k-sized SNN graph-partition model.

Workload (BASELINE.json configs[1]): synthetic PBMC3k-like SNN graph, n = 2638 cells (k=5, dim=15,
trim 15; no PBMC data ships with the reference), `clustering_bqm` model (BQM_clustering.py:29-47,
gamma_factor 0.05, k 8) resident in HBM; ONE STEP = one anneal of 4096 replicas x 1000 sweeps per GPU
(explicit geometric beta schedule, seed 1234).  The timed kernel is the one `MI355XSampler.sample_qubo`
runs for this model: K2, the CSR form (sparse cut term + uniform pair term; dE = Q_i . x evaluated from
the CSR rows).  The dense form (K1w, 27.8 MB fp32 Q) and the Potts kernel run beside it, untimed
(`other_kernels`); `--kernel dense` makes the dense kernel the timed one.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Multi-GPU: replicas shard (weak scaling: 4096 replicas per GPU, global ids rank*4096..), the model is
replicated, and each step ends with the ONE exchange the path has: a 64-bit MIN all-reduce of the
packed (energy, replica id) key + a broadcast of the winner's labels (RCCL over xGMI).

Rank 0 prints one JSON line.  `roofline.achieved` = algorithmic bytes per update (SURVEY.md section 8d:
CSR deg_i*8 + 8 averaged over the variables; dense 4n) x updates per launch / mean launch time (HIP
events).  The model (0.34 MB in CSR form, 27.8 MB dense) lives in L2 / Infinity Cache, so this is an
effective-bandwidth figure; `traffic` is what the fabric counters saw.  `cpu_baseline` times the oracle's
neal restatement on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CELLS, K_NN, DIM, ORD, N_CLUSTERS = 2638, 5, 15, 15, 9
SPREAD = 3.0        # cluster spread of the surrogate: the clusters overlap, ONE connected component (as real SNN graphs have a giant one)
REPLICAS_PER_GPU, SWEEPS, SEED = 4096, 1000, 1234
HBM_PEAK_GBPS = 8000.0


def build_workload():
    from scrna_seq_qannealing_clustering_amd import graphs, models
    nodes, eu, ev, w, _ = graphs.synthetic_snn(N_CELLS, K_NN, DIM, ORD, N_CLUSTERS, seed=0, spread=SPREAD)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    m = models.build_bqm_qubo(G, 0.05, k=8)
    Qs = np.ascontiguousarray(m.dense_Qs().astype(np.float32))
    betas = models.make_beta_schedule(SWEEPS, models.default_beta_range(m))
    return m, Qs, betas, (eu, ev, w), G


def real_neal_probe(Qs, betas):
    """if the real dwave-neal happens to be importable on this host, time it on a bounded
    sample as well (probe only -- nothing is installed or fetched).  Returns None when it is absent."""
    sampler = None
    for mod, attr in (("neal", "SimulatedAnnealingSampler"), ("dwave.samplers", "SimulatedAnnealingSampler")):
        try:
            sampler = getattr(__import__(mod, fromlist=[attr]), attr)()
            break
        except Exception:
            continue
    if sampler is None:
        return None
    n = Qs.shape[0]
    iu, ju = np.triu_indices(n, 1)
    Q = {(int(i), int(i)): float(Qs[i, i]) for i in range(n)}
    Q.update({(int(i), int(j)): 2.0 * float(Qs[i, j]) for i, j in zip(iu, ju) if Qs[i, j] != 0.0})
    sweeps = 50
    t0 = time.perf_counter()
    ss = sampler.sample_qubo(Q, num_reads=4, num_sweeps=sweeps, beta_range=(float(betas[0]), float(betas[-1])), seed=SEED)
    t = time.perf_counter() - t0
    return {"value": 4 * sweeps * n / t, "unit": "spin-flip updates/s", "cores": 1, "seconds": t,
            "best_energy": float(ss.first.energy), "sample": "real dwave-neal, 4 reads x %d sweeps, same Q" % sweeps}


def cpu_baseline(Qs, betas, seconds_target=15.0):
    """The oracle's restatement of dwave-neal (Ising, fp64, xorshift128+, sequential sweeps) on a bounded
    sample of the SAME workload: same Q, every (len/sweeps)-th beta of the same schedule."""
    from oracle import sa_oracle as so
    h, J, off = so.qubo_to_ising_dense(Qs.astype(np.float64))
    # the GPU box shares its host: one GPU's share is 16 cores (more threads only oversubscribe)
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("MI_CPU_THREADS", "16")))
    sweeps = 50
    sub = betas[:: max(1, len(betas) // sweeps)][:sweeps]
    # calibrate on 1 read, 1 thread ("how neal runs": reads are sequential, single-threaded)
    t0 = time.perf_counter()
    _, e1, st1 = so.sa_ising_neal_dense(h, J, 1, sub, seed=SEED, threads=1)
    t1 = time.perf_counter() - t0
    single = float(st1[0]) / t1
    reads = int(max(cores, min(64 * cores, (seconds_target / max(t1, 1e-3)) * cores)))
    t0 = time.perf_counter()
    _, en, st = so.sa_ising_neal_dense(h, J, reads, sub, seed=SEED, threads=cores)
    t = time.perf_counter() - t0
    # energy at the SAME sweep count as the GPU run (the metric's "best QUBO energy vs neal"): one read per
    # thread over the full schedule
    t0 = time.perf_counter()
    _, en_full, _ = so.sa_ising_neal_dense(h, J, cores, betas, seed=SEED + 1, threads=cores)
    t_full = time.perf_counter() - t0
    return {
        "value": float(st[0]) / t, "unit": "spin-flip updates/s", "cores": cores, "kind": "port",
        "single_thread_value": single,
        "best_energy": float((en + off).min()),
        "same_sweeps": {"reads": cores, "sweeps": int(len(betas)), "best_energy": float((en_full + off).min()),
                        "mean_energy": float((en_full + off).mean()), "seconds": t_full},
        "sample": "oracle neal restatement (fp64 Ising, xorshift128+), same dense Q, %d reads x %d sweeps "
                  "(every %d-th beta of the 1000-sweep schedule), OpenMP over reads on %d threads; "
                  "real dwave-neal is not installable offline" % (reads, len(sub), max(1, len(betas) // sweeps), cores),
        "seconds": t,
        "real_neal": real_neal_probe(Qs, betas),         # null: not importable on this host
    }


def pmc_traffic(replicas, sweeps, launches, kernel):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same workload
    (profiles/r01_pmc_traffic.json, written by scripts/pmc_traffic.py on the GPU box: FETCH_SIZE doubled
    per the gfx950 note + WRITE_SIZE, averaged over the launches of one step).  None when no profile matches
    this launch shape."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    if (rec.get("replicas") == replicas and rec.get("sweeps") == sweeps and rec.get("launches") == launches
            and rec.get("kernel") == kernel):
        return float(rec["hbm_bytes_per_launch"])
    return None


def binding_resource(kernel):
    """What actually binds the timed kernel, from the committed SQ-counter passes (profiles/r01_k2_binding.json,
    scripts/pmc_k2.sh + scripts/k2_binding.py): the byte model of `roofline` is an effective-bandwidth figure
    for a model that lives in L2."""
    if not kernel.startswith("k_anneal_csr_rank1"):
        return None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r01_k2_binding.json")))
    except (OSError, ValueError):
        return None
    return {k: rec[k] for k in ("resource", "utilisation", "salu_issue_utilisation", "l2_hit_rate", "source")}


def other_kernels(m, Qs, betas, graph, rank_device, headline):
    """Short untimed-region runs of the other kernels on the same graph (reported beside the headline, never
    part of `value`): the same QUBO on the other binary kernel, and K3 = BASELINE config 3 (DQM K=8)."""
    from scrna_seq_qannealing_clustering_amd import models
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
    out = {}
    n = m.num_variables
    R, S = REPLICAS_PER_GPU, 200
    b = models.make_beta_schedule(S, models.default_beta_range(m))
    if headline == "csr":
        p = Problem.dense(Qs, offset=0.0, device=rank_device)
        name, kname = "dense_bqm", "k_anneal_dense_wg<44,4>"
    else:
        p = Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                              float(np.float32(m.c_pair)), device=rank_device, order="slots")
        name, kname = "csr_rank1_bqm", "k_anneal_csr_rank1<16, true>"
    with p:
        p.anneal(R, b, SEED)
        ms = p.kernel_ms()
        st, en, info = p.fetch()
        out[name] = {"kernel": kname, "replicas": R, "sweeps": S, "kernel_ms": ms,
                     "updates_per_s": R * S * n / (ms * 1e-3),
                     "best_energy": float(m.energies(st[int(np.argmin(en))][None, :])[0]),
                     "acceptance": info["accepted"] / info["proposals"]}
    pm = models.build_dqm_potts(graph, 8, 0.005)
    with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), n, 8,
                           lin_offset=pm.lin_offset, device=rank_device, order="slots") as p:
        b = models.make_beta_schedule(S, default_potts_beta_range(pm))
        p.anneal(R, b, SEED)
        ms = p.kernel_ms()
        lab, en, info = p.fetch()
        out["potts_dqm_k8"] = {"kernel": "k_anneal_potts<16>", "replicas": R, "sweeps": S, "kernel_ms": ms,
                               "updates_per_s": R * S * n / (ms * 1e-3), "best_energy": float(en.min()),
                               "acceptance": info["accepted"] / info["proposals"]}
    # K4: the replica-batched energy x^T Q x of the headline's final states on the matrix cores (the one
    # GEMM-shaped step of the path: 2 n^2 flop per state), checked against the exact fp64 path
    from scrna_seq_qannealing_clustering_amd.engine import energy_dense
    X = st.astype(np.uint8)
    energy_dense(Qs, X[:64], device=rank_device, path=2)                           # warm
    e_mfma, ms = energy_dense(Qs, X, device=rank_device, path=2, return_ms=True)
    e_exact = energy_dense(Qs, X[:256], device=rank_device, path=1)
    flops = 2.0 * n * n * len(X)
    out["energy_mfma"] = {"kernel": "k_energy_dense_mfma", "states": int(len(X)), "kernel_ms": ms,
                          "tflops": flops / (ms * 1e-3) / 1e12, "peak_tflops_f32_input_mfma": 157.3,
                          "frac_of_mfma_peak": flops / (ms * 1e-3) / 157.3e12,
                          "max_rel_diff_vs_exact_fp64": float(np.max(np.abs(e_mfma[:256] - e_exact) /
                                                                     np.maximum(1.0, np.abs(e_exact))))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU)
    ap.add_argument("--sweeps", type=int, default=SWEEPS)
    ap.add_argument("--kernel", choices=("csr", "dense"), default="csr",
                    help="timed kernel: csr = K2 (what the sampler runs for this model), dense = K1w")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from scrna_seq_qannealing_clustering_amd import distributed as D
    from scrna_seq_qannealing_clustering_amd.engine import Problem

    # MI_BENCH_BACKEND=gloo + MI_BENCH_DEVICE=0 rehearse the N>1 code path with several ranks on ONE GPU
    # (the real multi-GPU run uses RCCL, one rank per GPU, launched by torch.distributed.run)
    rank, world, local = D.init_from_env(backend=os.environ.get("MI_BENCH_BACKEND"))
    if "MI_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MI_BENCH_DEVICE"])
    coll_dev = "cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu"
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)

    m, Qs, betas, (eu, ev, w), graph = build_workload()
    if args.sweeps != SWEEPS:
        from scrna_seq_qannealing_clustering_amd import models
        betas = models.make_beta_schedule(args.sweeps, models.default_beta_range(m))
    R = args.replicas
    n = Qs.shape[0]
    t_up0 = time.perf_counter()
    if args.kernel == "dense":                                     # model resident in HBM before timing
        prob = Problem.dense(Qs, offset=0.0, device=local)
        kernel_name, bytes_per_update = "k_anneal_dense_wg<44,4>", 4.0 * n
        layout = "dense fp32 Q 27.8 MB"
    else:
        prob = Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                                 float(np.float32(m.c_pair)), device=local, order="slots")   # as the sampler does
        kernel_name = "k_anneal_csr_rank1<16, true>"          # byte-state variant (n <= 9216)
        bytes_per_update = 8.0 * float(np.diff(m.rowptr).mean()) + 8.0      # deg_i*(4+4) + 8
        layout = "CSR (cut term) + uniform pair term, %.1f neighbours per cell on average" % float(np.diff(m.rowptr).mean())

    upload_ms = (time.perf_counter() - t_up0) * 1e3                # host arrays -> HBM (incl. the sweep ordering)

    def step(i):
        prob.anneal(R, betas, SEED + i, replica_offset=rank * R)
        idx, e, key, state = prob.best()                           # K5 on device; waits for the anneal
        return D.global_best(key, state)                           # C1 + C2 (identity at N=1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    kernel_ms = []
    best = None
    for i in range(args.steps):
        best = step(args.warmup + i)
        kernel_ms.append(prob.kernel_ms())                         # HIP events on the engine's stream
    launches = prob.launch_count()                                 # a long schedule is served by several launches
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    t_f0 = time.perf_counter()
    _, en, info = prob.fetch()
    fetch_ms = (time.perf_counter() - t_f0) * 1e3                  # R x n state bytes + R energies -> host
    updates_per_step = world * R * len(betas) * n
    value = updates_per_step * args.steps / elapsed
    k_ms = float(np.mean(kernel_ms))                               # all launches of one step
    launch_ms = k_ms / launches                                    # = rocprofv3's average duration of the kernel
    sweeps_per_launch = len(betas) / launches
    alg_bytes = bytes_per_update * R * sweeps_per_launch * n       # per launch (one GPU)
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    if args.kernel == "dense":
        row_bytes = info["accepted"] * (((n + 255) // 256) * 256 * 4)   # padded rows the accepted flips consumed
    else:
        row_bytes = R * len(betas) * ((n + 63) // 64) * 64 * (16 * 8 + 4 + 4 + 32)   # per slot: adjacency, lin, meta, 4 in-slot entries
    best_state = best[3]
    cut_edges = int(np.sum(best_state[eu] != best_state[ev]))
    out = {
        "metric": "spin-flip updates/sec (node) + best QUBO energy vs neal, PBMC3k SNN",
        "value": value, "unit": "spin-flip updates/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PBMC3k-sized synthetic SNN (n=2638, k=5, dim=15, trim 15), clustering_bqm QUBO "
                               "(gamma_factor 0.05, k 8), %s, resident in HBM, %d replicas/GPU x %d sweeps, "
                               "geometric beta, seed 1234" % (layout, R, len(betas)),
                   "kernel": args.kernel,
                   "n": n, "replicas_per_gpu": R, "sweeps": int(len(betas)), "parallelism": "replicas sharded x%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(R, len(betas), launches, kernel_name),
                     "kernel": kernel_name, "kernel_ms": launch_ms,
                     "launches_per_step": launches, "sweeps_per_launch": sweeps_per_launch,
                     "kernel_ms_per_step": k_ms,
                     "algorithmic_bytes_per_update": bytes_per_update,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "binding_resource": binding_resource(kernel_name),
                     "rows_GBps": row_bytes / (k_ms * 1e-3) / 1e9,
                     "acceptance": info["accepted"] / info["proposals"]},
        "best_energy": float(m.energies(best_state[None, :])[0]),
        "mean_energy": float(np.mean(en)),
        "replicas_at_best_energy": int(np.sum(en <= en.min() + 1e-6 * abs(en.min()))),
        "best_energy_device_f32": float(best[0]),
        "host_buffers": {"model_upload_ms": upload_ms, "results_fetch_ms": fetch_ms,
                         "note": "outside the timed region: `value` is measured with the model resident in HBM"},
        "best_cut_edges": cut_edges,
        "energy_lower_bound": float(-m.info["gamma"] * n * n / 4),
    }
    if rank == 0:
        if world == 1:
            out["other_kernels"] = other_kernels(m, Qs, betas, graph, local, args.kernel)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(Qs, betas)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    prob.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
