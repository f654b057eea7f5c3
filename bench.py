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
packed (fp64 energy, replica id) key + a broadcast of the winner's labels (RCCL over xGMI).
`python bench.py --gpus N` typed without a launcher starts the N ranks itself, as child processes.

Rank 0 prints one JSON line.  The model (0.34 MB in CSR form) lives in the L2s, never in HBM traffic terms, so the
`roofline` object of the CSR kernel is an L2 roofline: `achieved` = bytes the kernel's loads request from L2
per launch (adjacency + linear terms of every slot of every sweep of every wavefront, counted from the launch
shape; the TCC_REQ counter of profiles/r03_* agrees) / mean launch time (HIP events), `peak` = 34.5 TB/s
(MI355X_MICROARCH.md, L2 aggregate), `traffic` = what the fabric counters saw (HBM side).  `effective` keeps
SURVEY.md 8d's per-update byte model (deg_i*8 + 8) as an effective-bandwidth figure with no fraction.  The one
shape whose Q really lives in HBM (n = 50 000 dense, 10.6 GB; kernel K1g) is reported under
`other_kernels.dense_xl_50k` with its MFMA roofline.
`cpu_baseline` times the oracle on this host: the neal restatement (value), the SAME chain as the GPU kernel
(`same_chain`) and the neal restatement at equal reads and sweeps (`equal_reads`).
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC for RCCL on this driver: must be in the environment before the first HIP call of the process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CELLS, K_NN, DIM, ORD, N_CLUSTERS = 2638, 5, 15, 15, 9
SPREAD = 3.0        # cluster spread of the surrogate: the clusters overlap, ONE connected component (as real SNN graphs have a giant one)
REPLICAS_PER_GPU, SWEEPS, SEED = 4096, 1000, 1234
HBM_PEAK_GBPS = 8000.0
L2_PEAK_GBPS = 34500.0          # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s (8 XCDs x 4 MiB)


def build_workload():
    from scrna_seq_qannealing_clustering_amd import graphs, models
    nodes, eu, ev, w, truth = graphs.synthetic_snn(N_CELLS, K_NN, DIM, ORD, N_CLUSTERS, seed=0, spread=SPREAD)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    G.truth = truth                                  # planted cluster of every cell (reads_500 cuts a subgraph out)
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


def wilson_interval(hits, n, z=1.96):
    """95 % Wilson score interval of a binomial share."""
    if n == 0:
        return [0.0, 1.0]
    ph = hits / n
    den = 1.0 + z * z / n
    mid = (ph + z * z / (2 * n)) / den
    half = z * np.sqrt(ph * (1 - ph) / n + z * z / (4.0 * n * n)) / den
    return [float(max(0.0, mid - half)), float(min(1.0, mid + half))]


def equal_reads_stats(e_neal, e_gpu, best_known):
    """Two samplers at EQUAL reads and sweeps, as distributions: share of the reads that end at the best energy known for
    the model (`hit`, within 1e-9 relative) with 95 % Wilson bounds, mean +- standard error, and the two differences in
    units of their standard errors.  Both chains are Metropolis sweeps over the same schedule: neither is expected to be
    lower; which best-of-N wins in one run is chance."""
    e_neal, e_gpu = np.asarray(e_neal, dtype=np.float64), np.asarray(e_gpu, dtype=np.float64)
    tol = 1e-9 * abs(best_known)
    out = {}
    for name, e in (("neal", e_neal), ("gpu", e_gpu)):
        hits = int(np.sum(e <= best_known + tol))
        out[name] = {"reads": int(len(e)), "best_energy": float(e.min()), "mean_energy": float(e.mean()),
                     "mean_standard_error": float(e.std(ddof=1) / np.sqrt(len(e))),
                     "hits_at_best_known": hits, "hit_rate": hits / len(e), "hit_rate_95": wilson_interval(hits, len(e))}
    pn, pg = out["neal"]["hit_rate"], out["gpu"]["hit_rate"]
    pool = (out["neal"]["hits_at_best_known"] + out["gpu"]["hits_at_best_known"]) / (len(e_neal) + len(e_gpu))
    se_hit = float(np.sqrt(max(pool * (1 - pool), 1e-12) * (1.0 / len(e_neal) + 1.0 / len(e_gpu))))
    se_mean = float(np.hypot(out["neal"]["mean_standard_error"], out["gpu"]["mean_standard_error"]))
    out["best_known_energy"] = float(best_known)
    out["hit_rate_difference_sigmas"] = float((pg - pn) / se_hit)
    out["mean_difference_standard_errors"] = float((out["gpu"]["mean_energy"] - out["neal"]["mean_energy"]) / se_mean)
    return out


EQUAL_READS = 256           # reads on both sides of the equal-reads comparison (neal restatement: ~23 s on 16 cores)


def cpu_baseline(m, Qs, betas, edges, eq_gpu_states, eq_gpu_energies, best_known, perm, seconds_target=12.0):
    """The oracle on this host, on BOUNDED samples of the same workload:
      value        the restatement of dwave-neal (fp64 Ising, xorshift128+, dense couplings as neal's adjacency
                   lists would hold this QUBO, sequential sweeps) -- every (len/sweeps)-th beta of the schedule;
      same_chain   the chain the GPU kernel runs (CSR + uniform pair term, fp32, Philox: oracle 2b) on the same
                   threads -- the like-for-like algorithm (O(deg) per update instead of neal's O(n) per accepted flip);
      equal_reads  neal restatement, EQUAL_READS reads x the FULL schedule (seed SEED + 1), against GPU replicas
                   0 .. EQUAL_READS - 1 of seed SEED (`eq_gpu_*`: annealed once more for this, whatever --steps /
                   --warmup were): hit rates at the best energy known, means, best cuts (equal_reads_stats)."""
    from oracle import sa_oracle as so
    eu, ev = edges
    h, J, off = so.qubo_to_ising_dense(Qs.astype(np.float64))
    # the GPU box shares its host: one GPU's share is 16 cores (more threads only oversubscribe)
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), int(os.environ.get("MI_CPU_THREADS", "16")))
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
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
    # the GPU kernel's own chain on the CPU (the renumbered model the device runs)
    from scrna_seq_qannealing_clustering_amd import models
    rp, cc, vv = models.permute_csr(m.rowptr, m.col, m.val.astype(np.float32), perm)
    lin = m.lin.astype(np.float32)[perm]
    sc_reads, sc_sub = 8 * cores, betas[::10]
    t0 = time.perf_counter()
    sc_st, sc_en, sc_stats = so.sa_csr_rank1_philox(rp, cc, vv, lin, float(np.float32(m.c_pair)), sc_reads, sc_sub, SEED)
    t_sc = time.perf_counter() - t0
    # equal reads, equal sweeps
    R_eq = len(eq_gpu_energies)
    t0 = time.perf_counter()
    sp, _, _ = so.sa_ising_neal_dense(h, J, R_eq, betas, seed=SEED + 1, threads=cores)
    t_eq = time.perf_counter() - t0
    xn = ((sp + 1) // 2).astype(np.uint8)
    e_neal = m.energies(xn)
    e_gpu = np.asarray(eq_gpu_energies)
    known = min(float(best_known), float(e_neal.min()), float(e_gpu.min()))
    eq = equal_reads_stats(e_neal, e_gpu, known)
    eq["neal"]["best_cut_edges"] = int(so.cut_edges(eu, ev, xn[int(np.argmin(e_neal))][None, :])[0])
    eq["gpu"]["best_cut_edges"] = int(so.cut_edges(eu, ev, np.ascontiguousarray(eq_gpu_states[int(np.argmin(e_gpu))])[None, :])[0])
    eq.update({"reads": R_eq, "sweeps": int(len(betas)), "seconds": t_eq,
               "seeds": {"neal": SEED + 1, "gpu": SEED, "gpu_replicas": "0..%d" % (R_eq - 1)},
               "note": "same schedule, same number of reads and sweeps; both chains are Metropolis sweeps (the same "
                       "stationary distributions), so the two are statistically indistinguishable: hit rates at the best "
                       "known energy within their binomial bounds, means within their standard errors "
                       "(tests/test_gpu_sampler.py asserts 3 sigma / 4 s.e.); which best-of-%d is lower in one run is chance" % R_eq})
    return {
        "value": float(st[0]) / t, "unit": "spin-flip updates/s", "cores": cores, "kind": "port",
        "single_thread_value": single,
        "best_energy": float((en + off).min()),
        "sample": "oracle neal restatement (fp64 Ising, xorshift128+), same dense Q, %d reads x %d sweeps "
                  "(every %d-th beta of the 1000-sweep schedule), OpenMP over reads on %d threads; "
                  "real dwave-neal is not installable offline" % (reads, len(sub), max(1, len(betas) // sweeps), cores),
        "seconds": t,
        "same_chain": {"value": float(sc_stats[0]) / t_sc, "unit": "spin-flip updates/s", "cores": cores,
                       "best_energy": float(sc_en.min()), "seconds": t_sc,
                       "sample": "oracle 2b (the GPU kernel's chain: CSR + uniform pair, fp32, Philox), %d reads x %d sweeps "
                                 "(every 10th beta), OpenMP over reads" % (sc_reads, len(sc_sub))},
        "equal_reads": eq,
        "real_neal": real_neal_probe(Qs, betas),         # null: not importable on this host
    }


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def pmc_traffic(replicas, sweeps, launches, kernel):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same workload
    (profiles/r03_pmc_traffic.json, written by scripts/pmc_traffic.py on the GPU box: FETCH_SIZE doubled
    per the gfx950 note + WRITE_SIZE, averaged over the launches of one step).  None when no profile matches
    this launch shape and kernel."""
    rec = _profile_json("r03_pmc_traffic.json")
    if rec and (rec.get("replicas") == replicas and rec.get("sweeps") == sweeps and rec.get("launches") == launches
                and rec.get("kernel") == kernel):
        return float(rec["hbm_bytes_per_launch"])
    return None


def binding_resource(kernel, replicas, sweeps):
    """The counters of the committed SQ / TCC passes over this kernel at this launch shape
    (profiles/r03_k2_binding.json, scripts/pmc_k2.sh + scripts/k2_binding.py): L2 request bytes, LDS busy, VALU and
    per-wavefront issue.  None when the file was taken on another kernel or shape."""
    rec = _profile_json("r03_k2_binding.json")
    if rec and rec.get("kernel") == kernel and rec.get("replicas") == replicas and rec.get("sweeps") == sweeps:
        return {k: rec[k] for k in rec if k not in ("kernel", "replicas", "sweeps")}
    return None


def l2_request_bytes_per_launch(kernel, R, sweeps, n, D):
    """Bytes the anneal kernel's vector loads request from L2 in one launch: per slot of 64 variables the packed
    adjacency (D entries x 64 lanes x (4 B neighbour + 4 B value)) and the linear terms (256 B); one wavefront
    carries two replicas in the pair kernel, one otherwise.  Initial states / epilogue (once per launch) are left out."""
    slots = (n + 63) // 64
    waves = (R + 1) // 2 if "pair" in kernel else R
    return float(waves) * sweeps * slots * (D * 64 * 8 + 256)


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
        name = "dense_bqm"
    else:
        p = Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                              float(np.float32(m.c_pair)), device=rank_device, order="padded")
        name = "csr_rank1_bqm"
    with p:
        p.anneal(R, b, SEED)
        ms = p.kernel_ms()
        st, en, info = p.fetch()
        out[name] = {"kernel": p.kernel_name(), "replicas": R, "sweeps": S, "kernel_ms": ms,
                     "updates_per_s": R * S * n / (ms * 1e-3),
                     "best_energy": float(m.energies(st[int(np.argmin(en))][None, :])[0]),
                     "acceptance": info["accepted"] / info["proposals"]}
    pm = models.build_dqm_potts(graph, 8, 0.005)
    with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), n, 8,
                           lin_offset=pm.lin_offset, device=rank_device, order="padded") as p:
        b = models.make_beta_schedule(S, default_potts_beta_range(pm))
        p.anneal(R, b, SEED)
        ms = p.kernel_ms()
        lab, en, info = p.fetch()
        out["potts_dqm_k8"] = {"kernel": p.kernel_name(), "replicas": R, "sweeps": S, "kernel_ms": ms,
                               "updates_per_s": R * S * n / (ms * 1e-3), "best_energy": float(en.min()),
                               "acceptance": info["accepted"] / info["proposals"]}
    # K4: the replica-batched energy x^T Q x of the headline's final states on the matrix cores (the one
    # GEMM-shaped step of the path: 2 n^2 flop per state), checked against the exact fp64 path
    from scrna_seq_qannealing_clustering_amd.engine import energy_dense
    X = st.astype(np.uint8)
    energy_dense(Qs, X[:64], device=rank_device, path=2)                           # warm
    e_mfma, ms = energy_dense(Qs, X, device=rank_device, path=2, return_ms=True)
    e_exact = energy_dense(Qs, X[:256], device=rank_device, path=1)
    # the kernel multiplies only the 128 x 128 blocks (I, K >= I) of the symmetric Qs (an off-diagonal block counts
    # twice): `tflops` / `frac_of_mfma_peak` count the flops the matrix cores EXECUTE; `dense_equivalent_tflops` is the
    # 2 n^2 flop per state of the plain contraction over the same time
    T, rt = (n + 127) // 128, (len(X) + 127) // 128
    executed = T * (T + 1) // 2 * rt * 2.0 * 128 ** 3
    out["energy_mfma"] = {"kernel": "k_energy_dense_mfma", "states": int(len(X)), "kernel_ms": ms,
                          "tflops": executed / (ms * 1e-3) / 1e12, "peak_tflops_f32_input_mfma": 157.3,
                          "frac_of_mfma_peak": executed / (ms * 1e-3) / 157.3e12,
                          "dense_equivalent_tflops": 2.0 * n * n * len(X) / (ms * 1e-3) / 1e12,
                          "timed": "transpose of the states + MFMA kernel (one call)",
                          "max_rel_diff_vs_exact_fp64": float(np.max(np.abs(e_mfma[:256] - e_exact) /
                                                                     np.maximum(1.0, np.abs(e_exact))))}
    out["reads_500"] = reads_500(m, graph, rank_device)
    out["bisection_workflow"] = bisection_workflow(graph, rank_device)
    if os.environ.get("MI_BENCH_SKIP_50K") != "1":
        out["dense_xl_50k"] = dense_xl_50k(rank_device)
    return out


def reads_500(m, graph, rank_device, reads=500, sweeps=1000):
    """The reference's own call shape: `num_reads = 500` (BQM_clustering.py:52; 5000 at :240) x 1000 sweeps, on the whole
    graph and on a subgraph of the size its recursive bisection reaches (:113-203; here the cells of one planted
    cluster, n ~ 340).  500 reads are far fewer wavefronts than the chip has SIMDs (1024): the kernel time is one
    wavefront's latency through the sweeps, not throughput."""
    from scrna_seq_qannealing_clustering_amd import graphs, models
    from scrna_seq_qannealing_clustering_amd.engine import Problem, layout_block_for
    out = {"reads": reads, "sweeps": sweeps, "simds": 1024}
    keep = np.flatnonzero(graph.truth == 3)
    new_id = -np.ones(len(graph.truth), dtype=np.int64)
    new_id[keep] = np.arange(len(keep))
    eu, ev, w = graph._eu, graph._ev, graph._w
    inside = (new_id[eu] >= 0) & (new_id[ev] >= 0)
    sub = graphs.EdgeListGraph([str(i) for i in range(len(keep))], new_id[eu[inside]].astype(np.int32),
                               new_id[ev[inside]].astype(np.int32), w[inside])
    for name, mm in (("whole_graph", m), ("bisection_subgraph", models.build_bqm_qubo(sub, 0.05, k=8))):
        nn = mm.num_variables
        b = models.make_beta_schedule(sweeps, models.default_beta_range(mm))
        with Problem.csr_rank1(mm.rowptr, mm.col, mm.val.astype(np.float32), mm.lin.astype(np.float32),
                               float(np.float32(mm.c_pair)), device=rank_device, order="padded",
                               energy_model=(mm.val, mm.lin, mm.c_pair),
                               block=layout_block_for(nn, reads, int(np.diff(mm.rowptr).max()))) as p:   # as sampler.py
            p.anneal(reads, b, SEED)                                                   # warm (first launch of this shape)
            p.anneal(reads, b, SEED)
            ms, kname, slots = p.kernel_ms(), p.kernel_name(), p.n_dev // 64
        waves = reads_wavefronts(kname, reads)
        out[name] = {"n": nn, "device_slots": slots, "kernel": kname, "kernel_ms": ms,
                     "updates_per_s": reads * sweeps * nn / (ms * 1e-3), "wavefronts": waves,
                     "wavefronts_per_simd": waves / 1024.0}
    return out


def bisection_workflow(graph, rank_device, levels=3):
    """The reference's own workflow on the bench graph: `clustering_bqm` (BQM_clustering.py:25-204) with
    terminate_on="iter_limit" -- a recursive bisection, 2^(levels + 1) - 1 = 15 sampler calls of 500 reads x 1000 sweeps on
    shrinking subgraphs, through the drop-in sampler (model build, seats, upload, anneal, fp64 energies, SampleSet per
    call; the second half of every split enqueued while the first half's subtree is worked through).  Wall time of the
    whole call, second run (the first one pays the one-time initialisations)."""
    from scrna_seq_qannealing_clustering_amd import MI355XSampler
    from scrna_seq_qannealing_clustering_amd.clustering import clustering_bqm
    from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges
    G = graph_from_edges(list(graph.nodes), graph._eu, graph._ev, graph._w)
    smp = MI355XSampler(device=rank_device)
    calls = []
    inner, inner_async = smp.sample_qubo, smp.sample_qubo_async

    def counted(Q, **kw):
        calls.append(1)
        return inner(Q, **kw)

    def counted_async(Q, **kw):
        calls.append(1)
        return inner_async(Q, **kw)
    smp.sample_qubo, smp.sample_qubo_async = counted, counted_async
    walls = []
    for _ in range(3):
        calls.clear()
        t0 = time.perf_counter()
        clustering_bqm(G, 0, None, "mi355x", 0.05, 0, "iter_limit", 5, levels, 0, sampler=smp, sampler_kwargs={"seed": SEED})
        walls.append(time.perf_counter() - t0)
    return {"sampler_calls": len(calls), "reads": 500, "sweeps": 1000, "levels": levels + 1, "wall_s": min(walls[1:]),
            "first_run_wall_s": walls[0], "what": "clustering_bqm(terminate_on='iter_limit') on the bench graph through MI355XSampler"}


def reads_wavefronts(kernel, reads):
    """Wavefronts a run of `reads` replicas puts on the chip: the pair kernel carries two replicas per wavefront, the
    split kernel (few reads: the idle SIMDs take a share of every replica's variables) several wavefronts per replica."""
    import re
    if "pair" in kernel:
        return (reads + 1) // 2
    mt = re.search(r"split<\d+,\s*(\d+)>", kernel)
    return reads * int(mt.group(1)) if mt else reads


def dense_xl_50k(rank_device, n=50000, replicas=1024, sweeps=4):
    """BASELINE config 4 in its literal form: a synthetic 50 000-cell SNN graph built on the GPU (snn.build_snn), the
    clustering_bqm QUBO as a dense fp32 matrix (10.6 GB resident in HBM), one GPU's share of the replicas (8192 / 8), the
    first `sweeps` sweeps of the 1000-step schedule (the hot end: > 99 % of the proposals are accepted).  Kernel K1g:
    all replicas walk the rows together, 64 rows per DIAG, the row updates of a group of 512 rows as one GEMM-shaped
    pass over F[column][replica] on the matrix cores.  `mfma_roofline` = its flop (2 x rows x columns x replicas per
    pass, the field initialisation pass included) against the f32-input MFMA peak; `field_traffic` = the
    read-modify-write of F per group of 512 rows; `hbm_side` = what rocprofv3 FETCH_SIZE x 2 saw on this shape
    (profiles/r02_dense50k.json).  K1x (a workgroup per replica, one Q row per accepted flip) is what batches below
    256 replicas run: 3.5e7 updates/s on this model."""
    from scrna_seq_qannealing_clustering_amd import models, snn
    from scrna_seq_qannealing_clustering_amd.engine import Problem
    rng = np.random.RandomState(1)
    centers = rng.normal(scale=4.0, size=(30, 15))
    X = (centers[rng.randint(0, 30, size=n)] + rng.normal(size=(n, 15))).astype(np.float32)
    t0 = time.perf_counter()
    g = snn.build_snn(X, 5, 0.0, 15, device=rank_device)
    m = models.build_bqm_qubo(g.to_graph(), 0.05)
    Qs = np.full((n, n), np.float32(m.c_pair / 2.0), dtype=np.float32)          # Qs_ij = (c_pair + S_ij) / 2
    rows = np.repeat(np.arange(n), np.diff(m.rowptr))
    Qs[rows, m.col] += (m.val / 2.0).astype(np.float32)
    Qs[np.arange(n), np.arange(n)] = m.lin.astype(np.float32)
    t_build = time.perf_counter() - t0
    betas = models.make_beta_schedule(1000, models.default_beta_range(m))[:sweeps]
    t0 = time.perf_counter()
    with Problem.dense(Qs, device=rank_device) as p:
        t_upload = time.perf_counter() - t0
        del Qs
        p.anneal(replicas, betas, SEED)
        ms = p.kernel_ms()
        kname = p.kernel_name()
        _, en, info = p.fetch(states=False)
    nblocks, ncols, rp = (n + 63) // 64, ((n + 255) // 256) * 256, ((replicas + 255) // 256) * 256
    passes = sweeps + 1                                                           # + the field initialisation pass
    flop = 2.0 * (64 * nblocks) * ncols * rp * passes
    f_bytes = 8.0 * ncols * rp * ((nblocks + 7) // 8) * passes
    rec = _profile_json("r02_dense50k.json")
    hbm = None
    if rec and rec.get("replicas") == replicas and rec.get("sweeps") == sweeps and rec.get("n") == n:
        gbps = rec["pmc"]["fabric_read_bytes_x2"] / (ms * 1e-3) / 1e9
        hbm = {"bytes": rec["pmc"]["fabric_read_bytes_x2"], "GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
               "source": "profiles/r02_dense50k.json: rocprofv3 --pmc FETCH_SIZE x 2 on this shape (scripts/pmc_dense50k.sh), "
                         "bytes of that run / this run's kernel time"}
    return {"kernel": kname, "n": n, "replicas": replicas, "sweeps": sweeps, "kernel_ms": ms,
            "updates_per_s": replicas * sweeps * n / (ms * 1e-3), "acceptance": info["accepted"] / info["proposals"],
            "dense_Q_bytes": 4 * n * (((n + 4095) // 4096) * 4096), "graph_edges": int(len(m.col) // 2),
            "mfma_roofline": {"bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                              "frac": flop / (ms * 1e-3) / 157.3e12,
                              "model": "2 x (64 x blocks) rows x padded columns x padded replicas per pass, all rows, "
                                       "accepted or not (v_mfma_f32_16x16x4_f32 chained in row order: bit-exact)"},
            "field_traffic": {"GBps": f_bytes / (ms * 1e-3) / 1e9,
                              "note": "read + write of the cached fields F[column][replica] once per group of 512 rows"},
            "hbm_side": hbm,
            "host_build_s": t_build, "upload_s": t_upload, "best_energy": float(en.min())}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU)
    ap.add_argument("--sweeps", type=int, default=SWEEPS)
    ap.add_argument("--kernel", choices=("csr", "dense"), default="csr",
                    help="timed kernel: csr = K2 (what the sampler runs for this model), dense = K1w")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_launch_command(args, argv, port=None):
    """`python bench.py --gpus N` typed WITHOUT a launcher: the command that starts the N ranks (one per GPU) as
    CHILD processes -- torch.distributed.run on this same file with the same arguments.  None when this process
    already is a rank (WORLD_SIZE set by a launcher) or N = 1."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.abspath(__file__)] + list(argv)


def launch_ranks(cmd):
    """Runs the rank processes as children of this one (which has not touched the GPU: nothing above imports torch or
    loads the HIP library -- a process that has may not be replaced or re-exec'd on this pool), relays rank 0's JSON
    line and returns the children's exit code."""
    import subprocess
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse_args()
    cmd = rank_launch_command(args, sys.argv[1:])
    if cmd is not None:
        raise SystemExit(launch_ranks(cmd))

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
        bytes_per_update = 4.0 * n
        layout = "dense fp32 Q 27.8 MB"
    else:
        prob = Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                                 float(np.float32(m.c_pair)), device=local, order="padded",
                                 energy_model=(m.val, m.lin, m.c_pair))               # exactly as sampler.py creates it
        bytes_per_update = 8.0 * float(np.diff(m.rowptr).mean()) + 8.0      # deg_i*(4+4) + 8
        layout = "CSR (cut term) + uniform pair term, %.1f neighbours per cell on average" % float(np.diff(m.rowptr).mean())

    upload_ms = (time.perf_counter() - t_up0) * 1e3                # host arrays -> HBM (incl. the sweep ordering)

    def step(i):
        prob.anneal(R, betas, SEED + i, replica_offset=rank * R)
        idx, e, key, state = prob.best()                           # K5 on device (exact fp64 argmin); waits for the anneal
        return D.global_best_f64(e, rank * R + idx, state, num_reads=world * R)    # C1 + C2 (identity at N=1)

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

    kernel_name = prob.kernel_name()                               # what the library launched for this model and R
    t_f0 = time.perf_counter()
    states, en, info = prob.fetch()
    fetch_ms = (time.perf_counter() - t_f0) * 1e3                  # R x n state bytes + R energies -> host
    eq_states = eq_en = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the GPU side of cpu_baseline.equal_reads: replicas 0 .. EQUAL_READS - 1 of seed SEED over the full schedule,
        # whatever --steps / --warmup / --replicas were (a replica's chain depends on (seed, global id) only)
        prob.anneal(EQUAL_READS, betas, SEED)
        eq_states, eq_en, _ = prob.fetch()
    updates_per_step = world * R * len(betas) * n
    value = updates_per_step * args.steps / elapsed
    k_ms = float(np.mean(kernel_ms))                               # all launches of one step
    launch_ms = k_ms / launches                                    # = rocprofv3's average duration of the kernel
    sweeps_per_launch = len(betas) / launches
    alg_bytes = bytes_per_update * R * sweeps_per_launch * n       # per launch (one GPU): SURVEY 8d's per-update model
    effective = alg_bytes / (launch_ms * 1e-3) / 1e9
    accept = info["accepted"] / info["proposals"]
    if args.kernel == "dense":
        # Two kernels serve a cooling run: K1m the hot chunks (fields as MFMA accumulators: every row
        # costs 2 * 16 * n_pad flop per workgroup, accepted or not), K1w the rest (rows leave the LDS ring only for accepted
        # flips).  The roofline object prices the MFMA work of the K1m chunks against the f32-input MFMA peak over the
        # WHOLE step time (K1w's share of the time included): a lower bound of the fraction.
        ds = prob.debug_stats()
        mfma_chunks = int(ds[15])
        mfma_sweeps = min(len(betas), 8 + 32 * max(mfma_chunks - 1, 0)) if mfma_chunks else 0
        n_pad = ((n + 255) // 256) * 256
        flop = 2.0 * 16 * n_pad * (((n + 15) // 16) * 16) * ((R + 15) // 16) * mfma_sweeps
        roofline = {"bound": "mfma", "achieved": flop / (k_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                    "frac": flop / (k_ms * 1e-3) / 157.3e12, "traffic": None,
                    "mfma_sweeps": mfma_sweeps, "mfma_chunks": mfma_chunks, "workgroup_kernel_chunks": int(ds[14]),
                    "model": "flop of the sweeps K1m served (2 x 16 x n_pad x rows per workgroup and sweep) / the time of the whole "
                             "step, K1w's sweeps included; K1m alone: 0.49 of the peak (profiles/r02_k1m_binding.json)"}
    else:
        ell_width = 16 if np.diff(m.rowptr).max() <= 16 else (32 if np.diff(m.rowptr).max() <= 32 else 64)
        l2_bytes = l2_request_bytes_per_launch(kernel_name, R, sweeps_per_launch, prob.n_dev, ell_width)
        l2_gbps = l2_bytes / (launch_ms * 1e-3) / 1e9
        roofline = {"bound": "l2", "achieved": l2_gbps, "peak": L2_PEAK_GBPS, "unit": "GB/s", "frac": l2_gbps / L2_PEAK_GBPS,
                    "traffic": pmc_traffic(R, len(betas), launches, kernel_name),
                    "l2_request_bytes_per_launch": l2_bytes,
                    "model": "bytes the kernel's loads request from L2: per 64-variable slot the packed adjacency "
                             "(%d x 64 x 8 B) + linear terms (256 B), once per wavefront (two replicas per wavefront in the "
                             "pair kernel), every slot of every sweep; peak = L2 aggregate (MI355X_MICROARCH.md)" % ell_width,
                    "binding_resource": binding_resource(kernel_name, R, len(betas)),
                    "what_binds": "this path, nearly: with the thresholds computed by a second wavefront per workgroup (round 3) the "
                                  "sweeping wavefronts of a CU fetch 8 x 8.25 KB of adjacency per slot, ~57 B/clk of the 64 B/clk a CU's "
                                  "vector-memory path delivers at the clock the chip holds under this load (1.97 GHz in the counter passes); "
                                  "a timing-only build without adjacency traffic runs 25.2 -> 21.3 ms, without the threshold wavefront's "
                                  "arithmetic 24.0, without accept rounds 24.9, with conflict-free LDS gathers 25.7 "
                                  "(profiles/r03_d_k2p_tw_timing_experiments.txt); LDS 74 % busy, wavefronts issuing 35 % / waiting for an "
                                  "issue slot 25 % / parked on s_waitcnt 40 % of their cycles (profiles/r03_k2_binding.json)"}
    roofline.update({"kernel": kernel_name, "kernel_ms": launch_ms, "launches_per_step": launches,
                     "sweeps_per_launch": sweeps_per_launch, "kernel_ms_per_step": k_ms, "acceptance": accept})
    best_state = best[3]
    cut_edges = int(np.sum(best_state[eu] != best_state[ev]))
    out = {
        "metric": "spin-flip updates/sec (node) + best QUBO energy vs neal, PBMC3k SNN",
        "value": value, "unit": "spin-flip updates/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PBMC3k-sized synthetic SNN (n=2638, k=5, dim=15, trim 15; 9 overlapping clusters, one connected "
                               "component), clustering_bqm QUBO "
                               "(gamma_factor 0.05, k 8), %s, resident in HBM, %d replicas/GPU x %d sweeps, "
                               "geometric beta, seed 1234" % (layout, R, len(betas)),
                   "kernel": args.kernel,
                   "n": n, "device_slots": prob.n_dev // 64 if args.kernel == "csr" else None, "replicas_per_gpu": R, "sweeps": int(len(betas)), "parallelism": "replicas sharded x%d" % world},
        "roofline": roofline,
        "effective": {"algorithmic_bytes_per_update": bytes_per_update, "algorithmic_bytes_per_launch": alg_bytes,
                      "GBps": effective, "note": "SURVEY.md 8d byte model x updates / time: an effective-bandwidth figure "
                      "(the model is cache resident and shared by the replicas), not a fraction of any peak"},
        "best_energy": float(m.energies(best_state[None, :])[0]),
        "mean_energy": float(np.mean(en)),
        "replicas_at_best_energy": int(np.sum(en <= en.min() + 1e-6 * abs(en.min()))),
        "best_energy_device": float(best[0]), "best_replica_global_id": int(best[1]),
        "host_buffers": {"model_upload_ms": upload_ms, "results_fetch_ms": fetch_ms,
                         "note": "outside the timed region: `value` is measured with the model resident in HBM"},
        "best_cut_edges": cut_edges,
        "energy_lower_bound": float(-m.info["gamma"] * n * n / 4),
    }
    if rank == 0:
        if world == 1:
            out["other_kernels"] = other_kernels(m, Qs, betas, graph, local, args.kernel)
        if not args.no_cpu_baseline and world == 1:
            # the order the device sweeps the caller's variables in (holes of the padded layout take no proposal)
            perm = prob.perm if prob.perm is not None else (np.argsort(prob._inv) if prob._inv is not None else np.arange(n))
            out["cpu_baseline"] = cpu_baseline(m, Qs, betas, (eu, ev), eq_states, eq_en, float(np.min(en)), perm)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    prob.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
