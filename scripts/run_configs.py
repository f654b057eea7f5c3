#!/usr/bin/env python3
"""Runs the BASELINE.json configurations other than the bench headline at the share ONE MI355X gets of them
and writes profiles/r01_configs.json (updates/s, acceptance, best energies, graph-construction times).
The graphs are built on the GPU (snn.build_snn) from synthetic point clouds: the reference ships no PBMC or
kidney data.
  config 3  PBMC3k-sized SNN (n=2638), DQM K=8 (K3), 4096 replicas x 1000 sweeps
  config 4  synthetic SNN n=50000, 8192 replicas over 8 GPUs -> 1024 replicas here, clustering_bqm on K2
            (the model in CSR form: no 10 GB dense Q is ever needed), 200 sweeps timed
  config 5  kidney-sized SNN n=10605 (k=10, dim=30, ord=15), DQM K=15 with parallel tempering:
            8 rungs x 128 chains per GPU, rounds of 10 sweeps
usage: run_configs.py [--quick]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scrna_seq_qannealing_clustering_amd import models, snn, tempering  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402
from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range  # noqa: E402


def cloud(n, dim, clusters, seed):
    rng = np.random.RandomState(seed)
    centers = rng.normal(scale=4.0, size=(clusters, dim))
    lab = rng.randint(0, clusters, size=n)
    return (centers[lab] + rng.normal(size=(n, dim))).astype(np.float32), lab


def purity(labels, truth):
    """fraction of cells whose cluster's majority planted label is their own"""
    tot = 0
    for c in np.unique(labels):
        tot += np.bincount(truth[labels == c]).max()
    return tot / len(labels)


ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
a = ap.parse_args()
out = {}
f32 = np.float32

# ---- config 3 -------------------------------------------------------------------------------------------
X, truth = cloud(2638, 15, 9, 0)
g = snn.build_snn(X, 5, 0.0, 15)
pm = models.build_dqm_potts(g.to_graph(), 8, 0.005)
S = 100 if a.quick else 1000
with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(f32), float(f32(pm.c_pair)), 2638, 8, lin_offset=pm.lin_offset,
                       order="padded") as p:
    p.anneal(4096, models.make_beta_schedule(S, default_potts_beta_range(pm)), 1234)
    ms = p.kernel_ms()
    p_kernel3 = p.kernel_name()
    lab, en, info = p.fetch()
out["config3_dqm_k8_n2638"] = {
    "kernel": p_kernel3, "replicas": 4096, "sweeps": S, "kernel_ms": ms,
    "updates_per_s": 4096 * S * 2638 / (ms * 1e-3), "acceptance": info["accepted"] / info["proposals"],
    "best_energy": float(en.min()), "purity_of_best": purity(lab[int(np.argmin(en))], truth), "snn_build_ms": g.timing}
print(json.dumps(out["config3_dqm_k8_n2638"]), flush=True)

# ---- config 4 -------------------------------------------------------------------------------------------
n4 = 50000
X, truth = cloud(n4, 15, 30, 1)
t0 = time.perf_counter()
g = snn.build_snn(X, 5, 0.0, 15)
t_build = time.perf_counter() - t0
m = models.build_bqm_qubo(g.to_graph(), 0.05)
S = 20 if a.quick else 200
with Problem.csr_rank1(m.rowptr, m.col, m.val.astype(f32), m.lin.astype(f32), float(f32(m.c_pair)), order="padded") as p:
    p.anneal(1024, models.make_beta_schedule(S, models.default_beta_range(m)), 1234)
    ms = p.kernel_ms()
    p_kernel4 = p.kernel_name()
    st, en, info = p.fetch()
best = st[int(np.argmin(en))]
out["config4_bqm_n50000"] = {
    "kernel": p_kernel4, "replicas": 1024, "sweeps": S, "kernel_ms": ms,
    "updates_per_s": 1024 * S * n4 / (ms * 1e-3), "acceptance": info["accepted"] / info["proposals"],
    "best_energy": float(m.energies(best[None, :])[0]), "best_split": [int(best.sum()), int(n4 - best.sum())],
    "edges": int(len(g.col) // 2), "snn_build_ms": g.timing, "snn_build_wall_s": t_build,
    "dense_Q_bytes_not_materialised": 4 * n4 * n4,
    "naive_hbm_ceiling_updates_per_s": 8.0e12 / (4.0 * n4)}
print(json.dumps(out["config4_bqm_n50000"]), flush=True)

# ---- config 5 -------------------------------------------------------------------------------------------
n5 = 10605
X, truth = cloud(n5, 30, 15, 2)
g = snn.build_snn(X, 10, 0.0, 15)
pm = models.build_dqm_potts(g.to_graph(), 15, 0.005)
rungs, chains, rounds, sweeps_per_round = 8, 128, (4 if a.quick else 40), 10
lo, hi = default_potts_beta_range(pm)
ladder = np.geomspace(lo * 20, hi / 20, rungs)
t0 = time.perf_counter()
prob = Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(f32), float(f32(pm.c_pair)), n5, 15, lin_offset=pm.lin_offset,
                         order="padded")
kernel_ms = [0.0]


class TimedEngine(tempering.ProblemEngine):
    def round(self, *args, **kw):
        super().round(*args, **kw)
        kernel_ms[0] += self.problem.kernel_ms()


res = tempering.parallel_tempering(TimedEngine(prob, 1234), ladder, chains, rounds, sweeps_per_round, 1234)
wall_timed = time.perf_counter() - t0
# the same run as the product runs it: exchange on the device, no per-round read of anything (history off) --
# the wall time of the whole tempering against the sum of its anneal kernels
t0 = time.perf_counter()
res2 = tempering.parallel_tempering(tempering.ProblemEngine(prob, 1234), ladder, chains, rounds, sweeps_per_round, 1234,
                                    history=False)
wall = time.perf_counter() - t0
assert np.array_equal(res2["rung"], res["rung"]) and np.array_equal(res2["energies"], res["energies"])
lab = res["local_states"]
pt_kernel = prob.kernel_name()
prob.close()
out["config5_dqm_k15_n10605_tempering"] = {
    "kernel": pt_kernel + " + k_pt_exchange", "wall_s_with_per_round_timing_and_history": wall_timed,
    "wall_over_kernel": wall / (kernel_ms[0] * 1e-3), "rungs": rungs, "chains_per_rung": chains, "rounds": rounds,
    "sweeps_per_round": sweeps_per_round, "kernel_ms": kernel_ms[0], "wall_s": wall,
    "updates_per_s": rungs * chains * rounds * sweeps_per_round * n5 / (kernel_ms[0] * 1e-3),
    "best_energy": res["best_energy"], "purity_of_best": purity(lab[res["best_replica"]], truth),
    "swap_rate": res["swap_rate"], "best_energy_by_round": res["history"][:: max(1, rounds // 8)],
    "snn_build_ms": g.timing}
print(json.dumps(out["config5_dqm_k15_n10605_tempering"]), flush=True)

for folder in ("profiles", "gpurun_out"):            # gpurun_out/ is what travels back from the GPU box
    os.makedirs(os.path.join(ROOT, folder), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, folder, "r03_configs.json"), "w"), indent=1)
