#!/usr/bin/env python3
"""BASELINE config 4 in its literal form: n = 50000 synthetic SNN, the clustering_bqm QUBO as a DENSE fp32 Q
(10 GB) resident in HBM, annealed by K1x (one workgroup per replica, Q rows streamed from HBM on accepted
flips).  Prints one JSON line with updates/s and the algorithmic HBM rate (4 n_pad bytes per ACCEPTED flip).
The same model on K2 (CSR form) is in profiles/r01_configs.json.
usage: run_dense50k.py [--n N] [--replicas R] [--sweeps S]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import models, snn  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--replicas", type=int, default=256)
ap.add_argument("--sweeps", type=int, default=2)
ap.add_argument("--start", type=int, default=450, help="first beta of the 1000-step schedule to use")
ap.add_argument("--stride", type=int, default=1, help="take every stride-th beta (a short run over the whole range)")
ap.add_argument("--cold", type=int, default=-1, help="xl_cold_permille option (-1: the default)")
ap.add_argument("--chain", type=int, default=0, help="xl_chain option: 0 auto, 1 DIAG + small pass per block, 2 fused chain kernel")
ap.add_argument("--batched", type=int, default=0, help="xl_batched option: 0 auto, 1 K1g, 2 K1x")
a = ap.parse_args()
n = a.n
rng = np.random.RandomState(1)
centers = rng.normal(scale=4.0, size=(30, 15))
X = (centers[rng.randint(0, 30, size=n)] + rng.normal(size=(n, 15))).astype(np.float32)
g = snn.build_snn(X, 5, 0.0, 15)
m = models.build_bqm_qubo(g.to_graph(), 0.05)
t0 = time.perf_counter()
Qs = np.full((n, n), np.float32(m.c_pair / 2.0), dtype=np.float32)          # Qs_ij = (c_pair + S_ij) / 2
rows = np.repeat(np.arange(n), np.diff(m.rowptr))
Qs[rows, m.col] += (m.val / 2.0).astype(np.float32)
Qs[np.arange(n), np.arange(n)] = m.lin.astype(np.float32)
t_build = time.perf_counter() - t0
betas = models.make_beta_schedule(1000, models.default_beta_range(m))[a.start::a.stride][:a.sweeps]   # temperatures from the middle of the schedule
t0 = time.perf_counter()
with Problem.dense(Qs) as p:
    t_upload = time.perf_counter() - t0
    p.set_option("xl_batched", a.batched)
    p.set_option("xl_chain", a.chain)
    if a.cold >= 0:
        p.set_option("xl_cold_permille", a.cold)
    p.anneal(a.replicas, betas, 1234)
    ms = p.kernel_ms()
    p_kernel_name = p.kernel_name()
    st, en, info = p.fetch()
n_pad = ((n + 4095) // 4096) * 4096
init_rows = int(st.shape[0] * n / 2)                       # field initialisation streams ~n/2 rows per replica
out = {"kernel": p_kernel_name, "n": n, "replicas": a.replicas, "sweeps": a.sweeps,
       "kernel_ms": ms, "updates_per_s": a.replicas * a.sweeps * n / (ms * 1e-3),
       "acceptance": info["accepted"] / info["proposals"],
       "rows_streamed": info["accepted"] + init_rows,
       "algorithmic_GBps": (info["accepted"] + init_rows) * 4.0 * n_pad / (ms * 1e-3) / 1e9,
       "naive_hbm_ceiling_updates_per_s": 8.0e12 / (4.0 * n), "dense_Q_bytes": 4 * n * n_pad,
       "host_build_s": t_build, "upload_s": t_upload, "best_energy": float(en.min())}
print(json.dumps(out))
