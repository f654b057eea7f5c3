#!/usr/bin/env python3
"""K3f against k_anneal_potts on the bench graph (development helper): same chain, so --check compares the labels.
usage: perf_k3_fast.py [K] [replicas] [sweeps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scrna_seq_qannealing_clustering_amd import models
from scrna_seq_qannealing_clustering_amd.engine import Problem
from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m, Qs, betas, _, graph = bench.build_workload()
pm = models.build_dqm_potts(graph, K, 0.005)
n = pm.num_variables
b = models.make_beta_schedule(S, default_potts_beta_range(pm))
with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), n, K,
                       lin_offset=pm.lin_offset, order="padded") as p:
    ref = None
    for rnd in range(3):
        for fast in (2, 0):
            p.set_option("k3_fast", fast)
            p.anneal(R, b, 1234)
            ms = p.kernel_ms()
            lab, en, info = p.fetch()
            if ref is None:
                ref = lab.copy()
            same = np.array_equal(ref, lab)
            print("K=%d k3_fast=%d  %8.2f ms  %.3e upd/s  acc %.3f  minE %.4f  %s  %s" % (
                K, fast, ms, R * S * n / ms * 1e3, info["accepted"] / info["proposals"], en.min(), p.kernel_name(),
                "same labels" if same else "LABELS DIFFER"), flush=True)
