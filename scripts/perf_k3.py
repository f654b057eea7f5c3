#!/usr/bin/env python3
"""Timing / phase counters of K3 (Potts) on the bench graph, K = 8 (development helper)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scrna_seq_qannealing_clustering_amd import models
from scrna_seq_qannealing_clustering_amd.engine import Problem
from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
m, Qs, betas, _, graph = bench.build_workload()
pm = models.build_dqm_potts(graph, 8, 0.005)
R, S, n = int(os.environ.get("K3_R", "4096")), 200, 2638
b = models.make_beta_schedule(S, default_potts_beta_range(pm))
for order in (("slots",) if os.environ.get("K3_ONLY_SLOTS") else ("slots", None)):
    with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), n, 8,
                           lin_offset=pm.lin_offset, order=order) as p:
        for _ in range(2):
            p.anneal(R, b, 1234)
            ms = p.kernel_ms()
            lab, en, info = p.fetch()
            ds = p.debug_stats()
            print("order=%s  %.2f ms  %.3e upd/s  acc %.3f" % (order, ms, R * S * n / ms * 1e3, info["accepted"] / info["proposals"]))
            if ds[8:13].any():
                print("   cycles/wave/sweep: pre %.0f loop %.0f field-sum %.0f slot-top %.0f" % (
                    ds[8] / R / S, ds[9] / R / S, ds[11] / R / S, ds[12] / R / S))
