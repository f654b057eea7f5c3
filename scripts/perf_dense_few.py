#!/usr/bin/env python3
"""Dense (arbitrary-QUBO) kernels at FEW replicas: which variant should serve a run of 64 ... 1024 reads?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scrna_seq_qannealing_clustering_amd import models
from scrna_seq_qannealing_clustering_amd.engine import Problem
m, Qs, betas, _, graph = bench.build_workload()
n = m.num_variables
S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = models.make_beta_schedule(S, models.default_beta_range(m))
with Problem.dense(Qs) as p:
    for R in (64, 128, 256, 500, 1024, 2048):
        ref = None
        for variant in (0, 1, 2):
            p.set_option("variant", variant)
            for rep in range(2):
                p.anneal(R, b, 1234)
                ms = p.kernel_ms()
                st, en, info = p.fetch()
            if ref is None:
                ref = st.copy()
            print("R = %5d  variant %d  %9.2f ms  %.3e upd/s  launches %d  %s  %s" % (
                R, variant, ms, R * S * n / ms * 1e3, p.launch_count(), p.kernel_name(), "same" if np.array_equal(ref, st) else "DIFFERENT"), flush=True)
