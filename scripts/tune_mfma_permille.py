#!/usr/bin/env python3
"""The dense bench run (4096 replicas x 1000 sweeps, default schedule) against the hand-over threshold between the two
dense kernels (`mfma_permille`: a chunk that accepted at least this share hands the next one to K1m)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

m, Qs, betas, _, _graph = bench.build_workload()
vals = [int(v) for v in sys.argv[1:]] or [300, 400, 500, 600, 700]
with Problem.dense(Qs) as p:
    for rnd in range(2):
        for v in vals:
            p.set_option("mfma_permille", v)
            p.anneal(4096, betas, 1234)
            ms = p.kernel_ms()
            _, en, info = p.fetch(states=False)
            ds = p.debug_stats()
            print("mfma_permille %4d  %8.2f ms  chunks K1w %d / K1m %d  best %.4f  acc %.3f" % (
                v, ms, ds[14], ds[15], en.min(), info["accepted"] / info["proposals"]), flush=True)
