#!/usr/bin/env python3
"""Interleaved A/B timing of kernel options on the bench workload (development helper).
usage: perf_ab.py [--sweeps S] [--replicas R] [--rounds K] opt=val,opt=val ...   (each arg = one arm)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scrna_seq_qannealing_clustering_amd import models  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=100)
ap.add_argument("--replicas", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--check", action="store_true")
ap.add_argument("--beta-range", default=None, help="lo,hi (default: the model's neal-style range)")
ap.add_argument("arms", nargs="*", default=["pace=0", "pace=1"])
a = ap.parse_args()
m, Qs, betas_full, _, _graph = bench.build_workload()
brange = tuple(float(x) for x in a.beta_range.split(',')) if a.beta_range else models.default_beta_range(m)
betas = models.make_beta_schedule(a.sweeps, brange)
n = Qs.shape[0]
ref = None
with Problem.dense(Qs) as p:
    for rnd in range(a.rounds):
        for arm in a.arms:
            for kv in arm.split(","):
                if kv:
                    k, v = kv.split("=")
                    p.set_option(k, int(v))
            p.anneal(a.replicas, betas, 1234)
            ms = p.kernel_ms()
            st, en, info = p.fetch()
            pw = p.debug_pace()
            print('   pace: started %d disabled %d timeouts %d pop %s arrivals %s' % (pw[0], pw[1], pw[2], pw[32::32][:8].tolist(), pw[33::32][:8].tolist()))
            if a.check:
                if ref is None:
                    ref = st.copy()
                assert np.array_equal(ref, st), "arm %s changed the results" % arm
            print("round %d %-24s %9.2f ms  %.3e upd/s  rows %.1f GB/s  acc %.3f  minE %.3f" % (
                rnd, arm, ms, a.replicas * a.sweeps * n / ms * 1e3,
                info["accepted"] * 44 * 256 / ms / 1e6, info["accepted"] / info["proposals"], en.min()),
                flush=True)
