#!/usr/bin/env python3
"""Where the dense kernels spend a 1000-sweep run: the bench schedule cut into segments that continue each
other (states and random stream), kernel time and accepted flips per segment (development helper).
usage: perf_dense_segments.py [--segments 20] [--replicas 4096] [opt=val,...]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--segments", type=int, default=20)
ap.add_argument("--replicas", type=int, default=4096)
ap.add_argument("opts", nargs="?", default="")
a = ap.parse_args()
m, Qs, betas, _, _graph = bench.build_workload()
n = Qs.shape[0]
S = len(betas)
seg = S // a.segments
with Problem.dense(Qs) as p:
    for kv in a.opts.split(","):
        if kv:
            k, v = kv.split("=")
            p.set_option(k, int(v))
    for rep in range(2):
        total = 0.0
        for i in range(a.segments):
            p.anneal(a.replicas, betas[i * seg:(i + 1) * seg], 1234, sweep_offset=i * seg, continue_run=i > 0)
            ms = p.kernel_ms()
            _, _, info = p.fetch(states=False)
            total += ms
            if rep == 1:
                acc = info["accepted"] / info["proposals"]
                flips_per_sweep_replica = info["accepted"] / a.replicas / seg
                print("sweeps %4d-%4d  beta %8.3f  %8.2f ms  %7.3f ms/sweep  acc %.4f  flips/sweep/replica %7.1f  "
                      "us per flip-round %.2f  %s" % (
                          i * seg, (i + 1) * seg, betas[i * seg], ms, ms / seg, acc, flips_per_sweep_replica,
                          ms / seg * 1e3 / max(flips_per_sweep_replica, 1e-9), p.kernel_name()), flush=True)
        print("total %.1f ms  %.3e upd/s" % (total, a.replicas * S * n / total * 1e3), flush=True)
