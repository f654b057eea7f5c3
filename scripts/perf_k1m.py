#!/usr/bin/env python3
"""K1m (dense chain on the matrix cores) at a constant temperature: time per sweep with parts switched off
(development helper).  usage: perf_k1m.py [--beta B] [--sweeps S] debug=0 debug=1 ..."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--beta", type=float, default=0.01)
ap.add_argument("--sweeps", type=int, default=100)
ap.add_argument("--replicas", type=int, default=4096)
ap.add_argument("--variant", type=int, default=3)
ap.add_argument("arms", nargs="*", default=["debug=0"])
a = ap.parse_args()
m, Qs, _, _, _graph = bench.build_workload()
n = Qs.shape[0]
betas = np.full(a.sweeps, a.beta)
with Problem.dense(Qs) as p:
    p.set_option("variant", a.variant)
    p.set_option("chunk_sweeps", 0)
    for rnd in range(2):
        for arm in a.arms:
            for kv in arm.split(","):
                k, v = kv.split("=")
                p.set_option(k, int(v))
            p.anneal(a.replicas, betas, 1234)
            ms = p.kernel_ms()
            _, _, info = p.fetch(states=False)
            units = a.sweeps * 4 * ((n + 15) // 16)
            print("%-12s %8.2f ms  %.3f ms/sweep  %6.0f cycles/unit (2.4 GHz)  acc %.3f  %.3e upd/s  %s" % (
                arm, ms, ms / a.sweeps, ms * 1e-3 * 2.4e9 / units, info["accepted"] / info["proposals"],
                a.replicas * a.sweeps * n / ms * 1e3, p.kernel_name()), flush=True)
            ds = p.debug_stats()
            if ds[4:14].any():                     # a build with -DMI_K1M_TICKS: s_memtime ticks of the wave debug>>8, workgroup 0
                per = float(units) / 4.0
                print("     wave %d ticks per unit phase g=0..3: wait %s | after-barrier part: total %s of which bookkeeping %.0f DIAG %.0f (avg per unit)" % (
                    int(dict(kv.split("=") for kv in arm.split(","))["debug"]) >> 8,
                    " ".join("%6.0f" % (ds[4 + g] / per) for g in range(4)),
                    " ".join("%6.0f" % (ds[8 + g] / per) for g in range(4)), ds[12] / units, ds[13] / units), flush=True)
