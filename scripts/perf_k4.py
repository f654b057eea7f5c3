#!/usr/bin/env python3
"""K4 (replica-batched energies on the matrix cores) alone: time of the whole call (transpose + MFMA kernel) for
R states of an n-variable dense model, and the deviation from the exact fp64 path (development helper).
usage: perf_k4.py [--n 2638] [--states 4096]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd.engine import energy_dense  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2638)
ap.add_argument("--states", type=int, default=4096)
a = ap.parse_args()
rs = np.random.RandomState(3)
A = rs.standard_normal((a.n, a.n)).astype(np.float32)
Qs = np.ascontiguousarray((A + A.T) / 2)
X = (rs.rand(a.states, a.n) < 0.4).astype(np.uint8)
energy_dense(Qs, X[:64], path=2)
times = []
for _ in range(6):
    e, ms = energy_dense(Qs, X, path=2, return_ms=True)
    times.append(ms)
exact = energy_dense(Qs, X[:256], path=1)
flops = 2.0 * a.n * a.n * a.states
T, rt = (a.n + 127) // 128, (a.states + 127) // 128
executed = T * (T + 1) // 2 * rt * 2.0 * 128 ** 3          # upper block triangle of the symmetric matrix
ms = min(times)
print(json.dumps({"n": a.n, "states": a.states, "call_ms": times, "dense_equivalent_tflops": flops / ms / 1e9,
                  "executed_tflops": executed / ms / 1e9, "executed_frac_of_157.3": executed / ms / 1e9 / 157.3,
                  "max_rel_diff_vs_fp64": float(np.max(np.abs(e[:256] - exact) / np.maximum(1.0, np.abs(exact))))}))
