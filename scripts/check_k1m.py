#!/usr/bin/env python3
"""K1m alone (variant 3) against the workgroup kernel alone (variant 2) on random dense models of several sizes: the same
chain, so states and energies must be identical (development helper; the shipped checks are in tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

rng = np.random.default_rng(5)
bad = 0
for n in (100, 250, 700, 1500, 2638):
    Qs = rng.standard_normal((n, n)).astype(np.float32)
    Qs = np.triu(Qs) + np.triu(Qs, 1).T
    for beta in (0.002, 0.05):
        betas = np.full(12, beta)
        with Problem.dense(np.ascontiguousarray(Qs)) as p:
            p.set_option("chunk_sweeps", 0)
            out = []
            for variant in (3, 2):
                p.set_option("variant", variant)
                p.anneal(48, betas, 7, resync_interval=5)
                name = p.kernel_name()
                st, en, info = p.fetch()
                out.append((st, en, info["accepted"], name))
        same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
        bad += not same
        print("n=%5d beta=%g  %s  acc %.3f  %s | %s" % (n, beta, "same" if same else "DIFFERENT", out[0][2] / (48 * 12 * n),
                                                    out[0][3], out[1][3]), flush=True)
sys.exit(1 if bad else 0)
