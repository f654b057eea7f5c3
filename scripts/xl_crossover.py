#!/usr/bin/env python3
"""K1g (all replicas together, MFMA passes, hand-over to K1x at the cold end) against K1x alone for SMALL replica
counts and sizes just above 4096: a complete hot-to-cold schedule on a random dense QUBO (development helper; the
numbers behind the `xl_batched` default, profiles/r02_xl_crossover.json).
usage: xl_crossover.py [--sweeps 64]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import models  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=64)
a = ap.parse_args()
rows = []
for n in (4500, 8192, 20000):
    rng = np.random.RandomState(n)
    A = rng.standard_normal((n, n)).astype(np.float32)
    A *= (rng.rand(n, n) < 0.02)
    Qs = np.triu(A, 1)
    Qs = np.ascontiguousarray(Qs + Qs.T)
    del A
    Qs[np.arange(n), np.arange(n)] = rng.standard_normal(n).astype(np.float32)
    # neal's default range for this model (hot: half the flips of the smallest field accepted; cold: 1 % of the largest)
    absrow = np.abs(Qs).sum(axis=1)
    mind = np.abs(Qs)[np.abs(Qs) > 0].min()
    betas = np.geomspace(np.log(2) / absrow.max(), np.log(100) / max(mind, 1e-3), a.sweeps)
    with Problem.dense(Qs) as p:
        for R in (1, 8, 64):
            rec = {"n": n, "replicas": R, "sweeps": a.sweeps}
            ref = None
            for mode, key in ((2, "k1x_ms"), (1, "k1g_handover_ms")):
                p.set_option("xl_batched", mode)
                p.anneal(R, betas, 7)
                p.anneal(R, betas, 7)
                rec[key] = p.kernel_ms()
                st, en, info = p.fetch()
                if ref is None:
                    ref = st
                    rec["acceptance"] = info["accepted"] / info["proposals"]
                else:
                    rec["identical"] = bool(np.array_equal(ref, st))
                    rec["kernel"] = p.kernel_name()
            rows.append(rec)
            print(json.dumps(rec), flush=True)
print(json.dumps({"xl_crossover": rows}))
