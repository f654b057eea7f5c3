#!/usr/bin/env python3
"""A/B timing of the structured kernels on the bench graph (development helper).
usage: perf_k2.py [--sweeps S] [--replicas R] [--n N] variant=0 variant=1 ..."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import graphs, models  # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=200)
ap.add_argument("--replicas", type=int, default=4096)
ap.add_argument("--n", type=int, default=2638)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--check", action="store_true")
ap.add_argument("--order", default=None)
ap.add_argument("--block", type=int, default=64, help="seats per edge-free block of the padded layout (128 / 256: the few-replica kernel)")
ap.add_argument("--spread", type=float, default=3.0, help="cluster spread of the surrogate graph (3.0 = bench.py workload)")
ap.add_argument("arms", nargs="*", default=["k2_waves=0"])
a = ap.parse_args()
nodes, eu, ev, w, _ = graphs.synthetic_snn(a.n, 5, 15, 15, 9, seed=0, spread=a.spread)
G = graphs.EdgeListGraph(nodes, eu, ev, w)
m = models.build_bqm_qubo(G, 0.05, k=8)
betas = models.make_beta_schedule(a.sweeps, models.default_beta_range(m))
ref = None
with Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                       float(np.float32(m.c_pair)), order=a.order, block=a.block) as p:
    print("n = %d, device slots = %d" % (a.n, p.n_dev // 64))
    for rnd in range(a.rounds):
        for arm in a.arms:
            for kv in arm.split(","):
                k, v = kv.split("=")
                p.set_option(k, int(v))
            p.anneal(a.replicas, betas, 1234)
            ms = p.kernel_ms()
            st, en, info = p.fetch()
            ds = p.debug_stats()
            if ds[8:13].any():
                tot = float(ds[8:13].sum())
                if "split" in p.kernel_name():
                    steps = (p.n_dev // 64) // int(p.kernel_name().split(",")[1].strip(" >"))
                    print("   K2s cycles/step (wave 0): [adjacency wait + LDS reads issued %.0f, threshold + random words %.0f, rest of the gathers %.0f]  "
                          "sum %.0f  first solve %.0f  publish+barrier+read %.0f  further passes %.0f" % tuple(
                              float(x) / a.replicas / a.sweeps / steps for x in (ds[13], ds[6], ds[8], ds[9], ds[10], ds[11], ds[12])))
                elif "pair" in p.kernel_name():
                    slots = p.n_dev // 64
                    print("   K2p cycles per slot pair (sweeping wavefront; s_memtime ticks): top+prefetch %.0f  gathers issued->arrived %.0f  "
                          "field sums + first masks %.0f  rounds + commit %.0f  barrier %.0f" % tuple(
                              float(x) / (a.replicas / 2) / a.sweeps / slots for x in (ds[8], ds[9], ds[10], ds[11], ds[12])))
                else:
                    print("   phase cycles/wave/sweep: pre %.0f loop %.0f wait %.0f field-sum %.0f slot-top %.0f" % tuple(
                        float(x) / a.replicas / a.sweeps for x in ds[8:13]))
            if a.check:
                if ref is None:
                    ref = st.copy()
                assert np.array_equal(ref, st), "arm %s changed the results" % arm
            print("round %d %-20s %9.2f ms  %.3e upd/s  acc %.3f  minE %.3f  %s" % (
                rnd, arm, ms, a.replicas * a.sweeps * a.n / ms * 1e3, info["accepted"] / info["proposals"], en.min(),
                p.kernel_name()), flush=True)
