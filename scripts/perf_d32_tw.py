#!/usr/bin/env python3
"""The 32-wide adjacency layout (degree cap 30: the reference's `..._k8_dim15_30.gexf` graphs, main.py:110): K2p with and
without its threshold wavefront, 4096 replicas x 200 sweeps; --check compares the states."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import graphs, models
from scrna_seq_qannealing_clustering_amd.engine import Problem
R, S, n = 4096, 200, 2638
nodes, eu, ev, w, _ = graphs.synthetic_snn(n, 8, 15, 30, 9, seed=0, spread=3.0)
m = models.build_bqm_qubo(graphs.EdgeListGraph(nodes, eu, ev, w), 0.05)
deg = np.diff(m.rowptr)
b = models.make_beta_schedule(S, models.default_beta_range(m))
with Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                       float(np.float32(m.c_pair)), order="padded") as p:
    print("max degree %d mean %.1f, device slots %d" % (deg.max(), deg.mean(), p.n_dev // 64))
    ref = None
    for rnd in range(3):
        for tw in (2, 0):
            p.set_option("k2_tw", tw)
            p.anneal(R, b, 1)
            ms = p.kernel_ms()
            st, _, info = p.fetch()
            if ref is None:
                ref = st.copy()
            print("k2_tw=%d  %8.2f ms  %.3e upd/s  acc %.3f  %s  %s" % (tw, ms, R * S * n / ms * 1e3, info["accepted"] / info["proposals"],
                                                                  p.kernel_name(), "same" if np.array_equal(ref, st) else "DIFFERENT"), flush=True)
