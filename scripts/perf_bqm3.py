#!/usr/bin/env python3
"""`clustering_bqm_3`'s model (cut term + squared size window with slack bits, BQM_clustering.py:363-380) on the bench
graph: the structured kernels (weighted pair term) against the dense ones, through the sampler."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scrna_seq_qannealing_clustering_amd import MI355XSampler, models
m, Qs, betas, _, graph = bench.build_workload()
n = m.num_variables
base = models.build_bqm3_cut_qubo(graph, k=8)
W = float(np.sum(graph.w)) if hasattr(graph, "w") else 1.0
pen = models.add_size_window_penalty(base, lb=200, ub=n / 3, lagrange_multiplier=0.05 * W / n)
print("n = %d (+ %d slack bits, weights %s)" % (n, pen.num_variables - n, None if pen.weights is None else pen.weights[n:].tolist()))
for reads, sweeps in ((4096, 200), (500, 1000)):
    for kernel in ("csr", "dense"):
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            ss = MI355XSampler().sample_qubo(pen, num_reads=reads, num_sweeps=sweeps, seed=3, kernel=kernel)
            wall = time.perf_counter() - t0
            ms = ss.info["timing"]["kernel_ms"] if "timing" in ss.info else float("nan")
        print("%5d reads x %4d sweeps  kernel=%-5s  kernels %8.2f ms  %.3e upd/s  wall %.3f s  best E %.4f  size %d" % (
            reads, sweeps, kernel, ms, reads * sweeps * pen.num_variables / ms * 1e3, wall, ss.first.energy,
            int(sum(ss.first.sample[v] for v in pen.variables[:n]))), flush=True)
