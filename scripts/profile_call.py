#!/usr/bin/env python3
"""cProfile of ONE drop-in call, `MI355XSampler().sample_qubo(model, num_reads=500, num_sweeps=1000)` at n = 2638 and 342:
where the host time of a call goes (development helper)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from scrna_seq_qannealing_clustering_amd import MI355XSampler, models
m, Qs, betas, _, graph = bench.build_workload()
s = MI355XSampler()
s.sample_qubo(m, num_reads=500, num_sweeps=10, seed=1)           # warm
for rep in range(3):
    t0 = time.perf_counter()
    ss = s.sample_qubo(m, num_reads=500, num_sweeps=1000, seed=1)
    t1 = time.perf_counter()
    print("call %.2f ms  (kernel %.2f, upload %.2f, anneal+fetch %.2f)" % ((t1 - t0) * 1e3, ss.info["timing"]["kernel_ms"],
          ss.info["timing"]["upload_s"] * 1e3, ss.info["timing"]["anneal_s"] * 1e3))
pr = cProfile.Profile()
pr.enable()
ss = s.sample_qubo(m, num_reads=500, num_sweeps=1000, seed=1)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(18)
print(out.getvalue()[:4500])
