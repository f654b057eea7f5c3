#!/usr/bin/env python3
"""A/B of the sampler's layout choice on the 15-call bisection (same process, same box): blocks of 64 seats everywhere
against "auto" (the 128-seat layout where it needs few enough blocks).  Prints wall and kernel time of both, alternating."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import MI355XSampler, engine, sampler as sampler_mod          # noqa: E402
from scrna_seq_qannealing_clustering_amd.clustering import clustering_bqm               # noqa: E402
from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn  # noqa: E402

nodes, eu, ev, w, _ = synthetic_snn(2638)
G = graph_from_edges(nodes, eu, ev, w)
s = MI355XSampler()
kern = []
orig = s.sample_qubo


def traced(model, **kw):
    r = orig(model, **kw)
    kern.append(r.info["timing"]["kernel_ms"])
    return r


s.sample_qubo = traced
auto = engine.layout_block_for
for rep in range(4):
    for name, fn in (("64", lambda n, r, d=16: 64), ("auto", auto)):
        sampler_mod.layout_block_for = fn
        kern.clear()
        t0 = time.perf_counter()
        clustering_bqm(G, 0, None, "mi355x", 0.05, 0, "iter_limit", 5, 3, 0, sampler=s, sampler_kwargs={"seed": 7})
        print("run %d  layout %-4s  wall %.3f s  kernels %.1f ms" % (rep, name, time.perf_counter() - t0, sum(kern)), flush=True)
