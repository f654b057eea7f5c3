#!/usr/bin/env python3
"""Wall time of the reference's own workflow on one GPU: recursive bisection (BQM_clustering.py:25-204, `clustering_bqm`
with terminate_on="iter_limit") of the PBMC3k-sized SNN graph, 15 sampler calls of 500 reads x 1000 sweeps on
shrinking subgraphs -- where the time goes between model build, problem creation, anneal and SampleSet (cProfile).
Since round 3 the second half of every bisection is enqueued before the first half's subtree is worked through
(MI355XSampler.sample_qubo_async): kernels of sibling calls overlap, so their sum can exceed what the wall clock saw."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import MI355XSampler                           # noqa: E402
from scrna_seq_qannealing_clustering_amd.clustering import clustering_bqm               # noqa: E402
from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn  # noqa: E402

nodes, eu, ev, w, _ = synthetic_snn(2638)
G = graph_from_edges(nodes, eu, ev, w)
s = MI355XSampler()
calls = []
orig = s.sample_qubo


def traced(model, **kw):
    t0 = time.perf_counter()
    r = orig(model, **kw)
    calls.append((model.num_variables, time.perf_counter() - t0, r.info["timing"]))
    return r


s.sample_qubo = traced
orig_async = s.sample_qubo_async


def traced_async(model, **kw):                  # (the second half of every bisection is enqueued ahead: clustering.py)
    t0 = time.perf_counter()
    pend = orig_async(model, **kw)
    t_launch = time.perf_counter() - t0
    res = pend.result

    def result():
        t1 = time.perf_counter()
        r = res()
        calls.append((model.num_variables, t_launch + time.perf_counter() - t1, r.info["timing"]))
        return r
    pend.result = result
    return pend


s.sample_qubo_async = traced_async
clustering_bqm(G.subgraph(list(G.nodes)[:300]), 0, None, "mi355x", 0.05, 0, "once", 5, 3, 0, sampler=s)   # warm
calls.clear()
# (1) the plain wall time, no profiler attached; (2) the same run under cProfile for the split
for rep in range(3):
    t0 = time.perf_counter()
    clustering_bqm(G, 0, None, "mi355x", 0.05, 0, "iter_limit", 5, 3, 0, sampler=s,
                   sampler_kwargs={"seed": 7} if "--seed" in sys.argv else None)
    wall_plain = time.perf_counter() - t0
    print("recursive bisection, 4 levels, no profiler, run %d: %.3f s wall, %d sampler calls, kernels %.1f ms" % (
        rep, wall_plain, len(calls), sum(c[2]["kernel_ms"] for c in calls)))
    calls.clear()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
clustering_bqm(G, 0, None, "mi355x", 0.05, 0, "iter_limit", 5, 3, 0, sampler=s,
               sampler_kwargs={"seed": 7} if "--seed" in sys.argv else None)
pr.disable()
wall = time.perf_counter() - t0
print("recursive bisection, 4 levels: %.3f s wall, %d sampler calls" % (wall, len(calls)))
for n, t, tm in calls:
    print("   n = %5d   call %.1f ms   (upload %.1f, anneal+fetch %.1f, kernel %.2f)" % (
        n, t * 1e3, tm["upload_s"] * 1e3, tm["anneal_s"] * 1e3, tm["kernel_ms"]))
print("   sum of kernels %.1f ms = %.0f %% of the wall time" % (
    sum(c[2]["kernel_ms"] for c in calls), sum(c[2]["kernel_ms"] for c in calls) / wall / 10))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print(out.getvalue()[:7000])
