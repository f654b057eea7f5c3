#!/usr/bin/env python3
"""Wall time of the reference's k-way calls on one GPU (DQM_clustering.py:24-47, CQM_clustering.py:26-55) on the
PBMC3k-sized graph: model build, upload, anneal (the sampler's default 256 reads x 1000 sweeps), SampleSet."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import MI355XSampler, models                  # noqa: E402
from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn  # noqa: E402

nodes, eu, ev, w, _ = synthetic_snn(2638)
G = graph_from_edges(nodes, eu, ev, w)
s = MI355XSampler()
for K in (8, 15):
    for rep in range(3):
        t0 = time.perf_counter()
        pm = models.build_dqm_potts(G, K, 0.005)
        t1 = time.perf_counter()
        ss = s.sample_dqm(pm)
        t2 = time.perf_counter()
        print("K = %2d  model %.1f ms  sample_dqm %.1f ms (upload %.1f, anneal+fetch %.1f, kernel %.2f)  best E %.3f" % (
            K, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ss.info["timing"]["upload_s"] * 1e3, ss.info["timing"]["anneal_s"] * 1e3,
            ss.info["timing"]["kernel_ms"], ss.first.energy), flush=True)
pr = cProfile.Profile()
pr.enable()
pm = models.build_dqm_potts(G, 8, 0.005)
ss = s.sample_dqm(pm)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
print(out.getvalue()[:5000])
