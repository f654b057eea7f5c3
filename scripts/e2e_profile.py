"""Where the wall time of one drop-in call goes (n = 2638, 4096 reads x 1000 sweeps): model build, sampler
call (upload + anneal + fetch + SampleSet), split by cProfile.  GPU box."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import MI355XSampler, build_bqm_qubo          # noqa: E402
from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn  # noqa: E402

t0 = time.perf_counter()
nodes, eu, ev, w, _ = synthetic_snn(2638)
G = graph_from_edges(nodes, eu, ev, w)
t1 = time.perf_counter()
model = build_bqm_qubo(G, 0.05)
t2 = time.perf_counter()
s = MI355XSampler()
s.sample_qubo(model, num_reads=64, num_sweeps=10, seed=1)                              # warm the device
t3 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
ss = s.sample_qubo(model, num_reads=4096, num_sweeps=1000, seed=1234)
first = ss.first
pr.disable()
t4 = time.perf_counter()
print("graph %.3f s, build_bqm_qubo %.3f s, warm-up %.3f s, sample_qubo(4096 x 1000) %.3f s, E = %.4f"
      % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, first.energy))
print("timing info:", {k: v for k, v in ss.info.items() if "ms" in k or "time" in k})
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(18)
print(out.getvalue()[:6000])
# the reference's own call shape: a dict of n(n+1)/2 entries
if "--dict" in sys.argv:
    from itertools import combinations
    from collections import defaultdict
    t5 = time.perf_counter()
    Q = defaultdict(int)
    gamma = model.info["gamma"]
    for u, v, d in G.edges(data=True):
        Q[(u, u)] += 8 * d["weight"]; Q[(v, v)] += 8 * d["weight"]; Q[(u, v)] += -16 * d["weight"]
    for i in G.nodes:
        Q[(i, i)] += gamma * (1 - len(G.nodes))
    for i, j in combinations(G.nodes, 2):
        Q[(i, j)] += 2 * gamma
    t6 = time.perf_counter()
    ss2 = s.sample_qubo(Q, num_reads=4096, num_sweeps=1000, seed=1234)
    t7 = time.perf_counter()
    print("reference-style dict: build %.3f s (the reference's own cost), sample_qubo(dict) %.3f s, E = %.4f"
          % (t6 - t5, t7 - t6, ss2.first.energy))

# the k-way call (DQM_clustering.py:45) and the dict lifting, profiled the same way
from scrna_seq_qannealing_clustering_amd import build_dqm_potts            # noqa: E402
t8 = time.perf_counter()
pm = build_dqm_potts(G, 8, 0.005)
t9 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
sd = s.sample_dqm(pm, num_reads=4096, num_sweeps=1000, seed=1234)
_ = sd.first
pr.disable()
t10 = time.perf_counter()
print("build_dqm_potts %.3f s, sample_dqm(4096 x 1000) %.3f s (kernel %.1f ms), E = %.4f"
      % (t9 - t8, t10 - t9, sd.info["timing"]["kernel_ms"], sd.first.energy))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(10)
print(out.getvalue()[:3500])
if "--dict" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    from scrna_seq_qannealing_clustering_amd import qubo_dict_to_model
    qm = qubo_dict_to_model(Q)
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(12)
    print(out.getvalue()[:4000])
