"""cProfile of the parallel-tempering driver loop on config 5's shape (development helper)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import models, tempering
from scrna_seq_qannealing_clustering_amd.engine import Problem
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
n, K = 10605, 15
rng = np.random.RandomState(1)
eu = np.concatenate([np.arange(n)] * 3); ev = np.concatenate([(np.arange(n) + d) % n for d in (1, 7, 131)])
lo, hi = np.minimum(eu, ev), np.maximum(eu, ev)
w = rng.choice([1 / 9, 0.25, 3 / 7, 2 / 3, 1.0], size=len(lo))
rowptr, col, val = models._csr_from_edges(n, lo.astype(np.int32), hi.astype(np.int32), -2.0 * w)
prob = Problem.potts_csr(rowptr, col, val.astype(np.float32), 0.01, n, K, order="slots")
ladder = np.geomspace(0.5, 30.0, 8)
eng = tempering.ProblemEngine(prob, 1234)
tempering.parallel_tempering(eng, ladder, 128, 2, 10, 1234)      # warm
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
res = tempering.parallel_tempering(eng, ladder, 128, 40, 10, 1234)
pr.disable(); wall = time.perf_counter() - t0
print("wall %.3f s" % wall)
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(14); print(out.getvalue()[:3500])
