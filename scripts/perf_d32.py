"""K2 / K3 on wider graphs: degree cap 30 (the reference's `..._k8_dim15_30.gexf` graphs, main.py:110) and 60 --
the 32- and 64-wide adjacency layouts.  Development helper."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import graphs, models
from scrna_seq_qannealing_clustering_amd.engine import Problem
from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range

R, S, n = 4096, 200, 2638
for k, ord_ in ((5, 15), (8, 30), (12, 60), (20, 120)):
    nodes, eu, ev, w, _ = graphs.synthetic_snn(n, k, 15, ord_, 9, seed=0)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    m = models.build_bqm_qubo(G, 0.05)
    deg = np.diff(m.rowptr)
    b = models.make_beta_schedule(S, models.default_beta_range(m))
    with Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32),
                           float(np.float32(m.c_pair)), order="padded") as p:
        p.anneal(R, b, 1); p.anneal(R, b, 1)
        ms = p.kernel_ms(); _, _, info = p.fetch()
    print("k=%d ord=%d  max degree %d mean %.1f   K2 %.2f ms  %.3e upd/s  acc %.3f" % (
        k, ord_, deg.max(), deg.mean(), ms, R * S * n / ms * 1e3, info["accepted"] / info["proposals"]), flush=True)
    pm = models.build_dqm_potts(G, 8, 0.005)
    pb = models.make_beta_schedule(S, default_potts_beta_range(pm))
    with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), n, 8,
                           lin_offset=pm.lin_offset, order="padded") as p:
        p.anneal(R, pb, 1); p.anneal(R, pb, 1)
        ms = p.kernel_ms(); _, _, info = p.fetch()
    print("                                        K3 %.2f ms  %.3e upd/s  acc %.3f" % (
        ms, R * S * n / ms * 1e3, info["accepted"] / info["proposals"]), flush=True)
