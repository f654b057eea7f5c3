import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from scrna_seq_qannealing_clustering_amd import graphs, models
from scrna_seq_qannealing_clustering_amd.engine import Problem
from scrna_seq_qannealing_clustering_amd.sampler import default_potts_beta_range
R, S, n = 4096, 200, 2638
for k, ord_ in ((5, 15), (8, 30), (12, 60)):
    nodes, eu, ev, w, _ = graphs.synthetic_snn(n, k, 15, ord_, 9, seed=0)
    G = graphs.EdgeListGraph(nodes, eu, ev, w)
    pm = models.build_dqm_potts(G, 8, 0.005)
    pb = models.make_beta_schedule(S, default_potts_beta_range(pm))
    with Problem.potts_csr(pm.rowptr, pm.col, pm.val.astype(np.float32), float(np.float32(pm.c_pair)), n, 8,
                           lin_offset=pm.lin_offset, order="slots") as p:
        ref = None
        for mode in (0, 99, 0, 99):
            p.set_option("k2_waves", mode)
            p.anneal(R, pb, 1)
            ms = p.kernel_ms(); lab, _, info = p.fetch()
            if ref is None: ref = lab
            assert np.array_equal(ref, lab)
            print("k=%d ord=%d mode %d K3 %.2f ms  %.3e upd/s" % (k, ord_, mode, ms, R * S * n / ms * 1e3), flush=True)
