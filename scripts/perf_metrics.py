"""M1 (all-pairs Jaccard statistics) and the SNN build at the sizes of the BASELINE configs: device times and
the implied pair / word rates.  Development helper (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import metrics, snn

rs = np.random.RandomState(0)
for n, genes, K in ((2638, 2000, 9), (10605, 2000, 15), (50000, 2000, 30)):
    X = (rs.rand(n, genes) < 0.08).astype(np.uint8)              # ~8 % of the genes expressed per cell
    lab = rs.randint(0, K, n)
    bits = metrics.pack_expression(X)
    r = metrics.jaccard_pass(bits, lab, K)
    r = metrics.jaccard_pass(bits, lab, K)
    ms = r["kernel_ms"]
    words = bits.shape[1]
    pairs = n * n
    print("M1 n=%d genes=%d (%d words)  %.2f ms  %.3e pairs/s  %.3e word-ops/s" % (
        n, genes, words, ms, pairs / ms * 1e3, pairs * words / ms * 1e3), flush=True)
    P = rs.normal(size=(n, 15)).astype(np.float32)
    g = snn.build_snn(P, 5, 0.0, 15)
    t = g.timing
    print("   SNN build: kNN %.2f ms (%.3e distance-pairs/s), SNN rows %.2f ms, trim %.2f ms" % (
        t["knn_ms"], n * n / t["knn_ms"] * 1e3, t["snn_ms"], t["trim_ms"]), flush=True)
