import sys, time, numpy as np
sys.path.insert(0, ".")
from scrna_seq_qannealing_clustering_amd import snn
from oracle import snn_oracle as sn
rng = np.random.RandomState(1)
for n, k, dim, ordd in ((300, 5, 15, 15), (64, 2, 1, 1), (2638, 5, 15, 15), (1500, 33, 50, 20), (20000, 5, 15, 15), (50000, 5, 15, 15)):
    c = rng.normal(scale=4.0, size=(30, dim)); X = (c[rng.randint(0, 30, size=n)] + rng.normal(size=(n, dim))).astype(np.float32)
    t0 = time.time(); g = snn.build_snn(X, k, 0.0, ordd); t1 = time.time()
    print(n, k, ordd, "%.3fs" % (t1 - t0), g.timing, flush=True)
    if n <= 20000:
        nn, rp, col, sh = sn.snn_graph(X, k, 0.0, ordd)
        print("   equal to oracle:", np.array_equal(g.rowptr, rp) and np.array_equal(g.col, col) and np.array_equal(g.shared, sh), flush=True)
