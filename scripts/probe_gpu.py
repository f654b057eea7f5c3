#!/usr/bin/env python3
"""Ad-hoc GPU probe: parity of the dense kernel vs the oracle on a fixture, then a timing run.
(Development helper; the real checks live in tests/ and bench.py.)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import _lib, models, graphs
from scrna_seq_qannealing_clustering_amd.engine import Problem
from oracle import sa_oracle as orc

def load_fixture(name):
    d = json.load(open(os.path.join("tests/golden/graphs", name + ".json")))
    w = np.array([float.fromhex(x) for x in d["weight_hex"]])
    return graphs.graph_from_edges(d["nodes"], d["edge_u"], d["edge_v"], w)

print("devices:", _lib.device_count(), _lib.device_info(0))
G = load_fixture("noisy_circles")
m = models.build_bqm_qubo(G, 0.05)
Qs = m.dense_Qs().astype(np.float32)
hot, cold = models.default_beta_range(m)
print("beta range", hot, cold)
for (R, S, resync) in [(8, 20, 0), (64, 200, 0), (16, 100, 7)]:
    betas = models.make_beta_schedule(S, (hot, cold))
    with Problem.dense(Qs) as p:
        p.anneal(R, betas, 1234, resync_interval=resync)
        st, en, info = p.fetch()
        ms = p.kernel_ms()
        bi, be, key, bs = p.best()
    ost, oen, ostats = orc.sa_dense_philox(Qs, R, betas, 1234, resync_interval=resync)
    same = np.array_equal(st, ost)
    print(f"R={R} S={S} resync={resync}: states bit-exact={same}  maxdE={np.max(np.abs(en-oen)):.3e} "
          f"acc gpu={info['accepted']} orc={int(ostats[1])} ms={ms:.3f} best={be:.6f}@{bi} minE={en.min():.6f}")
    if not same:
        bad = np.argwhere(st != ost)
        print("  first mismatches:", bad[:5], " n mismatching replicas:", len(set(bad[:,0])))

# timing on the PBMC3k-sized surrogate
t0 = time.time()
nodes, eu, ev, w, truth = graphs.synthetic_snn(2638, 5, 15, 15, 9, seed=0)
Gs = graphs.EdgeListGraph(nodes, eu, ev, w)
m2 = models.build_bqm_qubo(Gs, 0.05)
Q2 = m2.dense_Qs().astype(np.float32)
hot, cold = models.default_beta_range(m2)
print("surrogate built in %.1fs, m=%d, beta=(%g,%g)" % (time.time() - t0, len(w), hot, cold))
with Problem.dense(Q2) as p:
    for (R, S) in [(256, 50), (4096, 50), (4096, 200)]:
        betas = models.make_beta_schedule(S, (hot, cold))
        p.anneal(R, betas, 1234)
        ms = p.kernel_ms()
        st, en, info = p.fetch()
        ups = R * S * 2638 / (ms * 1e-3)
        print(f"n=2638 R={R} S={S}: {ms:.1f} ms  {ups:.3e} updates/s  acc_rate={info['accepted']/info['proposals']:.3f} "
              f"minE={en.min():.4f} rowGB/s={info['accepted']*44*64*4/(ms*1e-3)/1e9:.1f}")
    # small-R parity at n=2638
    betas = models.make_beta_schedule(10, (hot, cold))
    p.anneal(4, betas, 99)
    st, en, info = p.fetch()
ost, oen, ostats = orc.sa_dense_philox(Q2, 4, betas, 99)
print("n=2638 parity: bit-exact=", np.array_equal(st, ost), "maxdE=", np.max(np.abs(en - oen)), info['accepted'], int(ostats[1]))
