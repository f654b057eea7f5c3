#!/usr/bin/env python3
"""Compute the known-answer values of tests/golden/kat_values.json.

Inputs: the graph fixtures (tests/golden/graphs/*.json, dumped from the reference's bundled
R/benchmarks/*.gexf by make_graph_fixtures.py).  Method: the literal restatement of the reference's
builders in oracle/model_oracle.py (BQM_clustering.py:29-47, DQM_clustering.py:29-43), fp64.  The
values reproduce SURVEY.md section 8c digit for digit (asserted in tests/test_oracle_kat.py against
constants copied from the survey), which is what pins the model side of the oracle.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from conftest import GRAPH_NAMES, load_fixture  # noqa: E402
from oracle import model_oracle as mo  # noqa: E402
from oracle import sa_oracle as so  # noqa: E402


def bfs_nodes(fx, start, count):
    adj = {v: [] for v in fx.nodes}
    for u, v, _ in fx.edges:
        adj[u].append(v)
        adj[v].append(u)
    seen, order, queue = {start}, [start], [start]
    while queue and len(order) < count:     # nx.bfs_edges order: neighbours in adjacency order
        cur = queue.pop(0)
        for nb in adj[cur]:
            if nb not in seen:
                seen.add(nb)
                order.append(nb)
                queue.append(nb)
                if len(order) == count:
                    break
    return order


def dense_from_dict(Q, nodes):
    idx = {v: i for i, v in enumerate(nodes)}
    n = len(nodes)
    Qs = np.zeros((n, n))
    for (u, v), b in Q.items():
        i, j = idx[u], idx[v]
        if i == j:
            Qs[i, i] += b
        else:
            Qs[i, j] += b / 2
            Qs[j, i] += b / 2
    return Qs


def main():
    out = {}
    for name in GRAPH_NAMES:
        fx = load_fixture(name)
        Q, gamma = mo.q_bqm(fx.nodes, fx.edges, 0.05, edges_weights=fx.W, k=8)
        half = {v: int(int(v) < 128) for v in fx.nodes}
        comp = fx.components()
        comp0 = {v: int(comp[i] == comp[0]) for i, v in enumerate(fx.nodes)}
        ent = {
            "n": len(fx.nodes), "m": len(fx.edges), "W": fx.W, "gamma": gamma, "lenQ": len(Q),
            "half_E_dict": mo.qubo_energy(Q, half),
            "half_E_closed": mo.bqm_energy_closed_form(fx.nodes, fx.edges, gamma, half),
            "half_cut_edges": mo.cut_edges(fx.edges, half),
            "half_cut_w": sum(w for u, v, w in fx.edges if half[u] != half[v]),
            "comp0_size": int(sum(comp0.values())),
            "comp0_E_dict": mo.qubo_energy(Q, comp0),
            "comp0_E_closed": mo.bqm_energy_closed_form(fx.nodes, fx.edges, gamma, comp0),
            "comp0_cut_edges": mo.cut_edges(fx.edges, comp0),
            "bound": -gamma * len(fx.nodes) ** 2 / 4,
            "ones_E_dict": mo.qubo_energy(Q, {v: 1 for v in fx.nodes}),
        }
        out[name] = ent
    # DQM KAT on circles (K=3, gamma=0.005, labels = component id)
    fx = load_fixture("noisy_circles")
    lin, quad = mo.dqm_model(fx.nodes, fx.edges, 3, 0.005)
    comp = fx.components()
    ids = {c: k for k, c in enumerate(sorted(set(comp.tolist())))}
    labels = {v: ids[int(comp[i])] for i, v in enumerate(fx.nodes)}
    out["dqm_circles"] = {
        "K": 3, "gamma": 0.005, "sum_lin": float(sum(lin[v][0] for v in fx.nodes)),
        "E_pairwise": mo.dqm_energy(lin, quad, labels),
    }
    # brute-force KATs on 20-node BFS-induced subgraphs
    for gname, gf in (("noisy_moons", 1.0), ("noisy_moons", 0.05), ("aniso", 0.05)):
        fx = load_fixture(gname)
        sub = bfs_nodes(fx, "0", 20)
        keep = set(sub)
        edges = [(u, v, w) for u, v, w in fx.edges if u in keep and v in keep]
        # subgraph node order = the BFS list ("bit i = i-th listed node")
        W = sum(w for _, _, w in edges)
        Q, gamma = mo.q_bqm(sub, edges, gf, edges_weights=W, k=8)
        Qs = dense_from_dict(Q, sub)
        mn, am, nm, se = so.bruteforce_qubo(Qs)
        x = {v: (am >> i) & 1 for i, v in enumerate(sub)}
        out["brute_%s_gf%s" % (gname, str(gf).replace(".", "p"))] = {
            "nodes": sub, "m": len(edges), "W": W, "gamma": gamma, "min_E": mn, "argmin": am,
            "num_min": nm, "second_E": se, "cut_edges": mo.cut_edges(edges, x),
            "cut_w": sum(w for u, v, w in edges if x[u] != x[v]),
        }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_values.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main()
