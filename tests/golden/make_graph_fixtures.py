#!/usr/bin/env python3
"""Generate tests/golden/graphs/*.json from the reference's bundled GEXF graphs.

Run ONLY in the build container (needs /root/reference).  It imports the reference's own
loader ``Python_Functions/create_graphs.py:create_graph`` (create_graphs.py:5-8) so that node
order, ``G.edges`` order and the float64 weights are exactly what the reference's builders
(`BQM_clustering.py:38-47`, `DQM_clustering.py:30-43`) would iterate over.  The output is data
only: node ids, edge endpoints (as indices into the node list, in ``G.edges`` order) and the
weights as IEEE-754 hex strings.  The six distinct graphs of R/benchmarks are dumped (the other
eight files there are byte-level copies).
"""
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graphs")
NAMES = ["noisy_circles", "noisy_moons", "varied", "aniso", "blobs", "no_structure"]


def main():
    sys.path.insert(0, REF)
    from Python_Functions.create_graphs import create_graph  # reference loader

    os.makedirs(OUT, exist_ok=True)
    for name in NAMES:
        G, _pos = create_graph(os.path.join(REF, "R", "benchmarks", "graph_%s.gexf" % name))
        nodes = list(G.nodes)
        idx = {v: i for i, v in enumerate(nodes)}
        eu, ev, wh = [], [], []
        for u, v in G.edges:
            eu.append(idx[u])
            ev.append(idx[v])
            wh.append(float(G.get_edge_data(u, v)["weight"]).hex())
        doc = {
            "source": "R/benchmarks/graph_%s.gexf" % name,
            "loader": "Python_Functions/create_graphs.py:create_graph",
            "nodes": nodes,
            "edge_u": eu,
            "edge_v": ev,
            "weight_hex": wh,
            "total_weight_hex": float(G.size(weight="weight")).hex(),
        }
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
        print(name, len(nodes), len(eu), G.size(weight="weight"))


if __name__ == "__main__":
    main()
