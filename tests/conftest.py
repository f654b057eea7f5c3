import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GRAPH_NAMES = ["noisy_circles", "noisy_moons", "varied", "aniso", "blobs", "no_structure"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Fixture:
    """One of the reference's bundled R/benchmarks graphs as dumped by tests/golden/make_graph_fixtures.py."""

    def __init__(self, name):
        d = json.load(open(os.path.join(GOLDEN, "graphs", name + ".json")))
        self.name = name
        self.nodes = d["nodes"]
        self.eu = np.asarray(d["edge_u"], dtype=np.int32)
        self.ev = np.asarray(d["edge_v"], dtype=np.int32)
        self.w = np.asarray([float.fromhex(x) for x in d["weight_hex"]], dtype=np.float64)
        self.W = float.fromhex(d["total_weight_hex"])
        self.edges = [(self.nodes[a], self.nodes[b], float(c))
                      for a, b, c in zip(self.eu.tolist(), self.ev.tolist(), self.w.tolist())]

    def graph(self):
        from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges
        return graph_from_edges(self.nodes, self.eu, self.ev, self.w)

    def components(self):
        parent = list(range(len(self.nodes)))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for a, b in zip(self.eu.tolist(), self.ev.tolist()):
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[rb] = ra
        return np.asarray([find(i) for i in range(len(self.nodes))])


_cache = {}


def load_fixture(name):
    if name not in _cache:
        _cache[name] = Fixture(name)
    return _cache[name]


@pytest.fixture(scope="session")
def kat():
    return json.load(open(os.path.join(GOLDEN, "kat_values.json")))


@pytest.fixture(scope="session", params=GRAPH_NAMES)
def any_graph(request):
    return load_fixture(request.param)


@pytest.fixture(scope="session")
def circles():
    return load_fixture("noisy_circles")


def have_gpu():
    try:
        from scrna_seq_qannealing_clustering_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False
