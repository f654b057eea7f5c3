"""The steps around the solved model: file-name scheme, labelled GEXF output, component bookkeeping
(`main.py:46-76`, `plot_and_save.py`, `other_tools.py:71-87`, `QA_subsampling.py:119-129`).  CPU only."""
import os

import networkx as nx
import pytest

from conftest import load_fixture
from scrna_seq_qannealing_clustering_amd import outputs
from scrna_seq_qannealing_clustering_amd.clustering import one_hot_sample
from scrna_seq_qannealing_clustering_amd.sampleset import SampleSet


def test_define_dirs_matches_the_reference_scheme():
    # main.py:84-99 defaults: n=256,k=5,ord=15,dim=15,g_type=1,gamma=0.005,gamma_factor=0.05
    d = outputs.define_dirs(256, 5, 15, 15, 0.005, 0.05, "benchmark_100_no_structure", 1)
    assert d["name"] == "256_graph_snn_k5_dim15_trimmed_15"
    assert d["graph_in"] == "./DatasetsIn/256_graph_snn_k5_dim15_trimmed_15.gexf"
    assert d["graph_in_csv"] == "./DatasetsIn/256_graph_snn_k5_dim15_trimmed_15.csv"
    assert d["graph_in_pru"] == "./DatasetsIn/256_pru_graph_snn_k5_dim15_trimmed_15benchmark_100_no_structure.gexf"
    assert d["graph_out_bqm"] == ("./DatasetsOut/256_graph_snn_k5_dim15_gf005_trimmed_15"
                                  "benchmark_100_no_structure_out.gexf")
    assert d["graph_out_dqm"] == ("./DatasetsOut/256_dqm_graph_snn_k5_dim15_g0005_trimmed_15"
                                  "benchmark_100_no_structure.gexf")
    assert d["graph_out_pru2"] == "./DatasetsOut/256_pru_graph_snn_k5_dim15_trimmed_15benchmark_100_no_structure2.gexf"
    assert d["img_out_bqm"] == ("./PlotsOut/256_bqm_graph_snn_k5_dim15_gf005_trimmed_15"
                                "benchmark_100_no_structure_out.png")
    assert d["img_out_p3"] == "./PlotsOut/256_pru_graph_snn_k5_dim15_trimmed_15benchmark_100_no_structure_out3.png"
    assert d["embedding"] == "./Embedding/256_graph_snn_k5_dim15_trimmed_15.json"
    assert len(d) == 18
    e = outputs.define_dirs(1000, 10, 30, 15, 0.005, 0.05, "", 0)     # the kidney graphs (main.py:106-108)
    assert e["name"] == "1000_graph_snn_k10_dim30_15"


def _toy():
    G = nx.Graph()
    G.add_weighted_edges_from([("0", "1", 1.0), ("1", "2", 0.25), ("2", "3", 1.0), ("3", "0", 0.5)])
    return G


def test_bqm_output_uses_the_last_attribute(tmp_path):
    G = _toy()
    for v, c in zip("0123", (7, 7, 150, 150)):
        G.nodes[v]["label1"] = c
    G.nodes["2"]["label2"] = 9                                     # a deeper split relabels two nodes
    G.nodes["3"]["label2"] = 170
    assert outputs.last_labels(G) == {"0": 7, "1": 7, "2": 9, "3": 170}
    dirs = outputs.define_dirs(4, 5, 15, 15, 0.005, 0.05, "", 1, root=str(tmp_path))
    cut, uncut = outputs.plot_and_save_graph_out_bqm(G, None, dirs)
    assert sorted(cut) == [("0", "3"), ("1", "2"), ("2", "3")] and uncut == [("0", "1")]
    H = nx.read_gexf(dirs["graph_out_bqm"])
    assert list(H.nodes) == list(G.nodes)
    assert H.nodes["2"]["label1"] == 150 and H.nodes["2"]["label2"] == 9 and "label2" not in H.nodes["0"]
    assert H["1"]["2"]["weight"] == 0.25
    assert not os.path.exists(dirs["img_out_bqm"])                 # no layout, no picture


def test_bqm_output_draws_when_a_layout_is_given(tmp_path):
    pytest.importorskip("matplotlib")
    G = _toy()
    for v, c in zip("0123", (7, 7, 150, 150)):
        G.nodes[v]["label1"] = c
    dirs = outputs.define_dirs(4, 5, 15, 15, 0.005, 0.05, "", 1, root=str(tmp_path))
    pos = nx.circular_layout(G)
    outputs.plot_and_save_graph_in(G, pos, dirs)
    outputs.plot_and_save_graph_out_bqm(G, pos, dirs)
    assert os.path.getsize(dirs["img_in"]) > 0 and os.path.getsize(dirs["img_out_bqm"]) > 0


def _sampleset(labels, vartype="DISCRETE"):
    import numpy as np
    return SampleSet(np.asarray([list(labels.values())]), np.asarray([-1.0]), list(labels), vartype=vartype)


def test_dqm_and_cqm_outputs_write_label1(tmp_path):
    G = _toy()
    ss = _sampleset({"0": 2, "1": 2, "2": 0, "3": 1})
    dirs = outputs.define_dirs(4, 5, 15, 15, 0.005, 0.05, "", 1, root=str(tmp_path))
    outputs.plot_and_save_graph_out_dqm(G, None, dirs, ss)
    H = nx.read_gexf(dirs["graph_out_dqm"])
    assert [H.nodes[v]["label1"] for v in "0123"] == [2, 2, 0, 1]

    # CQM: the label-dict sample and the reference's one-hot sample give the same file
    G1, G2 = _toy(), _toy()
    outputs.plot_and_save_graph_out_cqm(G1, None, dirs, ss, 3)
    a = nx.read_gexf(dirs["graph_out_cqm"])

    class OneHot:
        class first:
            sample = {k.replace("v_", "v_"): v for k, v in one_hot_sample(ss.first.sample, 3).items()}
    outputs.plot_and_save_graph_out_cqm(G2, None, dirs, OneHot, 3)
    b = nx.read_gexf(dirs["graph_out_cqm"])
    assert [a.nodes[v]["label1"] for v in "0123"] == [b.nodes[v]["label1"] for v in "0123"] == [2, 2, 0, 1]


def test_cqm_multi_writes_one_graph_per_sample(tmp_path):
    """plot_and_save.py:104-126: the first number_of_samples - 1 samples, one labelled GEXF each."""
    import numpy as np
    G = _toy()
    ss = SampleSet(np.asarray([[2, 2, 0, 1], [0, 1, 1, 2], [1, 1, 1, 0]]), np.asarray([-3.0, -2.0, -1.0]), list("0123"),
                   vartype="DISCRETE")
    paths = outputs.plot_and_save_graph_out_cqm_multi(G, None, {}, ss, 3, 3, out_dir=str(tmp_path / "multi"))
    assert [p.split("/")[-1] for p in paths] == ["sample_number0.gexf", "sample_number1.gexf"]
    a, b = nx.read_gexf(paths[0]), nx.read_gexf(paths[1])
    assert [a.nodes[v]["label1"] for v in "0123"] == [2, 2, 0, 1]            # samples come sorted by energy
    assert [b.nodes[v]["label1"] for v in "0123"] == [0, 1, 1, 2]


def test_cqm_2_addresses_nodes_by_subindex(tmp_path):
    G = _toy()
    for sub, v in enumerate(["3", "2", "1", "0"]):
        G.nodes[v]["subindex"] = sub
    ss = _sampleset({"0": 1, "1": 1, "2": 0, "3": 0})
    dirs = outputs.define_dirs(4, 5, 15, 15, 0.005, 0.05, "", 1, root=str(tmp_path))
    outputs.plot_and_save_graph_out_cqm_2(G, None, dirs, ss, 2)
    assert [G.nodes[v]["z_cluster"] for v in "0123"] == [1, 1, 0, 0]

    G2 = _toy()
    for sub, v in enumerate(["3", "2", "1", "0"]):
        G2.nodes[v]["subindex"] = sub

    class OneHot:
        class first:
            sample = one_hot_sample(ss.first.sample, 2, G)
    outputs.plot_and_save_graph_out_cqm_2(G2, None, dirs, OneHot, 2)
    assert [G2.nodes[v]["z_cluster"] for v in "0123"] == [1, 1, 0, 0]


def test_subsampling_outputs(tmp_path):
    G = _toy()
    for v, keep in zip("0123", (1, 0, 0, 1)):
        G.nodes[v]["label1"] = keep
    dirs = outputs.define_dirs(4, 5, 15, 15, 0.005, 0.05, "", 1, root=str(tmp_path))
    inc, exc = outputs.plot_and_save_graph_out_mvc(G, None, dirs)
    assert sorted(inc) == [("0", "1"), ("0", "3"), ("2", "3")] and exc == [("1", "2")]
    H = outputs.prune_graph(G, None, dirs)
    assert sorted(H.nodes) == ["0", "3"] and list(H.edges) in ([("0", "3")], [("3", "0")])
    assert sorted(nx.read_gexf(dirs["graph_out_pru2"]).nodes) == ["0", "3"]


def test_disconnected_components_on_the_blobs_fixture():
    fx = load_fixture("blobs")                                     # three components of 86 / 85 / 85 cells
    G = fx.graph()
    G.add_node("lonely")
    G2, S, lengths = outputs.disconnected_components(G)
    assert G2 is G and lengths == [86, 85, 85, 1] and len(S) == 4
    for s in S:
        if len(s) > 15:
            assert [G.nodes[v]["subindex"] for v in s.nodes()] == list(range(len(s)))
            assert all(G.nodes[v]["valid"] == 1 for v in s.nodes())
    assert G.nodes["lonely"]["valid"] == 0 and "subindex" not in G.nodes["lonely"]
