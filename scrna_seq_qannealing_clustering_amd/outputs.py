"""What the reference does with a solved model: labelled GEXF files (and optional pictures).

Counterparts of `Python_Functions/plot_and_save.py`, `other_tools.disconnected_components`
(`other_tools.py:71-87`) and `QA_subsampling.prune_graph` (`QA_subsampling.py:119-129`) with the reference's
names, argument order and -- the part downstream R notebooks depend on -- the node-attribute contract of the
written files:

    BQM drivers      every node carries ``label<iteration>`` colour ints; the partition a file encodes is the
                     LAST attribute a node received (`plot_and_save.py:16-17`)
    DQM / CQM        ``label1`` = cluster id of `sampleset.first.sample` (`:41-44`, `:61-63`)
    CQM on a subgraph  ``z_cluster`` per node, ``label1`` keyed by ``subindex`` (`:70-83`)
    sub-sampling     ``label1`` = 1 for kept nodes (`:86-102`)

Pictures are drawn only when a layout ``pos`` is handed over and matplotlib is importable: `spring_layout` is
O(n^2) per iteration and is what makes the reference's loader unusable beyond ~10k nodes, so
``pos=None`` -- what ``create_graph(..., layout=False)`` returns -- skips them and only writes the graph.
"""
from __future__ import annotations

import os
from collections import defaultdict

import networkx as nx

_TYPE_NAMES = ["_", "_trimmed_", "_negedges_", "_trimmed_negedges_"]


def define_dirs(n, k, dim, ord, g, gf, custom, type, root="."):
    """File-name scheme of `main.py:46-76`: n cells, k of the kNN, PCA dims, degree cap, gamma (DQM/CQM),
    gamma_factor (BQM), free-form suffix, graph type index into ("_", "_trimmed_", "_negedges_",
    "_trimmed_negedges_").  ``root`` replaces the reference's hard-wired "./"."""
    g = str(g).replace(".", "")
    gf = str(gf).replace(".", "")
    t = _TYPE_NAMES[type]
    stem = "_k%s_dim%s" % (k, dim)
    tail = "%s%s" % (t, ord)

    def p(folder, prefix, mid, suffix):
        return os.path.join(root, folder, "%s%s%s%s%s%s" % (n, prefix, stem, mid, tail, suffix))

    return {
        "name": "%s_graph_snn%s%s" % (n, stem, tail),
        "graph_in": p("DatasetsIn", "_graph_snn", "", ".gexf"),
        "graph_in_csv": p("DatasetsIn", "_graph_snn", "", ".csv"),
        "graph_in_pru": p("DatasetsIn", "_pru_graph_snn", "", custom + ".gexf"),
        "graph_out_bqm": p("DatasetsOut", "_graph_snn", "_gf" + gf, custom + "_out.gexf"),
        "graph_out_dqm": p("DatasetsOut", "_dqm_graph_snn", "_g" + g, custom + ".gexf"),
        "graph_out_cqm": p("DatasetsOut", "_cqm_graph_snn", "_g" + g, custom + ".gexf"),
        "graph_out_pru1": p("DatasetsOut", "_pru_graph_snn", "", custom + ".gexf"),
        "graph_out_pru2": p("DatasetsOut", "_pru_graph_snn", "", custom + "2.gexf"),
        "img_in": p("PlotsIn", "_graph_snn", "", custom + ".png"),
        "img_out_bqm": p("PlotsOut", "_bqm_graph_snn", "_gf" + gf, custom + "_out.png"),
        "img_out_dqm": p("PlotsOut", "_dqm_graph_snn", "_g" + g, custom + "_out.png"),
        "img_out_cqm": p("PlotsOut", "_cqm_graph_snn", "_g" + g, custom + "_out.png"),
        "img_out_p1": p("PlotsOut", "_pru_graph_snn", "", custom + "_out1.png"),
        "img_out_p2": p("PlotsOut", "_pru_graph_snn", "", custom + "_out2.png"),
        "img_out_p3": p("PlotsOut", "_pru_graph_snn", "", custom + "_out3.png"),
        "embedding": p("Embedding", "_graph_snn", "", ".json"),
        "embedding_pru": p("Embedding", "_pru_graph_snn", "", ".json"),
    }


def _ensure_parent(path):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)


def _canvas(pos):
    """matplotlib's pyplot with a cleared axis, or None when no picture is wanted / possible."""
    if pos is None:
        return None
    try:
        import matplotlib
        matplotlib.use("agg")
        from matplotlib import pyplot as plt
    except Exception:                                       # pragma: no cover - image without matplotlib
        return None
    plt.cla()
    return plt


def _save(plt, path):
    _ensure_parent(path)
    plt.savefig(path, bbox_inches="tight")


def _write(G, path):
    _ensure_parent(path)
    nx.write_gexf(G, path)


def last_labels(G):
    """node -> the attribute it received last (`plot_and_save.py:16`: ``list(G.nodes[u].values())[-1]``)."""
    return {u: list(d.values())[-1] for u, d in G.nodes(data=True)}


def split_edges(G):
    """(cut, uncut) edge lists under :func:`last_labels` (`plot_and_save.py:16-17`)."""
    lab = last_labels(G)
    cut, uncut = [], []
    for u, v in G.edges:
        (cut if lab[u] != lab[v] else uncut).append((u, v))
    return cut, uncut


def plot_and_save_graph_in(G, pos, dirs):
    """`plot_and_save.py:8-13`: picture of the input graph (nothing is written without a layout)."""
    plt = _canvas(pos)
    if plt is None:
        return
    nx.draw_networkx_nodes(G, pos, node_size=10, nodelist=G.nodes)
    nx.draw_networkx_edges(G, pos, edgelist=G.edges, style="solid", alpha=0.5, width=1)
    _save(plt, dirs["img_in"])


def plot_and_save_graph_out_bqm(G, pos, dirs):
    """`plot_and_save.py:15-34`: the recursive bisection's result.  Returns (cut, uncut) edge lists."""
    cut, uncut = split_edges(G)
    plt = _canvas(pos)
    if plt is not None:
        colors = [int(v) for v in last_labels(G).values()]
        nx.draw_networkx_nodes(G, pos, node_size=10, nodelist=G.nodes, node_color=colors)
        nx.draw_networkx_edges(G, pos, edgelist=cut, style="dashdot", alpha=0.5, width=1)
        nx.draw_networkx_edges(G, pos, edgelist=uncut, style="solid", width=1)
        _save(plt, dirs["img_out_bqm"])
    _write(G, dirs["graph_out_bqm"])
    return cut, uncut


def plot_and_save_graph_out_dqm(G, pos, dirs, sampleset):
    """`plot_and_save.py:36-44`: ``label1`` = case of the best sample."""
    lut = sampleset.first.sample
    plt = _canvas(pos)
    if plt is not None:
        nx.draw(G, pos=pos, with_labels=False, node_color=[lut[v] for v in G.nodes], node_size=10,
                cmap=plt.cm.rainbow)
        _save(plt, dirs["img_out_dqm"])
    nx.set_node_attributes(G, {v: int(lut[v]) for v in G.nodes}, name="label1")
    _write(G, dirs["graph_out_dqm"])


def _cluster_of(sample, key, num_of_clusters):
    """A CQM sample is either the reference's one-hot dict {'v_<key>,<p>': 0/1} (`CQM_clustering.py:34`) or
    this package's label dict {node: p}."""
    for p in range(num_of_clusters):
        if sample.get("v_%s,%d" % (key, p), 0) == 1:
            return p
    return None


def plot_and_save_graph_out_cqm(G, pos, dirs, sampleset_cqm, num_of_clusters):
    """`plot_and_save.py:46-63`.  Accepts the label-dict samples :func:`clustering.clustering_cqm` returns as
    well as the reference's one-hot samples (node ids must then be integers in 0..n-1, as there)."""
    sample = sampleset_cqm.first.sample
    labels = defaultdict(int)
    for node in G.nodes:
        p = sample[node] if node in sample else _cluster_of(sample, int(node), num_of_clusters)
        if p is not None:
            labels[node] = int(p)
    plt = _canvas(pos)
    if plt is not None:
        nx.draw(G, pos=pos, with_labels=False, node_color=[labels.get(v, -1) for v in G.nodes], node_size=10,
                cmap=plt.cm.rainbow)
        _save(plt, dirs["img_out_cqm"])
    nx.set_node_attributes(G, dict(labels), name="label1")
    _write(G, dirs["graph_out_cqm"])


def plot_and_save_graph_out_cqm_2(G, pos, dirs, sampleset_cqm, num_of_clusters):
    """`plot_and_save.py:65-83`: the subgraph variant -- nodes are addressed by their ``subindex``."""
    sample = sampleset_cqm.first.sample
    labels = {}
    for node in G.nodes:
        sub = G.nodes[node]["subindex"]
        p = sample[node] if node in sample else _cluster_of(sample, int(sub), num_of_clusters)
        if p is not None:
            G.nodes[node]["z_cluster"] = int(p)
            labels[sub] = int(p)
    plt = _canvas(pos)
    if plt is not None:
        nx.draw(G, pos=pos, with_labels=False, node_color=[G.nodes[v].get("z_cluster", -1) for v in G.nodes],
                node_size=10, cmap=plt.cm.rainbow)
        _save(plt, dirs["img_out_cqm"])
    nx.set_node_attributes(G, labels, name="label1")        # keyed by subindex, as the reference does (:82)
    _write(G, dirs["graph_out_cqm"])


def plot_and_save_graph_out_cqm_multi(G, pos, dirs, sampleset_cqm, num_of_clusters, number_of_samples,
                                      out_dir: str = "./graphs_multi_samples"):
    """`plot_and_save.py:104-126`: one labelled graph per sample for the first ``number_of_samples - 1`` samples of the
    set (the reference slices ``samples()[:number_of_samples-1]``), written as ``<out_dir>/sample_number<i>.gexf``
    (+ ``.png`` when a layout is given) with the cluster in node attribute ``label1``.  Returns the written paths."""
    samples = list(sampleset_cqm.samples()[:max(int(number_of_samples) - 1, 0)])
    paths = []
    for i, sample in enumerate(samples):
        graph_name = os.path.join(out_dir, "sample_number" + str(i))
        labels = defaultdict(int)
        for node in G.nodes:
            p = sample[node] if node in sample else _cluster_of(sample, int(node), num_of_clusters)
            if p is not None:
                labels[node] = int(p)
        plt = _canvas(pos)
        if plt is not None:
            nx.draw(G, pos=pos, with_labels=False, node_color=[labels.get(v, -1) for v in G.nodes], node_size=10,
                    cmap=plt.cm.rainbow)
            _save(plt, graph_name + ".png")
        nx.set_node_attributes(G, dict(labels), name="label1")
        _write(G, graph_name + ".gexf")
        paths.append(graph_name + ".gexf")
    return paths


def plot_and_save_graph_out_mvc(G, pos, dirs):
    """`plot_and_save.py:85-102`: the sub-sampling result (``label1`` = 1 kept).  Returns the edges that touch a
    kept node and the rest."""
    kept = {u for u, d in G.nodes(data=True) if d["label1"] == 1}
    included, excluded = [], []
    for u, v in G.edges:
        (included if (u in kept or v in kept) else excluded).append((u, v))
    plt = _canvas(pos)
    if plt is not None:
        lab = {u: d["label1"] for u, d in G.nodes(data=True)}
        nx.draw_networkx_nodes(G, pos, node_size=5, nodelist=G.nodes, node_color=list(lab.values()))
        nx.draw_networkx_labels(G, pos, labels=lab, font_size=5, font_color="r")
        nx.draw_networkx_edges(G, pos, edgelist=excluded, style="dashdot", alpha=0.5, width=0.5)
        nx.draw_networkx_edges(G, pos, edgelist=included, style="solid", width=1)
        _save(plt, dirs["img_out_p1"])
    _write(G, dirs["graph_out_pru1"])
    return included, excluded


def prune_graph(G, pos, dirs):
    """`QA_subsampling.py:119-129`: the subgraph induced by the kept nodes, written as ``graph_out_pru2``."""
    H = G.subgraph([u for u, d in G.nodes(data=True) if d["label1"] == 1])
    _write(H, dirs["graph_out_pru2"])
    plt = _canvas(pos)
    if plt is not None:
        nx.draw_networkx_nodes(H, pos, node_size=20, nodelist=H.nodes)
        nx.draw_networkx_edges(H, pos, edgelist=H.edges, style="solid", width=1)
        _save(plt, dirs["img_out_p2"])
    return H


def disconnected_components(G, min_valid: int = 15, verbose: bool = False):
    """`other_tools.py:71-87`: component copies S (in `nx.connected_components` order), their sizes in
    descending order, and the ``subindex`` / ``valid`` node attributes: components with more than 15 nodes
    are valid and their nodes are numbered 0.. in component order, the rest get ``valid = 0``."""
    comps = list(nx.connected_components(G))
    lengths = sorted((len(c) for c in comps), reverse=True)
    if verbose:
        print(lengths)
    S = [G.subgraph(c).copy() for c in comps]
    for s in S:
        if len(s) > min_valid:
            for sub, node in enumerate(s.nodes()):
                G.nodes[node]["subindex"] = sub
                G.nodes[node]["valid"] = 1
        else:
            for node in s.nodes():
                G.nodes[node]["valid"] = 0
    return G, S, lengths
