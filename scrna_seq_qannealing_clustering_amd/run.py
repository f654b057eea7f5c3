"""`python -m scrna_seq_qannealing_clustering_amd.run` -- the reference's `main.py` as a command.

`main.py:78-161` is a script of notebook-style cells: parameters at the top (`:78-97`), graph import
(`:113-116`), then one block per method (`:127-161`), each followed by its `plot_and_save_*` call.  This entry
runs the same blocks with the same parameter names and defaults, the MI355X sampler standing where the
script constructs a D-Wave sampler; ``--method`` selects the block(s).

    python -m scrna_seq_qannealing_clustering_amd.run --graph R/benchmarks/graph_noisy_circles.gexf \
           --method bqm --out /tmp/out --num-reads 512

Prints one JSON line per method (energy of the best sample, sizes, output file).
"""
from __future__ import annotations

import argparse
import json
import sys

METHODS = ("subsampling", "subsampling_2", "dqm", "cqm", "cqm_2", "bqm", "bqm_2", "bqm_3")


def _parser():
    ap = argparse.ArgumentParser(prog="scrna_seq_qannealing_clustering_amd.run", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--graph", help="input .gexf (create_graph) or .csv (create_graph_csv); default: the "
                    "automatic name dirs['graph_in'] under --root")
    ap.add_argument("--method", action="append", choices=METHODS + ("all",), help="block(s) of main.py to run")
    ap.add_argument("--root", default=".", help="folder holding DatasetsIn/ (the reference uses ./)")
    ap.add_argument("--out", default=None, help="folder for DatasetsOut/ PlotsOut/ (default: --root)")
    ap.add_argument("--layout", action="store_true", help="compute spring_layout and draw the pictures")
    # main.py:84-97, same names and defaults
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--ord", type=int, default=15)
    ap.add_argument("--dim", type=int, default=15)
    ap.add_argument("--g-type", type=int, default=1)
    ap.add_argument("--color", type=int, default=0)
    ap.add_argument("--gamma-factor", type=float, default=0.05)
    ap.add_argument("--gamma", type=float, default=0.005)
    ap.add_argument("--custom", default="")
    ap.add_argument("--terminate-on", default="conf", choices=("conf", "min_size", "once", "iter_limit"))
    ap.add_argument("--size-limit", type=int, default=40)
    ap.add_argument("--num-of-clusters", type=int, default=3)
    ap.add_argument("--iter-limit", type=int, default=2)
    ap.add_argument("--chain-strength", type=float, default=20)
    # sampler
    ap.add_argument("--num-reads", type=int, default=None)
    ap.add_argument("--num-sweeps", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--device", type=int, default=None)
    return ap


def main(argv=None):
    a = _parser().parse_args(argv)
    methods = a.method or ["bqm"]
    if "all" in methods:
        methods = list(METHODS)

    from . import clustering, outputs
    from .graphs import create_graph, create_graph_csv
    from .sampler import MI355XSampler

    dirs_in = outputs.define_dirs(a.n, a.k, a.dim, a.ord, a.gamma, a.gamma_factor, a.custom, a.g_type, root=a.root)
    dirs = outputs.define_dirs(a.n, a.k, a.dim, a.ord, a.gamma, a.gamma_factor, a.custom, a.g_type,
                               root=a.out or a.root)
    path = a.graph or dirs_in["graph_in"]
    if path.endswith(".csv"):                                         # main.py:113-116
        G, pos = create_graph_csv({"graph_in_csv": path}, layout=a.layout)
    else:
        G, pos = create_graph(path, layout=a.layout)
    outputs.plot_and_save_graph_in(G, pos, dirs)                      # :119
    G, S, lengths = outputs.disconnected_components(G)                # :122-123

    sampler = MI355XSampler(device=a.device) if a.device is not None else MI355XSampler()
    skw = {}
    for name in ("num_reads", "num_sweeps", "seed"):
        if getattr(a, name) is not None:
            skw[name] = getattr(a, name)
    solver = "mi355x"

    def report(method, response, out, **extra):
        rec = {"method": method, "nodes": G.number_of_nodes(), "edges": G.number_of_edges(),
               "components": lengths[:8], "out": out}
        if response is not None:
            rec["energy"] = float(response.first.energy)
        rec.update(extra)
        print(json.dumps(rec))

    for m in methods:
        if m == "subsampling":                                        # :127-131
            r = clustering.graph_subsampling(G, 7, solver, sampler=sampler, sampler_kwargs=skw)
            outputs.plot_and_save_graph_out_mvc(G, pos, dirs)
            H = outputs.prune_graph(G, pos, dirs)
            report(m, r, dirs["graph_out_pru2"], kept=H.number_of_nodes())
        elif m == "subsampling_2":                                    # :128
            kept = clustering.graph_subsampling_2(G, 10, sampler=sampler, sampler_kwargs=skw)
            outputs.plot_and_save_graph_out_mvc(G, pos, dirs)
            report(m, None, dirs["graph_out_pru1"], kept=len(kept))
        elif m == "dqm":                                              # :135-136
            r = clustering.clustering_dqm(G, a.num_of_clusters, a.gamma, sampler=sampler, sampler_kwargs=skw)
            outputs.plot_and_save_graph_out_dqm(G, pos, dirs, r)
            report(m, r, dirs["graph_out_dqm"], sizes=_sizes(r.first.sample))
        elif m == "cqm":                                              # :139-140
            r = clustering.clustering_cqm(G, a.num_of_clusters, sampler=sampler, sampler_kwargs=skw)
            outputs.plot_and_save_graph_out_cqm(G, pos, dirs, r, a.num_of_clusters)
            report(m, r, dirs["graph_out_cqm"], sizes=_sizes(r.first.sample))
        elif m == "cqm_2":                                            # :143-145: the largest valid component
            H = max(S, key=len)
            for node in H.nodes:
                H.nodes[node]["subindex"] = G.nodes[node]["subindex"] if "subindex" in G.nodes[node] else None
            if any(H.nodes[v]["subindex"] is None for v in H.nodes):
                for sub, node in enumerate(H.nodes):
                    H.nodes[node]["subindex"] = sub
            r = clustering.clustering_cqm_2(H, a.num_of_clusters, sampler=sampler, sampler_kwargs=skw)
            outputs.plot_and_save_graph_out_cqm_2(H, None if pos is None else pos, dirs, r, a.num_of_clusters)
            report(m, r, dirs["graph_out_cqm"], sizes=_sizes(r.first.sample))
        elif m == "bqm":                                              # :148-150
            r = clustering.clustering_bqm(G, 1, dirs, solver, a.gamma_factor, a.color, a.terminate_on,
                                          a.size_limit, a.iter_limit, a.chain_strength, sampler=sampler,
                                          sampler_kwargs=skw)
            cut, uncut = outputs.plot_and_save_graph_out_bqm(G, pos, dirs)
            report(m, r, dirs["graph_out_bqm"], cut_edges=len(cut), uncut_edges=len(uncut))
        elif m == "bqm_2":                                            # :153-155
            r = clustering.clustering_bqm_2(G, 1, dirs, solver, 0.01, a.color, a.terminate_on, a.size_limit, 1, 1,
                                            sampler=sampler, sampler_kwargs=skw)
            cut, uncut = outputs.plot_and_save_graph_out_bqm(G, pos, dirs)
            report(m, r, dirs["graph_out_bqm"], cut_edges=len(cut), uncut_edges=len(uncut))
        elif m == "bqm_3":                                            # :159-161
            r = clustering.clustering_bqm_3(G, 1, dirs, solver, a.gamma_factor, a.color, a.terminate_on,
                                            a.size_limit, sampler=sampler, sampler_kwargs=skw)
            cut, uncut = outputs.plot_and_save_graph_out_bqm(G, pos, dirs)
            report(m, r, dirs["graph_out_bqm"], cut_edges=len(cut), uncut_edges=len(uncut))
    return 0


def _sizes(sample):
    out = {}
    for lab in sample.values():
        out[int(lab)] = out.get(int(lab), 0) + 1
    return [out[k] for k in sorted(out)]


if __name__ == "__main__":
    sys.exit(main())
