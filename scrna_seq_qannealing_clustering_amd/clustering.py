"""Reference-shaped clustering drivers: the callers of the sampler boundary, runnable with `MI355XSampler`.

Each function keeps the reference's name, positional signature, termination rules and node-attribute
contract (``label<iteration>`` for the BQM drivers, written through ``G.subgraph`` views to the parent
graph; the DQM driver returns the sampleset for `plot_and_save_graph_out_dqm` to turn into ``label1``):

    clustering_bqm     /root/reference/Python_Functions/BQM_clustering.py:25-204
    clustering_bqm_2   /root/reference/Python_Functions/BQM_clustering.py:206-351
    clustering_bqm_3   /root/reference/Python_Functions/BQM_clustering.py:353-427
    clustering_dqm     /root/reference/Python_Functions/DQM_clustering.py:24-47

Differences, all deliberate:
  * the sampler: every ``solver`` string ("hybrid", "fixed_embedding", "embedding_composite", "mi355x")
    is served by ``MI355XSampler`` (pass ``sampler=`` to supply another dimod-style sampler); the QPU
    keyword arguments the reference passes (``label``, ``chain_strength``, ``num_reads``,
    ``return_embedding``) are forwarded unchanged; the embedding JSON cache (:60-82) has no meaning here;
  * the model is built in array form (models.py) -- same coefficients as the reference's dict loops;
  * reference bug fixed: the recursive calls of `clustering_bqm` omit ``chain_strength`` (:129-130,
    :158-159, :175-176, :202-203) and raise TypeError on the first recursion; here it is passed on;
  * `clustering_bqm_3` indexes its constraint terms by node LABEL (``x[int(n)]``, :375), which only works
    for labels '0'..'n-1' in order; here the constraint always covers every node once;
  * printing is opt-in (``verbose=True`` reproduces the reference's console output).
"""
from __future__ import annotations

import random
from typing import Optional

import numpy as np

from .models import (RootGraphArrays, add_size_window_penalty, build_bqm2_qubo, build_bqm3_cut_qubo, build_bqm_qubo,
                     build_dqm_potts)


def _sampler(sampler):
    if sampler is not None:
        return sampler
    from .sampler import MI355XSampler
    return MI355XSampler()


def _print_top(response, verbose):
    if not verbose:
        return
    print('-' * 60)
    print('{:>15s}{:>15s}{:^15s}{:^15s}'.format('Set 0', 'Set 1', 'Energy', 'Num. of occurrences'))
    print('-' * 60)
    i = 0
    for sample, E, occur in response.data(fields=['sample', 'energy', "num_occurrences"]):
        S0 = [k for k, v in sample.items() if v == 0]
        S1 = [k for k, v in sample.items() if v == 1]
        print('{:>15s}{:>15s}{:^15s}{:^15s}'.format(str(S0), str(S1), str(E), str(occur)))
        if i > 3:
            break
        i = i + 1


def _split(G, response):
    lut = response.first.sample
    S0 = [node for node in G.nodes if not lut[node]]
    S1 = [node for node in G.nodes if lut[node]]
    return S0, S1


def _colour(G, nodes, label, lo, hi):
    col = random.randint(lo, hi)
    for i in nodes:
        G.nodes(data=True)[i][label] = col


def _solve(G, model, dirs, solver, sampler, num_reads, chain_strength, sampler_kwargs, pending=False):
    name_spec = ''.join([dirs["name"], "_", solver]) if dirs and "name" in dirs else solver
    kw = dict(label=name_spec)
    if solver != "hybrid":                       # :75 / :85 pass the QPU arguments, :57 only the label
        kw.update(chain_strength=chain_strength, num_reads=num_reads)
        if solver == "fixed_embedding":
            kw["return_embedding"] = True
    kw.update(sampler_kwargs or {})
    smp = _sampler(sampler)
    if pending:                                  # enqueue only (a sampler without the asynchronous entry: None)
        return smp.sample_qubo_async(model, **kw) if hasattr(smp, "sample_qubo_async") else None
    return smp.sample_qubo(model, **kw)


def clustering_bqm(G, iteration, dirs, solver, gamma_factor, color, terminate_on, size_limit, iter_limit,
                   chain_strength, sampler=None, sampler_kwargs: Optional[dict] = None, verbose=False, _arrays=None,
                   _prefetched=None):
    """Recursive 2-way partition with the balanced-cut QUBO (BQM_clustering.py:25-204).  ``_arrays``: the root graph's
    adjacency as arrays, built by the outermost call and handed down the recursion (models.RootGraphArrays).
    ``_prefetched``: (model, pending sampler call) of this subgraph, enqueued by the parent before it descended into the
    sibling -- the two halves of a bisection are independent, so the second one anneals while the first one's subtree is
    worked through (same results, same order of everything the caller sees; a sampler without ``sample_qubo_async`` runs
    the reference's sequence)."""
    if _arrays is None:
        _arrays = RootGraphArrays.of(G)
    model = _prefetched[0] if _prefetched else build_bqm_qubo(G, gamma_factor, k=8, arrays=_arrays)     # :29-47
    if verbose:
        print("gamma: ", model.info["gamma"])
        print("... Running on MI355X ...")
    response = (_prefetched[1].result() if _prefetched
                else _solve(G, model, dirs, solver, sampler, 500, chain_strength, sampler_kwargs))   # :52-85
    _print_top(response, verbose)                                     # :88-102
    label = "label" + str(iteration)                                  # :104
    S0, S1 = _split(G, response)                                      # :105-109
    if verbose:
        print("S0 length: ", len(S0))
        print("S1 length: ", len(S1))

    def recurse():
        G0, G1 = G.subgraph(S0), G.subgraph(S1)
        # the second half's call is enqueued before the first half's subtree is worked through (it does not depend on it)
        m1 = build_bqm_qubo(G1, gamma_factor, k=8, arrays=_arrays)
        p1 = _solve(G1, m1, dirs, solver, sampler, 500, chain_strength, sampler_kwargs, pending=True)
        for sub, pre in ((G0, None), (G1, (m1, p1) if p1 is not None else None)):
            clustering_bqm(sub, iteration + 1, dirs, solver, gamma_factor, color + 20,
                           terminate_on, size_limit, iter_limit, chain_strength, sampler=sampler,
                           sampler_kwargs=sampler_kwargs, verbose=verbose, _arrays=_arrays, _prefetched=pre)

    if terminate_on == "min_size":                                    # :113-130
        if len(S0) > size_limit and len(S1) > size_limit and iteration < iter_limit:
            _colour(G, S0, label, 0, 100)
            _colour(G, S1, label, 120, 220)
            recurse()
    elif terminate_on == "conf":                                      # :132-181
        energy = response.record.energy
        if len(energy) > 3:
            if energy[3] > 0.1 or energy[3] < -0.1:
                ratio = energy[0] / energy[3]
            else:
                if verbose:
                    print("error: 3rd lowest energy too close to zero; check your results")
                _colour(G, G.nodes, label, 0, 100)
                return response
            if verbose:
                print("energies", energy[:3])
                print("ratio:", ratio)
                print("difference:", np.abs(energy[0] - energy[3]))
            if ratio > 1.5 and min(len(S0), len(S1)) > 5 and iteration < iter_limit:
                _colour(G, S0, label, 0, 100)
                _colour(G, S1, label, 120, 220)
                recurse()
            _colour(G, G.nodes, label, 0, 100)                        # :160-163 (overwrites, as written)
            return response
        elif min(len(S0), len(S1)) > 5 and iteration < iter_limit:
            _colour(G, S0, label, 0, 100)
            _colour(G, S1, label, 120, 220)
            recurse()
        else:
            _colour(G, G.nodes, label, 0, 100)
            return response
    elif terminate_on == "once":                                      # :183-190
        _colour(G, S0, label, 0, 100)
        _colour(G, S1, label, 120, 220)
    elif terminate_on == "iter_limit":                                # :192-203
        if iteration < iter_limit:
            _colour(G, S0, label, 0, 100)
            _colour(G, S1, label, 120, 220)
            recurse()
    return


def clustering_bqm_2(G, iteration, dirs, solver, gamma_factor, color, terminate_on, size_limit, k,
                     chain_strength, sampler=None, sampler_kwargs: Optional[dict] = None, verbose=False, _arrays=None,
                     _prefetched=None):
    """Recursive 2-way partition with the linear-penalty QUBO (BQM_clustering.py:206-351).  As in the
    reference the ``chain_strength`` argument is replaced by mean(w) * mean(deg) * 2 (:220).  ``_prefetched``: as in
    :func:`clustering_bqm`."""
    if _arrays is None:
        _arrays = RootGraphArrays.of(G)
    model = _prefetched[0] if _prefetched else build_bqm2_qubo(G, gamma_factor, k, arrays=_arrays)      # :210-236
    chain_strength = model.info["chain_strength"]
    if verbose:
        print("gamma: ", model.info["gamma"])
        print("chain_strength: ", chain_strength)
    response = (_prefetched[1].result() if _prefetched
                else _solve(G, model, dirs, solver, sampler, 5000, chain_strength, sampler_kwargs))  # :240-273
    _print_top(response, verbose)
    label = "label" + str(iteration)
    S0, S1 = _split(G, response)
    if verbose:
        print("S0 length: ", len(S0))
        print("S1 length: ", len(S1))

    def recurse():                                                    # :317-318, :338-339
        G0, G1 = G.subgraph(S0), G.subgraph(S1)
        m1 = build_bqm2_qubo(G1, gamma_factor, k, arrays=_arrays)     # the second half enqueued ahead (see clustering_bqm)
        p1 = _solve(G1, m1, dirs, solver, sampler, 5000, m1.info["chain_strength"], sampler_kwargs, pending=True)
        for sub, pre in ((G0, None), (G1, (m1, p1) if p1 is not None else None)):
            clustering_bqm_2(sub, iteration + 1, dirs, solver, gamma_factor, color + 20,
                             terminate_on, size_limit, k, chain_strength, sampler=sampler,
                             sampler_kwargs=sampler_kwargs, verbose=verbose, _arrays=_arrays, _prefetched=pre)

    if terminate_on == "min_size":                                    # :302-318
        for i in S0:                                                  # deterministic colours here (:306, :311)
            G.nodes(data=True)[i][label] = 100 - color
        for i in S1:
            G.nodes(data=True)[i][label] = color - 100
        if len(S0) > size_limit and len(S1) > size_limit:
            recurse()
    elif terminate_on == "conf":                                      # :320-339 (absolute energy gap rule)
        energy = response.record.energy
        if len(energy) > 3:
            difference = np.abs(energy[0] - energy[3])
            if verbose:
                print("energies", energy[:10])
                print("difference:", difference)
            if difference > 10 and min(len(S0), len(S1)) > 5:
                _colour(G, S0, label, 0, 100)
                _colour(G, S1, label, 120, 220)
                recurse()
    elif terminate_on == "once":                                      # :341-350
        _colour(G, S0, label, 0, 100)
        _colour(G, S1, label, 120, 220)
        return response
    return


def clustering_bqm_3(G, iteration, dirs, solver, gamma_factor, color, terminate_on, size_limit,
                     sampler=None, sampler_kwargs: Optional[dict] = None, verbose=False):
    """Single bipartition with an explicit size window ``size_limit <= |S1| <= n/6`` enforced by slack
    bits (BQM_clustering.py:353-427; `add_linear_inequality_constraint` with lagrange = gamma)."""
    cut = build_bqm3_cut_qubo(G, k=8)                                 # :363-369
    n = cut.num_variables
    from .models import graph_arrays_and_weight
    gamma = gamma_factor * graph_arrays_and_weight(G)[4] / n          # :357-359
    model = add_size_window_penalty(cut, lb=size_limit, ub=n / 6, lagrange_multiplier=gamma)  # :373-380
    kw = dict(max_iter=1, num_reads=1, qpu_reads=100, tabu_timeout=200,
              qpu_params={'label': 'Notebook - Hybrid Computing 1'})  # :386 (Kerberos arguments)
    kw.pop("num_reads")                                               # Kerberos' 1 read would waste the GPU
    kw.update(sampler_kwargs or {})
    response = _sampler(sampler).sample(model, **kw)
    _print_top(response, verbose)
    label = "label" + str(iteration)
    S0, S1 = _split(G, response)                                      # slack variables are not graph nodes
    if verbose:
        print("S0 length: ", len(S0))
        print("S1 length: ", len(S1))
    _colour(G, S0, label, 0, 100)                                     # :419-425
    _colour(G, S1, label, 120, 220)
    return response


def clustering_dqm(G, num_of_clusters, gamma, sampler=None, sampler_kwargs: Optional[dict] = None,
                   verbose=False):
    """k-way clustering with the reference's DQM (DQM_clustering.py:24-47) solved in Potts form."""
    model = build_dqm_potts(G, num_of_clusters, gamma)                # :29-43
    kw = dict(label='DQM - scRAN-seq')                                # :45
    kw.update(sampler_kwargs or {})
    sampleset = _sampler(sampler).sample_dqm(model, **kw)
    if verbose:
        print("Energy: {}\nSolution: {}".format(sampleset.first.energy, sampleset.first.sample))   # :46
    return sampleset


def clustering_cqm(G, num_of_clusters, min_cluster_size: int = 20, sampler=None,
                   sampler_kwargs: Optional[dict] = None, verbose=False):
    """`clustering_cqm` (CQM_clustering.py:26-55): one-hot k-way model whose objective keeps heavy edges
    inside clusters, every cluster holding at least 20 cells.  The reference builds n*K binaries plus n + K
    constraints for the Leap hybrid CQM solver; here the one-hot constraints are the state space itself
    (one label per node) and the size constraints restrict the moves, so every returned sample is feasible.
    ``one_hot_sample(sampleset.first.sample, K)`` gives the reference's ``v_{i},{k}`` dictionary."""
    from .models import build_cqm_potts
    model = build_cqm_potts(G, num_of_clusters, min_cluster_size)     # :33-48
    kw = dict(label='CQM - scRAN-seq')                                # :53
    kw.update(sampler_kwargs or {})
    sampleset = _sampler(sampler).sample_dqm(model, **kw)
    if verbose:
        print("Energy: {}\nSolution: {}".format(sampleset.first.energy, sampleset.first.sample))   # :54
    return sampleset


def clustering_cqm_2(G, num_of_clusters, min_cluster_size: int = 20, sampler=None,
                     sampler_kwargs: Optional[dict] = None, verbose=False):
    """`clustering_cqm_2` (CQM_clustering.py:57-90): the same model on a SUBGRAPH whose nodes carry a
    ``"subindex"`` attribute (their position in the subgraph) -- the reference only uses it to name the
    binaries ``v_<subindex>,<k>``.  Solved exactly like :func:`clustering_cqm`;
    ``one_hot_sample(sample, K, G)`` names the binaries by subindex."""
    return clustering_cqm(G, num_of_clusters, min_cluster_size, sampler, sampler_kwargs, verbose)


def one_hot_sample(sample, num_of_clusters, G=None):
    """label dict {node: k}  ->  the CQM's binary dict {'v_<node>,<k>': 0/1} (CQM_clustering.py:34); with ``G``
    given, nodes are named by their ``"subindex"`` attribute (:65)."""
    if G is not None:
        return {"v_%s,%d" % (G.nodes[node]["subindex"], k): int(k == lab)
                for node, lab in sample.items() for k in range(num_of_clusters)}
    return {"v_%s,%d" % (node, k): int(k == lab) for node, lab in sample.items() for k in range(num_of_clusters)}


def graph_subsampling(G, gamma, solver="mi355x", sampler=None, sampler_kwargs: Optional[dict] = None):
    """`graph_subsampling` (QA_subsampling.py:25-96): the pruning QUBO, solved, ``label1`` written to the nodes
    (1 = kept)."""
    from .models import build_subsampling_qubo
    kw = dict(label="prun_data", chain_strength=4, num_reads=100)     # :37-38,42,56
    kw.update(sampler_kwargs or {})
    response = _sampler(sampler).sample_qubo(build_subsampling_qubo(G, gamma), **kw)
    lut = response.first.sample                                       # :80
    for node in G.nodes:                                              # :83-94
        G.nodes[node]["label1"] = 1 if lut[node] else 0
    return response


def graph_subsampling_2(G, gamma=None, sampler=None, sampler_kwargs: Optional[dict] = None, lagrange: float = 2.0):
    """`graph_subsampling_2` (QA_subsampling.py:98-118): a maximum independent set through the sampler, as
    ``dnx.maximum_independent_set(G, sampler=...)`` does it; returns the node list S and writes ``label1``."""
    from .models import build_mis_qubo
    kw = dict(num_reads=10, label='graph_subsampling_2', time_limit=3.0)      # :102
    kw.update(sampler_kwargs or {})
    response = _sampler(sampler).sample_qubo(build_mis_qubo(G, lagrange), **kw)
    sample = response.first.sample
    S = [node for node in G.nodes if sample[node] > 0]
    inside = set(S)
    for node in G.nodes:                                              # :110-116
        G.nodes[node]["label1"] = 1 if node in inside else 0
    return S
