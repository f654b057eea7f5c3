#!/usr/bin/env python3
"""Prototype: small graphs (n ~ 300, 5 slots, every slot keeps internal edges -> serial accept loop) against a PADDED
layout: more, partly filled slots so that no slot holds an edge; holes = isolated dummies with a huge linear term,
initial states given explicitly (zeros at the holes).  Timing only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scrna_seq_qannealing_clustering_amd import models
from scrna_seq_qannealing_clustering_amd.engine import Problem
from scrna_seq_qannealing_clustering_amd.graphs import graph_from_edges, synthetic_snn

def colour(rowptr, col, n, S):
    deg = np.diff(rowptr)
    order = np.lexsort((np.arange(n), -deg))
    where = -np.ones(n, dtype=np.int64); fill = np.zeros(S, dtype=np.int64)
    for v in order:
        nb = set(where[col[rowptr[v]:rowptr[v + 1]]].tolist())
        cand = [s for s in range(S) if fill[s] < 64 and s not in nb]
        if not cand:
            return None
        s = min(cand, key=lambda t: (fill[t], t))
        where[v] = s; fill[s] += 1
    return where

R, sweeps = int(os.environ.get("R", 500)), 1000
from scrna_seq_qannealing_clustering_amd import MI355XSampler
from scrna_seq_qannealing_clustering_amd.clustering import clustering_bqm
nodes, eu, ev, w, lab = synthetic_snn(2638)
G = graph_from_edges(nodes, eu, ev, w)
smp = MI355XSampler(); captured = []; orig = smp.sample_qubo
def traced(model, **kw):
    captured.append(model); return orig(model, **kw)
smp.sample_qubo = traced
clustering_bqm(G, 0, None, "mi355x", 0.05, 0, "iter_limit", 5, 3, 0, sampler=smp)     # the models a 4-level bisection solves
for m in captured[3:5] + captured[2:3] + captured[1:2]:
    n = m.num_variables
    betas = models.make_beta_schedule(sweeps, models.default_beta_range(m))
    with Problem.csr_rank1(m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32), float(np.float32(m.c_pair)), order="slots") as p:
        p.anneal(R, betas, 1); p.anneal(R, betas, 1)
        ms0 = p.kernel_ms(); st, en, _ = p.fetch(); k0 = p.kernel_name()
    S0 = (n + 63) // 64
    for S in range(S0, 4 * S0 + 8):
        where = colour(m.rowptr, m.col, n, S)
        if where is not None:
            break
    pos = np.empty(n, dtype=np.int64); cnt = np.zeros(S, dtype=np.int64)
    for v in range(n):
        pos[v] = where[v] * 64 + cnt[where[v]]; cnt[where[v]] += 1
    N = S * 64
    rows = np.repeat(np.arange(n), np.diff(m.rowptr))
    r2, c2 = pos[rows], pos[m.col]
    o = np.lexsort((c2, r2))
    rowptr2 = np.zeros(N + 1, dtype=np.int32); np.add.at(rowptr2, r2 + 1, 1); rowptr2 = np.cumsum(rowptr2).astype(np.int32)
    col2, val2 = c2[o].astype(np.int32), m.val[o].astype(np.float32)
    lin2 = np.full(N, 1e30, dtype=np.float32); lin2[pos] = m.lin
    init = np.zeros((R, N), dtype=np.uint8); init[:, pos] = np.random.RandomState(0).randint(0, 2, size=(R, n))
    with Problem.csr_rank1(rowptr2, col2, val2, lin2, float(np.float32(m.c_pair))) as p:
        p.anneal(R, betas, 1, initial_states=init); p.anneal(R, betas, 1, initial_states=init)
        ms1 = p.kernel_ms(); st1, en1, _ = p.fetch(); k1 = p.kernel_name()
    e1 = m.energies(st1[:, pos])
    print("n = %4d  R = %d: %d full slots (%s) %.2f ms, best %.3f | padded to %d slots (%s) %.2f ms, best %.3f" % (
        n, R, S0, k0, ms0, m.energies(st).min(), S, k1, ms1, e1.min()), flush=True)
