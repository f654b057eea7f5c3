#!/usr/bin/env python3
"""Where the per-call "upload" of a structured binary model goes (planning, padding, packing, device allocation)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                     # noqa: E402
from scrna_seq_qannealing_clustering_amd import models                           # noqa: E402
from scrna_seq_qannealing_clustering_amd.engine import Problem                   # noqa: E402

m, Qs, betas, _, G = bench.build_workload()
args = (m.rowptr, m.col, m.val.astype(np.float32), m.lin.astype(np.float32), float(np.float32(m.c_pair)))
for rep in range(3):
    t0 = time.perf_counter()
    pos, ns, _ = models.padded_slot_layout(m.rowptr, m.col)
    t1 = time.perf_counter()
    models.padded_slot_layout(m.rowptr, m.col, slot=128)
    t2 = time.perf_counter()
    rp, cc, vv = models.pad_csr(m.rowptr, m.col, args[2], pos, ns * 64)
    t3 = time.perf_counter()
    with Problem.csr_rank1(*args, order="padded", energy_model=(m.val, m.lin, m.c_pair), block="auto") as p:
        t4 = time.perf_counter()
    t5 = time.perf_counter()
    with Problem.csr_rank1(rp, cc, vv, np.zeros(ns * 64, np.float32), args[4]) as p:
        t6 = time.perf_counter()
    print("plan64 %.2f ms, plan128 %.2f, pad_csr %.2f, Problem.csr_rank1(padded, auto, fp64 model) %.2f, close %.2f, bare create %.2f"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3))
