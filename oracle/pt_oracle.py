"""Replica-exchange step of parallel tempering, restated on the CPU.  TEST INFRASTRUCTURE ONLY.

Mirror of kernel K6 (scrna_seq_qannealing_clustering_amd/csrc/mi_sa.hip:k_pt_exchange), written as the plain loop
over chains and neighbouring rungs.  Replica g belongs to chain g // T and holds rung ``rung[g]``; round ``rnd``
proposes the pairs (k, k+1), k = rnd & 1, +2, ...; with a, b the holders of rungs k and k+1,

    arg = (beta_k - beta_{k+1}) * (E_a - E_b)                      (fp64)
    exchange iff  arg >= 0  or  -arg < neglog_u(word(i = chain*T + k, s = rnd, g = 0xffffffff, tag = 3))

which is Metropolis min(1, exp(arg)) with the chain's own logarithm (sa_oracle.c:orc_neglog_u) and Philox stream
(orc_chain_word).  The reference never exchanges anything (it calls a remote sampler once); the rule is the textbook
one (Swendsen-Wang 1986 / Geyer 1991 replica exchange), as BASELINE config 5 asks for.
"""
import numpy as np

from . import sa_oracle as so


def exchange_step(energies, rung, ladder, num_temps, rnd, seed):
    """Returns (new rung array, proposed, accepted)."""
    energies = np.asarray(energies, dtype=np.float64)
    ladder = np.asarray(ladder, dtype=np.float64)
    T = int(num_temps)
    R = len(energies)
    chains = R // T
    rung = np.array(rung, dtype=np.int64, copy=True)
    proposed = accepted = 0
    for c in range(chains):
        holder = np.empty(T, dtype=np.int64)
        for t in range(T):
            holder[rung[c * T + t]] = c * T + t
        for k in range(rnd & 1, T - 1, 2):
            a, b = int(holder[k]), int(holder[k + 1])
            arg = (ladder[k] - ladder[k + 1]) * (energies[a] - energies[b])
            proposed += 1
            if arg >= 0.0 or -arg < float(np.float32(so.neglog_u(so.chain_word(seed, c * T + k, rnd, 0xFFFFFFFF, 3)))):
                rung[a], rung[b] = k + 1, k
                accepted += 1
    return rung, proposed, accepted
