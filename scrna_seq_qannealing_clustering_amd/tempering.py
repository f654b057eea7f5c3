"""Parallel tempering (replica exchange) on top of the anneal kernels -- BASELINE config 5.

R = T x C replicas: C independent chains of T temperature rungs.  A round is ``sweeps_per_round`` sweeps
with every replica at the constant beta of the rung it currently holds (``MI_F_BETA_PER_REPLICA``),
continuing from the states left in HBM (``MI_F_CONTINUE``) with the random stream advanced by
``sweep_offset``; then neighbouring rungs of each chain propose to exchange with the Metropolis rule
``min(1, exp((beta_k - beta_{k+1}) (E_k - E_{k+1})))``, even pairs on even rounds, odd pairs on odd rounds.

**Temperatures move, states do not**: an accepted exchange swaps the two replicas' rung indices only.
Across GPUs the per-round exchange is ONE all-gather of the R energies (``distributed.gather_energies``,
RCCL on GPUs / gloo in the CPU tests); every rank then derives the identical exchange decisions from a
shared counter-based stream (seed, round), so no state ever crosses xGMI.

The exchange itself is O(R) host arithmetic on 8 bytes per replica; the sweeps are the kernels.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import distributed as D


def geometric_ladder(beta_hot: float, beta_cold: float, num_temps: int) -> np.ndarray:
    if num_temps < 2:
        raise ValueError("a tempering ladder needs at least two temperatures")
    return np.geomspace(float(beta_hot), float(beta_cold), int(num_temps))


class ProblemEngine:
    """Adapter: ``engine.Problem`` -> the three calls the tempering driver needs."""

    def __init__(self, problem, seed: int, resync_interval: int = 0):
        self.problem = problem
        self.seed = int(seed)
        self.resync = int(resync_interval)

    def round(self, betas_local, num_sweeps, sweep_offset, replica_offset, first, initial_states=None):
        self.problem.anneal(len(betas_local), betas_local, self.seed, replica_offset=replica_offset,
                            initial_states=initial_states if first else None, resync_interval=self.resync,
                            sweep_offset=sweep_offset, continue_run=not first, num_sweeps=num_sweeps)

    def energies(self) -> np.ndarray:
        return self.problem.fetch(states=False)[1]

    def states(self) -> np.ndarray:
        return self.problem.fetch(energies=False)[0]


def exchange_step(energies: np.ndarray, rung: np.ndarray, ladder: np.ndarray, num_temps: int, rnd: int,
                  seed: int):
    """One exchange phase over ALL replicas (identical on every rank).  ``rung[g]`` = ladder index held by
    global replica g; chain of g = g // num_temps.  Returns (new rung array, proposed, accepted)."""
    R = len(energies)
    chains = R // num_temps
    rung = rung.copy()
    energies = np.asarray(energies, dtype=np.float64)
    holder = np.empty((chains, num_temps), dtype=np.int64)          # holder[c, k] = replica holding rung k
    g = np.arange(R)
    holder[g // num_temps, rung] = g
    rs = np.random.RandomState([seed & 0x7FFFFFFF, (seed >> 31) & 0x7FFFFFFF, rnd & 0x7FFFFFFF, 0x5157])
    u = rs.random_sample((chains, num_temps))
    ks = np.arange(rnd & 1, num_temps - 1, 2)                       # disjoint pairs (k, k+1): all at once
    a, b = holder[:, ks], holder[:, ks + 1]
    arg = (ladder[ks] - ladder[ks + 1])[None, :] * (energies[a] - energies[b])
    acc = (arg >= 0.0) | (u[:, ks] < np.exp(np.minimum(arg, 0.0)))
    up = np.broadcast_to(ks[None, :], acc.shape)
    rung[a[acc]] = up[acc] + 1
    rung[b[acc]] = up[acc]
    proposed, accepted = int(acc.size), int(np.count_nonzero(acc))
    return rung, proposed, accepted


def parallel_tempering(engine, ladder, chains: int, rounds: int, sweeps_per_round: int, seed: int,
                       rank: int = 0, world: int = 1, group=None, initial_states: Optional[np.ndarray] = None):
    """Run PT on this rank's shard of the R = len(ladder) * chains replicas.

    Returns a dict: ``energies`` (all R, global order), ``rung`` (final rung of every replica),
    ``local_states`` (this rank's final states), ``best_energy`` / ``best_replica`` (global),
    ``swap_rate``, ``history`` (best energy after every round)."""
    ladder = np.ascontiguousarray(ladder, dtype=np.float64)
    T = len(ladder)
    R = T * int(chains)
    lo, hi = D.shard_range(R, rank, world)
    rung = np.arange(R, dtype=np.int64) % T
    proposed = accepted = 0
    history = []
    energies = None
    for rnd in range(int(rounds)):
        engine.round(ladder[rung[lo:hi]], int(sweeps_per_round), rnd * int(sweeps_per_round), lo, rnd == 0,
                     initial_states)
        energies = D.gather_energies(engine.energies(), group=group)          # (C3) R doubles
        history.append(float(energies.min()))
        if rnd + 1 < rounds:
            rung, p, a = exchange_step(energies, rung, ladder, T, rnd, seed)
            proposed += p
            accepted += a
    best = int(np.argmin(energies))
    return {
        "energies": energies, "rung": rung, "local_range": (lo, hi), "local_states": engine.states(),
        "best_energy": float(energies[best]), "best_replica": best,
        "swap_rate": (accepted / proposed) if proposed else 0.0, "history": history,
    }
