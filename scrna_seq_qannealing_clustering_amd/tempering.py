"""Parallel tempering (replica exchange) on top of the anneal kernels -- BASELINE config 5.

R = T x C replicas: C independent chains of T temperature rungs.  A round is ``sweeps_per_round`` sweeps
with every replica at the constant beta of the rung it currently holds, continuing from the states left in HBM
(``MI_F_CONTINUE``) with the random stream advanced by ``sweep_offset``; then neighbouring rungs of each chain
propose to exchange with the Metropolis rule ``min(1, exp((beta_k - beta_{k+1}) (E_k - E_{k+1})))``, even pairs
on even rounds, odd pairs on odd rounds.

**Temperatures move, states do not**: an accepted exchange swaps the two replicas' rung indices only.  The
exchange is kernel K6 (`csrc/mi_sa.hip:k_pt_exchange`, C ABI ``mi_sa_tempering_*``): on one GPU the energies,
the rungs and the per-replica temperatures the next round anneals at never leave HBM -- a round is two kernel
launches and no copy.  Across GPUs the per-round exchange is ONE all-gather of the R energies
(``distributed.gather_energies``: RCCL from and into HBM on GPUs, gloo through the host in the CPU tests); every rank then runs the same exchange
kernel on the same energies with the same counter-based stream (seed, round), so no state ever crosses xGMI.

The driver below only sequences rounds; an *engine* supplies ``begin / round / exchange / energies / states /
rungs`` (``ProblemEngine`` = the GPU; the tests have one backed by the CPU oracle).
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
    """``engine.Problem`` behind the tempering driver: rounds and exchanges are kernels on the problem's stream."""

    def __init__(self, problem, seed: int, resync_interval: int = 0):
        self.problem = problem
        self.seed = int(seed)
        self.resync = int(resync_interval)

    def begin(self, ladder, chains, lo, hi):
        self.problem.tempering_begin(ladder, chains, lo, hi - lo)
        self._n_local, self._lo = hi - lo, lo

    def round(self, num_sweeps, sweep_offset, first, initial_states=None):
        self.problem.anneal(self._n_local, None, self.seed, replica_offset=self._lo,
                            initial_states=initial_states if first else None, resync_interval=self.resync,
                            sweep_offset=sweep_offset, continue_run=not first, num_sweeps=num_sweeps)

    def exchange(self, rnd, seed, all_energies=None):
        if all_energies is not None and not isinstance(all_energies, np.ndarray):      # a device tensor (RCCL all-gather)
            self.problem.tempering_exchange_device(rnd, seed, all_energies)
        else:
            self.problem.tempering_exchange(rnd, seed, all_energies)

    def energies_device(self):
        """This rank's energies as a torch tensor aliasing the library's HBM buffer (the all-gather's send buffer)."""
        return self.problem.device_energies()

    def energies(self) -> np.ndarray:
        return self.problem.fetch(states=False)[1]

    def states(self) -> np.ndarray:
        return self.problem.fetch(energies=False)[0]

    def rungs(self):
        return self.problem.tempering_state()


def parallel_tempering(engine, ladder, chains: int, rounds: int, sweeps_per_round: int, seed: int,
                       rank: int = 0, world: int = 1, group=None, initial_states: Optional[np.ndarray] = None,
                       history: bool = True):
    """Run PT on this rank's shard of the R = len(ladder) * chains replicas.

    Returns a dict: ``energies`` (all R, global order), ``rung`` (final rung of every replica),
    ``local_states`` (this rank's final states), ``best_energy`` / ``best_replica`` (global),
    ``swap_rate``, ``history`` (best energy after every round; ``history=False`` skips the per-round read of the
    energies -- on one GPU a round then involves no host copy at all)."""
    ladder = np.ascontiguousarray(ladder, dtype=np.float64)
    T = len(ladder)
    R = T * int(chains)
    lo, hi = D.shard_range(R, rank, world)
    engine.begin(ladder, int(chains), lo, hi)
    hist = []
    # across GPUs (RCCL): the per-round exchange of energies is ONE all-gather from and into HBM -- the engine hands out
    # a tensor aliasing its energy buffer, the gathered tensor goes to the exchange kernel as it is; gloo (CPU tests,
    # rehearsals on one GPU) stages through the host
    on_device = False
    if world > 1 and hasattr(engine, "energies_device"):
        import torch.distributed as dist
        on_device = dist.get_backend(group) == "nccl"

    def all_energies():
        if on_device:
            return D.gather_energies(engine.energies_device(), group=group, num_reads=R)       # (C3) R doubles, in HBM
        return D.gather_energies(engine.energies(), group=group, num_reads=R)                  # (C3)

    for rnd in range(int(rounds)):
        engine.round(int(sweeps_per_round), rnd * int(sweeps_per_round), rnd == 0, initial_states)
        last = rnd + 1 == rounds
        if world > 1:
            if last and not history:
                break
            energies = all_energies()
            if history:
                hist.append(float(energies.min()))
            if not last:
                engine.exchange(rnd, seed, energies)
        else:
            if history:
                hist.append(float(engine.energies().min()))
            if not last:
                engine.exchange(rnd, seed, None)
    if int(rounds) < 1:                          # no round ran: the initial rungs, no states
        return {"energies": np.zeros(0), "rung": np.arange(R, dtype=np.int64) % T, "local_range": (lo, hi),
                "local_states": None, "best_energy": None, "best_replica": None, "swap_rate": 0.0, "history": hist}
    energies = all_energies() if world > 1 else engine.energies()
    if not isinstance(energies, np.ndarray):
        energies = energies.cpu().numpy()
    rung, proposed, accepted = engine.rungs()
    best = int(np.argmin(energies))
    return {
        "energies": energies, "rung": rung, "local_range": (lo, hi), "local_states": engine.states(),
        "best_energy": float(energies[best]), "best_replica": best,
        "swap_rate": (accepted / proposed) if proposed else 0.0, "history": hist,
    }
