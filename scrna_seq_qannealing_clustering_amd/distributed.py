"""Multi-GPU: replicas shard across ranks, one process per GPU, ONE integer MIN all-reduce for the
global best.

Replicas are independent Markov chains, so the path shards with no data-path collective: rank ``r``
of ``G`` runs global replica ids ``[r * R_local, (r + 1) * R_local)`` on its own GPU with the model
replicated in its HBM; a replica's random stream is keyed by its GLOBAL id, so results do not depend
on ``G``.  The only exchange is at the end:

  (C1) all-reduce(MIN) of one packed 64-bit key per rank  ``(sortable(float E_best) << 32) | global id``
       -- RCCL has no MINLOC, the packing makes an integer MIN do it (8 bytes per GPU over xGMI);
  (C2) broadcast of the winner's n labels (n bytes) from its owner rank.  The owner needs no second
       collective: replicas are sharded contiguously (``shard_range``), so the winning global id names its rank.

``torch.distributed`` is the transport (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

``HSA_ENABLE_IPC_MODE_LEGACY=0`` (dmabuf IPC: what RCCL needs on this driver) only takes effect when it is in
the environment BEFORE the process makes its first HIP call, so it is set here at import time; a launcher
that touches the GPU before importing this module has to export it itself.
"""
from __future__ import annotations

import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from typing import Optional, Tuple      # noqa: E402

import numpy as np                      # noqa: E402

_SIGN = 1 << 63


def pack_key(energy: float, global_id: int) -> int:
    """Order-preserving packing of (float32(energy), id) into an unsigned 64-bit integer."""
    b = int(np.float32(energy).view(np.uint32))
    s = (~b & 0xFFFFFFFF) if (b & 0x80000000) else (b | 0x80000000)
    return (s << 32) | (int(global_id) & 0xFFFFFFFF)


def unpack_key(key: int) -> Tuple[float, int]:
    s = (key >> 32) & 0xFFFFFFFF
    b = (s & 0x7FFFFFFF) if (s & 0x80000000) else (~s & 0xFFFFFFFF)
    return float(np.uint32(b).view(np.float32)), int(key & 0xFFFFFFFF)


def _to_signed(key: int) -> int:
    """uint64 order -> int64 order (torch has no uint64 reductions): flip the top bit."""
    k = key ^ _SIGN
    return k - (1 << 64) if k >= _SIGN else k


def _from_signed(k: int) -> int:
    return (k + (1 << 64) if k < 0 else k) ^ _SIGN


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (as torchrun sets them).
    Returns (rank, world_size, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(num_reads: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous global replica ids [lo, hi) of a rank; remainders go to the low ranks."""
    base, rem = divmod(int(num_reads), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(global_id: int, num_reads: int, world: int, base: int = 0) -> int:
    """Rank that ran global replica ``global_id`` when ``num_reads`` replicas with ids ``base ..`` were sharded by
    ``shard_range`` (remainders to the low ranks)."""
    g = int(global_id) - int(base)
    q, rem = divmod(int(num_reads), int(world))
    if g < 0 or g >= num_reads:
        raise ValueError("replica id %d is outside the run [%d, %d)" % (global_id, base, base + num_reads))
    if q == 0:
        return g
    cut = rem * (q + 1)                         # the first `rem` ranks hold q + 1 replicas each
    return g // (q + 1) if g < cut else rem + (g - cut) // q


def global_best(local_key: int, local_state: np.ndarray, group=None, device=None,
                num_reads: Optional[int] = None, base: int = 0):
    """C1 + C2.  ``local_key`` = this rank's packed best key (``Problem.best()[2]``), ``local_state``
    its n labels.  ``num_reads`` = replicas over ALL ranks, ``base`` = the first global id of the run (the
    winner's rank follows from its id and the contiguous sharding, ``owner_of``).  Returns
    ``(energy_f32, global_replica_id, owner_rank, state)`` identical on every rank.  Without an initialised
    process group it is the identity.  ONE 8-byte MIN all-reduce and ONE n-byte broadcast."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        e, gid = unpack_key(local_key)
        return e, gid, 0, np.asarray(local_state).copy()
    world = dist.get_world_size(group)
    if device is None:
        device = (torch.device("cuda", torch.cuda.current_device())
                  if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    k = torch.tensor([_to_signed(local_key)], dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)                      # (C1)
    best = _from_signed(int(k.item()))
    e, gid = unpack_key(best)
    if num_reads is None:
        raise ValueError("global_best needs num_reads (replicas over all ranks) to name the winner's rank")
    owner = owner_of(gid, num_reads, world, base)
    local_state = np.ascontiguousarray(local_state)
    # labels travel as they are stored: one byte per variable for binary states, two for Potts labels
    # (viewed as bytes: RCCL reduces no uint16)
    st = torch.from_numpy(local_state.view(np.uint8).copy()).to(device)
    dist.broadcast(st, src=dist.get_global_rank(group, owner) if group is not None else owner,
                   group=group)                                                 # (C2)
    return e, gid, owner, st.cpu().numpy().view(local_state.dtype)


def gather_energies(local_energy: np.ndarray, group=None, device=None) -> np.ndarray:
    """All-gather of the per-replica energies (R floats per rank) in global replica order; used for
    the full SampleSet and for the exchange step of parallel tempering."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(local_energy, dtype=np.float64).copy()
    world = dist.get_world_size(group)
    if device is None:
        device = (torch.device("cuda", torch.cuda.current_device())
                  if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    n_local = torch.tensor([len(local_energy)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    m = max(int(s.item()) for s in sizes)
    buf = torch.zeros(m, dtype=torch.float64, device=device)
    buf[: len(local_energy)] = torch.from_numpy(np.asarray(local_energy, dtype=np.float64)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return np.concatenate([o[: int(s.item())].cpu().numpy() for o, s in zip(out, sizes)])
