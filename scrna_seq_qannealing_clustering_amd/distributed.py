"""Multi-GPU: replicas shard across ranks, one process per GPU, ONE integer MIN all-reduce for the
global best.

Replicas are independent Markov chains, so the path shards with no data-path collective: rank ``r``
of ``G`` runs global replica ids ``[r * R_local, (r + 1) * R_local)`` on its own GPU with the model
replicated in its HBM; a replica's random stream is keyed by its GLOBAL id, so results do not depend
on ``G``.  The only exchange is at the end:

  (C1) all-reduce(MIN) of one packed 64-bit key per rank -- RCCL has no MINLOC, the packing makes an integer MIN do
       it (8 bytes per GPU over xGMI): the sortable bits of the fp64 energy with the lowest ceil(log2 num_reads)
       mantissa bits carrying the replica id (``pack_key64``; the fp32 key ``(sortable(float E) << 32) | global id``
       the device reduction writes stays available as ``pack_key`` / ``global_best``);
  (C2) broadcast of the winner's n labels + its exact fp64 energy (n + 8 bytes) from its owner rank.  The owner needs
       no second collective: replicas are sharded contiguously (``shard_range``), so the winning global id names its rank;
  (C3) parallel tempering only: ONE all-gather of the R energies per exchange round (``gather_energies``).

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


def pack_key64(energy: float, rel_id: int, id_bits: int) -> int:
    """Order-preserving packing of an fp64 energy AND a replica id into ONE unsigned 64-bit word: the sortable form of
    the double with its lowest ``id_bits`` mantissa bits replaced by the id (relative to the first id of the run).
    A MIN over such words is the lowest energy at 2^-(52 - id_bits) relative resolution (2^-37 = 7e-12 for 32768
    replicas; the fp32 key of ``pack_key`` resolves 6e-8, i.e. 0.008 at |E| = 1e5), ties go to the lowest id."""
    if not 0 <= int(rel_id) < (1 << id_bits):
        raise ValueError("replica id %d does not fit %d bits" % (rel_id, id_bits))
    b = int(np.float64(energy).view(np.uint64))
    s = (~b & 0xFFFFFFFFFFFFFFFF) if (b >> 63) else (b | _SIGN)
    return (s & ~((1 << id_bits) - 1) & 0xFFFFFFFFFFFFFFFF) | int(rel_id)


def unpack_key64(key: int, id_bits: int) -> Tuple[float, int]:
    """``(energy with its lowest id_bits mantissa bits cleared (in the sortable form), relative id)``."""
    rel = int(key) & ((1 << id_bits) - 1)
    s = int(key) & ~((1 << id_bits) - 1) & 0xFFFFFFFFFFFFFFFF
    b = (s & ~_SIGN) if (s & _SIGN) else (~s & 0xFFFFFFFFFFFFFFFF)
    return float(np.uint64(b).view(np.float64)), rel


def id_bits_for(num_reads: int) -> int:
    return max(1, int(num_reads - 1).bit_length())


def _collective_device(group, device):
    import torch
    import torch.distributed as dist
    if device is not None:
        return device
    return (torch.device("cuda", torch.cuda.current_device())
            if dist.get_backend(group) == "nccl" else torch.device("cpu"))


def _active(group) -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def global_best(local_key: int, local_state: np.ndarray, group=None, device=None,
                num_reads: Optional[int] = None, base: int = 0):
    """C1 + C2 on the fp32 key the device reduction writes (``Problem.best()[2]``, ``mi_sa_best``): between GPUs the
    comparison has fp32 resolution.  ``global_best_f64`` is the form the drivers use.  ``num_reads`` = replicas over
    ALL ranks, ``base`` = the first global id of the run.  Returns ``(energy_f32, global_replica_id, owner_rank,
    state)`` identical on every rank; the identity without a process group."""
    import torch
    import torch.distributed as dist
    if not _active(group):
        e, gid = unpack_key(local_key)
        return e, gid, 0, np.asarray(local_state).copy()
    if num_reads is None:                                   # (checked BEFORE any collective is issued)
        raise ValueError("global_best needs num_reads (replicas over all ranks) to name the winner's rank")
    world = dist.get_world_size(group)
    device = _collective_device(group, device)
    k = torch.tensor([_to_signed(local_key)], dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)                      # (C1)
    e, gid = unpack_key(_from_signed(int(k.item())))
    owner = owner_of(gid, num_reads, world, base)
    state, _ = _broadcast_winner(local_state, 0.0, owner, group, device)       # (C2)
    return e, gid, owner, state


def _broadcast_winner(local_state, local_energy, owner, group, device):
    """(C2) the winner's labels as they are stored (one byte per variable for binary states, two for Potts labels,
    viewed as bytes: RCCL reduces no uint16) followed by the 8 bytes of its exact fp64 energy, from its owner."""
    import torch
    import torch.distributed as dist
    local_state = np.ascontiguousarray(local_state)
    payload = np.concatenate([local_state.view(np.uint8).ravel(), np.array([local_energy], dtype=np.float64).view(np.uint8)])
    st = torch.from_numpy(payload).to(device)
    dist.broadcast(st, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
    out = st.cpu().numpy()
    return out[:-8].view(local_state.dtype).reshape(local_state.shape), float(out[-8:].view(np.float64)[0])


def global_best_f64(local_energy: float, local_gid: int, local_state: np.ndarray, num_reads: int, base: int = 0,
                    group=None, device=None):
    """C1 + C2 with the energy compared in fp64: ``local_energy`` / ``local_gid`` = this rank's best replica
    (``Problem.best()``: exact fp64 argmin on the device, ties to the lowest index), ``local_state`` its n labels.
    ONE 8-byte MIN all-reduce of ``pack_key64`` (the double's sortable bits with the lowest ceil(log2 num_reads)
    mantissa bits carrying the replica id: 2^-37 relative resolution at 32768 replicas, ties to the lowest id) and ONE
    broadcast of n + 8 bytes from the owner, whom the winning id names (``owner_of``): the winner's labels and its
    exact fp64 energy.  Returns ``(energy, global_replica_id, owner_rank, state)``, identical on every rank; the
    identity without a process group."""
    import torch
    import torch.distributed as dist
    num_reads = int(num_reads)
    rel = int(local_gid) - int(base)
    if rel < 0 or rel >= num_reads:                          # (before any collective)
        raise ValueError("replica id %d is outside the run [%d, %d)" % (local_gid, base, base + num_reads))
    if not _active(group):
        return float(local_energy), int(local_gid), 0, np.asarray(local_state).copy()
    world = dist.get_world_size(group)
    device = _collective_device(group, device)
    bits = id_bits_for(num_reads)
    k = torch.tensor([_to_signed(pack_key64(local_energy, rel, bits))], dtype=torch.int64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)                      # (C1)
    _, rel_best = unpack_key64(_from_signed(int(k.item())), bits)
    gid = int(base) + rel_best
    owner = owner_of(gid, num_reads, world, base)
    state, energy = _broadcast_winner(local_state, float(local_energy), owner, group, device)   # (C2)
    return energy, gid, owner, state


def gather_energies(local_energy, group=None, device=None, num_reads: Optional[int] = None):
    """(C3) all-gather of the per-replica energies in global replica order: the full SampleSet, and the exchange step
    of parallel tempering (one per round).  ``num_reads`` = replicas over all ranks: every rank's count then follows
    from ``shard_range`` and the gather is ONE collective (uneven shards are padded to the largest); without it the
    counts are gathered first (a second collective).  ``local_energy``: a numpy array (staged through ``device``), or
    a torch tensor already on the collective's device -- then nothing passes through the host and a tensor is returned."""
    import torch
    import torch.distributed as dist
    is_tensor = isinstance(local_energy, torch.Tensor)
    if not _active(group):
        return local_energy.clone() if is_tensor else np.asarray(local_energy, dtype=np.float64).copy()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = local_energy.device if is_tensor else _collective_device(group, device)
    n_mine = int(local_energy.shape[0])
    if num_reads is not None:
        counts = [hi - lo for lo, hi in (shard_range(num_reads, r, world) for r in range(world))]
        if counts[rank] != n_mine:
            raise ValueError("rank %d holds %d energies, its shard of %d replicas has %d" % (rank, n_mine, num_reads, counts[rank]))
    else:
        n_local = torch.tensor([n_mine], dtype=torch.int64, device=device)
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local, group=group)
        counts = [int(s.item()) for s in sizes]
    m = max(counts)
    src = local_energy.to(torch.float64) if is_tensor else torch.from_numpy(np.asarray(local_energy, dtype=np.float64)).to(device)
    if n_mine < m:
        src = torch.cat([src, torch.zeros(m - n_mine, dtype=torch.float64, device=device)])
    out = torch.empty(world * m, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group)
    if min(counts) < m:
        out = torch.cat([out[r * m: r * m + c] for r, c in enumerate(counts)])
    return out if is_tensor else out.cpu().numpy()
