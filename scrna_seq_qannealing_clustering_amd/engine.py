"""Thin object layer over the C ABI: one ``Problem`` = one model resident in HBM on one GPU."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib


_POTTS_MAX_N = 40000          # mi_sa_problem_create_potts_csr_f32's limit


def layout_block_for(n: int, num_reads: int, max_degree: int = 16):
    """Seats per edge-free block the sampler lays a structured binary model out in: 64 (one wavefront sweeps a slot per
    step), or ``"auto"`` for the few reads the reference asks for (``num_reads = 500``, BQM_clustering.py:52), where
    every wavefront has a SIMD to itself and a run takes (steps per sweep) x (one wavefront's time per step):
    ``Problem.csr_rank1`` then plans the 64-seat AND the 128-seat layout and takes the wider one when it needs few
    enough blocks -- one wavefront sweeps such a block per step (K2w, csrc/sparse_split_kernels.hip: with the thresholds
    coming from the workgroup's second wavefront a two-slot step takes 1.86 times a one-slot step) -- which large sparse
    graphs do (n = 2638: 22 blocks against 42 slots) and small or clustered ones, whose block count is set by the
    colours they need, do not."""
    if num_reads > 1024 or max_degree > 32 or n > 16384 or n < 1024:     # (below ~1000 variables the colours decide)
        return 64
    return "auto"


# a two-slot step of the wide kernel against a one-slot step (MI355X, a wavefront alone on its SIMD, thresholds from the
# second wavefront of its workgroup, field sums staged behind counted LDS waits: 439 ns per 128-seat block, 236 ns per
# 64-seat slot at n = 2638, 500 reads)
_WIDE_STEP_RATIO = 439.0 / 236.0


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class Problem:
    """A model uploaded to one MI355X.  ``anneal`` is asynchronous; ``fetch``/``best`` wait."""

    def __init__(self, handle, kind, n, num_cases, device, perm=None, seats=None, n_dev=None):
        self._h = handle
        self.kind = kind
        self.n = n                       # variables of the CALLER's model
        self.num_cases = num_cases
        self.device = device
        self._last = None
        self.perm = perm                 # device variable j is the caller's variable perm[j] (None: identity / padded)
        # device column of the caller's variable i (None: identity), and the device-side variable count: larger than
        # n under order="padded", where the seats no variable sits in are holes that stay 0
        self._inv = None if perm is None else np.argsort(perm)
        if seats is not None:
            self._inv = np.asarray(seats, dtype=np.int64)
        self.n_dev = int(n if n_dev is None else n_dev)

    # -- constructors ---------------------------------------------------------------------------
    @classmethod
    def dense(cls, Qs: np.ndarray, offset: float = 0.0, device: int = 0) -> "Problem":
        """``E(x) = x^T Qs x + offset`` with Qs symmetric (diag = linear terms)."""
        Qs = np.ascontiguousarray(Qs, dtype=np.float32)
        if Qs.ndim != 2 or Qs.shape[0] != Qs.shape[1]:
            raise ValueError("Qs must be a square matrix")
        if not np.array_equal(Qs, Qs.T):
            raise ValueError("Qs must be symmetric (use (Q + Q.T)/2 with the diagonal kept)")
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.mi_sa_problem_create_dense_f32(_ptr(Qs, C.c_float), Qs.shape[0],
                                                      float(offset), int(device), C.byref(h)))
        return cls(h, _lib.KIND_DENSE, Qs.shape[0], 2, device)

    @classmethod
    def csr_rank1(cls, rowptr, col, val, lin, c_pair: float, offset: float = 0.0,
                  device: int = 0, order: Optional[str] = None, energy_model=None, block=64, weights=None) -> "Problem":
        """``order="slots"`` renumbers the variables on the device so that the 64 variables a wavefront
        sweeps together are (as far as possible) mutually non-adjacent -- the kernel's integer fast path
        (models.slot_independent_order).  States go in and come out in the CALLER's order either way; the
        chain is a different (equally valid) sweep order, so results differ from ``order=None`` runs.
        ``order="padded"``: the same with holes allowed -- as many blocks of 64 seats as it takes to keep EVERY edge
        between blocks (models.padded_slot_layout); what small or strongly clustered graphs need (the subgraphs of the
        reference's recursive bisection, its 256-node benchmark graphs), where no packed order is edge-free.

        ``block`` (64, 128, 256 or "auto"; ``order="padded"`` only): seats per edge-free block.  128 / 256 lay the model
        out for the few-replica kernels (csrc/sparse_split_kernels.hip: one wavefront sweeps a whole block per step, K2w,
        or -- option ``k2_wide`` = 2 -- a workgroup of 2 / 4 wavefronts does, K2s), which the library picks for runs of up
        to 1024 replicas; "auto" plans both the 64- and the 128-seat layout and keeps the faster; ``layout_block_for`` is
        the sampler's choice.

        ``energy_model=(val64, lin64, c_pair64)``: the caller's fp64 coefficients (same CSR structure); the
        reported energies are then evaluated on the device in that model (the chain itself runs in fp32).

        ``weights`` (positive integers, ``order="padded"`` only): weights of the pair term, ``c_pair a_i a_j`` on pair
        (i, j) -- the slack bits of a squared linear constraint (models.add_size_window_penalty).  The variables whose
        weight is not 1 (at most 64, without sparse couplings) get a 64-seat slot of their own behind the others."""
        if weights is not None:
            weights = np.asarray(weights, dtype=np.int64)
            if np.all(weights == 1):
                weights = None
        if weights is not None:
            return cls._csr_rank1_weighted(rowptr, col, val, lin, c_pair, offset, device, order, energy_model, weights)
        perm = None
        val64 = lin64 = None
        if energy_model is not None:
            val64, lin64 = np.asarray(energy_model[0], dtype=np.float64), np.asarray(energy_model[1], dtype=np.float64)
        if order == "slots":
            from .models import permute_csr, slot_independent_order
            perm = slot_independent_order(rowptr, col)
            if val64 is not None:
                rowptr, col, val, val64 = permute_csr(rowptr, col, val, perm, also=val64)
                lin64 = lin64[perm]
            else:
                rowptr, col, val = permute_csr(rowptr, col, val, perm)
            lin = np.asarray(lin)[perm]
        elif order == "padded":
            # seats with holes (models.padded_slot_layout): the fewest blocks of 64 that keep every edge BETWEEN blocks.
            # A hole is a variable without couplings whose linear term is +inf: the kernels start it at 0 and
            # never flip it; its fp64 energy coefficients are 0.
            from .models import pad_csr, padded_slot_layout
            if block == "auto":
                # both layouts planned (a millisecond each); the 128-seat one when its steps take less time in all
                s64 = padded_slot_layout(rowptr, col, slot=64)
                s128 = padded_slot_layout(rowptr, col, slot=128)
                wide = s128[2] == 0 and s64[2] == 0 and (s128[1] + (s128[1] & 1)) * _WIDE_STEP_RATIO < s64[1]
                block = 128 if wide else 64
                seats, nslots, clashes = s128 if wide else s64
            elif block in (64, 128, 256):
                seats, nslots, clashes = padded_slot_layout(rowptr, col, slot=block)
            else:
                raise ValueError("block must be 64, 128, 256 or 'auto'")
            if block > 64 and clashes:               # no edge-free layout in blocks this wide: the 64-seat layout
                return cls.csr_rank1(rowptr, col, val, lin, c_pair, offset, device, "padded", energy_model)
            # rows wider than 64 neighbours (untrimmed SNN graphs) run on the runtime-width kernel, whose cost per slot is
            # the adjacency it streams whether a seat holds a variable or not: a layout that needs many more slots than the
            # packed one costs more than the serial accept loop it avoids (degree 110: 82 slots against 42, 1.4e10
            # against 2.6e10 updates/s) -- the packed slot-independent order there
            if int(np.diff(np.asarray(rowptr)).max(initial=0)) > 64 and nslots * block * 5 > ((len(lin) + 63) // 64) * 64 * 6:
                return cls.csr_rank1(rowptr, col, val, lin, c_pair, offset, device, "slots", energy_model)
            if block == 128 and nslots % 2:          # (the two-wavefront kernel takes whole groups of four 64-seat slots)
                nslots += 1
            n_caller, n_dev = len(lin), nslots * block
            if n_dev > (1 << 20) >= n_caller:        # the holes would push the model over the kernel's size limit
                return cls.csr_rank1(rowptr, col, val, lin, c_pair, offset, device, "slots", energy_model)
            if val64 is not None:
                rowptr, col, val, val64 = pad_csr(rowptr, col, val, seats, n_dev, also=val64)     # (one sort for both)
                l64 = np.zeros(n_dev, dtype=np.float64)
                l64[seats] = lin64
                lin64 = l64
            else:
                rowptr, col, val = pad_csr(rowptr, col, val, seats, n_dev)
            lpad = np.full(n_dev, np.inf, dtype=np.float32)
            lpad[seats] = np.asarray(lin, dtype=np.float32)
            lin = lpad
        elif order is not None:
            raise ValueError("order must be None, 'slots' or 'padded'")
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        lin = np.ascontiguousarray(lin, dtype=np.float32)
        n = len(lin)
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.mi_sa_problem_create_csr_rank1_f32(
            _ptr(rowptr, C.c_int32), _ptr(col, C.c_int32), _ptr(val, C.c_float),
            _ptr(lin, C.c_float), float(c_pair), n, float(offset), int(device), C.byref(h)))
        if order == "padded":
            prob = cls(h, _lib.KIND_CSR_RANK1, n_caller, 2, device, seats=seats, n_dev=n_dev)
        else:
            prob = cls(h, _lib.KIND_CSR_RANK1, n, 2, device, perm=perm)
        if val64 is not None:
            prob._set_energy_model(val64, lin64, float(energy_model[2]), len(val))
        return prob

    @classmethod
    def _csr_rank1_weighted(cls, rowptr, col, val, lin, c_pair, offset, device, order, energy_model, weights) -> "Problem":
        """The padded layout of a model with pair-term weights: the unit-weight variables as usual (64-seat slots free of
        internal edges), the others -- few, no sparse couplings -- in one more slot; mi_sa_problem_set_pair_weights."""
        from .models import pad_csr, padded_slot_layout
        if order != "padded":
            raise ValueError("a model with pair-term weights needs order='padded'")
        rowptr = np.asarray(rowptr, dtype=np.int64)
        col = np.asarray(col, dtype=np.int64)
        n_caller = len(lin)
        deg = np.diff(rowptr)
        heavy = np.flatnonzero(weights != 1)
        light = np.flatnonzero(weights == 1)
        if np.any(weights < 1) or len(heavy) > 64 or np.any(deg[heavy] != 0):
            raise ValueError("pair-term weights: positive integers; at most 64 variables with a weight other than 1, "
                             "and those without sparse couplings")
        # the layout of the unit-weight variables alone (the others have no edges): their CSR renumbered 0 .. len(light) - 1
        renum = np.full(n_caller, -1, dtype=np.int64)
        renum[light] = np.arange(len(light))
        rp_l = np.concatenate([[0], np.cumsum(deg[light])]).astype(np.int32)
        keep = np.repeat(weights == 1, deg)
        seats_l, nslots, _ = padded_slot_layout(rp_l, renum[col[keep]].astype(np.int32), slot=64)
        seats = np.empty(n_caller, dtype=np.int64)
        seats[light] = seats_l
        seats[heavy] = nslots * 64 + np.arange(len(heavy))
        n_dev = (nslots + 1) * 64
        val64 = lin64 = None
        if energy_model is not None:
            val64, lin64 = np.asarray(energy_model[0], dtype=np.float64), np.asarray(energy_model[1], dtype=np.float64)
            rp, cc, vv, val64 = pad_csr(rowptr, col, val, seats, n_dev, also=val64)
            l64 = np.zeros(n_dev, dtype=np.float64)
            l64[seats] = lin64
            lin64 = l64
        else:
            rp, cc, vv = pad_csr(rowptr, col, val, seats, n_dev)
        lpad = np.full(n_dev, np.inf, dtype=np.float32)
        lpad[seats] = np.asarray(lin, dtype=np.float32)
        rp = np.ascontiguousarray(rp, dtype=np.int32)
        cc = np.ascontiguousarray(cc, dtype=np.int32)
        vv = np.ascontiguousarray(vv, dtype=np.float32)
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.mi_sa_problem_create_csr_rank1_f32(
            _ptr(rp, C.c_int32), _ptr(cc, C.c_int32), _ptr(vv, C.c_float),
            _ptr(lpad, C.c_float), float(c_pair), n_dev, float(offset), int(device), C.byref(h)))
        prob = cls(h, _lib.KIND_CSR_RANK1, n_caller, 2, device, seats=seats, n_dev=n_dev)
        wdev = np.ones(n_dev, dtype=np.int32)
        wdev[seats] = weights
        rc = lib.mi_sa_problem_set_pair_weights(h, _ptr(wdev, C.c_int32))
        if rc:
            prob.close()
            _lib.check(rc)
        if val64 is not None:
            prob._set_energy_model(val64, lin64, float(energy_model[2]), len(vv))
        return prob

    @classmethod
    def potts_csr(cls, rowptr, col, val, c_pair: float, n: int, num_cases: int,
                  lin_offset: float = 0.0, device: int = 0, order: Optional[str] = None,
                  energy_model=None) -> "Problem":
        """``order="slots"`` / ``"padded"``: as in :meth:`csr_rank1` (labels go in and come out in the caller's order).
        ``energy_model=(val64, c_pair64)``: fp64 coefficients for the reported energies."""
        perm = None
        val64 = None if energy_model is None else np.asarray(energy_model[0], dtype=np.float64)
        if order == "slots":
            from .models import permute_csr, slot_independent_order
            perm = slot_independent_order(rowptr, col)
            if val64 is not None:
                rowptr, col, val, val64 = permute_csr(rowptr, col, val, perm, also=val64)
            else:
                rowptr, col, val = permute_csr(rowptr, col, val, perm)
        elif order == "padded":
            # seats with holes, as in csr_rank1; a hole of a Potts model is marked through mi_sa_problem_set_absent:
            # label 0, in no cluster, never proposed
            from .models import pad_csr, padded_slot_layout
            seats, nslots, _ = padded_slot_layout(rowptr, col)
            if nslots * 64 > _POTTS_MAX_N >= int(n):  # the holes would push the model over the kernel's size limit (its
                # LDS budget is checked against the padded size): the packed order, clashing slots on the general path
                return cls.potts_csr(rowptr, col, val, c_pair, n, num_cases, lin_offset, device, "slots", energy_model)
            n_caller, n = int(n), nslots * 64
            if val64 is not None:
                rowptr, col, val, val64 = pad_csr(rowptr, col, val, seats, n, also=val64)
            else:
                rowptr, col, val = pad_csr(rowptr, col, val, seats, n)
        elif order is not None:
            raise ValueError("order must be None, 'slots' or 'padded'")
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float32)
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.mi_sa_problem_create_potts_csr_f32(
            _ptr(rowptr, C.c_int32), _ptr(col, C.c_int32), _ptr(val, C.c_float), float(c_pair),
            int(n), int(num_cases), float(lin_offset), int(device), C.byref(h)))
        if order == "padded":
            prob = cls(h, _lib.KIND_POTTS_CSR, n_caller, int(num_cases), device, seats=seats, n_dev=n)
            absent = np.ones(n, dtype=np.uint8)
            absent[seats] = 0
            rc = lib.mi_sa_problem_set_absent(h, absent.ctypes.data_as(C.POINTER(C.c_uint8)))
            if rc:
                prob.close()
                _lib.check(rc)
        else:
            prob = cls(h, _lib.KIND_POTTS_CSR, int(n), int(num_cases), device, perm=perm)
        if val64 is not None:
            prob._set_energy_model(val64, None, float(energy_model[1]), len(val))
        return prob

    def _set_energy_model(self, val64, lin64, c_pair64, nnz):
        if len(val64) != nnz or (lin64 is not None and len(lin64) != self.n_dev):
            self.close()
            raise ValueError("energy_model must have the structure of the fp32 model")
        val64 = np.ascontiguousarray(val64, dtype=np.float64)
        lin64 = None if lin64 is None else np.ascontiguousarray(lin64, dtype=np.float64)
        try:
            _lib.check(_lib.load().mi_sa_problem_set_energy_model_f64(
                self._h, _ptr(val64, C.c_double), None if lin64 is None else _ptr(lin64, C.c_double), float(c_pair64)))
        except Exception:
            self.close()
            raise

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if self._h is not None and self._h.value:
            _lib.load().mi_sa_problem_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key: str, value: int):
        _lib.check(_lib.load().mi_sa_set_option(self._h, key.encode(), int(value)))

    def debug_pace(self, words: int = 288):
        out = np.zeros(words, dtype=np.uint32)
        _lib.check(_lib.load().mi_sa_debug_pace(self._h, _ptr(out, C.c_uint32), words))
        return out

    def debug_stats(self, words: int = 16):
        out = np.zeros(words, dtype=np.uint64)
        _lib.check(_lib.load().mi_sa_debug_stats(self._h, _ptr(out, C.c_uint64), words))
        return out

    # -- the anneal -----------------------------------------------------------------------------
    @property
    def state_dtype(self):
        return np.uint16 if self.kind == _lib.KIND_POTTS_CSR else np.uint8

    def anneal(self, num_reads: int, betas, seed: int, replica_offset: int = 0,
               initial_states: Optional[np.ndarray] = None, resync_interval: int = 0,
               sweep_offset: int = 0, continue_run: bool = False, num_sweeps: Optional[int] = None):
        """``betas``: one per sweep (default), or -- when ``num_sweeps`` is given -- one per REPLICA, held
        constant for ``num_sweeps`` sweeps (a tempering rung); ``betas=None`` with ``num_sweeps``: every replica
        at the temperature the device-side tempering state holds for it (``tempering_begin`` / ``_exchange``).
        ``continue_run`` starts from the states the previous call left on the device; ``sweep_offset`` continues
        its random stream."""
        resident = betas is None
        if resident and num_sweeps is None:
            raise ValueError("betas=None (temperatures resident on the device) needs num_sweeps")
        betas = None if resident else np.ascontiguousarray(betas, dtype=np.float64)
        per_replica = num_sweeps is not None
        if per_replica and not resident and len(betas) != num_reads:
            raise ValueError("per-replica betas need one entry per replica")
        flags = (1 if continue_run else 0) | (2 if per_replica else 0) | (4 if resident else 0)
        sweeps = int(num_sweeps) if per_replica else len(betas)
        init = None
        if initial_states is not None:
            init = np.ascontiguousarray(initial_states, dtype=self.state_dtype)
            if init.shape != (num_reads, self.n):
                raise ValueError("initial_states must have shape (num_reads, n) = (%d, %d)"
                                 % (num_reads, self.n))
            if self._inv is not None:
                dev = np.zeros((num_reads, self.n_dev), dtype=self.state_dtype)
                dev[:, self._inv] = init
                init = dev
        _lib.check(_lib.load().mi_sa_anneal_ex(
            self._h, int(num_reads), C.c_uint32(int(replica_offset) & 0xFFFFFFFF), sweeps,
            _ptr(betas, C.c_double), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
            init.ctypes.data_as(C.c_void_p) if init is not None else None, int(resync_interval),
            C.c_uint32(int(sweep_offset) & 0xFFFFFFFF), C.c_uint32(flags)))
        self._last = (int(num_reads), sweeps)

    # -- parallel tempering: exchange step on the device (K6) ----------------------------------------
    def tempering_begin(self, ladder_betas, chains: int, first_replica: int, num_local: int):
        lad = np.ascontiguousarray(ladder_betas, dtype=np.float64)
        self._pt_total = len(lad) * int(chains)
        _lib.check(_lib.load().mi_sa_tempering_begin(self._h, _ptr(lad, C.c_double), len(lad), int(chains),
                                                     C.c_uint32(int(first_replica)), int(num_local)))

    def tempering_exchange(self, rnd: int, seed: int, all_energies: Optional[np.ndarray] = None):
        """Neighbouring rungs of every chain exchange (Metropolis on a counter-based stream of (seed, round)).
        ``all_energies`` = None when this GPU owns every replica (nothing leaves HBM), else the all-gathered
        energies of the run in global replica order."""
        en = None if all_energies is None else np.ascontiguousarray(all_energies, dtype=np.float64)
        if en is not None and len(en) != self._pt_total:
            raise ValueError("all_energies must hold the %d energies of the run" % self._pt_total)
        _lib.check(_lib.load().mi_sa_tempering_exchange(self._h, C.c_uint32(int(rnd) & 0xFFFFFFFF),
                                                        C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _ptr(en, C.c_double)))

    def tempering_exchange_device(self, rnd: int, seed: int, all_energies_dev):
        """The same exchange with the all-gathered energies already in HBM: ``all_energies_dev`` = a contiguous float64
        torch tensor on this problem's GPU (the output of the RCCL all-gather), read in place."""
        if all_energies_dev.numel() != self._pt_total or not all_energies_dev.is_contiguous():
            raise ValueError("all_energies_dev must hold the %d energies of the run, contiguous" % self._pt_total)
        _lib.check(_lib.load().mi_sa_tempering_exchange_dev(self._h, C.c_uint32(int(rnd) & 0xFFFFFFFF),
                                                            C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                                            C.c_void_p(all_energies_dev.data_ptr())))

    def device_energies(self):
        """The energies of the last run as a float64 torch tensor that ALIASES the library's HBM buffer (no copy;
        waits for the run; valid until the next anneal on this handle): the send buffer of a GPU-to-GPU collective."""
        import torch
        d_en, R = C.c_void_p(), C.c_int(0)
        _lib.check(_lib.load().mi_sa_device_results(self._h, None, C.byref(d_en), C.byref(R)))

        class _Alias:                     # the array-interface protocol torch.as_tensor understands for device memory
            __cuda_array_interface__ = {"shape": (int(R.value),), "typestr": "<f8", "data": (int(d_en.value), False),
                                        "version": 2, "strides": None}
        return torch.as_tensor(_Alias(), device=torch.device("cuda", self.device))

    def tempering_state(self):
        """``(rung of every replica of the run, exchanges proposed, exchanges accepted)``."""
        rung = np.empty(self._pt_total, dtype=np.int32)
        prop, acc = C.c_uint64(0), C.c_uint64(0)
        _lib.check(_lib.load().mi_sa_tempering_state(self._h, _ptr(rung, C.c_int32), C.byref(prop), C.byref(acc)))
        return rung.astype(np.int64), int(prop.value), int(acc.value)

    def sync(self):
        _lib.check(_lib.load().mi_sa_sync(self._h))

    def kernel_ms(self) -> float:
        ms = C.c_float(0.0)
        _lib.check(_lib.load().mi_sa_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def launch_count(self) -> int:
        """Kernel launches that served the last anneal (kernel_ms() / launch_count() = mean launch time)."""
        k = C.c_int(0)
        _lib.check(_lib.load().mi_sa_last_launch_count(self._h, C.byref(k)))
        return int(k.value)

    def kernel_name(self) -> str:
        """The kernel(s) that served the last anneal, as a profiler names them."""
        buf = C.create_string_buffer(256)
        _lib.check(_lib.load().mi_sa_last_kernel_name(self._h, buf, 256))
        return buf.value.decode()

    def fetch(self, states: bool = True, energies: bool = True):
        if self._last is None:
            raise RuntimeError("fetch() before anneal()")
        R = self._last[0]
        st = np.empty((R, self.n_dev), dtype=self.state_dtype) if states else None
        en = np.empty(R, dtype=np.float64) if energies else None
        stats = np.zeros(3, dtype=np.uint64)
        _lib.check(_lib.load().mi_sa_fetch(
            self._h, st.ctypes.data_as(C.c_void_p) if st is not None else None,
            _ptr(en, C.c_double), _ptr(stats, C.c_uint64)))
        info = {"proposals": int(R) * int(self._last[1]) * int(self.n),
                "accepted": int(stats[1]), "row_bytes": int(stats[2])}
        if st is not None and self._inv is not None:
            # caller's variable i sits in device column inv[i]: a column gather (np.take is the fast form)
            st = np.take(st, self._inv, axis=1)
        return st, en, info

    def best(self, want_state: bool = True):
        idx = C.c_int(0)
        en = C.c_double(0.0)
        key = C.c_uint64(0)
        st = np.empty(self.n_dev, dtype=self.state_dtype) if want_state else None
        _lib.check(_lib.load().mi_sa_best(
            self._h, C.byref(idx), C.byref(en), C.byref(key),
            st.ctypes.data_as(C.c_void_p) if st is not None else None))
        if st is not None and self._inv is not None:
            st = st[self._inv]
        return int(idx.value), float(en.value), int(key.value), st


def energy_dense(Qs: np.ndarray, X: np.ndarray, offset: float = 0.0, device: int = 0, path: int = 0,
                 return_ms: bool = False):
    """Batched ``E_r = x_r^T Qs x_r + offset`` on the GPU (kernel K4).  ``path``: 0 auto (MFMA for
    batches of >= 32 states), 1 exact-fp64 VALU, 2 f32-input MFMA."""
    Qs = np.ascontiguousarray(Qs, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.uint8)
    if X.ndim != 2 or X.shape[1] != Qs.shape[0]:
        raise ValueError("X must have shape (R, n)")
    out = np.empty(X.shape[0], dtype=np.float64)
    ms = C.c_float(0.0)
    _lib.check(_lib.load().mi_energy_dense_f32_ex(_ptr(Qs, C.c_float), Qs.shape[0], _ptr(X, C.c_uint8),
                                                  X.shape[0], float(offset), _ptr(out, C.c_double),
                                                  int(device), int(path), C.byref(ms)))
    return (out, float(ms.value)) if return_ms else out


def energy_dense_f64(Qs: np.ndarray, X: np.ndarray, offset: float = 0.0, device: int = 0) -> np.ndarray:
    """``E_r = x_r^T Qs x_r + offset`` over an fp64 matrix on the GPU (every entry added once into fp64): the
    caller-model energies of a dense problem's samples."""
    Qs = np.ascontiguousarray(Qs, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.uint8)
    if X.ndim != 2 or Qs.ndim != 2 or Qs.shape[0] != Qs.shape[1] or X.shape[1] != Qs.shape[0]:
        raise ValueError("Qs must be n x n and X must have shape (R, n)")
    out = np.empty(X.shape[0], dtype=np.float64)
    _lib.check(_lib.load().mi_energy_dense_f64(_ptr(Qs, C.c_double), Qs.shape[0], _ptr(X, C.c_uint8), X.shape[0],
                                               float(offset), _ptr(out, C.c_double), int(device)))
    return out
