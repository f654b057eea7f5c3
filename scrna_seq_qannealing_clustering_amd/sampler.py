"""``MI355XSampler`` -- the drop-in for the sampler objects the reference constructs in-line.

The reference solves its models by calling a dimod ``Sampler``:

    LeapHybridSampler().sample_qubo(Q, label=...)                       BQM_clustering.py:56-57
    FixedEmbeddingComposite(DWaveSampler(...), emb).sample_qubo(Q, label=, chain_strength=,
        num_reads=, return_embedding=True)                              BQM_clustering.py:67-75
    EmbeddingComposite(DWaveSampler()).sample_qubo(Q, ...)              BQM_clustering.py:84-85
    hybrid.KerberosSampler().sample(bqm, max_iter=, num_reads=, ...)    BQM_clustering.py:386
    LeapHybridDQMSampler().sample_dqm(dqm, label=...)                   DQM_clustering.py:45

``MI355XSampler`` offers the same four methods (plus ``sample_ising``) and returns a SampleSet
(sampleset.py).  QPU-only keyword arguments (``label``, ``chain_strength``, ``return_embedding``,
``time_limit``, Kerberos' ``max_iter/qpu_reads/tabu_timeout/qpu_params`` ...) are accepted and
ignored; the annealing arguments follow ``neal.SimulatedAnnealingSampler.sample`` (``num_reads``,
``num_sweeps``, ``beta_range``, ``beta_schedule_type``, ``beta_schedule``,
``num_sweeps_per_beta``, ``seed``, ``initial_states``, ``initial_states_generator``).

Everything numeric runs in the HIP library (libmi_sa.so); there is no CPU path.
"""
from __future__ import annotations

import os
import time
from typing import Any, Dict, Hashable, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .bqm import BinaryQuadraticModel, DiscreteQuadraticModel
from .engine import Problem, energy_dense_f64, layout_block_for
from .models import (PottsModel, QuboModel, _csr_from_edges, default_beta_range,
                     make_beta_schedule, qubo_dict_to_model)
from .sampleset import SampleSet

# keyword arguments of the samplers the reference uses that have no meaning for an annealer on a GPU
_IGNORED_KWARGS = frozenset((
    "label", "chain_strength", "return_embedding", "time_limit", "annealing_time", "answer_mode",
    "auto_scale", "max_iter", "qpu_reads", "tabu_timeout", "qpu_params", "qpu_sampler",
    "convergence", "energy_threshold", "max_subproblem_size", "sa_reads", "sa_sweeps",
    "embedding_parameters", "chain_break_method", "chain_break_fraction", "warnings",
    "interrupt_function", "programming_thermalization", "readout_thermalization",
    "reduce_intersample_correlation", "num_spin_reversal_transforms", "flux_drift_compensation",
))

_DEFAULT_NUM_READS = 256


def _default_device() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


class PendingSampleSet:
    """A sampler call whose anneal is enqueued on the GPU (``MI355XSampler.sample_qubo_async``); ``result()`` waits for it."""

    def __init__(self, steps):
        self._steps = steps
        self._value = None
        self._done = False
        self._advance()                                   # runs up to the enqueued anneal (or to the end: an empty model)

    def _advance(self):
        try:
            next(self._steps)
        except StopIteration as stop:
            self._value, self._done = stop.value, True

    def result(self) -> SampleSet:
        while not self._done:
            self._advance()
        return self._value

    def __del__(self):                                    # an abandoned call still releases its device buffers
        try:
            if not self._done:
                self._steps.close()
        except Exception:
            pass


class MI355XSampler:
    """Replica-parallel simulated annealing on one MI355X (one wavefront per replica)."""

    parameters = {
        "num_reads": [], "num_sweeps": [], "beta_range": [], "beta_schedule_type": [],
        "beta_schedule": [], "num_sweeps_per_beta": [], "seed": [], "initial_states": [],
        "initial_states_generator": [], "resync_interval": [], "kernel": [], "min_cluster_size": [],
        **{k: [] for k in _IGNORED_KWARGS},
    }
    properties = {"category": "software", "beta_schedule_options": ("linear", "geometric", "custom"),
                  "engine": "libmi_sa (HIP, gfx950)"}

    def __init__(self, device: Optional[int] = None, replica_offset: int = 0):
        self.device = _default_device() if device is None else int(device)
        self.replica_offset = int(replica_offset)
        _lib.load()                              # fail now, loudly, if the engine is not built

    # ------------------------------------------------------------------------------------------
    # public dimod-style surface
    # ------------------------------------------------------------------------------------------
    def sample_qubo(self, Q, **kwargs) -> SampleSet:
        """``Q``: dict ``{(u, v): bias}`` (BQM_clustering.py:36-47) or a ``models.QuboModel``."""
        offset = kwargs.pop("offset", 0.0)
        model = Q if isinstance(Q, QuboModel) else qubo_dict_to_model(Q, offset=offset)
        return self._sample_binary(model, "BINARY", kwargs)

    def sample_qubo_async(self, Q, **kwargs) -> "PendingSampleSet":
        """``sample_qubo`` in two halves: the model is uploaded and the anneal ENQUEUED on the problem's own stream when this
        returns; ``.result()`` waits for it and builds the SampleSet.  Independent calls overlap on the GPU -- what the
        two halves of a bisection are (clustering.py prefetches the second half while it descends into the first): with 500
        reads a call is one wavefront's latency on half of the chip's SIMDs."""
        offset = kwargs.pop("offset", 0.0)
        model = Q if isinstance(Q, QuboModel) else qubo_dict_to_model(Q, offset=offset)
        return PendingSampleSet(self._binary_steps(model, "BINARY", kwargs))

    def sample_ising(self, h, J, **kwargs) -> SampleSet:
        bqm = BinaryQuadraticModel.from_ising(h, J, kwargs.pop("offset", 0.0))
        return self.sample(bqm, **kwargs)

    def sample(self, bqm, **kwargs) -> SampleSet:
        """``bqm``: ``bqm.BinaryQuadraticModel``, a real ``dimod.BinaryQuadraticModel`` (duck-typed
        on ``linear`` / ``quadratic`` / ``offset`` / ``vartype``) or a ``models.QuboModel``."""
        if isinstance(bqm, QuboModel):
            return self._sample_binary(bqm, "BINARY", kwargs)
        vartype = getattr(bqm.vartype, "name", str(bqm.vartype))
        linear = dict(bqm.linear)
        quadratic = dict(bqm.quadratic)
        offset = float(bqm.offset)
        if vartype == "SPIN":
            # s = 2x - 1:  h s -> 2h x - h ;  J s_u s_v -> 4J x_u x_v - 2J x_u - 2J x_v + J
            Q: Dict[Tuple[Hashable, Hashable], float] = {}
            lin = {v: 2.0 * b for v, b in linear.items()}
            off = offset - sum(linear.values())
            for (u, v), b in quadratic.items():
                Q[(u, v)] = 4.0 * b
                lin[u] = lin.get(u, 0.0) - 2.0 * b
                lin[v] = lin.get(v, 0.0) - 2.0 * b
                off += b
            Qd = {(v, v): b for v, b in lin.items()}
            Qd.update(Q)
            model = qubo_dict_to_model(Qd, offset=off)
        elif vartype == "BINARY":
            Qd = {(v, v): b for v, b in linear.items()}
            Qd.update(quadratic)
            model = qubo_dict_to_model(Qd, offset=offset)
        else:
            raise ValueError("unsupported vartype %r" % (vartype,))
        return self._sample_binary(model, vartype, kwargs)

    def sample_dqm(self, dqm, **kwargs) -> SampleSet:
        """``dqm``: ``models.PottsModel``, ``bqm.DiscreteQuadraticModel`` or a real
        ``dimod.DiscreteQuadraticModel`` whose biases have the Potts shape `clustering_dqm` builds
        (DQM_clustering.py:29-43): case-independent linear biases, equal-case quadratic biases."""
        model = dqm if isinstance(dqm, PottsModel) else dqm_to_potts(dqm)
        return self._sample_potts(model, kwargs)

    # ------------------------------------------------------------------------------------------
    # internals
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _split_kwargs(kwargs: Dict[str, Any]):
        ignored = sorted(k for k in kwargs if k in _IGNORED_KWARGS)
        unknown = sorted(k for k in kwargs if k not in MI355XSampler.parameters)
        if unknown:
            raise TypeError("MI355XSampler got unexpected keyword argument(s): %s" % ", ".join(unknown))
        return {k: v for k, v in kwargs.items() if k not in _IGNORED_KWARGS}, ignored

    @staticmethod
    def _seed(seed) -> int:
        if seed is None:
            return int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
        if not isinstance(seed, (int, np.integer)) or seed < 0:
            raise TypeError("'seed' should be a non-negative integer")
        return int(seed)

    def _schedule(self, kw, range_fn):
        num_sweeps = int(kw.get("num_sweeps", 1000))
        if num_sweeps < 0:
            raise ValueError("'num_sweeps' should be a non-negative integer")
        stype = kw.get("beta_schedule_type", "geometric")
        beta_range = kw.get("beta_range")
        if stype != "custom" and beta_range is None:
            beta_range = range_fn()
        betas = make_beta_schedule(num_sweeps, beta_range if beta_range is not None else (1.0, 1.0),
                                   stype, int(kw.get("num_sweeps_per_beta", 1)),
                                   kw.get("beta_schedule"))
        if len(betas) and (np.any(betas <= 0) or not np.all(np.isfinite(betas))):
            raise ValueError("beta schedule must be positive and finite on this engine")
        return betas, (None if beta_range is None else [float(beta_range[0]), float(beta_range[1])]), stype

    @staticmethod
    def _initial_states(kw, variables, num_reads, vartype):
        """neal semantics: ``initial_states`` = (array, labels) / array / SampleSet; generator
        'none' | 'tile' | 'random'.  Returns (num_reads, array-or-None in BINARY/label form)."""
        init = kw.get("initial_states")
        gen = kw.get("initial_states_generator", "random")
        if gen not in ("none", "tile", "random"):
            raise ValueError("unknown value for 'initial_states_generator'")
        if init is None:
            if gen == "none":
                raise ValueError("initial_states_generator='none' requires initial_states")
            return (num_reads if num_reads is not None else _DEFAULT_NUM_READS), None
        labels = None
        if isinstance(init, SampleSet):
            arr, labels = init.record["sample"], init.variables
            if init.vartype == "SPIN" and vartype != "DISCRETE":
                arr = (arr + 1) // 2
        elif isinstance(init, tuple) and len(init) == 2:
            arr, labels = np.asarray(init[0]), list(init[1])
        elif isinstance(init, Mapping):
            labels = list(init.keys())
            arr = np.asarray([[init[v] for v in labels]])
        else:
            arr = np.asarray(init)
        if arr.ndim == 1:
            arr = arr[None, :]
        if labels is not None:
            pos = {v: i for i, v in enumerate(labels)}
            try:
                arr = arr[:, [pos[v] for v in variables]]
            except KeyError as exc:
                raise ValueError("initial_states is missing variable %r" % (exc.args[0],)) from exc
        if arr.shape[1] != len(variables):
            raise ValueError("initial_states has %d columns, model has %d variables"
                             % (arr.shape[1], len(variables)))
        if vartype == "SPIN" and not isinstance(init, SampleSet):
            if np.any((arr != 1) & (arr != -1)):
                raise ValueError("SPIN initial_states must be +-1")
            arr = (arr + 1) // 2
        if num_reads is None:
            num_reads = arr.shape[0]
        have = arr.shape[0]
        if have > num_reads:
            arr = arr[:num_reads]
        elif have < num_reads:
            if gen == "none":
                raise ValueError("insufficient initial states for num_reads with generator 'none'")
            if gen == "tile":
                reps = -(-num_reads // have)
                arr = np.tile(arr, (reps, 1))[:num_reads]
            else:                                 # 'random': remaining replicas start randomly --
                return num_reads, ("partial", arr)  # handled by the caller (device RNG rows)
        return num_reads, arr

    def _sample_binary(self, model: QuboModel, vartype: str, kwargs) -> SampleSet:
        return PendingSampleSet(self._binary_steps(model, vartype, kwargs)).result()

    def _binary_steps(self, model: QuboModel, vartype: str, kwargs):
        """Generator: everything up to the enqueued anneal, ONE ``yield``, then the fetch and the SampleSet (its return value)."""
        kw, ignored = self._split_kwargs(dict(kwargs))
        t0 = time.perf_counter()
        n = model.num_variables
        if n == 0:
            return SampleSet(np.zeros((0, 0), dtype=np.int8), np.zeros(0), [], vartype,
                             info={"ignored_kwargs": ignored})
        seed = self._seed(kw.get("seed"))
        num_reads, init = self._initial_states(kw, model.variables, kw.get("num_reads"), vartype)
        if num_reads < 1:
            raise ValueError("'num_reads' should be a positive integer")
        betas, beta_range, stype = self._schedule(kw, lambda: default_beta_range(model))
        kernel = kw.get("kernel", "auto")
        if kernel not in ("auto", "dense", "csr"):
            raise ValueError("kernel must be 'auto', 'dense' or 'csr'")
        # structured models (sparse couplings + one uniform pair term: every graph-partition QUBO of the
        # reference) run on the CSR kernel at any size -- it is the faster one and needs no n x n matrix;
        # (rows wider than 64 neighbours on its runtime-width form); general QUBOs run on the dense kernels (n <= 65536)
        max_deg = int(np.diff(model.rowptr).max()) if n else 0
        use_csr = (kernel == "csr") or (kernel == "auto" and model._dense is None and (max_deg <= 4096 or n > 65536))
        if use_csr and model._dense is not None:
            raise ValueError("kernel='csr' needs a structured (CSR + uniform pair) model")
        if use_csr:
            prob = Problem.csr_rank1(model.rowptr, model.col, model.val.astype(np.float32),
                                     model.lin.astype(np.float32), float(np.float32(model.c_pair)),
                                     offset=model.offset, device=self.device, order="padded",
                                     energy_model=(model.val, model.lin, model.c_pair),
                                     block=layout_block_for(n, num_reads, max_deg), weights=model.weights)
        else:
            dense64 = model.dense_Qs()
            prob = Problem.dense(_symmetric_f32(dense64), offset=model.offset, device=self.device)
        with prob:
            init_arr = init
            if isinstance(init, tuple):           # partial initial states + random remainder
                have = init[1]
                prob.anneal(num_reads, betas[:0], seed, self.replica_offset)   # RNG init states only
                rnd, _, _ = prob.fetch(energies=False)
                rnd[: have.shape[0]] = have
                init_arr = rnd
            t1 = time.perf_counter()
            prob.anneal(num_reads, betas, seed, self.replica_offset, init_arr,
                        int(kw.get("resync_interval", 0)))
            yield None                                # (enqueued: what follows waits for the run)
            states, dev_energy, stats = prob.fetch()
            kernel_ms = prob.kernel_ms()
            t2 = time.perf_counter()
        # energies in the caller's fp64 coefficients (what dimod's SampleSet.from_samples_bqm evaluates on the
        # host): the structured kernel evaluates them itself (energy_model above); the dense kernels report
        # the energies of the fp32 matrix they anneal, so that path runs the fp64 energy kernel on the samples
        energies = dev_energy if use_csr else energy_dense_f64(dense64, states, model.offset, self.device)
        samples = states.astype(np.int8)
        if vartype == "SPIN":
            samples = 2 * samples - 1
        info = {
            "beta_range": beta_range, "beta_schedule_type": stype, "seed": seed,
            "num_sweeps": int(len(betas)), "num_reads": int(num_reads),
            "kernel": "csr_rank1" if use_csr else "dense", "device": self.device,
            "timing": {"upload_s": t1 - t0, "anneal_s": t2 - t1, "kernel_ms": kernel_ms},
            "updates_per_s": (stats["proposals"] / (kernel_ms * 1e-3)) if kernel_ms > 0 else None,
            "accepted": stats["accepted"], "proposals": stats["proposals"],
            "energy_evaluation": "device fp64 (caller's coefficients)",
            "device_energy_max_abs_diff": float(np.max(np.abs(dev_energy - energies))),
            "ignored_kwargs": ignored,
        }
        if kwargs.get("return_embedding"):
            info["embedding_context"] = {"embedding": {}}    # read at BQM_clustering.py:79
        return SampleSet(samples, energies, model.variables, vartype, info=info)

    def _sample_potts(self, model: PottsModel, kwargs) -> SampleSet:
        kw, ignored = self._split_kwargs(dict(kwargs))
        t0 = time.perf_counter()
        n = model.num_variables
        if n == 0:
            return SampleSet(np.zeros((0, 0), dtype=np.int32), np.zeros(0), [], "DISCRETE",
                             info={"ignored_kwargs": ignored})
        seed = self._seed(kw.get("seed"))
        num_reads, init = self._initial_states(kw, model.variables, kw.get("num_reads"), "DISCRETE")
        betas, beta_range, stype = self._schedule(kw, lambda: default_potts_beta_range(model))
        prob = Problem.potts_csr(model.rowptr, model.col, model.val.astype(np.float32),
                                 float(np.float32(model.c_pair)), n, model.num_cases,
                                 lin_offset=model.lin_offset, device=self.device, order="padded",
                                 energy_model=(model.val, model.c_pair))
        # the CQM's "every cluster has at least m members" (CQM_clustering.py:46-48): a hard constraint on moves
        min_size = int(kw.get("min_cluster_size", model.info.get("min_cluster_size", 0)) or 0)
        if min_size * model.num_cases > n:
            raise ValueError("min_cluster_size %d x %d clusters exceeds the %d variables" % (min_size, model.num_cases, n))
        with prob:
            init_arr = init
            if isinstance(init, tuple) or (min_size > 0 and init is None):
                have = init[1] if isinstance(init, tuple) else np.zeros((0, n), dtype=np.uint16)
                prob.anneal(num_reads, betas[:0], seed, self.replica_offset)
                rnd, _, _ = prob.fetch(energies=False)
                rnd[: have.shape[0]] = have
                init_arr = rnd
            if min_size > 0:
                init_arr = _make_feasible(np.array(init_arr, dtype=np.uint16, copy=True), model.num_cases, min_size)
                prob.set_option("min_cluster_size", min_size)
            t1 = time.perf_counter()
            prob.anneal(num_reads, betas, seed, self.replica_offset, init_arr)
            labels, dev_energy, stats = prob.fetch()
            kernel_ms = prob.kernel_ms()
            t2 = time.perf_counter()
        energies = dev_energy                    # evaluated on the device in the model's fp64 coefficients
        info = {
            "beta_range": beta_range, "beta_schedule_type": stype, "seed": seed,
            "num_sweeps": int(len(betas)), "num_reads": int(num_reads), "kernel": "potts_csr",
            "device": self.device,
            "timing": {"upload_s": t1 - t0, "anneal_s": t2 - t1, "kernel_ms": kernel_ms},
            "updates_per_s": (stats["proposals"] / (kernel_ms * 1e-3)) if kernel_ms > 0 else None,
            "accepted": stats["accepted"], "proposals": stats["proposals"],
            "energy_evaluation": "device fp64 (caller's coefficients)",
            "ignored_kwargs": ignored,
        }
        return SampleSet(labels.astype(np.int32), energies, model.variables, "DISCRETE", info=info)


def _make_feasible(labels: np.ndarray, K: int, min_size: int) -> np.ndarray:
    """Deterministic repair of initial labelings that leave a cluster below ``min_size``: members of the
    largest clusters (highest indices first) are moved into the deficient ones."""
    for r in range(labels.shape[0]):
        cnt = np.bincount(labels[r], minlength=K)
        for c in np.where(cnt < min_size)[0]:
            while cnt[c] < min_size:
                big = int(np.argmax(cnt))
                i = int(np.where(labels[r] == big)[0][-1])
                labels[r, i] = c
                cnt[big] -= 1
                cnt[c] += 1
    return labels


def _symmetric_f32(Qs64: np.ndarray) -> np.ndarray:
    """fp64 symmetric -> fp32 symmetric (rounding each entry once keeps symmetry)."""
    Q = np.asarray(Qs64, dtype=np.float64).astype(np.float32)
    return np.ascontiguousarray(Q)


def default_potts_beta_range(model: PottsModel) -> Tuple[float, float]:
    """Hot: the largest possible single-move |dE| gets acceptance 1/2; cold: the smallest non-zero
    coupling gets acceptance 1/100 (the neal rule applied to the move set of the Potts chain)."""
    n = model.num_variables
    rows = np.repeat(np.arange(n), np.diff(model.rowptr))
    full = model.val + model.c_pair
    abs_sum = np.zeros(n)
    np.add.at(abs_sum, rows, np.abs(full) - abs(model.c_pair))
    abs_sum += abs(model.c_pair) * (n - 1)
    max_field = float(abs_sum.max()) if n else 1.0
    cands = np.abs(full[full != 0.0])
    if model.c_pair != 0.0:
        cands = np.concatenate([cands, [abs(model.c_pair)]])
    min_bias = float(cands.min()) if len(cands) else 1.0
    if max_field <= 0:
        max_field = 1.0
    return float(np.log(2.0) / max_field), float(np.log(100.0) / min_bias)


def dqm_to_potts(dqm) -> PottsModel:
    """Reduce a (look-alike or real dimod) DiscreteQuadraticModel to Potts form, verifying that it
    has the shape `clustering_dqm` builds.  Raises NotImplementedError otherwise."""
    variables = list(dqm.variables)
    n = len(variables)
    if n == 0:
        return PottsModel([], 0, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0), 0.0,
                          np.zeros(0))
    K = int(dqm.num_cases(variables[0]))
    index = {v: i for i, v in enumerate(variables)}
    lin = np.zeros(n)
    for v in variables:
        if int(dqm.num_cases(v)) != K:
            raise NotImplementedError("all variables must have the same number of cases")
        b = np.asarray(dqm.get_linear(v), dtype=np.float64)
        if np.any(b != b[0]):
            raise NotImplementedError("case-dependent linear biases are not in Potts form")
        lin[index[v]] = b[0]
    if isinstance(dqm, DiscreteQuadraticModel):
        pairs = [(u, v, tab) for (u, v), tab in dqm.interactions()]
    else:                                         # real dimod DQM
        pairs = []
        for u in variables:
            for v in dqm.adj[u]:
                if index[u] < index[v]:
                    pairs.append((u, v, dqm.get_quadratic(u, v)))
    eu = np.empty(len(pairs), dtype=np.int32)
    ev = np.empty(len(pairs), dtype=np.int32)
    b = np.empty(len(pairs), dtype=np.float64)
    for k, (u, v, tab) in enumerate(pairs):
        vals = set()
        for (cu, cv), x in tab.items():
            if cu != cv:
                if x != 0.0:
                    raise NotImplementedError("unequal-case quadratic biases are not in Potts form")
            else:
                vals.add(x)
        if len(vals) > 1 or (vals and len([1 for (cu, cv) in tab if cu == cv]) != K):
            raise NotImplementedError("case-dependent quadratic biases are not in Potts form")
        eu[k], ev[k] = index[u], index[v]
        b[k] = vals.pop() if vals else 0.0
    c_pair = 0.0
    npairs = n * (n - 1) // 2
    if len(b) and 2 * len(b) >= npairs:
        vals_u, counts = np.unique(b, return_counts=True)
        top = int(np.argmax(counts))
        if 2 * counts[top] >= npairs:
            c_pair = float(vals_u[top])
    if c_pair != 0.0:
        dense = np.zeros((n, n))
        lo, hi = np.minimum(eu, ev), np.maximum(eu, ev)
        dense[lo, hi] = b
        iu, ju = np.triu_indices(n, 1)
        resid = dense[iu, ju] - c_pair
        nz = resid != 0.0
        rowptr, col, val = _csr_from_edges(n, iu[nz].astype(np.int32), ju[nz].astype(np.int32),
                                           resid[nz])
    else:
        rowptr, col, val = _csr_from_edges(n, eu, ev, b)
    return PottsModel(variables, K, rowptr, col, val, c_pair, lin, info={"kind": "dqm"})
