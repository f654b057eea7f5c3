"""Minimal stand-ins for the two dimod model classes the reference constructs itself, so that code
written against the reference runs unchanged when ``dimod`` is not installed:

* ``BinaryQuadraticModel.from_qubo`` + ``add_linear_inequality_constraint``
  (BQM_clustering.py:371-380, `clustering_bqm_3`)
* ``DiscreteQuadraticModel`` with ``add_variable`` / ``set_linear`` / ``set_quadratic``
  (DQM_clustering.py:29-43, `clustering_dqm`)

``MI355XSampler`` accepts these, the real dimod objects (duck-typed), and the array-form models of
``models.py``.
"""
from __future__ import annotations

from typing import Dict, Hashable, Iterable, List, Sequence, Tuple

import numpy as np


class BinaryQuadraticModel:
    def __init__(self, linear=None, quadratic=None, offset: float = 0.0, vartype: str = "BINARY"):
        self.linear: Dict[Hashable, float] = {}
        self.quadratic: Dict[Tuple[Hashable, Hashable], float] = {}
        self.offset = float(offset)
        self.vartype = getattr(vartype, "name", vartype)
        if self.vartype not in ("BINARY", "SPIN"):
            raise ValueError("vartype must be 'BINARY' or 'SPIN'")
        for v, b in (linear or {}).items():
            self.add_linear(v, b)
        for (u, v), b in (quadratic or {}).items():
            self.add_quadratic(u, v, b)

    # construction ---------------------------------------------------------------------------
    @classmethod
    def from_qubo(cls, Q, offset: float = 0.0) -> "BinaryQuadraticModel":
        bqm = cls(offset=offset, vartype="BINARY")
        for (u, v), b in Q.items():
            if u == v:
                bqm.add_linear(u, b)
            else:
                bqm.add_quadratic(u, v, b)
        return bqm

    @classmethod
    def from_ising(cls, h, J, offset: float = 0.0) -> "BinaryQuadraticModel":
        if not isinstance(h, dict):
            h = dict(enumerate(h))
        return cls(h, J, offset, "SPIN")

    def add_variable(self, v, bias: float = 0.0):
        self.linear[v] = self.linear.get(v, 0.0) + bias
        return v

    add_linear = add_variable

    def add_quadratic(self, u, v, bias: float):
        if u == v:
            raise ValueError("no self-loops allowed: {!r}".format(u))
        self.linear.setdefault(u, 0.0)
        self.linear.setdefault(v, 0.0)
        key = (v, u) if (v, u) in self.quadratic else (u, v)
        self.quadratic[key] = self.quadratic.get(key, 0.0) + bias

    @property
    def variables(self) -> List[Hashable]:
        return list(self.linear.keys())

    @property
    def num_variables(self) -> int:
        return len(self.linear)

    def energy(self, sample) -> float:
        e = self.offset
        for v, b in self.linear.items():
            e += b * sample[v]
        for (u, v), b in self.quadratic.items():
            e += b * sample[u] * sample[v]
        return e

    # the one constraint helper the reference uses ----------------------------------------------
    def add_linear_inequality_constraint(self, terms: Iterable[Tuple[Hashable, int]],
                                         lagrange_multiplier: float, label: str,
                                         constant: int = 0, lb: float = 0, ub: float = 0,
                                         cross_zero: bool = False, penalization_method="slack"):
        """``lb <= sum_i a_i x_i + constant <= ub`` the way dimod adds it (BQM_clustering.py:376-380 calls it with
        ``lb = size_limit``, ``ub = n / 6`` -- fractional -- and ``lagrange_multiplier = gamma``): binary slack variables
        ``slack_<label>_<j>`` with POSITIVE coefficients spanning ``[0, int(ub_c - lb_c)]`` and the equality penalty
        ``lagrange * (sum_i a_i x_i + sum_j c_j s_j - ub_c)^2`` with ``ub_c = min(sum of positive a_i, ub - constant)``
        kept as it is (not floored).  See :func:`inequality_slack`.  Returns the slack terms."""
        if self.vartype != "BINARY":
            raise ValueError("inequality constraints are supported for BINARY models")
        terms = list(terms)
        coeffs, ub_c = inequality_slack([a for _, a in terms], lb, ub, constant, label)
        if coeffs is None:
            return []                                            # feasible for every state: dimod adds nothing
        slack = [("slack_%s_%d" % (label, j), c) for j, c in enumerate(coeffs)]
        allterms = terms + slack
        const = -ub_c
        lam = float(lagrange_multiplier)
        # lam * (sum a z + const)^2, z binary  => z^2 = z
        for k, (v, a) in enumerate(allterms):
            self.add_linear(v, lam * (a * a + 2.0 * a * const))
            for (u, a2) in allterms[k + 1:]:
                self.add_quadratic(v, u, lam * 2.0 * a * a2)
        self.offset += lam * const * const
        return slack


def inequality_slack(coefficients, lb, ub, constant=0, label="constraint"):
    """Slack coefficients and right-hand side of dimod's ``add_linear_inequality_constraint`` [upstream dimod >= 0.10,
    restated from its published source, not importable here: parity unpinned]:
    ``ub_c = min(sum of positive coefficients, ub - constant)``, ``lb_c = max(sum of negative coefficients, lb - constant)``;
    nothing to add when every state is feasible (returns ``(None, ub_c)``); ``ValueError`` when ``ub_c < lb_c``;
    ``slack_upper_bound = int(ub_c - lb_c)``; no slack when that is 0 (a plain equality at ``ub_c``); else coefficients
    ``2^j`` for ``j < floor(log2(slack_upper_bound))`` plus the remainder ``slack_upper_bound - 2^floor(log2) + 1``.
    The penalty is ``lagrange * (terms + slack - ub_c)^2`` -- with a fractional ``ub`` (the reference passes ``n / 6``) its
    minimum over the slack is not zero."""
    pos = float(sum(a for a in coefficients if a > 0))
    neg = float(sum(a for a in coefficients if a < 0))
    ub_c = min(pos, float(ub) - constant)
    lb_c = max(neg, float(lb) - constant)
    if pos <= ub_c and neg >= lb_c:
        return None, ub_c
    if ub_c < lb_c:
        raise ValueError("The given constraint (%s) is infeasible with any value for state variables." % label)
    slack_upper_bound = int(ub_c - lb_c)
    if slack_upper_bound == 0:
        return [], ub_c
    num_slack = int(np.floor(np.log2(slack_upper_bound)))
    coeffs = [2 ** j for j in range(num_slack)]
    if slack_upper_bound - 2 ** num_slack >= 0:
        coeffs.append(slack_upper_bound - 2 ** num_slack + 1)
    return coeffs, ub_c


class DiscreteQuadraticModel:
    """One discrete variable per label with ``num_cases`` cases; linear bias per case; quadratic
    biases per (case_u, case_v) for variable pairs.  ``set_`` overwrites, as in dimod."""

    def __init__(self):
        self._cases: Dict[Hashable, int] = {}
        self._linear: Dict[Hashable, np.ndarray] = {}
        self._quadratic: Dict[Tuple[Hashable, Hashable], Dict[Tuple[int, int], float]] = {}
        self.offset = 0.0

    def add_variable(self, num_cases: int, label=None):
        if label is None:
            label = len(self._cases)
        if label in self._cases:
            raise ValueError("variable {!r} already exists".format(label))
        if num_cases < 1:
            raise ValueError("discrete variables must have at least one case")
        self._cases[label] = int(num_cases)
        self._linear[label] = np.zeros(int(num_cases), dtype=np.float64)
        return label

    @property
    def variables(self) -> List[Hashable]:
        return list(self._cases.keys())

    def num_variables(self) -> int:
        return len(self._cases)

    def num_cases(self, v=None) -> int:
        if v is None:
            return int(sum(self._cases.values()))
        return self._cases[v]

    def set_linear(self, v, biases: Sequence[float]):
        b = np.asarray(biases, dtype=np.float64)
        if b.shape != (self._cases[v],):
            raise ValueError("wrong number of biases for variable {!r}".format(v))
        self._linear[v] = b.copy()

    def get_linear(self, v) -> np.ndarray:
        return self._linear[v].copy()

    def set_linear_case(self, v, case: int, bias: float):
        self._linear[v][case] = bias

    def set_quadratic(self, u, v, biases):
        if u == v:
            raise ValueError("there cannot be a quadratic interaction between a variable and itself")
        if u not in self._cases or v not in self._cases:
            raise ValueError("unknown variable")
        key, flip = ((v, u), True) if (v, u) in self._quadratic else ((u, v), False)
        if isinstance(biases, dict):
            tab = {((cv, cu) if flip else (cu, cv)): float(b) for (cu, cv), b in biases.items()}
        else:
            arr = np.asarray(biases, dtype=np.float64)
            tab = {((j, i) if flip else (i, j)): float(arr[i, j])
                   for i in range(arr.shape[0]) for j in range(arr.shape[1]) if arr[i, j] != 0.0}
        self._quadratic[key] = tab

    def get_quadratic(self, u, v, array: bool = False):
        if (u, v) in self._quadratic:
            tab = dict(self._quadratic[(u, v)])
        elif (v, u) in self._quadratic:
            tab = {(b, a): x for (a, b), x in self._quadratic[(v, u)].items()}
        else:
            raise ValueError("no interaction between {!r} and {!r}".format(u, v))
        if not array:
            return tab
        out = np.zeros((self._cases[u], self._cases[v]))
        for (a, b), x in tab.items():
            out[a, b] = x
        return out

    def num_variable_interactions(self) -> int:
        return len(self._quadratic)

    def interactions(self):
        return self._quadratic.items()

    def energy(self, sample) -> float:
        e = self.offset
        for v, lab in sample.items():
            e += self._linear[v][lab]
        for (u, v), tab in self._quadratic.items():
            e += tab.get((sample[u], sample[v]), 0.0)
        return e
