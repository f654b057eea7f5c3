"""ctypes wrapper over oracle/liboracle.so (built by oracle/Makefile).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("sa_oracle.c", "snn_oracle.c")]
    if force or not os.path.exists(so) or any(
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so) for src in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_neglog_u.restype = C.c_float
        _LIB.orc_neglog_u.argtypes = [C.c_uint32]
        _LIB.orc_chain_word.restype = C.c_uint32
        _LIB.orc_chain_word.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def neglog_u(r):
    return float(lib().orc_neglog_u(C.c_uint32(int(r))))


def chain_word(seed, i, s, g, tag):
    return int(lib().orc_chain_word(seed, i, s, g, tag))


def sa_dense_philox(Qs, R, betas, seed, offset=0.0, replica_offset=0, init=None,
                    resync_interval=0, sweep_offset=0, num_sweeps=None):
    """betas: one per sweep, or (num_sweeps given) one per replica held for num_sweeps sweeps."""
    Qs = np.ascontiguousarray(Qs, dtype=np.float32)
    n = Qs.shape[0]
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    states = np.zeros((R, n), dtype=np.uint8)
    energy = np.zeros(R, dtype=np.float64)
    stats = np.zeros(2, dtype=np.uint64)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.uint8)
    rc = lib().orc_sa_dense_philox(
        _p(Qs, C.c_float), C.c_int(n), C.c_double(offset), C.c_int(R), C.c_uint32(replica_offset),
        C.c_int(len(betas) if num_sweeps is None else num_sweeps), _p(betas, C.c_double), C.c_uint64(seed),
        _p(init, C.c_uint8), C.c_int(resync_interval), _p(states, C.c_uint8), _p(energy, C.c_double),
        _p(stats, C.c_uint64), C.c_uint32(sweep_offset), C.c_int(0 if num_sweeps is None else 1))
    assert rc == 0
    return states, energy, stats


def sa_csr_rank1_philox(rowptr, col, val, lin, c_pair, R, betas, seed, offset=0.0,
                        replica_offset=0, init=None, resync_interval=0, sweep_offset=0, num_sweeps=None, weights=None):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    lin = np.ascontiguousarray(lin, dtype=np.float32)
    n = len(lin)
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    states = np.zeros((R, n), dtype=np.uint8)
    energy = np.zeros(R, dtype=np.float64)
    stats = np.zeros(2, dtype=np.uint64)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.uint8)
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.int32)
    rc = lib().orc_sa_csr_rank1_philox_w(
        _p(rowptr, C.c_int), _p(col, C.c_int), _p(val, C.c_float), _p(lin, C.c_float),
        C.c_float(c_pair), C.c_int(n), C.c_double(offset), C.c_int(R), C.c_uint32(replica_offset),
        C.c_int(len(betas) if num_sweeps is None else num_sweeps), _p(betas, C.c_double), C.c_uint64(seed),
        _p(init, C.c_uint8), C.c_int(resync_interval), _p(states, C.c_uint8), _p(energy, C.c_double),
        _p(stats, C.c_uint64), C.c_uint32(sweep_offset), C.c_int(0 if num_sweeps is None else 1), _p(weights, C.c_int))
    assert rc == 0
    return states, energy, stats


def potts_csr_philox(rowptr, col, val, c_pair, n, K, R, betas, seed, lin_offset=0.0,
                     replica_offset=0, init=None, sweep_offset=0, num_sweeps=None, min_size=0, absent=None):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    labels = np.zeros((R, n), dtype=np.uint16)
    energy = np.zeros(R, dtype=np.float64)
    stats = np.zeros(2, dtype=np.uint64)
    if init is not None:
        init = np.ascontiguousarray(init, dtype=np.uint16)
    if absent is not None:
        absent = np.ascontiguousarray(absent, dtype=np.uint8)
    rc = lib().orc_potts_csr_philox_absent(
        _p(rowptr, C.c_int), _p(col, C.c_int), _p(val, C.c_float), C.c_float(c_pair), C.c_int(n),
        C.c_int(K), C.c_double(lin_offset), C.c_int(R), C.c_uint32(replica_offset),
        C.c_int(len(betas) if num_sweeps is None else num_sweeps), _p(betas, C.c_double), C.c_uint64(seed),
        _p(init, C.c_uint16), _p(labels, C.c_uint16), _p(energy, C.c_double), _p(stats, C.c_uint64),
        C.c_uint32(sweep_offset), C.c_int(0 if num_sweeps is None else 1), C.c_int(int(min_size)),
        _p(absent, C.c_uint8))
    assert rc == 0
    return labels, energy, stats


def qubo_to_ising_dense(Qs):
    """Symmetric Qs (diag = linear) -> (h, J dense symmetric zero-diag, offset): E_qubo(x) =
    E_ising(2x-1) + offset.  With x=(s+1)/2: x^T Qs x = sum_i Qs_ii (s_i+1)/2 +
    sum_{i<j} 2 Qs_ij (s_i+1)(s_j+1)/4.  (neal: h_i=Q_ii/2+sum_j Q_ij/4, J_ij=Q_ij/4 with Q_ij the
    upper-triangular coefficient = 2 Qs_ij.)"""
    Qs = np.asarray(Qs, dtype=np.float64)
    d = np.diag(Qs).copy()
    off = Qs - np.diag(d)
    J = off / 2.0
    h = d / 2.0 + off.sum(axis=1) / 2.0
    offset = d.sum() / 2.0 + off.sum() / 4.0
    return h, J, offset


def sa_ising_neal_dense(h, J, num_reads, beta_schedule, seed, sweeps_per_beta=1, init_spins=None,
                        threads=1, rng=None):
    h = np.ascontiguousarray(h, dtype=np.float64)
    J = np.ascontiguousarray(J, dtype=np.float64)
    n = len(h)
    bs = np.ascontiguousarray(beta_schedule, dtype=np.float64)
    if init_spins is None:
        rng = rng or np.random.RandomState(seed & 0xFFFFFFFF)
        init_spins = (2 * rng.randint(0, 2, size=(num_reads, n)) - 1)
    states = np.ascontiguousarray(init_spins, dtype=np.int8).copy()
    energies = np.zeros(num_reads, dtype=np.float64)
    stats = np.zeros(2, dtype=np.uint64)
    done = lib().orc_sa_ising_neal_dense(
        _p(states, C.c_int8), _p(energies, C.c_double), C.c_int(num_reads), C.c_int(n),
        _p(h, C.c_double), _p(J, C.c_double), C.c_int(sweeps_per_beta), _p(bs, C.c_double),
        C.c_int(len(bs)), C.c_uint64(seed), C.c_int(threads), _p(stats, C.c_uint64))
    assert done == num_reads
    return states, energies, stats


def sa_ising_neal_csr(h, nbr_ptr, nbr, nbr_J, num_reads, beta_schedule, seed, sweeps_per_beta=1,
                      init_spins=None, rng=None):
    h = np.ascontiguousarray(h, dtype=np.float64)
    n = len(h)
    nbr_ptr = np.ascontiguousarray(nbr_ptr, dtype=np.int32)
    nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    nbr_J = np.ascontiguousarray(nbr_J, dtype=np.float64)
    bs = np.ascontiguousarray(beta_schedule, dtype=np.float64)
    if init_spins is None:
        rng = rng or np.random.RandomState(seed & 0xFFFFFFFF)
        init_spins = (2 * rng.randint(0, 2, size=(num_reads, n)) - 1)
    states = np.ascontiguousarray(init_spins, dtype=np.int8).copy()
    energies = np.zeros(num_reads, dtype=np.float64)
    stats = np.zeros(2, dtype=np.uint64)
    done = lib().orc_sa_ising_neal(
        _p(states, C.c_int8), _p(energies, C.c_double), C.c_int(num_reads), C.c_int(n),
        _p(h, C.c_double), _p(nbr_ptr, C.c_int), _p(nbr, C.c_int), _p(nbr_J, C.c_double),
        C.c_int(sweeps_per_beta), _p(bs, C.c_double), C.c_int(len(bs)), C.c_uint64(seed),
        _p(stats, C.c_uint64))
    assert done == num_reads
    return states, energies, stats


def energy_dense_f64(Qs, X, offset=0.0):
    X = np.ascontiguousarray(X, dtype=np.uint8)
    R, n = X.shape
    out = np.zeros(R, dtype=np.float64)
    if np.asarray(Qs).dtype == np.float64:
        Qd = np.ascontiguousarray(Qs, dtype=np.float64)
        lib().orc_energy_dense_f64d(_p(Qd, C.c_double), C.c_int(n), _p(X, C.c_uint8), C.c_int(R),
                                    C.c_double(offset), _p(out, C.c_double))
    else:
        Qf = np.ascontiguousarray(Qs, dtype=np.float32)
        lib().orc_energy_dense_f64(_p(Qf, C.c_float), C.c_int(n), _p(X, C.c_uint8), C.c_int(R),
                                   C.c_double(offset), _p(out, C.c_double))
    return out


def cut_edges(eu, ev, X):
    eu = np.ascontiguousarray(eu, dtype=np.int32)
    ev = np.ascontiguousarray(ev, dtype=np.int32)
    X = np.ascontiguousarray(X)
    R, n = X.shape
    out = np.zeros(R, dtype=np.int64)
    if X.dtype == np.uint16:
        lib().orc_cut_edges_u16(_p(eu, C.c_int), _p(ev, C.c_int), C.c_int(len(eu)),
                                _p(X, C.c_uint16), C.c_int(n), C.c_int(R), _p(out, C.c_int64))
    else:
        X = np.ascontiguousarray(X, dtype=np.uint8)
        lib().orc_cut_edges_u8(_p(eu, C.c_int), _p(ev, C.c_int), C.c_int(len(eu)),
                               _p(X, C.c_uint8), C.c_int(n), C.c_int(R), _p(out, C.c_int64))
    return out


def bruteforce_qubo(Qs, offset=0.0):
    Qd = np.ascontiguousarray(Qs, dtype=np.float64)
    n = Qd.shape[0]
    mn = C.c_double()
    am = C.c_uint64()
    nm = C.c_uint64()
    se = C.c_double()
    rc = lib().orc_bruteforce_qubo(_p(Qd, C.c_double), C.c_int(n), C.c_double(offset),
                                   C.byref(mn), C.byref(am), C.byref(nm), C.byref(se))
    assert rc == 0
    return mn.value, am.value, nm.value, se.value
