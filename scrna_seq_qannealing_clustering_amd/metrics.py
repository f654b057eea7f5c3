"""Cluster-quality metrics on the GPU: the numbers the reference publishes for its clusterings.

Reference (R, after the clustering): `/root/reference/R/pbmc3k/Pbmc3k_benchmark_clusters.Rmd`
  :36,47,69  within-cluster average Jaccard distance   ``mean(proxy::dist(cells, method = "jaccard"))``
  :82-94     silhouette widths                          ``cluster::silhouette(labels, dist)``
  :98-112    ``fpc::cluster.stats(dist, labels)``        -> ``R/pbmc3k/QA_benchmark.csv`` ...
One all-pairs pass on the device (csrc/metrics_kernels.hip, include/mi_metrics.h) returns per-cell distance
sums per cluster, squared-distance sums, cluster diameters and the separation matrix; every reported number
is a closed form of those, evaluated here in fp64.  No n x n matrix is built unless ``return_distances``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def pack_expression(X: np.ndarray) -> np.ndarray:
    """(n cells x g genes) -> (n, ceil(g / 64)) uint64 bit rows of the non-zero pattern (bit b of word w =
    gene 64 w + b), the input of the device pass."""
    B = (np.asarray(X) != 0)
    n, g = B.shape
    W = (g + 63) // 64
    padded = np.zeros((n, W * 64), dtype=bool)
    padded[:, :g] = B
    # little-endian bit order inside bytes, little-endian bytes inside the 64-bit word
    return np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little")).view("<u8").reshape(n, W)


def jaccard_pass(bits: np.ndarray, labels: np.ndarray, K: int, device: int = 0, return_distances: bool = False):
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n, W = bits.shape
    rowsum = np.empty((n, K))
    sq_all, sq_in = np.empty(n), np.empty(n)
    diam, sep = np.empty(K), np.empty((K, K))
    D = np.empty((n, n), dtype=np.float32) if return_distances else None
    ms = C.c_float(0.0)
    f64p, f32p = C.POINTER(C.c_double), C.POINTER(C.c_float)
    _lib.check(_lib.load().mi_jaccard_cluster_stats(
        bits.ctypes.data_as(C.POINTER(C.c_uint64)), n, W, labels.ctypes.data_as(C.POINTER(C.c_int32)), int(K),
        int(device), rowsum.ctypes.data_as(f64p), sq_all.ctypes.data_as(f64p), sq_in.ctypes.data_as(f64p),
        diam.ctypes.data_as(f64p), sep.ctypes.data_as(f64p), D.ctypes.data_as(f32p) if D is not None else None,
        C.byref(ms)))
    return {"rowsum": rowsum, "sq_all": sq_all, "sq_within": sq_in, "diameter": diam, "separation.matrix": sep,
            "distances": D, "kernel_ms": float(ms.value)}


def cluster_stats(X: np.ndarray, labels, device: int = 0, return_distances: bool = False) -> dict:
    """``X``: cells x genes expression (any dtype; only the zero / non-zero pattern enters the binary Jaccard
    distance) or pre-packed uint64 bit rows; ``labels``: one non-negative cluster id per cell.  Returns the
    distance-based fields of ``fpc::cluster.stats`` under fpc's names, ``sil.widths`` (``cluster::silhouette``)
    and ``within.average.distance`` (the per-cluster mean the notebook computes at :36)."""
    X = np.asarray(X)
    bits = X if X.dtype == np.uint64 else pack_expression(X)
    labels = np.asarray(labels)
    if labels.ndim != 1 or len(labels) != bits.shape[0]:
        raise ValueError("labels must have one entry per cell")
    uniq, lab = np.unique(labels, return_inverse=True)             # cluster ids need not be 0..K-1 (random colours)
    K, n = len(uniq), len(lab)
    r = jaccard_pass(bits, lab, K, device, return_distances)
    rs, sizes = r["rowsum"], np.bincount(lab, minlength=K).astype(np.int64)
    own = rs[np.arange(n), lab]
    a = np.where(sizes[lab] > 1, own / np.maximum(sizes[lab] - 1, 1), 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        means = rs / sizes[None, :]
    means[np.arange(n), lab] = np.inf
    b = means.min(axis=1) if K > 1 else np.zeros(n)
    m = np.maximum(a, b)
    sil = np.where((sizes[lab] > 1) & (K > 1) & (m > 0), (b - a) / np.where(m > 0, m, 1.0), 0.0)
    pair_sum = np.zeros((K, K))                                     # sum of d over ordered pairs (i in c, j in c')
    np.add.at(pair_sum, lab, rs)
    n_within = int((sizes * (sizes - 1) // 2).sum())
    n_between = n * (n - 1) // 2 - n_within
    within_sum = np.trace(pair_sum) / 2.0
    between_sum = (pair_sum.sum() - np.trace(pair_sum)) / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        avgd = np.where(sizes > 1, np.diag(pair_sum) / np.maximum(sizes * (sizes - 1), 1), np.nan)
        avbm = pair_sum / (sizes[:, None] * sizes[None, :])
    np.fill_diagonal(avbm, 0.0)
    sepm = r["separation.matrix"]
    off = ~np.eye(K, dtype=bool)
    sep = np.where(off, sepm, np.inf).min(axis=1) if K > 1 else np.full(K, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        toother = (pair_sum.sum(axis=1) - np.diag(pair_sum)) / (sizes * (n - sizes))
    wss = float((np.bincount(lab, weights=r["sq_within"], minlength=K) / 2.0 / np.maximum(sizes, 1)).sum())
    tss = float(r["sq_all"].sum() / 2.0 / n)
    diam = r["diameter"]
    avg_within, avg_between = float(a.sum() / n), (between_sum / n_between if n_between else float("nan"))
    # Pearson correlation of the pair distances with the 0/1 "different clusters" indicator
    npairs = n * (n - 1) // 2
    if n_between and n_within:
        sd, sdd = within_sum + between_sum, float(r["sq_all"].sum() / 2.0)
        cov = between_sum / npairs - (sd / npairs) * (n_between / npairs)
        var_d = sdd / npairs - (sd / npairs) ** 2
        var_i = (n_between / npairs) * (1.0 - n_between / npairs)
        gamma = cov / np.sqrt(var_d * var_i) if var_d > 0 and var_i > 0 else float("nan")
    else:
        gamma = float("nan")
    p = sizes / n
    out = {
        "n": n, "cluster.number": K, "cluster.ids": uniq, "cluster.size": sizes, "min.cluster.size": int(sizes.min()),
        "diameter": diam, "average.distance": avgd, "within.average.distance": avgd, "separation": sep,
        "average.toother": toother, "separation.matrix": sepm, "ave.between.matrix": avbm,
        "average.between": avg_between, "average.within": avg_within, "n.between": int(n_between),
        "n.within": n_within, "max.diameter": float(diam.max()), "min.separation": float(sep.min()),
        "within.cluster.ss": wss,
        "clus.avg.silwidths": np.bincount(lab, weights=sil, minlength=K) / np.maximum(sizes, 1),
        "avg.silwidth": float(sil.mean()), "sil.widths": sil, "pearsongamma": float(gamma),
        "dunn": float(sep.min() / diam.max()) if diam.max() > 0 else float("nan"),
        "dunn2": float(np.nanmin(np.where(off, avbm, np.nan)) / np.nanmax(avgd)) if K > 1 and np.any(sizes > 1) else float("nan"),
        "entropy": float(-(p * np.log(p)).sum()), "wb.ratio": avg_within / avg_between if n_between else float("nan"),
        "ch": ((tss - wss) / (K - 1)) / (wss / (n - K)) if K > 1 and n > K and wss > 0 else float("nan"),
        "kernel_ms": r["kernel_ms"],
    }
    if return_distances:
        out["distances"] = r["distances"]
    return out
