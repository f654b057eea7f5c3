"""Cluster-quality metrics on the GPU (csrc/metrics_kernels.hip through include/mi_metrics.h) against
oracle/metrics_oracle.py.  Distances are fp64 on both sides; sums differ only by summation order
(tolerance 1e-12 relative).  GPU only."""
import numpy as np
import pytest

from oracle import metrics_oracle as mo
from scrna_seq_qannealing_clustering_amd import _lib, metrics

pytestmark = pytest.mark.gpu

SCALARS = ("average.between", "average.within", "max.diameter", "min.separation", "within.cluster.ss", "avg.silwidth",
           "pearsongamma", "dunn", "dunn2", "entropy", "wb.ratio", "ch")
VECTORS = ("cluster.size", "diameter", "average.distance", "separation", "average.toother", "separation.matrix",
           "ave.between.matrix", "clus.avg.silwidths", "sil.widths")


def expression(n, g, seed, density=0.12):
    rng = np.random.RandomState(seed)
    groups = rng.randint(0, 4, size=n)
    base = rng.rand(4, g) < density
    return ((base[groups] ^ (rng.rand(n, g) < 0.05)) * rng.rand(n, g)).astype(np.float32), groups


@pytest.mark.parametrize("n,g,K", [(300, 500, 4), (513, 64, 9), (65, 1000, 2), (1000, 3001, 15), (40, 7, 1)])
def test_cluster_stats_equal_oracle(n, g, K):
    X, groups = expression(n, g, seed=n + g)
    labels = np.random.RandomState(K).randint(0, K, size=n) if K != 4 else groups
    labels[:K] = np.arange(K)                                   # every cluster non-empty
    st = metrics.cluster_stats(X, labels, return_distances=True)
    D = mo.jaccard_distance_matrix(X)
    ref = mo.cluster_stats(D, labels)
    assert np.allclose(st["distances"], D.astype(np.float32), rtol=0, atol=1e-7)
    assert st["n"] == ref["n"] and st["cluster.number"] == ref["cluster.number"]
    assert st["n.within"] == ref["n.within"] and st["n.between"] == ref["n.between"]
    for k in VECTORS:
        assert np.allclose(st[k], ref[k], rtol=1e-12, atol=1e-12, equal_nan=True), k
    for k in SCALARS:
        assert np.isclose(st[k], ref[k], rtol=1e-10, atol=1e-12, equal_nan=True), k


def test_arbitrary_cluster_ids_singletons_and_empty_rows():
    """The reference's labels are random colour integers (BQM_clustering.py:116-124), not 0..K-1; singletons
    have silhouette 0; two cells without any expressed gene have distance 0."""
    X, _ = expression(120, 90, seed=5)
    X[3] = 0
    X[77] = 0
    labels = np.array([17, 203, 5, 88])[np.random.RandomState(2).randint(0, 4, size=120)]
    labels[10] = 999                                            # a singleton cluster
    st = metrics.cluster_stats(X, labels)
    uniq, lab = np.unique(labels, return_inverse=True)
    ref = mo.cluster_stats(mo.jaccard_distance_matrix(X), lab)
    assert st["cluster.ids"].tolist() == uniq.tolist()
    assert st["sil.widths"][10] == 0.0
    for k in ("sil.widths", "diameter", "separation.matrix", "ave.between.matrix"):
        assert np.allclose(st[k], ref[k], rtol=1e-12, atol=1e-12, equal_nan=True), k


def test_limits_are_reported():
    bits = np.zeros((10, 300), dtype=np.uint64)                 # 640 B per word of gene row > 160 KB of LDS
    with pytest.raises(_lib.MiSaError) as ei:
        metrics.jaccard_pass(bits, np.zeros(10, dtype=np.int32), 1)
    assert ei.value.code == -5
    with pytest.raises(_lib.MiSaError):
        metrics.jaccard_pass(np.zeros((4, 1), dtype=np.uint64), np.array([0, 1, 2, 5], dtype=np.int32), 3)
