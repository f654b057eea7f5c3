"""Pins oracle/metrics_oracle.py: the Jaccard distance matrix against scipy's, the silhouette widths against
scikit-learn's independent implementation, and a hand-computed 4-point case.  CPU only.  (The R packages
the reference calls -- proxy, cluster, fpc -- are absent; see the oracle's header.)"""
import numpy as np
from scipy.spatial.distance import pdist, squareform
from sklearn.metrics import silhouette_samples

from oracle import metrics_oracle as mo


def test_distance_and_silhouette_against_independent_implementations():
    rng = np.random.RandomState(0)
    X = (rng.rand(200, 300) < 0.15) * rng.rand(200, 300)
    lab = rng.randint(0, 5, size=200)
    D = mo.jaccard_distance_matrix(X)
    assert np.allclose(D, squareform(pdist(X != 0, "jaccard")), rtol=0, atol=1e-15)
    assert np.allclose(mo.silhouette_widths(D, lab), silhouette_samples(D, lab, metric="precomputed"), rtol=0, atol=1e-15)


def test_hand_computed_case():
    # genes: A={0,1}, B={0,1}, C={2,3}, D={2}  -> d(A,B)=0, d(C,D)=1/2, every cross distance 1
    X = np.array([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 0]])
    st = mo.cluster_stats(mo.jaccard_distance_matrix(X), np.array([0, 0, 1, 1]))
    assert st["diameter"].tolist() == [0.0, 0.5] and st["separation"].tolist() == [1.0, 1.0]
    assert st["average.between"] == 1.0 and st["average.within"] == 0.25 and st["dunn"] == 2.0
    assert st["n.within"] == 2 and st["n.between"] == 4
    assert np.allclose(st["sil.widths"], [1.0, 1.0, 0.5, 0.5])
    assert st["within.cluster.ss"] == 0.125
