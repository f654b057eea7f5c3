"""CPU restatement of the cluster-quality numbers the reference publishes.  TEST INFRASTRUCTURE ONLY.

Reference: /root/reference/R/pbmc3k/Pbmc3k_benchmark_clusters.Rmd
   :36, :47, :69   mean(proxy::dist(cells x genes, method = "jaccard"))          within-cluster average distance
   :82-94          cluster::silhouette(labels, proxy::dist(..., "jaccard"))      silhouette widths
   :98-112         fpc::cluster.stats(dist, labels)  ->  R/pbmc3k/{QA,Seurat,Kmeans}_benchmark.csv
proxy, cluster and fpc are un-vendored R packages (absent here; R itself is absent), and the reference holds
the OUTPUT csv files of those calls but not their inputs, so nothing here can be checked against them:
**parity unpinned** vs the R packages.  What is restated is their published definitions:
   Jaccard (binary) dissimilarity  d = 1 - |A & B| / |A | B|  on the non-zero pattern of each cell's row
   silhouette  s(i) = (b - a) / max(a, b),  a = mean distance to the own cluster (others only), b = smallest
       mean distance to another cluster; s = 0 for singleton clusters           (Rousseeuw 1987; cluster::silhouette)
   cluster.stats fields (fpc manual): cluster.size, diameter, average.distance, separation, average.toother,
       separation.matrix, ave.between.matrix, average.between, average.within, n.between, n.within,
       max.diameter, min.separation, within.cluster.ss, clus.avg.silwidths, avg.silwidth, pearsongamma, dunn,
       dunn2, entropy, wb.ratio, ch
tests/test_metrics_oracle.py pins the silhouette against scikit-learn's independent implementation.
Two empty rows have distance 0 here (R gives NaN for 0/0); real cells are never empty.
"""
import numpy as np


def jaccard_distance_matrix(X):
    B = (np.asarray(X) != 0)
    Bi = B.astype(np.int64)
    inter = Bi @ Bi.T
    cnt = Bi.sum(axis=1)
    union = cnt[:, None] + cnt[None, :] - inter
    with np.errstate(invalid="ignore", divide="ignore"):
        D = np.where(union > 0, 1.0 - inter / np.maximum(union, 1), 0.0)
    np.fill_diagonal(D, 0.0)
    return D


def silhouette_widths(D, labels):
    labels = np.asarray(labels)
    n = len(labels)
    K = int(labels.max()) + 1
    sizes = np.bincount(labels, minlength=K)
    sums = np.stack([D[:, labels == c].sum(axis=1) for c in range(K)], axis=1)      # n x K
    s = np.zeros(n)
    for i in range(n):
        ci = labels[i]
        if sizes[ci] <= 1:
            continue
        a = sums[i, ci] / (sizes[ci] - 1)
        others = [sums[i, c] / sizes[c] for c in range(K) if c != ci and sizes[c] > 0]
        if not others:
            continue
        b = min(others)
        s[i] = (b - a) / max(a, b) if max(a, b) > 0 else 0.0
    return s


def cluster_stats(D, labels):
    """The distance-based fields of fpc::cluster.stats (no alt.clustering, no median / gap statistics)."""
    labels = np.asarray(labels)
    n = len(labels)
    K = int(labels.max()) + 1
    sizes = np.bincount(labels, minlength=K)
    same = labels[:, None] == labels[None, :]
    iu = np.triu_indices(n, 1)
    d_up, same_up = D[iu], same[iu]
    out = {"n": n, "cluster.number": K, "cluster.size": sizes, "min.cluster.size": int(sizes.min())}
    diam = np.zeros(K)
    avgd = np.full(K, np.nan)
    sep = np.full(K, np.inf)
    toother = np.full(K, np.nan)
    sepm = np.full((K, K), np.inf)
    avbm = np.full((K, K), np.nan)
    wss = 0.0
    for c in range(K):
        ic = np.where(labels == c)[0]
        sub = D[np.ix_(ic, ic)]
        if len(ic) > 1:
            up = sub[np.triu_indices(len(ic), 1)]
            diam[c] = up.max()
            avgd[c] = up.mean()
            wss += (up ** 2).sum() / len(ic)
        oc = np.where(labels != c)[0]
        if len(oc):
            cross = D[np.ix_(ic, oc)]
            sep[c] = cross.min()
            toother[c] = cross.mean()
        for c2 in range(K):
            if c2 != c and sizes[c2] and sizes[c]:
                blk = D[np.ix_(ic, np.where(labels == c2)[0])]
                sepm[c, c2] = blk.min()
                avbm[c, c2] = blk.mean()
    np.fill_diagonal(sepm, 0.0)
    np.fill_diagonal(avbm, 0.0)
    sil = silhouette_widths(D, labels)
    n_within = int(same_up.sum())
    n_between = int((~same_up).sum())
    # average.within: every observation has the same weight (mean over points of the mean distance to its cluster)
    sums_own = np.array([D[i, labels == labels[i]].sum() for i in range(n)])
    aw_terms = np.where(sizes[labels] > 1, sums_own / np.maximum(sizes[labels] - 1, 1), 0.0)
    avg_within = aw_terms.sum() / n
    avg_between = d_up[~same_up].mean() if n_between else np.nan
    p = sizes[sizes > 0] / n
    tss = (d_up ** 2).sum() / n
    out.update({
        "diameter": diam, "average.distance": avgd, "separation": sep, "average.toother": toother,
        "separation.matrix": sepm, "ave.between.matrix": avbm, "average.between": avg_between,
        "average.within": avg_within, "n.between": n_between, "n.within": n_within,
        "max.diameter": diam.max(), "min.separation": sep.min(), "within.cluster.ss": wss,
        "clus.avg.silwidths": np.array([sil[labels == c].mean() if sizes[c] else np.nan for c in range(K)]),
        "avg.silwidth": sil.mean(), "sil.widths": sil,
        "pearsongamma": float(np.corrcoef(d_up, (~same_up).astype(float))[0, 1]) if n_between and n_within else np.nan,
        "dunn": sep.min() / diam.max() if diam.max() > 0 else np.nan,
        "dunn2": np.nanmin(avbm[~np.eye(K, dtype=bool)]) / np.nanmax(avgd) if K > 1 else np.nan,
        "entropy": float(-(p * np.log(p)).sum()), "wb.ratio": avg_within / avg_between,
        "ch": ((tss - wss) / (K - 1)) / (wss / (n - K)) if K > 1 and n > K and wss > 0 else np.nan,
    })
    return out
