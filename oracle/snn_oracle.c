/*
 * snn_oracle.c -- CPU restatement of the SNN-graph construction that feeds the clustering path.
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and nothing else.
 *
 * What it restates (the reference does this step in R, outside its Python package):
 *   /root/reference/R/pbmc3k/Pbmc3k_prepare_data_for_QA_clustering.Rmd:67  FindNeighbors(dims=1:dim,
 *        k.param=k, compute.SNN=TRUE, prune.SNN=coff)          -> kNN (self included) + Jaccard SNN
 *   :70-72   snn <- graphs[["SCT_snn"]] - diag(n)              -> zero diagonal
 *   :75-79   for (i in 1:n) { to_delete <- order(snn[,i], decreasing=TRUE)[(ord+1):n];
 *                             snn[,i][to_delete] <- 0; snn[i,][to_delete] <- 0 }
 *            -> sequential, in-place, symmetric top-`ord` trim (R's order() is stable: ties keep index order)
 * Seurat itself is an un-vendored R dependency (absent here); its published SNN definition is restated:
 *   N(i) = the k nearest points of i, i included;  s_ij = |N(i) & N(j)|;  w_ij = s_ij / (2k - s_ij);
 *   w_ij < prune -> 0.  Seurat's default neighbour search is approximate (annoy); this restates the exact one.
 * Parity pin: tests/test_snn_oracle.py checks this file against a literal dense numpy restatement of the
 * Rmd lines above (scrna_seq_qannealing_clustering_amd/graphs.py:snn_from_points) on seeded point clouds.
 *
 * Arithmetic fixed here and mirrored bit for bit by csrc/snn_kernels.hip:
 *   d(i,j) = fp32 chain  d = fmaf(x_ic - x_jc, x_ic - x_jc, d), c = 0..dim-1 ;  neighbours ordered by (d, j).
 * Everything after the kNN is integer work on s_ij (the trim ranks by s: w is monotone in s).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* nn[i*k + 0] = i ; nn[i*k + 1..k-1] = the k-1 nearest other points, ascending (d, j). */
int orc_knn_f32(const float *X, int n, int dim, int k, int32_t *nn)
{
    if (k < 1 || k > n) return -1;
    const int kk = k - 1;
#pragma omp parallel
    {
        float *bd = (float *)malloc(sizeof(float) * (size_t)(kk > 0 ? kk : 1));
        int32_t *bj = (int32_t *)malloc(sizeof(int32_t) * (size_t)(kk > 0 ? kk : 1));
#pragma omp for schedule(static)
        for (int i = 0; i < n; ++i) {
            int cnt = 0;
            const float *xi = X + (size_t)i * dim;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const float *xj = X + (size_t)j * dim;
                float d = 0.0f;
                for (int c = 0; c < dim; ++c) {
                    const float diff = xi[c] - xj[c];
                    d = fmaf(diff, diff, d);
                }
                if (cnt == kk && !(d < bd[kk - 1])) continue;        /* j ascending: equal d never displaces */
                if (kk == 0) continue;
                int p = cnt < kk ? cnt : kk - 1;
                while (p > 0 && d < bd[p - 1]) { bd[p] = bd[p - 1]; bj[p] = bj[p - 1]; --p; }
                bd[p] = d; bj[p] = j;
                if (cnt < kk) ++cnt;
            }
            nn[(size_t)i * k] = i;
            for (int p = 0; p < kk; ++p) nn[(size_t)i * k + 1 + p] = bj[p];
        }
        free(bd); free(bj);
    }
    return 0;
}

/* reverse neighbour lists: rn_idx[rn_ptr[m] .. rn_ptr[m+1]) = { j : m in N(j) }, ascending j */
static void reverse_lists(const int32_t *nn, int n, int k, int64_t *rn_ptr, int32_t *rn_idx)
{
    memset(rn_ptr, 0, sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t e = 0; e < (int64_t)n * k; ++e) rn_ptr[nn[e] + 1]++;
    for (int m = 0; m < n; ++m) rn_ptr[m + 1] += rn_ptr[m];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    memcpy(cur, rn_ptr, sizeof(int64_t) * (size_t)n);
    for (int j = 0; j < n; ++j)
        for (int p = 0; p < k; ++p) rn_idx[cur[nn[(size_t)j * k + p]]++] = j;
    free(cur);
}

/* Shared-neighbour rows.  Two calls: col == NULL counts (fills rowptr[0..n]); otherwise fills col/shared
 * (rows ascending by column).  Entries: j != i, s_ij > 0, s/(2k-s) >= prune (fp64). */
int orc_snn_rows(const int32_t *nn, int n, int k, double prune, int64_t *rowptr, int32_t *col, int32_t *shared)
{
    int64_t *rn_ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int32_t *rn_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * k);
    reverse_lists(nn, n, k, rn_ptr, rn_idx);
    if (!col) rowptr[0] = 0;
#pragma omp parallel
    {
        int32_t *cnt = (int32_t *)calloc((size_t)n, sizeof(int32_t));
        int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n; ++i) {
            int nc = 0;
            for (int p = 0; p < k; ++p) {
                const int m = nn[(size_t)i * k + p];
                for (int64_t e = rn_ptr[m]; e < rn_ptr[m + 1]; ++e) {
                    const int j = rn_idx[e];
                    if (cnt[j]++ == 0) cand[nc++] = j;
                }
            }
            /* ascending column order */
            for (int a = 1; a < nc; ++a) {
                const int32_t v = cand[a];
                int b = a;
                while (b > 0 && cand[b - 1] > v) { cand[b] = cand[b - 1]; --b; }
                cand[b] = v;
            }
            int64_t out = col ? rowptr[i] : 0;
            int64_t deg = 0;
            for (int a = 0; a < nc; ++a) {
                const int j = cand[a];
                const int s = cnt[j];
                cnt[j] = 0;
                if (j == i) continue;
                if ((double)s / (2.0 * k - (double)s) < prune) continue;
                if (col) { col[out] = j; shared[out] = s; ++out; }
                ++deg;
            }
            if (!col) rowptr[i + 1] = deg;
        }
        free(cnt); free(cand);
    }
    if (!col)
        for (int i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    free(rn_ptr); free(rn_idx);
    return 0;
}

/* The Rmd :75-79 loop on the CSR form.  alive[e] (one byte per stored entry, all 1 on entry) is cleared for
 * every deleted entry and its mirror.  ord <= 0: nothing is trimmed. */
int orc_snn_trim(int n, const int64_t *rowptr, const int32_t *col, const int32_t *shared, int ord, uint8_t *alive)
{
    if (ord <= 0) return 0;
    int64_t maxdeg = 0;
    for (int i = 0; i < n; ++i)
        if (rowptr[i + 1] - rowptr[i] > maxdeg) maxdeg = rowptr[i + 1] - rowptr[i];
    int64_t *ent = (int64_t *)malloc(sizeof(int64_t) * (size_t)(maxdeg > 0 ? maxdeg : 1));
    for (int i = 0; i < n; ++i) {
        int cnt = 0;
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e)
            if (alive[e]) ent[cnt++] = e;
        if (cnt <= ord) continue;
        /* stable order by shared count descending (entries are already ascending by column) */
        for (int a = 1; a < cnt; ++a) {
            const int64_t v = ent[a];
            int b = a;
            while (b > 0 && shared[ent[b - 1]] < shared[v]) { ent[b] = ent[b - 1]; --b; }
            ent[b] = v;
        }
        for (int a = ord; a < cnt; ++a) {
            const int64_t e = ent[a];
            alive[e] = 0;
            const int j = col[e];
            int64_t lo = rowptr[j], hi = rowptr[j + 1] - 1;
            while (lo <= hi) {
                const int64_t mid = (lo + hi) / 2;
                if (col[mid] == i) { alive[mid] = 0; break; }
                if (col[mid] < i) lo = mid + 1; else hi = mid - 1;
            }
        }
    }
    free(ent);
    return 0;
}

/* ---- the optional variants of the notebooks (Pbmc3k_general_data_preparation.Rmd:77-123, Kidney_data.Rmd:235-266) ----
 * The stored rows double as the COLUMNS of the symmetric SNN matrix: entry e of row i with col[e] = r is A[r, i]. */

/* :77-83 "UNSYMMETRIC": for (i) { to_delete <- order(snn[,i], decreasing=TRUE)[(ord+1):n]; snn[,i][to_delete] <- 0 }
 * -- only column i is written, so the columns are independent.  key = what the column is ranked by. */
int orc_snn_trim_cols(int n, const int64_t *rowptr, const int32_t *key, int ord, uint8_t *alive)
{
    if (ord <= 0) return 0;
    for (int i = 0; i < n; ++i) {
        const int64_t b = rowptr[i], e1 = rowptr[i + 1];
        if (e1 - b <= ord) continue;
        for (int64_t e = b; e < e1; ++e) {
            int rank = 0;
            for (int64_t f = b; f < e1; ++f)                      /* stable order(): heavier first, ties by row index */
                rank += (key[f] > key[e] || (key[f] == key[e] && f < e)) ? 1 : 0;
            if (rank >= ord) alive[e] = 0;
        }
    }
    return 0;
}

static int64_t mirror_of(const int64_t *rowptr, const int32_t *col, int r, int i)
{
    int64_t lo = rowptr[r], hi = rowptr[r + 1] - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) / 2;
        if (col[mid] == i) return mid;
        if (col[mid] < i) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* mode 1 (:85-101): mutual[i,] <- snn[i,] & snn[,i]; snn[i,] <- old[i,] + bonus * mutual[i,]  -> code 1 on entries present
 * in both directions, support unchanged.  mode 2 (:103-113): snn[i,] <- old[i,] + old[,i]  -> support = union, code 2 where
 * both were present (the weight doubles). */
int orc_snn_enhance(int n, int mode, const int64_t *rowptr, const int32_t *col, const uint8_t *alive_in,
                    uint8_t *alive_out, uint8_t *code)
{
    for (int i = 0; i < n; ++i)
        for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
            const int64_t m = mirror_of(rowptr, col, col[e], i);
            const int here = alive_in[e] != 0, there = m >= 0 && alive_in[m] != 0;
            if (mode == 1) { alive_out[e] = (uint8_t)here; code[e] = (uint8_t)(here && there ? 1 : 0); }
            else { alive_out[e] = (uint8_t)(here || there); code[e] = (uint8_t)(here && there ? 2 : 0); }
        }
    return 0;
}
