/*
 * mi_snn.h -- C ABI of the SNN-graph construction on MI355X (part of libmi_sa.so).
 *
 * This is the step immediately BEFORE the anneal path.  The reference does
 * it in R, outside its Python package:
 *     /root/reference/R/pbmc3k/Pbmc3k_prepare_data_for_QA_clustering.Rmd:67   FindNeighbors(dims = 1:dim,
 *          k.param = k, compute.SNN = TRUE, prune.SNN = coff)     kNN (self included) + Jaccard SNN
 *     :70-72   snn <- graphs[["SCT_snn"]] - diag(n)
 *     :75-79   sequential, in-place, symmetric top-`ord` trim of every column (stable order())
 *     (same loop: R/kidney/Kidney_data.Rmd:210-266, R/pbmc3k/Pbmc3k_general_data_preparation.Rmd:59-123)
 * and hands the result to Python as a GEXF file (create_graphs.py:5-8).  mi_snn_build_f32 replaces that whole
 * step: points in, trimmed SNN graph out (CSR of shared-neighbour COUNTS s_ij; the Jaccard weight is
 * w_ij = s_ij / (2k - s_ij), evaluated by the caller in fp64 exactly as R does).
 *
 * Arithmetic (mirrored bit for bit by oracle/snn_oracle.c): exact kNN with fp32 squared distances
 *     d(i,j) = fmaf chain over c of (x_ic - x_jc)^2,   neighbours ordered by (d, j);
 * everything after the kNN is integer work.  Seurat's default neighbour search is approximate (annoy): an
 * exact search is what its `nn.method = "rann", eps = 0` computes.
 *
 * Conventions as in mi_sa.h: plain C types, caller-allocated host arrays, 0 / negative MI_E* return codes,
 * mi_last_error() for the message.
 */
#ifndef MI_SNN_H
#define MI_SNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_snn_graph mi_snn_graph;

/* X: n x dim row-major fp32 (e.g. PCA coordinates), 1 <= dim <= 64; k = k.param (the point itself counts,
 * 2 <= k <= 64, k <= n); prune = prune.SNN (weights below it are dropped; 0 keeps everything);
 * ord = degree cap of the trim (<= 0: no trim).  Builds the graph on `device` and keeps it in HBM. */
int mi_snn_build_f32(const float *X, int n, int dim, int k, double prune, int ord, int device,
                     mi_snn_graph **out);

/* The optional variants of the notebooks' graph preparation (all `eval=FALSE` chunks the user runs by hand):
 *   MI_SNN_TRIM_UNSYMMETRIC  first trim: every COLUMN keeps its `ord` heaviest entries, rows are not touched
 *                            (Pbmc3k_general_data_preparation.Rmd:77-83, Kidney_data.Rmd:235-242): the matrix
 *                            becomes asymmetric;
 *   MI_SNN_ENHANCE_MUTUAL    "Enhance shared edges", Method 2 (Rmd :85-101, Kidney :252-266): entries present in
 *                            both directions get `bonus` added (2 in the PBMC notebook, 1 in the kidney one);
 *   MI_SNN_ENHANCE_SUM       A + t(A) (Rmd :103-113, Kidney :246-250, "Method 1");
 *   ord2 > 0                 "limitation of nodes degrees #2" (Rmd :116-123): the sequential symmetric trim again,
 *                            on the ENHANCED weights (ranked in fp64 as R ranks them).  Built for symmetric
 *                            matrices: after a symmetric first trim, or after A + t(A).
 * The stored rows are the COLUMNS of the result: entry e of row i with col[e] = r is A[r, i] (for a symmetric
 * result the distinction vanishes).  mi_snn_fetch_codes: per stored entry 0 = w, 1 = w + bonus, 2 = w + w with
 * w = shared / (2k - shared); the caller evaluates the weights in fp64 exactly as R does. */
#define MI_SNN_TRIM_UNSYMMETRIC 1u
#define MI_SNN_ENHANCE_MUTUAL   2u
#define MI_SNN_ENHANCE_SUM       4u
int mi_snn_build_ex_f32(const float *X, int n, int dim, int k, double prune, int ord, uint32_t flags, double bonus,
                        int ord2, int device, mi_snn_graph **out);
int mi_snn_fetch_codes(mi_snn_graph *g, uint8_t *code);

/* The ROUNDING variant of the notebooks (Pbmc3k_normalization_simulated_data.Rmd:597-616): after `- diag(n)`
 *     snn <- round(snn, digits = 2)                                    (:599, :602)
 *     snn[snn < 0.16 & snn != 0] <- -0.3      ("also negative edges", :603-605, id_type 3)
 * and then the same sequential symmetric trim (:611-616) -- which now ranks by the ROUNDED weights (rounding can make
 * different shared-neighbour counts tie; ties go to the lower row index, R's stable order()).  round_digits = the
 * `digits` argument (0 .. 6; the weight is rounded in fp64 to the nearest multiple of 10^-digits, halves to even);
 * negative_below > 0 turns every entry whose rounded weight is below it (and not 0) into a negative edge, 0 keeps all
 * entries positive.  R's order(decreasing = TRUE) puts a negative entry BELOW the zeros of its column, so the trim of
 * that column deletes it (with its mirror) whenever the column holds at least `ord` non-negative positions, zeros
 * included -- always, on a graph of more than a few dozen cells: with ord > 0 the negative entries are therefore
 * dropped BEFORE the trim and the result holds none (a graph with n - (entries of its densest column) < ord is refused
 * with MI_EUNSUPPORTED); with ord <= 0 (no trim) they stay and mi_snn_fetch_codes marks them 3.  The caller evaluates
 * round(shared / (2k - shared), digits) in fp64 and substitutes its negative value (-0.3) on code 3. */
int mi_snn_build_rounded_f32(const float *X, int n, int dim, int k, double prune, int ord, int round_digits,
                             double negative_below, int device, mi_snn_graph **out);

/* nnz = stored (directed) entries of the final graph (= 2 x edges when it is symmetric); max_degree over its rows. */
int mi_snn_info(const mi_snn_graph *g, int *n, int *k, int64_t *nnz, int *max_degree);

/* Copy to host (each pointer nullable): nn  n x k neighbour indices (column 0 = the point itself, then
 * ascending (distance, index)); rowptr n+1; col / shared nnz entries, rows ascending by column. */
int mi_snn_fetch(mi_snn_graph *g, int32_t *nn, int64_t *rowptr, int32_t *col, int32_t *shared);

/* Device time of the three stages of the build (HIP events), milliseconds. */
int mi_snn_kernel_ms(const mi_snn_graph *g, float *knn_ms, float *snn_ms, float *trim_ms);

int mi_snn_destroy(mi_snn_graph *g);

#ifdef __cplusplus
}
#endif
#endif /* MI_SNN_H */
