// metrics_kernels.hip -- all-pairs binary-Jaccard statistics for cluster-quality metrics (gfx950).
// C ABI: include/mi_metrics.h.  Replaces the O(n^2 g) core of proxy::dist + cluster::silhouette +
// fpc::cluster.stats in /root/reference/R/pbmc3k/Pbmc3k_benchmark_clusters.Rmd:36-112.
//
// M1 k_jaccard_stats: one thread per cell i (64 cells per workgroup), all other cells j streamed through an
// LDS tile of 16 bit-packed rows that every thread reads at the same address (broadcast); the thread's own
// row sits in LDS with an odd word stride (conflict-free).  Per pair: `words` x (AND, OR, 2 popcounts), one
// fp64 division.  Per-thread accumulators (distance sums per cluster, min distance per cluster) live in LDS as
// [cluster][thread].  Columns are visited in index order, so every sum is reproducible run to run.
// Bound: integer VALU (v_bcnt) / LDS broadcast reads; HBM traffic is the bit matrix once per workgroup.
#include <cmath>
#include <vector>

#include "../../include/mi_metrics.h"
#include "mi_sa_device.h"

namespace mi_sa_impl {
namespace {

constexpr int kMetRows = 64, kMetTile = 16;

__global__ void __launch_bounds__(kMetRows) k_jaccard_stats(const unsigned long long *__restrict__ bits, int n, int W,
                                                            const int *__restrict__ labels, int K,
                                                            double *__restrict__ rowsum, double *__restrict__ rowsq_all,
                                                            double *__restrict__ rowsq_within,
                                                            unsigned long long *__restrict__ diam_bits,
                                                            unsigned long long *__restrict__ sep_bits,
                                                            float *__restrict__ out_D)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int WS = W | 1;                                                     // odd stride: conflict-free own rows
    unsigned long long *tileI = reinterpret_cast<unsigned long long *>(lds);            // [64][WS]
    unsigned long long *tileJ = tileI + (size_t)kMetRows * WS;                           // [16][W]
    double *acc = reinterpret_cast<double *>(tileJ + (size_t)kMetTile * W);             // [K][64] distance sums
    double *mind = acc + (size_t)K * kMetRows;                                           // [K][64] min distances
    int *labJ = reinterpret_cast<int *>(mind + (size_t)K * kMetRows);                    // [16]
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kMetRows + tid;
    const bool live = i < n;
    for (int w = 0; w < W; ++w) tileI[tid * WS + w] = live ? bits[(size_t)i * W + w] : 0ull;
    for (int c = 0; c < K; ++c) { acc[c * kMetRows + tid] = 0.0; mind[c * kMetRows + tid] = INFINITY; }
    const int ci = live ? labels[i] : 0;
    double sq_all = 0.0, sq_in = 0.0, dmax = 0.0;
    for (int j0 = 0; j0 < n; j0 += kMetTile) {
        __syncthreads();
        for (int e = tid; e < kMetTile * W; e += kMetRows) {
            const int jj = e / W, w = e - jj * W;
            tileJ[e] = (j0 + jj < n) ? bits[(size_t)(j0 + jj) * W + w] : 0ull;
        }
        if (tid < kMetTile) labJ[tid] = (j0 + tid < n) ? labels[j0 + tid] : 0;
        __syncthreads();
        const int lim = n - j0 < kMetTile ? n - j0 : kMetTile;
        for (int jj = 0; jj < lim; ++jj) {
            int inter = 0, uni = 0;
            for (int w = 0; w < W; ++w) {
                const unsigned long long x = tileI[tid * WS + w], y = tileJ[jj * W + w];
                inter += __popcll(x & y);
                uni += __popcll(x | y);
            }
            const int j = j0 + jj;
            const double d = (uni > 0 && j != i) ? 1.0 - (double)inter / (double)uni : 0.0;
            if (live && out_D) out_D[(size_t)i * n + j] = (float)d;
            if (!live || j == i) continue;
            const int cj = labJ[jj];
            acc[cj * kMetRows + tid] += d;
            sq_all += d * d;
            if (cj == ci) { sq_in += d * d; dmax = d > dmax ? d : dmax; }
            else { const double m0 = mind[cj * kMetRows + tid]; mind[cj * kMetRows + tid] = d < m0 ? d : m0; }
        }
    }
    if (!live) return;
    for (int c = 0; c < K; ++c) rowsum[(size_t)i * K + c] = acc[c * kMetRows + tid];
    rowsq_all[i] = sq_all;
    rowsq_within[i] = sq_in;
    // non-negative doubles order like their bit patterns
    atomicMax(&diam_bits[ci], (unsigned long long)__double_as_longlong(dmax));
    for (int c = 0; c < K; ++c)
        if (c != ci) atomicMin(&sep_bits[ci * K + c], (unsigned long long)__double_as_longlong(mind[c * kMetRows + tid]));
}

}  // namespace
}  // namespace mi_sa_impl
using namespace mi_sa_impl;

extern "C" int mi_jaccard_cluster_stats(const uint64_t *bits, int n, int words, const int32_t *labels, int K, int device,
                                        double *rowsum, double *rowsq_all, double *rowsq_within, double *diameter,
                                        double *separation, float *out_D, float *out_kernel_ms)
{
    if (!bits || !labels || !rowsum || !rowsq_all || !rowsq_within || !diameter || !separation)
        return fail(MI_EINVAL, "NULL argument");
    if (n < 1 || words < 1 || K < 1) return fail(MI_EINVAL, "n, words and K must be >= 1");
    if (K > 64) return fail(MI_EUNSUPPORTED, "at most 64 clusters (got %d)", K);
    const size_t lds = ((size_t)kMetRows * (words | 1) + (size_t)kMetTile * words) * 8 + 2 * (size_t)K * kMetRows * 8 + 64;
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "gene rows of %d words with %d clusters exceed the LDS plan (%zu B)", words, K, lds);
    for (int i = 0; i < n; ++i)
        if (labels[i] < 0 || labels[i] >= K) return fail(MI_EINVAL, "label %d of cell %d outside [0, %d)", labels[i], i, K);
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(MI_ENODEV, "no HIP device visible");
    if (device < 0 || device >= cnt) return fail(MI_EINVAL, "device %d out of range [0,%d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_bits = nullptr, *d_diam = nullptr, *d_sep = nullptr;
    int *d_lab = nullptr;
    double *d_rowsum = nullptr, *d_sqa = nullptr, *d_sqw = nullptr;
    float *d_D = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&d_bits, (size_t)n * words * 8));
        HIP_TRY(hipMalloc((void **)&d_lab, (size_t)n * 4));
        HIP_TRY(hipMalloc((void **)&d_rowsum, (size_t)n * K * 8));
        HIP_TRY(hipMalloc((void **)&d_sqa, (size_t)n * 8));
        HIP_TRY(hipMalloc((void **)&d_sqw, (size_t)n * 8));
        HIP_TRY(hipMalloc((void **)&d_diam, (size_t)K * 8));
        HIP_TRY(hipMalloc((void **)&d_sep, (size_t)K * K * 8));
        if (out_D) HIP_TRY(hipMalloc((void **)&d_D, (size_t)n * n * 4));
        HIP_TRY(hipMemcpy(d_bits, bits, (size_t)n * words * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_lab, labels, (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_diam, 0, (size_t)K * 8));
        std::vector<double> inf((size_t)K * K, INFINITY);
        HIP_TRY(hipMemcpy(d_sep, inf.data(), inf.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_jaccard_stats), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_jaccard_stats, dim3((n + kMetRows - 1) / kMetRows), dim3(kMetRows), lds, 0, d_bits, n, words, d_lab, K,
                           d_rowsum, d_sqa, d_sqw, d_diam, d_sep, d_D);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, 0));
        HIP_TRY(hipEventSynchronize(e1));
        if (out_kernel_ms) HIP_TRY(hipEventElapsedTime(out_kernel_ms, e0, e1));
        HIP_TRY(hipMemcpy(rowsum, d_rowsum, (size_t)n * K * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(rowsq_all, d_sqa, (size_t)n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(rowsq_within, d_sqw, (size_t)n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(diameter, d_diam, (size_t)K * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(separation, d_sep, (size_t)K * K * 8, hipMemcpyDeviceToHost));
        for (int c = 0; c < K; ++c) separation[c * K + c] = 0.0;
        if (out_D) HIP_TRY(hipMemcpy(out_D, d_D, (size_t)n * n * 4, hipMemcpyDeviceToHost));
        return MI_OK;
    }();
    void *bufs[] = {d_bits, d_lab, d_rowsum, d_sqa, d_sqw, d_diam, d_sep, d_D};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}
