// metrics_kernels.hip -- all-pairs binary-Jaccard statistics for cluster-quality metrics (gfx950).
// C ABI: include/mi_metrics.h.  Replaces the O(n^2 g) core of proxy::dist + cluster::silhouette +
// fpc::cluster.stats in /root/reference/R/pbmc3k/Pbmc3k_benchmark_clusters.Rmd:36-112.
//
// M1 k_jaccard_stats: one thread per cell i (64 cells per workgroup), all other cells j streamed through an
// LDS tile of 16 bit-packed rows that every thread reads at the same address (broadcast); the thread's own
// row sits in LDS with an odd word stride (conflict-free).  Per pair: `words` x (AND, OR, 2 popcounts), one
// fp64 division.  Per-thread accumulators (distance sums per cluster, min distance per cluster) live in LDS as
// [cluster][thread].  The column range is cut into `gridDim.y` slices (one wave per 64 cells alone would leave
// most of the chip's 1024 SIMDs without a wave: 42 waves at n = 2638, 782 at n = 50 000); every (cell block,
// slice) workgroup writes its partial sums to its own plane and k_reduce_slices adds the planes in slice order,
// so every sum is still reproducible run to run.
// Bound: integer VALU (v_bcnt) / LDS broadcast reads; HBM traffic is the bit matrix once per workgroup.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mi_metrics.h"
#include "mi_sa_device.h"

namespace mi_sa_impl {
namespace {

constexpr int kMetRows = 64, kMetTile = 16;

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// The cells arrive SORTED BY CLUSTER (stable: index order inside a cluster; `orig` = original index).  The column
// j is the same for every thread of the wave, so "which cluster sum does this distance go to" is wave-uniform:
// the running sum / minimum of the current column cluster are two registers per thread, flushed when the cluster
// changes -- no per-thread accumulator table in LDS (it capped the residency at three waves per CU for K = 30).
// Per (cell, cluster) the distances are still added in index order.
__global__ void __launch_bounds__(kMetRows) k_jaccard_stats(const unsigned long long *__restrict__ bits, int n, int W,
                                                            const int *__restrict__ labels, const int *__restrict__ orig,
                                                            int K, int slice_len,
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
    int *labJ = reinterpret_cast<int *>(tileJ + (size_t)kMetTile * W);                  // [16]
    int *origJ = labJ + kMetTile;                                                        // [16]
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kMetRows + tid;
    const bool live = i < n;
    for (int w = 0; w < W; ++w) tileI[tid * WS + w] = live ? bits[(size_t)i * W + w] : 0ull;
    const int ci = live ? labels[i] : -1;
    const int oi = live ? orig[i] : 0;
    const size_t plane = blockIdx.y;                                          // this slice's plane of partial sums
    // one cluster for the whole wave (the usual case after sorting): one atomic per wave instead of one per thread
    const int c_first = __builtin_amdgcn_readfirstlane(ci);
    const bool uniform_ci = __ballot(ci == c_first || !live) == ~0ull;
    double sq_all = 0.0, sq_in = 0.0, dmax = 0.0;
    int cur = -1;                                                             // cluster of the columns being summed
    double acc = 0.0, mn = INFINITY;
    auto flush = [&]() {                                                      // wave-uniform control flow
        if (cur < 0) return;
        if (live) rowsum[(plane * n + oi) * K + cur] = acc;                   // one run per (cell, cluster, slice)
        if (uniform_ci) {
            if (cur != c_first) {
                const double m = wave_min_f64(live ? mn : INFINITY);
                if (tid == 0 && c_first >= 0) atomicMin(&sep_bits[c_first * K + cur], (unsigned long long)__double_as_longlong(m));
            }
        } else if (live && cur != ci) {
            atomicMin(&sep_bits[ci * K + cur], (unsigned long long)__double_as_longlong(mn));
        }
        acc = 0.0;
        mn = INFINITY;
    };
    const int j_begin = blockIdx.y * slice_len;                               // slice_len is a multiple of the tile
    const int j_end = j_begin + slice_len < n ? j_begin + slice_len : n;
    for (int j0 = j_begin; j0 < j_end; j0 += kMetTile) {
        __syncthreads();
        for (int e = tid; e < kMetTile * W; e += kMetRows) {
            const int jj = e / W, w = e - jj * W;
            tileJ[e] = (j0 + jj < n) ? bits[(size_t)(j0 + jj) * W + w] : 0ull;
        }
        if (tid < kMetTile) {
            labJ[tid] = (j0 + tid < n) ? labels[j0 + tid] : 0;
            origJ[tid] = (j0 + tid < n) ? orig[j0 + tid] : 0;
        }
        __syncthreads();
        const int lim = j_end - j0 < kMetTile ? j_end - j0 : kMetTile;
        for (int jj = 0; jj < lim; ++jj) {
            const int cj = labJ[jj];                                          // the same for every thread
            if (cj != cur) { flush(); cur = cj; }
            int inter = 0, uni = 0;
            for (int w = 0; w < W; ++w) {
                const unsigned long long x = tileI[tid * WS + w], y = tileJ[jj * W + w];
                inter += __popcll(x & y);
                uni += __popcll(x | y);
            }
            const int j = j0 + jj;
            const double d = (uni > 0 && j != i) ? 1.0 - (double)inter / (double)uni : 0.0;
            if (live && out_D) out_D[(size_t)oi * n + origJ[jj]] = (float)d;
            if (!live || j == i) continue;
            acc += d;
            sq_all += d * d;
            if (cj == ci) { sq_in += d * d; dmax = d > dmax ? d : dmax; }
            else mn = d < mn ? d : mn;
        }
    }
    flush();
    if (live) {
        rowsq_all[plane * n + oi] = sq_all;
        rowsq_within[plane * n + oi] = sq_in;
    }
    // non-negative doubles order like their bit patterns
    if (uniform_ci) {
        const double m = wave_max_f64(live ? dmax : 0.0);
        if (tid == 0 && c_first >= 0) atomicMax(&diam_bits[c_first], (unsigned long long)__double_as_longlong(m));
    } else if (live) {
        atomicMax(&diam_bits[ci], (unsigned long long)__double_as_longlong(dmax));
    }
}

// out[e] = sum over the S planes of partial[s][e], in slice order
__global__ void __launch_bounds__(256) k_reduce_slices(const double *__restrict__ partial, int S, size_t count,
                                                       double *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    double v = 0.0;
    for (int sidx = 0; sidx < S; ++sidx) v += partial[(size_t)sidx * count + e];
    out[e] = v;
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
    const size_t lds = ((size_t)kMetRows * (words | 1) + (size_t)kMetTile * words) * 8 + 2 * kMetTile * 4 + 64;
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "gene rows of %d words exceed the LDS plan (%zu B)", words, lds);
    for (int i = 0; i < n; ++i)
        if (labels[i] < 0 || labels[i] >= K) return fail(MI_EINVAL, "label %d of cell %d outside [0, %d)", labels[i], i, K);
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(MI_ENODEV, "no HIP device visible");
    if (device < 0 || device >= cnt) return fail(MI_EINVAL, "device %d out of range [0,%d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_bits = nullptr, *d_diam = nullptr, *d_sep = nullptr;
    int *d_lab = nullptr, *d_orig = nullptr;
    double *d_rowsum = nullptr, *d_sqa = nullptr, *d_sqw = nullptr, *d_part = nullptr;
    // column slices: about four waves per SIMD in total, whole tiles per slice
    const int blocks = (n + kMetRows - 1) / kMetRows;
    int S = (4096 + blocks - 1) / blocks;
    S = S < 1 ? 1 : (S > 64 ? 64 : S);
    const int slice_len = (((n + S - 1) / S + kMetTile - 1) / kMetTile) * kMetTile;
    S = (n + slice_len - 1) / slice_len;
    const size_t per_plane = (size_t)n * K + 2 * (size_t)n;                   // rowsum, rowsq_all, rowsq_within
    float *d_D = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = guarded([&]() -> int {
        HIP_TRY(hipMalloc((void **)&d_bits, (size_t)n * words * 8));
        HIP_TRY(hipMalloc((void **)&d_lab, (size_t)n * 4));
        HIP_TRY(hipMalloc((void **)&d_rowsum, (size_t)n * K * 8));
        HIP_TRY(hipMalloc((void **)&d_sqa, (size_t)n * 8));
        HIP_TRY(hipMalloc((void **)&d_sqw, (size_t)n * 8));
        HIP_TRY(hipMalloc((void **)&d_part, (size_t)S * per_plane * 8));
        HIP_TRY(hipMalloc((void **)&d_diam, (size_t)K * 8));
        HIP_TRY(hipMalloc((void **)&d_sep, (size_t)K * K * 8));
        if (out_D) HIP_TRY(hipMalloc((void **)&d_D, (size_t)n * n * 4));
        // cells sorted by cluster (counting sort: stable, index order inside a cluster)
        std::vector<int> start((size_t)K + 1, 0), order((size_t)n), lab_s((size_t)n);
        for (int i = 0; i < n; ++i) start[labels[i] + 1]++;
        for (int c = 0; c < K; ++c) start[c + 1] += start[c];
        for (int i = 0; i < n; ++i) order[start[labels[i]]++] = i;
        std::vector<uint64_t> bits_s((size_t)n * words);
        for (int p = 0; p < n; ++p) {
            lab_s[p] = labels[order[p]];
            memcpy(&bits_s[(size_t)p * words], &bits[(size_t)order[p] * words], (size_t)words * 8);
        }
        HIP_TRY(hipMalloc((void **)&d_orig, (size_t)n * 4));
        HIP_TRY(hipMemcpy(d_bits, bits_s.data(), (size_t)n * words * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_lab, lab_s.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_orig, order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(d_part, 0, (size_t)S * per_plane * 8));              // clusters absent from a slice stay 0
        HIP_TRY(hipMemset(d_diam, 0, (size_t)K * 8));
        std::vector<double> inf((size_t)K * K, INFINITY);
        HIP_TRY(hipMemcpy(d_sep, inf.data(), inf.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_jaccard_stats), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipEventRecord(e0, 0));
        // planes: [S][n][K] distance sums, then [S][n] squared sums (all), then [S][n] squared sums (within)
        double *p_rowsum = d_part, *p_sqa = d_part + (size_t)S * n * K, *p_sqw = p_sqa + (size_t)S * n;
        hipLaunchKernelGGL(k_jaccard_stats, dim3(blocks, S), dim3(kMetRows), lds, 0, d_bits, n, words, d_lab, d_orig, K, slice_len,
                           p_rowsum, p_sqa, p_sqw, d_diam, d_sep, d_D);
        HIP_TRY(hipGetLastError());
        const size_t c1 = (size_t)n * K, c2 = (size_t)n;
        hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)((c1 + 255) / 256)), dim3(256), 0, 0, p_rowsum, S, c1, d_rowsum);
        hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)((c2 + 255) / 256)), dim3(256), 0, 0, p_sqa, S, c2, d_sqa);
        hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)((c2 + 255) / 256)), dim3(256), 0, 0, p_sqw, S, c2, d_sqw);
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
    });
    void *bufs[] = {d_orig, d_bits, d_lab, d_rowsum, d_sqa, d_sqw, d_part, d_diam, d_sep, d_D};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}
