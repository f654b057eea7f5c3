// snn_kernels.hip -- SNN-graph construction on MI355X (gfx950): exact kNN -> shared-neighbour counts ->
// prune -> zero diagonal -> sequential symmetric top-`ord` trim.  C ABI: include/mi_snn.h.
//
// Replaces the R step that produces the reference's input graphs
// (/root/reference/R/pbmc3k/Pbmc3k_prepare_data_for_QA_clustering.Rmd:67-79; the GEXF file it writes is
// what create_graphs.py:5-8 loads).  Arithmetic is fixed in oracle/snn_oracle.c and mirrored here:
//   d(i,j) = fp32 fmaf chain over the coordinates; neighbours ordered by (d, j); the rest is integers.
//
// Kernels
//   S1 k_knn<DP,KM>    one query point per thread, candidates streamed through an LDS tile that every thread
//                      reads at the same address (broadcast), running top-KM list in registers (unrolled
//                      compare-swap insertion).  VALU-bound: n^2 * dim fused multiply-adds.
//   S2 k_rn_*          reverse-neighbour lists RN(m) = { j : m in N(j) } (count, scan, fill).
//   S3 k_snn_rows<P>   one workgroup per row i, persistent: shared counts s_ij accumulated in byte counters in
//                      LDS (one per point) by walking RN(m) for m in N(i); a second walk reads-and-clears
//                      them (atomic AND: the first visitor of a duplicate candidate wins).  Pass 0 counts the
//                      row, pass 1 emits it, bitonic-sorted by column.  Integer / LDS-atomic bound.
//   S4 k_trim_par      the reference's trim loop is sequential and in place (what column i sees depends on
//                      what columns < i deleted); column i only depends on its smaller NEIGHBOURS, so many
//                      columns are in flight, each waiting for those (k_trim = the one-workgroup fallback).
//   S5 k_compact_*     drops the deleted entries.
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/mi_snn.h"
#include <algorithm>
#include <vector>

#include "mi_sa_device.h"

namespace mi_sa_impl {
namespace {

constexpr int kKnnThreads = 64;       // one wavefront per workgroup: 782 workgroups at n = 50 000 (3 per CU)

template <int DP, int KM>
__global__ void __launch_bounds__(kKnnThreads) k_knn(const float *__restrict__ X, int n, int dim, int k,
                                                     int32_t *__restrict__ nn)
{
    __shared__ __attribute__((aligned(16))) float tile[kKnnThreads * DP];
    const int i = blockIdx.x * kKnnThreads + threadIdx.x;
    float xq[DP];
#pragma unroll
    for (int c = 0; c < DP; ++c) xq[c] = (i < n && c < dim) ? X[(size_t)i * dim + c] : 0.0f;
    float bd[KM];
    int bj[KM];
#pragma unroll
    for (int p = 0; p < KM; ++p) { bd[p] = INFINITY; bj[p] = INT_MAX; }

    for (int j0 = 0; j0 < n; j0 += kKnnThreads) {
        __syncthreads();
        {
            const int j = j0 + (int)threadIdx.x;
#pragma unroll
            for (int c = 0; c < DP; ++c)
                tile[threadIdx.x * DP + c] = (j < n && c < dim) ? X[(size_t)j * dim + c] : 0.0f;
        }
        __syncthreads();
        const int lim = n - j0 < kKnnThreads ? n - j0 : kKnnThreads;
        // four candidates at a time: four independent fmaf chains in flight (one chain is 16 dependent
        // instructions), then the (rare) insertions in candidate order.  Rows past `lim` in the tile are
        // zero-filled and masked out by j < n.
        for (int jj = 0; jj < lim; jj += 4) {
            float d4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 *tp = reinterpret_cast<const f32x4 *>(tile + (jj + u) * DP);
                float d = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < DP / 4; ++c4) {
                    const f32x4 v = tp[c4];               // same address in every lane: LDS broadcast
                    float df;
                    df = xq[4 * c4 + 0] - v.x; d = __fmaf_rn(df, df, d);
                    df = xq[4 * c4 + 1] - v.y; d = __fmaf_rn(df, df, d);
                    df = xq[4 * c4 + 2] - v.z; d = __fmaf_rn(df, df, d);
                    df = xq[4 * c4 + 3] - v.w; d = __fmaf_rn(df, df, d);
                }
                d4[u] = d;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + jj + u;
                const float d = d4[u];
                if (j < n && j != i && d < bd[KM - 1]) {  // candidates arrive by ascending j: equal d never displaces
                    bd[KM - 1] = d;
                    bj[KM - 1] = j;
#pragma unroll
                    for (int p = KM - 1; p > 0; --p) {
                        if (bd[p] < bd[p - 1]) {
                            const float td = bd[p]; bd[p] = bd[p - 1]; bd[p - 1] = td;
                            const int tj = bj[p]; bj[p] = bj[p - 1]; bj[p - 1] = tj;
                        }
                    }
                }
            }
        }
    }
    if (i < n) {
        nn[(size_t)i * k] = i;
#pragma unroll
        for (int p = 0; p < KM; ++p)
            if (p < k - 1) nn[(size_t)i * k + 1 + p] = bj[p];
    }
}

__global__ void __launch_bounds__(256) k_rn_count(const int32_t *__restrict__ nn, long long total, int *__restrict__ cnt)
{
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
        atomicAdd(&cnt[nn[e]], 1);
}

// out[0] = 0, out[i+1] = in[0] + .. + in[i]  (single workgroup; n up to millions is a few dozen passes)
__global__ void __launch_bounds__(1024) k_scan_exclusive(const int *__restrict__ in, int *__restrict__ out, int n)
{
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) { carry = 0; out[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int v = i < n ? in[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) out[i + 1] = carry + part[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_rn_fill(const int32_t *__restrict__ nn, int n, int k,
                                                 const int *__restrict__ rn_ptr, int *__restrict__ cursor,
                                                 int32_t *__restrict__ rn_idx)
{
    const long long total = (long long)n * k;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int m = nn[e];
        const int pos = atomicAdd(&cursor[m], 1);
        rn_idx[rn_ptr[m] + pos] = (int32_t)(e / k);
    }
}

constexpr int kRowCap = 4096;          // candidates a row may emit (LDS sort buffer): hubs beyond it are an error

// PASS 0: deg[i] = entries of row i.  PASS 1: col/shared at rowptr[i], ascending by column.
template <int PASS>
__global__ void __launch_bounds__(256) k_snn_rows(const int32_t *__restrict__ nn, int n, int k, double prune,
                                                  const int *__restrict__ rn_ptr, const int32_t *__restrict__ rn_idx,
                                                  int *__restrict__ deg, const int *__restrict__ rowptr,
                                                  int32_t *__restrict__ col, int32_t *__restrict__ shared,
                                                  int *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned int *cnt32 = reinterpret_cast<unsigned int *>(lds);               // one byte per point
    const int words = (n + 3) / 4;
    unsigned long long *rowbuf = reinterpret_cast<unsigned long long *>(lds + (size_t)((words * 4 + 15) / 16) * 16);
    __shared__ int nrow;
    for (int w = threadIdx.x; w < words; w += 256) cnt32[w] = 0u;
    if (threadIdx.x == 0) nrow = 0;
    __syncthreads();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        for (int p = 0; p < k; ++p) {
            const int m = nn[(size_t)i * k + p];
            for (int e = rn_ptr[m] + (int)threadIdx.x; e < rn_ptr[m + 1]; e += 256) {
                const int j = rn_idx[e];
                atomicAdd(&cnt32[j >> 2], 1u << (8 * (j & 3)));
            }
        }
        __syncthreads();
        for (int p = 0; p < k; ++p) {
            const int m = nn[(size_t)i * k + p];
            for (int e = rn_ptr[m] + (int)threadIdx.x; e < rn_ptr[m + 1]; e += 256) {
                const int j = rn_idx[e];
                const int sh = 8 * (j & 3);
                const unsigned int old = atomicAnd(&cnt32[j >> 2], ~(0xffu << sh));
                const int s = (int)((old >> sh) & 0xffu);
                if (s == 0 || j == i) continue;
                if ((double)s / (2.0 * (double)k - (double)s) < prune) continue;
                const int idx = atomicAdd(&nrow, 1);
                if (PASS == 1 && idx < kRowCap) rowbuf[idx] = ((unsigned long long)(unsigned int)j << 32) | (unsigned int)s;
            }
        }
        __syncthreads();
        const int cntrow = nrow;
        if (PASS == 0) {
            if (threadIdx.x == 0) {
                deg[i] = cntrow;
                if (cntrow > kRowCap) atomicExch(err, 1);
            }
        } else if (cntrow <= kRowCap) {
            int m2 = 1;
            while (m2 < cntrow) m2 <<= 1;
            for (int e = cntrow + (int)threadIdx.x; e < m2; e += 256) rowbuf[e] = ~0ull;
            __syncthreads();
            for (int size = 2; size <= m2; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int e = threadIdx.x; e < m2; e += 256) {
                        const int partner = e ^ stride;
                        if (partner > e) {
                            const bool up = (e & size) == 0;
                            const unsigned long long a = rowbuf[e], b = rowbuf[partner];
                            if ((a > b) == up) { rowbuf[e] = b; rowbuf[partner] = a; }
                        }
                    }
                    __syncthreads();
                }
            const int base = rowptr[i];
            for (int e = threadIdx.x; e < cntrow; e += 256) {
                col[base + e] = (int32_t)(rowbuf[e] >> 32);
                shared[base + e] = (int32_t)(rowbuf[e] & 0xffffffffull);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) nrow = 0;
        __syncthreads();
    }
}

// The reference's trim loop (Rmd :75-79) on the CSR form: for i = 0..n-1 in order, keep the `ord` largest
// entries of column i (ties: lower row index first -- R's stable order()) and delete the others together
// with their mirrors.  One workgroup, sequential over i.
constexpr int kTrimThreads = 256;
__global__ void __launch_bounds__(kTrimThreads) k_trim(int n, int ord, const int *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col,
                                                       const int32_t *__restrict__ shared, unsigned char *alive)
{
    __shared__ int key[kRowCap];
    __shared__ int nalive;
    for (int i = 0; i < n; ++i) {
        const int base = rowptr[i], deg = rowptr[i + 1] - base;
        if (deg <= ord) continue;                                   // uniform: nothing can be deleted here
        if (threadIdx.x == 0) nalive = 0;
        __syncthreads();
        int mine = 0;
        for (int e = threadIdx.x; e < deg; e += kTrimThreads) {
            const unsigned char al = __hip_atomic_load(&alive[base + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            key[e] = al ? shared[base + e] : -1;
            mine += al ? 1 : 0;
        }
        if (mine) atomicAdd(&nalive, mine);
        __syncthreads();
        if (nalive > ord) {                                         // uniform
            for (int e = threadIdx.x; e < deg; e += kTrimThreads) {
                const int ke = key[e];
                if (ke < 0) continue;
                int rank = 0;
                for (int f = 0; f < deg; ++f) {
                    const int kf = key[f];
                    rank += (kf > ke || (kf == ke && f < e)) ? 1 : 0;
                }
                if (rank >= ord) {
                    __hip_atomic_store(&alive[base + e], (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int j = col[base + e];
                    int lo = rowptr[j], hi = rowptr[j + 1] - 1;
                    while (lo <= hi) {
                        const int mid = (lo + hi) >> 1;
                        const int cm = col[mid];
                        if (cm == i) {
                            __hip_atomic_store(&alive[mid], (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        if (cm < i) lo = mid + 1; else hi = mid - 1;
                    }
                }
            }
        }
        __threadfence();
        __syncthreads();
    }
}

// The same loop, many columns in flight.  Column i only depends on the columns j < i that are its neighbours
// in the UNTRIMMED graph (step j deletes entries of row/column j only), so workgroups claim columns in index
// order from a counter and each waits -- bounded by a clock AND an iteration cap -- until those neighbours are
// done.  A column is claimed only by a running workgroup and only after every smaller column has been
// claimed, so the smallest unfinished column never waits on anything unfinished: no deadlock.  A wait that
// hits its bound raises ctrl[1] (the host then redoes the trim with the sequential kernel) and proceeds.
//   ctrl[0] = next column to claim, ctrl[1] = error flag;  done[i] = 1 once column i has been processed.
constexpr int kTrimParThreads = 64;
constexpr long long kTrimWaitTicks = 500000;       // 5 ms of the 100 MHz realtime clock
__global__ void __launch_bounds__(kTrimParThreads) k_trim_par(int n, int ord, const int *__restrict__ rowptr,
                                                              const int32_t *__restrict__ col,
                                                              const int32_t *__restrict__ shared, unsigned char *alive,
                                                              unsigned int *ctrl, unsigned int *done)
{
    __shared__ int key[kRowCap];
    __shared__ int nalive, claimed, waits_ok;
    for (int iter = 0; iter <= n; ++iter) {                         // a workgroup can never claim more than n columns
        if (threadIdx.x == 0) { claimed = (int)atomicAdd(&ctrl[0], 1u); nalive = 0; waits_ok = 1; }
        __syncthreads();
        const int i = claimed;
        if (i >= n) return;                                         // uniform
        const int base = rowptr[i], deg = rowptr[i + 1] - base;
        if (deg > ord) {                                            // uniform: only such columns can delete
            // wait for the smaller neighbours (one lane per entry; the column list is ascending)
            for (int e = threadIdx.x; e < deg; e += kTrimParThreads) {
                const int j = col[base + e];
                if (j >= i) break;
                const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
                int spins = 0;
                while (__hip_atomic_load(&done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    if (++spins > 200000 || (long long)__builtin_amdgcn_s_memrealtime() - t0 > kTrimWaitTicks) {
                        waits_ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __threadfence();                                        // acquire side: the flags were read relaxed
            __syncthreads();
            if (!waits_ok && threadIdx.x == 0) __hip_atomic_store(&ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int mine = 0;
            for (int e = threadIdx.x; e < deg; e += kTrimParThreads) {
                const unsigned char al = __hip_atomic_load(&alive[base + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                key[e] = al ? shared[base + e] : -1;
                mine += al ? 1 : 0;
            }
            if (mine) atomicAdd(&nalive, mine);
            __syncthreads();
            if (nalive > ord) {                                     // uniform
                for (int e = threadIdx.x; e < deg; e += kTrimParThreads) {
                    const int ke = key[e];
                    if (ke < 0) continue;
                    int rank = 0;
                    for (int f = 0; f < deg; ++f) {
                        const int kf = key[f];
                        rank += (kf > ke || (kf == ke && f < e)) ? 1 : 0;
                    }
                    if (rank >= ord) {
                        __hip_atomic_store(&alive[base + e], (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int j = col[base + e];
                        int lo = rowptr[j], hi = rowptr[j + 1] - 1;
                        while (lo <= hi) {
                            const int mid = (lo + hi) >> 1;
                            const int cm = col[mid];
                            if (cm == i) {
                                __hip_atomic_store(&alive[mid], (unsigned char)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                            if (cm < i) lo = mid + 1; else hi = mid - 1;
                        }
                    }
                }
            }
            __threadfence();                                        // release side: deletions before the flag
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&done[i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// The optional graph variants of the notebooks (Pbmc3k_general_data_preparation.Rmd:77-123, Kidney_data.Rmd:235-266)
// ------------------------------------------------------------------------------------------------
// The stored rows double as the COLUMNS of the (symmetric) SNN matrix: entry e of row i with col[e] = r is A[r, i].
// "UNSYMMETRIC" trim (Rmd :77-83): column i keeps its `ord` heaviest entries (R's stable order(): heavier first, ties by
// row index) and only column i is written, so the columns are independent: one wavefront per column, rank of an
// entry = entries of the column that precede it in that order.
__global__ void __launch_bounds__(256) k_trim_cols(int n, int ord, const int *__restrict__ rowptr,
                                                   const int32_t *__restrict__ col, const int32_t *__restrict__ key,
                                                   unsigned char *__restrict__ alive)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int b = rowptr[i], e1 = rowptr[i + 1];
    if (e1 - b <= ord) return;
    for (int e = b + lane; e < e1; e += 64) {
        const int ke = key[e];
        int rank = 0;
        for (int f = b; f < e1; ++f) {                          // entries are stored ascending by row index
            const int kf = key[f];
            rank += (kf > ke || (kf == ke && f < e)) ? 1 : 0;
        }
        if (rank >= ord) alive[e] = 0;
    }
}

// position of the mirror entry (row r, column i) of entry (row i, column r), or -1
__device__ __forceinline__ int mirror_of(const int *__restrict__ rowptr, const int32_t *__restrict__ col, int r, int i)
{
    int lo = rowptr[r], hi = rowptr[r + 1] - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int cm = col[mid];
        if (cm == i) return mid;
        if (cm < i) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// "Enhance shared edges".  mode 1 (Rmd :85-101, "Method 2"): an entry that is present in BOTH directions is marked
// (code 1: the caller adds the bonus, + 2 * mutual resp. + mutual in the kidney notebook); the support is unchanged.
// mode 2 (Rmd :103-113, A + t(A)): the support becomes the union of the two directions, an entry present in both is
// marked code 2 (its weight doubles).
__global__ void __launch_bounds__(256) k_enhance(int n, int mode, const int *__restrict__ rowptr,
                                                 const int32_t *__restrict__ col,
                                                 const unsigned char *__restrict__ alive_in,
                                                 unsigned char *__restrict__ alive_out, unsigned char *__restrict__ code)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        const int m = mirror_of(rowptr, col, col[e], i);
        const bool here = alive_in[e] != 0, there = m >= 0 && alive_in[m] != 0;
        if (mode == 1) {
            alive_out[e] = here ? 1 : 0;
            code[e] = (here && there) ? 1 : 0;
        } else {
            alive_out[e] = (here || there) ? 1 : 0;
            code[e] = (here && there) ? 2 : 0;
        }
    }
}

// rank of an entry's enhanced weight among the values it can take (table built on the host in fp64 from s / (2k - s),
// the bonus and the doubling): the trim kernels only compare
__global__ void __launch_bounds__(256) k_make_keys(long long nnz, const int32_t *__restrict__ shared,
                                                   const unsigned char *__restrict__ code,
                                                   const int32_t *__restrict__ table, int32_t *__restrict__ key)
{
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < nnz; e += (long long)gridDim.x * 256)
        key[e] = table[shared[e] * 3 + (code ? code[e] : 0)];
}

// the rounding variant (Pbmc3k_normalization_simulated_data.Rmd:597-606): key = rank of round(w, digits) among the values
// it can take; an entry whose rounded weight the notebook turns into a NEGATIVE edge (table_neg) is dropped when a trim
// follows (R ranks it below every zero of its column: the trim deletes it, see mi_snn.h) and marked code 3 otherwise
__global__ void __launch_bounds__(256) k_round_keys(long long nnz, const int32_t *__restrict__ shared,
                                                    const int32_t *__restrict__ table, int32_t *__restrict__ key,
                                                    unsigned char *__restrict__ alive, unsigned char *__restrict__ code, int drop_negative)
{
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < nnz; e += (long long)gridDim.x * 256) {
        const int32_t t = table[shared[e]];
        key[e] = t & 0xffff;
        const bool neg = (t >> 16) != 0;
        if (neg && drop_negative) alive[e] = 0;
        code[e] = neg ? (unsigned char)3 : (unsigned char)0;
    }
}

__global__ void __launch_bounds__(256) k_compact_count(int n, const int *__restrict__ rowptr,
                                                       const unsigned char *__restrict__ alive, int *__restrict__ deg,
                                                       int *__restrict__ maxdeg)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = 0;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) c += alive[e] ? 1 : 0;
    deg[i] = c;
    atomicMax(maxdeg, c);
}

__global__ void __launch_bounds__(256) k_compact_fill(int n, const int *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const int32_t *__restrict__ shared,
                                                      const unsigned char *__restrict__ alive,
                                                      const int *__restrict__ out_ptr, int32_t *__restrict__ out_col,
                                                      int32_t *__restrict__ out_shared,
                                                      const unsigned char *__restrict__ code,
                                                      unsigned char *__restrict__ out_code)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int o = out_ptr[i];
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
        if (alive[e]) {
            out_col[o] = col[e];
            out_shared[o] = shared[e];
            if (out_code) out_code[o] = code ? code[e] : (unsigned char)0;
            ++o;
        }
}

template <int DP>
int launch_knn(const float *dX, int n, int dim, int k, int32_t *d_nn, hipStream_t st)
{
    const int blocks = (n + kKnnThreads - 1) / kKnnThreads;
    const int kk = k - 1;
    if (kk <= 8) hipLaunchKernelGGL((k_knn<DP, 8>), dim3(blocks), dim3(kKnnThreads), 0, st, dX, n, dim, k, d_nn);
    else if (kk <= 16) hipLaunchKernelGGL((k_knn<DP, 16>), dim3(blocks), dim3(kKnnThreads), 0, st, dX, n, dim, k, d_nn);
    else if (kk <= 32) hipLaunchKernelGGL((k_knn<DP, 32>), dim3(blocks), dim3(kKnnThreads), 0, st, dX, n, dim, k, d_nn);
    else hipLaunchKernelGGL((k_knn<DP, 64>), dim3(blocks), dim3(kKnnThreads), 0, st, dX, n, dim, k, d_nn);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace
}  // namespace mi_sa_impl
using namespace mi_sa_impl;

struct mi_snn_graph {
    int n = 0, dim = 0, k = 0, ord = 0, device = 0, max_degree = 0;
    long long nnz = 0;
    float ms_knn = 0, ms_snn = 0, ms_trim = 0;
    int32_t *d_nn = nullptr;
    int *d_ptr = nullptr;            // final rowptr (n+1)
    int32_t *d_col = nullptr, *d_shared = nullptr;
    unsigned char *d_code = nullptr;  // per entry: 0 plain, 1 present in both directions (mutual bonus), 2 doubled (A + t(A))
    int flags = 0, ord2 = 0;
};

extern "C" {

int mi_snn_destroy(mi_snn_graph *g)
{
    if (!g) return MI_OK;
    (void)hipSetDevice(g->device);
    void *bufs[] = {g->d_nn, g->d_ptr, g->d_col, g->d_shared, g->d_code};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete g;
    return MI_OK;
}

int mi_snn_build_f32(const float *X, int n, int dim, int k, double prune, int ord, int device, mi_snn_graph **out)
{
    return mi_snn_build_ex_f32(X, n, dim, k, prune, ord, 0u, 0.0, 0, device, out);
}

static int snn_build_impl(const float *X, int n, int dim, int k, double prune, int ord, uint32_t flags, double bonus,
                          int ord2, int round_digits, double negative_below, int device, mi_snn_graph **out);

int mi_snn_build_ex_f32(const float *X, int n, int dim, int k, double prune, int ord, uint32_t flags, double bonus,
                        int ord2, int device, mi_snn_graph **out)
{
    return snn_build_impl(X, n, dim, k, prune, ord, flags, bonus, ord2, -1, 0.0, device, out);
}

int mi_snn_build_rounded_f32(const float *X, int n, int dim, int k, double prune, int ord, int round_digits,
                             double negative_below, int device, mi_snn_graph **out)
{
    if (round_digits < 0 || round_digits > 6) return fail(MI_EINVAL, "round_digits must be 0 .. 6 (got %d)", round_digits);
    if (!(negative_below >= 0.0)) return fail(MI_EINVAL, "negative_below must be >= 0 (0 = no negative edges)");
    return snn_build_impl(X, n, dim, k, prune, ord, 0u, 0.0, 0, round_digits, negative_below, device, out);
}

static int snn_build_impl(const float *X, int n, int dim, int k, double prune, int ord, uint32_t flags, double bonus,
                          int ord2, int round_digits, double negative_below, int device, mi_snn_graph **out)
{
    if (!X || !out) return fail(MI_EINVAL, "NULL argument");
    if (flags & ~(uint32_t)(MI_SNN_TRIM_UNSYMMETRIC | MI_SNN_ENHANCE_MUTUAL | MI_SNN_ENHANCE_SUM))
        return fail(MI_EINVAL, "unknown flags 0x%x", flags);
    if ((flags & MI_SNN_ENHANCE_MUTUAL) && (flags & MI_SNN_ENHANCE_SUM))
        return fail(MI_EINVAL, "choose one enhancement: mutual bonus or A + t(A)");
    if ((flags & MI_SNN_TRIM_UNSYMMETRIC) && ord <= 0) return fail(MI_EINVAL, "the unsymmetric trim needs ord > 0");
    if (ord2 > 0 && (flags & MI_SNN_TRIM_UNSYMMETRIC) && !(flags & MI_SNN_ENHANCE_SUM))
        return fail(MI_EUNSUPPORTED, "the second trim is built for symmetric matrices (symmetric first trim, or A + t(A))");
    if (!(bonus >= 0.0)) return fail(MI_EINVAL, "bonus must be >= 0");
    if (n < 2 || dim < 1 || dim > 64) return fail(MI_EINVAL, "need n >= 2 and 1 <= dim <= 64 (got n=%d dim=%d)", n, dim);
    if (k < 2 || k > 64 || k > n) return fail(MI_EINVAL, "need 2 <= k <= min(64, n) (got k=%d)", k);
    if (!(prune >= 0.0)) return fail(MI_EINVAL, "prune must be >= 0");
    if ((size_t)n + kRowCap * 8 + 64 > 160 * 1024)
        return fail(MI_EUNSUPPORTED, "SNN kernel keeps one byte per point in LDS: n <= %d (got %d)", 160 * 1024 - kRowCap * 8 - 64, n);
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(MI_ENODEV, "no HIP device visible");
    if (device < 0 || device >= cnt) return fail(MI_EINVAL, "device %d out of range [0,%d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    mi_snn_graph *g = new (std::nothrow) mi_snn_graph();
    if (!g) return fail(MI_ENOMEM, "out of host memory");
    g->n = n; g->dim = dim; g->k = k; g->ord = ord; g->device = device; g->flags = (int)flags; g->ord2 = ord2;

    float *dX = nullptr;
    int *d_cnt = nullptr, *d_rn_ptr = nullptr, *d_cursor = nullptr, *d_deg = nullptr, *d_ptr0 = nullptr, *d_err = nullptr;
    int32_t *d_rn_idx = nullptr, *d_col0 = nullptr, *d_sh0 = nullptr;
    unsigned char *d_alive = nullptr, *d_alive2 = nullptr, *d_code0 = nullptr;
    int32_t *d_key = nullptr, *d_table = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = nullptr;
    int rc = guarded([&]() -> int {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipMalloc((void **)&dX, (size_t)n * dim * sizeof(float)));
        HIP_TRY(hipMemcpy(dX, X, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&g->d_nn, (size_t)n * k * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&d_cnt, (size_t)(n + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_rn_ptr, (size_t)(n + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_cursor, (size_t)(n + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_rn_idx, (size_t)n * k * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&d_deg, (size_t)(n + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_ptr0, (size_t)(n + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&g->d_ptr, (size_t)(n + 1) * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&d_err, 2 * sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)(n + 1) * sizeof(int), st));
        HIP_TRY(hipMemsetAsync(d_cursor, 0, (size_t)(n + 1) * sizeof(int), st));
        HIP_TRY(hipMemsetAsync(d_err, 0, 2 * sizeof(int), st));

        // S1: exact kNN
        HIP_TRY(hipEventRecord(ev[0], st));
        int r2 = dim <= 16 ? launch_knn<16>(dX, n, dim, k, g->d_nn, st)
                           : (dim <= 32 ? launch_knn<32>(dX, n, dim, k, g->d_nn, st) : launch_knn<64>(dX, n, dim, k, g->d_nn, st));
        if (r2) return r2;
        HIP_TRY(hipEventRecord(ev[1], st));

        // S2: reverse-neighbour lists
        const long long total = (long long)n * k;
        const int gblocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(k_rn_count, dim3(gblocks), dim3(256), 0, st, g->d_nn, total, d_cnt);
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, d_cnt, d_rn_ptr, n);
        hipLaunchKernelGGL(k_rn_fill, dim3(gblocks), dim3(256), 0, st, g->d_nn, n, k, d_rn_ptr, d_cursor, d_rn_idx);
        HIP_TRY(hipGetLastError());

        // S3: shared-neighbour rows (count, scan, emit)
        const size_t lds = (size_t)(((n + 3) / 4 * 4 + 15) / 16) * 16 + (size_t)kRowCap * 8;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_snn_rows<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_snn_rows<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int rblocks = n < cus * 2 ? n : cus * 2;
        hipLaunchKernelGGL(k_snn_rows<0>, dim3(rblocks), dim3(256), lds, st, g->d_nn, n, k, prune, d_rn_ptr, d_rn_idx, d_deg,
                           (const int *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, d_err);
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, d_deg, d_ptr0, n);
        HIP_TRY(hipGetLastError());
        int h_err = 0, nnz0 = 0;
        HIP_TRY(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&nnz0, d_ptr0 + n, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_err) return fail(MI_EUNSUPPORTED, "a point shares neighbours with more than %d others (hub): not supported", kRowCap);
        if (nnz0 < 0) return fail(MI_EUNSUPPORTED, "SNN graph has more than 2^31 entries");
        HIP_TRY(hipMalloc((void **)&d_col0, (size_t)(nnz0 > 0 ? nnz0 : 1) * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&d_sh0, (size_t)(nnz0 > 0 ? nnz0 : 1) * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&d_alive, (size_t)(nnz0 > 0 ? nnz0 : 1)));
        HIP_TRY(hipMemsetAsync(d_alive, 1, (size_t)(nnz0 > 0 ? nnz0 : 1), st));
        hipLaunchKernelGGL(k_snn_rows<1>, dim3(rblocks), dim3(256), lds, st, g->d_nn, n, k, prune, d_rn_ptr, d_rn_idx, d_deg,
                           (const int *)d_ptr0, d_col0, d_sh0, d_err);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[2], st));

        // S4: trim(s) and enhancement, S5: compaction
        // the sequential symmetric trim on `keys` (shared counts, or ranks of the enhanced weights) with degree cap `cap`
        auto trim_symmetric = [&](const int32_t *keys, int cap) -> int {
            if (!getenv("MI_SNN_TRIM_SEQUENTIAL")) {
                // many columns in flight (k_trim_par); the sequential kernel is the fallback should a dependency
                // wait ever hit its bound (MI_SNN_TRIM_SEQUENTIAL=1 forces it)
                unsigned int *d_tctrl = nullptr, *d_done = nullptr;
                unsigned char *d_save = nullptr;
                HIP_TRY(hipMalloc((void **)&d_tctrl, 2 * sizeof(unsigned int)));
                HIP_TRY(hipMalloc((void **)&d_done, (size_t)n * sizeof(unsigned int)));
                HIP_TRY(hipMalloc((void **)&d_save, (size_t)(nnz0 > 0 ? nnz0 : 1)));
                HIP_TRY(hipMemcpyAsync(d_save, d_alive, (size_t)(nnz0 > 0 ? nnz0 : 1), hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemsetAsync(d_tctrl, 0, 2 * sizeof(unsigned int), st));
                HIP_TRY(hipMemsetAsync(d_done, 0, (size_t)n * sizeof(unsigned int), st));
                const int tgrid = cus * 2;
                hipLaunchKernelGGL(k_trim_par, dim3(tgrid), dim3(kTrimParThreads), 0, st, n, cap, (const int *)d_ptr0, d_col0, keys,
                                   d_alive, d_tctrl, d_done);
                unsigned int terr = 0;
                HIP_TRY(hipMemcpyAsync(&terr, d_tctrl + 1, sizeof terr, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                (void)hipFree(d_tctrl);
                (void)hipFree(d_done);
                if (terr) {
                    fprintf(stderr, "mi_snn: parallel trim hit a wait bound; redoing sequentially\n");
                    HIP_TRY(hipMemcpyAsync(d_alive, d_save, (size_t)(nnz0 > 0 ? nnz0 : 1), hipMemcpyDeviceToDevice, st));
                    hipLaunchKernelGGL(k_trim, dim3(1), dim3(kTrimThreads), 0, st, n, cap, (const int *)d_ptr0, d_col0, keys, d_alive);
                    HIP_TRY(hipStreamSynchronize(st));
                }
                (void)hipFree(d_save);
            } else {
                hipLaunchKernelGGL(k_trim, dim3(1), dim3(kTrimThreads), 0, st, n, cap, (const int *)d_ptr0, d_col0, keys, d_alive);
            }
            HIP_TRY(hipGetLastError());
            return MI_OK;
        };
        if (round_digits >= 0) {
            // rounded weights (and the notebook's negative edges): the trim ranks by round(s / (2k - s), digits) -- rounding
            // can make different counts tie, and ties go to the lower row index -- evaluated in fp64 as R holds them
            const double scale = std::pow(10.0, round_digits);
            std::vector<double> wr((size_t)k + 1, 0.0);
            for (int sct = 1; sct <= k; ++sct) wr[(size_t)sct] = std::nearbyint((double)sct / (2.0 * k - (double)sct) * scale) / scale;
            std::vector<double> uniq(wr);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            std::vector<int32_t> table((size_t)k + 1);
            bool any_neg = false;
            for (int sct = 0; sct <= k; ++sct) {
                const bool neg = negative_below > 0.0 && wr[(size_t)sct] < negative_below && wr[(size_t)sct] != 0.0;
                any_neg = any_neg || (neg && sct > 0);
                table[(size_t)sct] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), wr[(size_t)sct]) - uniq.begin()) | (neg ? 0x10000 : 0);
            }
            if (any_neg && ord > 0) {
                // R sorts a column as [positives, zeros, negatives]: a negative entry survives the trim of its column only
                // if the column has fewer than ord non-negative ENTRIES, zeros and the diagonal included -- i.e. n - (its
                // negatives) < ord.  With every column at least ord non-negative positions long all of them are deleted,
                // which is what dropping them before the trim computes; a graph that small is refused.
                std::vector<int> hdeg((size_t)n);
                HIP_TRY(hipMemcpyAsync(hdeg.data(), d_deg, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                int maxdeg0 = 0;
                for (int i = 0; i < n; ++i) maxdeg0 = std::max(maxdeg0, hdeg[(size_t)i]);
                if (n - maxdeg0 < ord)
                    return fail(MI_EUNSUPPORTED, "negative edges on a graph this small (n = %d, densest column %d entries, ord = %d) "
                                "could survive the trim; not supported", n, maxdeg0, ord);
            }
            HIP_TRY(hipMalloc((void **)&d_table, table.size() * sizeof(int32_t)));
            HIP_TRY(hipMalloc((void **)&d_key, (size_t)(nnz0 > 0 ? nnz0 : 1) * sizeof(int32_t)));
            HIP_TRY(hipMalloc((void **)&d_code0, (size_t)(nnz0 > 0 ? nnz0 : 1)));
            HIP_TRY(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_round_keys, dim3(1024), dim3(256), 0, st, (long long)nnz0, d_sh0, d_table, d_key, d_alive, d_code0,
                               ord > 0 ? 1 : 0);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));                   // (table goes out of scope)
            if (ord > 0) {
                const int r3 = trim_symmetric(d_key, ord);
                if (r3) return r3;
            }
        } else if (ord > 0 && (flags & MI_SNN_TRIM_UNSYMMETRIC)) {
            hipLaunchKernelGGL(k_trim_cols, dim3((n + 3) / 4), dim3(256), 0, st, n, ord, (const int *)d_ptr0, d_col0, d_sh0, d_alive);
            HIP_TRY(hipGetLastError());
        } else if (ord > 0) {
            const int r3 = trim_symmetric(d_sh0, ord);
            if (r3) return r3;
        }
        if (flags & (MI_SNN_ENHANCE_MUTUAL | MI_SNN_ENHANCE_SUM)) {
            HIP_TRY(hipMalloc((void **)&d_alive2, (size_t)(nnz0 > 0 ? nnz0 : 1)));
            HIP_TRY(hipMalloc((void **)&d_code0, (size_t)(nnz0 > 0 ? nnz0 : 1)));
            hipLaunchKernelGGL(k_enhance, dim3((n + 255) / 256), dim3(256), 0, st, n, (flags & MI_SNN_ENHANCE_MUTUAL) ? 1 : 2,
                               (const int *)d_ptr0, d_col0, d_alive, d_alive2, d_code0);
            HIP_TRY(hipGetLastError());
            std::swap(d_alive, d_alive2);
        }
        if (ord2 > 0) {
            // ranks of the values an entry can hold, in fp64 as R holds them: w = s / (2k - s), w + bonus, w + w
            std::vector<double> vals;
            for (int sct = 0; sct <= k; ++sct) {
                const double w = sct == 0 ? 0.0 : (double)sct / (2.0 * k - (double)sct);
                vals.push_back(w); vals.push_back(w + bonus); vals.push_back(w + w);
            }
            std::vector<double> uniq(vals);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            std::vector<int32_t> table(vals.size());
            for (size_t q = 0; q < vals.size(); ++q)
                table[q] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), vals[q]) - uniq.begin());
            HIP_TRY(hipMalloc((void **)&d_table, table.size() * sizeof(int32_t)));
            HIP_TRY(hipMalloc((void **)&d_key, (size_t)(nnz0 > 0 ? nnz0 : 1) * sizeof(int32_t)));
            HIP_TRY(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_make_keys, dim3(1024), dim3(256), 0, st, (long long)nnz0, d_sh0, d_code0, d_table, d_key);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));                   // (table goes out of scope)
            const int r3 = trim_symmetric(d_key, ord2);
            if (r3) return r3;
        }
        HIP_TRY(hipMemsetAsync(d_err + 1, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_compact_count, dim3((n + 255) / 256), dim3(256), 0, st, n, (const int *)d_ptr0, d_alive, d_deg, d_err + 1);
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, d_deg, g->d_ptr, n);
        HIP_TRY(hipGetLastError());
        int nnz1 = 0;
        HIP_TRY(hipMemcpyAsync(&nnz1, g->d_ptr + n, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&g->max_degree, d_err + 1, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        g->nnz = nnz1;
        HIP_TRY(hipMalloc((void **)&g->d_col, (size_t)(nnz1 > 0 ? nnz1 : 1) * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&g->d_shared, (size_t)(nnz1 > 0 ? nnz1 : 1) * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&g->d_code, (size_t)(nnz1 > 0 ? nnz1 : 1)));
        hipLaunchKernelGGL(k_compact_fill, dim3((n + 255) / 256), dim3(256), 0, st, n, (const int *)d_ptr0, d_col0, d_sh0, d_alive,
                           (const int *)g->d_ptr, g->d_col, g->d_shared, (const unsigned char *)d_code0, g->d_code);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[3], st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipEventElapsedTime(&g->ms_knn, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&g->ms_snn, ev[1], ev[2]));
        HIP_TRY(hipEventElapsedTime(&g->ms_trim, ev[2], ev[3]));
        return MI_OK;
    });
    void *tmp[] = {dX, d_cnt, d_rn_ptr, d_cursor, d_rn_idx, d_deg, d_ptr0, d_err, d_col0, d_sh0, d_alive, d_alive2, d_code0, d_key, d_table};
    for (void *b : tmp)
        if (b) (void)hipFree(b);
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    if (rc) { mi_snn_destroy(g); return rc; }
    *out = g;
    return MI_OK;
}

int mi_snn_info(const mi_snn_graph *g, int *n, int *k, int64_t *nnz, int *max_degree)
{
    if (!g) return fail(MI_EINVAL, "NULL graph");
    if (n) *n = g->n;
    if (k) *k = g->k;
    if (nnz) *nnz = g->nnz;
    if (max_degree) *max_degree = g->max_degree;
    return MI_OK;
}

int mi_snn_fetch(mi_snn_graph *g, int32_t *nn, int64_t *rowptr, int32_t *col, int32_t *shared)
{
    if (!g) return fail(MI_EINVAL, "NULL graph");
    HIP_TRY(hipSetDevice(g->device));
    if (nn) HIP_TRY(hipMemcpy(nn, g->d_nn, (size_t)g->n * g->k * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (rowptr) {
        const int rc = guarded([&]() -> int {
            std::vector<int> tmp((size_t)g->n + 1);
            HIP_TRY(hipMemcpy(tmp.data(), g->d_ptr, tmp.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < tmp.size(); ++i) rowptr[i] = tmp[i];
            return MI_OK;
        });
        if (rc) return rc;
    }
    if (col && g->nnz) HIP_TRY(hipMemcpy(col, g->d_col, (size_t)g->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (shared && g->nnz) HIP_TRY(hipMemcpy(shared, g->d_shared, (size_t)g->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_snn_fetch_codes(mi_snn_graph *g, uint8_t *code)
{
    if (!g || !code) return fail(MI_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(g->device));
    if (g->nnz) HIP_TRY(hipMemcpy(code, g->d_code, (size_t)g->nnz, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_snn_kernel_ms(const mi_snn_graph *g, float *knn_ms, float *snn_ms, float *trim_ms)
{
    if (!g) return fail(MI_EINVAL, "NULL graph");
    if (knn_ms) *knn_ms = g->ms_knn;
    if (snn_ms) *snn_ms = g->ms_snn;
    if (trim_ms) *trim_ms = g->ms_trim;
    return MI_OK;
}

}  // extern "C"
