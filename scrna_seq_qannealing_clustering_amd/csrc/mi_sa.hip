// mi_sa.hip -- MI355X (gfx950) simulated-annealing engine: kernels + C ABI (include/mi_sa.h).
//
// Chain specification (shared with the CPU oracle by DESIGN.md, not by code):
//   * dense binary model  E(x) = x^T Qs x + offset.  Device matrix Q2 = 2*Qs off-diagonal, 0 on the
//     diagonal; diag = Qs_ii.  Cached local field f_i = diag_i + sum_j Q2_ij x_j.
//   * proposal (variable i, sweep s, global replica g): accepted iff
//         (x_i ? -f_i : f_i)  <  neglog_u(philox(i, s, g, 0)) * T_s ,   T_s = (float)(1/beta_s)
//     variables visited in index order 0..n-1; an accepted flip adds +-Q2 row i to f.
//   * one 64-lane wavefront owns one replica.  Variable i lives on lane (i & 63), slot t = i >> 6;
//     the field of slot t is VGPR f[t] (fully unrolled, NT slots).  Within a slot all 64 lanes test
//     their proposal at once; the LOWEST accepting lane is committed, its Q2 row is streamed
//     (16 B/lane coalesced loads from the slot-permuted matrix) into f, and only lanes above it are
//     re-tested -- exactly the sequential sweep order of the oracle, with rejected proposals free.
//
// Reference call sites served: BQM_clustering.py:57,75,85,245,263,273,386 ; DQM_clustering.py:45.

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mi_sa.h"

#include "mi_sa_device.h"

namespace mi_sa_impl {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;

thread_local std::string g_kernel;

void note_kernel(const char *fmt, ...)
{
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g_kernel.find(buf) == std::string::npos) g_kernel += (g_kernel.empty() ? "" : " + ") + std::string(buf);
}

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}



// ------------------------------------------------------------------------------------------------
// K5: best-of-replicas: packed (sortable(float E) << 32 | global id) minimum
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sortable_f32(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// One workgroup: the replica with the lowest fp64 energy (ties: the lowest index) -- exactly the record a sorted
// SampleSet puts first.  The packed key it writes for the cross-GPU exchange carries float(E): between GPUs the
// comparison has fp32 resolution (1 part in 1.7e7 of |E|), inside one GPU it is exact.
__global__ void __launch_bounds__(1024) k_best(const double *__restrict__ energy, int R,
                                               uint32_t replica_offset,
                                               unsigned long long *__restrict__ out_key)
{
    __shared__ double s_e[16];
    __shared__ int s_i[16];
    double be = INFINITY;
    int bi = 0x7fffffff;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const double e = energy[r];
        if (e < be || (e == be && r < bi)) { be = e; bi = r; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oe = __shfl_xor(be, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (oe < be || (oe == be && oi < bi)) { be = oe; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = be; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (s_e[w] < be || (s_e[w] == be && s_i[w] < bi)) { be = s_e[w]; bi = s_i[w]; }
        if (bi == 0x7fffffff) bi = 0;                       // every energy NaN: report replica 0
        out_key[0] = ((unsigned long long)sortable_f32((float)energy[bi]) << 32) |
                     (unsigned long long)(replica_offset + (uint32_t)bi);
    }
}

// ------------------------------------------------------------------------------------------------
// K6: replica exchange of parallel tempering, on the device
// ------------------------------------------------------------------------------------------------
// R = chains x T replicas, global replica g belongs to chain g / T and holds ladder rung rung[g].  One workgroup per
// chain.  Round `rnd` proposes the disjoint neighbour pairs (k, k+1), k = (rnd & 1), +2, ...: with a, b the replicas
// holding rungs k and k+1,
//     arg = (beta_k - beta_{k+1}) (E_a - E_b)                       (fp64)
//     exchange iff  arg >= 0  or  -arg < neglog_u(word(i = chain T + k, s = rnd, g = 0xffffffff, tag 3))
// -- Metropolis, min(1, e^arg), with the chain's own bit-reproducible logarithm and Philox stream, so every GPU
// (and the oracle, oracle/pt_oracle.py) takes the same decisions from the same energies.  Temperatures move, states
// never do: the kernel swaps the two rung indices and rewrites temps[] of the replicas this GPU owns.
__global__ void __launch_bounds__(256) k_pt_exchange(const double *__restrict__ energy, int *__restrict__ rung,
                                                     const double *__restrict__ betas,
                                                     const float *__restrict__ ladder_temps,
                                                     float *__restrict__ temps_local, int T, int lo, int hi,
                                                     uint32_t rnd, uint32_t seed_lo, uint32_t seed_hi,
                                                     unsigned long long *__restrict__ stats)
{
    extern __shared__ int holder[];                          // holder[k] = replica of this chain on rung k
    const int c = blockIdx.x;
    for (int t = threadIdx.x; t < T; t += blockDim.x) holder[rung[c * T + t]] = c * T + t;
    __syncthreads();
    unsigned int proposed = 0, accepted = 0;
    for (int k = (int)(rnd & 1u) + 2 * (int)threadIdx.x; k + 1 < T; k += 2 * (int)blockDim.x) {
        const int a = holder[k], b = holder[k + 1];
        const double arg = (betas[k] - betas[k + 1]) * (energy[a] - energy[b]);
        ++proposed;
        const bool acc = arg >= 0.0 ||
                         -arg < (double)neglog_u(chain_word_dev((uint32_t)(c * T + k), rnd, 0xffffffffu, 3u, seed_lo, seed_hi));
        if (acc) {
            rung[a] = k + 1;
            rung[b] = k;
            ++accepted;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int g = c * T + t;
        if (g >= lo && g < hi) temps_local[g - lo] = ladder_temps[rung[g]];
    }
    if (proposed) { atomicAdd(&stats[0], (unsigned long long)proposed); atomicAdd(&stats[1], (unsigned long long)accepted); }
}

}  // namespace mi_sa_impl
using namespace mi_sa_impl;

// ================================================================================================
// host side
// ================================================================================================
struct mi_sa_problem {
    int kind = 0, n = 0, K = 0, device = 0;
    double offset = 0.0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // dense
    int NT = 0;
    float *d_Qp = nullptr;
    float *d_Qm = nullptr;       // K1m: plain row-major Q2 + diagonal row (NT <= 44)
    float *d_Qs = nullptr;       // plain row-major copy (energy kernel), allocated lazily
    float *d_Q2xl = nullptr, *d_diagxl = nullptr;   // K1x (n > 4096): padded rows of 2*Qs, diagonal
    void *d_xg = nullptr;                    // K1g workspace (fields of all replicas, state words, signs, thresholds)
    size_t xg_bytes = 0;
    int opt_xl_batched = 0;                  // n > 4096: 0 auto (K1g for >= 256 replicas or n >= 16384), 1 always K1g, 2 always K1x
    int opt_xl_chain = 0;                    // K1g, chain of a group of blocks: 0 auto (fused up to 512 replicas), 1 a DIAG and a small pass per block, 2 fused
    int opt_xl_chunk = 8;                    // K1g: sweeps per chunk of a cooling run (the hand-over to K1x is decided per chunk)
    int opt_xl_cold_permille = 20;           // hand the rest of the run to K1x when a chunk accepted less than this share (0 = never)
    int opt_xl_async = 1;                    // that cooling run is driven by a worker thread: mi_sa_anneal returns at once (0 = in the caller)
    std::thread worker;                      // joined by the next call on this problem (settle)
    int worker_rc = MI_OK;
    std::string worker_err, worker_kernel;
    int xl_chunks = 0;
    // structured kinds (slot-ELL)
    int slots = 0, D = 0;
    float c_pair = 0.0f;
    uint32_t *d_ell_col = nullptr;
    float *d_ell_val = nullptr, *d_lin = nullptr;
    double *d_ell_val64 = nullptr, *d_lin64 = nullptr;   // optional fp64 energy model (mi_sa_problem_set_energy_model_f64)
    double c_pair64 = 0.0;
    std::vector<int32_t> h_rowptr;           // structured kinds: the CSR row pointers given at creation
    uint2 *d_rows = nullptr;                 // K2: row-major adjacency, in-slot neighbours first
    uint32_t *d_meta = nullptr;              // K2 / K3: in-slot count | degree << 8 | absent << 31
    std::vector<uint32_t> h_meta;            // host copy (mi_sa_problem_set_absent)
    std::vector<uint8_t> h_hole;             // structured binary: positions whose linear term is +inf (mi_sa_problem_set_pair_weights)
    uint4 *d_adj4 = nullptr;                 // K2: packed slot adjacency (see EllArgs::adj4)
    uint4 *d_adj4p = nullptr;                // K2p (two replicas per wavefront): the same with neighbour word = 4 * index; null = not eligible
    uint32_t *d_slot_flags = nullptr;        // K2: slots with internal edges
    int k2_state_bytes = 0;                  // K2: byte-per-variable state (16 replicas x n bytes fit one CU's LDS)
    int32_t *d_wgt = nullptr;                // K2 family: the 64 pair-term weights of the weighted slot (mi_sa_problem_set_pair_weights)
    int wslot = -1;                          // ... its index; -1: every weight is 1
    int k2_free_block = 0;                   // K2s: widest block of seats (256 / 128 / 64; 0 = none) that holds no edge anywhere in the model
    int cus = 0;
    // run buffers
    int cap_R = 0, cap_sweeps = 0;
    int last_R = 0;
    uint32_t last_offset = 0;
    bool has_run = false;
    float *d_temps = nullptr;
    void *d_init = nullptr;
    void *d_states = nullptr;
    double *d_energy = nullptr;
    unsigned long long *d_stats = nullptr;   // 4 words: proposals, accepted, bytes, best-key
    unsigned int *d_pace = nullptr;          // kPaceWords per launch chunk
    int opt_pace = 1;                        // sweep pacing on/off (speed only)
    int opt_variant = 0;                     // 0 auto, 1 wave-per-replica (K1), 2 workgroup/LDS ring (K1w), 3 MFMA (K1m)
    int opt_mfma_permille = 600;             // chunks that accept >= this share of their proposals hand the next one to K1m (0 = never)

    int opt_chunk_sweeps = 32;               // K1w/K1m: sweeps per launch of a chunked run (0 = one launch)
    float *d_fields = nullptr;               // cached fields between the launches of a chunked run
    unsigned int *d_ctrl = nullptr;          // kernel-scheduling words
    int cap_fields_R = 0;
    int opt_ondemand_permille = 40;          // K1w: on-demand sweeps below this acceptance (per mille); 0 = always stream
    int opt_debug = 0;                       // DenseArgs::debug (diagnostic timing only; results are wrong)
    int opt_min_cluster_size = 0;            // K3: hard lower bound on every cluster's size (CQM_clustering.py:46-48)
    int opt_k2_waves = 0;                    // K2: replicas per workgroup (0 = auto)
    int opt_k2_pair = 0;                     // K2p: 0 auto (runs of more replicas than the chip has SIMDs), 1 always when eligible, 2 never
    int opt_k2_split = 0;                    // K2s (csrc/sparse_split_kernels.hip): 0 auto (few replicas: its one-wavefront form), 1 always when eligible (2 / 4 wavefronts per replica on models laid out in blocks of 128 / 256 seats), 2 never
    int opt_k2_wide = 0;                     // models laid out in blocks of 128 / 256 seats, few replicas: 0 / 1 one wavefront sweeps a block per step (K2w), 2 a workgroup of 2 / 4 wavefronts does (K2s)
    int opt_k3_fast = 0;                     // K3f (csrc/potts_fast_kernels.hip): 0 auto (when the model is eligible), 2 never
    int opt_k2_tw = 0;                       // K2p with a threshold wavefront per workgroup: 0 auto (when built for the width), 1 on, 2 off
    int opt_k2_split_max = 1024;             // ... auto: runs of up to this many replicas (a wavefront per SIMD at most)
    int opt_unit_rows = 0;                   // K1w ring unit (rows per rendezvous): 0 auto, 2 or 4
    int resident_waves = 0;                  // co-resident wavefronts of the anneal kernel on this device
    int last_launches = 1;                   // kernel launches that served the last anneal
    std::string last_kernel;                 // ... and the kernel(s) they ran
    size_t state_elem = 1;
    // parallel tempering (mi_sa_tempering_*): ladder, rung of every replica of the run, exchange statistics
    int pt_T = 0, pt_chains = 0, pt_lo = 0, pt_R_local = 0;
    int *d_pt_rung = nullptr;
    double *d_pt_betas = nullptr, *d_pt_energy = nullptr;
    float *d_pt_ladder = nullptr;
    float *d_pt_temps = nullptr;             // per-replica temperatures of the next tempering round (its own buffer: an ordinary anneal on the same handle rewrites d_temps)
    unsigned long long *d_pt_stats = nullptr;
};

namespace {

int select_device(int device)
{
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(MI_ENODEV, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= cnt) return fail(MI_EINVAL, "device %d out of range [0,%d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    return MI_OK;
}

int ensure_run_buffers(mi_sa_problem *p, int R, int num_sweeps, bool need_init)
{
    if (R > p->cap_R) {
        if (p->d_states) (void)hipFree(p->d_states);
        if (p->d_energy) (void)hipFree(p->d_energy);
        if (p->d_init) { (void)hipFree(p->d_init); p->d_init = nullptr; }
        p->d_states = nullptr; p->d_energy = nullptr;
        HIP_TRY(hipMalloc(&p->d_states, (size_t)R * p->n * p->state_elem));
        HIP_TRY(hipMalloc((void **)&p->d_energy, (size_t)R * sizeof(double)));
        p->cap_R = R;
    }
    if (need_init && !p->d_init) HIP_TRY(hipMalloc(&p->d_init, (size_t)p->cap_R * p->n * p->state_elem));
    if (num_sweeps > p->cap_sweeps || !p->d_temps) {
        if (p->d_temps) (void)hipFree(p->d_temps);
        p->d_temps = nullptr;
        HIP_TRY(hipMalloc((void **)&p->d_temps, (size_t)(num_sweeps > 0 ? num_sweeps : 1) * sizeof(float)));
        p->cap_sweeps = num_sweeps;
    }
    return MI_OK;
}

int dispatch_dense(mi_sa_problem *p, const DenseArgs &a, hipStream_t st)
{
    if (p->opt_chunk_sweeps > 0 && a.num_sweeps > p->opt_chunk_sweeps && p->opt_variant != 1 && a.R >= 32 &&
        (p->cap_fields_R < a.R || !p->d_fields)) {
        if (p->d_fields) (void)hipFree(p->d_fields);
        p->d_fields = nullptr;
        HIP_TRY(hipMalloc((void **)&p->d_fields, (size_t)a.R * p->NT * 64 * sizeof(float)));
        p->cap_fields_R = a.R;
    }
    if (!p->d_ctrl) {
        HIP_TRY(hipMalloc((void **)&p->d_ctrl, kCtrlWords * sizeof(unsigned int)));
        HIP_TRY(hipMemset(p->d_ctrl, 0, kCtrlWords * sizeof(unsigned int)));
    }
    DenseLaunchCtx ctx{p->device, p->opt_pace, p->opt_variant, p->opt_unit_rows, p->opt_ondemand_permille,
                       p->opt_chunk_sweeps, p->opt_mfma_permille, p->d_fields, p->d_ctrl, p->d_pace, &p->resident_waves,
                       &p->last_launches};
    p->last_launches = 1;
    switch (p->NT) {
#define MI_CASE(N) case N: return mi_launch_dense_nt##N(ctx, a, st);
        MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16) MI_CASE(20) MI_CASE(24) MI_CASE(28)
        MI_CASE(32) MI_CASE(36) MI_CASE(40) MI_CASE(44) MI_CASE(48) MI_CASE(52) MI_CASE(56)
        MI_CASE(60) MI_CASE(64)
#undef MI_CASE
    }
    return fail(MI_EUNSUPPORTED, "dense kernel not built for NT=%d", p->NT);
}

constexpr int kMaxDenseN = 64 * 64;        // register-per-wave kernels (K1, K1w, K1m)
constexpr int kMaxDenseXlN = 16 * 4096;    // workgroup-per-replica kernel (K1x)

}  // namespace

// A cooling run on the batched large-model kernels decides its hand-over per chunk on the host (anneal_ex_impl), in a worker
// thread of the problem.  Every entry point that takes the problem joins that thread first; its error becomes the error of
// the joining call.
static int settle(mi_sa_problem *p)
{
    if (!p || !p->worker.joinable()) return MI_OK;
    p->worker.join();
    p->last_kernel = p->worker_kernel;
    if (p->worker_rc) {
        const int rc = p->worker_rc;
        p->worker_rc = MI_OK;
        return fail(rc, "%s", p->worker_err.c_str());
    }
    return MI_OK;
}

extern "C" {

const char *mi_last_error(void) { return g_err.c_str(); }

int mi_abi_version(void) { return 1; }

int mi_device_count(int *out_count)
{
    if (!out_count) return fail(MI_EINVAL, "out_count is NULL");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) { *out_count = 0; return fail(MI_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *out_count = cnt;
    return MI_OK;
}

int mi_device_info(int device, char *name, int len, int *out_cus, uint64_t *out_hbm_bytes)
{
    int rc = select_device(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && len > 0) snprintf(name, (size_t)len, "%s (%s)", prop.name, prop.gcnArchName);
    if (out_cus) *out_cus = prop.multiProcessorCount;
    if (out_hbm_bytes) *out_hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return MI_OK;
}

static int problem_common_init(mi_sa_problem *p, int device)
{
    p->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&p->ev0));
    HIP_TRY(hipEventCreate(&p->ev1));
    HIP_TRY(hipMalloc((void **)&p->d_stats, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(p->d_stats, 0, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void **)&p->d_pace, kMaxChunks * kPaceWords * sizeof(unsigned int)));
    HIP_TRY(hipDeviceGetAttribute(&p->cus, hipDeviceAttributeMultiprocessorCount, device));
    return MI_OK;
}

int mi_sa_problem_create_dense_f32(const float *Qs, int n, double offset, int device,
                                   mi_sa_problem **out)
{
    if (!Qs || !out) return fail(MI_EINVAL, "NULL argument");
    if (n < 1) return fail(MI_EINVAL, "n must be >= 1 (got %d)", n);
    if (n > kMaxDenseXlN)
        return fail(MI_EUNSUPPORTED, "dense kernels support n <= %d (got %d)", kMaxDenseXlN, n);
    int rc = select_device(device);
    if (rc) return rc;
    mi_sa_problem *p = new (std::nothrow) mi_sa_problem();
    if (!p) return fail(MI_ENOMEM, "out of host memory");
    p->kind = MI_KIND_DENSE; p->n = n; p->offset = offset; p->state_elem = 1;
    rc = problem_common_init(p, device);
    if (rc) { mi_sa_problem_destroy(p); return rc; }
    if (n > kMaxDenseN) {
        // K1x: Q stays in HBM as n padded rows of 2*Qs (zero diagonal), uploaded in blocks of rows
        p->xl_chunks = (n + 4095) / 4096;
        const size_t xstride = (size_t)p->xl_chunks * 4096;
        rc = guarded([&]() -> int {
            HIP_TRY(hipMalloc((void **)&p->d_Q2xl, (size_t)n * xstride * sizeof(float)));
            HIP_TRY(hipMalloc((void **)&p->d_diagxl, xstride * sizeof(float)));
            const int rows_per_block = 256;
            std::vector<float> blk((size_t)rows_per_block * xstride), hd(xstride, 0.0f);
            for (int r0 = 0; r0 < n; r0 += rows_per_block) {
                const int nr = n - r0 < rows_per_block ? n - r0 : rows_per_block;
                std::fill(blk.begin(), blk.begin() + (size_t)nr * xstride, 0.0f);
                for (int i = 0; i < nr; ++i) {
                    const float *row = Qs + (size_t)(r0 + i) * n;
                    float *dst = blk.data() + (size_t)i * xstride;
                    for (int j = 0; j < n; ++j) dst[j] = row[j] + row[j];
                    dst[r0 + i] = 0.0f;
                    hd[r0 + i] = row[r0 + i];
                }
                HIP_TRY(hipMemcpy(p->d_Q2xl + (size_t)r0 * xstride, blk.data(), (size_t)nr * xstride * sizeof(float), hipMemcpyHostToDevice));
            }
            HIP_TRY(hipMemcpy(p->d_diagxl, hd.data(), xstride * sizeof(float), hipMemcpyHostToDevice));
            return MI_OK;
        });
        if (rc) { mi_sa_problem_destroy(p); return rc; }
        *out = p;
        return MI_OK;
    }
    rc = guarded([&]() -> int {
        const int slots = (n + 63) / 64;
        p->NT = ((slots + 3) / 4) * 4;
        const size_t stride = (size_t)p->NT * 64;
        // host-side permute: Qp[i][(g*64+lane)*4+c] = 2*Qs[i][64*(4g+c)+lane] (0 on diagonal / padding)
        const int diag_row = slots * 64;
        std::vector<float> hp((size_t)(diag_row + 1) * stride, 0.0f);
        for (int i = 0; i < n; ++i) {
            const float *row = Qs + (size_t)i * n;
            float *dst = hp.data() + (size_t)i * stride;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const int t = j >> 6, lane = j & 63;
                dst[((size_t)(t >> 2) * 64 + lane) * 4 + (t & 3)] = row[j] + row[j];
            }
            hp[(size_t)diag_row * stride + ((size_t)((i >> 6) >> 2) * 64 + (i & 63)) * 4 + ((i >> 6) & 3)] = row[i];
        }
        HIP_TRY(hipMalloc((void **)&p->d_Qp, hp.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(p->d_Qp, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
        if (p->NT <= kMaxMfmaNT) {
            // K1m layout: plain row-major Q2 (zero diagonal), NPAD = 64*NT columns, NPAD rows + the diagonal row
            const size_t npad = (size_t)p->NT * 64;
            std::vector<float> hm((npad + 1) * npad, 0.0f);
            for (int i = 0; i < n; ++i) {
                const float *row = Qs + (size_t)i * n;
                float *dst = hm.data() + (size_t)i * npad;
                for (int j = 0; j < n; ++j) dst[j] = (j == i) ? 0.0f : row[j] + row[j];
                hm[npad * npad + i] = row[i];
            }
            HIP_TRY(hipMalloc((void **)&p->d_Qm, hm.size() * sizeof(float)));
            HIP_TRY(hipMemcpy(p->d_Qm, hm.data(), hm.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        return MI_OK;
    });
    if (rc) { mi_sa_problem_destroy(p); return rc; }
    *out = p;
    return MI_OK;
}

// CSR (both directions stored) -> slot-ELL device arrays (D = 16 / 32 / 64)
static int upload_slot_ell(mi_sa_problem *p, const int32_t *rowptr, const int32_t *col, const float *val, int n)
{
    int maxdeg = 0;
    for (int i = 0; i < n; ++i) {
        const int d = rowptr[i + 1] - rowptr[i];
        if (d < 0) return fail(MI_EINVAL, "rowptr is not monotone at %d", i);
        if (d > maxdeg) maxdeg = d;
    }
    // rows up to 64 wide are register resident in K2 and K3; wider ones (any multiple of 16 up to 4096) run on the
    // runtime-width forms of the same kernels, which read the adjacency from L2 inside the field sum
    if (maxdeg > 4096)
        return fail(MI_EUNSUPPORTED, "max degree %d exceeds the widest adjacency layout (4096); use the dense kernel", maxdeg);
    const int D = maxdeg <= 16 ? 16 : (maxdeg <= 32 ? 32 : (maxdeg <= 64 ? 64 : ((maxdeg + 15) / 16) * 16));
    const int slots = (n + 63) / 64;
    std::vector<uint32_t> hc((size_t)slots * D * 64);
    std::vector<float> hv((size_t)slots * D * 64, 0.0f);
    for (int t = 0; t < slots; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = t * 64 + lane;
            for (int k = 0; k < D; ++k) hc[((size_t)t * D + k) * 64 + lane] = (uint32_t)(i < n ? i : 0);
            if (i >= n) continue;
            for (int e = rowptr[i], k = 0; e < rowptr[i + 1]; ++e, ++k) {
                if (col[e] < 0 || col[e] >= n || col[e] == i)
                    return fail(MI_EINVAL, "bad column %d in row %d", col[e], i);
                hc[((size_t)t * D + k) * 64 + lane] = (uint32_t)col[e];
                hv[((size_t)t * D + k) * 64 + lane] = val[e];
            }
        }
    HIP_TRY(hipMalloc((void **)&p->d_ell_col, hc.size() * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&p->d_ell_val, hv.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(p->d_ell_col, hc.data(), hc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_ell_val, hv.data(), hv.size() * sizeof(float), hipMemcpyHostToDevice));
    p->slots = slots;
    p->D = D;
    p->h_rowptr.assign(rowptr, rowptr + n + 1);
    {
        // row-major copy (K2, K3): neighbours in the variable's own 64-slot first
        std::vector<uint2> hr((size_t)slots * 64 * D);
        std::vector<uint32_t> hm((size_t)slots * 64, 0u);
        for (int i = 0; i < slots * 64; ++i) {
            uint2 *row = hr.data() + (size_t)i * D;
            for (int k = 0; k < D; ++k) row[k] = make_uint2((uint32_t)(i < n ? i : 0), 0u);   // (self, +0.0f)
            if (i >= n) { hm[i] = 0x80000000u; continue; }    // bit 31: no variable at this position (K3 reads it)
            int k = 0, nin = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
                    const bool in_slot = (col[e] >> 6) == (i >> 6);
                    if (in_slot != (pass == 0)) continue;
                    uint32_t bits;
                    memcpy(&bits, &val[e], 4);
                    row[k++] = make_uint2((uint32_t)col[e], bits);
                    nin += in_slot ? 1 : 0;
                }
            hm[i] = (uint32_t)nin | ((uint32_t)k << 8);
        }
        HIP_TRY(hipMalloc((void **)&p->d_rows, hr.size() * sizeof(uint2)));
        HIP_TRY(hipMalloc((void **)&p->d_meta, hm.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(p->d_rows, hr.data(), hr.size() * sizeof(uint2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_meta, hm.data(), hm.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        p->h_meta = hm;
        if (p->kind == MI_KIND_POTTS_CSR && (D == 16 || D == 32) && (size_t)slots * 128 + 256 <= 160 * 1024) {
            // K3f's register image of a slot (csrc/potts_fast_kernels.hip), for models whose every slot is free of internal
            // edges: groups of four (neighbour, value) per lane, the neighbour as the LDS byte address of its 16-bit label cell
            bool any_in_slot = false;
            for (size_t i = 0; i < hm.size(); ++i) any_in_slot = any_in_slot || (hm[i] & 0xffu) != 0u;
            if (!any_in_slot) {
                const int G = D / 4;
                std::vector<uint32_t> ha((size_t)slots * G * 2 * 64 * 4, 0u);
                for (int t = 0; t < slots; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int k = 0; k < D; ++k) {
                            const size_t base = (((size_t)t * G + k / 4) * 2) * 256 + (size_t)lane * 4 + (k & 3);
                            ha[base] = 2u * hc[((size_t)t * D + k) * 64 + lane];
                            memcpy(&ha[base + 256], &hv[((size_t)t * D + k) * 64 + lane], 4);
                        }
                HIP_TRY(hipMalloc((void **)&p->d_adj4p, ha.size() * sizeof(uint32_t)));
                HIP_TRY(hipMemcpy(p->d_adj4p, ha.data(), ha.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
        }
        if (p->kind == MI_KIND_CSR_RANK1) {
            // K2's register image of a slot: groups of four (neighbour, value) per lane, the neighbour already
            // translated into where its state bit lives in LDS (the state masks start at LDS address 0)
            const int G = D / 4;
            // state in LDS: a half per variable while 16 replicas fit one CU (n <= 4608), else a byte (n <= 9216), else a bit
            // (MI_K2_STATE = bit | byte | half narrows the choice: A/B timing of the three forms on one model)
            const char *force = getenv("MI_K2_STATE");
#ifdef MI_K2_DEBUG_BUILD
            const bool debug_linear = getenv("MI_K2_DEBUG_LINEAR") != nullptr;   // (read once, not per adjacency entry)
#endif
            const bool fits_half = (size_t)slots * 128 * 16 <= 144 * 1024, fits_byte = (size_t)slots * 64 * 16 <= 144 * 1024;
            p->k2_state_bytes = fits_half ? 2 : (fits_byte ? 1 : 0);
            if (force && !strcmp(force, "byte") && fits_byte) p->k2_state_bytes = 1;
            if (force && !strcmp(force, "bit")) p->k2_state_bytes = 0;
            std::vector<uint32_t> ha((size_t)slots * G * 2 * 64 * 4, 0u), hf((size_t)slots, 0u);
            for (int t = 0; t < slots; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    if (hm[(size_t)t * 64 + lane] & 0xffu) hf[t] = 1u;
                    for (int k = 0; k < D; ++k) {
                        const uint32_t c = hc[((size_t)t * D + k) * 64 + lane];
                        uint32_t vb;
                        memcpy(&vb, &hv[((size_t)t * D + k) * 64 + lane], 4);
                        const size_t base = (((size_t)t * G + k / 4) * 2) * 256 + (size_t)lane * 4 + (k & 3);
#ifdef MI_K2_DEBUG_BUILD                                           /* timing-only builds: never in the shipped library */
                        if (debug_linear) {                       // TIMING ONLY (wrong chain): conflict-free gathers
                            ha[base] = (uint32_t)((lane * 2 + ((k * 128) % (slots * 128))));
                            ha[base + 256] = vb;
                            continue;
                        }
#endif
                        ha[base] = p->k2_state_bytes == 2 ? 2u * c
                                 : (p->k2_state_bytes == 1 ? c : ((((c >> 5) * 4u) << 8) | (c & 31u)));
                        ha[base + 256] = vb;
                    }
                }
            HIP_TRY(hipMalloc((void **)&p->d_adj4, ha.size() * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void **)&p->d_slot_flags, hf.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(p->d_adj4, ha.data(), ha.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_slot_flags, hf.data(), hf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            // K2p: every slot free of internal edges, D = 16 / 32, 8 wavefronts x 4 bytes per variable fit a CU's LDS
            bool any_general = false;
            for (int t = 0; t < slots; ++t) any_general = any_general || hf[t] != 0u;
            // K2s: the widest block of whole slots that holds no edge (a layout planned with slot = 128 / 256 seats)
            p->k2_free_block = any_general ? 0 : 64;
            for (int B : {256, 128}) {
                if (any_general || slots % 4 != 0) continue;         // (whole groups of four slots: one Philox block each)
                bool ok = true;
                for (int i = 0; i < n && ok; ++i)
                    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                        if (col[e] / B == i / B) { ok = false; break; }
                if (ok) { p->k2_free_block = B; break; }
            }
            // (the pair packing serves K2p and the few-replica kernels: 4 bytes of LDS per seat and replica pair / replica;
            // whether a RUN fits the CUs' LDS is decided per launch, mi_sa_anneal)
            if (!any_general && (D == 16 || D == 32) && (size_t)slots * 256 + 4096 <= 160 * 1024) {
                for (int t = 0; t < slots; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int k = 0; k < D; ++k)
                            ha[(((size_t)t * G + k / 4) * 2) * 256 + (size_t)lane * 4 + (k & 3)] =
#ifdef MI_K2_DEBUG_BUILD
                                debug_linear ? (uint32_t)(lane * 4 + (k * 256) % (slots * 256)) :   // TIMING ONLY: conflict-free gathers
#endif
                                4u * hc[((size_t)t * D + k) * 64 + lane];
                HIP_TRY(hipMalloc((void **)&p->d_adj4p, ha.size() * sizeof(uint32_t)));
                HIP_TRY(hipMemcpy(p->d_adj4p, ha.data(), ha.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            }
        }
    }
    return MI_OK;
}

// Host-only planning step (no device is touched): the slot-independent sweep order of the structured kernels.
static int plan_slot_order_impl(const int32_t *rowptr, const int32_t *col, int n, int slot, int64_t *perm)
{
    const int nslots = (n + slot - 1) / slot;
    if (nslots <= 1 || n > (1 << 18)) {                      // the greedy pass is O(n * slots): identity beyond 262144
        for (int i = 0; i < n; ++i) perm[i] = i;
        return MI_OK;
    }
    for (int i = 0; i < n; ++i)
        if (rowptr[i + 1] < rowptr[i]) return fail(MI_EINVAL, "rowptr is not monotone at %d", i);
    // variables by descending degree, ties by index (stable counting sort)
    int maxdeg = 0;
    for (int i = 0; i < n; ++i) maxdeg = std::max(maxdeg, rowptr[i + 1] - rowptr[i]);
    std::vector<int> start((size_t)maxdeg + 2, 0), order((size_t)n);
    for (int i = 0; i < n; ++i) start[(size_t)(maxdeg - (rowptr[i + 1] - rowptr[i])) + 1]++;
    for (int d = 0; d <= maxdeg; ++d) start[(size_t)d + 1] += start[(size_t)d];
    for (int i = 0; i < n; ++i) order[(size_t)start[(size_t)(maxdeg - (rowptr[i + 1] - rowptr[i]))]++] = i;
    std::vector<int> fill((size_t)nslots, 0), cap((size_t)nslots, slot), where((size_t)n, -1), stamp((size_t)nslots, -1);
    cap[(size_t)nslots - 1] = n - slot * (nslots - 1);
    for (int v : order) {
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            if (col[e] < 0 || col[e] >= n) return fail(MI_EINVAL, "bad column %d in row %d", col[e], v);
            const int w = where[(size_t)col[e]];
            if (w >= 0) stamp[(size_t)w] = v;                // slot w holds a neighbour of v
        }
        // least filled slot that holds no neighbour (ties: lowest index); one that does only as a last resort
        int best = -1, best_nb = -1;
        for (int s = 0; s < nslots; ++s) {
            if (fill[(size_t)s] >= cap[(size_t)s]) continue;
            if (stamp[(size_t)s] == v) { if (best_nb < 0 || fill[(size_t)s] < fill[(size_t)best_nb]) best_nb = s; }
            else if (best < 0 || fill[(size_t)s] < fill[(size_t)best]) best = s;
        }
        const int s = best >= 0 ? best : best_nb;
        where[(size_t)v] = s;
        fill[(size_t)s]++;
    }
    // by slot, then by original index
    std::vector<int> pos((size_t)nslots + 1, 0);
    for (int i = 0; i < n; ++i) pos[(size_t)where[(size_t)i] + 1]++;
    for (int s = 0; s < nslots; ++s) pos[(size_t)s + 1] += pos[(size_t)s];
    for (int i = 0; i < n; ++i) perm[pos[(size_t)where[(size_t)i]]++] = i;
    return MI_OK;
}

// One greedy pass with `nslots` slots of `slot` seats each (no short last slot: holes may sit anywhere).  Returns the
// number of variables that had to share a slot with a neighbour; where[i] = slot of variable i.
static int greedy_slots(const int32_t *rowptr, const int32_t *col, int n, int slot, int nslots,
                        const std::vector<int> &order, std::vector<int> &where)
{
    std::vector<int> fill((size_t)nslots, 0), stamp((size_t)nslots, -1);
    where.assign((size_t)n, -1);
    int clashes = 0;
    for (int v : order) {
        for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            const int w = where[(size_t)col[e]];
            if (w >= 0) stamp[(size_t)w] = v;                // slot w holds a neighbour of v
        }
        int best = -1, best_nb = -1;
        for (int s = 0; s < nslots; ++s) {
            if (fill[(size_t)s] >= slot) continue;
            if (stamp[(size_t)s] == v) { if (best_nb < 0 || fill[(size_t)s] < fill[(size_t)best_nb]) best_nb = s; }
            else if (best < 0 || fill[(size_t)s] < fill[(size_t)best]) best = s;
        }
        const int s = best >= 0 ? best : best_nb;
        if (best < 0) ++clashes;
        where[(size_t)v] = s;
        fill[(size_t)s]++;
    }
    return clashes;
}

static int plan_slot_layout_impl(const int32_t *rowptr, const int32_t *col, int n, int slot, int max_slots,
                                 int64_t *pos, int *out_slots, int *out_clashes)
{
    const int s0 = (n + slot - 1) / slot;
    if (max_slots < s0) max_slots = s0;
    for (int i = 0; i < n; ++i) {
        if (rowptr[i + 1] < rowptr[i]) return fail(MI_EINVAL, "rowptr is not monotone at %d", i);
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
            if (col[e] < 0 || col[e] >= n) return fail(MI_EINVAL, "bad column %d in row %d", col[e], i);
    }
    if (n > (1 << 18)) {
        // a greedy pass is O(n * slots) and repeats while the layout grows: beyond 262144 variables the packed identity
        // layout is returned (as mi_sa_plan_slot_order does); clashes = variables with a neighbour in their own slot
        int clashes = 0;
        for (int i = 0; i < n; ++i) {
            pos[i] = i;
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
                if (col[e] / slot == i / slot) { ++clashes; break; }
        }
        *out_slots = s0;
        if (out_clashes) *out_clashes = clashes;
        return MI_OK;
    }
    // variables by descending degree, ties by index (stable counting sort) -- as mi_sa_plan_slot_order
    int maxdeg = 0;
    for (int i = 0; i < n; ++i) maxdeg = std::max(maxdeg, rowptr[i + 1] - rowptr[i]);
    std::vector<int> start((size_t)maxdeg + 2, 0), order((size_t)n);
    for (int i = 0; i < n; ++i) start[(size_t)(maxdeg - (rowptr[i + 1] - rowptr[i])) + 1]++;
    for (int d = 0; d <= maxdeg; ++d) start[(size_t)d + 1] += start[(size_t)d];
    for (int i = 0; i < n; ++i) order[(size_t)start[(size_t)(maxdeg - (rowptr[i + 1] - rowptr[i]))]++] = i;
    // fewest slots (from the fully packed count up, +1/8 per try) that leave no edge inside a slot; the packed
    // layout's clashes stay if max_slots does not suffice
    std::vector<int> where, first_where;
    int nslots = s0, clashes = 0, first_clashes = 0;
    for (;;) {
        clashes = greedy_slots(rowptr, col, n, slot, nslots, order, where);
        if (nslots == s0) { first_where = where; first_clashes = clashes; }
        if (clashes == 0) break;
        const int next = nslots + std::max(1, nslots / 8);
        if (next > max_slots) { where = first_where; clashes = first_clashes; nslots = s0; break; }
        nslots = next;
    }
    // Small graphs: when holes were needed, the block count is set by how many COLOURS the graph needs, not by n / slot,
    // and the balanced greedy pass wastes some (10 blocks for a 342-cell cluster that 8 colour).  A saturation-degree
    // colouring (DSATUR: always the uncoloured variable that sees the most colours, ties by degree, then index; lowest
    // colour with a free seat) is tried as well, and kept when it needs fewer blocks -- every block is a dependent step
    // of a sweep, so 8 instead of 10 is 20 % of a small model's kernel time.  O(n^2): graphs up to 2048 variables.
    if (clashes == 0 && nslots > s0 && n <= 2048) {
        const int C = nslots;                                        // only fewer colours than the greedy result are of interest
        std::vector<unsigned char> seen((size_t)n * C, 0);
        std::vector<int> sat((size_t)n, 0), colour((size_t)n, -1), fill2((size_t)C, 0);
        int used = 0;
        bool ok = true;
        for (int step = 0; step < n && ok; ++step) {
            int v = -1;
            for (int u = 0; u < n; ++u) {
                if (colour[(size_t)u] >= 0) continue;
                if (v < 0) { v = u; continue; }
                const int du = rowptr[u + 1] - rowptr[u], dv = rowptr[v + 1] - rowptr[v];
                if (sat[(size_t)u] > sat[(size_t)v] || (sat[(size_t)u] == sat[(size_t)v] && du > dv)) v = u;
            }
            int c = 0;
            while (c < C && (seen[(size_t)v * C + c] || fill2[(size_t)c] >= slot)) ++c;
            if (c >= C) { ok = false; break; }
            colour[(size_t)v] = c;
            fill2[(size_t)c]++;
            used = std::max(used, c + 1);
            if (used >= C) { ok = false; break; }                    // no better than the greedy layout
            for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
                const int u = col[e];
                if (!seen[(size_t)u * C + c]) { seen[(size_t)u * C + c] = 1; sat[(size_t)u]++; }
            }
        }
        if (ok && used < nslots && used >= s0) {
            where = colour;
            nslots = used;
        }
    }
    std::vector<int> seat((size_t)nslots, 0);
    for (int i = 0; i < n; ++i) pos[i] = (int64_t)where[(size_t)i] * slot + seat[(size_t)where[(size_t)i]]++;   // by index inside a slot
    *out_slots = nslots;
    if (out_clashes) *out_clashes = clashes;
    return MI_OK;
}

int mi_sa_plan_slot_layout(const int32_t *rowptr, const int32_t *col, int n, int slot, int max_slots,
                           int64_t *out_pos, int *out_slots, int *out_clashes)
{
    if (!rowptr || !out_pos || !out_slots || (n > 0 && rowptr[n] > 0 && !col)) return fail(MI_EINVAL, "NULL argument");
    if (n < 1 || slot < 1) return fail(MI_EINVAL, "n and slot must be >= 1");
    return guarded([&]() -> int { return plan_slot_layout_impl(rowptr, col, n, slot, max_slots, out_pos, out_slots, out_clashes); });
}

int mi_sa_plan_slot_order(const int32_t *rowptr, const int32_t *col, int n, int slot, int64_t *out_perm)
{
    if (!rowptr || !out_perm || (n > 0 && rowptr[n] > 0 && !col)) return fail(MI_EINVAL, "NULL argument");
    if (n < 0 || slot < 1) return fail(MI_EINVAL, "n must be >= 0 and slot >= 1");
    return guarded([&]() -> int { return plan_slot_order_impl(rowptr, col, n, slot, out_perm); });
}

int mi_sa_problem_create_csr_rank1_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                       const float *lin, float c_pair, int n, double offset, int device,
                                       mi_sa_problem **out)
{
    if (!rowptr || !lin || !out || (rowptr[n > 0 ? n : 0] > 0 && (!col || !val))) return fail(MI_EINVAL, "NULL argument");
    if (n < 1) return fail(MI_EINVAL, "n must be >= 1 (got %d)", n);
    if (n > (1 << 20)) return fail(MI_EUNSUPPORTED, "csr_rank1 kernel supports n <= 1048576 (got %d)", n);
    int rc = select_device(device);
    if (rc) return rc;
    mi_sa_problem *p = new (std::nothrow) mi_sa_problem();
    if (!p) return fail(MI_ENOMEM, "out of host memory");
    p->kind = MI_KIND_CSR_RANK1; p->n = n; p->K = 2; p->offset = offset; p->state_elem = 1; p->c_pair = c_pair;
    rc = problem_common_init(p, device);
    if (!rc) rc = guarded([&]() -> int { return upload_slot_ell(p, rowptr, col, val, n); });
    if (!rc) rc = guarded([&]() -> int {
        // the lanes past n carry lin = +inf: their dE is +inf, never accepted (K2 has no per-lane bound check)
        std::vector<float> hl((size_t)p->slots * 64, INFINITY);
        for (int i = 0; i < n; ++i) hl[i] = lin[i];
        p->h_hole.assign((size_t)n, 0);
        for (int i = 0; i < n; ++i) p->h_hole[(size_t)i] = std::isinf(lin[i]) ? 1 : 0;
        HIP_TRY(hipMalloc((void **)&p->d_lin, hl.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(p->d_lin, hl.data(), hl.size() * sizeof(float), hipMemcpyHostToDevice));
        return MI_OK;
    });
    if (rc) { mi_sa_problem_destroy(p); return rc; }
    *out = p;
    return MI_OK;
}

int mi_sa_problem_create_potts_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                                       float c_pair, int n, int K, double lin_offset, int device,
                                       mi_sa_problem **out)
{
    if (!rowptr || !out || (rowptr[n > 0 ? n : 0] > 0 && (!col || !val))) return fail(MI_EINVAL, "NULL argument");
    if (n < 1) return fail(MI_EINVAL, "n must be >= 1 (got %d)", n);
    if (K < 1 || K > 64) return fail(MI_EUNSUPPORTED, "potts kernel supports 1 <= K <= 64 cases (got %d)", K);
    if (n > 40000) return fail(MI_EUNSUPPORTED, "potts kernel supports n <= 40000 (got %d)", n);
    int rc = select_device(device);
    if (rc) return rc;
    mi_sa_problem *p = new (std::nothrow) mi_sa_problem();
    if (!p) return fail(MI_ENOMEM, "out of host memory");
    p->kind = MI_KIND_POTTS_CSR; p->n = n; p->K = K; p->offset = lin_offset; p->state_elem = 2; p->c_pair = c_pair;
    rc = problem_common_init(p, device);
    if (!rc) rc = guarded([&]() -> int { return upload_slot_ell(p, rowptr, col, val, n); });
    if (rc) { mi_sa_problem_destroy(p); return rc; }
    *out = p;
    return MI_OK;
}

static int set_energy_model_impl(mi_sa_problem *p, const double *val, const double *lin, double c_pair)
{
    if (const int rc_w = settle(p)) return rc_w;
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (p->kind != MI_KIND_CSR_RANK1 && p->kind != MI_KIND_POTTS_CSR)
        return fail(MI_EUNSUPPORTED, "an fp64 energy model is defined for the structured kinds only");
    const int n = p->n, D = p->D;
    const int64_t nnz = p->h_rowptr.empty() ? 0 : p->h_rowptr[n];
    if ((nnz > 0 && !val) || (p->kind == MI_KIND_CSR_RANK1 && !lin)) return fail(MI_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    std::vector<double> hv((size_t)p->slots * D * 64, 0.0), hl((size_t)p->slots * 64, 0.0);
    for (int i = 0; i < n; ++i) {
        const int t = i >> 6, lane = i & 63;
        for (int e = p->h_rowptr[i], k = 0; e < p->h_rowptr[i + 1]; ++e, ++k) hv[((size_t)t * D + k) * 64 + lane] = val[e];
        if (lin) hl[i] = lin[i];
    }
    if (!p->d_ell_val64) HIP_TRY(hipMalloc((void **)&p->d_ell_val64, hv.size() * sizeof(double)));
    if (!p->d_lin64) HIP_TRY(hipMalloc((void **)&p->d_lin64, hl.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(p->d_ell_val64, hv.data(), hv.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_lin64, hl.data(), hl.size() * sizeof(double), hipMemcpyHostToDevice));
    p->c_pair64 = c_pair;
    return MI_OK;
}

int mi_sa_problem_set_absent(mi_sa_problem *p, const uint8_t *absent)
{
    if (!p || !absent) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (p->kind != MI_KIND_POTTS_CSR)
        return fail(MI_EUNSUPPORTED, "holes of a Potts model only (a binary CSR model marks them by lin = +inf)");
    return guarded([&]() -> int {
        for (int i = 0; i < p->n; ++i) {
            if (absent[i] && (p->h_meta[(size_t)i] & 0x00ffff00u))
                return fail(MI_EINVAL, "variable %d is marked absent but has couplings", i);
            p->h_meta[(size_t)i] = (p->h_meta[(size_t)i] & 0x7fffffffu) | (absent[i] ? 0x80000000u : 0u);
        }
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpy(p->d_meta, p->h_meta.data(), p->h_meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        return MI_OK;
    });
}

int mi_sa_problem_set_pair_weights(mi_sa_problem *p, const int32_t *weights)
{
    if (!p || !weights) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (p->kind != MI_KIND_CSR_RANK1)
        return fail(MI_EUNSUPPORTED, "pair-term weights: structured binary (CSR + uniform pair) problems only");
    return guarded([&]() -> int {
        int wslot = -1;
        for (int i = 0; i < p->n; ++i) {
            if (p->h_meta[(size_t)i] >> 31) continue;          // (never set for this kind; holes are marked by lin = +inf)
            const bool hole = !p->h_hole.empty() && p->h_hole[(size_t)i];
            if (hole) continue;
            if (weights[i] < 1) return fail(MI_EINVAL, "weight %d of variable %d: weights are positive integers", weights[i], i);
            if (weights[i] > (1 << 20)) return fail(MI_EINVAL, "weight %d of variable %d exceeds 2^20", weights[i], i);
            if (weights[i] == 1) continue;
            if (p->h_rowptr[(size_t)i + 1] != p->h_rowptr[(size_t)i])
                return fail(MI_EINVAL, "variable %d has weight %d and sparse couplings: weighted variables couple through the pair term only", i, weights[i]);
            if (wslot >= 0 && wslot != i / 64)
                return fail(MI_EINVAL, "variables with weights other than 1 in slots %d and %d: they must share one 64-variable slot", wslot, i / 64);
            wslot = i / 64;
        }
        std::vector<int32_t> hw(64, 0);
        if (wslot >= 0) {
            for (int l = 0; l < 64; ++l) {
                const int i = wslot * 64 + l;
                if (i >= p->n || (!p->h_hole.empty() && p->h_hole[(size_t)i])) continue;
                hw[(size_t)l] = weights[i];
            }
            // a variable of weight 1 may share the slot (it is swept by the same serial loop) -- but it must have no sparse
            // couplings either: the loop does not update neighbours
            for (int l = 0; l < 64; ++l) {
                const int i = wslot * 64 + l;
                if (i < p->n && hw[(size_t)l] != 0 && p->h_rowptr[(size_t)i + 1] != p->h_rowptr[(size_t)i])
                    return fail(MI_EINVAL, "variable %d shares the weighted slot %d and has sparse couplings", i, wslot);
            }
        }
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (!p->d_wgt) HIP_TRY(hipMalloc((void **)&p->d_wgt, 64 * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->d_wgt, hw.data(), 64 * sizeof(int32_t), hipMemcpyHostToDevice));
        p->wslot = wslot;
        return MI_OK;
    });
}

int mi_sa_problem_set_energy_model_f64(mi_sa_problem *p, const double *val, const double *lin, double c_pair)
{
    return guarded([&]() -> int { return set_energy_model_impl(p, val, lin, c_pair); });
}

int mi_sa_problem_destroy(mi_sa_problem *p)
{
    if (!p) return MI_OK;
    (void)settle(p);
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void *bufs[] = {p->d_wgt, p->d_xg, p->d_pt_rung, p->d_pt_betas, p->d_pt_energy, p->d_pt_ladder, p->d_pt_temps, p->d_pt_stats, p->d_adj4p, p->d_ell_val64, p->d_lin64, p->d_Q2xl, p->d_diagxl, p->d_rows, p->d_meta, p->d_adj4, p->d_slot_flags, p->d_Qm, p->d_fields, p->d_ctrl, p->d_ell_col, p->d_ell_val, p->d_lin, p->d_pace, p->d_Qp, p->d_Qs, p->d_temps, p->d_init, p->d_states, p->d_energy, p->d_stats};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return MI_OK;
}

int mi_sa_problem_info(const mi_sa_problem *p, int *kind, int *n, int *num_cases, int *device)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (kind) *kind = p->kind;
    if (n) *n = p->n;
    if (num_cases) *num_cases = p->K;
    if (device) *device = p->device;
    return MI_OK;
}

int mi_sa_debug_stats(mi_sa_problem *p, uint64_t *out, int words)
{
    if (!p || !out) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (words > 16) words = 16;
    HIP_TRY(hipMemcpy(out, p->d_stats, (size_t)words * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (words == 16 && p->d_ctrl) {                         // [14], [15]: chunks of the last scheduled dense run served by K1w / K1m
        unsigned int c[2] = {0, 0};
        HIP_TRY(hipMemcpy(c, p->d_ctrl + 4, sizeof c, hipMemcpyDeviceToHost));
        out[14] = c[0]; out[15] = c[1];
    }
    return MI_OK;
}

int mi_sa_debug_pace(mi_sa_problem *p, unsigned int *out, int words)
{
    if (!p || !out) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (words > kPaceWords) words = kPaceWords;
    HIP_TRY(hipMemcpy(out, p->d_pace, (size_t)words * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_sa_set_option(mi_sa_problem *p, const char *key, long value)
{
    if (!p || !key) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (!strcmp(key, "pace")) { p->opt_pace = value != 0; return MI_OK; }
    if (!strcmp(key, "xl_batched") && value >= 0 && value <= 2) { p->opt_xl_batched = (int)value; return MI_OK; }
    if (!strcmp(key, "xl_chunk") && value >= 1) { p->opt_xl_chunk = (int)value; return MI_OK; }
    if (!strcmp(key, "xl_chain") && value >= 0 && value <= 2) { p->opt_xl_chain = (int)value; return MI_OK; }
    if (!strcmp(key, "xl_cold_permille") && value >= 0 && value <= 1000) { p->opt_xl_cold_permille = (int)value; return MI_OK; }
    if (!strcmp(key, "xl_async") && value >= 0 && value <= 1) { p->opt_xl_async = (int)value; return MI_OK; }
    if (!strcmp(key, "mfma_permille") && value >= 0 && value <= 1000) { p->opt_mfma_permille = (int)value; return MI_OK; }
    if (!strcmp(key, "chunk_sweeps") && value >= 0) { p->opt_chunk_sweeps = (int)value; return MI_OK; }
    if (!strcmp(key, "ondemand_permille") && value >= 0 && value <= 1000) { p->opt_ondemand_permille = (int)value; return MI_OK; }
    if (!strcmp(key, "debug")) { p->opt_debug = (int)value; return MI_OK; }
    if (!strcmp(key, "k2_waves") && ((value >= 0 && value <= 16) || value == 99)) { p->opt_k2_waves = (int)value; return MI_OK; }   // (99: K3 keeps its serial move loop -- A/B timing)
    if (!strcmp(key, "k2_pair") && value >= 0 && value <= 2) { p->opt_k2_pair = (int)value; return MI_OK; }
    if (!strcmp(key, "k2_split") && value >= 0 && value <= 2) { p->opt_k2_split = (int)value; return MI_OK; }
    if (!strcmp(key, "k2_split_max") && value >= 0) { p->opt_k2_split_max = (int)value; return MI_OK; }
    if (!strcmp(key, "k2_wide") && value >= 0 && value <= 2) { p->opt_k2_wide = (int)value; return MI_OK; }
    if (!strcmp(key, "k2_tw") && value >= 0 && value <= 2) { p->opt_k2_tw = (int)value; return MI_OK; }
    if (!strcmp(key, "k3_fast") && value >= 0 && value <= 2) { p->opt_k3_fast = (int)value; return MI_OK; }
    if (!strcmp(key, "min_cluster_size") && value >= 0) {
        if (p->kind != MI_KIND_POTTS_CSR) return fail(MI_EINVAL, "min_cluster_size applies to Potts problems");
        p->opt_min_cluster_size = (int)value;
        return MI_OK;
    }
    if (!strcmp(key, "variant") && value >= 0 && value <= 4) { p->opt_variant = (int)value; return MI_OK; }
    if (!strcmp(key, "unit_rows") && (value == 0 || value == 2 || value == 4)) { p->opt_unit_rows = (int)value; return MI_OK; }
    return fail(MI_EINVAL, "unknown option '%s'", key);
}

static int anneal_ex_impl(mi_sa_problem *p, int R, uint32_t replica_offset, int num_sweeps,
                          const double *betas, uint64_t seed, const void *init, int resync_interval,
                          uint32_t sweep_offset, uint32_t flags)
{
    if (const int rc_w = settle(p)) return rc_w;
    const bool cont = (flags & MI_F_CONTINUE) != 0, resident = (flags & MI_F_TEMPS_RESIDENT) != 0;
    const bool per_replica = (flags & MI_F_BETA_PER_REPLICA) != 0 || resident;
    const int num_betas = per_replica ? R : num_sweeps;
    if (flags & ~(uint32_t)(MI_F_CONTINUE | MI_F_BETA_PER_REPLICA | MI_F_TEMPS_RESIDENT)) return fail(MI_EINVAL, "unknown flags 0x%x", flags);
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (resident && (p->pt_T == 0 || p->pt_R_local != R))
        return fail(MI_ESTATE, "MI_F_TEMPS_RESIDENT needs mi_sa_tempering_begin for %d replicas on this problem", R);
    if (R < 1) return fail(MI_EINVAL, "R must be >= 1 (got %d)", R);
    if (num_sweeps < 0) return fail(MI_EINVAL, "num_sweeps must be >= 0");
    if (num_betas > 0 && num_sweeps > 0 && !betas && !resident) return fail(MI_EINVAL, "betas is NULL");
    if (cont && (!p || !p->has_run || p->last_R != R))
        return fail(MI_ESTATE, "MI_F_CONTINUE needs a previous run with the same number of replicas");
    if (cont && init) return fail(MI_EINVAL, "MI_F_CONTINUE and init are mutually exclusive");
    if (resync_interval < 0) return fail(MI_EINVAL, "resync_interval must be >= 0");
    for (int s = 0; s < (num_sweeps > 0 && !resident ? num_betas : 0); ++s)
        if (!(betas[s] > 0.0) || !std::isfinite(betas[s]))
            return fail(MI_EINVAL, "betas[%d] = %g is not a positive finite number", s, betas[s]);
    HIP_TRY(hipSetDevice(p->device));
    int rc = ensure_run_buffers(p, R, num_betas, init != nullptr);
    if (rc) return rc;
    if (!resident) {                             // (tempering rounds: the exchange kernel keeps temps[] up to date)
        std::vector<float> temps((size_t)(num_betas > 0 ? num_betas : 1), 1.0f);
        for (int s = 0; s < (num_sweeps > 0 ? num_betas : 0); ++s) temps[s] = (float)(1.0 / betas[s]);
        // pageable-host async copies are staged synchronously by the runtime: the vector may go away
        HIP_TRY(hipMemcpyAsync(p->d_temps, temps.data(), temps.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
    }
    if (init)
        HIP_TRY(hipMemcpyAsync(p->d_init, init, (size_t)R * p->n * p->state_elem, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_stats, 0, 16 * sizeof(unsigned long long), p->stream));
    if (!resident || init)
        HIP_TRY(hipStreamSynchronize(p->stream));   // inputs resident before the timed region
    g_kernel.clear();
    const float *temps_buf = resident ? p->d_pt_temps : p->d_temps;

    if (p->kind == MI_KIND_DENSE && p->xl_chunks > 0) {
        DenseXlArgs a;
        a.Q2 = p->d_Q2xl; a.diag = p->d_diagxl; a.temps = temps_buf;
        a.init = cont ? (const uint8_t *)p->d_states : (init ? (const uint8_t *)p->d_init : nullptr);
        a.states = (uint8_t *)p->d_states; a.energy = p->d_energy; a.stats = p->d_stats;
        a.offset = p->offset; a.n = p->n; a.R = R; a.num_sweeps = num_sweeps; a.resync = resync_interval;
        a.replica_offset = replica_offset; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        a.sweep_offset = sweep_offset; a.temps_per_replica = per_replica ? 1 : 0;
        a.xg_chain = p->opt_xl_chain;
        p->last_launches = 1;
        // K1x pays per accepted flip (a barrier and the L2 latency of one Q row: ~1 us up to n = 8192, 3 us at 20 000,
        // 8 us at 50 000) and runs 256 replicas at a time; K1g pays ~25 us per 64 rows whatever the replica count
        // below 256.  Measured (profiles/r02_xl_crossover.json): K1x wins at n <= 8192 with <= 64 replicas (2-3x),
        // K1g from n = 20 000 at any count (1.1-2x over a whole schedule, 9x on its hot part at 50 000).
        const bool batched = p->opt_xl_batched == 1 || (p->opt_xl_batched == 0 && (R >= 256 || p->n >= 16384));
        if (batched) {
            const size_t need = mi_dense_xg_workspace_bytes(p->n, R);
            if (need > p->xg_bytes) {
                if (p->d_xg) HIP_TRY(hipFree(p->d_xg));
                p->d_xg = nullptr; p->xg_bytes = 0;
                HIP_TRY(hipMalloc(&p->d_xg, need));
                p->xg_bytes = need;
            }
        }
        HIP_TRY(hipEventRecord(p->ev0, p->stream));
        if (!batched) {
            rc = mi_launch_dense_xl(a, p->xl_chunks, p->stream);
        } else if (resync_interval > 0 || per_replica || num_sweeps <= p->opt_xl_chunk || p->opt_xl_cold_permille == 0) {
            rc = mi_launch_dense_xg(a, p->xl_chunks, p->d_xg, p->stream, 3);
        } else {
            // K1g costs the same hot or cold (two launches per 64 rows whether anything flips or not); K1x costs per
            // accepted flip.  Along a cooling schedule: K1g in chunks of sweeps while the chunks accept enough, then
            // K1x for the rest, continuing from K1g's states AND cached fields (same chain, bit for bit).  The hand-over
            // is decided on the host, chunk by chunk -- a device-side mode word as on the n <= 4096 scheduler would need
            // every chunk's launches enqueued in advance (1600 per sweep at n = 50 000, each an empty launch once the
            // run has gone cold: seconds) -- so a worker thread of the problem waits for the chunks and this call
            // returns at once, like every other anneal; the next call on the problem joins it (settle).
            auto cooling_run = [p, a, num_sweeps, R]() -> int {
                HIP_TRY(hipSetDevice(p->device));
                int rc = MI_OK, s0 = 0;
                while (!rc && s0 < num_sweeps) {
                    const int len = num_sweeps - s0 < p->opt_xl_chunk ? num_sweeps - s0 : p->opt_xl_chunk;
                    DenseXlArgs b = a;
                    b.num_sweeps = len; b.temps = a.temps + s0; b.sweep_offset = a.sweep_offset + (uint32_t)s0;
                    unsigned long long before = 0, after = 0;
                    HIP_TRY(hipMemcpyAsync(&before, p->d_stats + 1, sizeof before, hipMemcpyDeviceToHost, p->stream));
                    rc = mi_launch_dense_xg(b, p->xl_chunks, p->d_xg, p->stream, (s0 == 0 ? 1 : 0) | 2);
                    if (rc) break;
                    s0 += len;
                    if (s0 >= num_sweeps) break;
                    HIP_TRY(hipMemcpyAsync(&after, p->d_stats + 1, sizeof after, hipMemcpyDeviceToHost, p->stream));
                    HIP_TRY(hipStreamSynchronize(p->stream));
                    const double share = (double)(after - before) / ((double)R * (double)p->n * (double)len);
                    if (share * 1000.0 < (double)p->opt_xl_cold_permille) {
                        DenseXlArgs c = a;
                        c.num_sweeps = num_sweeps - s0; c.temps = a.temps + s0; c.sweep_offset = a.sweep_offset + (uint32_t)s0;
                        c.init = (const uint8_t *)p->d_states;             // written by the chunk that just ended
                        c.fields_in = mi_dense_xg_fields(p->d_xg);
                        c.fin_ncols = (p->n + 255) / 256 * 256;
                        rc = mi_launch_dense_xl(c, p->xl_chunks, p->stream);
                        break;
                    }
                }
                if (rc) return rc;
                HIP_TRY(hipEventRecord(p->ev1, p->stream));
                return MI_OK;
            };
            if (p->opt_xl_async) {
                p->last_R = R; p->last_offset = replica_offset; p->has_run = true;
                p->worker = std::thread([p, cooling_run]() {
                    g_kernel.clear(); g_err.clear();
                    p->worker_rc = cooling_run();
                    p->worker_err = g_err;
                    p->worker_kernel = g_kernel;
                });
                return MI_OK;
            }
            rc = cooling_run();
            if (rc) return rc;
            p->last_R = R; p->last_offset = replica_offset; p->has_run = true;
            p->last_kernel = g_kernel;
            return MI_OK;
        }
        if (rc) return rc;
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
    } else if (p->kind == MI_KIND_DENSE) {
        DenseArgs a;
        a.Qp = p->d_Qp; a.Qm = p->d_Qm; a.temps = temps_buf;
        a.init = cont ? (const uint8_t *)p->d_states : (init ? (const uint8_t *)p->d_init : nullptr);
        a.states = (uint8_t *)p->d_states; a.energy = p->d_energy; a.stats = p->d_stats; a.pace = nullptr;
        a.offset = p->offset; a.n = p->n; a.R = R; a.num_sweeps = num_sweeps; a.resync = resync_interval;
        a.replica_offset = replica_offset; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        a.debug = p->opt_debug; a.ondemand_flips = 0; a.sweep_offset = sweep_offset; a.temps_per_replica = per_replica ? 1 : 0;
        HIP_TRY(hipEventRecord(p->ev0, p->stream));
        rc = dispatch_dense(p, a, p->stream);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
    } else {
        EllArgs a;
        a.ell_col = p->d_ell_col; a.ell_val = p->d_ell_val; a.lin = p->d_lin; a.temps = temps_buf;
        a.init = cont ? p->d_states : (init ? p->d_init : nullptr); a.states = p->d_states; a.energy = p->d_energy; a.stats = p->d_stats;
        a.c_pair = p->c_pair; a.offset = p->offset; a.n = p->n; a.K = p->K; a.R = R; a.num_sweeps = num_sweeps;
        a.resync = resync_interval; a.slots = p->slots; a.D = p->D;
        a.replica_offset = replica_offset; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        a.sweep_offset = sweep_offset; a.temps_per_replica = per_replica ? 1 : 0;
        a.rows = p->d_rows; a.meta = p->d_meta; a.adj4 = p->d_adj4; a.slot_flags = p->d_slot_flags; a.state_bytes = p->k2_state_bytes; a.waves_override = p->opt_k2_waves; a.min_size = p->opt_min_cluster_size;
        a.ell_val64 = p->d_ell_val64; a.lin64 = p->d_lin64; a.c_pair64 = p->c_pair64;
        a.wgt = p->d_wgt; a.wslot = p->kind == MI_KIND_CSR_RANK1 ? p->wslot : -1;
        if (p->kind == MI_KIND_POTTS_CSR && init) {
            // labels must be < K: validated on the host copy (the device trusts them as cnt[] indices)
            const uint16_t *l = static_cast<const uint16_t *>(init);
            for (size_t k = 0; k < (size_t)R * p->n; ++k)
                if (l[k] >= (uint16_t)p->K) return fail(MI_EINVAL, "initial label %u >= K = %d", (unsigned)l[k], p->K);
        }
        p->last_launches = 1;
        HIP_TRY(hipEventRecord(p->ev0, p->stream));
        if (p->kind == MI_KIND_POTTS_CSR) {
            if (p->d_adj4p && p->opt_k3_fast != 2 && mi_potts_fast_eligible(p->D, p->K, a.min_size)) {
                a.adj4 = p->d_adj4p;                  // every slot free of internal edges: the lean kernel (same chain)
                // (up to 1024 replicas every wavefront has a SIMD to itself: a threshold wavefront beside each)
                rc = mi_launch_potts_fast(a, p->opt_k2_tw != 2 && R <= 1024, p->stream);
            } else {
                rc = mi_launch_potts(a, p->stream);
            }
        } else {
            // which of the kernels of the structured binary model (all run the same chain): an explicit option first;
            // otherwise few replicas -> K2s in its one-wavefront form (random words a few rounds per step, 32-bit state
            // cells: 5 % faster than K2 / K2p when every wavefront has a SIMD to itself; its 2 / 4-wavefront forms only on
            // request: measured break-even), more replicas than the chip has SIMDs -> two replicas per wavefront, else one
            const bool split_ok = p->k2_free_block >= 64 && p->d_adj4p != nullptr, pair_ok = p->d_adj4p != nullptr;
            const bool tw = p->opt_k2_tw != 2;         // a threshold wavefront beside the sweeping one (same chain)
            // these kernels keep 4 bytes of LDS per seat: the library's own choice takes them only when the workgroups of
            // the run are resident in ONE round (else K2 with its bit / byte state, 16 replicas per CU at any size)
            const long cus = p->cus > 0 ? p->cus : 256;
            auto one_round = [&](size_t lds_per_wg, long wgs) {
                return lds_per_wg * (size_t)((wgs + cus - 1) / cus) <= (size_t)160 * 1024;
            };
            const size_t cells = (size_t)p->slots * 256;
            // K2p: with its threshold wavefront 8 workgroups (16 replicas) fill a CU -- for runs of up to that many; beyond,
            // the kernel without it holds 16 workgroups per CU (6144 replicas: 4.3e11 against two rounds at 3.5e11).  Models
            // of up to 4608 variables keep 8 workgroups' cells per CU at any replica count (several rounds if need be);
            // larger ones take K2p only when one round holds the run.
            const long pair_wgs = ((long)R + 1) / 2;
            const long tw_rounds = (pair_wgs + 8 * cus - 1) / (8 * cus);
            const bool tw_pair = tw && p->D == 16 &&
                                 ((pair_wgs <= 8 * cus && one_round(cells + 4096, pair_wgs)) ||            // one round, or
                                  (10 * pair_wgs >= 9 * tw_rounds * 8 * cus && (cells + 4096) * 8 <= (size_t)160 * 1024));   // nearly full ones
            const bool pair_run = pair_ok && (cells * 8 <= (size_t)150 * 1024 || one_round(cells + (tw_pair ? 4096 : 0), pair_wgs));
            int choice = 0;
            // (a model with pair-term weights: the kernels that sweep its weighted slot are K2, K2p and K2w with one slot
            // per step beside a threshold wavefront)
            const bool weighted = p->wslot >= 0;
            if (weighted) {
                if (p->opt_k2_pair != 2 && pair_run && p->D == 16 && R > 1024) choice = 1;
                else if (p->opt_k2_split != 2 && tw && p->opt_k2_wide != 2 && split_ok &&
                         R <= p->opt_k2_split_max && one_round(cells + 2048, R)) choice = 2;
            } else
            if (p->opt_k2_split == 1 && split_ok) choice = 2;
            else if (p->opt_k2_pair == 1 && pair_ok) choice = 1;
            else if (p->opt_k2_split != 2 && split_ok && R <= p->opt_k2_split_max && one_round(cells + 2048, R)) choice = 2;
            else if (p->opt_k2_pair != 2 && pair_run && R > 1024) choice = 1;
            if (choice == 2 && weighted) {
                a.adj4 = p->d_adj4p;                  // (an edge-free layout in wider blocks is one in 64-seat slots too)
                rc = mi_launch_csr_rank1_wide(a, 1, true, p->stream);
            } else if (choice == 2 && p->k2_free_block > 64 && p->opt_k2_wide != 2 && (p->D == 16 || p->k2_free_block == 128)) {
                // blocks of 128 / 256 edge-free seats, few replicas: ONE wavefront sweeps a block per step
                a.adj4 = p->d_adj4p;
                rc = mi_launch_csr_rank1_wide(a, p->k2_free_block / 64, tw, p->stream);
            } else if (choice == 2 && p->k2_free_block == 64 && p->opt_k2_wide != 2 && tw) {
                a.adj4 = p->d_adj4p;                  // 64-seat layouts: one slot per step, thresholds from the second wavefront
                rc = mi_launch_csr_rank1_wide(a, 1, true, p->stream);
            } else if (choice == 2) {
                a.adj4 = p->d_adj4p;
                rc = mi_launch_csr_rank1_split(a, p->k2_free_block / 64, p->stream);
            } else if (choice == 1) {
                a.adj4 = p->d_adj4p;                  // two replicas per wavefront: half the adjacency traffic per update
                rc = mi_launch_csr_rank1_pair(a, p->opt_k2_pair == 1 ? (tw && p->D == 16 && pair_wgs <= 8 * cus) : tw_pair, p->stream);
            } else {
                // K2: every wavefront alone on its SIMD (up to 1024 replicas) -> a threshold wavefront beside it
                rc = mi_launch_csr_rank1(a, p->stream, tw && R <= 1024 && (p->D == 16 || p->D == 32) && p->k2_state_bytes <= 1);
            }
        }
        if (rc) return rc;
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
    }
    p->last_R = R; p->last_offset = replica_offset; p->has_run = true;
    p->last_kernel = g_kernel;
    return MI_OK;
}

int mi_sa_tempering_begin(mi_sa_problem *p, const double *ladder_betas, int T, int chains,
                          uint32_t first_replica, int R_local)
{
    if (!p || !ladder_betas) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (T < 2 || T > 1024) return fail(MI_EINVAL, "a tempering ladder has 2 .. 1024 temperatures (got %d)", T);
    if (chains < 1) return fail(MI_EINVAL, "chains must be >= 1");
    const long long total = (long long)T * chains;
    if (R_local < 1 || (long long)first_replica + R_local > total)
        return fail(MI_EINVAL, "local replicas [%u, %u + %d) do not lie inside the %lld replicas of the run", first_replica, first_replica, R_local, total);
    for (int k = 0; k < T; ++k)
        if (!(ladder_betas[k] > 0.0) || !std::isfinite(ladder_betas[k]))
            return fail(MI_EINVAL, "ladder beta %d = %g is not a positive finite number", k, ladder_betas[k]);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    int rc = ensure_run_buffers(p, R_local, R_local, false);
    if (rc) return rc;
    for (void *b : {(void *)p->d_pt_rung, (void *)p->d_pt_betas, (void *)p->d_pt_energy, (void *)p->d_pt_ladder, (void *)p->d_pt_temps, (void *)p->d_pt_stats})
        if (b) (void)hipFree(b);
    p->d_pt_rung = nullptr; p->d_pt_betas = nullptr; p->d_pt_energy = nullptr; p->d_pt_ladder = nullptr; p->d_pt_temps = nullptr; p->d_pt_stats = nullptr;
    p->pt_T = 0;
    rc = guarded([&]() -> int {
        std::vector<int> rung((size_t)total);
        for (long long g = 0; g < total; ++g) rung[(size_t)g] = (int)(g % T);
        std::vector<float> lt((size_t)T), local((size_t)R_local);
        for (int k = 0; k < T; ++k) lt[(size_t)k] = (float)(1.0 / ladder_betas[k]);
        for (int r = 0; r < R_local; ++r) local[(size_t)r] = lt[(size_t)(((long long)first_replica + r) % T)];
        HIP_TRY(hipMalloc((void **)&p->d_pt_rung, (size_t)total * sizeof(int)));
        HIP_TRY(hipMalloc((void **)&p->d_pt_betas, (size_t)T * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&p->d_pt_energy, (size_t)total * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&p->d_pt_ladder, (size_t)T * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&p->d_pt_stats, 2 * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&p->d_pt_temps, (size_t)R_local * sizeof(float)));
        HIP_TRY(hipMemcpy(p->d_pt_rung, rung.data(), rung.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_pt_betas, ladder_betas, (size_t)T * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_pt_ladder, lt.data(), lt.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_pt_temps, local.data(), local.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(p->d_pt_stats, 0, 2 * sizeof(unsigned long long)));
        return MI_OK;
    });
    if (rc) return rc;
    p->pt_T = T; p->pt_chains = chains; p->pt_lo = (int)first_replica; p->pt_R_local = R_local;
    return MI_OK;
}

static int tempering_exchange_impl(mi_sa_problem *p, uint32_t round, uint64_t seed, const double *all_energies, bool on_device)
{
    if (const int rc_w = settle(p)) return rc_w;
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (p->pt_T == 0) return fail(MI_ESTATE, "mi_sa_tempering_begin has not been called on this problem");
    if (!p->has_run || p->last_R != p->pt_R_local) return fail(MI_ESTATE, "no tempering round has run on this problem");
    const long long total = (long long)p->pt_T * p->pt_chains;
    if (!all_energies && p->pt_R_local != total)
        return fail(MI_EINVAL, "this GPU holds %d of the %lld replicas: the exchange needs all energies", p->pt_R_local, total);
    HIP_TRY(hipSetDevice(p->device));
    const double *en = p->d_energy;              // one GPU owns every replica: the energies never leave HBM
    if (all_energies && on_device) {
        en = all_energies;                       // the all-gather's output buffer, already in HBM
    } else if (all_energies) {
        HIP_TRY(hipMemcpyAsync(p->d_pt_energy, all_energies, (size_t)total * sizeof(double), hipMemcpyHostToDevice, p->stream));
        en = p->d_pt_energy;
    }
    hipLaunchKernelGGL(k_pt_exchange, dim3(p->pt_chains), dim3(256), (size_t)p->pt_T * sizeof(int), p->stream, en,
                       p->d_pt_rung, p->d_pt_betas, p->d_pt_ladder, p->d_pt_temps, p->pt_T, p->pt_lo, p->pt_lo + p->pt_R_local,
                       round, (uint32_t)seed, (uint32_t)(seed >> 32), p->d_pt_stats);
    HIP_TRY(hipGetLastError());
    if (all_energies) HIP_TRY(hipStreamSynchronize(p->stream));      // the caller's buffer may go away
    return MI_OK;
}

int mi_sa_tempering_exchange(mi_sa_problem *p, uint32_t round, uint64_t seed, const double *all_energies)
{
    return tempering_exchange_impl(p, round, seed, all_energies, false);
}

int mi_sa_tempering_exchange_dev(mi_sa_problem *p, uint32_t round, uint64_t seed, const double *d_all_energies)
{
    if (!d_all_energies) return fail(MI_EINVAL, "d_all_energies is NULL");
    return tempering_exchange_impl(p, round, seed, d_all_energies, true);
}

int mi_sa_device_results(mi_sa_problem *p, void **out_d_states, double **out_d_energy, int *out_R)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (const int rc_w = settle(p)) return rc_w;
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (out_d_states) *out_d_states = p->d_states;
    if (out_d_energy) *out_d_energy = p->d_energy;
    if (out_R) *out_R = p->last_R;
    return MI_OK;
}

int mi_sa_tempering_state(mi_sa_problem *p, int32_t *out_rung, uint64_t *out_proposed, uint64_t *out_accepted)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (const int rc_w = settle(p)) return rc_w;
    if (p->pt_T == 0) return fail(MI_ESTATE, "mi_sa_tempering_begin has not been called on this problem");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (out_rung)
        HIP_TRY(hipMemcpy(out_rung, p->d_pt_rung, (size_t)p->pt_T * p->pt_chains * sizeof(int), hipMemcpyDeviceToHost));
    unsigned long long st[2];
    HIP_TRY(hipMemcpy(st, p->d_pt_stats, sizeof st, hipMemcpyDeviceToHost));
    if (out_proposed) *out_proposed = st[0];
    if (out_accepted) *out_accepted = st[1];
    return MI_OK;
}

int mi_sa_anneal_ex(mi_sa_problem *p, int R, uint32_t replica_offset, int num_sweeps,
                    const double *betas, uint64_t seed, const void *init, int resync_interval,
                    uint32_t sweep_offset, uint32_t flags)
{
    return guarded([&]() -> int {
        return anneal_ex_impl(p, R, replica_offset, num_sweeps, betas, seed, init, resync_interval, sweep_offset, flags);
    });
}

int mi_sa_anneal(mi_sa_problem *p, int R, uint32_t replica_offset, int num_sweeps,
                 const double *betas, uint64_t seed, const void *init, int resync_interval)
{
    return mi_sa_anneal_ex(p, R, replica_offset, num_sweeps, betas, seed, init, resync_interval, 0u, 0u);
}

int mi_sa_sync(mi_sa_problem *p)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (const int rc_w = settle(p)) return rc_w;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MI_OK;
}

int mi_sa_last_kernel_ms(mi_sa_problem *p, float *out_ms)
{
    if (!p || !out_ms) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(out_ms, p->ev0, p->ev1));
    return MI_OK;
}

int mi_sa_last_launch_count(mi_sa_problem *p, int *out_launches)
{
    if (!p || !out_launches) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    *out_launches = p->last_launches;
    return MI_OK;
}

int mi_sa_last_kernel_name(mi_sa_problem *p, char *out, int len)
{
    if (!p || !out || len < 1) return fail(MI_EINVAL, "NULL argument");
    if (const int rc_w = settle(p)) return rc_w;
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    snprintf(out, (size_t)len, "%s", p->last_kernel.c_str());
    return MI_OK;
}

int mi_sa_fetch(mi_sa_problem *p, void *out_states, double *out_energy, uint64_t *out_stats)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (const int rc_w = settle(p)) return rc_w;
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (out_states)
        HIP_TRY(hipMemcpy(out_states, p->d_states, (size_t)p->last_R * p->n * p->state_elem, hipMemcpyDeviceToHost));
    if (out_energy)
        HIP_TRY(hipMemcpy(out_energy, p->d_energy, (size_t)p->last_R * sizeof(double), hipMemcpyDeviceToHost));
    if (out_stats) {
        unsigned long long st[4];
        HIP_TRY(hipMemcpy(st, p->d_stats, sizeof st, hipMemcpyDeviceToHost));
        out_stats[0] = st[0]; out_stats[1] = st[1]; out_stats[2] = st[2];
    }
    return MI_OK;
}

int mi_sa_best(mi_sa_problem *p, int *out_index, double *out_energy, uint64_t *out_key, void *out_state)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (const int rc_w = settle(p)) return rc_w;
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    HIP_TRY(hipSetDevice(p->device));
    unsigned long long init_key = ~0ull, key = 0;
    HIP_TRY(hipMemcpyAsync(p->d_stats + 3, &init_key, sizeof init_key, hipMemcpyHostToDevice, p->stream));
    hipLaunchKernelGGL(k_best, dim3(1), dim3(1024), 0, p->stream, p->d_energy, p->last_R, p->last_offset, p->d_stats + 3);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&key, p->d_stats + 3, sizeof key, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    const int idx = (int)((uint32_t)(key & 0xffffffffull) - p->last_offset);
    if (idx < 0 || idx >= p->last_R) return fail(MI_EHIP, "best-of reduction returned an invalid index %d", idx);
    if (out_index) *out_index = idx;
    if (out_key) *out_key = key;
    if (out_energy) HIP_TRY(hipMemcpy(out_energy, p->d_energy + idx, sizeof(double), hipMemcpyDeviceToHost));
    if (out_state)
        HIP_TRY(hipMemcpy(out_state, (const char *)p->d_states + (size_t)idx * p->n * p->state_elem,
                          (size_t)p->n * p->state_elem, hipMemcpyDeviceToHost));
    return MI_OK;
}

// ---- several GPUs from ONE process (callers without a process-per-GPU launcher) -------------------------------
// Replicas are independent chains: problem d (the same model created on device d) runs the contiguous shard d of the
// global replica ids; the anneals are asynchronous on each device's own stream, so the devices run concurrently; the
// best replica is the minimum of the per-device packed keys -- the same reduction distributed.global_best does with one
// RCCL all-reduce when there is one process per GPU.
static void shard_of(int R_total, int d, int ndev, int *lo, int *hi)
{
    const int base = R_total / ndev, rem = R_total % ndev;
    *lo = d * base + (d < rem ? d : rem);
    *hi = *lo + base + (d < rem ? 1 : 0);
}

static int multi_check(mi_sa_problem *const *problems, int ndev)
{
    if (!problems || ndev < 1) return fail(MI_EINVAL, "need ndev >= 1 problem handles");
    for (int d = 0; d < ndev; ++d) {
        if (!problems[d]) return fail(MI_EINVAL, "problem %d is NULL", d);
        if (problems[d]->kind != problems[0]->kind || problems[d]->n != problems[0]->n || problems[d]->K != problems[0]->K)
            return fail(MI_EINVAL, "problem %d is not the model of problem 0 (kind / size differ)", d);
    }
    return MI_OK;
}

int mi_multi_gpu_anneal(mi_sa_problem *const *problems, int ndev, int R_total, uint32_t replica_offset,
                        int num_sweeps, const double *betas, uint64_t seed, int resync_interval)
{
    int rc = multi_check(problems, ndev);
    if (rc) return rc;
    if (R_total < ndev) return fail(MI_EINVAL, "R_total = %d replicas cannot be sharded over %d devices", R_total, ndev);
    for (int d = 0; d < ndev; ++d) {
        int lo, hi;
        shard_of(R_total, d, ndev, &lo, &hi);
        rc = mi_sa_anneal_ex(problems[d], hi - lo, replica_offset + (uint32_t)lo, num_sweeps, betas, seed, nullptr,
                             resync_interval, 0u, 0u);
        if (rc) return rc;
    }
    return MI_OK;
}

int mi_multi_gpu_best(mi_sa_problem *const *problems, int ndev, int *out_owner, uint32_t *out_global_id,
                      double *out_energy, void *out_state)
{
    int rc = multi_check(problems, ndev);
    if (rc) return rc;
    // one process sees every device's exact fp64 minimum: lowest energy, ties to the lowest global id (the devices
    // hold ascending id ranges) -- the record a sorted SampleSet of all replicas puts first
    int owner = -1, best_idx = 0;
    uint64_t best_key = ~0ull;
    double best_e = 0.0;
    for (int d = 0; d < ndev; ++d) {
        int idx = 0;
        uint64_t key = 0;
        double e = 0.0;
        rc = mi_sa_best(problems[d], &idx, &e, &key, nullptr);
        if (rc) return rc;
        if (owner < 0 || e < best_e) { owner = d; best_key = key; best_idx = idx; best_e = e; }
    }
    mi_sa_problem *p = problems[owner];
    HIP_TRY(hipSetDevice(p->device));
    if (out_owner) *out_owner = owner;
    if (out_global_id) *out_global_id = (uint32_t)(best_key & 0xffffffffull);
    if (out_energy) HIP_TRY(hipMemcpy(out_energy, p->d_energy + best_idx, sizeof(double), hipMemcpyDeviceToHost));
    if (out_state)
        HIP_TRY(hipMemcpy(out_state, (const char *)p->d_states + (size_t)best_idx * p->n * p->state_elem,
                          (size_t)p->n * p->state_elem, hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_multi_gpu_fetch(mi_sa_problem *const *problems, int ndev, void *out_states, double *out_energy, uint64_t *out_stats)
{
    int rc = multi_check(problems, ndev);
    if (rc) return rc;
    size_t done = 0;
    uint64_t tot[3] = {0, 0, 0};
    for (int d = 0; d < ndev; ++d) {
        mi_sa_problem *p = problems[d];
        if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on problem %d", d);
        uint64_t st[3] = {0, 0, 0};
        rc = mi_sa_fetch(p, out_states ? (char *)out_states + done * p->n * p->state_elem : nullptr,
                         out_energy ? out_energy + done : nullptr, st);
        if (rc) return rc;
        for (int k = 0; k < 3; ++k) tot[k] += st[k];
        done += (size_t)p->last_R;
    }
    if (out_stats) { out_stats[0] = tot[0]; out_stats[1] = tot[1]; out_stats[2] = tot[2]; }
    return MI_OK;
}

int mi_sa_qubo_dense_f32(const float *Qs, int n, double offset, int R, int num_sweeps,
                         const double *betas, uint64_t seed, const uint8_t *init,
                         uint8_t *out_states, double *out_energy, uint64_t *out_stats, int device)
{
    mi_sa_problem *p = nullptr;
    int rc = mi_sa_problem_create_dense_f32(Qs, n, offset, device, &p);
    if (rc) return rc;
    rc = mi_sa_anneal(p, R, 0, num_sweeps, betas, seed, init, 0);
    if (!rc) rc = mi_sa_fetch(p, out_states, out_energy, out_stats);
    if (!rc && out_stats) out_stats[0] = (uint64_t)R * (uint64_t)num_sweeps * (uint64_t)n;
    mi_sa_problem_destroy(p);
    return rc;
}

int mi_energy_dense_f32_ex(const float *Qs, int n, const uint8_t *X, int R, double offset,
                           double *out_energy, int device, int path, float *out_kernel_ms)
{
    if (!Qs || !X || !out_energy) return fail(MI_EINVAL, "NULL argument");
    if (n < 1 || R < 1) return fail(MI_EINVAL, "n and R must be >= 1");
    if (path < 0 || path > 2) return fail(MI_EINVAL, "path must be 0 (auto), 1 (VALU) or 2 (MFMA)");
    if (path != 1 && (path == 2 || R >= 32)) {
        // the MFMA kernel multiplies only the blocks on and above the diagonal of a SYMMETRIC Qs: a matrix that is not
        // (e.g. an upper-triangular QUBO) goes to the exact path when the choice is the library's, and is refused when
        // the caller asked for the MFMA path by name
        bool symmetric = true;
        for (int i = 0; i < n && symmetric; ++i)
            for (int j = i + 1; j < n; ++j)
                if (Qs[(size_t)i * n + j] != Qs[(size_t)j * n + i]) { symmetric = false; break; }
        if (!symmetric) {
            if (path == 2) return fail(MI_EINVAL, "the MFMA energy path (path = 2) needs a symmetric Qs; use (Q + Q^T) / 2 or path 0 / 1");
            path = 1;
        }
    }
    if (path == 0) path = (R >= 32) ? 2 : 1;      // MFMA only when the batch is a real dense contraction
    int rc = select_device(device);
    if (rc) return rc;
    float *dQ = nullptr; uint8_t *dX = nullptr, *dXt = nullptr; double *dE = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    rc = [&]() -> int {
        // the MFMA kernel reads whole 128 x 128 blocks: rows padded to n_pad floats, n_pad rows, padding zero
        const size_t ldq = path == 2 ? ((size_t)n + 127) / 128 * 128 : (size_t)n;
        const size_t rows = path == 2 ? ldq : (size_t)n;
        HIP_TRY(hipMalloc((void **)&dQ, rows * ldq * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&dX, (size_t)R * n));
        HIP_TRY(hipMalloc((void **)&dE, (size_t)R * sizeof(double)));
        if (path == 2) {
            HIP_TRY(hipMalloc((void **)&dXt, mi_energy_dense_scratch_bytes(n, R)));
            HIP_TRY(hipMemset(dQ, 0, rows * ldq * sizeof(float)));
        }
        HIP_TRY(hipMemcpy2D(dQ, ldq * sizeof(float), Qs, (size_t)n * sizeof(float), (size_t)n * sizeof(float), (size_t)n,
                            hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dX, X, (size_t)R * n, hipMemcpyHostToDevice));
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, 0));
        int r2 = mi_launch_energy_dense(dQ, n, (int)ldq, dX, R, offset, dE, dXt, path, 0);
        if (r2) return r2;
        HIP_TRY(hipEventRecord(e1, 0));
        HIP_TRY(hipEventSynchronize(e1));
        if (out_kernel_ms) HIP_TRY(hipEventElapsedTime(out_kernel_ms, e0, e1));
        HIP_TRY(hipMemcpy(out_energy, dE, (size_t)R * sizeof(double), hipMemcpyDeviceToHost));
        return MI_OK;
    }();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (dQ) (void)hipFree(dQ);
    if (dX) (void)hipFree(dX);
    if (dXt) (void)hipFree(dXt);
    if (dE) (void)hipFree(dE);
    return rc;
}

int mi_energy_dense_f64(const double *Qs, int n, const uint8_t *X, int R, double offset,
                        double *out_energy, int device)
{
    if (!Qs || !X || !out_energy) return fail(MI_EINVAL, "NULL argument");
    if (n < 1 || R < 1) return fail(MI_EINVAL, "n and R must be >= 1");
    int rc = select_device(device);
    if (rc) return rc;
    double *dQ = nullptr, *dE = nullptr; uint8_t *dX = nullptr;
    rc = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&dQ, (size_t)n * n * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&dX, (size_t)R * n));
        HIP_TRY(hipMalloc((void **)&dE, (size_t)R * sizeof(double)));
        HIP_TRY(hipMemcpy(dQ, Qs, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dX, X, (size_t)R * n, hipMemcpyHostToDevice));
        int r2 = mi_launch_energy_dense_f64(dQ, n, dX, R, offset, dE, 0);
        if (r2) return r2;
        HIP_TRY(hipMemcpy(out_energy, dE, (size_t)R * sizeof(double), hipMemcpyDeviceToHost));
        return MI_OK;
    }();
    if (dQ) (void)hipFree(dQ);
    if (dX) (void)hipFree(dX);
    if (dE) (void)hipFree(dE);
    return rc;
}

int mi_energy_dense_f32(const float *Qs, int n, const uint8_t *X, int R, double offset,
                        double *out_energy, int device)
{
    return mi_energy_dense_f32_ex(Qs, n, X, R, offset, out_energy, device, 0, nullptr);
}

}  // extern "C"
