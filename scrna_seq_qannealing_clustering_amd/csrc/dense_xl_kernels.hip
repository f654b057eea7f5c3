// dense_xl_kernels.hip -- K1x: the dense-QUBO chain for models too large for register-per-wave fields
// (4096 < n <= 65536): ONE WORKGROUP of 512 threads per replica, the n cached fields spread over the
// workgroup's registers, Q in HBM.
//
// Same chain as K1 / K1w (oracle 2a), different mapping.  Thread t owns, in every
// 4096-column chunk k, the eight consecutive variables 4096 k + 8 t + c (8 waves of 256 VGPRs hold 65536
// fields): a Q row is read as two 16 B/lane loads per chunk, 32 contiguous bytes per lane.  The sequential sweep is kept
// exactly: in a chunk every thread finds its first accepting variable at or above the cursor, the lowest
// one in the workgroup (ballot inside a wave, one LDS word per wave, ONE barrier per commit with a
// double-buffered exchange) is committed, all 1024 threads add +-row to their fields, and the cursor moves
// past it.
// Bound: HBM.  An accepted flip streams 4 n_pad bytes of Q (213 KB at n = 50 000) and nothing about a
// 10 GB matrix is cacheable across replicas that have drifted apart: this is the kernel whose roofline is
// the HBM read roofline of SURVEY.md section 8d (4n bytes per ACCEPTED update).
#include "mi_sa_device.h"

namespace mi_sa_impl {
namespace {

constexpr int kXlThreads = 512, kXlChunk = 4096, kXlVpt = kXlChunk / kXlThreads;   // 8 variables per thread per chunk

template <int CH>
__global__ void __launch_bounds__(kXlThreads) k_anneal_dense_xl(DenseXlArgs a)
{
    constexpr int W = kXlThreads / 64, V = kXlVpt;               // waves per workgroup, variables per thread per chunk
    __shared__ unsigned int wave_min[2][W];
    __shared__ unsigned int bits_lds[kXlThreads];
    __shared__ double esum[W];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x;
    const uint32_t g = a.replica_offset + (uint32_t)r;
    const int n = a.n;
    const size_t stride = (size_t)CH * kXlChunk;                 // floats per row of Q2 (zero padded)

    float f[CH][V];
    unsigned int xb[CH];                                         // bit c = x of variable 4096 k + V tid + c
    static_for<0, CH>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        xb[k] = 0u;
#pragma unroll
        for (int c = 0; c < V; ++c) {
            const int i = k * kXlChunk + tid * V + c;
            unsigned int bit = 0u;
            if (i < n) bit = a.init ? (a.init[(size_t)r * n + i] ? 1u : 0u)
                                    : (chain_word_dev((uint32_t)i, 0u, g, 1u, a.seed_lo, a.seed_hi) >> 31);
            xb[k] |= bit << c;
        }
    });

    // f (+)= s * Q2[row]: two 16 B/lane loads per chunk
    auto add_row = [&](int row, float sgn) {
        const float *src = a.Q2 + (size_t)row * stride + tid * V;
        static_for<0, CH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int h = 0; h < V / 4; ++h) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(src + k * kXlChunk + 4 * h);
                f[k][4 * h + 0] = __fmaf_rn(sgn, q.x, f[k][4 * h + 0]);
                f[k][4 * h + 1] = __fmaf_rn(sgn, q.y, f[k][4 * h + 1]);
                f[k][4 * h + 2] = __fmaf_rn(sgn, q.z, f[k][4 * h + 2]);
                f[k][4 * h + 3] = __fmaf_rn(sgn, q.w, f[k][4 * h + 3]);
            }
        });
    };

    // f = diag ; then += Q2 row j for every j with x_j = 1, ascending j (the workgroup walks its own bits)
    auto field_init = [&]() {
        static_for<0, CH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int c = 0; c < V; ++c) f[k][c] = a.diag[k * kXlChunk + tid * V + c];
        });
        for (int k = 0; k < CH; ++k) {
            unsigned int mine = 0u;
            static_for<0, CH>([&](auto kc) { if (decltype(kc)::value == k) mine = xb[decltype(kc)::value]; });
            __syncthreads();
            bits_lds[tid] = mine;
            __syncthreads();
            for (int t = 0; t < kXlThreads; ++t) {
                unsigned int m = bits_lds[t];                    // same word for every thread
                while (m) {
                    const int c = __ffs((int)m) - 1;
                    m &= m - 1;
                    add_row(k * kXlChunk + t * V + c, 1.0f);     // fmaf(1, q, f) == f + q
                }
            }
        }
    };

    unsigned long long accepted = 0;
    unsigned int phase = 0;
    int until_resync = a.resync > 0 ? 1 : 0;
    for (int s = 0; s < a.num_sweeps; ++s) {
        bool init_now = (s == 0);
        if (a.resync > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now && s == 0 && a.fields_in) {
            // continue a run of K1g: its cached fields, exactly (a re-evaluation would round differently)
            static_for<0, CH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    const int i = k * kXlChunk + tid * V + c;
                    f[k][c] = i < a.fin_ncols ? a.fields_in[((size_t)(r >> 6) * a.fin_ncols + i) * 64 + (r & 63)] : 0.0f;
                }
            });
        } else if (init_now) {
            field_init();
        }
        const float T = a.temps[a.temps_per_replica ? r : s];
        static_for<0, CH>([&](auto kc) {                         // compile-time chunk index: f[] stays in registers
            constexpr int k = decltype(kc)::value;
            if (k * kXlChunk >= n) return;                       // uniform
            float thr[V];
#pragma unroll
            for (int c = 0; c < V; ++c) {
                const int i = k * kXlChunk + tid * V + c;
                thr[c] = (i < n) ? neglog_u(chain_word_dev((uint32_t)i, (uint32_t)s + a.sweep_offset, g, 0u, a.seed_lo, a.seed_hi)) * T
                                 : -INFINITY;
            }
            int cursor = k * kXlChunk;                           // variables below it are done for this sweep
            while (true) {
                // this thread's first accepting variable at or above the cursor
                const int base = k * kXlChunk + tid * V;
                unsigned int cand = 0xffffffffu;
#pragma unroll
                for (int c = V - 1; c >= 0; --c) {
                    const unsigned int xc = (xb[k] >> c) & 1u;
                    const float dE = xc ? -f[k][c] : f[k][c];
                    if (base + c >= cursor && dE < thr[c]) cand = ((unsigned int)(base + c) << 1) | xc;
                }
                // lowest lane with a candidate holds the wave's lowest variable (indices grow with the thread id)
                const unsigned long long bal = __ballot(cand != 0xffffffffu);
                unsigned int wmin = 0xffffffffu;
                if (bal) wmin = (unsigned int)__builtin_amdgcn_readlane((int)cand, __ffsll((unsigned long long)bal) - 1);
                if (lane == 0) wave_min[phase][wave] = wmin;
                __syncthreads();
                unsigned int best = 0xffffffffu;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const unsigned int v = wave_min[phase][w];
                    best = v < best ? v : best;
                }
                phase ^= 1u;                                     // the other buffer next time: one barrier per commit
                if (best == 0xffffffffu) break;                  // uniform
                const int i = (int)(best >> 1);
                if (i / V == k * kXlThreads + tid) xb[k] ^= 1u << (i % V);
                add_row(i, (best & 1u) ? -1.0f : 1.0f);
                cursor = i + 1;
                ++accepted;
            }
        });
    }

    // states out; energy E = 1/2 sum_i x_i (f_i + diag_i) from the cached fp32 fields, summed in fp64
    double e = 0.0;
    static_for<0, CH>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int c = 0; c < V; ++c) {
            const int i = k * kXlChunk + tid * V + c;
            const unsigned int xc = (xb[k] >> c) & 1u;
            if (i < n) {
                a.states[(size_t)r * n + i] = (uint8_t)xc;
                if (xc) e += 0.5 * ((double)f[k][c] + (double)a.diag[i]);
            }
        }
    });
    e = wave_sum_f64(e);
    __syncthreads();
    if (lane == 0) esum[wave] = e;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < W; ++w) tot += esum[w];
        a.energy[r] = tot + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

template <int CH>
int launch_xl(const DenseXlArgs &a, hipStream_t st)
{
    note_kernel("k_anneal_dense_xl<%d>", CH);
    hipLaunchKernelGGL((k_anneal_dense_xl<CH>), dim3(a.R), dim3(kXlThreads), 0, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

int mi_launch_dense_xl(const DenseXlArgs &a, int chunks, hipStream_t st)
{
    switch (chunks) {
#define MI_XL(N) case N: return launch_xl<N>(a, st);
        MI_XL(2) MI_XL(3) MI_XL(4) MI_XL(5) MI_XL(6) MI_XL(7) MI_XL(8) MI_XL(9) MI_XL(10) MI_XL(11) MI_XL(12)
        MI_XL(13) MI_XL(14) MI_XL(15) MI_XL(16)
#undef MI_XL
    }
    return fail(MI_EUNSUPPORTED, "K1x is built for 2..16 chunks of 4096 variables (got %d)", chunks);
}

}  // namespace mi_sa_impl
