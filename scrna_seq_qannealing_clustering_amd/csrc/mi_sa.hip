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

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mi_sa.h"

namespace mi_sa_impl {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;

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

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(MI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                               \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers: Philox4x32-10, -ln(u)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t PH_M0 = 0xD2511F53u, PH_M1 = 0xCD9E8D57u;
constexpr uint32_t PH_W0 = 0x9E3779B9u, PH_W1 = 0xBB67AE85u;

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(PH_M0, c0), lo0 = PH_M0 * c0;
        const uint32_t hi1 = __umulhi(PH_M1, c2), lo1 = PH_M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PH_W0; k1 += PH_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// -ln(u), u in (0,1] from the top 23 bits of r; every step one IEEE fp32 op or fma (bit-reproducible).
__device__ __forceinline__ float neglog_u(uint32_t r)
{
    const float mm = __uint_as_float(0x3f800000u | (r >> 9));
    const float u = 2.0f - mm;
    const uint32_t ub = __float_as_uint(u);
    int e = (int)(ub >> 23) - 127;
    float m = __uint_as_float((ub & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float t = m - 1.0f;
    float p = -0x1.9f9af6p-4f;
    p = __fmaf_rn(p, t, 0x1.4cd8dcp-3f);
    p = __fmaf_rn(p, t, -0x1.61491cp-3f);
    p = __fmaf_rn(p, t, 0x1.977bcp-3f);
    p = __fmaf_rn(p, t, -0x1.ff611p-3f);
    p = __fmaf_rn(p, t, 0x1.555a22p-2f);
    p = __fmaf_rn(p, t, -0x1.00007cp-1f);
    p = __fmaf_rn(p, t, 0x1.fffffep-1f);
    const float lnm = p * t;
    return __fmaf_rn(-(float)e, 0x1.62e43p-1f, -lnm);
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}


// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N) -- keeps every f[] index a
// constant so the field array is register-allocated at any NT (a pragma-unrolled loop falls back to
// scratch once the body grows past the unroller's budget).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&body)
{
    if constexpr (I < N) {
        body(std::integral_constant<int, I>{});
        static_for<I + 1, N>(body);
    }
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// Sweep pacing (speed only, never correctness)
// ------------------------------------------------------------------------------------------------
// Replicas visit the rows of Q in the same order but accept different flips, so they drift apart and
// their row fetches stop sharing the XCD's 4 MiB L2 (Q is ~30 MB): every fetch then comes from
// Infinity Cache / HBM.  Holding the replicas of ONE XCD together at each sweep boundary keeps them
// inside a window of a few hundred rows, which the L2 holds.  No data passes through this rendezvous:
// results are identical with it on, off, or timing out -- every wait is bounded by a wall-clock
// limit, so a launch whose waves are not all resident only loses time.
//   pace[0]            waves started (launch-wide)
//   pace[1]            pacing disabled (the start rendezvous timed out)
//   pace[2]            sweep waits that hit their time limit (diagnostic)
//   pace[32*(1+x)]     waves living on XCD x          (one 128-byte line per XCD)
//   pace[32*(1+x)+1]   sweep arrivals on XCD x (monotonic)
constexpr int kPaceWords = 32 * 9;
constexpr long long kPaceStartTicks = 400000;   // 4 ms of the 100 MHz realtime clock
constexpr long long kPaceSweepTicks = 200000;   // 2 ms

__device__ __forceinline__ unsigned int pace_load(const unsigned int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// returns the XCD population, or 0 when pacing is off for this launch
__device__ __forceinline__ unsigned int sweep_pace_begin(unsigned int *pace, unsigned int total_waves,
                                                         unsigned int &xcc)
{
    if (!pace) return 0;
    xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // HW_REG_XCC_ID[3:0]
    unsigned int pop = 0;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&pace[32 * (1 + xcc)], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(&pace[0], 1u);
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        bool ok = true;
        while (pace_load(&pace[0]) < total_waves) {
            if (pace_load(&pace[1]) != 0 ||
                (long long)__builtin_amdgcn_s_memrealtime() - t0 > kPaceStartTicks) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        if (!ok) __hip_atomic_store(&pace[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else pop = pace_load(&pace[32 * (1 + xcc)]);
    }
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)pop);
}

__device__ __forceinline__ void sweep_pace_arrive_wait(unsigned int *pace, unsigned int xcc,
                                                       unsigned int pop, unsigned int sweeps_done)
{
    if ((threadIdx.x & 63) == 0) {
        unsigned int *arr = &pace[32 * (1 + xcc) + 1];
        atomicAdd(arr, 1u);
        const unsigned int target = pop * sweeps_done;
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while (pace_load(arr) < target) {
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > kPaceSweepTicks) { atomicAdd(&pace[2], 1u); break; }
            __builtin_amdgcn_s_sleep(64);
        }
    }
    // the other 63 lanes re-converge with lane 0 here (same wave): nothing else to do
}

// ------------------------------------------------------------------------------------------------
// K1: dense binary chain, one wavefront per replica, fields in VGPRs
// ------------------------------------------------------------------------------------------------
struct DenseArgs {
    const float *Qp;        // slot-permuted Q2: row i, float4 index (g*64 + lane) holds columns
                            // 64*(4g+c)+lane, c = 0..3 ; row stride = NT*64 floats; row n = diagonal
    const float *temps;     // num_sweeps floats
    const uint8_t *init;    // nullable, R x n
    uint8_t *states;        // R x n
    double *energy;         // R
    unsigned long long *stats;  // [0] proposals [1] accepted [2] bytes
    unsigned int *pace;     // sweep pacing words (see sweep_pace_*), zeroed per launch; nullable
    double offset;
    int n, R, num_sweeps, resync;
    uint32_t replica_offset, seed_lo, seed_hi;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// f (+)= s * Q2[row].  The row is fetched with NT/4 buffer loads of 16 B/lane (1 KiB per
// wave-instruction, fully coalesced): descriptor in SGPRs, ONE VGPR of addressing (lane*16), the
// wave-uniform row offset in soffset -- flat global loads cost three 64-bit VGPR address pairs here
// and push the kernel into spilling its field registers.
template <int NT>
__device__ __forceinline__ void dense_add_row(float (&f)[NT], __amdgpu_buffer_rsrc_t rsrc, int row,
                                              int lane, float s)
{
    const int voff = lane * 16;
    const int soff = row * (NT * 64 * 4);
    u32x4 q[NT / 4];
#pragma unroll
    for (int g = 0; g < NT / 4; ++g)
        q[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + g * 1024, 0);
#pragma unroll
    for (int g = 0; g < NT / 4; ++g) {
        f[4 * g + 0] = __fmaf_rn(s, __uint_as_float(q[g].x), f[4 * g + 0]);
        f[4 * g + 1] = __fmaf_rn(s, __uint_as_float(q[g].y), f[4 * g + 1]);
        f[4 * g + 2] = __fmaf_rn(s, __uint_as_float(q[g].z), f[4 * g + 2]);
        f[4 * g + 3] = __fmaf_rn(s, __uint_as_float(q[g].w), f[4 * g + 3]);
    }
}

// f = diag ; then add row j for every j with x_j = 1, ascending j.  The diagonal is stored as row n
// of the permuted matrix, so every global access of the kernel goes through dense_add_row
// (0 + 1*d = d exactly).
template <int NT>
__device__ __forceinline__ void dense_field_init(float (&f)[NT], __amdgpu_buffer_rsrc_t rsrc, int n,
                                                 uint64_t xb, int lane)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) f[t] = 0.0f;
#pragma unroll 1
    for (int t = -1; t < NT; ++t) {             // runtime loop: one copy of the row update
        uint64_t m = (t < 0) ? 1ull : __ballot((xb >> t) & 1ull);
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            dense_add_row<NT>(f, rsrc, (t < 0) ? n : t * 64 + l, lane, 1.0f);   // n = index of the diagonal row
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(256, 4) k_anneal_dense(DenseArgs a)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;                       // wave-uniform
    const uint32_t g = a.replica_offset + (uint32_t)r;
    const int n = a.n;

    // whole permuted matrix behind one buffer descriptor built from kernel arguments: rows
    // 0..64*slots-1 (zero rows past n), then the diagonal as one more row
    const int diag_row = ((n + 63) >> 6) * 64;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.Qp), 0, (diag_row + 1) * (NT * 64 * 4), 0x00020000);

    float f[NT];
    uint64_t xb = 0;                            // bit t = x[64 t + lane]

    if (a.init) {
        const uint8_t *src = a.init + (size_t)r * n;
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            const int i = t * 64 + lane;
            if (i < n && src[i]) xb |= (1ull << t);
        }
    } else {
#pragma unroll 1
        for (int g4 = 0; g4 < NT / 4; ++g4) {
            uint32_t w[4];
            philox4x32_10((uint32_t)(g4 * 64 + lane), 0u, g, 1u, a.seed_lo, a.seed_hi, w);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * g4 + c;
                if (t * 64 + lane < n) xb |= ((uint64_t)(w[c] >> 31) << t);
            }
        }
    }

    unsigned int xcc = 0;
    const unsigned int pace_pop = sweep_pace_begin(a.pace, (unsigned int)a.R, xcc);

    unsigned long long accepted = 0;
    int until_resync = a.resync;
    // s == num_sweeps is the epilogue pass: exact fields from the final state, no sweep.
    for (int s = 0; s <= a.num_sweeps; ++s) {
        bool init_now = (s == 0) || (s == a.num_sweeps);
        if (a.resync > 0 && s > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now) dense_field_init<NT>(f, rsrc, diag_row, xb, lane);
        if (s == a.num_sweeps) break;
        // temperature of this sweep as a scalar (SGPR) operand
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[s])));
        uint32_t w[4];
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            // Opaque per-slot copies of n and lane: everything derived from them is then NOT
            // loop-invariant for LICM, which otherwise hoists ~NT masks, NT lane offsets and NT/4
            // Philox blocks out of the sweep loop and makes the kernel spill its field registers.
            int nn = n, ln = lane;
            asm volatile("" : "+s"(nn));
            asm volatile("" : "+v"(ln));
            const int left = nn - t * 64;       // variables remaining from this slot on (scalar)
            if (left > 0) {                     // wave-uniform
                if constexpr ((t & 3) == 0)
                    philox4x32_10((uint32_t)((t >> 2) * 64 + ln), (uint32_t)s, g, 0u, a.seed_lo,
                                  a.seed_hi, w);
                float thr = neglog_u(w[t & 3]) * T;
                if (ln >= left) thr = -INFINITY;
                float sg = ((xb >> t) & 1ull) ? -1.0f : 1.0f;
                uint64_t todo = ~0ull;
                while (true) {
                    const float dE = sg * f[t];
                    const uint64_t m = __ballot(dE < thr) & todo;
                    if (m == 0) break;
                    const int l = __ffsll((unsigned long long)m) - 1;
                    todo = (l == 63) ? 0ull : (~0ull << (l + 1));
                    const float sl = readlane_f(sg, l);
                    if (ln == l) { sg = -sg; xb ^= (1ull << t); }
                    dense_add_row<NT>(f, rsrc, t * 64 + l, ln, sl);
                    ++accepted;
                }
            }
        });
        if (pace_pop) sweep_pace_arrive_wait(a.pace, xcc, pace_pop, (unsigned int)(s + 1));
    }

    // E = 1/2 sum x_i (diag_i + f_i) in fp64 (f is exact for the final state here)
    float dg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dg[t] = 0.0f;
    dense_add_row<NT>(dg, rsrc, diag_row, lane, 1.0f);
    double e = 0.0;
    uint8_t *dst = a.states + (size_t)r * n + lane;
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const bool on = (xb >> t) & 1ull;
        if (t * 64 + lane < n) {
            dst[t * 64] = on ? 1 : 0;
            if (on) e += 0.5 * ((double)dg[t] + (double)f[t]);
        }
    });
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

// ------------------------------------------------------------------------------------------------
// K1w: dense binary chain, one WORKGROUP of 16 wavefronts = 16 replicas sharing Q rows through LDS
// ------------------------------------------------------------------------------------------------
// Same chain as k_anneal_dense (bit-identical results), different data movement.  All replicas visit
// the rows of Q in the same order, so a workgroup streams Q ONCE per sweep through an LDS ring and
// every accepted flip of its 16 replicas reads its row from LDS (ds_read_b128, conflict-free:
// 16 B/lane consecutive) instead of fetching 11 KB from L2 / Infinity Cache per flip.  HBM-side
// traffic drops from (accepted flips x row) to (rows per sweep) per workgroup, i.e. by
// 16 x acceptance rate, and no longer depends on the acceptance rate at all.
//   ring: U units of GR rows (row = NT*256 bytes, slot-permuted like the global matrix), filled by
//         LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per wave-instruction, no VGPRs), unit u+U-1
//         issued when unit u starts, retired with a COUNTED s_waitcnt vmcnt + raw s_barrier so
//         (U-2) units stay in flight across every barrier.
//   lockstep: the 16 waves rendezvous once per unit (GR rows); inside a unit each wave runs its own
//         accept/commit loop on the unit's rows.
constexpr int kWgWaves = 16;
constexpr int kLdsBytes = 160 * 1024;

template <int NT, int GR>
struct WgCfg {
    static constexpr int ROWB = NT * 256;
    static constexpr int UNITB = GR * ROWB;
    static constexpr int G = NT / 4;                    // 1 KiB pieces per row (<= 16)
    static constexpr int Ufit = kLdsBytes / UNITB;
    static constexpr int Ucap = 2 + 60 / GR;            // keeps (U-2)*GR within the 6-bit vmcnt
    static constexpr int U = Ufit < Ucap ? Ufit : Ucap;
    static constexpr bool ok = U >= 3 && (64 % GR) == 0;
};

// f (+)= s * row, the row read from the LDS ring (ds_read_b128, 16 B/lane consecutive: conflict-free).
// Done in two halves with a scheduling fence between them: LDS latency is short, and holding the
// whole row in registers at once (NT more VGPRs) is what made this kernel spill.
template <int NT>
__device__ __forceinline__ void dense_add_row_lds(float (&f)[NT], const char *row, int lane, float s)
{
    constexpr int G = NT / 4, H = (G + 1) / 2;
    const char *p = row + lane * 16;
    {
        f32x4 q[H];
#pragma unroll
        for (int g = 0; g < H; ++g) q[g] = *reinterpret_cast<const f32x4 *>(p + g * 1024);
#pragma unroll
        for (int g = 0; g < H; ++g) {
            f[4 * g + 0] = __fmaf_rn(s, q[g].x, f[4 * g + 0]);
            f[4 * g + 1] = __fmaf_rn(s, q[g].y, f[4 * g + 1]);
            f[4 * g + 2] = __fmaf_rn(s, q[g].z, f[4 * g + 2]);
            f[4 * g + 3] = __fmaf_rn(s, q[g].w, f[4 * g + 3]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G > H) {
        f32x4 q[G - H];
#pragma unroll
        for (int g = H; g < G; ++g) q[g - H] = *reinterpret_cast<const f32x4 *>(p + g * 1024);
#pragma unroll
        for (int g = H; g < G; ++g) {
            f[4 * g + 0] = __fmaf_rn(s, q[g - H].x, f[4 * g + 0]);
            f[4 * g + 1] = __fmaf_rn(s, q[g - H].y, f[4 * g + 1]);
            f[4 * g + 2] = __fmaf_rn(s, q[g - H].z, f[4 * g + 2]);
            f[4 * g + 3] = __fmaf_rn(s, q[g - H].w, f[4 * g + 3]);
        }
    }
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B from (buffer base + voff + soff) to lds_dst + lane*16
// (buffer_load_dwordx4 ... lds).  Kept in a non-template __device__ function: inside a kernel TEMPLATE
// the builtin makes hipcc silently drop the kernel's host-side launch stub (undefined symbol at load).
__device__ __forceinline__ void lds_dma_16(__amdgpu_buffer_rsrc_t rsrc, char *lds_dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 16,
                                             voff, soff, 0, 0);
}

template <int NT, int GR>
__global__ void __launch_bounds__(1024, 4) k_anneal_dense_wg(DenseArgs a)
{
    using C = WgCfg<NT, GR>;
    __shared__ __attribute__((aligned(16))) char ring[C::U * C::UNITB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x * kWgWaves + wave;
    const bool active = r < a.R;                 // idle waves still take part in DMA and barriers
    const uint32_t g = a.replica_offset + (uint32_t)r;
    const int n = a.n;
    const int slots_used = (n + 63) >> 6;
    const int units_per_sweep = slots_used * (64 / GR);
    const long long total_units = (long long)a.num_sweeps * units_per_sweep;

    // rows 0..64*slots_used-1 (zero rows past n) + the diagonal row at index 64*slots_used
    const int diag_row = slots_used * 64;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.Qp), 0, (diag_row + 1) * C::ROWB, 0x00020000);

    float f[NT];
    uint64_t xb = 0;
    if (active) {
        if (a.init) {
            const uint8_t *src = a.init + (size_t)r * n;
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                const int i = t * 64 + lane;
                if (i < n && src[i]) xb |= (1ull << t);
            }
        } else {
#pragma unroll 1
            for (int g4 = 0; g4 < NT / 4; ++g4) {
                uint32_t w[4];
                philox4x32_10((uint32_t)(g4 * 64 + lane), 0u, g, 1u, a.seed_lo, a.seed_hi, w);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int t = 4 * g4 + c;
                    if (t * 64 + lane < n) xb |= ((uint64_t)(w[c] >> 31) << t);
                }
            }
        }
    }

    // sweep pacing across the workgroups of one XCD (wave 0 of each workgroup takes part; the other
    // waves are held by the next unit barrier): keeps the 32 rings of an XCD within the L2 window
    unsigned int xcc = 0;
    unsigned int pace_pop = 0;
    if (wave == 0) pace_pop = sweep_pace_begin(a.pace, gridDim.x, xcc);

    // ---- ring bookkeeping (all wave-uniform) ----
    long long issued = 0;                        // units whose DMA has been issued
    int issue_row = 0;                           // first row of the next unit to issue
    int issue_slot = 0;                          // ring slot of the next unit to issue
    int cur_slot = 0;                            // ring slot of the unit being processed
    long long processed = 0;                     // units fully processed
    auto issue_unit = [&]() {
        if (issued < total_units) {
            if (wave < C::G) {
#pragma unroll
                for (int k = 0; k < GR; ++k)
                    lds_dma_16(rsrc, ring + issue_slot * C::UNITB + k * C::ROWB + wave * 1024, lane * 16,
                               (issue_row + k) * C::ROWB + wave * 1024);
            }
            ++issued;
            issue_row += GR;
            if (issue_row >= units_per_sweep * GR) issue_row = 0;
            issue_slot = (issue_slot + 1 == C::U) ? 0 : issue_slot + 1;
        }
    };

    unsigned long long accepted = 0;
    int until_resync = a.resync;
    for (int s = 0; s <= a.num_sweeps; ++s) {
        bool init_now = (s == 0) || (s == a.num_sweeps);
        if (a.resync > 0 && s > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now && active) dense_field_init<NT>(f, rsrc, diag_row, xb, lane);
        if (s == a.num_sweeps) break;
        if (s == 0) {
            // everything above used ordinary loads; from here on only LDS-DMA is in the VM queue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int u = 0; u < C::U - 1; ++u) issue_unit();
        }
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[s])));
        uint32_t w[4];
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            int nn = n, ln = lane;
            asm volatile("" : "+s"(nn));
            asm volatile("" : "+v"(ln));
            const int left = nn - t * 64;
            if (left > 0) {                     // wave-uniform, identical in every wave of the block
                if constexpr ((t & 3) == 0)
                    philox4x32_10((uint32_t)((t >> 2) * 64 + ln), (uint32_t)s, g, 0u, a.seed_lo,
                                  a.seed_hi, w);
                float thr = neglog_u(w[t & 3]) * T;
                if (ln >= left || !active) thr = -INFINITY;
                float sg = ((xb >> t) & 1ull) ? -1.0f : 1.0f;
                uint64_t todo = ~0ull;
#pragma unroll 1
                for (int j = 0; j < 64 / GR; ++j) {
                    // retire unit (this wave's pieces), rendezvous, refill the slot just vacated
                    // counted wait: (U-2) younger units stay in flight -- valid only while that many
                    // younger units HAVE been issued; at the tail of the run drain everything
                    if (issued - processed - 1 >= C::U - 2) {
                        if (wave < C::G)
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C::U - 2) * GR) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    issue_unit();
                    const uint64_t umask = (GR == 64) ? ~0ull : (((1ull << GR) - 1ull) << (j * GR));
                    const char *unit = ring + cur_slot * C::UNITB;
                    while (true) {
                        const float dE = sg * f[t];
                        const uint64_t m = __ballot(dE < thr) & todo & umask;
                        if (m == 0) break;
                        const int l = __ffsll((unsigned long long)m) - 1;
                        todo = (l == 63) ? 0ull : (~0ull << (l + 1));
                        const float sl = readlane_f(sg, l);
                        if (ln == l) { sg = -sg; xb ^= (1ull << t); }
                        dense_add_row_lds<NT>(f, unit + (l - j * GR) * C::ROWB, ln, sl);
                        ++accepted;
                    }
                    cur_slot = (cur_slot + 1 == C::U) ? 0 : cur_slot + 1;
                    ++processed;
                }
            }
        });
        if (pace_pop) sweep_pace_arrive_wait(a.pace, xcc, pace_pop, (unsigned int)(s + 1));
    }

    if (!active) return;
    float dg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dg[t] = 0.0f;
    dense_add_row<NT>(dg, rsrc, diag_row, lane, 1.0f);
    double e = 0.0;
    uint8_t *dst = a.states + (size_t)r * n + lane;
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const bool on = (xb >> t) & 1ull;
        if (t * 64 + lane < n) {
            dst[t * 64] = on ? 1 : 0;
            if (on) e += 0.5 * ((double)dg[t] + (double)f[t]);
        }
    });
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

// ------------------------------------------------------------------------------------------------
// K4 (VALU form): energies of arbitrary states, one wavefront per state
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_energy_dense_valu(const float *__restrict__ Qs, int n,
                                                           const uint8_t *__restrict__ X, int R,
                                                           double offset, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const uint8_t *x = X + (size_t)r * n;
    double e = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!x[i]) continue;                    // wave-uniform (same address for all lanes)
        const float *row = Qs + (size_t)i * n;
        float acc = 0.0f;
        for (int j = lane; j < n; j += 64)
            if (x[j]) acc += row[j];
        e += (double)acc;
    }
    e = wave_sum_f64(e);
    if (lane == 0) out[r] = e + offset;
}

// ------------------------------------------------------------------------------------------------
// K5: best-of-replicas: packed (sortable(float E) << 32 | global id) minimum
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sortable_f32(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void __launch_bounds__(256) k_best(const double *__restrict__ energy, int R,
                                              uint32_t replica_offset,
                                              unsigned long long *__restrict__ out_key)
{
    unsigned long long best = ~0ull;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const unsigned long long key =
            ((unsigned long long)sortable_f32((float)energy[r]) << 32) |
            (unsigned long long)(replica_offset + (uint32_t)r);
        best = key < best ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0) atomicMin(out_key, best);
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
    float *d_Qs = nullptr;       // plain row-major copy (energy kernel), allocated lazily
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
    int opt_variant = 0;                     // 0 auto, 1 wave-per-replica (K1), 2 workgroup/LDS ring (K1w)
    int opt_unit_rows = 0;                   // K1w ring unit (rows per rendezvous): 0 auto, 1/2/4
    int resident_waves = 0;                  // co-resident wavefronts of the anneal kernel on this device
    size_t state_elem = 1;
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

constexpr int kMaxChunks = 64;

// Launches the anneal in chunks of at most `resident` replicas (= wavefronts), so that every wave of
// a launch is co-resident and the sweep pacing rendezvous can complete; chunks run back to back on
// the stream.  Each chunk gets its own zeroed pacing words.
template <int NT>
int launch_dense(mi_sa_problem *p, DenseArgs a, hipStream_t st)
{
    if (p->resident_waves == 0) {
        int blocks_per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_anneal_dense<NT>, 256, 0));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device));
        p->resident_waves = blocks_per_cu * cus * 4;
        if (p->resident_waves < 4) return fail(MI_EHIP, "anneal kernel cannot be resident (occupancy 0)");
    }
    const int total = a.R;
    const uint32_t base_offset = a.replica_offset;
    const uint8_t *init0 = a.init;
    uint8_t *states0 = a.states;
    double *energy0 = a.energy;
    int chunk = p->resident_waves;
    if ((total + chunk - 1) / chunk > kMaxChunks) chunk = (total + kMaxChunks - 1) / kMaxChunks;
    const bool pace = p->opt_pace && a.num_sweeps > 1;
    if (pace)
        HIP_TRY(hipMemsetAsync(p->d_pace, 0, kMaxChunks * kPaceWords * sizeof(unsigned int), st));
    int c = 0;
    for (int lo = 0; lo < total; lo += chunk, ++c) {
        const int cnt = total - lo < chunk ? total - lo : chunk;
        a.R = cnt;
        a.replica_offset = base_offset + (uint32_t)lo;
        a.init = init0 ? init0 + (size_t)lo * a.n : nullptr;
        a.states = states0 + (size_t)lo * a.n;
        a.energy = energy0 + lo;
        a.pace = (pace && cnt <= p->resident_waves) ? p->d_pace + (size_t)c * kPaceWords : nullptr;
        hipLaunchKernelGGL(k_anneal_dense<NT>, dim3((cnt + 3) / 4), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
    }
    return MI_OK;
}

template <int NT, int GR>
int launch_dense_wg(mi_sa_problem *p, DenseArgs a, hipStream_t st)
{
    if constexpr (WgCfg<NT, GR>::ok) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device));
        const int blocks = (a.R + kWgWaves - 1) / kWgWaves;
        a.pace = nullptr;
        if (p->opt_pace && a.num_sweeps > 1 && blocks <= cus) {     // one 160 KB workgroup per CU
            HIP_TRY(hipMemsetAsync(p->d_pace, 0, kPaceWords * sizeof(unsigned int), st));
            a.pace = p->d_pace;
        }
        hipLaunchKernelGGL((k_anneal_dense_wg<NT, GR>), dim3((a.R + kWgWaves - 1) / kWgWaves), dim3(1024), 0, st, a);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    } else {
        return fail(MI_EUNSUPPORTED, "LDS ring does not fit for NT=%d unit_rows=%d", NT, GR);
    }
}

template <int NT>
int launch_dense_any(mi_sa_problem *p, const DenseArgs &a, hipStream_t st)
{
    int variant = p->opt_variant;
    if (variant == 0) variant = (a.R >= 2 * kWgWaves && a.num_sweeps > 0) ? 2 : 1;
    if (variant == 2) {
        int gr = p->opt_unit_rows ? p->opt_unit_rows : 2;
        if (gr == 4 && !WgCfg<NT, 4>::ok) gr = 2;
        if (gr == 2 && !WgCfg<NT, 2>::ok) gr = 1;
        switch (gr) {
            case 1: return launch_dense_wg<NT, 1>(p, a, st);
            case 2: return launch_dense_wg<NT, 2>(p, a, st);
            case 4: return launch_dense_wg<NT, 4>(p, a, st);
        }
    }
    return launch_dense<NT>(p, a, st);
}

int dispatch_dense(mi_sa_problem *p, const DenseArgs &a, hipStream_t st)
{
    switch (p->NT) {
#define MI_CASE(N) case N: return launch_dense_any<N>(p, a, st);
#ifdef MI_SA_DEV_NT   /* development builds: only NT=4 and one large size, to cut compile time */
        MI_CASE(4) MI_CASE(MI_SA_DEV_NT)
#else
        MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16) MI_CASE(20) MI_CASE(24) MI_CASE(28)
        MI_CASE(32) MI_CASE(36) MI_CASE(40) MI_CASE(44) MI_CASE(48) MI_CASE(52) MI_CASE(56)
        MI_CASE(60) MI_CASE(64)
#endif
#undef MI_CASE
    }
    return fail(MI_EUNSUPPORTED, "dense kernel not built for NT=%d", p->NT);
}

constexpr int kMaxDenseN = 64 * 64;

}  // namespace

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
    HIP_TRY(hipMalloc((void **)&p->d_stats, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(p->d_stats, 0, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void **)&p->d_pace, kMaxChunks * kPaceWords * sizeof(unsigned int)));
    return MI_OK;
}

int mi_sa_problem_create_dense_f32(const float *Qs, int n, double offset, int device,
                                   mi_sa_problem **out)
{
    if (!Qs || !out) return fail(MI_EINVAL, "NULL argument");
    if (n < 1) return fail(MI_EINVAL, "n must be >= 1 (got %d)", n);
    if (n > kMaxDenseN)
        return fail(MI_EUNSUPPORTED, "dense register-resident kernel supports n <= %d (got %d)", kMaxDenseN, n);
    int rc = select_device(device);
    if (rc) return rc;
    mi_sa_problem *p = new (std::nothrow) mi_sa_problem();
    if (!p) return fail(MI_ENOMEM, "out of host memory");
    p->kind = MI_KIND_DENSE; p->n = n; p->offset = offset; p->state_elem = 1;
    rc = problem_common_init(p, device);
    if (rc) { mi_sa_problem_destroy(p); return rc; }
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
    rc = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&p->d_Qp, hp.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(p->d_Qp, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice));
        return MI_OK;
    }();
    if (rc) { mi_sa_problem_destroy(p); return rc; }
    *out = p;
    return MI_OK;
}

int mi_sa_problem_create_csr_rank1_f32(const int32_t *, const int32_t *, const float *, const float *,
                                       float, int, double, int, mi_sa_problem **)
{
    return fail(MI_EUNSUPPORTED, "csr_rank1 kernel not built yet");
}

int mi_sa_problem_create_potts_csr_f32(const int32_t *, const int32_t *, const float *, float, int, int,
                                       double, int, mi_sa_problem **)
{
    return fail(MI_EUNSUPPORTED, "potts kernel not built yet");
}

int mi_sa_problem_destroy(mi_sa_problem *p)
{
    if (!p) return MI_OK;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void *bufs[] = {p->d_pace, p->d_Qp, p->d_Qs, p->d_temps, p->d_init, p->d_states, p->d_energy, p->d_stats};
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

int mi_sa_debug_pace(mi_sa_problem *p, unsigned int *out, int words)
{
    if (!p || !out) return fail(MI_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (words > kPaceWords) words = kPaceWords;
    HIP_TRY(hipMemcpy(out, p->d_pace, (size_t)words * sizeof(unsigned int), hipMemcpyDeviceToHost));
    return MI_OK;
}

int mi_sa_set_option(mi_sa_problem *p, const char *key, long value)
{
    if (!p || !key) return fail(MI_EINVAL, "NULL argument");
    if (!strcmp(key, "pace")) { p->opt_pace = value != 0; return MI_OK; }
    if (!strcmp(key, "variant") && value >= 0 && value <= 2) { p->opt_variant = (int)value; return MI_OK; }
    if (!strcmp(key, "unit_rows") && (value == 0 || value == 1 || value == 2 || value == 4)) { p->opt_unit_rows = (int)value; return MI_OK; }
    return fail(MI_EINVAL, "unknown option '%s'", key);
}

int mi_sa_anneal(mi_sa_problem *p, int R, uint32_t replica_offset, int num_sweeps,
                 const double *betas, uint64_t seed, const void *init, int resync_interval)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    if (R < 1) return fail(MI_EINVAL, "R must be >= 1 (got %d)", R);
    if (num_sweeps < 0) return fail(MI_EINVAL, "num_sweeps must be >= 0");
    if (num_sweeps > 0 && !betas) return fail(MI_EINVAL, "betas is NULL");
    if (resync_interval < 0) return fail(MI_EINVAL, "resync_interval must be >= 0");
    for (int s = 0; s < num_sweeps; ++s)
        if (!(betas[s] > 0.0) || !std::isfinite(betas[s]))
            return fail(MI_EINVAL, "betas[%d] = %g is not a positive finite number", s, betas[s]);
    HIP_TRY(hipSetDevice(p->device));
    int rc = ensure_run_buffers(p, R, num_sweeps, init != nullptr);
    if (rc) return rc;
    std::vector<float> temps((size_t)(num_sweeps > 0 ? num_sweeps : 1), 1.0f);
    for (int s = 0; s < num_sweeps; ++s) temps[s] = (float)(1.0 / betas[s]);
    // pageable-host async copies are staged synchronously by the runtime: the vector may go away
    HIP_TRY(hipMemcpyAsync(p->d_temps, temps.data(), temps.size() * sizeof(float), hipMemcpyHostToDevice, p->stream));
    if (init)
        HIP_TRY(hipMemcpyAsync(p->d_init, init, (size_t)R * p->n * p->state_elem, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemsetAsync(p->d_stats, 0, 4 * sizeof(unsigned long long), p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));   // inputs resident before the timed region

    if (p->kind == MI_KIND_DENSE) {
        DenseArgs a;
        a.Qp = p->d_Qp; a.temps = p->d_temps;
        a.init = init ? (const uint8_t *)p->d_init : nullptr;
        a.states = (uint8_t *)p->d_states; a.energy = p->d_energy; a.stats = p->d_stats; a.pace = nullptr;
        a.offset = p->offset; a.n = p->n; a.R = R; a.num_sweeps = num_sweeps; a.resync = resync_interval;
        a.replica_offset = replica_offset; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        HIP_TRY(hipEventRecord(p->ev0, p->stream));
        rc = dispatch_dense(p, a, p->stream);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(p->ev1, p->stream));
    } else {
        return fail(MI_EUNSUPPORTED, "kind %d not built yet", p->kind);
    }
    p->last_R = R; p->last_offset = replica_offset; p->has_run = true;
    return MI_OK;
}

int mi_sa_sync(mi_sa_problem *p)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MI_OK;
}

int mi_sa_last_kernel_ms(mi_sa_problem *p, float *out_ms)
{
    if (!p || !out_ms) return fail(MI_EINVAL, "NULL argument");
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(out_ms, p->ev0, p->ev1));
    return MI_OK;
}

int mi_sa_fetch(mi_sa_problem *p, void *out_states, double *out_energy, uint64_t *out_stats)
{
    if (!p) return fail(MI_EINVAL, "NULL problem");
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
    if (!p->has_run) return fail(MI_ESTATE, "no anneal has been run on this problem");
    HIP_TRY(hipSetDevice(p->device));
    unsigned long long init_key = ~0ull, key = 0;
    HIP_TRY(hipMemcpyAsync(p->d_stats + 3, &init_key, sizeof init_key, hipMemcpyHostToDevice, p->stream));
    const int blocks = (p->last_R + 255) / 256 < 1024 ? (p->last_R + 255) / 256 : 1024;
    hipLaunchKernelGGL(k_best, dim3(blocks), dim3(256), 0, p->stream, p->d_energy, p->last_R, p->last_offset, p->d_stats + 3);
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

int mi_energy_dense_f32(const float *Qs, int n, const uint8_t *X, int R, double offset,
                        double *out_energy, int device)
{
    if (!Qs || !X || !out_energy) return fail(MI_EINVAL, "NULL argument");
    if (n < 1 || R < 1) return fail(MI_EINVAL, "n and R must be >= 1");
    int rc = select_device(device);
    if (rc) return rc;
    float *dQ = nullptr; uint8_t *dX = nullptr; double *dE = nullptr;
    rc = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&dQ, (size_t)n * n * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&dX, (size_t)R * n));
        HIP_TRY(hipMalloc((void **)&dE, (size_t)R * sizeof(double)));
        HIP_TRY(hipMemcpy(dQ, Qs, (size_t)n * n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dX, X, (size_t)R * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_energy_dense_valu, dim3((R + 3) / 4), dim3(256), 0, 0, dQ, n, dX, R, offset, dE);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out_energy, dE, (size_t)R * sizeof(double), hipMemcpyDeviceToHost));
        return MI_OK;
    }();
    if (dQ) (void)hipFree(dQ);
    if (dX) (void)hipFree(dX);
    if (dE) (void)hipFree(dE);
    return rc;
}

}  // extern "C"
