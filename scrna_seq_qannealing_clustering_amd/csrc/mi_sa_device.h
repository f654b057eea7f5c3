// mi_sa_device.h -- shared device-side pieces of the MI355X annealing engine (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <exception>
#include <new>
#include <type_traits>

#include "../../include/mi_sa.h"

namespace mi_sa_impl {

int fail(int code, const char *fmt, ...);
void note_kernel(const char *fmt, ...);      // the launchers record which kernel serves the running anneal (mi_sa_last_kernel_name)

// No C++ exception may cross the C ABI: every extern "C" entry that allocates host memory runs its body through
// guarded(), which turns std::bad_alloc (and anything else) into an MI_E* code + mi_last_error() text.
template <typename F>
int guarded(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(MI_ENOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return fail(MI_EINVAL, "unexpected C++ exception: %s", e.what());
    } catch (...) {
        return fail(MI_EINVAL, "unexpected C++ exception");
    }
}

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ::mi_sa_impl::fail(MI_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                                      __FILE__, __LINE__);                                       \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers: Philox4x32-10, -ln(u)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t PH_M0 = 0xD2511F53u, PH_M1 = 0xCD9E8D57u;
constexpr uint32_t PH_W0 = 0x9E3779B9u, PH_W1 = 0xBB67AE85u;

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
    // both halves of each 32 x 32 product from ONE v_mad_u64_u32 (3.3 issue cycles against 2 x 3.1 for
    // v_mul_hi_u32 + v_mul_lo_u32, scripts/ubench_valu.hip)
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PH_M0 * c0, p1 = (uint64_t)PH_M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += PH_W0; k1 += PH_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// -ln(u), u in (0,1] from the top 23 bits of r; every step one IEEE fp32 op or fma (bit-reproducible).
// Range reduction without a compare/select pair: adding C = 2^23 - 0x3504f4 to the bits of u carries into the
// exponent field exactly when the mantissa exceeds that of 1.41421356f, which is the oracle's
// `if (m > 1.41421356f) { m *= 0.5f; e += 1; }` (both forms are exact; checked over all 2^23 inputs by
// tests/test_oracle_kat.py::test_neglog_reduction_forms_agree on the CPU restatement of this sequence).
__device__ __forceinline__ float neglog_u(uint32_t r)
{
    const float mm = __uint_as_float(0x3f800000u | (r >> 9));
    const float u = 2.0f - mm;
    const uint32_t ub = __float_as_uint(u);
    const int neg_e = 127 - (int)((ub + 0x004afb0cu) >> 23);           // -(e), e as the oracle counts it
    const float m = __uint_as_float(ub + ((uint32_t)neg_e << 23));     // u * 2^(-e) in [sqrt(1/2), sqrt(2))
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
    return __fmaf_rn((float)neg_e, 0x1.62e43p-1f, -lnm);
}

// neglog_u for TWO random words at once (the two replicas a wavefront of the pair kernel carries): every fp32 step as
// ONE packed instruction (v_pk_add / v_pk_mul / v_pk_fma_f32 are IEEE per component), the integer steps per word.  The
// range reduction in the form that packs: with t2 = the exponent field of (bits(u) + C) in place (E << 23),
//     m = u * 2^(127 - E)          as a multiply by the float whose bits are 0x7f000000 - t2 (exact: a power of two),
//     (float)(127 - E)             as fma((float)(int)t2, -2^-23, 127) (E * 2^23 converts exactly, the fma is exact),
// the same m and the same exponent the carry form above produces, so the same bits come out (all 2^23 inputs:
// oracle/sa_oracle.c orc_neglog_u_scaled, tests/test_oracle_kat.py::test_neglog_reduction_forms_agree).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t neglog_u2(uint32_t ra, uint32_t rb)
{
    const f32x2_t mm = {__uint_as_float(__builtin_amdgcn_alignbit(0x7fu, ra, 9)),       // 0x3f800000 | (r >> 9)
                        __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, rb, 9))};
    const f32x2_t u = f32x2_t{2.0f, 2.0f} - mm;
    const uint32_t ta = (__float_as_uint(u.x) + 0x004afb0cu) & 0x7f800000u;
    const uint32_t tb = (__float_as_uint(u.y) + 0x004afb0cu) & 0x7f800000u;
    const f32x2_t scale = {__uint_as_float(0x7f000000u - ta), __uint_as_float(0x7f000000u - tb)};
    const f32x2_t m = u * scale;
    const f32x2_t ef = {(float)(int)ta, (float)(int)tb};
    const f32x2_t neg_e = __builtin_elementwise_fma(ef, f32x2_t{-0x1p-23f, -0x1p-23f}, f32x2_t{127.0f, 127.0f});
    const f32x2_t t = m - f32x2_t{1.0f, 1.0f};
    f32x2_t p = {-0x1.9f9af6p-4f, -0x1.9f9af6p-4f};
    p = __builtin_elementwise_fma(p, t, f32x2_t{0x1.4cd8dcp-3f, 0x1.4cd8dcp-3f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{-0x1.61491cp-3f, -0x1.61491cp-3f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{0x1.977bcp-3f, 0x1.977bcp-3f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{-0x1.ff611p-3f, -0x1.ff611p-3f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{0x1.555a22p-2f, 0x1.555a22p-2f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{-0x1.00007cp-1f, -0x1.00007cp-1f});
    p = __builtin_elementwise_fma(p, t, f32x2_t{0x1.fffffep-1f, 0x1.fffffep-1f});
    const f32x2_t lnm = p * t;
    return __builtin_elementwise_fma(neg_e, f32x2_t{0x1.62e43p-1f, 0x1.62e43p-1f}, -lnm);
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

// wave-uniform read of a pacing word (every lane loads the same address; readfirstlane makes the
// uniformity visible to the compiler -- a lane-0-only spin loop inside the sweep loop makes hipcc treat
// the kernel's scalar bookkeeping as divergent and move it to VGPRs)
__device__ __forceinline__ unsigned int pace_load(const unsigned int *p)
{
    return (unsigned int)__builtin_amdgcn_readfirstlane(
        (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ long long pace_clock()
{
    return (long long)__builtin_amdgcn_s_memrealtime();
}

// returns the XCD population, or 0 when pacing is off for this launch
__device__ __forceinline__ unsigned int sweep_pace_begin(unsigned int *pace, unsigned int total_waves,
                                                         unsigned int &xcc)
{
    if (!pace) return 0;
    xcc = (unsigned int)__builtin_amdgcn_readfirstlane(
              (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11))) & 7u;   // HW_REG_XCC_ID[3:0]
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&pace[32 * (1 + xcc)], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(&pace[0], 1u);
    }
    const long long t0 = pace_clock();
    bool ok = true;
    while (pace_load(&pace[0]) < total_waves) {
        if (pace_load(&pace[1]) != 0 || pace_clock() - t0 > kPaceStartTicks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(32);
    }
    if (!ok) {
        if ((threadIdx.x & 63) == 0)
            __hip_atomic_store(&pace[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    return pace_load(&pace[32 * (1 + xcc)]);
}

// every participant ARRIVES every sweep (keeps the counter meaningful); only those that are about to
// stream Q again also WAIT for the others
__device__ __forceinline__ void sweep_pace_arrive_wait(unsigned int *pace, unsigned int xcc,
                                                       unsigned int pop, unsigned int sweeps_done,
                                                       bool wait = true)
{
    unsigned int *arr = &pace[32 * (1 + xcc) + 1];
    if ((threadIdx.x & 63) == 0) atomicAdd(arr, 1u);
    if (!wait) return;
    const unsigned int target = pop * sweeps_done;
    const long long t0 = pace_clock();
    while (pace_load(arr) < target) {
        if (pace_clock() - t0 > kPaceSweepTicks) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&pace[2], 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

// ------------------------------------------------------------------------------------------------
// K1: dense binary chain, one wavefront per replica, fields in VGPRs
// ------------------------------------------------------------------------------------------------
struct DenseArgs {
    const float *Qp;        // slot-permuted Q2: row i, float4 index (g*64 + lane) holds columns
                            // 64*(4g+c)+lane, c = 0..3 ; row stride = NT*64 floats; row n = diagonal
    const float *Qm;        // K1m: plain row-major Q2 (zero diagonal), 64*NT floats per row, rows 0..64*NT-1
                            // (zero past n) and the diagonal as row 64*NT; nullptr when not uploaded
    const float *temps;     // num_sweeps floats
    const uint8_t *init;    // nullable, R x n
    uint8_t *states;        // R x n
    double *energy;         // R
    unsigned long long *stats;  // [0] proposals [1] accepted [2] bytes
    unsigned int *pace;     // sweep pacing words (see sweep_pace_*), zeroed per launch; nullable
    double offset;
    int n, R, num_sweeps, resync;
    uint32_t replica_offset, seed_lo, seed_hi;
    uint32_t sweep_offset;  // added to the sweep index in the RNG counter (continuation of an earlier run)
    int temps_per_replica;  // 0: temps[s] per sweep; 1: temps[r], one constant temperature per replica
    float *fields;          // cached local fields between launches, canonical [R][64*NT] (nullable)
    unsigned int *ctrl;     // kernel-scheduling words (see dense_mfma / launch_dense_chunked), nullable
    int flags;              // kDenseFieldsIn | kDenseFieldsOut | kDenseNoEnergy
    int resync_first;       // sweeps until the first field re-synchronisation of this launch (resync > 0)
    int my_mode;            // mode value for which this kernel serves its chunk (otherwise it exits at once)
    int chunk_index;        // position of this launch in its run: its mode word is ctrl[kCtrlModes + chunk_index]
    unsigned int mode_up_flips;   // launch-wide accepted flips at or above which the NEXT launch uses K1m
    int ondemand_flips;     // K1w: sweeps whose predecessor had fewer accepted flips per workgroup run on demand (0 = never)
    int debug;              // diagnostic timing builds only: bit0 = skip LDS-DMA, bit1 = accept nothing (K1w);
                            // K1m: bit2 = no decision chain, bit3 = no MFMAs
};

constexpr int kDenseFieldsIn = 1;    // start from the cached fields in DenseArgs::fields (no re-initialisation)
constexpr int kDenseFieldsOut = 2;   // leave the cached fields there at the end
constexpr int kDenseNoEnergy = 4;    // not the last launch of a run: skip the energy epilogue

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));


typedef float f32x2 __attribute__((ext_vector_type(2)));

// arguments of the structured (slot-ELL) kernels K2 / K3 -- sparse_kernels.hip
struct EllArgs {
    const uint32_t *ell_col;   // [slots][D][64] neighbour indices (padding: the variable itself)
    const float *ell_val;      // [slots][D][64] S_ij            (padding: +0.0f)
    const float *lin;          // K2: slots*64 linear terms (zero padded); K3: unused
    const float *temps;        // num_sweeps
    const void *init;          // nullable: R x n uint8 (K2) / uint16 (K3)
    void *states;              // R x n uint8 (K2) / uint16 (K3)
    double *energy;            // R
    unsigned long long *stats; // [1] accepted
    float c_pair;
    double offset;             // K2: offset; K3: lin_offset
    int n, K, R, num_sweeps, resync, slots, D;
    uint32_t replica_offset, seed_lo, seed_hi;
    uint32_t sweep_offset;     // see DenseArgs
    int temps_per_replica;
    // K2: row-major copy of the adjacency with the neighbours inside the variable's own 64-slot first
    const uint2 *rows;         // [slots*64][D] (col, val bits), padding (self, +0)
    const uint32_t *meta;      // per variable: in-slot neighbour count | degree << 8
    const uint4 *adj4;         // K2: per slot, per group of 4 entries: [64][4] packed neighbours (LDS byte offset of
                               // the neighbour's 32-bit state word << 8 | bit), then [64][4] values
    int state_bytes;           // K2: adj4 holds plain neighbour indices and the state is one byte per variable in LDS
    const uint32_t *slot_flags;// K2: per slot, non-zero when some variable of the slot has an in-slot neighbour
    int waves_override;        // (unused)
    int min_size;              // K3: a move out of a cluster with exactly min_size members is rejected (0 = off)
    // optional fp64 coefficients for the final energies only (same layouts as ell_val / lin); null = use the fp32 model
    const double *ell_val64;
    const double *lin64;
    double c_pair64;
    // K2 family: weights of the uniform pair term (mi_sa_problem_set_pair_weights).  wslot >= 0: the one slot whose
    // variables carry weights other than 1 (wgt = its 64 weights, 0 at a hole); every kernel then carries
    // sum_j w_j z_j where it carried sum_j z_j and sweeps that slot with a serial loop.  wslot < 0: all weights 1.
    const int32_t *wgt = nullptr;
    int wslot = -1;
    int ring_off = 0;          // K2 with a threshold wavefront: byte offset of the ring of thresholds in LDS (set by its launcher)
};

// The weighted slot of a structured binary model (EllArgs::wslot): a sequential sweep over its lanes -- few variables,
// the slack bits of a squared constraint, coupled to everything through the pair term only -- with the oracle's expression
// f = g + (c * (float)w) * (float)(A - w z).  z: the lane's bit; A: sum_j w_j z_j, updated.  Returns the mask of flips.
__device__ __forceinline__ uint64_t weighted_slot_sweep(float g, float thr, int w, float c_pair, uint32_t z, int &A, int lane)
{
    const float cw = c_pair * (float)w;
    uint64_t todo = __ballot(w != 0), flipped = 0ull;
    uint32_t zz = z;
    while (todo != 0ull) {
        const int l = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1ull;
        const float f = g + cw * (float)(A - w * (int)zz);
        const float dE = zz ? -f : f;
        if ((__ballot(dE < thr) >> l) & 1ull) {                     // wave-uniform: lane l accepts
            const int wl = __builtin_amdgcn_readlane(w, l);
            const int zl = __builtin_amdgcn_readlane((int)zz, l);
            A += zl ? -wl : wl;
            if (lane == l) zz ^= 1u;
            flipped |= 1ull << l;
        }
    }
    return flipped;
}

// sum over the wavefront of a small non-negative integer (initialisation / epilogue of the weighted slot)
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
int mi_launch_csr_rank1(const EllArgs &, hipStream_t, bool tw = false);   // tw: + a threshold wavefront (runs of up to 1024 replicas; bit / byte state, 16 / 32 entries)
int mi_launch_csr_rank1_pair(const EllArgs &, bool tw, hipStream_t);   // sparse_pair_kernels.hip: two replicas per wavefront (tw: + a threshold wavefront)
int mi_launch_csr_rank1_split(const EllArgs &, int nw, hipStream_t);   // sparse_split_kernels.hip: nw wavefronts per replica
int mi_launch_csr_rank1_wide(const EllArgs &, int spb, bool tw, hipStream_t);   // ... one wavefront per replica, spb slots per step (tw: + a threshold wavefront; spb = 1 only so)
int mi_launch_potts(const EllArgs &, hipStream_t);
// potts_fast_kernels.hip: K3f, the Potts chain for models whose every slot is free of internal edges (K <= 16, 16 / 32
// entries per variable, no size constraint); adj4 = packed adjacency with neighbour word = 2 * index
bool mi_potts_fast_eligible(int D, int K, int min_size);
int mi_launch_potts_fast(const EllArgs &, bool tw, hipStream_t);

// K1x (dense_xl_kernels.hip): dense chain for 4096 < n <= 65536, one workgroup per replica
struct DenseXlArgs {
    const float *Q2;        // n rows x (chunks*4096) floats: 2*Qs off-diagonal, 0 on the diagonal and in the padding
    const float *diag;      // chunks*4096 floats (zero padded)
    const float *temps;
    const uint8_t *init;    // nullable, R x n
    uint8_t *states;        // R x n
    double *energy;         // R
    unsigned long long *stats;
    double offset;
    int n, R, num_sweeps, resync;
    uint32_t replica_offset, seed_lo, seed_hi, sweep_offset;
    int temps_per_replica;
    // K1x continuing a run of K1g: the cached fields of all replicas as K1g keeps them, F[r / 64][column][r % 64]
    // (fin_ncols columns per replica range); nullptr = build the fields from the states (as at the start of a run)
    const float *fields_in = nullptr;
    int fin_ncols = 0;
    int xg_chain = 0;       // K1g, the chain of a group of blocks: 0 auto, 1 = a DIAG and a small pass per block, 2 = one launch (k_xg_chain)
};
int mi_launch_dense_xl(const DenseXlArgs &, int chunks, hipStream_t);
// K1g (dense_xg_kernels.hip): the same run for many replicas at once, row updates as a GEMM-shaped pass per 64 rows
size_t mi_dense_xg_workspace_bytes(int n, int R);
// phase: bit 0 = first call of a run (state words, field initialisation), bit 1 = write states and energies
int mi_launch_dense_xg(const DenseXlArgs &, int chunks, void *workspace, hipStream_t, int phase);
const float *mi_dense_xg_fields(void *workspace);

// random word of (variable i, sweep s, global replica g, tag) -- the per-variable form of the chain's RNG
// addressing (one Philox block per call; the wave kernels share a block between four slots instead)
__device__ __forceinline__ uint32_t chain_word_dev(uint32_t i, uint32_t s, uint32_t g, uint32_t tag, uint32_t k0, uint32_t k1)
{
    uint32_t w[4];
    philox4x32_10(((i >> 8) << 6) | (i & 63u), s, g, tag, k0, k1, w);
    const uint32_t sel = (i >> 6) & 3u;
    return sel == 0 ? w[0] : (sel == 1 ? w[1] : (sel == 2 ? w[2] : w[3]));
}

// K4 launcher (energy_kernels.hip); all pointers are device pointers
size_t mi_energy_dense_scratch_bytes(int n, int R);
int mi_launch_energy_dense(const float *dQ, int n, int ldq, const uint8_t *dX, int R, double offset, double *dE,
                           uint8_t *dXt, int path, hipStream_t st);
int mi_launch_energy_dense_f64(const double *dQ, int n, const uint8_t *dX, int R, double offset, double *dE, hipStream_t st);

// Energy of the final state, E = sum_i x_i diag_i + 1/2 sum_{i,j} x_i x_j Q2_ij, with every fp32 matrix
// entry added EXACTLY once into fp64 accumulators (lane l sums its own columns over all set rows; one
// wave reduction at the end).  Independent of the cached fp32 fields, so the reported energies carry
// no accumulated rounding of the chain.
template <int NT>
__device__ __forceinline__ double dense_energy_f64(__amdgpu_buffer_rsrc_t rsrc, int diag_row, uint64_t xb,
                                                   int lane)
{
    constexpr int G = NT / 4;
    const int voff = lane * 16;
    double pair = 0.0, lin = 0.0;
#pragma unroll 1
    for (int t = -1; t < NT; ++t) {
        uint64_t m = (t < 0) ? 1ull : __ballot((xb >> t) & 1ull);
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const int soff = ((t < 0) ? diag_row : t * 64 + l) * (NT * 256);
            double acc = 0.0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + g * 1024, 0);
                acc += ((xb >> (4 * g + 0)) & 1ull) ? (double)__uint_as_float(q.x) : 0.0;
                acc += ((xb >> (4 * g + 1)) & 1ull) ? (double)__uint_as_float(q.y) : 0.0;
                acc += ((xb >> (4 * g + 2)) & 1ull) ? (double)__uint_as_float(q.z) : 0.0;
                acc += ((xb >> (4 * g + 3)) & 1ull) ? (double)__uint_as_float(q.w) : 0.0;
            }
            if (t < 0) lin = acc; else pair += acc;
        }
    }
    return wave_sum_f64(lin + 0.5 * pair);
}

// ---- kernel scheduling across the launches of a chunked run ------------------------------------------
// ctrl[kCtrlModes + c] = kernel that serves chunk c (kModeWg / kModeMfma), ctrl[1] = workgroups finished,
// ctrl[2] = flips accepted by the running launch, ctrl[4], ctrl[5] = chunks served by each kernel.  For every
// chunk BOTH kernels are enqueued; the one whose mode is not set exits at once.  The last workgroup of the
// kernel that did run picks the mode of the NEXT chunk from the acceptance the run has reached (a word per
// chunk, so the second kernel of the same chunk cannot see the update).  Results do not depend on the
// choice: same chain, same state.
constexpr unsigned int kModeWg = 0, kModeMfma = 1;
constexpr int kCtrlModes = 8, kCtrlWords = 4096;

__device__ __forceinline__ bool sched_my_turn(const DenseArgs &a)
{
    if (!a.ctrl) return true;
    return (unsigned int)__builtin_amdgcn_readfirstlane(
               (int)__hip_atomic_load(a.ctrl + kCtrlModes + a.chunk_index, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT)) == (unsigned int)a.my_mode;
}

// called by ONE thread per workgroup after the workgroup's results are stored; wg_flips = its accepted flips
__device__ __forceinline__ void sched_finish(const DenseArgs &a, unsigned int wg_flips)
{
    if (!a.ctrl) return;
    atomicAdd(&a.ctrl[2], wg_flips);
    __threadfence();
    const unsigned int ticket = atomicAdd(&a.ctrl[1], 1u);
    if (ticket == gridDim.x - 1) {
        __threadfence();
        const unsigned int total = __hip_atomic_load(&a.ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int next = (a.mode_up_flips != 0 && total >= a.mode_up_flips) ? kModeMfma : kModeWg;
        __hip_atomic_store(&a.ctrl[kCtrlModes + a.chunk_index + 1], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.ctrl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.ctrl[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&a.ctrl[4 + a.my_mode], 1u);          // diagnostics: launches served by each kernel
    }
}

// one translation unit per NT (dense_mfma_kernels.hip) for the sizes whose LDS plan fits (NT <= 44)
#define MI_DECLARE_MFMA(N) int mi_launch_dense_mfma_nt##N(const DenseArgs &, hipStream_t);
MI_DECLARE_MFMA(4) MI_DECLARE_MFMA(8) MI_DECLARE_MFMA(12) MI_DECLARE_MFMA(16) MI_DECLARE_MFMA(20)
MI_DECLARE_MFMA(24) MI_DECLARE_MFMA(28) MI_DECLARE_MFMA(32) MI_DECLARE_MFMA(36) MI_DECLARE_MFMA(40)
MI_DECLARE_MFMA(44)
#undef MI_DECLARE_MFMA
constexpr int kMaxMfmaNT = 44;

// what a per-NT launcher needs to know about the problem handle
struct DenseLaunchCtx {
    int device;
    int opt_pace, opt_variant, opt_unit_rows, opt_ondemand_permille, opt_chunk_sweeps, opt_mfma_permille;
    float *d_fields;               // R x 64*NT floats or nullptr
    unsigned int *d_ctrl;
    unsigned int *d_pace;          // kMaxChunks * kPaceWords words
    int *resident_waves;           // cached occupancy of the wave-per-replica kernel (0 = unknown)
    int *launches;                 // out: kernel launches that serve this anneal (a chunked run has several)
};

constexpr int kMaxChunks = 64;

// one translation unit per NT (dense_kernels.hip compiled with -DMI_NT=<NT>) defines its launcher
#define MI_DECLARE_DENSE(N) int mi_launch_dense_nt##N(const DenseLaunchCtx &, const DenseArgs &, hipStream_t);
MI_DECLARE_DENSE(4) MI_DECLARE_DENSE(8) MI_DECLARE_DENSE(12) MI_DECLARE_DENSE(16) MI_DECLARE_DENSE(20)
MI_DECLARE_DENSE(24) MI_DECLARE_DENSE(28) MI_DECLARE_DENSE(32) MI_DECLARE_DENSE(36) MI_DECLARE_DENSE(40)
MI_DECLARE_DENSE(44) MI_DECLARE_DENSE(48) MI_DECLARE_DENSE(52) MI_DECLARE_DENSE(56) MI_DECLARE_DENSE(60)
MI_DECLARE_DENSE(64)
#undef MI_DECLARE_DENSE

}  // namespace mi_sa_impl
