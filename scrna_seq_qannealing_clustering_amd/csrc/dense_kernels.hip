// dense_kernels.hip -- the dense-QUBO anneal kernels (K1 wave-per-replica, K1w workgroup/LDS ring) and
// their launcher for ONE field-register count NT = MI_NT (n <= 64*NT).  Compiled once per NT so that the
// sixteen sizes build in parallel (each is a fully unrolled ~10k-instruction kernel).
//
// Chain specification, layouts and the reference call sites served: see mi_sa.hip / DESIGN.md.
#include "mi_sa_device.h"

#ifndef MI_NT
#error "compile with -DMI_NT=<4,8,...,64>"
#endif

namespace mi_sa_impl {

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
    int until_resync = a.resync_first;           // counts down at the START of a sweep
    for (int s = 0; s < a.num_sweeps; ++s) {
        bool init_now = (s == 0);
        if (a.resync > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now) dense_field_init<NT>(f, rsrc, diag_row, xb, lane);
        // temperature of this sweep as a scalar (SGPR) operand
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(a.temps[a.temps_per_replica ? r : s])));
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
                    philox4x32_10((uint32_t)((t >> 2) * 64 + ln), (uint32_t)s + a.sweep_offset, g, 0u, a.seed_lo,
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

    // final states out; energies re-evaluated exactly (fp64 accumulation of the fp32 entries)
    uint8_t *dst = a.states + (size_t)r * n + lane;
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t * 64 + lane < n) dst[t * 64] = ((xb >> t) & 1ull) ? 1 : 0;
    });
    const double e = dense_energy_f64<NT>(rsrc, diag_row, xb, lane);
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
constexpr int kLdsBytes = 160 * 1024 - 256;   // ring budget; the last 256 bytes hold the sweep-mode words

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
//   * fields live as NT/2 register PAIRS so that each 16-byte piece is consumed by two v_pk_fma_f32;
//   * pieces are visited starting with the one that holds slot T (rotation by T/4) and `after_first`
//     runs right after it: the field of the slot being swept is final there, so the caller picks the
//     NEXT flip while the rest of this row is still being added (shortens the serial flip chain);
//   * a rolling window of W pieces is in flight (LDS latency is short; a whole row in registers at
//     once is what made this kernel spill).
template <int NT, int T, int W, typename F>
__device__ __forceinline__ void dense_add_row_lds(f32x2 (&f)[NT / 2], const char *row_lane, float s,
                                                  F &&after_first)
{
    constexpr int G = NT / 4, G0 = T / 4, WW = W < G ? W : G;
    const f32x2 s2 = {s, s};
    f32x4 q[WW];
#pragma unroll
    for (int k = 0; k < WW; ++k)
        q[k] = *reinterpret_cast<const f32x4 *>(row_lane + ((G0 + k) % G) * 1024);
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int g = (G0 + k) % G;
        const f32x2 lo = {q[k % WW].x, q[k % WW].y}, hi = {q[k % WW].z, q[k % WW].w};
        f[2 * g + 0] = __builtin_elementwise_fma(s2, lo, f[2 * g + 0]);
        f[2 * g + 1] = __builtin_elementwise_fma(s2, hi, f[2 * g + 1]);
        if (k + WW < G)
            q[k % WW] = *reinterpret_cast<const f32x4 *>(row_lane + ((G0 + k + WW) % G) * 1024);
        if (k == 0) after_first();
    }
}

// f (+)= s * Q2[row] straight from L2 / Infinity Cache (K1-style buffer loads) into the pair layout of
// K1w: used by the ON-DEMAND sweeps, where accepted flips are too rare to be worth streaming Q.  Two
// halves, so that at most NT/2+2 extra registers are live.
template <int NT>
__device__ __forceinline__ void dense_add_row_pairs(f32x2 (&f)[NT / 2], __amdgpu_buffer_rsrc_t rsrc, int row,
                                                    int lane, float s)
{
    constexpr int G = NT / 4, H = (G + 1) / 2;
    const int voff = lane * 16;
    const int soff = row * (NT * 256);
    const f32x2 s2 = {s, s};
    {
        u32x4 q[H];
#pragma unroll
        for (int g = 0; g < H; ++g) q[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + g * 1024, 0);
#pragma unroll
        for (int g = 0; g < H; ++g) {
            f[2 * g + 0] = __builtin_elementwise_fma(s2, f32x2{__uint_as_float(q[g].x), __uint_as_float(q[g].y)}, f[2 * g + 0]);
            f[2 * g + 1] = __builtin_elementwise_fma(s2, f32x2{__uint_as_float(q[g].z), __uint_as_float(q[g].w)}, f[2 * g + 1]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G > H) {
        u32x4 q[G - H];
#pragma unroll
        for (int g = H; g < G; ++g) q[g - H] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + g * 1024, 0);
#pragma unroll
        for (int g = H; g < G; ++g) {
            f[2 * g + 0] = __builtin_elementwise_fma(s2, f32x2{__uint_as_float(q[g - H].x), __uint_as_float(q[g - H].y)}, f[2 * g + 0]);
            f[2 * g + 1] = __builtin_elementwise_fma(s2, f32x2{__uint_as_float(q[g - H].z), __uint_as_float(q[g - H].w)}, f[2 * g + 1]);
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
    __shared__ __attribute__((aligned(16))) char ring[C::U * C::UNITB + 256];
    if (!sched_my_turn(a)) return;               // a chunk that the other dense kernel serves
    // last 256 bytes: three rotating words counting the workgroup's accepted flips per sweep
    unsigned int *flips_word = reinterpret_cast<unsigned int *>(ring + C::U * C::UNITB);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x * kWgWaves + wave;
    const bool active = r < a.R;                 // idle waves still take part in DMA and barriers
    const uint32_t g = a.replica_offset + (uint32_t)r;
    const int n = a.n;
    const int slots_used = (n + 63) >> 6;
    const int units_per_sweep = slots_used * (64 / GR);
    const int total_units = units_per_sweep;     // the ring is primed and drained once per STREAMED sweep

    // rows 0..64*slots_used-1 (zero rows past n) + the diagonal row at index 64*slots_used
    const int diag_row = slots_used * 64;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.Qp), 0, (diag_row + 1) * C::ROWB, 0x00020000);

    f32x2 f[NT / 2];                            // field of slot t = f[t >> 1][t & 1]
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
    // (kept in SGPRs by force: hipcc's uniformity analysis otherwise demotes them to VGPRs and wraps
    // every LDS-DMA in a waterfall loop)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    int issued = 0;                              // units whose DMA has been issued
    int issue_row = 0;                           // first row of the next unit to issue
    int issue_slot = 0;                          // ring slot of the next unit to issue
    int cur_slot = 0;                            // ring slot of the unit being processed
    int processed = 0;                           // units fully processed
    auto issue_unit = [&]() {
        if (issued < total_units) {
            if (wave < C::G && !(a.debug & 1)) {
#pragma unroll
                for (int k = 0; k < GR; ++k)
                    lds_dma_16(rsrc, ring + issue_slot * C::UNITB + k * C::ROWB + wave * 1024, lane * 16,
                               (issue_row + k) * C::ROWB + wave * 1024);
            }
            issued = uni(issued + 1);
            issue_row = uni(issue_row + GR >= units_per_sweep * GR ? 0 : issue_row + GR);
            issue_slot = uni((issue_slot + 1 == C::U) ? 0 : issue_slot + 1);
        }
    };

    // Sweep mode (identical in all 16 waves): STREAM = Q through the LDS ring, one rendezvous per unit;
    // ON-DEMAND = no streaming and no rendezvous, the (rare) accepted flips fetch their row from L2.
    // Chosen per sweep from the workgroup's accepted-flip count of the previous sweep: a geometric
    // schedule spends most of its sweeps cold (acceptance << 1 %), where streaming 30 MB per sweep per
    // workgroup to serve a handful of flips is all overhead.  Results do not depend on the mode.
    if (threadIdx.x < 4) flips_word[threadIdx.x] = 0u;     // [0..2] per-sweep counts, [3] whole launch
    __syncthreads();
    bool stream = true;
    unsigned long long accepted = 0;
    int until_resync = a.resync_first;           // counts down at the START of a sweep
    for (int s = 0; s < a.num_sweeps; ++s) {
        const unsigned long long accepted_before = accepted;
        bool init_now = (s == 0) && !(a.flags & kDenseFieldsIn);
        if (a.resync > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now && active) {
            float fs[NT];
            dense_field_init<NT>(fs, rsrc, diag_row, xb, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) f[t >> 1][t & 1] = fs[t];
        } else if (s == 0 && active) {
            // continue a run: the cached fields of the previous launch, canonical [replica][column]
            const float *src = a.fields + (size_t)r * (NT * 64) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) f[t >> 1][t & 1] = src[t * 64];
        }
        if (stream) {
            // everything above used ordinary loads; from here on only LDS-DMA is in the VM queue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issued = 0; processed = 0; issue_row = 0; issue_slot = 0; cur_slot = 0;
#pragma unroll 1
            for (int u = 0; u < C::U - 1; ++u) issue_unit();
        }
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(a.temps[a.temps_per_replica ? (active ? r : 0) : s])));
        uint32_t w[4];
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            int nn = n, ln = lane;
            asm volatile("" : "+s"(nn));
            asm volatile("" : "+v"(ln));
            const int left = nn - t * 64;
            if (left > 0) {                     // wave-uniform, identical in every wave of the block
                if constexpr ((t & 3) == 0)
                    philox4x32_10((uint32_t)((t >> 2) * 64 + ln), (uint32_t)s + a.sweep_offset, g, 0u, a.seed_lo,
                                  a.seed_hi, w);
                float thr = neglog_u(w[t & 3]) * T;
                if (ln >= left || !active || (a.debug & 2)) thr = -INFINITY;
                float sg = ((xb >> t) & 1ull) ? -1.0f : 1.0f;
                uint64_t todo = ~0ull;
                if (stream) {
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
                    uint64_t m = __ballot(sg * f[t >> 1][t & 1] < thr) & todo & umask;
                    while (m != 0) {
                        const int l = __ffsll((unsigned long long)m) - 1;
                        todo = (l == 63) ? 0ull : (~0ull << (l + 1));
                        const float sl = readlane_f(sg, l);
                        if (ln == l) { sg = -sg; xb ^= (1ull << t); }
                        dense_add_row_lds<NT, t, 8>(f, unit + (l - j * GR) * C::ROWB + ln * 16, sl, [&]() {
                            m = __ballot(sg * f[t >> 1][t & 1] < thr) & todo & umask;
                        });
                        ++accepted;
                    }
                    cur_slot = uni((cur_slot + 1 == C::U) ? 0 : cur_slot + 1);
                    processed = uni(processed + 1);
                }
                } else {
                    // on-demand: the plain lowest-accepting-lane loop over the whole slot (as K1)
                    while (true) {
                        const uint64_t m = __ballot(sg * f[t >> 1][t & 1] < thr) & todo;
                        if (m == 0) break;
                        const int l = __ffsll((unsigned long long)m) - 1;
                        todo = (~0ull << l) << 1;
                        const float sl = readlane_f(sg, l);
                        if (ln == l) { sg = -sg; xb ^= (1ull << t); }
                        dense_add_row_pairs<NT>(f, rsrc, t * 64 + l, ln, sl);
                        ++accepted;
                    }
                }
            }
        });
        // workgroup-wide accepted-flip count of this sweep -> mode of the next one.  Three rotating words:
        // the word of sweep s+2 is cleared after the rendezvous of sweep s, i.e. strictly before any wave
        // can add to it (that needs the rendezvous of sweep s+1).
        if (lane == 0 && accepted != accepted_before)
            atomicAdd(&flips_word[s % 3], (unsigned int)(accepted - accepted_before));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned int wg_flips = (unsigned int)__builtin_amdgcn_readfirstlane((int)flips_word[s % 3]);
        if (threadIdx.x == 0) flips_word[(s + 2) % 3] = 0u;
        stream = (a.ondemand_flips == 0) || (wg_flips >= (unsigned int)a.ondemand_flips);
        if (pace_pop) sweep_pace_arrive_wait(a.pace, xcc, pace_pop, (unsigned int)(s + 1), stream);
    }

    if (active) {
        uint8_t *dst = a.states + (size_t)r * n + lane;
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (t * 64 + lane < n) dst[t * 64] = ((xb >> t) & 1ull) ? 1 : 0;
        });
        if (a.flags & kDenseFieldsOut) {
            float *out = a.fields + (size_t)r * (NT * 64) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) out[t * 64] = f[t >> 1][t & 1];
        }
        if (lane == 0) {
            atomicAdd(&a.stats[1], accepted);
            atomicAdd(&flips_word[3], (unsigned int)accepted);
        }
        if (!(a.flags & kDenseNoEnergy)) {
            const double e = dense_energy_f64<NT>(rsrc, diag_row, xb, lane);
            if (lane == 0) a.energy[r] = e + a.offset;
        }
    }
    if (a.ctrl) {                                // wave-uniform: pick the kernel of the next chunk
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (threadIdx.x == 0) sched_finish(a, flips_word[3]);
    }
}

namespace {


// Launches the anneal in chunks of at most `resident` replicas (= wavefronts), so that every wave of
// a launch is co-resident and the sweep pacing rendezvous can complete; chunks run back to back on
// the stream.  Each chunk gets its own zeroed pacing words.
template <int NT>
int launch_dense(const DenseLaunchCtx &p, DenseArgs a, hipStream_t st)
{
    if ((*p.resident_waves) == 0) {
        int blocks_per_cu = 0, cus = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_anneal_dense<NT>, 256, 0));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device));
        (*p.resident_waves) = blocks_per_cu * cus * 4;
        if ((*p.resident_waves) < 4) return fail(MI_EHIP, "anneal kernel cannot be resident (occupancy 0)");
    }
    const int total = a.R;
    const uint32_t base_offset = a.replica_offset;
    const uint8_t *init0 = a.init;
    uint8_t *states0 = a.states;
    double *energy0 = a.energy;
    const float *temps0 = a.temps;
    int chunk = (*p.resident_waves);
    if ((total + chunk - 1) / chunk > kMaxChunks) chunk = (total + kMaxChunks - 1) / kMaxChunks;
    const bool pace = p.opt_pace && a.num_sweeps > 1;
    if (pace)
        HIP_TRY(hipMemsetAsync(p.d_pace, 0, kMaxChunks * kPaceWords * sizeof(unsigned int), st));
    int c = 0;
    for (int lo = 0; lo < total; lo += chunk, ++c) {
        const int cnt = total - lo < chunk ? total - lo : chunk;
        a.R = cnt;
        a.replica_offset = base_offset + (uint32_t)lo;
        a.init = init0 ? init0 + (size_t)lo * a.n : nullptr;
        a.states = states0 + (size_t)lo * a.n;
        a.energy = energy0 + lo;
        a.temps = temps0 + (a.temps_per_replica ? lo : 0);          // one temperature per replica: the chunk's own
        a.pace = (pace && cnt <= (*p.resident_waves)) ? p.d_pace + (size_t)c * kPaceWords : nullptr;
        note_kernel("k_anneal_dense<%d>", NT);
        hipLaunchKernelGGL(k_anneal_dense<NT>, dim3((cnt + 3) / 4), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
    }
    *p.launches = c;
    return MI_OK;
}

template <int NT, int GR>
int launch_dense_wg(const DenseLaunchCtx &p, DenseArgs a, hipStream_t st)
{
    if constexpr (WgCfg<NT, GR>::ok) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device));
        const int blocks = (a.R + kWgWaves - 1) / kWgWaves;
        a.ondemand_flips = (int)((long long)p.opt_ondemand_permille * a.n * kWgWaves / 1000);
        a.pace = nullptr;
        if (p.opt_pace && a.num_sweeps > 1 && blocks <= cus) {     // one 160 KB workgroup per CU
            HIP_TRY(hipMemsetAsync(p.d_pace, 0, kPaceWords * sizeof(unsigned int), st));
            a.pace = p.d_pace;
        }
        note_kernel("k_anneal_dense_wg<%d,%d>", NT, GR);
        hipLaunchKernelGGL((k_anneal_dense_wg<NT, GR>), dim3((a.R + kWgWaves - 1) / kWgWaves), dim3(1024), 0, st, a);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    } else {
        return fail(MI_EUNSUPPORTED, "LDS ring does not fit for NT=%d unit_rows=%d", NT, GR);
    }
}

// sweeps until the first field re-synchronisation of a launch that starts at sweep s0 of its run
inline int resync_first_for(int resync, int s0)
{
    return resync > 0 ? ((resync - s0 % resync) % resync) + 1 : 0;
}

template <int NT>
int launch_dense_wg_any(const DenseLaunchCtx &p, const DenseArgs &a, hipStream_t st)
{
    int gr = p.opt_unit_rows ? p.opt_unit_rows : 4;       // 4 rows per rendezvous measured fastest
    if (gr == 4 && !WgCfg<NT, 4>::ok) gr = 2;
    switch (gr) {
        case 2: return launch_dense_wg<NT, 2>(p, a, st);
        case 4: return launch_dense_wg<NT, 4>(p, a, st);
    }
    return fail(MI_EUNSUPPORTED, "LDS ring unit of %d rows is not built for NT=%d", gr, NT);
}

// K1w over a long schedule: the run is cut into launches of `chunk` sweeps.  State (bits in
// DenseArgs::states, cached fields in DenseArgs::fields) persists in HBM between launches, so the cut
// points are invisible to the chain -- results are bit-identical to one launch -- and each launch can
// be served by whichever kernel suits the acceptance rate the run has reached.
template <int NT>
int launch_dense_mfma(const DenseArgs &a, hipStream_t st)
{
#if MI_NT <= 44
#define MI_CAT2(x, y) x##y
#define MI_CAT(x, y) MI_CAT2(x, y)
    return MI_CAT(mi_launch_dense_mfma_nt, MI_NT)(a, st);
#undef MI_CAT
#undef MI_CAT2
#else
    (void)a; (void)st;
    return fail(MI_EUNSUPPORTED, "K1m is not built for NT=%d (n > %d)", NT, kMaxMfmaNT * 64);
#endif
}

template <int NT>
int launch_dense_chunked(const DenseLaunchCtx &p, const DenseArgs &a, hipStream_t st, int variant)
{
    const int chunk = p.opt_chunk_sweeps;
    // variant 4 = scheduled: every chunk is offered to both kernels, the device-side mode word decides
    const bool scheduled = (variant == 4);
    if (variant == 3 && (chunk <= 0 || a.num_sweeps <= chunk || !p.d_fields)) {
        DenseArgs b = a;
        b.fields = nullptr; b.ctrl = nullptr; b.flags = 0; b.my_mode = 0; b.mode_up_flips = 0; b.chunk_index = 0;
        b.resync_first = resync_first_for(a.resync, 0);
        return launch_dense_mfma<NT>(b, st);
    }
    if (chunk <= 0 || a.num_sweeps <= chunk || !p.d_fields) {
        DenseArgs b = a;
        b.fields = nullptr; b.ctrl = nullptr; b.flags = 0; b.my_mode = 0; b.mode_up_flips = 0; b.chunk_index = 0;
        b.resync_first = resync_first_for(a.resync, 0);
        return launch_dense_wg_any<NT>(p, b, st);   // (variant 4 never gets here: it requires chunking)
    }
    if (scheduled) {
        if ((a.num_sweeps + chunk - 1) / chunk + 2 > kCtrlWords - kCtrlModes)
            return fail(MI_EINVAL, "too many chunks for the scheduler: raise chunk_sweeps");
        HIP_TRY(hipMemsetAsync(p.d_ctrl, 0, kCtrlWords * sizeof(unsigned int), st));
        const unsigned int start = kModeMfma;            // start hot; a short first chunk calibrates
        HIP_TRY(hipMemcpyAsync(p.d_ctrl + kCtrlModes, &start, sizeof start, hipMemcpyHostToDevice, st));
    }
    int chunk_index = 0;
    const int first_len = scheduled ? (chunk < 8 ? chunk : 8) : chunk;
    for (int s0 = 0, len = first_len; s0 < a.num_sweeps; s0 += len, len = chunk) {
        const bool first = (s0 == 0), last = (s0 + len >= a.num_sweeps);
        DenseArgs b = a;
        b.num_sweeps = last ? a.num_sweeps - s0 : len;
        b.temps = a.temps_per_replica ? a.temps : a.temps + s0;
        b.sweep_offset = a.sweep_offset + (uint32_t)s0;
        b.init = first ? a.init : a.states;
        b.fields = p.d_fields; b.ctrl = nullptr; b.my_mode = 0; b.mode_up_flips = 0;
        b.flags = (first ? 0 : kDenseFieldsIn) | (last ? 0 : (kDenseFieldsOut | kDenseNoEnergy));
        b.resync_first = resync_first_for(a.resync, s0);
        b.chunk_index = chunk_index++;
        int rc;
        if (scheduled) {
            // next chunk goes to K1m when this one accepted at least opt_mfma_permille of its proposals
            b.ctrl = p.d_ctrl;
            b.mode_up_flips = (unsigned int)((double)p.opt_mfma_permille * 1e-3 * (double)a.R * a.n * b.num_sweeps);
            if (b.mode_up_flips == 0) b.mode_up_flips = 1;
            b.my_mode = (int)kModeMfma;
            rc = launch_dense_mfma<NT>(b, st);
            if (rc) return rc;
            b.my_mode = (int)kModeWg;
            rc = launch_dense_wg_any<NT>(p, b, st);
        } else {
            rc = (variant == 3) ? launch_dense_mfma<NT>(b, st) : launch_dense_wg_any<NT>(p, b, st);
        }
        if (rc) return rc;
    }
    *p.launches = chunk_index;
    return MI_OK;
}

template <int NT>
int launch_dense_any(const DenseLaunchCtx &p, const DenseArgs &a, hipStream_t st)
{
    int variant = p.opt_variant;
    if (variant == 0) {
        variant = (a.R >= 2 * kWgWaves && a.num_sweeps > 0) ? 2 : 1;
        // 128 .. 1024 replicas: a run is one chain's latency whatever serves it (85 ms per 200 sweeps at n = 2638 from 64 to
        // 2048 replicas), and with at most one wavefront per SIMD the wave-per-replica kernel is the quicker one by 7-10 %
        // (scripts/perf_dense_few.py); same chain, bit for bit
        if (variant == 2 && a.R >= 128 && a.R <= 1024) variant = 1;
        // long schedules on sizes K1m is built for: let the device alternate K1m (hot) and K1w (cold)
        if (variant == 2 && NT <= kMaxMfmaNT && a.Qm && p.d_fields && p.opt_chunk_sweeps > 0 &&
            a.num_sweeps > p.opt_chunk_sweeps && p.opt_mfma_permille > 0)
            variant = 4;
    }
    if (variant >= 2) return launch_dense_chunked<NT>(p, a, st, variant);
    DenseArgs b = a;
    b.fields = nullptr; b.ctrl = nullptr; b.flags = 0; b.my_mode = 0; b.mode_up_flips = 0; b.chunk_index = 0;
    b.resync_first = resync_first_for(a.resync, 0);
    return launch_dense<NT>(p, b, st);
}

}  // namespace

#define MI_CAT2(a, b) a##b
#define MI_CAT(a, b) MI_CAT2(a, b)
int MI_CAT(mi_launch_dense_nt, MI_NT)(const DenseLaunchCtx &ctx, const DenseArgs &a, hipStream_t st)
{
    return launch_dense_any<MI_NT>(ctx, a, st);
}

}  // namespace mi_sa_impl
