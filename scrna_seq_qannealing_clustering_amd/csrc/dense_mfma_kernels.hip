// dense_mfma_kernels.hip -- K1m: the dense-QUBO chain with the row updates on the matrix cores (gfx950).
//
// Same chain as K1 / K1w, bit for bit; what changes is WHO holds the cached fields.
// A workgroup = 16 wavefronts = 16 replicas, but the fields of all 16 replicas live TRANSPOSED, as the
// accumulators of f32-input MFMAs:  wave w owns the columns [w*S, (w+1)*S) (S = 4*NT = n_pad/16) of every
// replica, as NT/4 tiles of v_mfma_f32_16x16x4_f32 accumulators
//       C[i][r] = f_{column w*S + 16*tile + i}(replica r)        (lane l: r = l & 15, i = 4*(l >> 4) + reg).
// The flips a unit of 4 consecutive rows R0..R0+3 causes in the 16 replicas are ONE rank-4 update
//       F[:, r] += sum_k Q2[R0+k][:] * s_k(r),      s_k(r) in {-1, 0, +1},
// i.e. per tile one MFMA with A[i][k] = Q2[R0+k][column i] (read from the LDS ring, 4 B/lane) and
// B[k][r] = s_k(r).  The f32-input MFMA is an exact fp32 fmaf chain in k order
//       D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// which is precisely the oracle's "f += sgn * Q2" applied for the accepted rows in row order (a rejected
// row has s = 0 and fma(a, 0, C) = C).  So a Q row read from LDS once serves all 16 replicas: LDS traffic
// per unit drops from (accepted flips) x 11 KB to 45 KB flat, and the kernel is bound by the fp32 FMA rate
// (the MFMA pipe) instead of the LDS->VGPR fill rate that bounds K1w at the hot end of a schedule.
//
// The accept/reject decisions of a unit are made by the 16 lanes that hold the four fields involved
// (owner wave R0 / S, tile (R0 % S) / 16, lane quarter (R0 % 16) / 4), sequentially over the 4 rows, with
// the intra-unit couplings applied by the same fmaf chain the MFMA will execute.  Thresholds come from the
// same Philox addressing as everywhere else (wave w draws for replica w, 4 slots per block) through LDS.
//
// Used for the hot part of a schedule only (acceptance above ~10 %): the launcher alternates K1m and K1w
// per chunk of sweeps (mi_sa_device.h "kernel scheduling"); both leave bits + cached fields in HBM.
#include "mi_sa_device.h"

#ifndef MI_NT
#error "compile with -DMI_NT=<4,8,...,44>"
#endif

namespace mi_sa_impl {

namespace {

typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <int NT>
struct MfCfg {
    static constexpr int M = NT / 4;                 // 16x16 tiles per wave
    static constexpr int NPAD = NT * 64;             // padded number of variables
    static constexpr int S = NPAD / 16;              // columns per wave (= 16 * M)
    static constexpr int ROWS = (NPAD + 16) * 4;     // LDS row stride: +64 B makes the A reads conflict-free
    static constexpr int UNITB = 4 * ROWS;           // a unit = 4 rows = the K of one MFMA
    static constexpr int U = 3;
    static constexpr int G = NT / 4;                 // 1 KiB pieces per row
    static constexpr int RING = U * UNITB;
    static constexpr int THR = RING;                 // float thr[4][64][16]
    static constexpr int XB = THR + 4 * 64 * 16 * 4; // uint16 xbits[NPAD]  (bit r = x of replica r)
    static constexpr int SB = XB + NPAD * 2;         // float S[2][4][16]
    static constexpr int FL = SB + 2 * 4 * 16 * 4;   // uint32 flag[2], wg_flips
    static constexpr int TOTAL = FL + 64;
    static constexpr bool ok = TOTAL <= 160 * 1024;
};

__device__ __forceinline__ void lds_dma_16(__amdgpu_buffer_rsrc_t rsrc, char *lds_dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 16,
                                             voff, soff, 0, 0);
}

template <int NT>
__global__ void __launch_bounds__(1024, 4) k_anneal_dense_mfma(DenseArgs a)
{
    using C = MfCfg<NT>;
    __shared__ __attribute__((aligned(16))) char lds[C::TOTAL];
    if (!sched_my_turn(a)) return;

    float *thrbuf = reinterpret_cast<float *>(lds + C::THR);
    unsigned short *xbits = reinterpret_cast<unsigned short *>(lds + C::XB);
    float *Sbuf = reinterpret_cast<float *>(lds + C::SB);
    unsigned int *flag = reinterpret_cast<unsigned int *>(lds + C::FL);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rbase = blockIdx.x * 16;
    const int r_me = rbase + wave;                   // the replica this wave draws random numbers / does I/O for
    const bool active_me = r_me < a.R;
    const uint32_t g_me = a.replica_offset + (uint32_t)r_me;
    const int n = a.n;
    const int rows_used = ((n + 3) >> 2) << 2;       // rows that can hold a proposal, in whole units
    const int total_units = rows_used >> 2;
    const int col0 = wave * C::S;                    // first column of this wave's slab
    const int lr = lane & 15, lq = lane >> 4;        // MFMA lane coordinates: replica / k (or row quarter)

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.Qm), 0, (C::NPAD + 1) * C::NPAD * 4, 0x00020000);

    // ---- state bits -> xbits (one 16-bit word per row, bit r = replica r) via a byte image in the ring area
    {
        unsigned char *img = reinterpret_cast<unsigned char *>(lds);         // [16][NPAD]
        for (int t = 0; t < NT; t += 4) {
            uint32_t w[4] = {0, 0, 0, 0};
            if (!a.init && active_me)
                philox4x32_10((uint32_t)((t >> 2) * 64 + lane), 0u, g_me, 1u, a.seed_lo, a.seed_hi, w);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = (t + c) * 64 + lane;
                unsigned char bit = 0;
                if (active_me && i < n) bit = a.init ? (a.init[(size_t)r_me * n + i] ? 1 : 0) : (unsigned char)(w[c] >> 31);
                img[wave * C::NPAD + i] = bit;
            }
        }
        __syncthreads();
        unsigned short mine[(C::NPAD + 1023) / 1024];
#pragma unroll
        for (int k = 0; k < (C::NPAD + 1023) / 1024; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            unsigned int msk = 0;
            if (i < C::NPAD)
                for (int w2 = 0; w2 < 16; ++w2) msk |= (unsigned int)img[w2 * C::NPAD + i] << w2;
            mine[k] = (unsigned short)msk;
        }
        __syncthreads();                                  // the image (ring area) is dead from here on
#pragma unroll
        for (int k = 0; k < (C::NPAD + 1023) / 1024; ++k) {
            const int i = k * 1024 + (int)threadIdx.x;
            if (i < C::NPAD) xbits[i] = mine[k];
        }
        if (threadIdx.x < 3) flag[threadIdx.x] = 0u;
        __syncthreads();
    }

    // ---- fields: from the previous launch, or diag (the forced pass below then adds the rows with x = 1)
    f32x4acc acc[C::M];
    const bool fields_in = (a.flags & kDenseFieldsIn) != 0;
#pragma unroll
    for (int tau = 0; tau < C::M; ++tau) {
        const int j = col0 + tau * 16 + 4 * lq;          // 4 consecutive columns of replica lr
        if (fields_in) {
            const int rr = rbase + lr;
            acc[tau] = (rr < a.R) ? *reinterpret_cast<const f32x4acc *>(a.fields + (size_t)rr * C::NPAD + j)
                                  : f32x4acc{0, 0, 0, 0};
        } else {
            acc[tau] = *reinterpret_cast<const f32x4acc *>(a.Qm + (size_t)C::NPAD * C::NPAD + j);   // diagonal row
        }
    }

    // ---- ring bookkeeping (wave-uniform, forced into SGPRs) ----
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    int issued = 0, processed = 0, issue_slot = 0, cur_slot = 0;
    auto issue_unit = [&]() {
        if (issued < total_units) {
            if (wave < C::G) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    lds_dma_16(rsrc, lds + issue_slot * C::UNITB + k * C::ROWS + wave * 1024, lane * 16,
                               (issued * 4 + k) * (C::NPAD * 4) + wave * 1024);
            }
            issued = uni(issued + 1);
            issue_slot = uni((issue_slot + 1 == C::U) ? 0 : issue_slot + 1);
        }
    };

    unsigned long long accepted = 0;
    int until_resync = a.resync_first;
    // pass -1 is the forced field-initialisation pass (s_k = x_k, no decisions); it also serves re-syncs
    for (int s = (fields_in ? 0 : -1); s < a.num_sweeps; ++s) {
        bool force = (s < 0);
        if (s >= 0 && a.resync > 0 && --until_resync == 0) {
            // exact re-initialisation: acc = diag, then one forced pass, then redo this sweep index normally
            until_resync = a.resync + 1;                  // the redo of this s decrements once more
#pragma unroll
            for (int tau = 0; tau < C::M; ++tau)
                acc[tau] = *reinterpret_cast<const f32x4acc *>(a.Qm + (size_t)C::NPAD * C::NPAD + col0 + tau * 16 + 4 * lq);
            force = true;
        }
        const float T_me = force ? 1.0f
                                 : __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(
                                       a.temps[a.temps_per_replica ? (active_me ? r_me : 0) : s])));
        // prime the ring for this pass
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issued = 0; processed = 0; issue_slot = 0; cur_slot = 0;
        issue_unit();
        issue_unit();

        bool done = false;
#pragma unroll 1
        for (int wstar = 0; wstar < 16 && !done; ++wstar) {
            static_for<0, C::M>([&](auto tc) {
                constexpr int taustar = decltype(tc)::value;
#pragma unroll 1
                for (int qstar = 0; qstar < 4 && !done; ++qstar) {
                    const int R0 = wstar * C::S + taustar * 16 + qstar * 4;
                    if (R0 >= rows_used) { done = true; break; }        // wave-uniform
                    const int u = R0 >> 2;
                    // thresholds of the next four slots (256 rows): wave w draws for replica w
                    if (!force && (R0 & 255) == 0) {
                        uint32_t w4[4];
                        philox4x32_10((uint32_t)((R0 >> 8) * 64 + lane), (uint32_t)s + a.sweep_offset, g_me, 0u,
                                      a.seed_lo, a.seed_hi, w4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float thr = neglog_u(w4[c]) * T_me;
                            if (R0 + c * 64 + lane >= n || !active_me) thr = -INFINITY;
                            thrbuf[(c * 64 + lane) * 16 + wave] = thr;
                        }
                    }
                    // rendezvous A: ring unit u has landed (every wave waited for its own pieces); also
                    // publishes the thresholds written above
                    if (issued - processed - 1 >= 1) {
                        if (wave < C::G) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // 1 younger unit in flight
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    issue_unit();                                         // into the slot unit u-1 vacated
                    const char *unit = lds + cur_slot * C::UNITB;
                    float *Su = Sbuf + (u & 1) * 64;

                    if (wave == wstar) {                                  // wave-uniform
                        const bool mine = (lq == qstar);                  // the 16 lanes holding these fields
                        const float *urow = reinterpret_cast<const float *>(unit);
                        constexpr int RS = C::ROWS / 4;                   // row stride in floats
                        float t0 = acc[taustar][0], t1 = acc[taustar][1], t2 = acc[taustar][2], t3 = acc[taustar][3];
                        const unsigned int x0w = xbits[R0], x1w = xbits[R0 + 1], x2w = xbits[R0 + 2], x3w = xbits[R0 + 3];
                        const int x0 = (x0w >> lr) & 1, x1 = (x1w >> lr) & 1, x2 = (x2w >> lr) & 1, x3 = (x3w >> lr) & 1;
                        float s0, s1, s2, s3;
                        int a0, a1, a2, a3;
                        if (force) {
                            s0 = (float)x0; s1 = (float)x1; s2 = (float)x2; s3 = (float)x3;
                            a0 = a1 = a2 = a3 = 0;
                        } else {
                            const int ib = ((R0 >> 6) & 3) * 64 + (R0 & 63);
                            const float h0 = thrbuf[(ib + 0) * 16 + lr], h1 = thrbuf[(ib + 1) * 16 + lr];
                            const float h2 = thrbuf[(ib + 2) * 16 + lr], h3 = thrbuf[(ib + 3) * 16 + lr];
                            // couplings inside the unit: c_kk' = Q2[R0+k][R0+k'] (row k of the unit, column R0+k')
                            const float c01 = urow[0 * RS + R0 + 1], c02 = urow[0 * RS + R0 + 2], c03 = urow[0 * RS + R0 + 3];
                            const float c12 = urow[1 * RS + R0 + 2], c13 = urow[1 * RS + R0 + 3];
                            const float c23 = urow[2 * RS + R0 + 3];
                            a0 = mine && ((x0 ? -t0 : t0) < h0);
                            s0 = a0 ? (x0 ? -1.0f : 1.0f) : 0.0f;
                            t1 = __fmaf_rn(c01, s0, t1);
                            a1 = mine && ((x1 ? -t1 : t1) < h1);
                            s1 = a1 ? (x1 ? -1.0f : 1.0f) : 0.0f;
                            t2 = __fmaf_rn(c12, s1, __fmaf_rn(c02, s0, t2));
                            a2 = mine && ((x2 ? -t2 : t2) < h2);
                            s2 = a2 ? (x2 ? -1.0f : 1.0f) : 0.0f;
                            t3 = __fmaf_rn(c23, s2, __fmaf_rn(c13, s1, __fmaf_rn(c03, s0, t3)));
                            a3 = mine && ((x3 ? -t3 : t3) < h3);
                            s3 = a3 ? (x3 ? -1.0f : 1.0f) : 0.0f;
                        }
                        if (mine) { Su[0 * 16 + lr] = s0; Su[1 * 16 + lr] = s1; Su[2 * 16 + lr] = s2; Su[3 * 16 + lr] = s3; }
                        // new state bits of the four rows: ballots over the 16 active lanes of this quarter
                        const int sh = 16 * qstar;
                        const unsigned long long b0 = __ballot(mine && (x0 ^ a0)), b1 = __ballot(mine && (x1 ^ a1));
                        const unsigned long long b2 = __ballot(mine && (x2 ^ a2)), b3 = __ballot(mine && (x3 ^ a3));
                        const unsigned long long f0 = __ballot(a0), f1 = __ballot(a1), f2 = __ballot(a2), f3 = __ballot(a3);
                        const unsigned long long nz = __ballot(mine && ((s0 != 0.0f) | (s1 != 0.0f) | (s2 != 0.0f) | (s3 != 0.0f)));
                        if (mine && lr == 0) {
                            if (!force) {
                                xbits[R0] = (unsigned short)(b0 >> sh); xbits[R0 + 1] = (unsigned short)(b1 >> sh);
                                xbits[R0 + 2] = (unsigned short)(b2 >> sh); xbits[R0 + 3] = (unsigned short)(b3 >> sh);
                            }
                            flag[u & 1] = (nz != 0) ? 1u : 0u;
                        }
                        accepted += (unsigned long long)(__popcll(f0) + __popcll(f1) + __popcll(f2) + __popcll(f3));
                    }
                    // rendezvous B: the signs of unit u are published
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (__builtin_amdgcn_readfirstlane((int)flag[u & 1]) != 0) {
                        const float b = Su[lq * 16 + lr];                                   // B[k = lq][r = lr]
                        const char *arow = unit + lq * C::ROWS + (col0 + lr) * 4;           // A[i = lr][k = lq]
#pragma unroll
                        for (int tau = 0; tau < C::M; ++tau) {
                            const float av = *reinterpret_cast<const float *>(arow + tau * 64);
                            acc[tau] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[tau], 0, 0, 0);
                        }
                    }
                    cur_slot = uni((cur_slot + 1 == C::U) ? 0 : cur_slot + 1);
                    processed = uni(processed + 1);
                }
            });
        }
        if (force && s >= 0) --s;                     // the re-sync pass does not consume a sweep index
    }

    // ---- results: states (wave w = replica w), cached fields, energies, flip count, next kernel ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (active_me) {
        uint8_t *dst = a.states + (size_t)r_me * n;
        for (int t = 0; t < NT; ++t) {
            const int i = t * 64 + lane;
            if (i < n) dst[i] = (uint8_t)((xbits[i] >> wave) & 1);
        }
    }
    if (a.flags & kDenseFieldsOut) {
        const int rr = rbase + lr;
        if (rr < a.R) {
#pragma unroll
            for (int tau = 0; tau < C::M; ++tau)
                *reinterpret_cast<f32x4acc *>(a.fields + (size_t)rr * C::NPAD + col0 + tau * 16 + 4 * lq) = acc[tau];
        }
    }
    if (lane == 0 && accepted) {
        atomicAdd(&a.stats[1], accepted);
        atomicAdd(&flag[2], (unsigned int)accepted);
    }
    if (!(a.flags & kDenseNoEnergy) && active_me) {
        // exact fp64 energy from the slot-permuted matrix (same evaluator as K1 / K1w)
        uint64_t xb = 0;
        for (int t = 0; t < NT; ++t) xb |= (uint64_t)((xbits[t * 64 + lane] >> wave) & 1) << t;
        const int diag_row = ((n + 63) >> 6) * 64;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.Qp), 0, (diag_row + 1) * (NT * 256), 0x00020000);
        const double e = dense_energy_f64<NT>(rp, diag_row, xb, lane);
        if (lane == 0) a.energy[r_me] = e + a.offset;
    }
    __syncthreads();
    if (threadIdx.x == 0) sched_finish(a, flag[2]);
}

}  // namespace

#define MI_CAT2(a, b) a##b
#define MI_CAT(a, b) MI_CAT2(a, b)
int MI_CAT(mi_launch_dense_mfma_nt, MI_NT)(const DenseArgs &a, hipStream_t st)
{
    if constexpr (MfCfg<MI_NT>::ok) {
        if (!a.Qm) return fail(MI_EINVAL, "K1m needs the row-major matrix copy");
        note_kernel("k_anneal_dense_mfma<%d>", MI_NT);
        hipLaunchKernelGGL((k_anneal_dense_mfma<MI_NT>), dim3((a.R + 15) / 16), dim3(1024), 0, st, a);
        HIP_TRY(hipGetLastError());
        return MI_OK;
    } else {
        return fail(MI_EUNSUPPORTED, "K1m LDS plan does not fit for NT=%d", MI_NT);
    }
}

}  // namespace mi_sa_impl
