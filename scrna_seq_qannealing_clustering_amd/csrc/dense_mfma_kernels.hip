// dense_mfma_kernels.hip -- K1m: the dense-QUBO chain with the row updates on the matrix cores (gfx950).
//
// Same chain as K1 / K1w, bit for bit; what changes is WHO holds the cached fields.
// A workgroup = 16 wavefronts = 16 replicas, but the fields of all 16 replicas live TRANSPOSED, as the
// accumulators of f32-input MFMAs.  The n_pad = 64*NT columns are cut into blocks of 16; block c belongs to
// wave c % 16 as its tile c / 16 (round robin, so consecutive blocks sit on different SIMDs):
//       acc[tile][reg] of lane l = f_{column 16*c + 4*(l >> 4) + reg}(replica l & 15).
// The flips that 4 consecutive rows R..R+3 cause in the 16 replicas are ONE rank-4 update
//       F[:, r] += sum_k Q2[R+k][:] * s_k(r),      s_k(r) in {-1, 0, +1},
// i.e. per tile one v_mfma_f32_16x16x4_f32 with A[i][k] = Q2[R+k][column i] (read from the LDS ring, 4 B per
// lane) and B[k][r] = s_k(r).  The f32-input MFMA is an exact fp32 fmaf chain in k order
//       D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// which is precisely the oracle's "f += sgn * Q2" applied for the accepted rows in row order (a rejected
// row has s = 0 and fma(a, 0, C) = C).  A Q row read from LDS once serves all 16 replicas: LDS traffic per
// 4 rows is 45 KB flat instead of (accepted flips) x 11 KB, and the kernel is bound by the fp32 MFMA rate
// (256 flop/clk/CU: 2*16*n_pad^2 flop per sweep and workgroup) instead of the LDS->VGPR fill rate that bounds
// K1w at the hot end of a schedule.
//
// Schedule (round 2).  Decisions are taken per BLOCK of 16 rows, updates applied per UNIT of 4 rows (the K of
// one MFMA; the LDS ring holds 3 units of 4 x 11 KB):
//   * DIAG(c): the owner wave of block c walks its 16 rows sequentially for the 16 replicas at once.  Lane
//     4r + q holds the fields t_j of replica r for the columns j = q, 4+q, 8+q, 12+q of the block; step k
//     decides row k in the lanes q = k % 4, broadcasts s_k inside the quad (DPP) and applies the couplings
//     Q2[16c+k][16c+j] -- the same fmaf chain, in the same order, that the MFMAs apply to the accumulators
//     afterwards (fields of rows already decided are dead and may take garbage).  It leaves s_k(r) as the B
//     operands of the block's four MFMA units, the new state bits, and a 4-bit "unit has a flip" word.
//   * look-ahead: DIAG(c+1) runs in the LAST unit of block c, on a copy of the accumulator tile taken one unit
//     earlier plus that unit's four rows applied by hand -- so it overlaps the MFMAs of the other 15 waves
//     and the chain never waits for a whole panel update.  One s_barrier per unit (it was two, with the
//     decisions of 4 rows serialised between them: 2900 cycles per unit, now the 16 x 11 MFMAs = 1408).
//   * the 16x16 coupling block of DIAG(c+1) is fetched by LDS-DMA three units ahead (wave 15, which loads no
//     ring piece), transposed on the way ([k][q][m], one ds_read_b128 per step).
// Thresholds come from the same Philox addressing as everywhere else (wave w draws for replica w, 4 slots =
// 256 rows per block of random numbers), through LDS.
//
// Used for the hot part of a schedule (acceptance above ~10 %): the launcher alternates K1m and K1w per
// chunk of sweeps (mi_sa_device.h "kernel scheduling"); both leave bits + cached fields in HBM.
#include "mi_sa_device.h"

#ifndef MI_NT
#error "compile with -DMI_NT=<4,8,...,44>"
#endif

// diagnostic timing builds (-DMI_K1M_DEBUG; results are wrong): DenseArgs::debug bit0 = no ring DMA, bit2 = no
// decision chain, bit3 = no MFMAs, bit5 = no hand-applied rows (diag_pre).  -DMI_K1M_TICKS: s_memtime phase counters of the wave `debug >> 8` in stats[4..13]
// (per phase g: the rendezvous; operand reads; MFMA issue; what follows them -- printed by scripts/perf_k1m.py).
#ifdef MI_K1M_DEBUG
#define K1M_DBG(bit) ((a.debug & (bit)) != 0)
#else
#define K1M_DBG(bit) false
#endif

namespace mi_sa_impl {

namespace {

typedef float f32x4acc __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const float lds_f32;
typedef __attribute__((address_space(3))) const f32x4acc lds_f32x4;

template <int NT>
struct MfCfg {
    static constexpr int M = NT / 4;                 // 16x16 tiles per wave
    static constexpr int NPAD = NT * 64;             // padded number of variables
    static constexpr int NB = NPAD / 16;             // 16-row blocks (= 16 * M)
    static constexpr int ROWS = (NPAD + 16) * 4;     // LDS row stride: +64 B makes the A reads conflict-free
    static constexpr int UNITB = 4 * ROWS;           // a unit = 4 rows = the K of one MFMA
    static constexpr int U = 3;
    static constexpr int G = NT / 4;                 // 1 KiB pieces per row
    static constexpr int RING = U * UNITB;
    static constexpr int THRS = 260;                 // floats per replica: 256 rows + 4 (conflict-free reads)
    static constexpr int THR = RING;                 // float thr[16][THRS]
    static constexpr int XR = THR + 16 * THRS * 4;   // uint16 xr[NB][16]: bit k = x of row 16*block + k
    static constexpr int SB = XR + NB * 16 * 2;      // float S[2][4][64]: B operands of a block's 4 units
    static constexpr int CB = SB + 2 * 256 * 4;      // float C[16][4][4]: coupling block, [k][q][m]
    static constexpr int EB = CB + 256 * 4;          // float E[16][4][4]: previous block's rows x this block's columns
    static constexpr int TB = EB + 256 * 4;          // float T[16][20]: accumulator tile on its way to DIAG
    static constexpr int FL = TB + 16 * 20 * 4;      // uint32 flag[2] (units with a flip), wg_flips
    static constexpr int TOTAL = FL + 16;
    static constexpr bool ok = TOTAL <= 160 * 1024 && G <= 15;
};

__device__ __forceinline__ void lds_dma_16(__amdgpu_buffer_rsrc_t rsrc, char *lds_dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 16,
                                             voff, soff, 0, 0);
}

// 64 lanes x 4 B, each lane from its own offset (a gather), to lds_dst + lane*4
__device__ __forceinline__ void lds_dma_4(__amdgpu_buffer_rsrc_t rsrc, char *lds_dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 4,
                                             voff, soff, 0, 0);
}

template <int QK>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), QK * 0x55, 0xf, 0xf, false));
}

template <int CTRL>
__device__ __forceinline__ unsigned int quad_perm(unsigned int v)
{
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

template <int NT>
__global__ void __launch_bounds__(1024, 4) k_anneal_dense_mfma(DenseArgs a)
{
    using C = MfCfg<NT>;
    __shared__ __attribute__((aligned(16))) char lds[C::TOTAL];
    if (!sched_my_turn(a)) return;

    float *thr = reinterpret_cast<float *>(lds + C::THR);
    unsigned short *xr = reinterpret_cast<unsigned short *>(lds + C::XR);
    float *Sbuf = reinterpret_cast<float *>(lds + C::SB);
    float *Cbuf = reinterpret_cast<float *>(lds + C::CB);
    float *Ebuf = reinterpret_cast<float *>(lds + C::EB);
    float *Tbuf = reinterpret_cast<float *>(lds + C::TB);
    unsigned int *flag = reinterpret_cast<unsigned int *>(lds + C::FL);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rbase = blockIdx.x * 16;
    const int r_me = rbase + wave;                   // the replica this wave draws random numbers / does I/O for
    const bool active_me = r_me < a.R;
    const uint32_t g_me = a.replica_offset + (uint32_t)r_me;
    const int n = a.n;
    const int nbu = (n + 15) >> 4;                   // blocks that can hold a proposal
    const int units_per_pass = 4 * nbu;
    const bool last_tile_live = (C::M - 1) * 16 + wave < nbu;   // does this wave's last tile hold live columns
    const int lr = lane & 15, lq = lane >> 4;        // MFMA lane coordinates: replica / k (or column quarter)
    const int dr = lane >> 2, dq = lane & 3;         // DIAG lane coordinates: replica / column residue
    const bool loader = wave < 4;                    // waves 0..3 stream the ring (row = wave of every unit)

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.Qm), 0, (C::NPAD + 1) * C::NPAD * 4, 0x00020000);

    // ---- state bits -> xr (one 16-bit word per block and replica): wave w packs replica w ----
    for (int t = 0; t < NT; t += 4) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (!a.init && active_me)
            philox4x32_10((uint32_t)((t >> 2) * 64 + lane), 0u, g_me, 1u, a.seed_lo, a.seed_hi, w);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = (t + c) * 64 + lane;
            bool bit = false;
            if (active_me && i < n) bit = a.init ? (a.init[(size_t)r_me * n + i] != 0) : ((w[c] >> 31) != 0);
            const unsigned long long bal = __ballot(bit);
            if (lane < 4) xr[((t + c) * 4 + lane) * 16 + wave] = (unsigned short)(bal >> (16 * lane));
        }
    }
    if (threadIdx.x < 3) flag[threadIdx.x] = 0u;

    // ---- fields: from the previous launch, or diag (the forced pass below then adds the rows with x = 1)
    f32x4acc acc[C::M];
    const bool fields_in = (a.flags & kDenseFieldsIn) != 0;
    // the compiler waits for a load where its result is first used -- for these that would be a vmcnt wait in
    // front of the MFMAs, which drains the ring's LDS-DMA queue with it: consume the loads here
    auto pin_acc = [&]() {
#pragma unroll
        for (int tau = 0; tau < C::M; ++tau) asm volatile("" : "+v"(acc[tau]));
    };
    auto load_diag = [&]() {
        // (through the buffer resource: flat pointers would sit in 2 VGPRs per tile across the whole kernel)
#pragma unroll
        for (int tau = 0; tau < C::M; ++tau) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (wave * 16 + 4 * lq) * 4,
                                                                  C::NPAD * C::NPAD * 4 + tau * 1024, 0);
            acc[tau] = f32x4acc{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w)};
        }
        pin_acc();
    };
    if (fields_in) {
        const int rr = rbase + lr;
#pragma unroll
        for (int tau = 0; tau < C::M; ++tau)
            acc[tau] = (rr < a.R) ? *reinterpret_cast<const f32x4acc *>(a.fields + (size_t)rr * C::NPAD + (tau * 16 + wave) * 16 + 4 * lq)
                                  : f32x4acc{0, 0, 0, 0};
        pin_acc();
    } else {
        load_diag();
    }

    // ---- the passes of this launch: (sweep index, forced).  A forced pass takes no decisions, its "flips" are
    // the set bits (s_k = x_k): it rebuilds the fields from the diagonal -- at the start, and at every re-sync,
    // after which the same sweep index runs normally (same bookkeeping as K1 / K1w)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    int gen_s = fields_in ? 0 : -1, gen_until = a.resync_first;
    auto next_pass = [&](int &s_out, bool &force_out) -> bool {
        if (gen_s >= a.num_sweeps) return false;
        bool force = gen_s < 0;
        if (gen_s >= 0 && a.resync > 0 && --gen_until == 0) { gen_until = a.resync + 1; force = true; }
        s_out = gen_s; force_out = force;
        gen_s = (force && gen_s >= 0) ? gen_s : gen_s + 1;
        return true;
    };
    int total_passes = 0;
    {
        int s0, keep_s = gen_s, keep_u = gen_until; bool f0;
        while (next_pass(s0, f0)) ++total_passes;
        gen_s = keep_s; gen_until = keep_u;
    }
    const int total_units = total_passes * units_per_pass;

    // ---- ring bookkeeping (wave-uniform, forced into SGPRs); the stream runs on across passes ----
    int issued = 0, processed = 0, issue_slot = 0, cur_slot = 0, issue_row = 0;
    // (four loader waves -- one per SIMD -- bring in one row of the unit each, all G pieces of it: the bookkeeping
    // of the stream is then paid once per SIMD and overlaps the MFMAs of the three other waves there; every
    // instruction in this loop costs its issue slot in all waves that execute it)
    auto issue_unit = [&]() {
        if (issued < total_units) {
            if (!K1M_DBG(1)) {
#pragma unroll
                for (int p = 0; p < C::G; ++p)
                    lds_dma_16(rsrc, lds + issue_slot * C::UNITB + wave * C::ROWS + p * 1024, lane * 16,
                               (issue_row + wave) * (C::NPAD * 4) + p * 1024);
            }
            issued = uni(issued + 1);
            issue_row = uni(issue_row + 4 >= units_per_pass * 4 ? 0 : issue_row + 4);
            issue_slot = uni((issue_slot + 1 == C::U) ? 0 : issue_slot + 1);
        }
    };
    // a 16 x 16 block of Q2, gathered (and transposed on the way) as dst[k][q][m] = Q2[16rb + k][16cb + 4m + q]
    auto issue_block = [&](int dst, int rb, int cb) {
        // word 64i + lane of dst: k = 4i + (lane >> 4), q = (lane >> 2) & 3, m = lane & 3
        const int lane_off = ((lane >> 4) * C::NPAD + 4 * (lane & 3) + ((lane >> 2) & 3)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            lds_dma_4(rsrc, lds + dst + i * 256, lane_off, ((16 * rb + 4 * i) * C::NPAD + 16 * cb) * 4);
    };
    auto draw_thresholds = [&](int s_pass, int grp) {
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(
            a.temps[a.temps_per_replica ? (active_me ? r_me : 0) : s_pass])));
        uint32_t w4[4];
        philox4x32_10((uint32_t)(grp * 64 + lane), (uint32_t)s_pass + a.sweep_offset, g_me, 0u, a.seed_lo, a.seed_hi, w4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float th = neglog_u(w4[c]) * T;
            if (grp * 256 + c * 64 + lane >= n || !active_me) th = -INFINITY;
            thr[wave * C::THRS + c * 64 + lane] = th;
        }
    };
    auto dump_tile = [&](int tile) {                 // acc[tile] -> T[replica][column], for the wave's own DIAG
        static_for<0, C::M>([&](auto tc) {
            constexpr int tau = decltype(tc)::value;
            if (tau == tile) *reinterpret_cast<f32x4acc *>(Tbuf + lr * 20 + 4 * lq) = acc[tau];
        });
    };

    unsigned long long accepted = 0;

    // ---- DIAG of block nb (by its owner wave): the decisions of 16 rows x 16 replicas, in four pieces that run
    // in the four units of the block BEFORE it (state in d_*; only the owner's copy means anything):
    //   begin   the accumulator tile as it is at the start of the previous block -> t (DIAG lane layout)
    //   pre     the previous block's 16 rows applied by hand: its signs S are known, its couplings to this
    //           block's columns come from E -- the same fmaf chain the MFMAs apply to the accumulators meanwhile
    //   steps   the sequential walk over the 16 rows
    //   end     publishes the B operands, the state bits and the "unit has a flip" word
    float d_tq[4], d_hq[4], d_sown[4];
    unsigned int d_sx[4], d_xw = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) { d_tq[m] = 0.0f; d_hq[m] = 0.0f; d_sown[m] = 0.0f; d_sx[m] = 0u; }
    auto diag_begin = [&](int nb) {
        d_xw = xr[nb * 16 + dr];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            d_tq[m] = Tbuf[dr * 20 + 4 * m + dq];
            d_sx[m] = (d_xw << (31 - 4 * m - dq)) & 0x80000000u;
            d_sown[m] = 0.0f;
        }
    };
    // (the LDS bases go through an opaque copy so that every access below is base register + immediate offset:
    // left alone, the compiler materialises one address VGPR per row, hoists them all out of the loops, spills
    // them -- and every scratch reload is a vmcnt(0) wait that drains the ring's DMA queue)
    auto opaque = [](const float *p) {
        lds_f32 *q = (lds_f32 *)p;
        asm volatile("" : "+v"(q));
        return q;
    };
    auto diag_pre = [&](int gb_cur, auto k0c, auto k1c) {
        constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
        if (K1M_DBG(32)) return;
        lds_f32 *Sp = opaque(Sbuf + (gb_cur & 1) * 256 + dr);
        lds_f32 *Ep = opaque(Ebuf + dq * 4);
        // (the eight signs first, the coupling rows three LDS round trips deep: row by row with a full wait each, the
        // piece was eight LDS latencies long -- and it is the owner that the other fifteen waves wait for)
        float sv[K1 - K0];
        static_for<K0, K1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            sv[k - K0] = Sp[(k >> 2) * 64 + (k & 3) * 16];
        });
        f32x4acc e4[3];
        e4[0] = *(lds_f32x4 *)(Ep + K0 * 16);
        e4[1] = *(lds_f32x4 *)(Ep + (K0 + 1) * 16);
        static_for<K0, K1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k + 2 < K1) e4[(k + 2 - K0) % 3] = *(lds_f32x4 *)(Ep + (k + 2) * 16);
            const f32x4acc e = e4[(k - K0) % 3];
#pragma unroll
            for (int m = 0; m < 4; ++m) d_tq[m] = __fmaf_rn(e[m], sv[k - K0], d_tq[m]);
        });
    };
    auto diag_thresholds = [&](int nb) {
#pragma unroll
        for (int m = 0; m < 4; ++m) d_hq[m] = thr[dr * C::THRS + ((16 * nb + 4 * m + dq) & 255)];
    };
    auto diag_steps = [&](auto k0c, auto k1c) {
        constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
        if (K1M_DBG(4)) return;
        lds_f32 *Cp = opaque(Cbuf + dq * 4);
        f32x4acc cq[3];                                   // rows k, k+1, k+2: two LDS round trips stay in flight
        cq[0] = *(lds_f32x4 *)(Cp + K0 * 16);
        cq[1] = *(lds_f32x4 *)(Cp + (K0 + 1) * 16);
        static_for<K0, K1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int mk = k >> 2, qk = k & 3;
            if constexpr (k + 2 < K1) cq[(k + 2 - K0) % 3] = *(lds_f32x4 *)(Cp + (k + 2) * 16);
            const f32x4acc c4 = cq[(k - K0) % 3];
            const bool d = __uint_as_float(__float_as_uint(d_tq[mk]) ^ d_sx[mk]) < d_hq[mk];
            const float so = d ? __uint_as_float(0x3f800000u ^ d_sx[mk]) : 0.0f;
            const float sk = quad_bcast<qk>(so);
            if (dq == qk) d_sown[mk] = so;
#pragma unroll
            for (int m = mk; m < 4; ++m) d_tq[m] = __fmaf_rn(c4[m], sk, d_tq[m]);    // (m = mk: dead fields, see top)
        });
    };
    auto diag_end = [&](int nb, bool force, int gbn) {
        if (force) {
            d_xw = xr[nb * 16 + dr];
#pragma unroll
            for (int m = 0; m < 4; ++m) d_sown[m] = ((d_xw >> (4 * m + dq)) & 1u) ? 1.0f : 0.0f;
        }
        float *Sn = Sbuf + (gbn & 1) * 256;
        unsigned int nz = 0, word = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            Sn[m * 64 + dq * 16 + dr] = d_sown[m];
            const unsigned long long bal = __ballot(d_sown[m] != 0.0f);
            if (bal) nz |= 1u << m;
            if (!force) {
                accepted += (unsigned long long)__popcll(bal);
                word |= (d_sown[m] != 0.0f ? 1u : 0u) << (4 * m + dq);
            }
        }
        if (!force) {
            word |= quad_perm<0xB1>(word);
            word |= quad_perm<0x4E>(word);
            if (dq == 0) xr[nb * 16 + dr] = (unsigned short)(d_xw ^ word);
        }
        if (lane == 0) flag[gbn & 1] = nz;
    };
    using I0 = std::integral_constant<int, 0>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;

    int cur_s = 0, nxt_s = 0;
    bool cur_force = false, nxt_force = false;
    bool have_cur = next_pass(cur_s, cur_force);
    __syncthreads();                                  // xr / flag initialised
    if (have_cur) {
        // prologue: thresholds + couplings of block 0, the first two ring units, DIAG(0) by wave 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // from here on only LDS-DMA is in the VM queue
        if (!cur_force) draw_thresholds(cur_s, 0);
        if (wave == 15) {
            if (!cur_force) issue_block(C::CB, 0, 0);
            issue_block(C::EB, 0, nbu > 1 ? 1 : 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (loader) { issue_unit(); issue_unit(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0) {
            if (!cur_force) {
                dump_tile(0);
                diag_begin(0);
                diag_thresholds(0);
                diag_steps(I0{}, I16{});
            }
            diag_end(0, cur_force, 0);
        }
    }
    int gb = 0;                                       // running block count
#ifdef MI_K1M_TICKS
    unsigned long long tick_ = __builtin_amdgcn_s_memtime(), t_wait[4] = {0, 0, 0, 0}, t_body[4] = {0, 0, 0, 0}, t_diag = 0, t_thr = 0;
#define K1M_TICK(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tick_; tick_ = now_; } while (0)
#else
#define K1M_TICK(var) do { } while (0)
#endif
    float av[C::M], bv = 0.0f;                        // MFMA operands of the current unit
#pragma unroll
    for (int tau = 0; tau < C::M; ++tau) av[tau] = 0.0f;
    auto run_mfmas = [&]() {
#pragma unroll
        for (int tau = 0; tau < C::M; ++tau) {
            // (n_pad - n < 256 columns = one tile per wave: only the last tile can be all padding)
            if (tau < C::M - 1 || last_tile_live)
                acc[tau] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tau], bv, acc[tau], 0, 0, 0);
        }
    };
    while (have_cur) {
        const bool have_nxt = next_pass(nxt_s, nxt_force);
        if (cur_force && cur_s >= 0) load_diag();     // re-sync: rebuild from the diagonal
#pragma unroll 1
        for (int b = 0; b < nbu; ++b) {
            const bool wrap = (b + 1 == nbu);
            const bool nb_valid = !wrap || have_nxt;
            const int nb = wrap ? 0 : b + 1;
            const int np_s = wrap ? nxt_s : cur_s;
            const bool np_force = wrap ? nxt_force : cur_force;
            const bool own_next = nb_valid && wave == (nb & 15);
            const bool own_chain = own_next && !np_force;
            unsigned int nzw = 0;                     // bit g: unit g of this block has a flip in some replica
            static_for<0, 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                // rendezvous: ring unit landed (every loader waited for its own pieces), previous unit consumed,
                // everything the owner of this block published is visible
                if ((loader)) {
                    if (issued - processed >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::G) : "memory");   // 1 younger unit in flight
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if ((g == 0 || g == 2) && (wave == 15)) {
                    // wave 15: E of this block (issued in unit 2 of the previous one) / C of the next (in unit 0)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                K1M_TICK(t_wait[g]);
                const char *unit = lds + cur_slot * C::UNITB;
                if constexpr (g == 0) nzw = (unsigned int)__builtin_amdgcn_readfirstlane((int)flag[gb & 1]);
                const bool mfma_on = (((nzw >> g) & 1u) && !K1M_DBG(8)) || K1M_DBG(16);   // (bit4: MFMAs regardless)
                if (mfma_on) {
                    bv = Sbuf[(gb & 1) * 256 + g * 64 + lane];                            // B[k = lq][r = lr]
                    const char *arow = unit + lq * C::ROWS + (wave * 16 + lr) * 4;       // A[i = lr][k = lq]
#pragma unroll
                    for (int tau = 0; tau < C::M; ++tau) av[tau] = *reinterpret_cast<const float *>(arow + tau * 1024);
                }
                // (the copy of the owner's tile for DIAG(nb): as it is BEFORE this block's rows go in)
                if constexpr (g == 0) { if ((own_chain)) dump_tile(nb >> 4); }
                K1M_TICK(t_thr);
                if (mfma_on) run_mfmas();
                K1M_TICK(t_body[g]);
                // everything else AFTER the MFMAs are queued: the stream's bookkeeping (loader waves) and the owner's
                // piece of DIAG(nb) run under the matrix pipe, not in front of it
                if ((loader)) issue_unit();                             // into the slot the previous unit vacated
                if ((own_next)) {                             // a quarter of DIAG(nb) per unit
                    if constexpr (g == 0) { if (own_chain) { diag_begin(nb); diag_pre(gb, I0{}, I8{}); } }
                    if constexpr (g == 1) { if (own_chain) diag_pre(gb, I8{}, I16{}); }
                    if constexpr (g == 2) { if (own_chain) { diag_thresholds(nb); diag_steps(I0{}, I8{}); } }
                    if constexpr (g == 3) { if (own_chain) diag_steps(I8{}, I16{}); diag_end(nb, np_force, gb + 1); }
                }
                if constexpr (g == 0) {
                    if (nb_valid && !np_force) {
                        if ((wave == 15)) issue_block(C::CB, nb, nb);
                        if (((nb & 15) == 0)) {
                            // (opaque copy: LICM otherwise hoists the whole Philox block out of this branch
                            // and runs it once per BLOCK in all 16 waves)
                            int grp = nb >> 4;
                            asm volatile("" : "+s"(grp));
                            draw_thresholds(np_s, grp);
                        }
                    }
                }
                if constexpr (g == 2) {
                    // E for the block after this one: rows of block nb x columns of ITS successor (the E of this
                    // block is read in units 0 and 1: not before the rendezvous of unit 2)
                    if ((nb_valid && wave == 15)) issue_block(C::EB, nb, nb + 1 == nbu ? 0 : nb + 1);
                }
                cur_slot = uni((cur_slot + 1 == C::U) ? 0 : cur_slot + 1);
                processed = uni(processed + 1);
                K1M_TICK(t_diag);
            });
            gb = uni(gb + 1);
        }
        cur_s = nxt_s; cur_force = nxt_force; have_cur = have_nxt;
    }

    // ---- results: states (wave w = replica w), cached fields, energies, flip count, next kernel ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (active_me) {
        uint8_t *dst = a.states + (size_t)r_me * n;
        for (int t = 0; t < NT; ++t) {
            const int i = t * 64 + lane;
            if (i < n) dst[i] = (uint8_t)((xr[(i >> 4) * 16 + wave] >> (i & 15)) & 1);
        }
    }
    if (a.flags & kDenseFieldsOut) {
        const int rr = rbase + lr;
        if (rr < a.R) {
#pragma unroll
            for (int tau = 0; tau < C::M; ++tau)
                *reinterpret_cast<f32x4acc *>(a.fields + (size_t)rr * C::NPAD + (tau * 16 + wave) * 16 + 4 * lq) = acc[tau];
        }
    }
#ifdef MI_K1M_TICKS
    if (lane == 0 && blockIdx.x == 0 && wave == (a.debug >> 8)) {     // the wave picked by debug bits 8..11
        for (int g = 0; g < 4; ++g) { atomicAdd(&a.stats[4 + g], t_wait[g]); atomicAdd(&a.stats[8 + g], t_body[g]); }
        atomicAdd(&a.stats[12], t_thr);               // bookkeeping / loads after the rendezvous
        atomicAdd(&a.stats[13], t_diag);              // its DIAG pieces
    }
#endif
    if (lane == 0 && accepted) {
        atomicAdd(&a.stats[1], accepted);
        atomicAdd(&flag[2], (unsigned int)accepted);
    }
    if (!(a.flags & kDenseNoEnergy) && active_me) {
        // exact fp64 energy from the slot-permuted matrix (same evaluator as K1 / K1w)
        uint64_t xb = 0;
        for (int t = 0; t < NT; ++t) {
            const int i = t * 64 + lane;
            xb |= (uint64_t)((xr[(i >> 4) * 16 + wave] >> (i & 15)) & 1) << t;
        }
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
