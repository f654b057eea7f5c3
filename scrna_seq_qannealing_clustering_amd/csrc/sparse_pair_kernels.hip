// sparse_pair_kernels.hip -- K2p: the CSR + uniform-pair anneal with TWO replicas per wavefront (gfx950 only).
//
// Why.  With the accept mask found by fixed-point rounds (sparse_kernels.hip) K2 is bound by the vector-memory
// path: every wavefront re-reads its slot's adjacency (D x 64 x 8 bytes = 8 KB at D = 16) from L2 for every slot of
// every sweep -- 29 TB/s of the ~34.5 TB/s the L2s deliver (profiles/r02_*), 64 B/clk per CU.  The adjacency does
// not depend on the replica, so here a wavefront carries TWO replicas through the same slot: one set of adjacency
// registers, two states, two fields, two thresholds.  L2 traffic per update halves.
//
// State layout: one 32-bit cell per variable in LDS, [half x of replica A | half x of replica B] (0.0 / 1.0), so ONE
// ds_read_b32 per neighbour serves both replicas and v_fma_mix_f32 takes either half as it is (op_sel).  4 bytes per
// variable per wavefront: n <= 4608 keeps 8 wavefronts (16 replicas) per CU.
//
// The chain is K2's (oracle/sa_oracle.c 2b), bit for bit; only models whose every slot is free of internal edges
// (the slot-independent order) take this kernel -- the others run on k_anneal_csr_rank1.
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

typedef _Float16 half_t;

__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(half_t, (uint16_t)w); }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(half_t, (uint16_t)(w >> 16)); }

template <int D>
__global__ void __launch_bounds__(64, 2) k_anneal_csr_rank1_pair(EllArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];      // cell of variable i at byte 4 i
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x;                                    // replicas 2 pair, 2 pair + 1
    const int rA = 2 * pair, rB = 2 * pair + 1;
    if (rA >= a.R) return;
    const bool liveB = rB < a.R;                                    // an odd R leaves the last wavefront one idle seat
    const uint32_t gidA = a.replica_offset + (uint32_t)rA, gidB = gidA + 1u;
    const int n = a.n, slots = a.slots;
    const uint8_t *init = static_cast<const uint8_t *>(a.init);
    uint32_t *cell = reinterpret_cast<uint32_t *>(lds);

    int SA = 0, SB = 0;
    for (int tg = 0; tg * 4 < slots; ++tg) {
        uint32_t wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
        if (!init) {
            philox4x32_10((uint32_t)(tg * 64 + lane), 0u, gidA, 1u, a.seed_lo, a.seed_hi, wa);
            philox4x32_10((uint32_t)(tg * 64 + lane), 0u, gidB, 1u, a.seed_lo, a.seed_hi, wb);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = 4 * tg + c;
            if (t >= slots) break;
            const int i = t * 64 + lane;
            bool xa, xb;
            const bool real = i < n && a.lin[i] < INFINITY;         // (+inf linear term = hole of a padded layout: stays 0)
            if (init) {
                xa = real && init[(size_t)rA * n + i] != 0;
                xb = real && liveB && init[(size_t)rB * n + i] != 0;
            } else {
                xa = real && (wa[c] >> 31);
                xb = real && (wb[c] >> 31);
            }
            cell[i] = (xa ? 0x3c00u : 0u) | (xb ? 0x3c000000u : 0u);
            SA += __popcll(__ballot(xa));
            SB += __popcll(__ballot(xb));
        }
    }

    constexpr int G = D / 4;                                        // groups of four (neighbour, value) per lane
    const __amdgpu_buffer_rsrc_t rs_adj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.adj4), 0, slots * G * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.lin), 0, slots * 256, 0x00020000);
    struct SlotAdj { u32x4 col[G]; u32x4 val[G]; uint32_t lin; };
    const int lane16 = lane * 16;
#ifdef MI_K2P_NOFETCH
    // TIMING-ONLY build (`make dbg`, never the shipped library; wrong chain): the adjacency of slot 0 serves every slot, so
    // the sweep loop issues no vector-memory instruction -- what the kernel would cost without its L2 traffic
    auto fetch_real = [&](int t) {
#else
    auto fetch_adj = [&](int t) {
#endif
        SlotAdj p;
        const int tt = t < slots ? t : slots - 1;
        const int soff = tt * (G * 2048);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            // (constant parts of the offset fold into the instruction's 12-bit immediate)
            const int so = soff + (g / 2) * 4096, io = (g & 1) * 2048;
            p.col[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane16 + io, so, 0);
            p.val[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane16 + io + 1024, so, 0);
        }
        p.lin = __builtin_amdgcn_raw_buffer_load_b32(rs_lin, lane * 4, tt * 256, 0);
        return p;
    };
#ifdef MI_K2P_NOFETCH
    const SlotAdj adj0 = fetch_real(0);
    auto fetch_adj = [&](int) { return adj0; };
#endif

    unsigned long long accepted = 0;
    uint32_t wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
    float TA = 1.0f, TB = 1.0f;
    const float cp = a.c_pair;

    // one slot for both replicas; `wordA` / `wordB` = this slot's random words.
    // Order of a slot: (1) the LDS reads are ISSUED -- the lane's own cell and the 16 neighbour cells; (2) while they are in
    // flight the two thresholds are computed (neglog_u2: the fp32 steps as packed instructions, one for both replicas);
    // (3) one wait; (4) the field sums; (5) the accept masks.  The reads are inline asm (hipcc adds the zero base of the
    // dynamic LDS block to every address it computes itself), so the compiler does not count them: every register they
    // write, and the thresholds, pass THROUGH the wait statement ("+v"), which makes "used only after the wait" a
    // data dependence, not a scheduling accident -- and the wait is lgkmcnt(0), so whatever LDS or scalar-memory
    // operation the compiler may place before it is waited for as well (scripts/check_asm_lds.py checks the emitted
    // code for a read of such a register ahead of its wait at build time).
    auto slot_body = [&](int t, const SlotAdj &cur, uint32_t wordA, uint32_t wordB) {
        const int i = t * 64 + lane;
        uint32_t own;                                               // [x_A | x_B] of this lane's variable
        float gA = __uint_as_float(cur.lin), gB = gA;               // (lanes past n carry lin = +inf: never accepted)
        float thrA = 0.0f, thrB = 0.0f;
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            uint32_t word[16];
            // the packed neighbour word IS the LDS byte address of its cell (one wavefront per workgroup, no static LDS)
            if (g0 == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(own) : "v"(i * 4));
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("ds_read_b32 %0, %1" : "=v"(word[k]) : "v"(cur.col[g0 + k / 4][k & 3]));
            if (g0 == 0) {
                asm volatile("" : "+v"(wordA), "+v"(wordB));        // (keeps the threshold arithmetic behind the reads' issue)
                const f32x2_t thr = neglog_u2(wordA, wordB) * f32x2_t{TA, TB};
                thrA = thr.x;
                thrB = thr.y;
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(word[0]), "+v"(word[1]), "+v"(word[2]), "+v"(word[3]), "+v"(word[4]), "+v"(word[5]),
                               "+v"(word[6]), "+v"(word[7]), "+v"(word[8]), "+v"(word[9]), "+v"(word[10]), "+v"(word[11]),
                               "+v"(word[12]), "+v"(word[13]), "+v"(word[14]), "+v"(word[15]), "+v"(own), "+v"(thrA), "+v"(thrB)
                             :: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(word[0]), "+v"(word[1]), "+v"(word[2]), "+v"(word[3]), "+v"(word[4]), "+v"(word[5]),
                               "+v"(word[6]), "+v"(word[7]), "+v"(word[8]), "+v"(word[9]), "+v"(word[10]), "+v"(word[11]),
                               "+v"(word[12]), "+v"(word[13]), "+v"(word[14]), "+v"(word[15])
                             :: "memory");
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float v = __uint_as_float(cur.val[g0 + k / 4][k & 3]);
                gA = __builtin_fmaf(v, half_lo(word[k]), gA);       // fma(val, x, g): the oracle's conditional add
                gB = __builtin_fmaf(v, half_hi(word[k]), gB);
            }
        }
        const uint32_t xiA = (own >> 13) & 1u, xiB = own >> 29;     // 0x3c00 -> 1
        const uint64_t XA = __ballot((own & 0xffffu) != 0u), XB = __ballot((own >> 16) != 0u);
        const uint32_t sgA = xiA << 31, sgB = xiB << 31;            // dE = x ? -f : f
        const f32x2_t gs = {__uint_as_float(__float_as_uint(gA) ^ sgA), __uint_as_float(__float_as_uint(gB) ^ sgB)};
        const f32x2_t cs = {__uint_as_float(__float_as_uint(cp) ^ sgA), __uint_as_float(__float_as_uint(cp) ^ sgB)};
        const int ownA = SA - (int)xiA, ownB = SB - (int)xiB;
        // accept masks by fixed-point rounds (see k_anneal_csr_rank1): both replicas advance together; the oracle's
        // g + c * (float)(s - x) as one packed multiply and one packed add (no contraction)
        f32x2_t de = gs + cs * f32x2_t{(float)ownA, (float)ownB};
        bool mA = de.x < thrA, mB = de.y < thrB;
        uint64_t AA = __ballot(mA), AB = __ballot(mB);
        if ((AA | AB) != 0ull) {                                    // wave-uniform
            const int baseA = ownA - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(XA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)XA, 0u));
            const int baseB = ownB - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(XB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)XB, 0u));
            for (int round = 0; round < 66; ++round) {
                const uint64_t BA = AA ^ XA, BB = AB ^ XB;
                const int sA = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(BA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)BA, (uint32_t)baseA));
                const int sB = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(BB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)BB, (uint32_t)baseB));
                de = gs + cs * f32x2_t{(float)sA, (float)sB};
                mA = de.x < thrA;
                mB = de.y < thrB;
                const uint64_t NA = __ballot(mA), NB = __ballot(mB);
                const bool same = NA == AA && NB == AB;
                AA = NA;
                AB = NB;
                if (same) break;
            }
            SA += __popcll(AA & ~XA) - __popcll(AA & XA);
            SB += __popcll(AB & ~XB) - __popcll(AB & XB);
            accepted += (unsigned long long)(__popcll(AA) + (liveB ? __popcll(AB) : 0));
            // toggling a state is one XOR of its half; every lane stores its cell (unchanged cells keep their word)
            cell[i] = own ^ (mA ? 0x3c00u : 0u) ^ (mB ? 0x3c000000u : 0u);
        }
    };

    for (int s = 0; s < a.num_sweeps; ++s) {
        if (a.temps_per_replica) {
            TA = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[rA])));
            TB = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[liveB ? rB : rA])));
        } else {
            TA = TB = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[s])));
        }
        const uint32_t sw = (uint32_t)s + a.sweep_offset;
        // four slots per trip (one Philox block per replica), the adjacency one slot ahead in two register sets
        SlotAdj P = fetch_adj(0), Q;
#pragma unroll 1
        for (int t = 0; t < slots; t += 4) {
            philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gidA, 0u, a.seed_lo, a.seed_hi, wa);
            philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gidB, 0u, a.seed_lo, a.seed_hi, wb);
            Q = fetch_adj(t + 1);
            slot_body(t, P, wa[0], wb[0]);
            if (t + 1 < slots) {                                    // wave-uniform
                P = fetch_adj(t + 2);
                slot_body(t + 1, Q, wa[1], wb[1]);
                if (t + 2 < slots) {
                    Q = fetch_adj(t + 3);
                    slot_body(t + 2, P, wa[2], wb[2]);
                    P = fetch_adj(t + 4);
                    if (t + 3 < slots) slot_body(t + 3, Q, wa[3], wb[3]);
                }
            }
        }
    }

    // ---- epilogue: states out, exact fp64 energies (same sums as k_anneal_csr_rank1) ----
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1 && !liveB) break;
        const int r = rep ? rB : rA;
        const int sh = rep ? 16 : 0;
        uint8_t *dst = static_cast<uint8_t *>(a.states) + (size_t)r * n;
        int cnt = 0;
        double e = 0.0;
        for (int t = 0; t < slots; ++t) {
            const int i = t * 64 + lane;
            const bool on = ((cell[i] >> sh) & 0xffffu) != 0u;
            if (i < n) dst[i] = (uint8_t)on;
            cnt += __popcll(__ballot(on));
            if (!on) continue;
            double acc = 0.0;
            for (int k = 0; k < D; ++k) {
                const size_t at = ((size_t)t * D + k) * 64 + lane;
                const uint32_t cc = a.ell_col[at];
                const double vv = a.ell_val64 ? a.ell_val64[at] : (double)a.ell_val[at];
                if (((cell[cc] >> sh) & 0xffffu) != 0u) acc += vv;
            }
            e += (a.lin64 ? a.lin64[i] : (double)a.lin[i]) + 0.5 * acc;
        }
        e = wave_sum_f64(e);
        if (lane == 0) {
            const double cp64 = a.ell_val64 ? a.c_pair64 : (double)a.c_pair;
            a.energy[r] = e + cp64 * 0.5 * (double)cnt * (double)(cnt - 1) + a.offset;
        }
    }
    if (lane == 0) atomicAdd(&a.stats[1], accepted);
}

template <typename KernelT>
int launch_pair(KernelT kernel, const EllArgs &a, hipStream_t st)
{
    const size_t lds = (size_t)a.slots * 256;                      // 4 bytes per variable
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "csr_rank1 pair kernel: n = %d exceeds the state LDS budget", a.n);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel("k_anneal_csr_rank1_pair<%d>", a.D);
    hipLaunchKernelGGL(kernel, dim3((a.R + 1) / 2), dim3(64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

// a.adj4 must hold the pair packing (neighbour word = 4 * index)
int mi_launch_csr_rank1_pair(const EllArgs &a, hipStream_t st)
{
    if (!a.adj4) return fail(MI_EHIP, "csr_rank1 pair kernel: packed adjacency missing");
    if (a.D == 16) return launch_pair(k_anneal_csr_rank1_pair<16>, a, st);
    if (a.D == 32) return launch_pair(k_anneal_csr_rank1_pair<32>, a, st);
    return fail(MI_EUNSUPPORTED, "csr_rank1 pair kernel: slot-ELL width %d not built", a.D);
}

}  // namespace mi_sa_impl
