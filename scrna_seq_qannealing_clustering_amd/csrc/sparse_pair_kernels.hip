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

// phase cycle counters of the sweeping wavefront (`make dbg DBG_FLAGS=-DMI_K2P_PROFILE DBG_NAME=pprof`; perturbs the run)
#ifdef MI_K2P_PROFILE
#define K2P_TICK(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tick_; tick_ = now_; } while (0)
#else
#define K2P_TICK(var) do { } while (0)
#endif

__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(half_t, (uint16_t)w); }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(half_t, (uint16_t)(w >> 16)); }

// TW ("threshold wavefront", round 3): the workgroup has a SECOND wavefront that does the part of a sweep which does not
// depend on the state -- the random words (Philox) and the thresholds -ln(u) * T of both replicas -- one group of four
// slots ahead, and hands them over through a two-deep ring in LDS (one ds_write_b64 / ds_read_b64 per slot and lane,
// one s_barrier per group).  Why: with two wavefronts per SIMD the kernel is bound by what ONE wavefront can issue
// (a lone wavefront issues a vector instruction every 5.8 cycles, scripts/ubench_valu.hip; the SIMD could take one every
// 1.5-2), and a third of the instructions of a slot are these thresholds.  A 128-thread workgroup puts its two
// wavefronts on different SIMDs and every SIMD ends up with two sweeping and two threshold wavefronts
// (scripts/probe_placement.hip).  Same chain: the thresholds are the same function of (variable, sweep, replica).
// WGT: the model carries pair-term weights (mi_sa_problem_set_pair_weights; weighted_slot_sweep): a template switch, so
// that the other models' code is what it was.
template <int D, bool TW, bool WGT = false>
__global__ void __launch_bounds__(TW ? 128 : 64, TW ? 1 : 2) k_anneal_csr_rank1_pair(EllArgs a)
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
    // TW: the ring of thresholds behind the cells: 2 groups x 4 slots x 64 lanes x (thrA, thrB)
    const uint32_t ring_lane = (uint32_t)slots * 256u + (uint32_t)lane * 8u;

    if constexpr (TW) {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1) {
            // ---- the threshold wavefront ----
            uint32_t wa[4], wb[4];
            uint32_t buf = 0;
            for (int s = 0; s < a.num_sweeps; ++s) {
                float TA, TB;
                if (a.temps_per_replica) {
                    TA = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[rA])));
                    TB = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[liveB ? rB : rA])));
                } else {
                    TA = TB = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[s])));
                }
                const uint32_t sw = (uint32_t)s + a.sweep_offset;
#pragma unroll 1
                for (int t = 0; t < slots; t += 4) {
#ifdef MI_K2P_DBG_NOPROD   /* timing only: no random words */
                    for (int c = 0; c < 4; ++c) wa[c] = wb[c] = (uint32_t)(t + c) * 0x9E3779B9u + lane * 77u;
#else
                    philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gidA, 0u, a.seed_lo, a.seed_hi, wa);
                    philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gidB, 0u, a.seed_lo, a.seed_hi, wb);
#endif
                    const uint32_t at = ring_lane + buf;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x2_t thr = neglog_u2(wa[c], wb[c]) * f32x2_t{TA, TB};
                        asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(at), "v"(thr), "n"(c * 512) : "memory");
                    }
                    buf ^= 2048u;
                    // group (s, t) is in the ring: the sweeping wavefront passes the same barrier before it reads it, and
                    // passes the NEXT one only after it has read it -- so the buffer written next (the other one) is free
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            return;
        }
    }

    // (pair-term weights, mi_sa_problem_set_pair_weights: SA / SB are sum_j w_j x_j; the one slot whose lanes carry weights
    // other than 1 is swept by a serial loop -- weighted_slot_sweep)
    const int wslot = WGT ? a.wslot : -1;
    const int wl = wslot >= 0 ? a.wgt[lane] : 0;
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
            if (WGT && t == wslot) {
                SA += (int)wave_sum_i64(xa ? (long long)wl : 0ll);
                SB += (int)wave_sum_i64(xb ? (long long)wl : 0ll);
            } else {
                SA += __popcll(__ballot(xa));
                SB += __popcll(__ballot(xb));
            }
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

#ifdef MI_K2P_PROFILE
    unsigned long long tick_ = __builtin_amdgcn_s_memtime(), t_top = 0, t_gather = 0, t_sum = 0, t_rounds = 0, t_barrier = 0;
#endif
#ifndef MI_K2P_DBG_NOPRIO
    if constexpr (TW) __builtin_amdgcn_s_setprio(3);
#endif               // the sweeping wavefront is the critical path of its workgroup
    unsigned long long accepted = 0;
    uint32_t accA = 0, accB = 0;                                    // accepted moves of the running sweep, per replica
    uint32_t wa[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
    float TA = 1.0f, TB = 1.0f;
    const float cp = a.c_pair;
    uint32_t ring_buf = 0u;                                         // TW: which half of the ring holds the group being swept

    // one slot for both replicas; `wordA` / `wordB` = this slot's random words.
    // Order of a slot: (1) the LDS reads are ISSUED -- the lane's own cell and the 16 neighbour cells; (2) while they are in
    // flight the two thresholds are computed (neglog_u2: the fp32 steps as packed instructions, one for both replicas);
    // (3) one wait; (4) the field sums; (5) the accept masks.  The reads are inline asm (hipcc adds the zero base of the
    // dynamic LDS block to every address it computes itself), so the compiler does not count them: every register they
    // write, and the thresholds, pass THROUGH the wait statement ("+v"), which makes "used only after the wait" a
    // data dependence, not a scheduling accident -- and the wait is lgkmcnt(0), so whatever LDS or scalar-memory
    // operation the compiler may place before it is waited for as well (scripts/check_asm_lds.py checks the emitted
    // code for a read of such a register ahead of its wait at build time).
    auto slot_body = [&](auto c_in_group, int t, const SlotAdj &cur, uint32_t wordA, uint32_t wordB) {
        constexpr int C = decltype(c_in_group)::value;
        const int i = t * 64 + lane;
        uint32_t own;                                               // [x_A | x_B] of this lane's variable
        float gA = __uint_as_float(cur.lin), gB = gA;               // (lanes past n carry lin = +inf: never accepted)
        float thrA = 0.0f, thrB = 0.0f;
        f32x2_t thr2 = {0.0f, 0.0f};
        // what the accept masks need of the lane's OWN cell (TW: computed under the gathers, see below)
        uint32_t xiA = 0u, xiB = 0u, sgA = 0u, sgB = 0u;
        uint64_t XA = 0ull, XB = 0ull;
        f32x2_t cs = {0.0f, 0.0f}, csf = {0.0f, 0.0f};
        int ownA = 0, ownB = 0;
        auto own_terms = [&]() {
            xiA = (own >> 13) & 1u;                                 // 0x3c00 -> 1
            xiB = own >> 29;
            XA = __ballot((own & 0xffffu) != 0u);
            XB = __ballot((own >> 16) != 0u);
            sgA = xiA << 31;                                        // dE = x ? -f : f
            sgB = xiB << 31;
            cs = f32x2_t{__uint_as_float(__float_as_uint(cp) ^ sgA), __uint_as_float(__float_as_uint(cp) ^ sgB)};
            ownA = SA - (int)xiA;
            ownB = SB - (int)xiB;
            csf = cs * f32x2_t{(float)ownA, (float)ownB};           // the oracle's c * (float)(s - x): one packed multiply
        };
        K2P_TICK(t_top);
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            uint32_t word[16];
            // the packed neighbour word IS the LDS byte address of its cell (one wavefront per workgroup, no static LDS)
            if (g0 == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(own) : "v"(i * 4));
            // (TW: the ring read second, so that "at most 15 reads outstanding" below means the own cell and the thresholds are back)
            if (g0 == 0 && TW) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(thr2) : "v"(ring_lane + ring_buf), "n"(C * 512));
#pragma unroll
            for (int k = 0; k < 16; ++k)
#ifdef MI_K2P_DBG_LINEAR   /* timing only: conflict-free addresses */
                asm volatile("ds_read_b32 %0, %1" : "=v"(word[k]) : "v"((cur.col[g0 + k / 4][k & 3] & 0x3f00u) + lane * 4));
#else
                asm volatile("ds_read_b32 %0, %1" : "=v"(word[k]) : "v"(cur.col[g0 + k / 4][k & 3]));
#endif
            if (g0 == 0 && TW) {
                // 18 LDS reads are in flight and they return in order: with at most 15 outstanding the lane's own cell and
                // the thresholds are here -- everything that depends only on them is computed UNDER the gathers (the
                // second wait names its results, so it cannot sink below it)
                asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(own), "+v"(thr2) :: "memory");
#ifndef MI_K2P_DBG_TERMS_AFTER   /* (timing only: the terms after the full wait, as before) */
                own_terms();
#endif
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(word[0]), "+v"(word[1]), "+v"(word[2]), "+v"(word[3]), "+v"(word[4]), "+v"(word[5]),
                               "+v"(word[6]), "+v"(word[7]), "+v"(word[8]), "+v"(word[9]), "+v"(word[10]), "+v"(word[11]),
                               "+v"(word[12]), "+v"(word[13]), "+v"(word[14]), "+v"(word[15]), "+v"(csf)
                             :: "memory");
                thrA = thr2.x;
                thrB = thr2.y;
            } else if (g0 == 0) {
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
            K2P_TICK(t_gather);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float v = __uint_as_float(cur.val[g0 + k / 4][k & 3]);
                gA = __builtin_fmaf(v, half_lo(word[k]), gA);       // fma(val, x, g): the oracle's conditional add
                gB = __builtin_fmaf(v, half_hi(word[k]), gB);
            }
        }
#ifdef MI_K2P_DBG_TERMS_AFTER
        own_terms();
#else
        if constexpr (!TW) own_terms();
#endif
        if (WGT && t == wslot) {
            // ---- the slot of the weighted variables: a serial sweep per replica (few lanes, no sparse couplings) ----
            const uint64_t FA = weighted_slot_sweep(gA, thrA, wl, cp, xiA, SA, lane);
            const uint64_t FB = weighted_slot_sweep(gB, thrB, wl, cp, xiB, SB, lane);
            accA += (uint32_t)__popcll(FA);
            accB += (uint32_t)__popcll(FB);
            uint32_t fA, fB;
            asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(fA) : "v"(0x3c00u), "s"(FA));
            asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(fB) : "v"(0x3c000000u), "s"(FB));
            cell[i] = own ^ fA ^ fB;
            K2P_TICK(t_rounds);
            return;
        }
        const f32x2_t gs = {__uint_as_float(__float_as_uint(gA) ^ sgA), __uint_as_float(__float_as_uint(gB) ^ sgB)};
        // accept masks by fixed-point rounds (see k_anneal_csr_rank1): both replicas advance together; the oracle's
        // g + c * (float)(s - x) as one packed multiply and one packed add (no contraction)
        f32x2_t de = gs + csf;
        bool mA = de.x < thrA, mB = de.y < thrB;
        uint64_t AA = __ballot(mA), AB = __ballot(mB);
        K2P_TICK(t_sum);
        if ((AA | AB) != 0ull) {                                    // wave-uniform
            const int baseA = ownA - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(XA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)XA, 0u));
            const int baseB = ownB - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(XB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)XB, 0u));
            // (the loop ends by itself: a lane's decision depends on the masks of the lanes below it only, so after k rounds the
            // lowest k lanes are final whatever the data -- NaNs included, a compare with one is just false -- and round 65
            // repeats round 64.  Everything that leaves the loop is a wave-uniform mask: a per-lane flag live across it costs
            // six scalar instructions a round, a round counter five)
            uint64_t BA = AA ^ XA, BB = AB ^ XB;                    // the states after the moves guessed so far
#ifndef MI_K2P_DBG_NOROUNDS   /* (timing only: the first masks stand) */
#pragma nounroll
            for (;;) {
                const int sA = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(BA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)BA, (uint32_t)baseA));
                const int sB = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(BB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)BB, (uint32_t)baseB));
                de = gs + cs * f32x2_t{(float)sA, (float)sB};
                const uint64_t NBA = __ballot(de.x < thrA) ^ XA, NBB = __ballot(de.y < thrB) ^ XB;
                uint64_t d0 = NBA ^ BA, d1 = NBB ^ BB;
                asm("" : "+s"(d0), "+s"(d1));                       // (opaque: hipcc turns (a^b)|(c^d) == 0 into two compares and selects)
                BA = NBA;
                BB = NBB;
                if ((d0 | d1) == 0ull) break;
            }
#endif
            AA = BA ^ XA;
            AB = BB ^ XB;
            // sum(x) moves by popc(new states) - popc(old states) of the slot
            SA += __popcll(BA) - __popcll(XA);
            SB += __popcll(BB) - __popcll(XB);
            accA += (uint32_t)__popcll(AA);
            accB += (uint32_t)__popcll(AB);
            // toggling a state is one XOR of its half; every lane stores its cell (unchanged cells keep their word)
            uint32_t tA, tB;
            asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(tA) : "v"(0x3c00u), "s"(AA));
            asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(tB) : "v"(0x3c000000u), "s"(AB));
            cell[i] = own ^ tA ^ tB;
        }
        K2P_TICK(t_rounds);
    };

    using std::integral_constant;
    for (int s = 0; s < a.num_sweeps; ++s) {
        if constexpr (!TW) {
            if (a.temps_per_replica) {
                TA = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[rA])));
                TB = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[liveB ? rB : rA])));
            } else {
                TA = TB = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[s])));
            }
        }
        const uint32_t sw = (uint32_t)s + a.sweep_offset;
        // four slots per trip (one Philox block per replica), the adjacency one slot ahead in two register sets
        SlotAdj P = fetch_adj(0), Q;
#pragma unroll 1
        for (int t = 0; t < slots; t += 4) {
            if constexpr (TW) {
                K2P_TICK(t_top);
                __builtin_amdgcn_s_barrier();                       // this group's thresholds are in the ring
                K2P_TICK(t_barrier);
            } else {
                philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gidA, 0u, a.seed_lo, a.seed_hi, wa);
                philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gidB, 0u, a.seed_lo, a.seed_hi, wb);
            }
            Q = fetch_adj(t + 1);
            slot_body(integral_constant<int, 0>{}, t, P, wa[0], wb[0]);
            if (t + 1 < slots) {                                    // wave-uniform
                P = fetch_adj(t + 2);
                slot_body(integral_constant<int, 1>{}, t + 1, Q, wa[1], wb[1]);
                if (t + 2 < slots) {
                    Q = fetch_adj(t + 3);
                    slot_body(integral_constant<int, 2>{}, t + 2, P, wa[2], wb[2]);
                    P = fetch_adj(t + 4);
                    if (t + 3 < slots) slot_body(integral_constant<int, 3>{}, t + 3, Q, wa[3], wb[3]);
                }
            }
            if constexpr (TW) ring_buf ^= 2048u;
        }
        accepted += (unsigned long long)accA + (liveB ? (unsigned long long)accB : 0ull);
        accA = accB = 0;
    }

    // ---- epilogue: states out, exact fp64 energies (same sums as k_anneal_csr_rank1) ----
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1 && !liveB) break;
        const int r = rep ? rB : rA;
        const int sh = rep ? 16 : 0;
        uint8_t *dst = static_cast<uint8_t *>(a.states) + (size_t)r * n;
        long long cnt = 0, cnt2 = 0;                               // sum_j w_j x_j, sum_j w_j^2 x_j
        double e = 0.0;
        for (int t = 0; t < slots; ++t) {
            const int i = t * 64 + lane;
            const bool on = ((cell[i] >> sh) & 0xffffu) != 0u;
            if (i < n) dst[i] = (uint8_t)on;
            if (WGT && t == wslot) {
                cnt += wave_sum_i64(on ? (long long)wl : 0ll);
                cnt2 += wave_sum_i64(on ? (long long)wl * wl : 0ll);
            } else {
                const int c1 = __popcll(__ballot(on));
                cnt += c1;
                cnt2 += c1;
            }
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
            a.energy[r] = e + cp64 * 0.5 * ((double)cnt * (double)cnt - (double)cnt2) + a.offset;
        }
    }
    if (lane == 0) atomicAdd(&a.stats[1], accepted);
#ifdef MI_K2P_PROFILE
    if (lane == 0) {
        atomicAdd(&a.stats[8], t_top); atomicAdd(&a.stats[9], t_gather); atomicAdd(&a.stats[10], t_sum);
        atomicAdd(&a.stats[11], t_rounds); atomicAdd(&a.stats[12], t_barrier);
    }
#endif
}

template <typename KernelT>
int launch_pair(KernelT kernel, const EllArgs &a, bool tw, hipStream_t st)
{
    // 4 bytes per variable; TW: the two-deep ring of thresholds behind them (2 x 4 slots x 64 lanes x 8 bytes)
    const size_t lds = (size_t)a.slots * 256 + (tw ? 4096 : 0);
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "csr_rank1 pair kernel: n = %d exceeds the state LDS budget", a.n);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel(tw ? "k_anneal_csr_rank1_pair<%d, tw>" : "k_anneal_csr_rank1_pair<%d>", a.D);
    hipLaunchKernelGGL(kernel, dim3((a.R + 1) / 2), dim3(tw ? 128 : 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

// a.adj4 must hold the pair packing (neighbour word = 4 * index)
int mi_launch_csr_rank1_pair(const EllArgs &a, bool tw, hipStream_t st)
{
    if (!a.adj4) return fail(MI_EHIP, "csr_rank1 pair kernel: packed adjacency missing");
    // the ring costs LDS: beyond 64 slots only seven workgroups (14 replicas) fit a CU, and a run that fills the chip
    // (16 replicas per CU) would take two rounds -- such models keep the kernel without a threshold wavefront
    if (tw && ((size_t)a.slots * 256 + 4096) * 8 > 160 * 1024 && a.R > 2 * 7 * 256) tw = false;
    if (a.wslot >= 0) {                       // pair-term weights (16 entries per variable)
        if (a.D != 16) return fail(MI_EUNSUPPORTED, "csr_rank1 pair kernel: pair-term weights at slot-ELL width %d not built", a.D);
        return tw ? launch_pair(k_anneal_csr_rank1_pair<16, true, true>, a, true, st)
                  : launch_pair(k_anneal_csr_rank1_pair<16, false, true>, a, false, st);
    }
    if (a.D == 16 && tw) return launch_pair(k_anneal_csr_rank1_pair<16, true>, a, true, st);
    if (a.D == 16) return launch_pair(k_anneal_csr_rank1_pair<16, false>, a, false, st);
    if (a.D == 32) return launch_pair(k_anneal_csr_rank1_pair<32, false>, a, false, st);
    return fail(MI_EUNSUPPORTED, "csr_rank1 pair kernel: slot-ELL width %d not built", a.D);
}

}  // namespace mi_sa_impl
