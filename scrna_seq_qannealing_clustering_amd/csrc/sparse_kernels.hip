// sparse_kernels.hip -- anneal kernels for the STRUCTURED models of the reference (gfx950 only):
//
//   K2  k_anneal_csr_rank1<D>  binary model  E(x) = sum lin_i x_i + sum_{i<j} (c + S_ij) x_i x_j + offset
//       -- the graph-partition QUBO of BQM_clustering.py:38-47 (sparse cut term S + 2*gamma on EVERY
//       pair) without ever materialising the n x n matrix: an accepted flip touches deg(i) cached
//       fields and one integer (s = sum x), not n of them.
//   K3  k_anneal_potts<D>      k-way model   E(l) = lin_offset + sum_{u<v, l_u == l_v} (c + S_uv)
//       -- the DQM of DQM_clustering.py:29-43 in its native (one-hot-free) form.
//
// Layout ("slot-ELL"): variable i = 64 t + lane; the neighbours of the 64 variables of slot t are stored
// as ell[(t*D + k)*64 + lane], k < D (D = 16, 32 or 64 = padded max degree), so a wave loads a whole slot's
// adjacency with D coalesced 256-byte reads into registers.  Padding entries point at the variable
// itself with weight +0.0f (adding +0.0f is the identity, so padded and unpadded sums are bit-equal).
// One wavefront owns one replica; per-replica state that needs random access (cached fields g for K2,
// labels for K3) lives in LDS.  Chain specification: DESIGN.md / oracle/sa_oracle.c (2b), (2c).
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

constexpr int kSparseWaves = 4;      // wavefronts (replicas) per workgroup

// Philox words of the four slots 4*tg .. 4*tg+3 for this lane
__device__ __forceinline__ void slot_words(uint32_t (&w)[4], int tg, int lane, uint32_t s, uint32_t g,
                                           uint32_t tag, uint32_t k0, uint32_t k1)
{
    philox4x32_10((uint32_t)(tg * 64 + lane), s, g, tag, k0, k1, w);
}

// ------------------------------------------------------------------------------------------------
// K2
// ------------------------------------------------------------------------------------------------
// Chain (oracle/sa_oracle.c 2b): variables in index order, 64 at a time (one slot = one wavefront).  When a
// slot is entered every lane evaluates the sparse part of ITS variable's field fresh from the state,
//     g_i = lin_i ; for k in stored order: if x[col_k]: g_i += val_k            (this is dE = Q_i . x in CSR form)
// with the adjacency of the slot arriving as D coalesced 256-byte reads (prefetched one slot ahead) and the
// state bits read from LDS (one 64-bit mask per slot per replica -- the only per-replica memory: 8 bytes per
// 64 variables, so 16 replicas per CU stay resident at any n).  Nothing is cached between slots: no field
// array, no neighbour scatter after a flip, no atomics, nothing to re-synchronise.
// Inside the slot the serial accept -> commit -> re-test loop only touches what the next decision can depend
// on:  the integer s = sum x  ((float)(s - x_i) moves by +-1.0, exact)  and the g of the flipped variable's
// neighbours INSIDE this slot: the host stores those first in a row-major copy of the adjacency (`rows`,
// `meta` = their count), every lane preloads the first four (col, val) of its row, and the update is two
// v_readlane + one masked add per in-slot neighbour (0.36 per flip on the PBMC-sized SNN graph).  The flipped
// lane is never tested again in this slot, so its own state is patched after the loop (x ^= flipped), and
// the sign of a flip comes from the slot's state mask.  ~20 instructions per accepted flip.
#ifdef MI_K2_PROFILE
#define K2_TICK(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tick_; tick_ = now_; } while (0)
#else
#define K2_TICK(var) do { } while (0)
#endif

// XS: how the state lives in LDS.  0: one BIT per variable (any n).  1: one BYTE per variable (16 replicas x n
// bytes fit a CU): a neighbour's state is a ds_read_u8 at its index and enters the field sum as fma(val, x, g) --
// two VALU instructions per neighbour (convert, fma) instead of four (address shift, bit extract, mask, add).
// 2: one HALF per variable (0.0 / 1.0; n <= 4608): the fma takes the half as it is (v_fma_mix_f32 widens the
// operand exactly), ONE VALU instruction per neighbour, the same fp32 result bit for bit.
// TW (round 3): a second wavefront of the workgroup computes the random words and thresholds one group of four slots
// ahead and hands them over through a two-deep ring in LDS behind the state (k_anneal_csr_rank1_pair has the same
// arrangement) -- for runs of up to 1024 replicas, where every wavefront has a SIMD to itself and a third of a slot's
// instructions move to an idle one.  a.ring_off = byte offset of the ring (2 x 4 slots x 64 lanes x 4 bytes).
template <int D, int XS, bool TW = false>
__global__ void __launch_bounds__(TW ? 128 : 64, (TW ? 2 : (D <= 32 ? 4 : 2))) k_anneal_csr_rank1(EllArgs a)
{
    constexpr bool XB = XS != 0;                     // state addressed per variable (byte or half), not per bit
    typedef _Float16 half_t;
    // ONE wavefront per workgroup: the replica's state masks start at LDS address 0, so the host can store
    // every neighbour as a ready-made (LDS byte offset of its 32-bit state word << 8 | bit position).
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x;
    if (r >= a.R) return;
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots;
    uint64_t *xm = reinterpret_cast<uint64_t *>(lds);                          // bit l of xm[t] = x[64 t + l]
    const uint8_t *init = static_cast<const uint8_t *>(a.init);
    const uint2 *rows = a.rows;
    if constexpr (TW) {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1) {
            // ---- the threshold wavefront ----
            uint32_t tw_w[4];
            uint32_t buf = 0;
            const uint32_t at0 = (uint32_t)a.ring_off + (uint32_t)lane * 4u;
            for (int s2 = 0; s2 < a.num_sweeps; ++s2) {
                const float T2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s2])));
#pragma unroll 1
                for (int t = 0; t < slots; t += 4) {
                    slot_words(tw_w, t >> 2, lane, (uint32_t)s2 + a.sweep_offset, gid, 0u, a.seed_lo, a.seed_hi);
                    const f32x2_t l01 = neglog_u2(tw_w[0], tw_w[1]) * f32x2_t{T2, T2}, l23 = neglog_u2(tw_w[2], tw_w[3]) * f32x2_t{T2, T2};
                    const uint32_t at = at0 + buf;
                    asm volatile("ds_write_b32 %0, %1" :: "v"(at), "v"(l01.x) : "memory");
                    asm volatile("ds_write_b32 %0, %1 offset:256" :: "v"(at), "v"(l01.y) : "memory");
                    asm volatile("ds_write_b32 %0, %1 offset:512" :: "v"(at), "v"(l23.x) : "memory");
                    asm volatile("ds_write_b32 %0, %1 offset:768" :: "v"(at), "v"(l23.y) : "memory");
                    buf ^= 1024u;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            return;
        }
        __builtin_amdgcn_s_setprio(3);
    }
    uint32_t ring_buf = 1024u;                       // TW: the half of the ring the running group is in (toggled at its start)

    // (pair-term weights, mi_sa_problem_set_pair_weights: S is sum_j w_j x_j; only the lanes of ONE slot carry weights
    // other than 1, and that slot is swept by a serial loop -- weighted_slot_sweep)
    const int wl = a.wslot >= 0 ? a.wgt[lane] : 0;
    auto slot_sum = [&](int t, uint64_t m) -> int {
        return t == a.wslot ? (int)wave_sum_i64(((m >> lane) & 1ull) ? (long long)wl : 0ll) : __popcll(m);
    };
    int S = 0;
    if (init) {
        for (int t = 0; t < slots; ++t) {
            const int i = t * 64 + lane;
            const uint64_t m = __ballot(i < n && init[(size_t)r * n + i] && a.lin[i] < INFINITY);    // (a hole stays 0)
            if constexpr (XS == 2) reinterpret_cast<half_t *>(lds)[i] = (half_t)(float)((m >> lane) & 1ull);
            else if constexpr (XS == 1) lds[i] = (char)((m >> lane) & 1ull);
            else if (lane == 0) xm[t] = m;
            S += slot_sum(t, m);
        }
    } else {
        for (int tg = 0; tg * 4 < slots; ++tg) {
            uint32_t w[4];
            slot_words(w, tg, lane, 0u, gid, 1u, a.seed_lo, a.seed_hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * tg + c;
                if (t >= slots) break;
                // (a variable whose linear term is +inf is a HOLE of a padded layout, mi_sa_plan_slot_layout: it
                // starts at 0 and its dE = +inf is never accepted -- as the lanes past n)
                const uint64_t m = __ballot(t * 64 + lane < n && (w[c] >> 31) && a.lin[t * 64 + lane] < INFINITY);
                if constexpr (XS == 2) reinterpret_cast<half_t *>(lds)[t * 64 + lane] = (half_t)(float)((m >> lane) & 1ull);
                else if constexpr (XS == 1) lds[t * 64 + lane] = (char)((m >> lane) & 1ull);
                else if (lane == 0) xm[t] = m;
                S += slot_sum(t, m);
            }
        }
    }

    // The slot's adjacency does not depend on the chain: it is fetched one slot ahead into one of two register
    // buffers (the slot loop is unrolled by two, so no copies) with D/2 buffer_load_dwordx4 whose addresses
    // cost no VALU work: descriptor + (lane*16) + scalar slot offset + immediate.
    //   adj4 layout per slot: for g < D/4: [64 lanes][4] packed neighbours, then [64 lanes][4] values.
    // D = 16 / 32 / 64: register resident.  D = 0: rows of any width W = a.D (a multiple of 16): the adjacency is
    // read group by group inside the field sum (no prefetch) -- the same chain for the untrimmed graphs' wide rows.
    constexpr int G = D ? D / 4 : 1;
    const int W = D ? D : a.D, Gw = W / 4;
    const __amdgpu_buffer_rsrc_t rs_adj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.adj4), 0, slots * Gw * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.lin), 0, slots * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_flag = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(a.slot_flags), 0, slots * 4, 0x00020000);
    struct SlotAdj { u32x4 col[G]; u32x4 val[G]; uint32_t lin, flag; int soff; };
    auto fetch_adj = [&](int t) {
        SlotAdj p;
        const int tt = t < slots ? t : slots - 1;
        const int soff = tt * (Gw * 2048);
        p.soff = soff;
        if constexpr (D != 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                p.col[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16, soff + g * 2048, 0);
                p.val[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16, soff + g * 2048 + 1024, 0);
            }
        }
        p.lin = __builtin_amdgcn_raw_buffer_load_b32(rs_lin, lane * 4, tt * 256, 0);
        p.flag = __builtin_amdgcn_raw_buffer_load_b32(rs_flag, 0, tt * 4, 0);   // arrives with the prefetch: no stall
        return p;
    };

    unsigned long long accepted = 0;
#ifdef MI_K2_PROFILE
    unsigned long long tick_ = __builtin_amdgcn_s_memtime(), t_pre = 0, t_loop = 0, t_wait = 0, t_apply = 0, t_init = 0;
#endif
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    float T = 1.0f;
    int s = 0;

    auto slot_body = [&](int t, const SlotAdj &cur) {
        if ((t & 3) == 0) {
            if constexpr (TW) {
                __builtin_amdgcn_s_barrier();                 // this group's thresholds are in the ring
                ring_buf ^= 1024u;
            } else {
                slot_words(w, t >> 2, lane, (uint32_t)s + a.sweep_offset, gid, 0u, a.seed_lo, a.seed_hi);
            }
        }
        const int c = t & 3;
        const uint32_t wc = c == 0 ? w[0] : (c == 1 ? w[1] : (c == 2 ? w[2] : w[3]));
        const int i = t * 64 + lane;
        float thr_tw = 0.0f;
        if constexpr (TW) thr_tw = *reinterpret_cast<const float *>(lds + a.ring_off + ring_buf + (uint32_t)c * 256u + (uint32_t)lane * 4u);
        const bool general = __builtin_amdgcn_readfirstlane((int)cur.flag) != 0;   // some variable of this slot has a
                                                              // neighbour inside the slot
        K2_TICK(t_init);
        // this lane's own state word goes first (conflict-free: consecutive lanes, consecutive cells); it is back by
        // the time the neighbour gathers are
        uint32_t own = 0u;
        if constexpr (XS == 2) own = reinterpret_cast<const uint16_t *>(lds)[i];      // 0x0000 / 0x3c00 (half 0.0 / 1.0)
        else if constexpr (XS == 1) own = (uint32_t)(unsigned char)lds[i];            // 0 / 1
        // fresh field: lin_i + the stored neighbours whose bit is set, in stored order.  A clear bit adds +0.0f
        // (value AND mask): x + 0.0f == x for every x except -0.0f, and +-0.0f are the same number to every
        // comparison downstream -- no decision can differ from the oracle's skip.
        float gi = __uint_as_float(cur.lin);
        auto add16 = [&](const u32x4 *c4, const u32x4 *v4) {  // 16 state words in flight, then 16 adds
            uint32_t word[16];
            // The gathers are written out: one wavefront per workgroup and no static LDS put the state at LDS
            // address 0, so the packed neighbour word IS the address -- hipcc adds the (zero) base of the dynamic
            // LDS block to every one of them with a v_add_u32.  The compiler does not count asm LDS reads: the wait
            // that names all 16 destinations follows them (cdna_hip_programming.md 5.7, form ii).
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t pk = c4[k / 4][k & 3];
                if constexpr (XS == 2) asm volatile("ds_read_u16 %0, %1" : "=v"(word[k]) : "v"(pk));       // 2 * index
                else if constexpr (XS == 1) asm volatile("ds_read_u8 %0, %1" : "=v"(word[k]) : "v"(pk));   // the index
                else asm volatile("ds_read_b32 %0, %1" : "=v"(word[k]) : "v"(pk >> 8));
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(word[0]), "+v"(word[1]), "+v"(word[2]), "+v"(word[3]), "+v"(word[4]), "+v"(word[5]),
                           "+v"(word[6]), "+v"(word[7]), "+v"(word[8]), "+v"(word[9]), "+v"(word[10]), "+v"(word[11]),
                           "+v"(word[12]), "+v"(word[13]), "+v"(word[14]), "+v"(word[15])
                         :: "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t pk = c4[k / 4][k & 3];
                if constexpr (XS == 2) {
                    // the half widens exactly: fma(val, x, g), one rounding (v_fma_mix_f32)
                    const half_t hx = __builtin_bit_cast(half_t, (uint16_t)word[k]);
                    gi = __builtin_fmaf(__uint_as_float(v4[k / 4][k & 3]), (float)hx, gi);
                } else if constexpr (XS == 1) {
                    // fma(val, 1, g) = g + val and fma(val, 0, g) = g: the oracle's conditional add, one rounding
                    gi = __fmaf_rn(__uint_as_float(v4[k / 4][k & 3]), (float)word[k], gi);
                } else {
                    const int msk = __builtin_amdgcn_sbfe((int)word[k], pk, 1u);   // v_bfe_i32: bit pk[4:0] -> 0 / -1
                    gi = gi + __uint_as_float(v4[k / 4][k & 3] & (uint32_t)msk);
                }
            }
        };
        if constexpr (D != 0) {
#pragma unroll
            for (int g0 = 0; g0 < G; g0 += 4) add16(&cur.col[g0], &cur.val[g0]);
        } else {
#pragma unroll 1
            for (int g0 = 0; g0 < Gw; g0 += 4) {
                u32x4 c4[4], v4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c4[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16, cur.soff + (g0 + q) * 2048, 0);
                    v4[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16, cur.soff + (g0 + q) * 2048 + 1024, 0);
                }
                add16(c4, v4);
            }
        }
        K2_TICK(t_apply);
        // (the lanes past n carry lin = +inf: dE = +inf is never below any threshold, no per-lane bound check here)
        const float thr = TW ? thr_tw : neglog_u(wc) * T;
        uint64_t xm_t;                                        // the slot's 64 state bits as a scalar
        uint32_t xi;
        if constexpr (XS == 2) {
            xm_t = __ballot(own != 0u);
            xi = own >> 13;                                   // 0x3c00 -> 1
        } else if constexpr (XS == 1) {
            xm_t = __ballot(own != 0u);
            xi = own;
        } else {
            const uint64_t xm_v = xm[t];                      // same word in every lane: make it scalar
            xm_t = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xm_v >> 32)) << 32) |
                   (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xm_v);
            xi = (uint32_t)((xm_t >> lane) & 1ull);
        }
        const uint32_t sgnbit = xi << 31;                     // dE = x ? -f : f
        uint64_t todo = ~0ull, flipped = 0ull;
        bool mine = false;                                    // this lane's variable flips in this slot
        if (t == a.wslot) {
            // ---- the slot of the weighted variables: a serial sweep (few lanes, no sparse couplings) ----
            flipped = weighted_slot_sweep(gi, thr, wl, a.c_pair, xi, S, lane);
            mine = (flipped >> lane) & 1ull;
        } else if (!general) {
            // ---- no variable of this slot has a neighbour inside it: decisions depend on s alone -------------
            // Sequentially, lane i sees s = S + d_i with d_i = sum over the ACCEPTING lanes j < i of (+1 if x_j = 0,
            // -1 if x_j = 1), and accepts iff  +-(g_i + c (float)(S + d_i - x_i)) < thr_i  (the oracle's expression,
            // one multiply and one add).  That recurrence has exactly one solution -- the accept mask A of the
            // sequential sweep -- and it is found WITHOUT the serial loop: evaluate every lane at once under a
            // guessed mask, rebuild the mask, repeat until it reproduces itself.  After k rounds the lowest k lanes
            // are final (induction over the lane index), so the fixed point is reached in at most 65 rounds whatever
            // the data; on the SNN models it takes 2-3 (one round finds the mask at d = 0, the next ones only move
            // the few lanes whose threshold lies within |d_i| of the current sum), against up to 64 trips through a
            // dependent vector -> scalar -> vector chain.  Every round evaluates the exact fp32 predicate, so the
            // decisions are the oracle's bit for bit.
            //   d_i = popc_below_i(A ^ X) - popc_below_i(X)     (X = the slot's state mask)
            // because a lane with x_j = 0 contributes A_j and one with x_j = 1 contributes (1 - A_j) - 1.
            const float gs = __uint_as_float(__float_as_uint(gi) ^ sgnbit);
            const float cs = __uint_as_float(__float_as_uint(a.c_pair) ^ sgnbit);
            const int s_own = S - (int)xi;
            K2_TICK(t_pre);
            mine = gs + cs * (float)s_own < thr;
            uint64_t A = __ballot(mine);
            if (A != 0ull) {                                  // wave-uniform
                const int base = s_own - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(xm_t >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)xm_t, 0u));
                for (int round = 0; round < 66; ++round) {
                    const uint64_t Bm = A ^ xm_t;
                    const int s_i = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(Bm >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)Bm, (uint32_t)base));
                    mine = gs + cs * (float)s_i < thr;
                    const uint64_t A2 = __ballot(mine);
                    if (A2 == A) break;
                    A = A2;
                }
            }
            flipped = A;
            (void)todo;
            S += __popcll(flipped & ~xm_t) - __popcll(flipped & xm_t);
        } else {
            // (slots with internal edges: none under the slot-independent order; their row heads are fetched here)
            const uint32_t metav = a.meta[i];
            const uint4 e01 = *reinterpret_cast<const uint4 *>(rows + (size_t)i * W);
            const uint4 e23 = *reinterpret_cast<const uint4 *>(rows + (size_t)i * W + 2);
            const uint64_t has_in = __ballot((metav & 0xffu) != 0u);
            float Sf = (float)(S - (int)xi);
            K2_TICK(t_pre);
            while (true) {
                const float fi = gi + a.c_pair * Sf;
                const float dE = __uint_as_float(__float_as_uint(fi) ^ sgnbit);
                const uint64_t m = __ballot(dE < thr) & todo;
                if (m == 0) break;
                const int l = __ffsll((unsigned long long)m) - 1;
                todo = (~0ull << l) << 1;
                flipped |= 1ull << l;
                const bool xl = (xm_t >> l) & 1ull;           // the lane's bit BEFORE its (only) flip in this slot
                const float sgn = xl ? -1.0f : 1.0f;
                S += xl ? -1 : 1;
                Sf += sgn;
                if ((has_in >> l) & 1ull) {                   // wave-uniform: l has neighbours inside this slot
                    const int nin = (int)(__builtin_amdgcn_readlane((int)metav, l) & 0xff);
                    auto hit = [&](uint32_t cc, uint32_t vv) {
                        if (lane == (int)(cc & 63u)) gi = gi + sgn * __uint_as_float(vv);
                    };
                    hit(__builtin_amdgcn_readlane((int)e01.x, l), __builtin_amdgcn_readlane((int)e01.y, l));
                    if (nin > 1) hit(__builtin_amdgcn_readlane((int)e01.z, l), __builtin_amdgcn_readlane((int)e01.w, l));
                    if (nin > 2) hit(__builtin_amdgcn_readlane((int)e23.x, l), __builtin_amdgcn_readlane((int)e23.y, l));
                    if (nin > 3) hit(__builtin_amdgcn_readlane((int)e23.z, l), __builtin_amdgcn_readlane((int)e23.w, l));
                    for (int k = 4; k < nin; ++k) {
                        const uint2 e = rows[((size_t)t * 64 + l) * W + k];
                        hit(e.x, e.y);
                    }
                }
            }
        }
        if (flipped) {                                        // wave-uniform
            accepted += (unsigned long long)__popcll(flipped);
            if (general && t != a.wslot) mine = (flipped >> lane) & 1ull;
            // toggling a state cell is one XOR of the word read at the top of the slot
            if constexpr (XS == 2) { if (mine) reinterpret_cast<uint16_t *>(lds)[i] = (uint16_t)(own ^ 0x3c00u); }
            else if constexpr (XS == 1) { if (mine) lds[i] = (char)(own ^ 1u); }
            else if (lane == 0) xm[t] = xm_t ^ flipped;
        }
        K2_TICK(t_loop);
    };

    for (s = 0; s < a.num_sweeps; ++s) {
        T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s])));
        if constexpr (D == 16) {
            SlotAdj A = fetch_adj(0), B;
#pragma unroll 1
            for (int t = 0; t < slots; t += 2) {
                B = fetch_adj(t + 1);
                slot_body(t, A);
                if (t + 1 < slots) {                          // wave-uniform
                    A = fetch_adj(t + 2);
                    slot_body(t + 1, B);
                }
            }
        } else {                                              // wide rows: one buffer, hidden by the other waves
#pragma unroll 1
            for (int t = 0; t < slots; ++t) {
                const SlotAdj A = fetch_adj(t);
                slot_body(t, A);
            }
        }
    }
#ifdef MI_K2_PROFILE
    if (lane == 0) {
        atomicAdd(&a.stats[8], t_pre); atomicAdd(&a.stats[9], t_loop); atomicAdd(&a.stats[10], t_wait);
        atomicAdd(&a.stats[11], t_apply); atomicAdd(&a.stats[12], t_init);
    }
#endif

    // ---- epilogue: states out, exact fp64 energy ----
    uint8_t *dst = static_cast<uint8_t *>(a.states) + (size_t)r * n;
    long long cnt = 0, cnt2 = 0;                             // sum_j w_j x_j and sum_j w_j^2 x_j (both the count when w = 1)
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        const uint64_t m = XS == 2 ? __ballot(reinterpret_cast<const uint16_t *>(lds)[i] != 0)
                                   : (XS == 1 ? __ballot(lds[i] != 0) : xm[t]);
        const int on = (int)((m >> lane) & 1ull);
        if (i < n) dst[i] = (uint8_t)on;
        if (t == a.wslot) {
            cnt += wave_sum_i64(on ? (long long)wl : 0ll);
            cnt2 += wave_sum_i64(on ? (long long)wl * wl : 0ll);
        } else {
            cnt += __popcll(m);
            cnt2 += __popcll(m);
        }
        if (!on) continue;
        double acc = 0.0;
        for (int k = 0; k < W; ++k) {
            const uint32_t cc = a.ell_col[((size_t)t * W + k) * 64 + lane];
            const size_t at = ((size_t)t * W + k) * 64 + lane;
            const double vv = a.ell_val64 ? a.ell_val64[at] : (double)a.ell_val[at];
            const bool xc = XS == 2 ? (reinterpret_cast<const uint16_t *>(lds)[cc] != 0)
                                    : (XS == 1 ? (lds[cc] != 0) : (bool)((xm[cc >> 6] >> (cc & 63u)) & 1ull));
            if (xc) acc += vv;
        }
        e += (a.lin64 ? a.lin64[i] : (double)a.lin[i]) + 0.5 * acc;
    }
    e = wave_sum_f64(e);
    if (lane == 0) {
        const double cp = a.ell_val64 ? a.c_pair64 : (double)a.c_pair;
        a.energy[r] = e + cp * 0.5 * ((double)cnt * (double)cnt - (double)cnt2) + a.offset;     // (w = 1: cnt (cnt - 1) / 2 pairs)
        atomicAdd(&a.stats[1], accepted);
    }
}

// ------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------
// LDS per wave: labels, one byte per variable (K <= 64).  Lane q of the wave keeps cnt[q] (cluster sizes).
// Proposal of variable i: a = l_i, b = (a + 1 + word(i,s,g,2) mod (K-1)) mod K;
//   dE = c (cnt_b - cnt_a + 1) + hd,  hd = sum over the neighbours j of S_ij ([l_j = b] - [l_j = a])
// (ONE signed fp32 sum in stored neighbour order -- oracle 2c; round 3: it was the difference of two sums before).  Same loop economy as K2: the slot's
// adjacency is prefetched one slot ahead; inside the slot a commit moves two cluster sizes (every lane
// patches the sizes of ITS two labels with +-1.0, exact) and re-evaluates h only on the lanes that have the
// moved variable among their neighbours -- found from the mover's in-slot neighbour list (`rows` / `meta`,
// as in K2), not by scanning every lane's row.  A mover is never tested again in its slot, so its label is
// written after the loop -- unless it has in-slot neighbours, which read it back at once.
// D = 16 / 32 / 64: the slot's adjacency is register resident (prefetched for D = 16).  D = 0: rows of ANY width
// W = a.D (a multiple of 16; the untrimmed SNN graphs of the reference reach degrees of order k^2): the adjacency
// is read entry by entry from L2 inside the field sum -- the same chain, slower.
template <int D>
__global__ void __launch_bounds__(256, (D <= 32 ? 4 : 2)) k_anneal_potts(EllArgs a)
{
    const int W = D ? D : a.D;                               // adjacency entries per variable
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (r >= a.R) return;
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots, K = a.K;
    // LDS per wave: the labels (one byte per variable) and the K cluster sizes (one int each, 64 reserved)
    uint8_t *lab = reinterpret_cast<uint8_t *>(lds) + (size_t)wave * ((size_t)slots * 64 + 256);
    int *cnt = reinterpret_cast<int *>(lab + (size_t)slots * 64);
    const uint16_t *init = static_cast<const uint16_t *>(a.init);
    const uint2 *rows = a.rows;

    for (int tg = 0; tg * 4 < slots; ++tg) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (!init) slot_words(w, tg, lane, 0u, gid, 1u, a.seed_lo, a.seed_hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = 4 * tg + c;
            if (t >= slots) break;
            const int i = t * 64 + lane;
            uint32_t v = 0;
            // (bit 31 of meta: no variable sits at this position -- past n, or a hole of a padded layout,
            // mi_sa_plan_slot_layout / mi_sa_problem_set_absent: label 0, in no cluster, never proposed)
            const bool present = (a.meta[i] >> 31) == 0u;
            if (present) v = init ? (uint32_t)init[(size_t)r * n + i] : (w[c] % (uint32_t)K);
            lab[i] = (uint8_t)v;
        }
    }
    int cntv = 0;                                            // lane q: number of variables with label q
    for (int q = 0; q < K; ++q) {
        int c = 0;
        for (int t = 0; t < slots; ++t)
            c += __popcll(__ballot((a.meta[t * 64 + lane] >> 31) == 0u && lab[t * 64 + lane] == q));
        if (lane == q) cntv = c;
    }
    cnt[lane] = cntv;

    constexpr bool PF = (D == 16);
    constexpr int DR = D ? D : 1;                            // register-resident entries (none for D = 0)
    struct SlotAdj { uint32_t col[DR]; float val[DR]; uint32_t meta; int slot; };
    auto fetch_adj = [&](int t) {
        SlotAdj p;
        const int tt = t < slots ? t : slots - 1;
        p.slot = tt;
        p.meta = a.meta[tt * 64 + lane];                     // in-slot neighbour count of this lane's variable
        if constexpr (D != 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                p.col[k] = a.ell_col[((size_t)tt * D + k) * 64 + lane];
                p.val[k] = a.ell_val[((size_t)tt * D + k) * 64 + lane];
            }
        }
        return p;
    };

    unsigned long long accepted = 0;
#ifdef MI_K2_PROFILE
    unsigned long long tick_ = __builtin_amdgcn_s_memtime(), t_pre = 0, t_loop = 0, t_apply = 0, t_init = 0;
#endif
    const bool use_min = a.min_size > 0;                     // wave-uniform
    // w mod (K-1) for a 32-bit w without the 40-instruction integer division: q = mulhi(w, floor(2^32/d)) is
    // floor(w/d) or one less, so the remainder needs at most one correction (d = K-1 <= 63)
    const uint32_t dK = (uint32_t)(K > 1 ? K - 1 : 1);
    const uint32_t magic = dK == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / dK);   // (2^32 does not fit for d = 1)
    for (int s = 0; s < a.num_sweeps && K > 1; ++s) {
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(a.temps[a.temps_per_replica ? r : s])));
        SlotAdj nxt;
        if constexpr (PF) nxt = fetch_adj(0);
        uint32_t w0[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int t = 0; t < slots; ++t) {                    // ONE copy of the slot body
            if ((t & 3) == 0) {
                slot_words(w0, t >> 2, lane, (uint32_t)s + a.sweep_offset, gid, 0u, a.seed_lo, a.seed_hi);
                slot_words(w2, t >> 2, lane, (uint32_t)s + a.sweep_offset, gid, 2u, a.seed_lo, a.seed_hi);
            }
            const int c = t & 3;
            const uint32_t w0c = c == 0 ? w0[0] : (c == 1 ? w0[1] : (c == 2 ? w0[2] : w0[3]));
            const uint32_t w2c = c == 0 ? w2[0] : (c == 1 ? w2[1] : (c == 2 ? w2[2] : w2[3]));
            const int i = t * 64 + lane;
            SlotAdj cur;
            if constexpr (PF) cur = nxt; else cur = fetch_adj(t);
            const uint32_t metav = cur.meta;                 // (prefetched with the adjacency)
            const uint64_t has_in = __ballot((metav & 0xffu) != 0u);
            // the movers' in-slot neighbours (lane ids) are needed only in slots that have any -- none under the
            // slot-independent order; issued before the next slot's prefetch (loads return in order)
            uint4 e01 = make_uint4(0u, 0u, 0u, 0u), e23 = e01;
            if (has_in != 0ull) {
                e01 = *reinterpret_cast<const uint4 *>(rows + (size_t)i * W);
                e23 = *reinterpret_cast<const uint4 *>(rows + (size_t)i * W + 2);
                asm volatile("" ::: "memory");
            }
            if constexpr (PF) nxt = fetch_adj(t + 1);
            K2_TICK(t_init);
            float thr = neglog_u(w0c) * T;
            if (metav >> 31) thr = -INFINITY;                // nobody sits here
            const int la = lab[i];
            uint32_t rem = w2c - __umulhi(w2c, magic) * dK;  // in [0, 2 dK)
            rem = rem >= dK ? rem - dK : rem;
            int lb = la + 1 + (int)rem;                      // in [1, 2K - 2]
            lb = lb >= K ? lb - K : lb;
            float ha = 0.0f, hb = 0.0f;
            auto sum_h = [&]() {
                ha = 0.0f;
                hb = 0.0f;
                if constexpr (D != 0) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        // ONE signed sum (chain specification 2c, round 3): + S_ij for a neighbour with the target label,
                        // - S_ij for one with the lane's own label; it is kept in hb with ha = 0 (hb - ha below is exact)
                        const int lj = lab[cur.col[k]];
                        const float v = cur.val[k];
                        hb = hb + ((lj == lb) ? v : ((lj == la) ? -v : 0.0f));
                    }
                } else {
                    for (int k0 = 0; k0 < W; k0 += 16) {     // padding entries: (the variable itself, +0.0f)
                        uint32_t cj[16];
                        float vj[16];
#pragma unroll
                        for (int k = 0; k < 16; ++k) {       // 32 loads in flight, then the gathers, then the sums
                            const size_t at = ((size_t)cur.slot * W + k0 + k) * 64 + lane;
                            cj[k] = a.ell_col[at];
                            vj[k] = a.ell_val[at];
                        }
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const int lj = lab[cj[k]];
                            hb = hb + ((lj == lb) ? vj[k] : ((lj == la) ? -vj[k] : 0.0f));
                        }
                    }
                }
            };
            K2_TICK(t_pre);
            sum_h();
            K2_TICK(t_apply);
            int ia = cnt[la] - 1, ib = cnt[lb];              // sizes of this lane's two clusters
            // one-hot images of this lane's two labels (K <= 32): a move a_s -> b_s then updates the sizes with
            // AND + bit-count on the vector unit alone, no compare results travelling through SGPRs
            const uint32_t oa = 1u << (la & 31), ob = 1u << (lb & 31);
            const bool onehot = K <= 32;                     // wave-uniform
            uint64_t todo = ~0ull, flipped = 0ull;
            if (has_in != 0ull) asm volatile("" ::"v"(e01.x), "v"(e01.z), "v"(e23.x), "v"(e23.z));
            auto commit_loop = [&](auto use_min_c, auto onehot_c) {
            constexpr bool UM = decltype(use_min_c)::value, OH = decltype(onehot_c)::value;
            float hd = hb - ha;
            while (true) {
                // dE = (h_b + c n_b) - (h_a + c (n_a - 1)) as one fma of the integer size difference onto the
                // field difference (oracle 2c evaluates the same expression)
                const float dE = fmaf(a.c_pair, (float)(ib - ia), hd);
                // ia = (members of this lane's cluster) - 1: a move may not shrink a cluster below min_size
                const uint64_t m = (UM ? __ballot(dE < thr && ia >= a.min_size) : __ballot(dE < thr)) & todo;
                if (m == 0) break;
                const int l = __ffsll((unsigned long long)m) - 1;
                todo = (~0ull << l) << 1;
                flipped |= 1ull << l;
                // integer sizes (the per-cluster table `cntv` is brought up to date once per slot, after the loop)
                if constexpr (OH) {
                    const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane((int)oa, l);
                    const uint32_t sb = (uint32_t)__builtin_amdgcn_readlane((int)ob, l);
                    ia += (int)__builtin_popcount(oa & sb) - (int)__builtin_popcount(oa & sa);
                    ib += (int)__builtin_popcount(ob & sb) - (int)__builtin_popcount(ob & sa);
                } else {
                    const int a_s = __builtin_amdgcn_readlane(la, l);
                    const int b_s = __builtin_amdgcn_readlane(lb, l);
                    ia += (int)(la == b_s) - (int)(la == a_s);
                    ib += (int)(lb == b_s) - (int)(lb == a_s);
                }
                if ((has_in >> l) & 1ull) {                  // wave-uniform: l has neighbours inside this slot
                    if (lane == l) lab[i] = (uint8_t)lb;
                    const int nin = (int)(__builtin_amdgcn_readlane((int)metav, l) & 0xff);
                    bool touched = lane == (int)(__builtin_amdgcn_readlane((int)e01.x, l) & 63);
                    if (nin > 1) touched |= lane == (int)(__builtin_amdgcn_readlane((int)e01.z, l) & 63);
                    if (nin > 2) touched |= lane == (int)(__builtin_amdgcn_readlane((int)e23.x, l) & 63);
                    if (nin > 3) touched |= lane == (int)(__builtin_amdgcn_readlane((int)e23.z, l) & 63);
                    for (int k = 4; k < nin; ++k) touched |= lane == (int)(rows[((size_t)t * 64 + l) * W + k].x & 63u);
                    if (touched) { sum_h(); hd = hb - ha; }
                }
            }
            };
            // the two wave-uniform switches select one of four straight-line copies of the loop (inside it they
            // would be a ladder of taken branches on the serial path)
            if (has_in == 0ull && K <= 16 && a.waves_override != 99) {
                // ---- no variable of this slot has a neighbour inside it (every slot under the slot-independent
                // order): a lane's decision depends on the movers below it only through the sizes of ITS two clusters.
                // As in K2, the accept mask of the sequential sweep is the one fixed point of "evaluate every lane
                // under a guessed mask, rebuild the mask" -- reached in 2-3 rounds instead of one trip through a
                // dependent vector -> scalar -> vector chain per mover.  The sizes the movers below lane i leave
                // behind come from ONE wave-wide prefix sum: lane j contributes a byte per cluster, 1 + [j moves into
                // it] - [j moves out of it] (0, 1 or 2: never a borrow between bytes, sums <= 128), so after the scan
                // byte q of lane i holds i + (net change of cluster q by the movers below i), and the index cancels in
                // the difference of the lane's two clusters.  K <= 8 clusters fill one 64-bit value, K <= 16 two.
                const bool two = K > 8;                           // wave-uniform
                auto unit = [&](int q) -> unsigned long long { return 1ull << ((q & 7) * 8); };
                const unsigned long long ones = 0x0101010101010101ull;
                // this lane's contribution when it moves (la -> lb), per 64-bit half
                unsigned long long mv0 = ones, mv1 = ones;
                if (lb < 8) mv0 += unit(lb); else mv1 += unit(lb);
                if (la < 8) mv0 -= unit(la); else mv1 -= unit(la);
                const int sh_a = (la & 7) * 8, sh_b = (lb & 7) * 8;
                const float hd = hb - ha;
                const int d0 = ib - ia;                           // (cnt_b) - (cnt_a - 1) before any move of this slot
                auto scan32 = [&](uint32_t v) -> uint32_t {       // inclusive prefix sum over the 64 lanes (DPP)
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
                    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
                    return v;
                };
                auto scan64 = [&](unsigned long long v) -> unsigned long long {
                    return ((unsigned long long)scan32((uint32_t)(v >> 32)) << 32) | scan32((uint32_t)v);
                };
                bool mine = fmaf(a.c_pair, (float)d0, hd) < thr && (!use_min || ia >= a.min_size);
                uint64_t A = __ballot(mine);
                if (A != 0ull) {                                  // wave-uniform
                    for (int round = 0; round < 66; ++round) {
                        // exclusive prefix: the inclusive scan minus the lane's own contribution
                        const unsigned long long c0 = mine ? mv0 : ones, c1 = mine ? mv1 : ones;
                        const unsigned long long s0 = scan64(c0) - c0;
                        const unsigned long long s1 = two ? scan64(c1) - c1 : 0ull;
                        const int na = (int)(((la < 8 ? s0 : s1) >> sh_a) & 0xffull);   // lane index + net change of cluster a
                        const int nb = (int)(((lb < 8 ? s0 : s1) >> sh_b) & 0xffull);
                        mine = fmaf(a.c_pair, (float)(d0 + nb - na), hd) < thr && (!use_min || ia + (na - lane) >= a.min_size);
                        const uint64_t A2 = __ballot(mine);
                        if (A2 == A) break;
                        A = A2;
                    }
                }
                flipped = A;
            } else if (!use_min && onehot && has_in == 0ull) {
                // The common case -- no size constraint, K <= 32, no variable of this slot with a neighbour inside
                // it (every slot under the slot-independent order) -- hand-scheduled like K2's loop: the lanes
                // above the last mover are selected by EXEC and each lane carries only the DIFFERENCE of its two
                // cluster sizes.
                float t0, t1;
                int l_s;
                uint32_t sa_s, sb_s;
                int idf = ib - ia;
                const float hd = hb - ha;
                if (K <= 16) {
                    // the mover's clusters travel as 2 x their ids; W holds a signed 2-bit field per cluster
                    // (+1 at the target, -1 at the source) and W' the negated fields, so a lane's change is one
                    // v_bfe_i32 per label and one three-operand add: 8 VALU + 11 SALU per move
                    const int la2 = la * 2, lb2 = lb * 2;
                    uint32_t w1_s, w2_s;
                    asm volatile(
                        "0:\n\t"
                        "v_cvt_f32_i32 %[t0], %[id]\n\t"
                        "v_fma_f32 %[t0], %[c], %[t0], %[hd]\n\t"
                        "v_cmp_lt_f32 vcc, %[t0], %[thr]\n\t"
                        "s_cbranch_vccz 1f\n\t"
                        "s_ff1_i32_b64 %[l], vcc\n\t"
                        "s_bitset1_b64 %[fl], %[l]\n\t"
                        "s_lshl_b64 exec, -2, %[l]\n\t"
                        "s_nop 1\n\t"                          // SALU-written lane select: 4 wait states
                        "v_readlane_b32 %[sa], %[la2], %[l]\n\t"
                        "v_readlane_b32 %[sb], %[lb2], %[l]\n\t"
                        "s_nop 0\n\t"
                        "s_lshl_b32 %[w1], 3, %[sa]\n\t"        // W : -1 at the source cluster,
                        "s_lshl_b32 %[w2], 1, %[sb]\n\t"        //     +1 at the target cluster
                        "s_or_b32 %[w1], %[w1], %[w2]\n\t"
                        "s_lshl_b32 %[sa], 1, %[sa]\n\t"        // W': the negated fields (for this lane's a-label)
                        "s_lshl_b32 %[sb], 3, %[sb]\n\t"
                        "s_or_b32 %[sa], %[sa], %[sb]\n\t"
                        "s_nop 0\n\t"
                        "v_bfe_i32 %[t0], %[w1], %[lb2], 2\n\t"  // change of n_b
                        "v_bfe_i32 %[t1], %[sa], %[la2], 2\n\t"  // -(change of n_a)
                        "v_add3_u32 %[id], %[id], %[t0], %[t1]\n\t"
                        "s_branch 0b\n"
                        "1:\n\t"
                        "s_mov_b64 exec, -1\n\t"
                        : [id] "+v"(idf), [fl] "+s"(flipped), [t0] "=&v"(t0), [t1] "=&v"(t1), [l] "=&s"(l_s),
                          [sa] "=&s"(sa_s), [sb] "=&s"(sb_s), [w1] "=&s"(w1_s), [w2] "=&s"(w2_s)
                        : [c] "s"(a.c_pair), [hd] "v"(hd), [thr] "v"(thr), [la2] "v"(la2), [lb2] "v"(lb2)
                        : "vcc", "scc");
                } else {
                    // 16 < K <= 32: one-hot images of the mover's clusters, four 1-bit extracts per move
                    asm volatile(
                        "0:\n\t"
                        "v_cvt_f32_i32 %[t0], %[id]\n\t"
                        "v_fma_f32 %[t0], %[c], %[t0], %[hd]\n\t"
                        "v_cmp_lt_f32 vcc, %[t0], %[thr]\n\t"
                        "s_cbranch_vccz 1f\n\t"
                        "s_ff1_i32_b64 %[l], vcc\n\t"
                        "s_bitset1_b64 %[fl], %[l]\n\t"
                        "s_lshl_b64 exec, -2, %[l]\n\t"
                        "s_nop 1\n\t"
                        "v_readlane_b32 %[sa], %[oa], %[l]\n\t"
                        "v_readlane_b32 %[sb], %[ob], %[l]\n\t"
                        "s_nop 1\n\t"
                        "v_bfe_u32 %[t0], %[sb], %[lb], 1\n\t"      // [lb == b_s]
                        "v_bfe_u32 %[t1], %[sa], %[lb], 1\n\t"      // [lb == a_s]
                        "v_sub_u32 %[t0], %[t0], %[t1]\n\t"
                        "v_add_u32 %[id], %[id], %[t0]\n\t"
                        "v_bfe_u32 %[t0], %[sb], %[la], 1\n\t"      // [la == b_s]
                        "v_bfe_u32 %[t1], %[sa], %[la], 1\n\t"      // [la == a_s]
                        "v_sub_u32 %[t0], %[t0], %[t1]\n\t"
                        "v_sub_u32 %[id], %[id], %[t0]\n\t"
                        "s_branch 0b\n"
                        "1:\n\t"
                        "s_mov_b64 exec, -1\n\t"
                        : [id] "+v"(idf), [fl] "+s"(flipped), [t0] "=&v"(t0), [t1] "=&v"(t1), [l] "=&s"(l_s),
                          [sa] "=&s"(sa_s), [sb] "=&s"(sb_s)
                        : [c] "s"(a.c_pair), [hd] "v"(hd), [thr] "v"(thr), [oa] "v"(oa), [ob] "v"(ob), [la] "v"(la),
                          [lb] "v"(lb)
                        : "vcc", "scc");
                }
            } else if (use_min) { if (onehot) commit_loop(std::true_type{}, std::true_type{}); else commit_loop(std::true_type{}, std::false_type{}); }
            else { if (onehot) commit_loop(std::false_type{}, std::true_type{}); else commit_loop(std::false_type{}, std::false_type{}); }
            if (flipped) {                                   // wave-uniform
                accepted += (unsigned long long)__popcll(flipped);
                const bool moved = (flipped >> lane) & 1ull;
                if (moved) lab[i] = (uint8_t)lb;
                // cluster sizes: two LDS atomics per mover (the wave's LDS operations execute in order, so the
                // next slot's reads of cnt[] see them)
                if (moved) {
                    __hip_atomic_fetch_add(&cnt[lb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(&cnt[la], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            K2_TICK(t_loop);
        }
    }
#ifdef MI_K2_PROFILE
    if (lane == 0) {
        atomicAdd(&a.stats[8], t_pre); atomicAdd(&a.stats[9], t_loop); atomicAdd(&a.stats[11], t_apply); atomicAdd(&a.stats[12], t_init);
    }
#endif

    // ---- epilogue: labels out, exact fp64 energy ----
    uint16_t *dst = static_cast<uint16_t *>(a.states) + (size_t)r * n;
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        if (i >= n) continue;
        const int li = lab[i];
        dst[i] = (uint16_t)li;
        for (int k = 0; k < W; ++k) {
            const uint32_t cc = a.ell_col[((size_t)t * W + k) * 64 + lane];
            const size_t at = ((size_t)t * W + k) * 64 + lane;
            const double vv = a.ell_val64 ? a.ell_val64[at] : (double)a.ell_val[at];
            if ((int)cc > i && lab[cc] == li) e += vv;
        }
    }
    cntv = cnt[lane];
    if (lane < K) e += (a.ell_val64 ? a.c_pair64 : (double)a.c_pair) * 0.5 * (double)cntv * (double)(cntv - 1);
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

template <typename KernelT>
int launch_sparse(KernelT kernel, const EllArgs &a, size_t lds_per_wave, hipStream_t st)
{
    note_kernel("k_anneal_potts<%d>", a.D <= 64 ? a.D : 0);
    int waves = kSparseWaves;                    // fewer replicas per workgroup when their LDS state is large
    while (waves > 1 && lds_per_wave * waves > 160 * 1024) --waves;
    const size_t lds = lds_per_wave * waves;
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "model too large for the LDS-resident sparse kernel (%zu B)", lds);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kernel, dim3((a.R + waves - 1) / waves), dim3(waves * 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

template <typename KernelT>
int launch_csr_rank1(KernelT kernel, const EllArgs &a0, size_t lds, hipStream_t st, bool tw = false)
{
    // one wavefront = one replica = one workgroup; the only LDS is the state (a bit or a byte per variable) -- and, with a
    // threshold wavefront beside the sweeping one, the ring of thresholds behind it
    EllArgs a = a0;
    if (tw) {
        a.ring_off = (int)((lds + 15) / 16 * 16);
        lds = (size_t)a.ring_off + 2048;
    }
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "csr_rank1: n = %d exceeds the state LDS budget", a.n);
    if (!a.adj4 || !a.slot_flags) return fail(MI_EHIP, "csr_rank1: packed adjacency missing");
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel(tw ? "k_anneal_csr_rank1<%d, %d, tw>" : "k_anneal_csr_rank1<%d, %d>", a.D <= 64 ? a.D : 0, a.state_bytes);
    hipLaunchKernelGGL(kernel, dim3(a.R), dim3(tw ? 128 : 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

int mi_launch_csr_rank1(const EllArgs &a, hipStream_t st, bool tw)
{
    const size_t bits = (size_t)a.slots * 8, bytes = (size_t)a.slots * 64, halves = (size_t)a.slots * 128;
    if (tw) {          // built for the register-resident widths of large models: bit and byte state
        if (a.state_bytes == 1 && a.D == 16) return launch_csr_rank1(k_anneal_csr_rank1<16, 1, true>, a, bytes, st, true);
        if (a.state_bytes == 1 && a.D == 32) return launch_csr_rank1(k_anneal_csr_rank1<32, 1, true>, a, bytes, st, true);
        if (a.state_bytes == 0 && a.D == 16) return launch_csr_rank1(k_anneal_csr_rank1<16, 0, true>, a, bits, st, true);
        if (a.state_bytes == 0 && a.D == 32) return launch_csr_rank1(k_anneal_csr_rank1<32, 0, true>, a, bits, st, true);
    }
    if (a.state_bytes == 2) {
        if (a.D == 16) return launch_csr_rank1(k_anneal_csr_rank1<16, 2>, a, halves, st);
        if (a.D == 32) return launch_csr_rank1(k_anneal_csr_rank1<32, 2>, a, halves, st);
        if (a.D == 64) return launch_csr_rank1(k_anneal_csr_rank1<64, 2>, a, halves, st);
        if (a.D > 64 && a.D % 16 == 0) return launch_csr_rank1(k_anneal_csr_rank1<0, 2>, a, halves, st);
    } else if (a.state_bytes == 1) {
        if (a.D == 16) return launch_csr_rank1(k_anneal_csr_rank1<16, 1>, a, bytes, st);
        if (a.D == 32) return launch_csr_rank1(k_anneal_csr_rank1<32, 1>, a, bytes, st);
        if (a.D == 64) return launch_csr_rank1(k_anneal_csr_rank1<64, 1>, a, bytes, st);
        if (a.D > 64 && a.D % 16 == 0) return launch_csr_rank1(k_anneal_csr_rank1<0, 1>, a, bytes, st);
    } else {
        if (a.D == 16) return launch_csr_rank1(k_anneal_csr_rank1<16, 0>, a, bits, st);
        if (a.D == 32) return launch_csr_rank1(k_anneal_csr_rank1<32, 0>, a, bits, st);
        if (a.D == 64) return launch_csr_rank1(k_anneal_csr_rank1<64, 0>, a, bits, st);
        if (a.D > 64 && a.D % 16 == 0) return launch_csr_rank1(k_anneal_csr_rank1<0, 0>, a, bits, st);
    }
    return fail(MI_EUNSUPPORTED, "slot-ELL width %d not built", a.D);
}

int mi_launch_potts(const EllArgs &a, hipStream_t st)
{
    const size_t per_wave = (size_t)a.slots * 64 + 256;      // labels + cluster sizes
    if (a.D == 16) return launch_sparse(k_anneal_potts<16>, a, per_wave, st);
    if (a.D == 32) return launch_sparse(k_anneal_potts<32>, a, per_wave, st);
    if (a.D == 64) return launch_sparse(k_anneal_potts<64>, a, per_wave, st);
    if (a.D > 64 && a.D % 16 == 0) return launch_sparse(k_anneal_potts<0>, a, per_wave, st);     // rows of any width
    return fail(MI_EUNSUPPORTED, "slot-ELL width %d not built", a.D);
}

}  // namespace mi_sa_impl
