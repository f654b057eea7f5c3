// potts_fast_kernels.hip -- K3f: the Potts (DQM) chain for the common case, built around its field sum (gfx950 only).
//
// Which models: every 64-seat slot free of internal edges (what the sampler's layouts give), 2 <= K <= 16 labels, 16 or
// 32 adjacency entries per variable (a minimum cluster size included: template switch UM).  Everything else runs on k_anneal_potts
// (sparse_kernels.hip) -- the same chain (oracle/sa_oracle.c 2c), bit for bit.
//
// Why a kernel of its own.  With four wavefronts per SIMD k_anneal_potts is bound by the NUMBER of vector instructions
// (312 per slot, profiles/r02_k3_binding.json), and a third of them are the field sum: per neighbour a compare, a select
// and an add for EACH of the lane's two labels.  The chain asks for one number per proposal,
//     hd = sum_k S_ik * sigma_k,   sigma_k = [l_k == b] - [l_k == a]      (fp32, stored order, one fma per neighbour)
// and sigma_k is a lookup of the neighbour's label in a per-lane table with two non-zero entries:
//   * K <= 8 (KM = 8): the table is eight BYTES in a register pair -- 0x40 at b, 0xc0 at a, 0 elsewhere -- and the
//     neighbour's LDS cell, 16 bits holding (label << 8) | 0x0c, is the selector of a v_perm_b32: byte 1 of the result is
//     the table entry of the label, byte 0 the constant 0 (selector 0x0c), so the LOW HALF of the result IS the fp16
//     number +2.0 (0x4000), -2.0 (0xc000) or 0.0, which v_fma_mix_f32 takes as it is.  One permute + one fma per
//     neighbour.  (The sum comes out doubled -- exactly: a power of two commutes with every rounding -- and is compared
//     with the doubled threshold.)
//   * K <= 16 (KM = 16): the table is sixteen 2-bit fields in one register (01 at b, 11 at a), the cell is 2 * label,
//     and v_bfe_i32 with the cell as its offset returns +1, -1 or 0: three instructions per neighbour.
// The rest is k_anneal_potts' fast path made lean: the accept mask of a slot by fixed-point rounds over a DPP prefix
// sum of packed per-cluster size changes, with only wave-uniform masks alive across the loop; the cluster sizes a lane
// needs picked out of the scan by one more v_perm_b32 each; cluster sizes in LDS (two atomics per mover).
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

template <int KM>
__device__ __forceinline__ uint32_t enc_label(uint32_t q) { return KM == 8 ? ((q << 8) | 0x0cu) : (q << 1); }
template <int KM>
__device__ __forceinline__ uint32_t dec_label(uint32_t c) { return KM == 8 ? (c >> 8) : (c >> 1); }

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts)
__device__ __forceinline__ uint32_t wave_scan32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return v;
}

// per-lane select by a wave-uniform mask held in a scalar register pair
__device__ __forceinline__ uint32_t select_by_mask(uint32_t if_clear, uint32_t if_set, uint64_t mask)
{
    uint32_t out;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(if_clear), "v"(if_set), "s"(mask));
    return out;
}

// UM: a minimum cluster size is set (CQM_clustering.py:46-48 as a hard constraint): a move out of a cluster that would
// be left with fewer than min_size members is rejected whatever its dE -- the size a lane sees is the one the movers
// below it leave behind, which the same prefix scan delivers
// TW: a second wavefront of the workgroup computes what does not depend on the labels -- the threshold -ln(u) * T and the
// target offset word mod (K - 1) of every proposal -- one group of four slots ahead and hands them over through a two-deep
// ring in LDS (as k_anneal_csr_rank1_pair does): for runs of up to 1024 replicas (the sampler's default is 256 reads), where
// every wavefront has a SIMD to itself and that work moves to an idle one.
template <int D, int KM, bool UM, bool TW>
__global__ void __launch_bounds__(TW ? 128 : 64, (TW ? 2 : (D == 16 ? 4 : 2))) k_anneal_potts_fast(EllArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];      // cell of seat i at byte 2 i, then the K cluster sizes
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x;
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots, K = a.K;
    uint16_t *cell = reinterpret_cast<uint16_t *>(lds);
    const uint32_t cnt_base = (uint32_t)slots * 128u;
    int *cnt = reinterpret_cast<int *>(lds + cnt_base);
    const uint16_t *init = static_cast<const uint16_t *>(a.init);
    // w mod (K-1) without the integer division: q = mulhi(w, floor(2^32 / d)) is floor(w / d) or one less
    const uint32_t dK = (uint32_t)(K - 1);
    const uint32_t magic = dK == 1u ? 0xffffffffu : (uint32_t)(0x100000000ull / dK);
    // TW: the ring behind the cluster sizes, 2 groups x 4 slots x 64 lanes x (threshold, target offset)
    const uint32_t ring_lane = cnt_base + 256u + (uint32_t)lane * 8u;
    if constexpr (TW) {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1) {
            // ---- the threshold wavefront ----
            uint32_t pw0[4], pw2[4];
            uint32_t buf = 0;
            for (int s = 0; s < a.num_sweeps; ++s) {
                float Tp = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s])));
                if constexpr (KM == 8) Tp = 2.0f * Tp;
                const uint32_t sw = (uint32_t)s + a.sweep_offset;
#pragma unroll 1
                for (int t = 0; t < slots; t += 4) {
                    philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gid, 0u, a.seed_lo, a.seed_hi, pw0);
                    philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gid, 2u, a.seed_lo, a.seed_hi, pw2);
                    const f32x2_t l01 = neglog_u2(pw0[0], pw0[1]) * f32x2_t{Tp, Tp}, l23 = neglog_u2(pw0[2], pw0[3]) * f32x2_t{Tp, Tp};
                    const float thr4[4] = {l01.x, l01.y, l23.x, l23.y};
                    const uint32_t at = ring_lane + buf;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        uint32_t rem = pw2[c] - __umulhi(pw2[c], magic) * dK;     // in [0, 2 dK)
                        rem = rem >= dK ? rem - dK : rem;
                        const u32x2 pk = {__float_as_uint(thr4[c]), rem};
                        asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(at), "v"(pk), "n"(c * 512) : "memory");
                    }
                    buf ^= 2048u;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            return;
        }
        __builtin_amdgcn_s_setprio(3);
    }

    // ---- initial labels (the chain's tag-1 words, or the caller's / the previous launch's states) ----
    for (int tg = 0; tg * 4 < slots; ++tg) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (!init) philox4x32_10((uint32_t)(tg * 64 + lane), 0u, gid, 1u, a.seed_lo, a.seed_hi, w);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = 4 * tg + c;
            if (t >= slots) break;
            const int i = t * 64 + lane;
            // (bit 31 of meta: nobody sits here -- past n, or a hole of a padded layout: label 0, in no cluster, never proposed)
            const bool present = (a.meta[i] >> 31) == 0u;
            uint32_t v = 0u;
            if (present) v = init ? (uint32_t)init[(size_t)r * n + i] : (w[c] % (uint32_t)K);
            cell[i] = (uint16_t)enc_label<KM>(v);
        }
    }
    {
        int cntv = 0;                                               // lane q: number of variables with label q
        for (int q = 0; q < K; ++q) {
            int c = 0;
            for (int t = 0; t < slots; ++t)
                c += __popcll(__ballot((a.meta[t * 64 + lane] >> 31) == 0u && dec_label<KM>(cell[t * 64 + lane]) == (uint32_t)q));
            if (lane == q) cntv = c;
        }
        cnt[lane] = cntv;                                           // (64 words reserved)
    }

    constexpr int G = D / 4;
    const __amdgpu_buffer_rsrc_t rs_adj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.adj4), 0, slots * G * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_meta = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(a.meta), 0, slots * 256, 0x00020000);
    struct SlotAdj { u32x4 col[G]; u32x4 val[G]; uint32_t meta; };
    const int lane16 = lane * 16;
#ifdef MI_K3F_DBG_NOFETCH   /* timing only (wrong chain): slot 0's adjacency serves every slot, no vector-memory traffic in the sweep */
    auto fetch_real = [&](int t) {
#else
    auto fetch_adj = [&](int t) {
#endif
        SlotAdj p;
        const int tt = t < slots ? t : slots - 1;
        const int soff = tt * (G * 2048);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int so = soff + (g / 2) * 4096, io = (g & 1) * 2048;
            p.col[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane16 + io, so, 0);
            p.val[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane16 + io + 1024, so, 0);
        }
        p.meta = __builtin_amdgcn_raw_buffer_load_b32(rs_meta, lane * 4, tt * 256, 0);
        return p;
    };

#ifdef MI_K3F_DBG_NOFETCH
    const SlotAdj adj0 = fetch_real(0);
    auto fetch_adj = [&](int) { return adj0; };
#endif

    const float c_eff = KM == 8 ? 2.0f * a.c_pair : a.c_pair;       // (KM = 8: the field sum comes out doubled)
    const bool narrow = K <= 4;                                     // wave-uniform: the size bytes of all clusters fill one dword

    uint32_t sel[16];                                               // the gathered cells
    unsigned long long accepted = 0;
    uint32_t acc32 = 0;
    uint32_t w0[4] = {0u, 0u, 0u, 0u}, w2[4] = {0u, 0u, 0u, 0u};
    float T = 1.0f;

    typedef _Float16 half_t;
    uint32_t ring_buf = 0u;                                         // TW: which half of the ring holds the group being swept
    auto slot_body = [&](auto c_in_group, int t, const SlotAdj &cur, uint32_t w0c, uint32_t w2c) {
        constexpr int C = decltype(c_in_group)::value;
        const int i = t * 64 + lane;
        uint32_t own;
        float hd = 0.0f;
        float thr = 0.0f;
        u32x2 tw2 = {0u, 0u};                                       // TW: (threshold bits, target offset) from the ring
        uint32_t la = 0u, lb = 0u, tlo = 0u, thi = 0u;
        int na0, nb0;                                               // sizes of the lane's two clusters
        auto proposal = [&](uint32_t rem) {
            // a = the lane's label, b = (a + 1 + word mod (K - 1)) mod K; and the lookup table of sigma
            la = dec_label<KM>(own);
            lb = la + 1u + rem;                                     // in [1, 2K - 2]
            lb = lb >= (uint32_t)K ? lb - (uint32_t)K : lb;
            if constexpr (KM == 8) {
                const uint64_t tbl = (0xc0ull << (la * 8u)) | (0x40ull << (lb * 8u));
                tlo = (uint32_t)tbl;
                thi = (uint32_t)(tbl >> 32);
            } else {
                tlo = (3u << (la * 2u)) | (1u << (lb * 2u));
            }
        };
        auto fma_k = [&](int g0, int k) {
            const float v = __uint_as_float(cur.val[g0 + k / 4][k & 3]);
            if constexpr (KM == 8) {
                const uint32_t pk = __builtin_amdgcn_perm(thi, tlo, sel[k]);                     // low half: fp16 +2, -2 or 0
                hd = __builtin_fmaf(v, (float)__builtin_bit_cast(half_t, (uint16_t)pk), hd);
            } else {
                hd = __builtin_fmaf(v, (float)(int)__builtin_amdgcn_sbfe((int)tlo, sel[k], 2u), hd);   // +1, -1 or 0 (the builtin's result type is unsigned)
            }
        };
        if constexpr (TW && G == 4) {
            // A wavefront alone on its SIMD (runs of up to 1024 replicas) is parked on s_waitcnt half of its time if it waits
            // for all its LDS reads at once: the own cell and the ring entry go FIRST, the proposal and its table are worked
            // out while the gathers arrive, the two cluster sizes are requested then, and the fma chain follows the gathers
            // in stages of four behind counted waits (LDS reads return in order; every wait names what the stage before it
            // produced, so nothing sinks below it)
            asm volatile("ds_read_u16 %0, %1" : "=v"(own) : "v"(i * 2));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tw2) : "v"(ring_lane + ring_buf), "n"(C * 512));
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("ds_read_u16 %0, %1" : "=v"(sel[k]) : "v"(cur.col[k / 4][k & 3]));
            asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(own), "+v"(tw2), "+v"(sel[0]) :: "memory");        // 18 issued: the 3 oldest are back
            thr = __uint_as_float(tw2.x);
            if (cur.meta >> 31) thr = -INFINITY;                    // nobody sits here
            proposal(tw2.y);
            asm volatile("ds_read_b32 %0, %1" : "=v"(na0) : "v"(cnt_base + la * 4u));
            asm volatile("ds_read_b32 %0, %1" : "=v"(nb0) : "v"(cnt_base + lb * 4u));
            asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(sel[1]), "+v"(sel[2]), "+v"(sel[3]), "+v"(tlo), "+v"(thi) :: "memory");   // 20 issued: 6 back
#pragma unroll
            for (int k = 0; k < 4; ++k) fma_k(0, k);
            asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(sel[4]), "+v"(sel[5]), "+v"(sel[6]), "+v"(sel[7]), "+v"(hd) :: "memory");
#pragma unroll
            for (int k = 4; k < 8; ++k) fma_k(0, k);
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(sel[8]), "+v"(sel[9]), "+v"(sel[10]), "+v"(sel[11]), "+v"(hd) :: "memory");
#pragma unroll
            for (int k = 8; k < 12; ++k) fma_k(0, k);
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(sel[12]), "+v"(sel[13]), "+v"(sel[14]), "+v"(sel[15]), "+v"(hd) :: "memory");
#pragma unroll
            for (int k = 12; k < 16; ++k) fma_k(0, k);
        } else {
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            if (g0 == 0) asm volatile("ds_read_u16 %0, %1" : "=v"(own) : "v"(i * 2));
#pragma unroll
            for (int k = 0; k < 16; ++k)
#ifdef MI_K3F_DBG_LINEAR   /* timing only: conflict-free gather addresses */
                asm volatile("ds_read_u16 %0, %1" : "=v"(sel[k]) : "v"((cur.col[g0 + k / 4][k & 3] & 0x1f80u) + lane * 2));
#else
                asm volatile("ds_read_u16 %0, %1" : "=v"(sel[k]) : "v"(cur.col[g0 + k / 4][k & 3]));
#endif
            if (g0 == 0) {
                uint32_t rem;
                if constexpr (TW) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tw2) : "v"(ring_lane + ring_buf), "n"(C * 512));
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(sel[0]), "+v"(sel[1]), "+v"(sel[2]), "+v"(sel[3]), "+v"(sel[4]), "+v"(sel[5]),
                                   "+v"(sel[6]), "+v"(sel[7]), "+v"(sel[8]), "+v"(sel[9]), "+v"(sel[10]), "+v"(sel[11]),
                                   "+v"(sel[12]), "+v"(sel[13]), "+v"(sel[14]), "+v"(sel[15]), "+v"(own), "+v"(tw2)
                                 :: "memory");
                    thr = __uint_as_float(tw2.x);
                    rem = tw2.y;
                    if (cur.meta >> 31) thr = -INFINITY;            // nobody sits here
                } else {
                    asm volatile("" : "+v"(w0c));                   // (keeps the threshold arithmetic behind the reads' issue)
                    thr = neglog_u(w0c) * T;
                    if (cur.meta >> 31) thr = -INFINITY;            // nobody sits here
                    asm volatile("s_waitcnt lgkmcnt(0)"
                                 : "+v"(sel[0]), "+v"(sel[1]), "+v"(sel[2]), "+v"(sel[3]), "+v"(sel[4]), "+v"(sel[5]),
                                   "+v"(sel[6]), "+v"(sel[7]), "+v"(sel[8]), "+v"(sel[9]), "+v"(sel[10]), "+v"(sel[11]),
                                   "+v"(sel[12]), "+v"(sel[13]), "+v"(sel[14]), "+v"(sel[15]), "+v"(own), "+v"(thr)
                                 :: "memory");
                    rem = w2c - __umulhi(w2c, magic) * dK;          // in [0, 2 dK)
                    rem = rem >= dK ? rem - dK : rem;
                }
                proposal(rem);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(sel[0]), "+v"(sel[1]), "+v"(sel[2]), "+v"(sel[3]), "+v"(sel[4]), "+v"(sel[5]),
                               "+v"(sel[6]), "+v"(sel[7]), "+v"(sel[8]), "+v"(sel[9]), "+v"(sel[10]), "+v"(sel[11]),
                               "+v"(sel[12]), "+v"(sel[13]), "+v"(sel[14]), "+v"(sel[15])
                             :: "memory");
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) fma_k(g0, k);
        }
        // sizes of the lane's two clusters (the atomics of the previous slot are ordered before these reads)
        asm volatile("ds_read_b32 %0, %1" : "=v"(na0) : "v"(cnt_base + la * 4u));
        asm volatile("ds_read_b32 %0, %1" : "=v"(nb0) : "v"(cnt_base + lb * 4u));
        }
        // a mover's contribution to the packed per-cluster bytes: 1 everywhere, +1 at its target, -1 at its source (0, 1
        // or 2: no borrow between bytes, sums <= 128); byte q of the inclusive scan at lane i = (i + 1) + net change of
        // cluster q by the movers up to and including i.  K <= 8: one 64-bit value (K <= 4: its low dword), else two.
        const uint64_t ones = 0x0101010101010101ull;
        const uint64_t ub = 1ull << ((lb & 7u) * 8u), ua = 1ull << ((la & 7u) * 8u);
        uint64_t mvA, mvB = ones;
        if constexpr (KM == 8) {
            mvA = ones + ub - ua;
        } else {
            mvA = ones + (lb < 8u ? ub : 0ull) - (la < 8u ? ua : 0ull);
            mvB = ones + (lb < 8u ? 0ull : ub) - (la < 8u ? 0ull : ua);
        }
        // byte selectors of the lane's two clusters inside a 64-bit scan (v_perm_b32: the byte lands in byte 0, zeros above)
        const uint32_t pa = 0x0c0c0c00u | (la & 7u), pb = 0x0c0c0c00u | (lb & 7u);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(na0), "+v"(nb0) :: "memory");
        const int d0 = nb0 - na0 + 1;                               // cnt_b - (cnt_a - 1) before any move of this slot
        uint64_t A0 = __ballot(__builtin_fmaf(c_eff, (float)d0, hd) < thr);
        if constexpr (UM) A0 &= __ballot(na0 - 1 >= a.min_size);
        if (A0 != 0ull) {                                           // wave-uniform
            // fixed-point rounds (ends by itself: a lane's decision depends on the movers below it only, so after k rounds
            // the lowest k lanes are final); only the wave-uniform mask lives across the loop
            uint64_t A = A0;
#pragma nounroll
            for (;;) {
                const uint32_t own2 = select_by_mask(0u, 2u, A);    // the lane's own move inside the inclusive scan
                uint32_t sa, sb;
                if (KM == 8 && narrow) {
                    const uint32_t s0 = wave_scan32(select_by_mask((uint32_t)ones, (uint32_t)mvA, A));
                    sa = __builtin_amdgcn_perm(0u, s0, pa);
                    sb = __builtin_amdgcn_perm(0u, s0, pb);
                } else {
                    const uint32_t s0 = wave_scan32(select_by_mask((uint32_t)ones, (uint32_t)mvA, A));
                    const uint32_t s1 = wave_scan32(select_by_mask((uint32_t)ones, (uint32_t)(mvA >> 32), A));
                    sa = __builtin_amdgcn_perm(s1, s0, pa);
                    sb = __builtin_amdgcn_perm(s1, s0, pb);
                    if constexpr (KM == 16) {
                        const uint32_t s2 = wave_scan32(select_by_mask((uint32_t)ones, (uint32_t)mvB, A));
                        const uint32_t s3 = wave_scan32(select_by_mask((uint32_t)ones, (uint32_t)(mvB >> 32), A));
                        const uint32_t ta = __builtin_amdgcn_perm(s3, s2, pa), tb = __builtin_amdgcn_perm(s3, s2, pb);
                        sa = la < 8u ? sa : ta;
                        sb = lb < 8u ? sb : tb;
                    }
                }
                const int d = d0 + (int)sb - (int)sa - (int)own2;
                uint64_t A2 = __ballot(__builtin_fmaf(c_eff, (float)d, hd) < thr);
                if constexpr (UM) {
                    // members of the lane's cluster when its turn comes: the start value + the net change by the movers
                    // BELOW it = the inclusive scan byte minus the lane's own contribution (0 as a mover, 1 otherwise)
                    // minus the (lane) ones of the lanes below
                    const int size_a = na0 + (int)sa + (int)(own2 >> 1) - (lane + 1);
                    A2 &= __ballot(size_a - 1 >= a.min_size);
                }
                uint64_t df = A2 ^ A;
                asm("" : "+s"(df));
                A = A2;
                if (df == 0ull) break;
            }
            acc32 += (uint32_t)__popcll(A);
            // movers: the new cell (every lane stores: the others their old one), and the two cluster sizes under the mask
            // (the LDS operations of a wavefront execute in order: the next slot's reads of the sizes see them)
            const uint32_t newc = select_by_mask(own, enc_label<KM>(lb), A);
            asm volatile("ds_write_b16 %0, %1" :: "v"(i * 2), "v"(newc) : "memory");
#ifndef MI_K3F_DBG_NOATOMICS   /* (timing only: cluster sizes never move) */
            asm volatile("s_mov_b64 exec, %0\n\t"
                         "ds_add_u32 %1, %3\n\t"
                         "ds_sub_u32 %2, %3\n\t"
                         "s_mov_b64 exec, -1"
                         :: "s"(A), "v"(cnt_base + lb * 4u), "v"(cnt_base + la * 4u), "v"(1u) : "memory");
#endif
        }
    };

    for (int s = 0; s < a.num_sweeps; ++s) {
        T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s])));
        if constexpr (KM == 8) T = 2.0f * T;                        // (the doubled field sum against the doubled threshold)
        const uint32_t sw = (uint32_t)s + a.sweep_offset;
        SlotAdj P = fetch_adj(0), Q;
        using std::integral_constant;
#pragma unroll 1
        for (int t = 0; t < slots; t += 4) {
            if constexpr (TW) {
                __builtin_amdgcn_s_barrier();                       // this group's thresholds and target offsets are in the ring
            } else {
                philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gid, 0u, a.seed_lo, a.seed_hi, w0);
                philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gid, 2u, a.seed_lo, a.seed_hi, w2);
            }
            Q = fetch_adj(t + 1);
            slot_body(integral_constant<int, 0>{}, t, P, w0[0], w2[0]);
            if (t + 1 < slots) {                                    // wave-uniform
                P = fetch_adj(t + 2);
                slot_body(integral_constant<int, 1>{}, t + 1, Q, w0[1], w2[1]);
                if (t + 2 < slots) {
                    Q = fetch_adj(t + 3);
                    slot_body(integral_constant<int, 2>{}, t + 2, P, w0[2], w2[2]);
                    P = fetch_adj(t + 4);
                    if (t + 3 < slots) slot_body(integral_constant<int, 3>{}, t + 3, Q, w0[3], w2[3]);
                }
            }
            if constexpr (TW) ring_buf ^= 2048u;
        }
        accepted += acc32;
        acc32 = 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- epilogue: labels out, exact fp64 energy (the sums of k_anneal_potts) ----
    uint16_t *dst = static_cast<uint16_t *>(a.states) + (size_t)r * n;
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        if (i >= n) continue;
        const uint32_t li = dec_label<KM>(cell[i]);
        dst[i] = (uint16_t)li;
        for (int k = 0; k < D; ++k) {
            const size_t at = ((size_t)t * D + k) * 64 + lane;
            const uint32_t cc = a.ell_col[at];
            const double vv = a.ell_val64 ? a.ell_val64[at] : (double)a.ell_val[at];
            if ((int)cc > i && dec_label<KM>(cell[cc]) == li) e += vv;
        }
    }
    const int cntv = cnt[lane];
    if (lane < K) e += (a.ell_val64 ? a.c_pair64 : (double)a.c_pair) * 0.5 * (double)cntv * (double)(cntv - 1);
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

template <typename KernelT>
int launch_potts_fast(KernelT kernel, const EllArgs &a, int km, bool tw, hipStream_t st)
{
    const size_t lds = (size_t)a.slots * 128 + 256 + (tw ? 4096 : 0);   // 2 bytes per seat + the cluster sizes (+ the ring)
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "potts fast kernel: n = %d exceeds the label LDS budget", a.n);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel(tw ? "k_anneal_potts_fast<%d, %d, tw>" : "k_anneal_potts_fast<%d, %d>", a.D, km);
    hipLaunchKernelGGL(kernel, dim3(a.R), dim3(tw ? 128 : 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

template <int D, int KM>
int launch_potts_fast_dk(const EllArgs &a, bool tw, hipStream_t st)
{
    const bool um = a.min_size > 0;
    if (tw) return um ? launch_potts_fast(k_anneal_potts_fast<D, KM, true, true>, a, KM, true, st)
                      : launch_potts_fast(k_anneal_potts_fast<D, KM, false, true>, a, KM, true, st);
    return um ? launch_potts_fast(k_anneal_potts_fast<D, KM, true, false>, a, KM, false, st)
              : launch_potts_fast(k_anneal_potts_fast<D, KM, false, false>, a, KM, false, st);
}

}  // namespace

bool mi_potts_fast_eligible(int D, int K, int min_size)
{
    (void)min_size;                                                 // (any: the kernels with the size test are built too)
    return (D == 16 || D == 32) && K >= 2 && K <= 16;
}

// a.adj4 = the packed adjacency with neighbour word = 2 * index (the byte address of the neighbour's 16-bit cell);
// tw: a threshold wavefront beside the sweeping one (runs of up to 1024 replicas)
int mi_launch_potts_fast(const EllArgs &a, bool tw, hipStream_t st)
{
    if (!a.adj4) return fail(MI_EHIP, "potts fast kernel: packed adjacency missing");
    if (!mi_potts_fast_eligible(a.D, a.K, a.min_size)) return fail(MI_EUNSUPPORTED, "potts fast kernel: not built for this model");
    if (a.K <= 8) return a.D == 16 ? launch_potts_fast_dk<16, 8>(a, tw, st) : launch_potts_fast_dk<32, 8>(a, tw, st);
    return a.D == 16 ? launch_potts_fast_dk<16, 16>(a, tw, st) : launch_potts_fast_dk<32, 16>(a, tw, st);
}

}  // namespace mi_sa_impl
