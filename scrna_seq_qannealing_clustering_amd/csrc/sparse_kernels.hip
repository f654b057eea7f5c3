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
// LDS per wave: g[slots*64] floats (cached g_i = lin_i + sum_j S_ij x_j), then 2*D words of scratch
// through which the committing lane hands its adjacency row to lanes 0..D-1:
//   committing lane: 2D ds_write_b32;  lane k < D: reads its (col, val), then g[col] += sgn*val (one IEEE
//   fp32 add, the oracle's rounding).  (Measured: ds_write_b128 + ds_add_f32 in place of this was 15 % SLOWER.)
// At D = 16 a flip costs about as many instructions here as a whole dense row update in K1w, so K2's role
// is the sizes the dense kernels cannot hold (n > 4096), not speed at n ~ 2.6k.
// State bits: up to four 64-slot masks per lane (n <= 16384).
constexpr int kK2Masks = 4;

template <int D>
__device__ __forceinline__ void k2_apply_row(float *g, uint32_t *scr, const uint32_t (&colv)[D],
                                             const float (&valv)[D], int lane, int l, float sgn)
{
    if (lane == l) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            scr[k] = colv[k];
            scr[D + k] = __float_as_uint(valv[k]);
        }
    }
    // same wave: LDS operations execute in order, the reads below see the writes above
    if (lane < D) {
        const uint32_t c = scr[lane];
        const float v = __uint_as_float(scr[D + lane]);
        // padding entries are (self, +0.0f): several lanes may rewrite g[self] with the same value
        g[c] = g[c] + sgn * v;          // one fp32 add per touched field, in flip order (oracle 2b)
    }
}

template <int D>
__global__ void __launch_bounds__(256) k_anneal_csr_rank1(EllArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (r >= a.R) return;                                   // no workgroup-level synchronisation below
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots;
    const size_t per_wave = (size_t)slots * 64 * 4 + 2 * D * 4;
    float *g = reinterpret_cast<float *>(lds + wave * per_wave);
    uint32_t *scr = reinterpret_cast<uint32_t *>(lds + wave * per_wave + (size_t)slots * 64 * 4);
    const uint8_t *init = static_cast<const uint8_t *>(a.init);

    uint64_t xb0 = 0, xb1 = 0, xb2 = 0, xb3 = 0;            // bit (t & 63) of mask (t >> 6) = x[64 t + lane]
    auto get_bit = [&](int t) -> int {
        const uint64_t m = (t < 64) ? xb0 : (t < 128) ? xb1 : (t < 192) ? xb2 : xb3;   // t is wave-uniform
        return (int)((m >> (t & 63)) & 1ull);
    };
    auto xor_bit = [&](int t, uint64_t v) {
        const uint64_t b = v << (t & 63);
        if (t < 64) xb0 ^= b; else if (t < 128) xb1 ^= b; else if (t < 192) xb2 ^= b; else xb3 ^= b;
    };
    if (init) {
        for (int t = 0; t < slots; ++t) {
            const int i = t * 64 + lane;
            xor_bit(t, (i < n && init[(size_t)r * n + i]) ? 1ull : 0ull);
        }
    } else {
        for (int tg = 0; tg * 4 < slots; ++tg) {
            uint32_t w[4];
            slot_words(w, tg, lane, 0u, gid, 1u, a.seed_lo, a.seed_hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * tg + c;
                if (t < slots) xor_bit(t, (t * 64 + lane < n) ? (uint64_t)(w[c] >> 31) : 0ull);
            }
        }
    }

    int S = 0;
    auto load_slot = [&](int t, uint32_t (&colv)[D], float (&valv)[D]) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            colv[k] = a.ell_col[((size_t)t * D + k) * 64 + lane];
            valv[k] = a.ell_val[((size_t)t * D + k) * 64 + lane];
        }
    };
    // g = lin ; then for j ascending with x_j = 1: g[col] += val over row j ; S = popcount
    auto field_init = [&]() {
        for (int t = 0; t < slots; ++t) g[t * 64 + lane] = a.lin[t * 64 + lane];
        int cnt = 0;
        for (int t = 0; t < slots; ++t) {
            uint64_t m = __ballot(get_bit(t));
            if (m == 0) continue;
            uint32_t colv[D];
            float valv[D];
            load_slot(t, colv, valv);
            cnt += __popcll(m);
            while (m) {
                const int l = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                k2_apply_row<D>(g, scr, colv, valv, lane, l, 1.0f);
            }
        }
        S = cnt;
    };

    unsigned long long accepted = 0;
    int until_resync = a.resync;
    for (int s = 0; s < a.num_sweeps; ++s) {
        bool init_now = (s == 0);
        if (a.resync > 0 && s > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now) field_init();
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(a.temps[a.temps_per_replica ? r : s])));
        for (int tg = 0; tg * 4 < slots; ++tg) {
            uint32_t w[4];
            slot_words(w, tg, lane, (uint32_t)s + a.sweep_offset, gid, 0u, a.seed_lo, a.seed_hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * tg + c;
                if (t >= slots) break;                       // wave-uniform
                uint32_t colv[D];
                float valv[D];
                load_slot(t, colv, valv);
                float thr = neglog_u(w[c]) * T;
                if (t * 64 + lane >= n) thr = -INFINITY;
                int xi = get_bit(t);
                uint64_t todo = ~0ull;
                while (true) {
                    const float fi = g[t * 64 + lane] + a.c_pair * (float)(S - xi);
                    const float dE = xi ? -fi : fi;
                    const uint64_t m = __ballot(dE < thr) & todo;
                    if (m == 0) break;
                    const int l = __ffsll((unsigned long long)m) - 1;
                    todo = (~0ull << l) << 1;
                    const int xl = __builtin_amdgcn_readlane(xi, l);
                    k2_apply_row<D>(g, scr, colv, valv, lane, l, xl ? -1.0f : 1.0f);
                    S += xl ? -1 : 1;
                    if (lane == l) xi ^= 1;
                    xor_bit(t, lane == l ? 1ull : 0ull);
                    ++accepted;
                }
            }
        }
    }

    // ---- epilogue: states out, exact fp64 energy ----
    uint8_t *dst = static_cast<uint8_t *>(a.states) + (size_t)r * n;
    uint32_t *xw = reinterpret_cast<uint32_t *>(g);          // reuse the field array for the bits
    int cnt = 0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        const uint32_t on = (uint32_t)get_bit(t);
        xw[i] = on;
        if (i < n) dst[i] = (uint8_t)on;
        cnt += __popcll(__ballot(on));
    }
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        if (!get_bit(t)) continue;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) {
            const uint32_t cc = a.ell_col[((size_t)t * D + k) * 64 + lane];
            const float vv = a.ell_val[((size_t)t * D + k) * 64 + lane];
            if (xw[cc]) acc += (double)vv;
        }
        e += (double)a.lin[i] + 0.5 * acc;
    }
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + (double)a.c_pair * 0.5 * (double)cnt * (double)(cnt - 1) + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

// ------------------------------------------------------------------------------------------------
// K2 (second form): short decision chain + deferred neighbour updates
// ------------------------------------------------------------------------------------------------
// Same chain as k_anneal_csr_rank1 (bit-identical: oracle 2b), rebuilt around what bounds it: the serial
// accept -> commit -> re-test loop inside a 64-variable slot.  In the first form every commit pushes the
// committing lane's whole adjacency row through LDS (2D stores + a read-modify-write of D fields) before
// the next ballot.  Here the loop only touches what the NEXT decision of this slot can depend on:
//   * the integer s = sum x (every lane: (float)(s - x_i) +- 1.0, exact);
//   * the fields of the committing variable's neighbours INSIDE this slot, which live in registers (gi):
//     the host stores those neighbours first in each row (`meta` = their count), and the first four
//     (col, val) of every lane's row are preloaded, so the update is two v_readlane + one masked add per
//     in-slot neighbour (0.36 per flip on the PBMC-sized SNN graph).
//   Everything else about a flip is scalar bookkeeping: the flipped lane is never tested again in this
//   slot, so its own state is patched after the loop (x ^= flipped), and the sign of a flip comes from the
//   slot's state mask, not from a cross-lane read.  ~20 instructions per accepted flip.
// All other neighbours' fields cannot be read before the slot ends, so their updates are DEFERRED: after
// the loop the flipped lanes' rows are fetched from a row-major copy of the adjacency with the in-slot
// entries blanked (`rows_out`; 64/D rows per wave-load, all loads of a slot in flight together, L2
// resident) and applied with one no-return fp32 add per flip, in flip order (a wave's LDS operations
// execute in order, so every field receives its adds in the oracle's order).
// State bits live in LDS as one 64-bit mask per slot.  Fields (4 B per variable per replica) live in LDS for
// the first `lds_waves` wavefronts of a workgroup and in a global buffer (L2 / Infinity Cache resident,
// no-return global_atomic_add_f32) for the others: 16 replicas per CU stay resident at any n, e.g. 15 + 1 at
// n = 2638, where 16 x 10.5 KB of fields just miss the 160 KB of LDS -- one pass over 4096 replicas
// instead of two.
template <bool GG>
__device__ __forceinline__ void k2_field_add(float *p, float v)
{
    if constexpr (GG) (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)p, v);
    else (void)__builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)p, v, 0, 0, false);
}

// Applies rows {64 t + l : l in mask} (ascending l) to the field array, sign +1 (field initialisation,
// full rows) or by the variable's CURRENT bit in xnew (deferred flips on the blanked rows: x = 1 now
// means the flip was 0 -> 1).  Entries whose value is +-0 (padding, blanked) are skipped.  64/D rows per
// wave-load, up to 16 loads in flight; one field-add instruction per row, in row order.
template <int D, bool GG>
__device__ __forceinline__ void k2_apply_rows_impl(float *g, const uint2 *__restrict__ rows, int t, uint64_t mask,
                                                   uint64_t xnew, bool by_bit, int lane)
{
    constexpr int FPL = 64 / D, NB = 16;
    const int q = lane / D, k = lane % D;
    while (mask) {
        uint2 e[NB];
        int mine[NB];
        // issue: nothing in this loop consumes a loaded value, so all NB loads are in flight together
        int groups = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            mine[b] = -1;
            e[b] = make_uint2(0u, 0u);
            if (mask) {                                      // wave-uniform
                groups = b + 1;
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    if (mask) {
                        const int l = __ffsll((unsigned long long)mask) - 1;
                        mask &= mask - 1;
                        if (q == j) mine[b] = l;
                    }
                }
                if (mine[b] >= 0) e[b] = rows[((size_t)t * 64 + mine[b]) * D + k];
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < groups) {                                // wave-uniform
                const bool neg = by_bit && !((xnew >> (mine[b] & 63)) & 1ull);
                const uint32_t vb = e[b].y ^ (neg ? 0x80000000u : 0u);
                const bool active = mine[b] >= 0 && (vb << 1) != 0u;
#pragma unroll
                for (int j = 0; j < FPL; ++j) {
                    if (q == j && active) k2_field_add<GG>(g + e[b].x, __uint_as_float(vb));
                    asm volatile("" ::: "memory");           // one instruction per row, in row order
                }
            }
        }
    }
}

template <int D>
__device__ __forceinline__ void k2_apply_rows(float *g, const uint2 *__restrict__ rows, int t, uint64_t mask,
                                              uint64_t xnew, bool by_bit, int lane, bool gg)
{
    if (gg) k2_apply_rows_impl<D, true>(g, rows, t, mask, xnew, by_bit, lane);     // wave-uniform
    else k2_apply_rows_impl<D, false>(g, rows, t, mask, xnew, by_bit, lane);
}

#ifdef MI_K2_PROFILE
#define K2_TICK(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tick_; tick_ = now_; } while (0)
#else
#define K2_TICK(var) do { } while (0)
#endif

template <int D>
__global__ void __launch_bounds__(1024) k_anneal_csr_rank1_v2(EllArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waves = (int)(blockDim.x >> 6);
    const int r = blockIdx.x * waves + wave;
    if (r >= a.R) return;                                   // no workgroup-level synchronisation below
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots;
    const bool gg = wave >= a.lds_waves;                     // wave-uniform: this replica's fields are global
    uint64_t *xm = reinterpret_cast<uint64_t *>(lds) + (size_t)wave * slots;   // bit l of xm[t] = x[64 t + l]
    float *g = gg ? a.gbuf + ((size_t)blockIdx.x * (waves - a.lds_waves) + (wave - a.lds_waves)) * ((size_t)slots * 64)
                  : reinterpret_cast<float *>(lds + (size_t)waves * slots * 8 + (size_t)wave * a.g_bytes);
    const uint8_t *init = static_cast<const uint8_t *>(a.init);
    const uint2 *rows = a.rows;

    if (init) {
        for (int t = 0; t < slots; ++t) {
            const int i = t * 64 + lane;
            const uint64_t m = __ballot(i < n && init[(size_t)r * n + i]);
            if (lane == 0) xm[t] = m;
        }
    } else {
        for (int tg = 0; tg * 4 < slots; ++tg) {
            uint32_t w[4];
            slot_words(w, tg, lane, 0u, gid, 1u, a.seed_lo, a.seed_hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * tg + c;
                if (t >= slots) break;
                const uint64_t m = __ballot(t * 64 + lane < n && (w[c] >> 31));
                if (lane == 0) xm[t] = m;
            }
        }
    }

    int S = 0;
    // g = lin ; then for j ascending with x_j = 1: g[col] += val over row j ; S = popcount
    auto field_init = [&]() {
        for (int t = 0; t < slots; ++t)
            if (t * 64 + lane < n) g[t * 64 + lane] = a.lin[t * 64 + lane];
        int cnt = 0;
        for (int t = 0; t < slots; ++t) {
            const uint64_t m = xm[t];
            if (m == 0) continue;
            cnt += __popcll(m);
            k2_apply_rows<D>(g, rows, t, m, 0ull, false, lane, gg);
        }
        S = cnt;
    };

    // per-slot data that does not depend on the chain: fetched one slot ahead
    struct SlotPre { uint32_t meta; uint4 e01, e23; };
    auto prefetch = [&](int t) {
        SlotPre p;
        const int tt = t < slots ? t : slots - 1;
        const size_t i = (size_t)tt * 64 + lane;
        p.meta = a.meta[i];
        p.e01 = *reinterpret_cast<const uint4 *>(rows + i * D);
        p.e23 = *reinterpret_cast<const uint4 *>(rows + i * D + 2);
        return p;
    };

    unsigned long long accepted = 0;
#ifdef MI_K2_PROFILE
    unsigned long long tick_ = __builtin_amdgcn_s_memtime(), t_pre = 0, t_loop = 0, t_wait = 0, t_apply = 0, t_init = 0;
#endif
    int until_resync = a.resync;
    for (int s = 0; s < a.num_sweeps; ++s) {
        bool init_now = (s == 0);
        if (a.resync > 0 && s > 0 && --until_resync == 0) { init_now = true; until_resync = a.resync; }
        if (init_now) field_init();
        K2_TICK(t_init);
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(a.temps[a.temps_per_replica ? r : s])));
        SlotPre nxt = prefetch(0);
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int t = 0; t < slots; ++t) {                    // ONE copy of the slot body (instruction cache)
            {
                if ((t & 3) == 0)
                    slot_words(w, t >> 2, lane, (uint32_t)s + a.sweep_offset, gid, 0u, a.seed_lo, a.seed_hi);
                const int c = t & 3;
                const uint32_t wc = c == 0 ? w[0] : (c == 1 ? w[1] : (c == 2 ? w[2] : w[3]));
                const int i = t * 64 + lane;
                const SlotPre cur = nxt;
                nxt = prefetch(t + 1);
                const uint32_t metav = cur.meta;
                const uint4 e01 = cur.e01, e23 = cur.e23;
                float gi = 0.0f;
                if (i < n) gi = gg ? __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g[i];
                float thr = neglog_u(wc) * T;
                if (i >= n) thr = -INFINITY;
                const uint64_t xm_t = xm[t];
                const uint32_t xi = (uint32_t)((xm_t >> lane) & 1ull);
                const uint32_t sgnbit = xi << 31;            // dE = x ? -f : f
                float Sf = (float)(S - (int)xi);
                const uint64_t has_in = __ballot((metav & 0xffu) != 0u);
                uint64_t todo = ~0ull, flipped = 0ull;
                // This slot's prefetched registers are waited for HERE: left to hipcc, the wait lands at their
                // first use inside the loop, as a counted s_waitcnt that every iteration executes.
                asm volatile("" ::"v"(metav), "v"(e01.x), "v"(e01.y), "v"(e01.z), "v"(e01.w), "v"(e23.x),
                             "v"(e23.y), "v"(e23.z), "v"(e23.w));
                K2_TICK(t_pre);
                while (true) {
                    const float fi = gi + a.c_pair * Sf;
                    const float dE = __uint_as_float(__float_as_uint(fi) ^ sgnbit);
                    const uint64_t m = __ballot(dE < thr) & todo;
                    if (m == 0) break;
                    const int l = __ffsll((unsigned long long)m) - 1;
                    todo = (~0ull << l) << 1;
                    flipped |= 1ull << l;
                    const bool xl = (xm_t >> l) & 1ull;      // the lane's bit BEFORE its (only) flip in this slot
                    const float sgn = xl ? -1.0f : 1.0f;
                    S += xl ? -1 : 1;
                    Sf += sgn;
                    if ((has_in >> l) & 1ull) {              // wave-uniform: l has neighbours inside this slot
                        const int nin = (int)(__builtin_amdgcn_readlane((int)metav, l) & 0xff);
                        auto hit = [&](uint32_t cc, uint32_t vv) {
                            if (lane == (int)(cc & 63u)) gi = gi + sgn * __uint_as_float(vv);
                        };
                        hit(__builtin_amdgcn_readlane((int)e01.x, l), __builtin_amdgcn_readlane((int)e01.y, l));
                        if (nin > 1) hit(__builtin_amdgcn_readlane((int)e01.z, l), __builtin_amdgcn_readlane((int)e01.w, l));
                        if (nin > 2) hit(__builtin_amdgcn_readlane((int)e23.x, l), __builtin_amdgcn_readlane((int)e23.y, l));
                        if (nin > 3) hit(__builtin_amdgcn_readlane((int)e23.z, l), __builtin_amdgcn_readlane((int)e23.w, l));
                        for (int k = 4; k < nin; ++k) {
                            const uint2 e = rows[((size_t)t * 64 + l) * D + k];
                            hit(e.x, e.y);
                        }
                    }
                }
                K2_TICK(t_loop);
                if (flipped) {                               // wave-uniform
                    const uint64_t xnew = xm_t ^ flipped;
                    accepted += (unsigned long long)__popcll(flipped);
                    if (lane == 0) xm[t] = xnew;
                    if (i < n) g[i] = gi;
                    k2_apply_rows<D>(g, a.rows_out, t, flipped, xnew, true, lane, gg);
                }
                K2_TICK(t_apply);
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
    int cnt = 0;
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        const uint64_t m = xm[t];
        const int on = (int)((m >> lane) & 1ull);
        if (i < n) dst[i] = (uint8_t)on;
        cnt += __popcll(m);
        if (!on) continue;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) {
            const uint32_t cc = a.ell_col[((size_t)t * D + k) * 64 + lane];
            const float vv = a.ell_val[((size_t)t * D + k) * 64 + lane];
            if ((xm[cc >> 6] >> (cc & 63u)) & 1ull) acc += (double)vv;
        }
        e += (double)a.lin[i] + 0.5 * acc;
    }
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + (double)a.c_pair * 0.5 * (double)cnt * (double)(cnt - 1) + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

// ------------------------------------------------------------------------------------------------
// K3
// ------------------------------------------------------------------------------------------------
// LDS per wave: labels, one byte per variable (K <= 64).  Lane q of the wave keeps cnt[q] (cluster sizes).
// Proposal of variable i: a = l_i, b = (a + 1 + word(i,s,g,2) mod (K-1)) mod K;
//   dE = [h_b + c cnt_b] - [h_a + c (cnt_a - 1)],  h_q = sum of S_ij over neighbours j with l_j = q
// (h sums taken in stored neighbour order, fp32 -- oracle 2c).  After a commit every later lane of the
// slot re-evaluates from its register-resident adjacency row.
template <int D>
__global__ void __launch_bounds__(256) k_anneal_potts(EllArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (r >= a.R) return;
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots, K = a.K;
    uint8_t *lab = reinterpret_cast<uint8_t *>(lds) + (size_t)wave * slots * 64;
    const uint16_t *init = static_cast<const uint16_t *>(a.init);

    for (int tg = 0; tg * 4 < slots; ++tg) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (!init) slot_words(w, tg, lane, 0u, gid, 1u, a.seed_lo, a.seed_hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = 4 * tg + c;
            if (t >= slots) break;
            const int i = t * 64 + lane;
            uint32_t v = 0;
            if (i < n) v = init ? (uint32_t)init[(size_t)r * n + i] : (w[c] % (uint32_t)K);
            lab[i] = (uint8_t)v;
        }
    }
    int cntv = 0;                                            // lane q: number of variables with label q
    for (int q = 0; q < K; ++q) {
        int c = 0;
        for (int t = 0; t < slots; ++t)
            c += __popcll(__ballot(t * 64 + lane < n && lab[t * 64 + lane] == q));
        if (lane == q) cntv = c;
    }

    unsigned long long accepted = 0;
    for (int s = 0; s < a.num_sweeps && K > 1; ++s) {
        const float T = __int_as_float(__builtin_amdgcn_readfirstlane(
            __float_as_int(a.temps[a.temps_per_replica ? r : s])));
        for (int tg = 0; tg * 4 < slots; ++tg) {
            uint32_t w0[4], w2[4];
            slot_words(w0, tg, lane, (uint32_t)s + a.sweep_offset, gid, 0u, a.seed_lo, a.seed_hi);
            slot_words(w2, tg, lane, (uint32_t)s + a.sweep_offset, gid, 2u, a.seed_lo, a.seed_hi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int t = 4 * tg + c;
                if (t >= slots) break;
                const int i = t * 64 + lane;
                uint32_t colv[D];
                float valv[D];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    colv[k] = a.ell_col[((size_t)t * D + k) * 64 + lane];
                    valv[k] = a.ell_val[((size_t)t * D + k) * 64 + lane];
                }
                float thr = neglog_u(w0[c]) * T;
                if (i >= n) thr = -INFINITY;
                int la = lab[i];
                const int lb = (la + 1 + (int)(w2[c] % (uint32_t)(K - 1))) % K;
                uint64_t todo = ~0ull;
                // h_a / h_b are summed from scratch (stored neighbour order) at the slot start and again
                // only for lanes that have the committed variable among their neighbours -- for every
                // other lane the cached sums ARE what a fresh evaluation would give; cluster sizes change
                // for everybody and enter through cnt.
                float ha = 0.0f, hb = 0.0f;
                auto sum_h = [&]() {
                    ha = 0.0f;
                    hb = 0.0f;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const int lj = lab[colv[k]];
                        ha = ha + ((lj == la) ? valv[k] : 0.0f);
                        hb = hb + ((lj == lb) ? valv[k] : 0.0f);
                    }
                };
                sum_h();
                while (true) {
                    const int ca = __shfl(cntv, la, 64), cb = __shfl(cntv, lb, 64);
                    const float ea = ha + a.c_pair * (float)(ca - 1);
                    const float eb = hb + a.c_pair * (float)cb;
                    const float dE = eb - ea;
                    const uint64_t m = __ballot(dE < thr) & todo;
                    if (m == 0) break;
                    const int l = __ffsll((unsigned long long)m) - 1;
                    todo = (~0ull << l) << 1;
                    const int a_s = __builtin_amdgcn_readlane(la, l);
                    const int b_s = __builtin_amdgcn_readlane(lb, l);
                    if (lane == l) { lab[i] = (uint8_t)lb; la = lb; }
                    if (lane == a_s) cntv -= 1;
                    if (lane == b_s) cntv += 1;
                    const uint32_t moved = (uint32_t)(t * 64 + l);
                    bool touched = false;
#pragma unroll
                    for (int k = 0; k < D; ++k) touched |= (colv[k] == moved);
                    if (touched && lane > l) sum_h();          // padding (self) never equals `moved` for lane > l
                    ++accepted;
                }
            }
        }
    }

    // ---- epilogue: labels out, exact fp64 energy ----
    uint16_t *dst = static_cast<uint16_t *>(a.states) + (size_t)r * n;
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        if (i >= n) continue;
        const int li = lab[i];
        dst[i] = (uint16_t)li;
        for (int k = 0; k < D; ++k) {
            const uint32_t cc = a.ell_col[((size_t)t * D + k) * 64 + lane];
            const float vv = a.ell_val[((size_t)t * D + k) * 64 + lane];
            if ((int)cc > i && lab[cc] == li) e += (double)vv;
        }
    }
    if (lane < K) e += (double)a.c_pair * 0.5 * (double)cntv * (double)(cntv - 1);
    e = wave_sum_f64(e);
    if (lane == 0) {
        a.energy[r] = e + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

template <typename KernelT>
int launch_sparse(KernelT kernel, const EllArgs &a, size_t lds_per_wave, hipStream_t st)
{
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

}  // namespace

// K2 second form: 16 replicas per CU at any size -- as many as fit keep their fields in LDS, the others in
// the global buffer (mi_k2_plan, shared with the host code that sizes that buffer).
void mi_k2_plan(int n, int slots, int R, int cus, int waves_override, int lds_waves_override, K2Plan *out)
{
    const size_t lds_max = 160 * 1024;
    int waves = waves_override > 0 ? waves_override : (R + cus - 1) / cus;
    if (waves < 1) waves = 1;
    if (waves > 16) waves = 16;
    while (waves > 1 && (size_t)waves * slots * 8 > lds_max) --waves;
    const size_t g_bytes = (((size_t)n * 4 + 15) / 16) * 16;
    const size_t room = lds_max - (size_t)waves * slots * 8;
    int lds_waves = (int)(room / g_bytes);
    if (lds_waves > waves) lds_waves = waves;
    if (lds_waves_override >= 0 && lds_waves_override < lds_waves) lds_waves = lds_waves_override;
    out->waves = waves;
    out->lds_waves = lds_waves;
    out->g_bytes = (int)g_bytes;
    out->lds_bytes = (size_t)waves * slots * 8 + (size_t)lds_waves * g_bytes;
    out->blocks = (R + waves - 1) / waves;
    out->gbuf_floats = (size_t)out->blocks * (waves - lds_waves) * ((size_t)slots * 64);
}

namespace {

template <int D>
int launch_csr_rank1_v2(EllArgs a, hipStream_t st)
{
    K2Plan pl;
    mi_k2_plan(a.n, a.slots, a.R, a.cus > 0 ? a.cus : 256, a.waves_override, a.lds_waves_override, &pl);
    if ((size_t)a.slots * 8 > 160 * 1024) return fail(MI_EUNSUPPORTED, "csr_rank1: n = %d exceeds the state-mask LDS budget", a.n);
    if (pl.gbuf_floats > 0 && !a.gbuf) return fail(MI_EHIP, "csr_rank1: global field buffer missing");
    a.lds_waves = pl.lds_waves;
    a.g_bytes = pl.g_bytes;
    auto kernel = k_anneal_csr_rank1_v2<D>;
    if (pl.lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes));
    hipLaunchKernelGGL(kernel, dim3(pl.blocks), dim3(pl.waves * 64), pl.lds_bytes, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

int mi_launch_csr_rank1(const EllArgs &a, hipStream_t st)
{
    if ((a.variant & 15) != 1) {
        if (a.D == 16) return launch_csr_rank1_v2<16>(a, st);
        if (a.D == 32) return launch_csr_rank1_v2<32>(a, st);
        if (a.D == 64) return launch_csr_rank1_v2<64>(a, st);
        return fail(MI_EUNSUPPORTED, "slot-ELL width %d not built", a.D);
    }
    if (a.n > 16384) return fail(MI_EUNSUPPORTED, "csr_rank1 first form supports n <= 16384 (got %d)", a.n);
    const size_t per_wave = (size_t)a.slots * 64 * 4 + 2 * (size_t)a.D * 4;
    if (a.D == 16) return launch_sparse(k_anneal_csr_rank1<16>, a, per_wave, st);
    if (a.D == 32) return launch_sparse(k_anneal_csr_rank1<32>, a, per_wave, st);
    if (a.D == 64) return launch_sparse(k_anneal_csr_rank1<64>, a, per_wave, st);
    return fail(MI_EUNSUPPORTED, "slot-ELL width %d not built", a.D);
}

int mi_launch_potts(const EllArgs &a, hipStream_t st)
{
    const size_t per_wave = (size_t)a.slots * 64;
    if (a.D == 16) return launch_sparse(k_anneal_potts<16>, a, per_wave, st);
    if (a.D == 32) return launch_sparse(k_anneal_potts<32>, a, per_wave, st);
    if (a.D == 64) return launch_sparse(k_anneal_potts<64>, a, per_wave, st);
    return fail(MI_EUNSUPPORTED, "slot-ELL width %d not built", a.D);
}

}  // namespace mi_sa_impl
