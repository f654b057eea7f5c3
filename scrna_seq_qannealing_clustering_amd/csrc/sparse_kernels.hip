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

int mi_launch_csr_rank1(const EllArgs &a, hipStream_t st)
{
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
