// sparse_split_kernels.hip -- K2s: the CSR + uniform-pair anneal for FEW replicas (gfx950 only).
//
// Why.  The reference asks for num_reads = 500 (BQM_clustering.py:52) on graphs that shrink under its recursive
// bisection (:113-203).  500 replicas are 500 (K2) or 250 (K2p) wavefronts on a chip with 1024 SIMDs: every wavefront
// is alone on its SIMD and the run takes ONE wavefront's latency through slots x sweeps dependent steps of ~970 cycles
// (19 ms for n = 2638 x 1000 sweeps whether a wavefront carries one replica or two) while three quarters of the chip idle.
// Here a replica is swept by a WORKGROUP of NW wavefronts: the variables are laid out in BLOCKS of 64 NW mutually
// non-adjacent seats (mi_sa_plan_slot_layout with slot = 64 NW; holes allowed), wave w owns seats 64 w .. 64 w + 63 of
// every block, and a sweep is blocks = slots / NW dependent steps instead of slots.
//
// Same chain, bit for bit (oracle/sa_oracle.c 2b on the same padded model): inside a block the decisions interact only
// through s = sum x, so the accept mask of the SEQUENTIAL sweep over the block's 64 NW seats is the fixed point of
// "evaluate every seat under a guessed mask, rebuild the mask" (k_anneal_csr_rank1) -- here
// over NW wavefronts: wave w sees s = S + off_w + d_lane, off_w = the net change the waves below it make.  Every wave
// solves its own 64 seats for a given off_w (rounds inside the wave, no rendezvous), publishes its net change, and after
// ONE barrier every wave knows all of them; when the offsets the waves used reproduce themselves the block is done
// (wave k is final after k + 1 passes; in the cold two thirds of a schedule nothing flips and the first pass is the last).
// The states are written speculatively BEFORE that barrier (and patched if a later pass changes the mask), so the
// barrier that ends the exchange is also the one that orders this block's writes before the next block's gathers:
// one barrier per block in the common case.
//
// Random words: one Philox block serves four 64-seat slots; every wave computes the blocks of ITS slots (the same words
// K2 draws: counter (group, lane), sweep, replica), a few rounds per step under the LDS gathers of the running group
// instead of ten at the head of every fourth slot.  The state is one 32-bit LDS cell per seat (ds_read_b32 gathers cost
// a lone wavefront two thirds of what ds_read_u16 ones do); NW = 1 is the same step without the exchange: the
// one-wavefront-per-replica kernel for few replicas on models too small for wider blocks.
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

typedef _Float16 half_t;

#ifdef MI_K2_PROFILE     /* development build (`make prof`): cycles per phase of a step, wave 0 of every workgroup */
#define K2S_TICK(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tick_; tick_ = now_; } while (0)
#define K2S_TICK0(var) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tick_; tick_ = now_; } while (0)   /* (no wait: LDS reads stay in flight) */
#else
#define K2S_TICK(var) do { } while (0)
#define K2S_TICK0(var) do { } while (0)
#endif

constexpr int kCommBytes = 8 * 16;               // eight exchange slots of four ints (the waves' net changes)
constexpr int kRingBytes = 4 * 64 * 16;          // NW > 1: four groups of random words in flight, [4][64 lanes][4 words]

// One Philox4x32-10 block computed a few rounds per step: the ten rounds of the NEXT group of four slots ride in the
// shadow of this group's LDS gathers instead of standing, all ten, at the head of every fourth slot.
struct PhiloxPipe {
    uint32_t c0, c1, c2, c3;     // counter words after `done` rounds
    uint32_t k0, k1;             // round keys of the next round
    int done;
    __device__ __forceinline__ void start(uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t key0, uint32_t key1)
    {
        c0 = i0; c1 = i1; c2 = i2; c3 = i3; k0 = key0; k1 = key1; done = 0;
    }
    __device__ __forceinline__ void round()
    {
        const uint64_t p0 = (uint64_t)PH_M0 * c0, p1 = (uint64_t)PH_M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += PH_W0; k1 += PH_W1;
        ++done;
    }
};

template <int D, int NW>
__global__ void __launch_bounds__(64 * NW, 1) k_anneal_csr_rank1_split(EllArgs a)
{
    static_assert(NW == 1 || NW == 2 || NW == 4, "one, two or four wavefronts per replica");
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [0, 4 * 64 * slots): one 32-bit cell per seat, low half = x (0.0 / 1.0)
    const int lane = threadIdx.x & 63;
    const int w = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = blockIdx.x;                                       // grid = R workgroups
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots, blocks = slots / NW;        // slots is a multiple of NW (launcher)
    const int comm_at = slots * 256, ring_at = comm_at + kCommBytes;
    const uint8_t *init = static_cast<const uint8_t *>(a.init);
    int *comm = reinterpret_cast<int *>(lds + comm_at);
    const int gps = (slots + 3) / 4;                                // Philox groups (of four slots) per sweep

    // ---- initial state: wave w fills its own slots; S = sum x over the whole replica ----
    int S = 0;
    for (int b = 0; b < blocks; ++b) {
        const int t = b * NW + w, i = t * 64 + lane;
        bool x;
        const bool real = i < n && a.lin[i] < INFINITY;            // (+inf linear term = hole of a padded layout: stays 0)
        if (init) {
            x = real && init[(size_t)r * n + i] != 0;
        } else {
            uint32_t iw[4];
            philox4x32_10((uint32_t)((t >> 2) * 64 + lane), 0u, gid, 1u, a.seed_lo, a.seed_hi, iw);
            const int c = t & 3;
            x = real && ((c == 0 ? iw[0] : (c == 1 ? iw[1] : (c == 2 ? iw[2] : iw[3]))) >> 31);
        }
        reinterpret_cast<uint32_t *>(lds)[i] = x ? 0x3c00u : 0u;
        S += __popcll(__ballot(x));
    }
    if constexpr (NW > 1) {
        if (lane == 0) comm[w] = S;
        __syncthreads();
        S = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) S += __builtin_amdgcn_readfirstlane(comm[k]);
        __syncthreads();                                            // (comm[0..3] is exchange slot 0: read before reuse)
    }

    constexpr int G = D / 4;
    const __amdgpu_buffer_rsrc_t rs_adj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.adj4), 0, slots * G * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.lin), 0, slots * 256, 0x00020000);
    struct SlotAdj { u32x4 col[G]; u32x4 val[G]; uint32_t lin; };
    auto fetch_adj = [&](int t) {                                   // the pair kernel's packing: neighbour word = 4 * index
        SlotAdj p;
        const int tt = t < slots ? t : slots - 1;
        const int soff = tt * (G * 2048);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int so = soff + (g / 2) * 4096, io = (g & 1) * 2048;
            p.col[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16 + io, so, 0);
            p.val[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16 + io + 1024, so, 0);
        }
        p.lin = __builtin_amdgcn_raw_buffer_load_b32(rs_lin, lane * 4, tt * 256, 0);
        return p;
    };

#ifdef MI_K2_PROFILE
    unsigned long long tick_ = __builtin_amdgcn_s_memtime(), t_top = 0, t_sum = 0, t_solve = 0, t_xchg = 0, t_more = 0, t_issue = 0, t_arith = 0;
#endif
    unsigned long long accepted = 0;
    uint32_t xc = 0;                                                // exchanges so far (slot xc & 7 of comm)
    float T = 1.0f;
    const float cp = a.c_pair;
    uint32_t sw = a.sweep_offset;
    int s = 0;

    // Random words.  One Philox block serves four slots (one group).
    //   NW = 1: `cw` = the words of the group the wave is in, `nx` = the next group's block in the making, 3 + 3 + 2 + 2
    //   rounds over the group's four steps (whatever is missing when the group changes -- a short last group -- is
    //   finished there).
    //   NW > 1: the groups of the run, numbered q = sweep * groups_per_sweep + g in the order they are used, are computed
    //   by the waves in turn (q % NW), each wave spreading the ten rounds of its next group over the four steps before
    //   the one that needs it (slots is a multiple of four: a group per 4 / NW steps, every wave one group per four
    //   steps), and shared through a ring of four LDS entries: written during the step BEFORE the group's first one (the
    //   barrier that ends every step publishes it), read as one word per lane at the top of a step.
    uint32_t cw[4] = {0u, 0u, 0u, 0u};
    PhiloxPipe nx;
    int cur_g = -1;                                                 // NW = 1: group of `cw` within the sweep
    int next_q = w;                                                 // NW > 1: the group this wave produces next
    auto start_group = [&](int q) {                                 // counter of group q of the run
        const int qs = q / gps, qg = q - qs * gps;
        nx.start((uint32_t)(qg * 64 + lane), a.sweep_offset + (uint32_t)qs, gid, 0u, a.seed_lo, a.seed_hi);
    };
    auto publish_group = [&](int q) {
        const u32x4 v = {nx.c0, nx.c1, nx.c2, nx.c3};
        asm volatile("ds_write_b128 %0, %1" :: "v"(ring_at + (q & 3) * 1024 + lane * 16), "v"(v) : "memory");
    };
    (void)cw; (void)cur_g; (void)next_q;
    if constexpr (NW == 1) {
        nx.start((uint32_t)lane, sw, gid, 0u, a.seed_lo, a.seed_hi);       // the first slot's group: 0
    } else {
        start_group(next_q);                                        // this wave's first group, whole, before the first step
#pragma unroll
        for (int k = 0; k < 10; ++k) nx.round();
        publish_group(next_q);
        next_q += NW;
        start_group(next_q);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int ustep = 0;                                                  // steps done so far (all sweeps)

    // one block: this wave's slot t = b NW + w with its adjacency `cur`
    auto step = [&](int b, const SlotAdj &cur) {
        const int t = b * NW + w, i = t * 64 + lane;
        const int c = t & 3;
        if constexpr (NW == 1) {
            if ((t >> 2) != cur_g) {                                // wave-uniform: this slot opens a new group
                while (nx.done < 10) nx.round();
                cw[0] = nx.c0; cw[1] = nx.c1; cw[2] = nx.c2; cw[3] = nx.c3;
                cur_g = t >> 2;
                const int tn = ((t >> 2) + 1) * 4;                  // first slot of the next group, or of the next sweep
                const bool wrap = tn >= slots;
                nx.start((uint32_t)((wrap ? 0 : (tn >> 2)) * 64 + lane), wrap ? sw + 1u : sw, gid, 0u, a.seed_lo, a.seed_hi);
            }
        }
        // (1) LDS reads, issued together: (NW > 1: this slot's random word,) the lane's own cell, then the 16 neighbour
        // cells; they return in issue order
        uint32_t own, word[16], rword = 0u;
        if constexpr (NW > 1)
            asm volatile("ds_read_b32 %0, %1" : "=v"(rword) : "v"(ring_at + ((s * gps + (t >> 2)) & 3) * 1024 + lane * 16 + c * 4) : "memory");
        asm volatile("ds_read_b32 %0, %1" : "=v"(own) : "v"(i * 4) : "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k)
            asm volatile("ds_read_b32 %0, %1" : "=v"(word[k]) : "v"(cur.col[k / 4][k & 3]));
        K2S_TICK0(t_issue);
        // (2) under the gathers: the threshold of this slot and a share of the next group's random words
        // (the empty asm statements keep this arithmetic BEHIND the reads' issue: hipcc would hoist it above them)
        float thr;
        if constexpr (NW == 1) {
            rword = c == 0 ? cw[0] : (c == 1 ? cw[1] : (c == 2 ? cw[2] : cw[3]));
            asm volatile("" : "+v"(rword), "+v"(nx.c0), "+v"(nx.c1), "+v"(nx.c2), "+v"(nx.c3));
            nx.round(); nx.round();
            if (c < 2) nx.round();                                  // 3 + 3 + 2 + 2 over the group's four steps
            thr = neglog_u(rword) * T;
        } else {
            // this wave's share of the random words: its next group over the four steps that end one step before the
            // group's first (group q opens step q * 4 / NW: a group every 4 / NW steps, NW producers)
            asm volatile("" : "+v"(nx.c0), "+v"(nx.c1), "+v"(nx.c2), "+v"(nx.c3));
            const int due = next_q * (4 / NW) - 1;                  // the step during which group next_q must be published
            const int left = due - ustep;                           // 3, 2, 1, 0 inside the group's four steps
            if (left <= 3) {                                        // wave-uniform
                nx.round(); nx.round();
                if (left >= 2) nx.round();                          // 3 + 3 + 2 + 2
                if (left == 0) {
                    publish_group(next_q);
                    next_q += NW;
                    start_group(next_q);
                }
            }
            // the random word was issued first: at most the 15 youngest reads may still be in flight behind it
            asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(rword) :: "memory");
            thr = neglog_u(rword) * T;
        }
        K2S_TICK0(t_arith);
        K2S_TICK(t_top);
        // (3) the field sum, four neighbours at a time as they arrive (counted waits)
        float gi = __uint_as_float(cur.lin);                        // (lanes past n and holes carry lin = +inf: never accepted)
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            if (g0 > 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    asm volatile("ds_read_b32 %0, %1" : "=v"(word[k]) : "v"(cur.col[g0 + k / 4][k & 3]));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(word[0]), "+v"(word[1]), "+v"(word[2]), "+v"(word[3]), "+v"(own), "+v"(thr) :: "memory");
                if (q == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(word[4]), "+v"(word[5]), "+v"(word[6]), "+v"(word[7]) :: "memory");
                if (q == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(word[8]), "+v"(word[9]), "+v"(word[10]), "+v"(word[11]) :: "memory");
                if (q == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(word[12]), "+v"(word[13]), "+v"(word[14]), "+v"(word[15]) :: "memory");
#pragma unroll
                for (int k = 4 * q; k < 4 * q + 4; ++k) {
                    const half_t hx = __builtin_bit_cast(half_t, (uint16_t)word[k]);
                    gi = __builtin_fmaf(__uint_as_float(cur.val[g0 + k / 4][k & 3]), (float)hx, gi);   // fma(val, x, g)
                }
            }
        }
        K2S_TICK(t_sum);
        // (4) this wave's 64 seats for a given sum at the block's first seat of this wave (k_anneal_csr_rank1's rounds)
        const uint64_t X = __ballot(own != 0u);
        const uint32_t xi = own >> 13;                              // 0x3c00 -> 1
        const uint32_t sgnbit = xi << 31;                           // dE = x ? -f : f
        const float gs = __uint_as_float(__float_as_uint(gi) ^ sgnbit);
        const float cs = __uint_as_float(__float_as_uint(cp) ^ sgnbit);
        const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(X >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)X, 0u));
        auto solve = [&](int s_in, uint64_t A, bool fresh) -> uint64_t {
            const int base = s_in - (int)xi - below;
            if (fresh) {                                            // first guess: nobody below this lane moves
                A = __ballot(gs + cs * (float)(s_in - (int)xi) < thr);
                if (A == 0ull) return A;                            // (an empty mask reproduces itself)
            }
            for (int round = 0; round < 66; ++round) {
                const uint64_t Bm = A ^ X;
                const int s_i = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(Bm >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)Bm, (uint32_t)base));
                const uint64_t A2 = __ballot(gs + cs * (float)s_i < thr);
                if (A2 == A) break;
                A = A2;
            }
            return A;
        };
        uint64_t A = solve(S, 0ull, true), written = 0ull;
        K2S_TICK(t_solve);
        if constexpr (NW == 1) {
            if (A != 0ull) {                                        // wave-uniform
                if ((A >> lane) & 1ull) asm volatile("ds_write_b32 %0, %1" :: "v"(i * 4), "v"(own ^ 0x3c00u) : "memory");
                S += __popcll(A & ~X) - __popcll(A & X);
                accepted += (unsigned long long)__popcll(A);
            }
            (void)written; (void)xc; (void)comm;
        } else {
            int used[NW];                                           // the offsets the published changes were computed with
#pragma unroll
            for (int k = 0; k < NW; ++k) used[k] = 0;
            int total = 0;
            for (int pass = 0; pass < NW + 2; ++pass) {
                // states as this pass has them (toggle what differs from what is in LDS), then this wave's net change
                const uint64_t fix = A ^ written;
                if (fix != 0ull) {                                  // wave-uniform
                    if ((fix >> lane) & 1ull) {
                        own ^= 0x3c00u;
                        asm volatile("ds_write_b32 %0, %1" :: "v"(i * 4), "v"(own) : "memory");
                    }
                    written = A;
                }
                const int dw = __popcll(A & ~X) - __popcll(A & X);
                if (lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(comm_at + (int)(xc & 7u) * 16 + w * 4), "v"(dw) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                u32x4 dv;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(dv) : "v"(comm_at + (int)(xc & 7u) * 16) : "memory");
                ++xc;
                int d[4];
                d[0] = __builtin_amdgcn_readfirstlane((int)dv[0]);
                d[1] = __builtin_amdgcn_readfirstlane((int)dv[1]);
                d[2] = __builtin_amdgcn_readfirstlane((int)dv[2]);
                d[3] = __builtin_amdgcn_readfirstlane((int)dv[3]);
                // offsets these changes imply; done when every wave computed its change with exactly its offset
                int off[NW], run = 0;
                bool same = true;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    off[k] = run;
                    same = same && off[k] == used[k];
                    run += d[k];
                }
                total = run;
                if (pass == 0) K2S_TICK(t_xchg);
                if (same) break;
#pragma unroll
                for (int k = 0; k < NW; ++k) used[k] = off[k];
                int my = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) my = k == w ? off[k] : my;
                A = solve(S + my, A, false);
            }
            S += total;
            accepted += (unsigned long long)__popcll(A);
        }
        ++ustep;
        K2S_TICK(t_more);
    };

    for (s = 0; s < a.num_sweeps; ++s) {
        T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s])));
        sw = (uint32_t)s + a.sweep_offset;
        cur_g = -1;                                                 // (the first slot of a sweep always opens a group)
        SlotAdj P = fetch_adj(w), Q;
#pragma unroll 1
        for (int b = 0; b < blocks; b += 2) {
            Q = fetch_adj((b + 1) * NW + w);
            step(b, P);
            if (b + 1 < blocks) {                                   // wave-uniform
                P = fetch_adj((b + 2) * NW + w);
                step(b + 1, Q);
            }
        }
    }
    __syncthreads();
#ifdef MI_K2_PROFILE
    if (threadIdx.x == 0) {
        atomicAdd(&a.stats[8], t_top); atomicAdd(&a.stats[9], t_sum); atomicAdd(&a.stats[10], t_solve);
        atomicAdd(&a.stats[11], t_xchg); atomicAdd(&a.stats[12], t_more); atomicAdd(&a.stats[13], t_issue); atomicAdd(&a.stats[6], t_arith);
    }
#endif

    // ---- epilogue: states out, exact fp64 energy (the sums of k_anneal_csr_rank1, wave w over its own slots) ----
    uint8_t *dst = static_cast<uint8_t *>(a.states) + (size_t)r * n;
    const uint32_t *cell = reinterpret_cast<const uint32_t *>(lds);
    int cnt = 0;
    double e = 0.0;
    for (int b = 0; b < blocks; ++b) {
        const int t = b * NW + w, i = t * 64 + lane;
        const bool on = cell[i] != 0u;
        if (i < n) dst[i] = (uint8_t)on;
        cnt += __popcll(__ballot(on));
        if (!on) continue;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) {
            const size_t at = ((size_t)t * D + k) * 64 + lane;
            const uint32_t cc = a.ell_col[at];
            const double vv = a.ell_val64 ? a.ell_val64[at] : (double)a.ell_val[at];
            if (cell[cc] != 0u) acc += vv;
        }
        e += (a.lin64 ? a.lin64[i] : (double)a.lin[i]) + 0.5 * acc;
    }
    e = wave_sum_f64(e);
    if constexpr (NW > 1) {
        __syncthreads();                                            // (every wave is done reading the cells)
        double *esum = reinterpret_cast<double *>(lds);             // (the state is free now)
        int *csum = reinterpret_cast<int *>(lds + 64);
        if (lane == 0) { esum[w] = e; csum[w] = cnt; }
        __syncthreads();
        e = 0.0; cnt = 0;
        for (int k = 0; k < NW; ++k) { e += esum[k]; cnt += csum[k]; }
    }
    if (threadIdx.x == 0) {
        const double cp64 = a.ell_val64 ? a.c_pair64 : (double)a.c_pair;
        a.energy[r] = e + cp64 * 0.5 * (double)cnt * (double)(cnt - 1) + a.offset;
    }
    if (lane == 0) atomicAdd(&a.stats[1], accepted);
}

// ------------------------------------------------------------------------------------------------
// K2w: ONE wavefront per replica, SPB slots per step
// ------------------------------------------------------------------------------------------------
// The other way to use an edge-free block of 64 SPB seats (SPB = 2, 4) when replicas are few: one wavefront sweeps the
// whole block in one step -- the LDS reads of all its slots issued together, their field sums as SPB independent fma
// chains, their thresholds side by side (two at a time as packed arithmetic), and only the accept masks in sequence
// (slot j + 1 sees the sum slot j leaves).  A wavefront that is alone on its SIMD spends a slot of k_anneal_csr_rank1
// mostly WAITING (115 instructions in 950 cycles: LDS round trip, the 16-deep fma chain, vector -> scalar -> vector
// hops); two or four slots' worth of independent work per step fill those gaps, with no exchange between wavefronts.
// Same chain as every other kernel of the family on the same padded model, bit for bit.
// TW (round 3): a second wavefront in the workgroup -- on another SIMD, and with few replicas the chip has SIMDs to spare --
// computes the random words and the thresholds -ln(u) * T one group of four slots ahead and hands them over through a
// two-deep ring in LDS, one s_barrier per group (k_anneal_csr_rank1_pair has the same arrangement): a third of the
// instructions of a step leave the wavefront whose issue rate bounds the run.  SPB = 1 (TW only): the plain
// one-slot-per-step sweep, for the 64-seat layouts of small or strongly clustered models.
// WGT: the model carries pair-term weights (mi_sa_problem_set_pair_weights): the sum is sum_j w_j x_j and one slot is swept
// serially (weighted_slot_sweep) -- a template switch, so that the other models' code is what it was.
template <int D, int SPB, bool TW, bool WGT = false>
__global__ void __launch_bounds__(TW ? 128 : 64, 1) k_anneal_csr_rank1_wide(EllArgs a)
{
    static_assert(SPB == 2 || SPB == 4 || (SPB == 1 && TW), "one (with a threshold wavefront), two or four slots per step");
    extern __shared__ __attribute__((aligned(16))) char lds[];      // one 32-bit cell per seat, low half = x (0.0 / 1.0)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x;
    const uint32_t gid = a.replica_offset + (uint32_t)r;
    const int n = a.n, slots = a.slots, blocks = slots / SPB;       // slots is a multiple of SPB (launcher)
    const uint8_t *init = static_cast<const uint8_t *>(a.init);
    // TW: the ring behind the cells, two groups of [4 / SPB steps][64 lanes][SPB thresholds] (1 KB each)
    const uint32_t ring_lane = (uint32_t)slots * 256u + (uint32_t)lane * (4u * SPB);

    if constexpr (TW) {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 1) {
            // ---- the threshold wavefront ----
            uint32_t w[4];
            uint32_t buf = 0;
            for (int s = 0; s < a.num_sweeps; ++s) {
                const float T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s])));
                const uint32_t sw = (uint32_t)s + a.sweep_offset;
#pragma unroll 1
                for (int t = 0; t < slots; t += 4) {
                    philox4x32_10((uint32_t)((t >> 2) * 64 + lane), sw, gid, 0u, a.seed_lo, a.seed_hi, w);
                    const f32x2_t l01 = neglog_u2(w[0], w[1]) * f32x2_t{T, T}, l23 = neglog_u2(w[2], w[3]) * f32x2_t{T, T};
                    const uint32_t at = ring_lane + buf;
                    if constexpr (SPB == 4) {
                        const f32x4 v = {l01.x, l01.y, l23.x, l23.y};
                        asm volatile("ds_write_b128 %0, %1" :: "v"(at), "v"(v) : "memory");
                    } else if constexpr (SPB == 2) {
                        asm volatile("ds_write_b64 %0, %1" :: "v"(at), "v"(l01) : "memory");
                        asm volatile("ds_write_b64 %0, %1 offset:512" :: "v"(at), "v"(l23) : "memory");
                    } else {
                        asm volatile("ds_write_b32 %0, %1" :: "v"(at), "v"(l01.x) : "memory");
                        asm volatile("ds_write_b32 %0, %1 offset:256" :: "v"(at), "v"(l01.y) : "memory");
                        asm volatile("ds_write_b32 %0, %1 offset:512" :: "v"(at), "v"(l23.x) : "memory");
                        asm volatile("ds_write_b32 %0, %1 offset:768" :: "v"(at), "v"(l23.y) : "memory");
                    }
                    buf ^= 1024u;
                    // the sweeping wavefront passes this barrier before it reads the group and the next one only after it
                    // has read it: the buffer written next (the other one) is free by then
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
            return;
        }
        __builtin_amdgcn_s_setprio(3);
    }

    // (pair-term weights, mi_sa_problem_set_pair_weights: S is sum_j w_j x_j; the one slot whose lanes carry weights other
    // than 1 is swept by a serial loop -- weighted_slot_sweep)
    const int wslot = WGT ? a.wslot : -1;
    const int wl = wslot >= 0 ? a.wgt[lane] : 0;
    int S = 0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        bool x;
        const bool real = i < n && a.lin[i] < INFINITY;            // (+inf linear term = hole of a padded layout: stays 0)
        if (init) {
            x = real && init[(size_t)r * n + i] != 0;
        } else {
            uint32_t iw[4];
            philox4x32_10((uint32_t)((t >> 2) * 64 + lane), 0u, gid, 1u, a.seed_lo, a.seed_hi, iw);
            const int c = t & 3;
            x = real && ((c == 0 ? iw[0] : (c == 1 ? iw[1] : (c == 2 ? iw[2] : iw[3]))) >> 31);
        }
        reinterpret_cast<uint32_t *>(lds)[i] = x ? 0x3c00u : 0u;
        S += (WGT && t == wslot) ? (int)wave_sum_i64(x ? (long long)wl : 0ll) : __popcll(__ballot(x));
    }

    constexpr int G = D / 4;
    const __amdgpu_buffer_rsrc_t rs_adj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(a.adj4), 0, slots * G * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_lin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.lin), 0, slots * 256, 0x00020000);
    struct SlotAdj { u32x4 col[G]; u32x4 val[G]; uint32_t lin; };
    struct BlockAdj { SlotAdj s[SPB]; };
    auto fetch_block = [&](int b) {                                 // the pair kernel's packing: neighbour word = 4 * index
        BlockAdj p;
        const int bb = b < blocks ? b : blocks - 1;
#pragma unroll
        for (int j = 0; j < SPB; ++j) {
            const int tt = bb * SPB + j;
            const int soff = tt * (G * 2048);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int so = soff + (g / 2) * 4096, io = (g & 1) * 2048;
                p.s[j].col[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16 + io, so, 0);
                p.s[j].val[g] = __builtin_amdgcn_raw_buffer_load_b128(rs_adj, lane * 16 + io + 1024, so, 0);
            }
            p.s[j].lin = __builtin_amdgcn_raw_buffer_load_b32(rs_lin, lane * 4, tt * 256, 0);
        }
        return p;
    };

    unsigned long long accepted = 0;
    uint32_t acc32 = 0;                                             // accepted moves of the running sweep
    float T = 1.0f;
    const float cp = a.c_pair;
    uint32_t sw = a.sweep_offset;
    // random words: `cw` = the four words of the group the step is in, `nx` = the next group's block in the making
    // (10 SPB / 4 rounds per step under the gathers; a short last group's remainder is finished when the group changes)
    uint32_t cw[4] = {0u, 0u, 0u, 0u};
    PhiloxPipe nx;
    int cur_g = -1;
    nx.start((uint32_t)lane, sw, gid, 0u, a.seed_lo, a.seed_hi);

    uint32_t ring_buf = 1024u;                                      // TW: which half of the ring the running group is in (toggled at its start)
    auto step = [&](int b, const BlockAdj &cur) {
        const int t0 = b * SPB;
        if constexpr (TW) {
            if ((t0 & 3) == 0) {                                    // wave-uniform: this block opens a new group of four slots
                __builtin_amdgcn_s_barrier();                       // ... whose thresholds are in the ring
                ring_buf ^= 1024u;
            }
        } else if ((t0 >> 2) != cur_g) {
            while (nx.done < 10) nx.round();
            cw[0] = nx.c0; cw[1] = nx.c1; cw[2] = nx.c2; cw[3] = nx.c3;
            cur_g = t0 >> 2;
            const int tn = ((t0 >> 2) + 1) * 4;                     // first slot of the next group, or of the next sweep
            const bool wrap = tn >= slots;
            nx.start((uint32_t)((wrap ? 0 : (tn >> 2)) * 64 + lane), wrap ? sw + 1u : sw, gid, 0u, a.seed_lo, a.seed_hi);
        }
        // (1) the LDS reads of all SPB slots (at most 16 stay in flight; the issue of the later ones paces itself)
        uint32_t own[SPB], word[SPB][16];
        constexpr bool PIPE = TW && D == 16 && SPB <= 2;            // (the forms the sampler's layouts run: staged waits below)
        float thr[SPB];
        typedef float thr_vec_t __attribute__((ext_vector_type(SPB == 1 ? 2 : SPB)));
        thr_vec_t tv;                                               // TW: the step's thresholds as they come from the ring
        if constexpr (PIPE) {
            // issue order: the own cells, the thresholds, then the gathers position by position across the slots -- LDS reads
            // return in order, so "at most N outstanding" below names a prefix of this sequence
#pragma unroll
            for (int j = 0; j < SPB; ++j)
                asm volatile("ds_read_b32 %0, %1" : "=v"(own[j]) : "v"(((t0 + j) * 64 + lane) * 4) : "memory");
            const uint32_t at = ring_lane + ring_buf + (uint32_t)(t0 & 3) * 256u;
            if constexpr (SPB == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(tv) : "v"(at));
            else asm volatile("ds_read_b32 %0, %1" : "=v"(tv.x) : "v"(at));
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < SPB; ++j)
                    asm volatile("ds_read_b32 %0, %1" : "=v"(word[j][k]) : "v"(cur.s[j].col[k / 4][k & 3]));
        } else {
#pragma unroll
        for (int j = 0; j < SPB; ++j) {
            asm volatile("ds_read_b32 %0, %1" : "=v"(own[j]) : "v"(((t0 + j) * 64 + lane) * 4) : "memory");
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("ds_read_b32 %0, %1" : "=v"(word[j][k]) : "v"(cur.s[j].col[k / 4][k & 3]));
        }
        }
        if constexpr (PIPE) {
            // (2) (the thresholds were requested with the own cells)
        } else if constexpr (TW) {
            // (2) the SPB thresholds of the step from the ring: one read, waited for with the gathers (below)
            const uint32_t at = ring_lane + ring_buf + (uint32_t)(t0 & 3) * 256u;
            if constexpr (SPB == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(tv) : "v"(at));
            else if constexpr (SPB == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(tv) : "v"(at));
            else asm volatile("ds_read_b32 %0, %1" : "=v"(tv.x) : "v"(at));
        } else {
            // (2) behind their issue: the SPB thresholds (pairs of them as packed arithmetic) and a share of the next random words
            uint32_t rw[SPB];
#pragma unroll
            for (int j = 0; j < SPB; ++j) {
                const int c = (t0 + j) & 3;                         // (SPB = 4: c = j; SPB = 2: 0, 1 or 2, 3)
                rw[j] = c == 0 ? cw[0] : (c == 1 ? cw[1] : (c == 2 ? cw[2] : cw[3]));
            }
            asm volatile("" : "+v"(rw[0]), "+v"(rw[SPB - 1]), "+v"(nx.c0), "+v"(nx.c1), "+v"(nx.c2), "+v"(nx.c3));
#pragma unroll
            for (int j = 0; j + 1 < SPB; j += 2) {
                const f32x2_t l2 = neglog_u2(rw[j], rw[j + 1]) * f32x2_t{T, T};
                thr[j] = l2.x;
                thr[j + 1] = l2.y;
            }
#pragma unroll
            for (int k = 0; k < (10 * SPB) / 4; ++k) nx.round();    // (a group lasts 4 / SPB steps: never past ten rounds)
        }
        // (3) the field sums: SPB independent chains
        float gi[SPB];
#pragma unroll
        for (int j = 0; j < SPB; ++j) gi[j] = __uint_as_float(cur.s[j].lin);
        auto fma_k = [&](int j, int k) {
            const half_t hx = __builtin_bit_cast(half_t, (uint16_t)word[j][k]);
            gi[j] = __builtin_fmaf(__uint_as_float(cur.s[j].val[k / 4][k & 3]), (float)hx, gi[j]);   // fma(val, x, g)
        };
        if constexpr (PIPE && SPB == 2) {
            // 35 reads in flight: own 0, own 1, thresholds, then w[0][k], w[1][k] for k = 0 .. 15.  A wavefront alone on its
            // SIMD is parked on s_waitcnt half of its time if it waits for all of them (profiles/r03_k2w_binding.json): the two
            // fma chains advance in three stages behind counted waits -- at most 15 outstanding: k < 8 of both slots are back;
            // at most 7: k < 12; none: all -- each wait naming what the stage before it produced, so nothing sinks below it
            asm volatile("s_waitcnt lgkmcnt(15)"
                         : "+v"(word[0][0]), "+v"(word[0][1]), "+v"(word[0][2]), "+v"(word[0][3]), "+v"(word[0][4]), "+v"(word[0][5]),
                           "+v"(word[0][6]), "+v"(word[0][7]), "+v"(word[1][0]), "+v"(word[1][1]), "+v"(word[1][2]), "+v"(word[1][3]),
                           "+v"(word[1][4]), "+v"(word[1][5]), "+v"(word[1][6]), "+v"(word[1][7]), "+v"(own[0]), "+v"(own[1]), "+v"(tv)
                         :: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) { fma_k(0, k); fma_k(1, k); }
            asm volatile("s_waitcnt lgkmcnt(7)"
                         : "+v"(word[0][8]), "+v"(word[0][9]), "+v"(word[0][10]), "+v"(word[0][11]), "+v"(word[1][8]), "+v"(word[1][9]),
                           "+v"(word[1][10]), "+v"(word[1][11]), "+v"(gi[0]), "+v"(gi[1])
                         :: "memory");
#pragma unroll
            for (int k = 8; k < 12; ++k) { fma_k(0, k); fma_k(1, k); }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(word[0][12]), "+v"(word[0][13]), "+v"(word[0][14]), "+v"(word[0][15]), "+v"(word[1][12]), "+v"(word[1][13]),
                           "+v"(word[1][14]), "+v"(word[1][15]), "+v"(gi[0]), "+v"(gi[1])
                         :: "memory");
#pragma unroll
            for (int k = 12; k < 16; ++k) { fma_k(0, k); fma_k(1, k); }
            thr[0] = tv[0];
            thr[1] = tv[1];
        } else if constexpr (PIPE && SPB == 1) {
            // one slot per step: 18 reads in flight (own, threshold, 16 gathers), the fma chain in four stages of four
            asm volatile("s_waitcnt lgkmcnt(12)"
                         : "+v"(word[0][0]), "+v"(word[0][1]), "+v"(word[0][2]), "+v"(word[0][3]), "+v"(own[0]), "+v"(tv) :: "memory");
#pragma unroll
            for (int k = 0; k < 4; ++k) fma_k(0, k);
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(word[0][4]), "+v"(word[0][5]), "+v"(word[0][6]), "+v"(word[0][7]), "+v"(gi[0]) :: "memory");
#pragma unroll
            for (int k = 4; k < 8; ++k) fma_k(0, k);
            asm volatile("s_waitcnt lgkmcnt(4)"
                         : "+v"(word[0][8]), "+v"(word[0][9]), "+v"(word[0][10]), "+v"(word[0][11]), "+v"(gi[0]) :: "memory");
#pragma unroll
            for (int k = 8; k < 12; ++k) fma_k(0, k);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(word[0][12]), "+v"(word[0][13]), "+v"(word[0][14]), "+v"(word[0][15]), "+v"(gi[0]) :: "memory");
#pragma unroll
            for (int k = 12; k < 16; ++k) fma_k(0, k);
            thr[0] = tv[0];
        } else {
        if constexpr (TW) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv) :: "memory");
#pragma unroll
            for (int j = 0; j < SPB; ++j) thr[j] = tv[j];
        }
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            if (g0 > 0) {
#pragma unroll
                for (int j = 0; j < SPB; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        asm volatile("ds_read_b32 %0, %1" : "=v"(word[j][k]) : "v"(cur.s[j].col[g0 + k / 4][k & 3]));
            }
#pragma unroll
            for (int j = 0; j < SPB; ++j) {
                // slot j's reads are older than slot j + 1's: everything but the 17 (SPB - 1 - j) youngest has returned
                // (the hardware counter holds 15 at most, so for the early slots this is a full wait in effect)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(word[j][0]), "+v"(word[j][1]), "+v"(word[j][2]), "+v"(word[j][3]), "+v"(word[j][4]),
                               "+v"(word[j][5]), "+v"(word[j][6]), "+v"(word[j][7]), "+v"(word[j][8]), "+v"(word[j][9]),
                               "+v"(word[j][10]), "+v"(word[j][11]), "+v"(word[j][12]), "+v"(word[j][13]), "+v"(word[j][14]),
                               "+v"(word[j][15]), "+v"(own[j]), "+v"(thr[j])
                             :: "memory");
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < SPB; ++j) {
                    const half_t hx = __builtin_bit_cast(half_t, (uint16_t)word[j][k]);
                    gi[j] = __builtin_fmaf(__uint_as_float(cur.s[j].val[g0 + k / 4][k & 3]), (float)hx, gi[j]);   // fma(val, x, g)
                }
        }
        }
        // (4) the accept masks, slot after slot (k_anneal_csr_rank1's rounds); the block holds no edge, so a later slot's
        // field sum does not see an earlier slot's flips -- only the sum does
#pragma unroll
        for (int j = 0; j < SPB; ++j) {
            const int i = (t0 + j) * 64 + lane;
            const uint32_t xi = own[j] >> 13;                       // 0x3c00 -> 1
            if (WGT && t0 + j == wslot) {
                // ---- the slot of the weighted variables: a serial sweep (few lanes, no sparse couplings) ----
                const uint64_t F = weighted_slot_sweep(gi[j], thr[j], wl, cp, xi, S, lane);
                acc32 += (uint32_t)__popcll(F);
                uint32_t tg2;
                asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(tg2) : "v"(0x3c00u), "s"(F));
                asm volatile("ds_write_b32 %0, %1" :: "v"(i * 4), "v"(own[j] ^ tg2) : "memory");
                continue;
            }
            const uint64_t X = __ballot(own[j] != 0u);
            const uint32_t sgnbit = xi << 31;                       // dE = x ? -f : f
            const float gs = __uint_as_float(__float_as_uint(gi[j]) ^ sgnbit);
            const float cs = __uint_as_float(__float_as_uint(cp) ^ sgnbit);
            const uint64_t A0 = __ballot(gs + cs * (float)(S - (int)xi) < thr[j]);
            if (A0 != 0ull) {                                       // wave-uniform
                const int base = S - (int)xi - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(X >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)X, 0u));
                // rounds until the states after the moves reproduce themselves (ends by itself: after k rounds the lowest
                // k lanes are final); only wave-uniform masks live across the loop
                uint64_t Bm = A0 ^ X;
#pragma nounroll
                for (;;) {
                    const int s_i = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(Bm >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)Bm, (uint32_t)base));
                    const uint64_t B2 = __ballot(gs + cs * (float)s_i < thr[j]) ^ X;
                    uint64_t d = B2 ^ Bm;
                    asm("" : "+s"(d));
                    Bm = B2;
                    if (d == 0ull) break;
                }
                const uint64_t A = Bm ^ X;
                uint32_t tgl;
                asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(tgl) : "v"(0x3c00u), "s"(A));
                asm volatile("ds_write_b32 %0, %1" :: "v"(i * 4), "v"(own[j] ^ tgl) : "memory");
                S += __popcll(Bm) - __popcll(X);
                acc32 += (uint32_t)__popcll(A);
            }
        }
    };

    for (int s = 0; s < a.num_sweeps; ++s) {
        T = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.temps[a.temps_per_replica ? r : s])));
        sw = (uint32_t)s + a.sweep_offset;
        cur_g = -1;
        BlockAdj P = fetch_block(0), Q;
#pragma unroll 1
        for (int b = 0; b < blocks; b += 2) {
            Q = fetch_block(b + 1);
            step(b, P);
            if (b + 1 < blocks) {                                   // wave-uniform
                P = fetch_block(b + 2);
                step(b + 1, Q);
            }
        }
        accepted += acc32;
        acc32 = 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- epilogue: states out, exact fp64 energy (the sums of k_anneal_csr_rank1) ----
    uint8_t *dst = static_cast<uint8_t *>(a.states) + (size_t)r * n;
    const uint32_t *cell = reinterpret_cast<const uint32_t *>(lds);
    long long cnt = 0, cnt2 = 0;                                    // sum_j w_j x_j, sum_j w_j^2 x_j
    double e = 0.0;
    for (int t = 0; t < slots; ++t) {
        const int i = t * 64 + lane;
        const bool on = cell[i] != 0u;
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
            if (cell[cc] != 0u) acc += vv;
        }
        e += (a.lin64 ? a.lin64[i] : (double)a.lin[i]) + 0.5 * acc;
    }
    e = wave_sum_f64(e);
    if (lane == 0) {
        const double cp64 = a.ell_val64 ? a.c_pair64 : (double)a.c_pair;
        a.energy[r] = e + cp64 * 0.5 * ((double)cnt * (double)cnt - (double)cnt2) + a.offset;
        atomicAdd(&a.stats[1], accepted);
    }
}

template <typename KernelT>
int launch_wide(KernelT kernel, const EllArgs &a, int spb, bool tw, hipStream_t st)
{
    const size_t lds = (size_t)a.slots * 256 + (tw ? 2048 : 0);    // the cells; tw: + the ring of thresholds
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "csr_rank1 wide kernel: n = %d exceeds the state LDS budget", a.n);
    if (a.slots % spb != 0) return fail(MI_EINVAL, "csr_rank1 wide kernel: %d slots are not whole blocks of %d", a.slots, spb);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel(tw ? "k_anneal_csr_rank1_wide<%d, %d, tw>" : "k_anneal_csr_rank1_wide<%d, %d>", a.D, spb);
    hipLaunchKernelGGL(kernel, dim3(a.R), dim3(tw ? 128 : 64), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

template <typename KernelT>
int launch_split(KernelT kernel, const EllArgs &a, int nw, hipStream_t st)
{
    const size_t lds = (size_t)a.slots * 256 + kCommBytes + (nw > 1 ? kRingBytes : 0);
    if (lds > 160 * 1024) return fail(MI_EUNSUPPORTED, "csr_rank1 split kernel: n = %d exceeds the state LDS budget", a.n);
    if (a.slots % nw != 0) return fail(MI_EINVAL, "csr_rank1 split kernel: %d slots are not whole blocks of %d", a.slots, nw);
    if (nw > 1 && a.slots % 4 != 0)
        return fail(MI_EINVAL, "csr_rank1 split kernel: with %d wavefronts per replica the slots (%d) must come in whole groups of four", nw, a.slots);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    note_kernel("k_anneal_csr_rank1_split<%d, %d>", a.D, nw);
    hipLaunchKernelGGL(kernel, dim3(a.R), dim3(64 * nw), lds, st, a);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace

// one wavefront per replica, spb = 2 / 4 slots per step (a model whose every block of 64 * spb seats is free of internal
// edges; a.adj4 = the pair kernel's packing)
int mi_launch_csr_rank1_wide(const EllArgs &a, int spb, bool tw, hipStream_t st)
{
    if (!a.adj4) return fail(MI_EHIP, "csr_rank1 wide kernel: packed adjacency missing");
    if (a.wslot >= 0) {                       // pair-term weights: one slot per step beside a threshold wavefront
        if (tw && a.D == 16 && spb == 1) return launch_wide(k_anneal_csr_rank1_wide<16, 1, true, true>, a, spb, true, st);
        if (tw && a.D == 32 && spb == 1) return launch_wide(k_anneal_csr_rank1_wide<32, 1, true, true>, a, spb, true, st);
        return fail(MI_EUNSUPPORTED, "csr_rank1 wide kernel: a model with pair-term weights runs one slot per step beside a threshold wavefront");
    }
    if (tw) {
        if (a.D == 16 && spb == 1) return launch_wide(k_anneal_csr_rank1_wide<16, 1, true>, a, spb, true, st);
        if (a.D == 16 && spb == 2) return launch_wide(k_anneal_csr_rank1_wide<16, 2, true>, a, spb, true, st);
        if (a.D == 16 && spb == 4) return launch_wide(k_anneal_csr_rank1_wide<16, 4, true>, a, spb, true, st);
        if (a.D == 32 && spb == 1) return launch_wide(k_anneal_csr_rank1_wide<32, 1, true>, a, spb, true, st);
        if (a.D == 32 && spb == 2) return launch_wide(k_anneal_csr_rank1_wide<32, 2, true>, a, spb, true, st);
    } else {
        if (a.D == 16 && spb == 2) return launch_wide(k_anneal_csr_rank1_wide<16, 2, false>, a, spb, false, st);
        if (a.D == 16 && spb == 4) return launch_wide(k_anneal_csr_rank1_wide<16, 4, false>, a, spb, false, st);
        if (a.D == 32 && spb == 2) return launch_wide(k_anneal_csr_rank1_wide<32, 2, false>, a, spb, false, st);
    }
    return fail(MI_EUNSUPPORTED, "csr_rank1 wide kernel: width %d / %d slots per step%s not built", a.D, spb, tw ? " with a threshold wavefront" : "");
}

// a.adj4 must hold the pair kernel's packing (neighbour word = 4 * index) of a model whose every block of 64 * nw seats
// is free of internal edges
int mi_launch_csr_rank1_split(const EllArgs &a, int nw, hipStream_t st)
{
    if (!a.adj4) return fail(MI_EHIP, "csr_rank1 split kernel: packed adjacency missing");
    if (a.D == 16 && nw == 4) return launch_split(k_anneal_csr_rank1_split<16, 4>, a, nw, st);
    if (a.D == 16 && nw == 2) return launch_split(k_anneal_csr_rank1_split<16, 2>, a, nw, st);
    if (a.D == 16 && nw == 1) return launch_split(k_anneal_csr_rank1_split<16, 1>, a, nw, st);
    if (a.D == 32 && nw == 4) return launch_split(k_anneal_csr_rank1_split<32, 4>, a, nw, st);
    if (a.D == 32 && nw == 2) return launch_split(k_anneal_csr_rank1_split<32, 2>, a, nw, st);
    if (a.D == 32 && nw == 1) return launch_split(k_anneal_csr_rank1_split<32, 1>, a, nw, st);
    return fail(MI_EUNSUPPORTED, "csr_rank1 split kernel: width %d / %d wavefronts not built", a.D, nw);
}

}  // namespace mi_sa_impl
