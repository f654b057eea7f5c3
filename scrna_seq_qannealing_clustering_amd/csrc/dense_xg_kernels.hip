// dense_xg_kernels.hip -- K1g: the dense-QUBO chain for LARGE models (4096 < n <= 65536) and MANY replicas, with the
// row updates of all replicas as one GEMM-shaped pass per block of rows on the matrix cores (gfx950).
//
// Same chain as K1 / K1w / K1m / K1x, bit for bit (oracle 2a).  K1x gives a replica a workgroup
// and streams one 4 n-byte row of Q per accepted flip through it: at n = 50 000 that is 213 KB behind a barrier and a
// full L2 round trip per flip (7.7 us), and every replica fetches every row for itself.  Here all R replicas walk
// the rows TOGETHER, 64 rows (one "block") at a time, with the cached fields of all replicas in HBM / Infinity
// Cache as F[replica range][column][64 replicas]:
//   DIAG(b)   four lanes per replica: the 64 decisions of block b in sequence, on the replica's 64 fields of the
//             block's own columns and the 64 x 64 coupling block Q2[b][b] (LDS, broadcast reads) -- leaves the signs
//             S[k][r] in {-1, 0, +1} (0 = rejected), the new state bits, and a "some replica flipped" flag per
//             64 replicas;
//   PANEL(b)  F[:, r] += sum_k Q2[64 b + k][:] * S[k][r]  for ALL columns and replicas: tiles of
//             v_mfma_f32_16x16x4_f32 chained over the 64 rows.  The f32-input MFMA is an exact fp32 fmaf chain in
//             k order, D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))), i.e. the oracle's "f += sgn * Q2[row]"
//             for the accepted rows in row order (K1m relies on the same identity); a rejected row contributes
//             fma(q, 0, f) = f.  The block's own columns are updated by the same pass, so DIAG's private copies
//             are simply dropped.
// A Q row is read once per sweep for all replicas (10.6 GB per sweep at n = 50 000 instead of 213 KB per accepted
// flip and replica); what the pass costs is the read-modify-write of F (n_pad x R x 8 B per block) and
// 2 * 64 * n_pad * R flop per block on the matrix pipe.  The read-modify-write is paid once per GROUP of eight
// blocks: inside a group the later blocks' own columns get the earlier blocks' rows by small passes into a side
// buffer (Tm), so that their DIAGs can run before the group's full pass.  The chain of the NEXT group runs on a second
// stream with 8 compute units of its own beside the current group's full pass (events order them; see the launcher).
// Thresholds for four blocks at a time (one Philox block serves four 64-variable slots).
// Used for R >= 256 replicas or n >= 16384; K1x keeps small batches of small problems and the cold end of a cooling
// run (mi_sa.hip).
#include "mi_sa_device.h"
#include <cstdlib>
#include <vector>

namespace mi_sa_impl {
namespace {

typedef float f32x4acc __attribute__((ext_vector_type(4)));

constexpr int kXgB = 64;                 // rows per block = variables per slot of the RNG addressing
constexpr int kXgCols = 256;             // columns per PANEL workgroup (4 waves x 64)
constexpr int kXgReps = 64;              // replicas per PANEL workgroup (= one flag word)
constexpr int kXgGrp = 8;                // blocks per group: F is read and written once per 64 * kXgGrp rows

struct XgArgs {
    const float *Q2;        // n rows x stride floats (zero diagonal, zero padding)
    const float *diag;      // stride floats
    size_t stride;          // floats per row of Q2
    float *F;               // cached fields, [Rp / 64][ncols][64]: the 64 replicas of a PANEL workgroup contiguous per
                            // column, so that its 256 x 64 tile is ONE 64 KB run of memory (see fidx)
    unsigned long long *XT; // [nblocks][Rp] state bits of a block, bit k = x of variable 64 b + k
    float *S;               // [2][kXgGrp][64][Rp] signs of the blocks of a group (two groups in flight: parity)
    float *Tm;              // [Rp / 64][kXgGrp - 1][64][64] fields of the group's later blocks with the earlier blocks' rows applied
    float *TH;              // [kXgGrp][64][Rp] thresholds of the group's blocks
    unsigned int *flags;    // [2][kXgGrp][Rp / 64]: block j of the group flipped something in these 64 replicas
    const float *temps;
    const uint8_t *init;    // nullable, R x n
    uint8_t *states;        // R x n
    double *energy;
    unsigned long long *stats;
    double offset;
    int n, R, Rp, ncols, nblocks;
    uint32_t replica_offset, seed_lo, seed_hi;
    int temps_per_replica;
};

__device__ __forceinline__ size_t fidx(const XgArgs &a, int col, int r)
{
    return ((size_t)(r >> 6) * a.ncols + col) * 64 + (r & 63);
}

__device__ __forceinline__ size_t tmidx(int slot, int col, int r)
{
    return (((size_t)(r >> 6) * (kXgGrp - 1) + slot) * 64 + col) * 64 + (r & 63);
}

// ---- state bits: XT[b][r] from the given initial states or the chain's own random start (tag 1) ----
__global__ void __launch_bounds__(64) k_xg_init_state(XgArgs a)
{
    const int lane = threadIdx.x, r = blockIdx.x, tg = blockIdx.y;       // one wavefront per (replica, four blocks)
    const uint32_t g = a.replica_offset + (uint32_t)r;
    uint32_t w[4] = {0, 0, 0, 0};
    if (!a.init) philox4x32_10((uint32_t)(tg * 64 + lane), 0u, g, 1u, a.seed_lo, a.seed_hi, w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int b = 4 * tg + c, i = b * 64 + lane;
        if (b >= a.nblocks) break;
        bool bit = false;
        if (i < a.n) bit = a.init ? (a.init[(size_t)r * a.n + i] != 0) : ((w[c] >> 31) != 0);
        const unsigned long long m = __ballot(bit);
        if (lane == 0) a.XT[(size_t)b * a.Rp + r] = m;
    }
}

__global__ void __launch_bounds__(256) k_xg_fill_fields(XgArgs a)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;           // over ncols * Rp
    if (idx < (size_t)a.ncols * a.Rp) a.F[idx] = a.diag[(idx >> 6) % (size_t)a.ncols];
}

// thresholds of blocks 4 tg .. 4 tg + 3 for sweep s: TH[c][lane][r] = -ln(u) * T   (-inf: no such variable / replica);
// blockIdx.z = which four of the group's blocks (one launch per group)
__global__ void __launch_bounds__(256) k_xg_thresholds(XgArgs a, int tg0, uint32_t sweep, int s_local)
{
    const int tg = tg0 + (int)blockIdx.z, slot0 = 4 * (int)blockIdx.z;
    const int r = blockIdx.x * 256 + threadIdx.x, lane = blockIdx.y;
    if (r >= a.Rp) return;
    float th[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (r < a.R) {
        const float T = a.temps[a.temps_per_replica ? r : s_local];
        uint32_t w[4];
        philox4x32_10((uint32_t)(tg * 64 + lane), sweep, a.replica_offset + (uint32_t)r, 0u, a.seed_lo, a.seed_hi, w);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if ((4 * tg + c) * 64 + lane < a.n) th[c] = neglog_u(w[c]) * T;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) a.TH[((size_t)(slot0 + c) * 64 + lane) * a.Rp + r] = th[c];
}

// ---- DIAG(b): FOUR lanes per replica (a quad), 64 replicas per workgroup.  Lane q of a quad holds the fields of the
// block's columns q, 4 + q, ..., 60 + q; step k decides row k in lane q = k % 4, the sign goes round the quad by DPP,
// every lane applies the coupling row to its 16 fields (those of rows already decided are dead and may take
// garbage) -- a quarter of the fmaf chain per lane of the one-thread-per-replica form (25 -> 10 us per block).
// j = b % kXgGrp: position in its group of blocks -- block 0 reads its fields from F, the others from Tm (F plus the
// rows of the group's earlier blocks, k_xg_panel<true>) ----
template <int QK>
__device__ __forceinline__ float xg_quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), QK * 0x55, 0xf, 0xf, false));
}

__global__ void __launch_bounds__(256) k_xg_diag(XgArgs a, int b, int force, int par)
{
    __shared__ __attribute__((aligned(16))) float Ct[kXgB][4][16];       // Ct[k][q][m] = Q2[R0 + k][R0 + q + 4 m]
    __shared__ int any_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = tid & 3, r = blockIdx.x * 64 + (tid >> 2);
    const int R0 = b * kXgB, j = b % kXgGrp;
    float *Sj = a.S + ((size_t)par * kXgGrp + j) * kXgB * a.Rp;
    if (tid == 0) any_s = 0;
    if (!force) {
        for (int e = tid; e < kXgB * kXgB / 4; e += 256) {
            const int k = e >> 4, c = e & 15;                             // columns R0 + 4 c .. + 3: q = 0..3 at m = c
            f32x4acc v = {0, 0, 0, 0};
            if (R0 + k < a.n) v = *reinterpret_cast<const f32x4acc *>(a.Q2 + (size_t)(R0 + k) * a.stride + R0 + 4 * c);
            Ct[k][0][c] = v[0]; Ct[k][1][c] = v[1]; Ct[k][2][c] = v[2]; Ct[k][3][c] = v[3];
        }
    }
    __syncthreads();
    const unsigned long long xw = a.XT[(size_t)b * a.Rp + r];
    unsigned long long accepted = 0, word = 0;
    bool any = false;
    if (force) {
        // field (re)initialisation: the "flips" are the set bits (f = diag + sum of the rows with x = 1)
#pragma unroll
        for (int m = 0; m < 16; ++m) Sj[(size_t)(q + 4 * m) * a.Rp + r] = ((xw >> (q + 4 * m)) & 1ull) ? 1.0f : 0.0f;
        any = xw != 0ull;
    } else {
        float t[16], th[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            t[m] = j == 0 ? a.F[fidx(a, R0 + q + 4 * m, r)] : a.Tm[tmidx(j - 1, q + 4 * m, r)];
            th[m] = a.TH[((size_t)j * 64 + q + 4 * m) * a.Rp + r];
        }
        static_for<0, kXgB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int mk = k >> 2, qk = k & 3;
            const bool xk = ((xw >> k) & 1ull) != 0ull;
            const float dE = xk ? -t[mk] : t[mk];
            const bool acc = (q == qk) && dE < th[mk];                    // the lane that holds row k
            const float so = acc ? (xk ? -1.0f : 1.0f) : 0.0f;
            const float sk = xg_quad_bcast<qk>(so);
            if (q == qk) Sj[(size_t)k * a.Rp + r] = so;
            if (acc) { word |= 1ull << k; ++accepted; }
            if (__ballot(acc) != 0ull) {                                  // (wave-uniform: nobody flipped row k)
#pragma unroll
                for (int g4 = mk >> 2; g4 < 4; ++g4) {
                    const f32x4acc c4 = *reinterpret_cast<const f32x4acc *>(&Ct[k][q][4 * g4]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[4 * g4 + i] = __fmaf_rn(c4[i], sk, t[4 * g4 + i]);
                }
                any = any || acc;
            }
        });
        // the quad's accepted rows -> the replica's state word
        unsigned int lo = (unsigned int)word, hi = (unsigned int)(word >> 32);
        lo |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xf, 0xf, false);
        hi |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, false);
        lo |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xf, 0xf, false);
        hi |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xf, 0xf, false);
        if (q == 0) a.XT[(size_t)b * a.Rp + r] = xw ^ (((unsigned long long)hi << 32) | lo);
    }
    if (__ballot(any) != 0ull && lane == 0) atomicOr(&any_s, 1);
    __syncthreads();
    if (tid == 0) a.flags[((size_t)par * kXgGrp + j) * (a.Rp / kXgReps) + blockIdx.x] = any_s ? 1u : 0u;
    if (!force) {
        // accepted flips of this wavefront -> stats[1]
        unsigned long long tot = accepted;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
        if (lane == 0 && tot) atomicAdd(&a.stats[1], tot);
    }
}

// ---- CHAIN(g): the whole chain of a group in ONE launch, one workgroup per 64 replicas.  The DIAGs and small passes of
// a group never leave their replica range: the fields of the group's later blocks (7 x 64 columns x 64 replicas) live
// in LDS instead of Tm, DIAG(j) decides on them, and the rows of block j go onto the columns of blocks j + 1 .. by
// chained MFMAs straight out of and back into LDS (wave w: columns 16 w .. of every later block; its Q2 operands for
// the first two later blocks are loaded BEFORE the 64 sequential decisions and arrive under them).  Same arithmetic
// in the same order as k_xg_diag + k_xg_panel<true> (tests compare the two), 15 launches and their gaps fewer per
// group: below 256 replicas the chain IS the run time.  Signs, state words and flags go to HBM as before (the
// group's full pass reads them); thresholds come from k_xg_thresholds. ----
constexpr int kXgFS = 68;                // padded replica stride of the LDS fields (16-byte tile reads, 2-way at worst)

__global__ void __launch_bounds__(256, 1) k_xg_chain(XgArgs a, int g, int nbg, int par)
{
    __shared__ __attribute__((aligned(16))) float Fo[kXgGrp - 1][kXgB][kXgFS];   // fields of blocks 1 .. of the group
    __shared__ __attribute__((aligned(16))) float Ct[kXgB][4][16];               // Ct[k][q][m] = Q2[R0 + k][R0 + q + 4 m]
    __shared__ __attribute__((aligned(16))) float Sl[kXgB][kXgReps + 16];         // signs of the running block
    __shared__ int any_s[kXgGrp];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = tid & 3, rl = tid >> 2, r = blockIdx.x * 64 + rl;
    const int lr = lane & 15, lq = lane >> 4;
    const int ranges = a.Rp / kXgReps;
    if (tid < kXgGrp) any_s[tid] = 0;
    // the later blocks' fields: 64 columns x 64 replicas of a block are one 16 KB run of F
    for (int jb = 1; jb < nbg; ++jb) {
        const float *src = a.F + fidx(a, (kXgGrp * g + jb) * kXgB, blockIdx.x * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, col = e >> 4, r4 = (e & 15) * 4;
            *reinterpret_cast<f32x4acc *>(&Fo[jb - 1][col][r4]) = *reinterpret_cast<const f32x4acc *>(src + (size_t)col * 64 + r4);
        }
    }
    unsigned long long accepted = 0;
    // What a block's decisions need from HBM -- its coupling block, thresholds and state word -- is fetched one block ahead,
    // under the decisions of the block before (a round trip to memory in front of every block's first decision otherwise).
    auto load_ct = [&](f32x4acc (&dst)[4], int R0) {          // the 64 x 64 coupling block Q2[R0 ..][R0 ..]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, k = e >> 4, c = e & 15;
            dst[i] = f32x4acc{0, 0, 0, 0};
            if (R0 + k < a.n) dst[i] = *reinterpret_cast<const f32x4acc *>(a.Q2 + (size_t)(R0 + k) * a.stride + R0 + 4 * c);
        }
    };
    auto store_ct = [&](const f32x4acc (&src)[4]) {            // Ct[k][q][m] = Q2[R0 + k][R0 + q + 4 m]: columns 4 c .. + 3 are q = 0..3 at m = c
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, k = e >> 4, c = e & 15;
            Ct[k][0][c] = src[i][0]; Ct[k][1][c] = src[i][1]; Ct[k][2][c] = src[i][2]; Ct[k][3][c] = src[i][3];
        }
    };
    f32x4acc ctn[4];
    float th[16], thn[16];
    f32x2 t[8];                                                // the lane's 16 fields as pairs: two row updates per v_pk_fma_f32
    unsigned long long xw, xwn = 0;
    {
        const int b0 = kXgGrp * g;
        load_ct(ctn, b0 * kXgB);
        xw = a.XT[(size_t)b0 * a.Rp + r];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            th[m] = a.TH[((size_t)0 * 64 + q + 4 * m) * a.Rp + r];
            t[m >> 1][m & 1] = a.F[fidx(a, b0 * kXgB + q + 4 * m, r)];
        }
        store_ct(ctn);
    }
    for (int j = 0; j < nbg; ++j) {
        const int b = kXgGrp * g + j, R0 = b * kXgB;
        float *Sj = a.S + ((size_t)par * kXgGrp + j) * kXgB * a.Rp;
        // small passes: wave w owns columns 16 w .. 16 w + 15 of EVERY later block (the same work for all four waves whatever
        // the number of later blocks).  Its operands for the first two later blocks (rows of block j) are loaded here, in
        // flight under the 64 decisions; the ones after arrive under the MFMAs of the block before.
        float qa[16], qn[16];
        auto load_q = [&](float (&dst)[16], int jb) {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const int row = R0 + 4 * ks + lq;
                dst[ks] = row < a.n ? a.Q2[(size_t)row * a.stride + (size_t)(kXgGrp * g + jb) * kXgB + 16 * wave + lr] : 0.0f;
            }
        };
        if (j + 1 < nbg) load_q(qa, j + 1);
        if (j + 2 < nbg) load_q(qn, j + 2);
        __syncthreads();                                       // Ct staged; Fo carries the rows of blocks < j
        if (j > 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) t[m >> 1][m & 1] = Fo[j - 1][q + 4 * m][rl];
        }
        if (j + 1 < nbg) {                                     // the next block's inputs, in flight under this block's decisions
            load_ct(ctn, R0 + kXgB);
            xwn = a.XT[(size_t)(b + 1) * a.Rp + r];
#pragma unroll
            for (int m = 0; m < 16; ++m) thn[m] = a.TH[((size_t)(j + 1) * 64 + q + 4 * m) * a.Rp + r];
        }
        unsigned long long word = 0;
        bool any = false;
        static_for<0, kXgB>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int mk = k >> 2, qk = k & 3;
            // the coupling row of step k is read before its decision is known (at the hot end every row is flipped by
            // some replica of the wavefront): its LDS latency runs beside the decision instead of behind it
            f32x4acc c4[4];
#pragma unroll
            for (int g4 = mk >> 2; g4 < 4; ++g4) c4[g4] = *reinterpret_cast<const f32x4acc *>(&Ct[k][q][4 * g4]);
            const bool xk = ((xw >> k) & 1ull) != 0ull;
            const float tk = t[mk >> 1][mk & 1];
            const float dE = xk ? -tk : tk;
            const bool acc = (q == qk) && dE < th[mk];                    // the lane that holds row k
            const float so = acc ? (xk ? -1.0f : 1.0f) : 0.0f;
            const float sk = xg_quad_bcast<qk>(so);
            if (q == qk) Sl[k][rl] = so;
            if (acc) { word |= 1ull << k; ++accepted; }
            if (__ballot(acc) != 0ull) {                                  // (wave-uniform: nobody flipped row k)
                const f32x2 s2 = {sk, sk};
#pragma unroll
                for (int g4 = mk >> 2; g4 < 4; ++g4) {                    // (c * s + t per element, as __fmaf_rn(c, s, t))
                    t[2 * g4 + 0] = __builtin_elementwise_fma(f32x2{c4[g4][0], c4[g4][1]}, s2, t[2 * g4 + 0]);
                    t[2 * g4 + 1] = __builtin_elementwise_fma(f32x2{c4[g4][2], c4[g4][3]}, s2, t[2 * g4 + 1]);
                }
                any = any || acc;
            }
        });
        {   // the quad's accepted rows -> the replica's state word
            unsigned int lo = (unsigned int)word, hi = (unsigned int)(word >> 32);
            lo |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xf, 0xf, false);
            hi |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, false);
            lo |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xf, 0xf, false);
            hi |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xf, 0xf, false);
            if (q == 0) a.XT[(size_t)b * a.Rp + r] = xw ^ (((unsigned long long)hi << 32) | lo);
        }
        if (__ballot(any) != 0ull && lane == 0) atomicOr(&any_s[j], 1);
        __syncthreads();                                       // signs in Sl, the block's flag complete; Ct is free
        if (j + 1 < nbg) store_ct(ctn);
        for (int e = tid; e < kXgB * kXgReps / 4; e += 256) {  // ... and on their way to HBM for the group's full pass
            const int k = e >> 4, r4 = (e & 15) * 4;
            *reinterpret_cast<f32x4acc *>(Sj + (size_t)k * a.Rp + blockIdx.x * 64 + r4) = *reinterpret_cast<const f32x4acc *>(&Sl[k][r4]);
        }
        const bool live = any_s[j] != 0;
        if (tid == 0) a.flags[((size_t)par * kXgGrp + j) * ranges + blockIdx.x] = live ? 1u : 0u;
        if (live && j + 1 < nbg) {
            // rows of block j onto the columns of the later blocks: C[i = replica][j = column] tiles of this wave's 16 columns,
            // chained in row order; the signs (A operand) are the same for every later block
            float sa[16][4];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) sa[ks][rt] = Sl[4 * ks + lq][16 * rt + lr];                    // A[i = lr][k = lq]
#pragma unroll 1
            for (int jb = j + 1; jb < nbg; ++jb) {
                f32x4acc acc[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt] = *reinterpret_cast<const f32x4acc *>(&Fo[jb - 1][16 * wave + lr][16 * rt + 4 * lq]);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[ks][rt], qa[ks], acc[rt], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<f32x4acc *>(&Fo[jb - 1][16 * wave + lr][16 * rt + 4 * lq]) = acc[rt];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) qa[ks] = qn[ks];
                if (jb + 2 < nbg) load_q(qn, jb + 2);
            }
        }
        __syncthreads();                                       // Fo and Sl are free for the next block, its Ct is in place
        xw = xwn;
#pragma unroll
        for (int m = 0; m < 16; ++m) th[m] = thn[m];
    }
    // accepted flips of this wavefront -> stats[1]
    unsigned long long tot = accepted;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
    if (lane == 0 && tot) atomicAdd(&a.stats[1], tot);
}

// ---- PANEL: fields (+)= sum over rows of Q2[row][col] * S[row][r], the rows in order, as chained MFMAs ----
// workgroup = 256 columns x 64 replicas, wave w = 64 columns: 4 x 4 tiles, C[i = replica][j = column].
//   MINI = false: the whole group g (blocks G g .. G g + nbg - 1, G = kXgGrp; 64 nbg rows) onto ALL columns of F -- F is read and
//                 written once per 256 rows, which is what this pass costs besides the MFMAs;
//   MINI = true:  the rows of block G g + j alone onto the columns of the group's LATER blocks (workgroup y, wave w: block
//                 G g + j + 1 + 4 y + w), into Tm -- the fields the next DIAGs of the group decide on.  F itself is
//                 untouched until the group's full pass, which applies the same rows in the same order.
// The full pass of a group is launched in two parts: first the 256 columns that are the NEXT group's own (cy_only), then
// the rest (cy_skip) -- the next group's DIAGs need only the first part and run beside the second (dense_xg host loop).
// NCT = 16-column tiles per wave: 4 (a workgroup owns 256 columns) everywhere but in the first part of a full pass, whose
// few workgroups sit on the critical path of the next group's chain -- there 1 (64 columns per workgroup, four times
// the workgroups, a quarter of the matrix work each).
template <bool MINI, int NCT>
__global__ void __launch_bounds__(256, 4) k_xg_panel(XgArgs a, int g, int nbg, int j, int par, int cy_only, int cy_skip)
{
    constexpr int CW = 16 * NCT;                              // columns per wave
    constexpr int AS = kXgCols + 16, SS = kXgReps + 16;       // padded LDS strides: conflict-free operand reads
    constexpr int CR = 16;                                    // rows of Q2 per chunk in LDS (the next chunk waits in registers)
    __shared__ __attribute__((aligned(16))) float Apan[CR][AS];
    __shared__ __attribute__((aligned(16))) float Ssl[kXgB][SS];
    // workgroup -> tile, XCD-aware: consecutive workgroup ids go round the 8 XCDs, each with an L2 of its own.  The replica
    // ranges of one column range read the same rows of Q2: they get consecutive ids on ONE XCD (a contiguous chunk of
    // the tile list per XCD, bijective for any grid), so Q2 leaves HBM once per pass, not once per XCD.
    const int ranges = a.Rp / kXgReps;
    int tile = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (!MINI) {
        const int nwg = (int)(gridDim.x * gridDim.y), qq = nwg / 8, rr = nwg % 8, xcd = tile % 8;
        tile = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + tile / 8;
    }
    const int rx = tile % (int)gridDim.x, ty = tile / (int)gridDim.x;
    const int cy = cy_only >= 0 ? cy_only + ty / (4 / NCT) : ty;     // (NCT = 1: four workgroups per 256 columns)
    if (!MINI && cy_skip >= 0 && cy >= cy_skip && cy < cy_skip + kXgGrp / 4) return;
    const int yy = MINI ? ty : 0;                             // MINI: which four of the later blocks
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int rep0 = rx * kXgReps;
    const int jb0 = MINI ? j : 0, jb1 = MINI ? j + 1 : nbg;   // row blocks of the group this pass applies
    const int col0 = MINI ? (kXgGrp * g + j + 1 + 4 * yy) * kXgB : cy * kXgCols + (NCT == 1 ? 64 * (ty % 4) : 0);
    unsigned int live = 0;                                    // bit jj: block jj flipped a row in these 64 replicas
    for (int jj = jb0; jj < jb1; ++jj) live |= (a.flags[((size_t)par * kXgGrp + jj) * ranges + rx] != 0u ? 1u : 0u) << jj;
    live = (unsigned int)__builtin_amdgcn_readfirstlane((int)live);   // (workgroup-uniform: keeps the block index, and with it the
                                                                      // buffer descriptors below, in scalar registers)
    if (live == 0u && !(MINI && j == 0)) return;              // (the first MINI of a group also COPIES F into Tm)
    const bool wave_on = !MINI || 4 * yy + wave < nbg - 1 - j;   // MINI: one wave per later block of the group

    // chunk and sign loads through buffer descriptors with a UNIFORM base per chunk / block (scalar arithmetic) and
    // per-thread offsets that never change -- no 64-bit address arithmetic in vector registers (the kernel lives on 128
    // of them); rows past n are outside the descriptor and read as zero
    f32x4acc pre[NCT];                                        // 16 rows x 64 NCT columns: NCT pieces of 16 bytes per thread
    const int prow = NCT == 4 ? tid >> 6 : tid >> 4, pcol = NCT == 4 ? 4 * (tid & 63) : 4 * (tid & 15);
    const int vq = (int)(((size_t)prow * a.stride + pcol) * 4);
    const int sq = (int)(4 * a.stride * 4);                  // four rows on: the soffset step of a thread's next piece
    auto fetch_chunk = [&](int jj, int c) {                   // rows 16 c .. 16 c + 15 of block kXgGrp g + jj, columns col0 ..
        const int row0 = (kXgGrp * g + jj) * kXgB + CR * c;
        // (readfirstlane: hipcc clamps with a VECTOR med3, and a descriptor held in vector registers costs a waterfall loop per load)
        const int valid = __builtin_amdgcn_readfirstlane(a.n - row0 < 0 ? 0 : (a.n - row0 > CR ? CR : a.n - row0));
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.Q2 + (size_t)row0 * a.stride + col0), 0, (int)((size_t)valid * a.stride * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < NCT; ++i)
            pre[i] = __builtin_bit_cast(f32x4acc, __builtin_amdgcn_raw_buffer_load_b128(rq, vq, i * sq, 0));
    };
    // accumulators: tile (rt, ct): replicas rep0 + 16 rt + 4 lq + reg, column CW wave + 16 ct + lr of the workgroup's 64 NCT
    auto field_ptr = [&](bool dst, int ct, int rt) -> float * {
        const int r = rep0 + 16 * rt + 4 * lq;
        if (MINI) {
            if (!dst && j == 0) return a.F + fidx(a, col0 + CW * wave + 16 * ct + lr, r);
            return a.Tm + tmidx(j + 4 * yy + wave, 16 * ct + lr, r);
        }
        return a.F + fidx(a, col0 + CW * wave + 16 * ct + lr, r);
    };
    f32x4acc acc[4][NCT];
    if (wave_on) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = *reinterpret_cast<const f32x4acc *>(field_ptr(false, ct, rt));
    }
    // The chunks of all live blocks are ONE stream: chunk t + 1 (the next block's first chunk included) and the next
    // block's signs travel from HBM to registers under the MFMAs of chunk t, so that a block boundary costs a barrier
    // and two LDS writes, not a round trip to memory (it did: eight exposed fetches per workgroup, and the four
    // workgroups of a CU pay them in lock-step).
    f32x4acc spre[kXgB * kXgReps / 4 / 256];
    const int vs = ((tid >> 4) * a.Rp + (tid & 15) * 4) * 4;
    auto fetch_signs = [&](int jj) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            a.S + ((size_t)par * kXgGrp + jj) * kXgB * a.Rp + rep0, 0, kXgB * a.Rp * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < kXgB * kXgReps / 4 / 256; ++i)   // rows (tid >> 4) + 16 i
            spre[i] = __builtin_bit_cast(f32x4acc, __builtin_amdgcn_raw_buffer_load_b128(rs, vs, i * 16 * a.Rp * 4, 0));
    };
    unsigned int todo = live & ~((1u << jb0) - 1u) & ((jb1 >= 32) ? ~0u : ((1u << jb1) - 1u));
    if (todo != 0u) {
        int jj = __builtin_ctz(todo);
        fetch_chunk(jj, 0);
        fetch_signs(jj);
#pragma unroll 1
        while (jj >= 0) {
            todo &= todo - 1u;
            const int next = todo != 0u ? (int)__builtin_ctz(todo) : -1;
            __syncthreads();                                   // (Ssl and Apan of the previous block consumed)
#pragma unroll
            for (int i = 0; i < kXgB * kXgReps / 4 / 256; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<f32x4acc *>(&Ssl[e >> 4][(e & 15) * 4]) = spre[i];
            }
#pragma unroll 1
            for (int c = 0; c < kXgB / CR; ++c) {
                if (c > 0) __syncthreads();                    // (the previous chunk is consumed)
#pragma unroll
                for (int i = 0; i < NCT; ++i) *reinterpret_cast<f32x4acc *>(&Apan[4 * i + prow][pcol]) = pre[i];
                if (c + 1 < kXgB / CR) fetch_chunk(jj, c + 1);     // in flight under this chunk's MFMAs
                else if (next >= 0) fetch_chunk(next, 0);
                if (c == 1 && next >= 0) fetch_signs(next);
                __syncthreads();
                if (wave_on) {
#pragma unroll
                    for (int ks = 0; ks < CR / 4; ++ks) {          // k = CR c + 4 ks + lq
                        float sa[4], qb[NCT];
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) sa[rt] = Ssl[CR * c + 4 * ks + lq][16 * rt + lr];        // A[i = lr][k = lq]
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) qb[ct] = Apan[4 * ks + lq][CW * wave + 16 * ct + lr];  // B[k = lq][j = lr]
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                            for (int rt = 0; rt < 4; ++rt)
                                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[rt], qb[ct], acc[rt][ct], 0, 0, 0);
                    }
                }
            }
            jj = next;
        }
    }
    if (wave_on) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) *reinterpret_cast<f32x4acc *>(field_ptr(true, ct, rt)) = acc[rt][ct];
    }
}

// ---- states out; energy E = 1/2 sum_i x_i (f_i + diag_i) from the cached fp32 fields, summed in fp64 (as K1x) ----
__global__ void __launch_bounds__(64) k_xg_finish(XgArgs a)
{
    const int lane = threadIdx.x, r = blockIdx.x;                         // one wavefront per replica
    double e = 0.0;
    for (int b = 0; b < a.nblocks; ++b) {
        const unsigned long long xw = a.XT[(size_t)b * a.Rp + r];
        const int i = b * kXgB + lane;
        const bool x = ((xw >> lane) & 1ull) != 0ull;
        if (i < a.n) {
            a.states[(size_t)r * a.n + i] = x ? 1 : 0;
            if (x) e += 0.5 * ((double)a.F[fidx(a, i, r)] + (double)a.diag[i]);
        }
    }
    e = wave_sum_f64(e);
    if (lane == 0) a.energy[r] = e + a.offset;
}

}  // namespace

size_t mi_dense_xg_workspace_bytes(int n, int R)
{
    const size_t Rp = ((size_t)R + 255) / 256 * 256, ncols = ((size_t)n + kXgCols - 1) / kXgCols * kXgCols;
    const size_t nblocks = ((size_t)n + kXgB - 1) / kXgB;
    return ncols * Rp * 4 + nblocks * Rp * 8 + 2 * kXgGrp * (size_t)kXgB * Rp * 4 + (kXgGrp - 1) * 64 * Rp * 4 + kXgGrp * 64 * Rp * 4 +
           2 * kXgGrp * (Rp / 64) * 4 + 256;
}

// The whole run: (re)initialisation passes and sweeps as K1x orders them; two launches per block of 64 rows.
const float *mi_dense_xg_fields(void *workspace) { return static_cast<const float *>(workspace); }

namespace {
// the run's own streams and events: released on every way out of the launcher (the runtime defers the destruction of
// a stream or event that still has work in flight)
struct XgSync {
    hipStream_t sa = nullptr, sb = nullptr;
    bool own_streams = false;
    hipEvent_t ev[10] = {};
    int nev = 0;
    ~XgSync()
    {
        for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
        if (own_streams) { (void)hipStreamDestroy(sa); (void)hipStreamDestroy(sb); }
    }
    int event(hipEvent_t *out)
    {
        HIP_TRY(hipEventCreateWithFlags(out, hipEventDisableTiming));
        ev[nev++] = *out;
        return MI_OK;
    }
};
}  // namespace

int mi_launch_dense_xg(const DenseXlArgs &x, int chunks, void *workspace, hipStream_t st, int phase)
{
    const bool begin = (phase & 1) != 0, end = (phase & 2) != 0;
    XgArgs a;
    a.Q2 = x.Q2; a.diag = x.diag; a.stride = (size_t)chunks * 4096;
    a.n = x.n; a.R = x.R; a.Rp = (x.R + 255) / 256 * 256;       // (whole DIAG workgroups; the idle seats never accept)
    a.ncols = (x.n + kXgCols - 1) / kXgCols * kXgCols;
    a.nblocks = (x.n + kXgB - 1) / kXgB;
    char *w = static_cast<char *>(workspace);
    a.F = reinterpret_cast<float *>(w);                      w += (size_t)a.ncols * a.Rp * 4;
    a.XT = reinterpret_cast<unsigned long long *>(w);        w += (size_t)a.nblocks * a.Rp * 8;
    a.S = reinterpret_cast<float *>(w);                      w += 2 * kXgGrp * (size_t)kXgB * a.Rp * 4;
    a.Tm = reinterpret_cast<float *>(w);                     w += (kXgGrp - 1) * (size_t)64 * a.Rp * 4;
    a.TH = reinterpret_cast<float *>(w);                     w += (size_t)kXgGrp * 64 * a.Rp * 4;
    a.flags = reinterpret_cast<unsigned int *>(w);
    a.temps = x.temps; a.init = x.init; a.states = x.states; a.energy = x.energy; a.stats = x.stats; a.offset = x.offset;
    a.replica_offset = x.replica_offset; a.seed_lo = x.seed_lo; a.seed_hi = x.seed_hi;
    a.temps_per_replica = x.temps_per_replica;


    if (begin) {
        HIP_TRY(hipMemsetAsync(a.XT, 0, (size_t)a.nblocks * a.Rp * 8, st));     // (replicas past R: no bits)
        hipLaunchKernelGGL(k_xg_init_state, dim3(a.R, (a.nblocks + 3) / 4), dim3(64), 0, st, a);
    }
    const dim3 gdiag(a.Rp / 64), gpanel(a.Rp / kXgReps, a.ncols / kXgCols);
    const dim3 gthr((a.Rp + 255) / 256, 64);
    // Each stream gets its own compute units: the chain's kernels are small (4 .. 16 workgroups) and latency-critical,
    // and behind a full pass that keeps every CU filled from its 3000-workgroup grid they were not scheduled until
    // the pass had drained (measured: a 23 us DIAG took 234 us) -- 8 of the 256 CUs are set aside for them.
    // one chain launch per group (k_xg_chain, a whole CU per 64 replicas) / a DIAG and a small pass per block: the fused
    // kernel wins where the chain bounds the run (few replica ranges; 145 against 154 ms at n = 50 000, 64 replicas x 4
    // sweeps) and is neutral at 1024 replicas, where it would need 16 of the CUs
    const bool fused = x.xg_chain == 2 || (x.xg_chain == 0 && (x.R + 255) / 256 * 256 / kXgReps <= 8);
    note_kernel(fused ? "k_xg_chain + k_xg_panel (K1g, %d blocks of 64 rows in groups of %d)"
                      : "k_xg_diag + k_xg_panel (K1g, %d blocks of 64 rows in groups of %d)", a.nblocks, kXgGrp);
    XgSync sync;
    hipStream_t sa = st, sb = st;                             // (fallback without CU masks: one stream, same order)
    bool own_streams = false;
    {
        int dev = 0, ncu = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        const int words = (ncu + 31) / 32;
        std::vector<uint32_t> ma((size_t)words, 0u), mb((size_t)words, 0u);
        // (the fused chain kernel takes a whole CU per 64 replicas: 156 KB of LDS)
        const int ranges = a.Rp / kXgReps;
        const int chain_cus = fused ? (ranges < 8 ? 8 : (ranges > 32 ? 32 : ranges)) : 8;
        for (int c = 0; c < ncu; ++c) (c < chain_cus ? mb : ma)[(size_t)c / 32] |= 1u << (c % 32);
        hipStream_t ta = nullptr, tb = nullptr;
        const char *one = getenv("MI_XG_ONE_STREAM");       // (tests: the fallback order on the caller's stream)
        if (!(one && one[0] == '1') && ncu >= 64 && hipExtStreamCreateWithCUMask(&ta, (uint32_t)words, ma.data()) == hipSuccess) {
            if (hipExtStreamCreateWithCUMask(&tb, (uint32_t)words, mb.data()) == hipSuccess) {
                sa = ta; sb = tb; own_streams = true;
                sync.sa = ta; sync.sb = tb; sync.own_streams = true;
            } else {
                (void)hipStreamDestroy(ta);
            }
        }
        (void)hipGetLastError();
    }
    hipEvent_t ev_in, ev_out;
    if (int rc = sync.event(&ev_in)) return rc;
    if (int rc = sync.event(&ev_out)) return rc;
    HIP_TRY(hipEventRecord(ev_in, st));                       // everything enqueued on the caller's stream so far
    if (own_streams) { HIP_TRY(hipStreamWaitEvent(sa, ev_in, 0)); HIP_TRY(hipStreamWaitEvent(sb, ev_in, 0)); }
    // Two streams.  B runs the chain of a group (thresholds, DIAG, MINI); A rewrites F (the group's full pass), in two
    // parts: the 256 columns the NEXT group owns go to B, in front of that group's chain (a sixteen-workgroup launch);
    // all others to A, beside that chain.  A group's full pass needs its signs (evS);
    // the S / flag buffers of a parity are free again when the full pass two groups back is through (evP).  Forced
    // (re-initialisation) groups decide nothing and read no fields; the first real group after them waits for all of F.
    hipEvent_t evS[4], evP[4];
    for (int i = 0; i < 4; ++i) {
        if (int rc = sync.event(&evS[i])) return rc;
        if (int rc = sync.event(&evP[i])) return rc;
    }
    const int ngroups = (a.nblocks + kXgGrp - 1) / kXgGrp;
    long G = 0;                                               // running group count
    // the list of passes: (force, sweep index)
    struct Pass { int force; uint32_t sweep; int s_local; };
    std::vector<Pass> passes;
    {
        int until_resync = x.resync > 0 ? 1 : 0;
        for (int s = 0; s < x.num_sweeps; ++s) {
            bool init_now = (s == 0) && begin;                // (a continued run keeps its fields; it has no re-syncs)
            if (x.resync > 0 && --until_resync == 0) { init_now = true; until_resync = x.resync; }
            if (init_now) passes.push_back({1, 0u, 0});
            passes.push_back({0, (uint32_t)s + x.sweep_offset, s});
        }
        if (x.num_sweeps == 0 && begin) passes.push_back({1, 0u, 0});  // energies of the initial states
    }
    bool split_prev = false;                                  // the previous group's full pass was launched in two parts
    for (size_t pi = 0; pi < passes.size(); ++pi) {
        const Pass &ps = passes[pi];
        if (ps.force) {                                       // F = diag, on A, after everything that still reads F
            HIP_TRY(hipEventRecord(evS[(G + 3) & 3], sb));
            HIP_TRY(hipStreamWaitEvent(sa, evS[(G + 3) & 3], 0));
            const size_t cells = (size_t)a.ncols * a.Rp;
            hipLaunchKernelGGL(k_xg_fill_fields, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, sa, a);
        }
        for (int g = 0; g < ngroups; ++g, ++G) {
            const int nbg = a.nblocks - kXgGrp * g < kXgGrp ? a.nblocks - kXgGrp * g : kXgGrp;
            const int par = (int)(G & 1);
            const bool last_of_pass = g + 1 == ngroups;
            const bool has_next = !last_of_pass || pi + 1 < passes.size();
            const int next_force = last_of_pass ? (has_next ? passes[pi + 1].force : 1) : ps.force;
            // ---- stream B: the chain of group G
            if (G >= 2) HIP_TRY(hipStreamWaitEvent(sb, evP[(G - 2) & 3], 0));        // S / flags of this parity are free
            if (!ps.force && G >= 1 && !split_prev)                                   // its own columns of F are final: the
                HIP_TRY(hipStreamWaitEvent(sb, evP[(G - 1) & 3], 0));                 // previous full pass (or its first part, on B)
            if (!ps.force)
                hipLaunchKernelGGL(k_xg_thresholds, dim3(gthr.x, gthr.y, (nbg + 3) / 4), dim3(256), 0, sb, a, g * (kXgGrp / 4), ps.sweep,
                                   ps.s_local);
            if (fused && !ps.force) {
                hipLaunchKernelGGL(k_xg_chain, gdiag, dim3(256), 0, sb, a, g, nbg, par);
            } else {
                for (int j = 0; j < nbg; ++j) {
                    hipLaunchKernelGGL(k_xg_diag, gdiag, dim3(256), 0, sb, a, kXgGrp * g + j, ps.force, par);
                    if (!ps.force && j + 1 < nbg)
                        hipLaunchKernelGGL((k_xg_panel<true, 4>), dim3(a.Rp / kXgReps, (nbg - 1 - j + 3) / 4), dim3(256), 0, sb, a, g, nbg, j, par, -1, -1);
                }
            }
            HIP_TRY(hipEventRecord(evS[G & 3], sb));
            // ---- the group's full pass over F: the next group's own 256 columns on B (in front of that group's chain; they
            // must hold everything up to group G - 1 first), all others on A
            HIP_TRY(hipStreamWaitEvent(sa, evS[G & 3], 0));
            split_prev = has_next && !next_force && ngroups > 1;
            if (split_prev) {
                const int gn = last_of_pass ? 0 : g + 1;
                if (G >= 1) HIP_TRY(hipStreamWaitEvent(sb, evP[(G - 1) & 3], 0));
                const int cyn = gn * (kXgGrp / 4);                 // the next group's columns: kXgGrp / 4 workgroup ranges
                const int ncy = (a.ncols / kXgCols - cyn) < kXgGrp / 4 ? (a.ncols / kXgCols - cyn) : kXgGrp / 4;
                // (64-column workgroups where this part is on the critical path -- few replica ranges, the fused chain; with many
                // ranges it hides behind the full pass and its 8 CUs are better off with a quarter of the workgroups)
                if (fused) hipLaunchKernelGGL((k_xg_panel<false, 1>), dim3(a.Rp / kXgReps, 4 * ncy), dim3(256), 0, sb, a, g, nbg, 0, par, cyn, -1);
                else hipLaunchKernelGGL((k_xg_panel<false, 4>), dim3(a.Rp / kXgReps, ncy), dim3(256), 0, sb, a, g, nbg, 0, par, cyn, -1);
                hipLaunchKernelGGL((k_xg_panel<false, 4>), gpanel, dim3(256), 0, sa, a, g, nbg, 0, par, -1, cyn);
            } else {
                hipLaunchKernelGGL((k_xg_panel<false, 4>), gpanel, dim3(256), 0, sa, a, g, nbg, 0, par, -1, -1);
            }
            HIP_TRY(hipEventRecord(evP[G & 3], sa));
        }
    }
    if (end) hipLaunchKernelGGL(k_xg_finish, dim3(a.R), dim3(64), 0, sa, a);          // (A has waited for the last signs: evS)
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev_out, sa));
    if (own_streams) HIP_TRY(hipStreamWaitEvent(st, ev_out, 0));             // the caller's stream continues after the run
    return MI_OK;
}

}  // namespace mi_sa_impl
