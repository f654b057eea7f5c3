// energy_kernels.hip -- K4: batched energy evaluation  E_r = x_r^T Qs x_r + offset  (gfx950 only).
//
//   k_energy_dense_valu  one wavefront per state, every fp32 entry added once into fp64 (exact to 1e-15):
//                        for narrow batches.
//   k_energy_dense_mfma  Y = Qs * X on the matrix cores with the f32-INPUT MFMA (v_mfma_f32_32x32x2_f32:
//                        exact f32 products, f32 accumulate == an fmaf chain), LDS-tiled 128 x 128 x 32, fused
//                        with E_r = sum_i X_ir Y_ir; only the upper block triangle of the symmetric Qs is multiplied.
//                        Used only when the batch is a true dense contraction (R >= 32 states).  Partial
//                        dot products are kept in fp32 for at most 128 terms (+ a 16-way fp32 tree) and then
//                        folded into fp64; the error is ~1e-7 of sum|terms| (tolerance in the tests: 2e-6).
//
// Serves SampleSet energy re-evaluation for the sampler surface (BQM_clustering.py:93-98 prints these
// energies; the "conf" rule :133-146 divides them).
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename QT>
__global__ void __launch_bounds__(256) k_energy_dense_valu(const QT *__restrict__ Qs, int n, int ld,
                                                           const uint8_t *__restrict__ X, int R,
                                                           double offset, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const uint8_t *x = X + (size_t)r * n;
    double e = 0.0;
    for (int i = 0; i < n; ++i) {
        if (!x[i]) continue;                    // wave-uniform (same address for all lanes)
        const QT *row = Qs + (size_t)i * ld;
        for (int j = lane; j < n; j += 64)
            if (x[j]) e += (double)row[j];
    }
    e = wave_sum_f64(e);
    if (lane == 0) out[r] = e + offset;
}

// X [R][n] (state-major bytes) -> Xt [n_pad][Rpad] (variable-major, zero padded to multiples of 128 both ways) and the same
// states as BITS, Xm [n_pad][Rpad / 32] (bit r & 31 of word r >> 5): 64 x 64 byte tiles through LDS, so both the reads
// (along n) and the writes (along R) are contiguous.  The grid covers the padded extent (the padding is written, not
// cleared beforehand), and the first row of workgroups starts the energies at the model's offset.
__global__ void __launch_bounds__(256) k_transpose_states(const uint8_t *__restrict__ X, int R, int n,
                                                          uint8_t *__restrict__ Xt, unsigned int *__restrict__ Xm, int Rpad,
                                                          double offset, double *__restrict__ out)
{
    __shared__ unsigned char t[64][68];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int row = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int r = r0 + row, k = k0 + seg + b;
        t[row][seg + b] = (r < R && k < n) ? X[(size_t)r * n + k] : (uint8_t)0;
    }
    if (blockIdx.y == 0 && threadIdx.x < 64 && r0 + (int)threadIdx.x < R) out[r0 + threadIdx.x] = offset;
    __syncthreads();
    unsigned int bits = 0u;
#pragma unroll
    for (int b = 0; b < 16; ++b) bits |= (unsigned int)(t[seg + b][row] & 1u) << b;
    const unsigned int other = __shfl_xor(bits, 1, 64);           // the neighbouring 16 states of the same variable
#pragma unroll
    for (int b = 0; b < 16; ++b) Xt[(size_t)(k0 + row) * Rpad + r0 + seg + b] = t[seg + b][row];
    if ((threadIdx.x & 1) == 0) Xm[(size_t)(k0 + row) * (Rpad / 32) + (r0 + seg) / 32] = bits | (other << 16);
}

// E_r = sum_{i,k} X_ir Qs_ik X_kr as an LDS-tiled f32 GEMM on the matrix cores, Y = Qs * X, fused with the masked
// reduction sum_i X_ir Y_ir.  Qs is SYMMETRIC, so only the 128 x 128 blocks (I, K) with K >= I are multiplied:
// an off-diagonal block counts twice (an exact doubling of its partial sum).  Work unit = a contiguous piece of
// the row-major list of those blocks, for one tile of 128 states; E_r is linear in Y, so the pieces add
// independently (one fp64 atomic per state and unit) and are sized by the host to fill the chip evenly.
// Workgroup = 4 wavefronts (2 x 2), each 64 x 64 of the block as 2 x 2 accumulators of v_mfma_f32_32x32x2_f32
// (exact f32 products, f32 accumulate == an fmaf chain).  A block is walked in 4 chunks of 32 k: the chunk of Qs
// (A[i][k] = Qs[k][i], read as rows) and of the states (bytes widened to f32 at staging) sit double-buffered in LDS
// as [k][128] floats, operand reads conflict-free (lane l reads A[k0 + (l >> 5)][tile + (l & 31)]).
// Everything that is not an MFMA rides in the shadow of the MFMAs of the same wave, pinned there by scheduling
// barriers: operand reads two k-steps ahead; the LDS stores of chunk t + 1 in k-steps 0-3 of chunk t; the global loads
// of chunk t + 2 in k-steps 4-7.  (Left to the compiler these were phases of their own, and the two workgroups of a CU
// ran them in lock-step: 0.53 of the MFMA rate where the bare loop reaches 0.92, scripts/ubench_mfma.hip.)
// After every block the <= 128-term fp32 partial sums are folded into fp64 under the state mask of the row tile
// (bit tile Xs, prefetched one block ahead when the next block starts a new row).  Inputs are padded: Qp has
// n_pad = 128 * ceil(n / 128) rows of n_pad floats, Xt / Xm have n_pad rows, all padding zero.
// Operand maps of v_mfma_f32_32x32x2_f32:  A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds
// B[k = l >> 5][j = l & 31];  C/D: register q of lane l is C[row = (q & 3) + 8 (q >> 2) + 4 (l >> 5)][col = l & 31].
constexpr int kGemmTile = 128, kGemmKC = 32;

__global__ void __launch_bounds__(256, 2) k_energy_dense_mfma(const float *__restrict__ Qp, int ldq,
                                                              const uint8_t *__restrict__ Xt,
                                                              const unsigned int *__restrict__ Xm, int R, int Rpad,
                                                              int T, int pieces, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float As[2][kGemmKC][kGemmTile];
    __shared__ __attribute__((aligned(16))) float Bs[2][kGemmKC][kGemmTile];
    __shared__ unsigned int Xs[2][kGemmTile][kGemmTile / 32];     // X[i][r] of a row tile as bits (the reduction mask)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // unit -> (piece, state tile): every XCD (workgroup id mod 8) takes a contiguous range of the piece-major unit
    // list, so the workgroups that read the same blocks of Qs share an L2
    const int U = gridDim.x;
    const int w = (U % 8 == 0) ? (int)(blockIdx.x % 8) * (U / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int rtiles = Rpad / kGemmTile;
    const int piece = w / rtiles, r0 = (w % rtiles) * kGemmTile;
    const int S = T * (T + 1) / 2;
    const int s0 = (int)((long long)S * piece / pieces), s1 = (int)((long long)S * (piece + 1) / pieces);
    if (s0 >= s1) return;
    const int NT = 4 * (s1 - s0);
    const int wi = (wave & 1) * 64, wr = (wave >> 1) * 64;       // this wave's 64 x 64 corner inside the block
    const int half = lane >> 5, col = lane & 31;
    const int W = Rpad / 32;

    int cI = 0, cK = 0;                                          // block s0: row I holds the blocks K = I .. T - 1
    {
        int s = s0;
        while (s >= T - cI) { s -= T - cI; ++cI; }
        cK = cI + s;
    }
    int lI = cI, lK = cK, lcc = 0;                               // the loader's position in the chunk stream
    const int kk = tid >> 5, c4 = (tid & 31) * 4;
    // chunk loads through buffer descriptors rebuilt per chunk from a UNIFORM base (scalar arithmetic) with per-thread
    // offsets that never change: no address VGPRs are recomputed inside the loop (hipcc reused the registers of loads
    // still in flight for them, and waited for those loads)
    const int va = (kk * ldq + c4) * 4, vx = kk * Rpad + c4;
    __amdgpu_buffer_rsrc_t rq, rx;
    f32x4 st_a[4];
    unsigned int st_x[4];
    auto loader_begin = [&]() {                                  // descriptors of the loader's chunk, then advance it
        const size_t krow = (size_t)(lK * kGemmTile + lcc * kGemmKC);
        rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Qp + krow * ldq + (size_t)lI * kGemmTile), 0,
                                               kGemmKC * ldq * 4, 0x00020000);
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(Xt + krow * Rpad + r0), 0, kGemmKC * Rpad, 0x00020000);
        if (++lcc == 4) { lcc = 0; if (++lK == T) { ++lI; lK = lI; } }
    };
    auto load_piece = [&](int q) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rq, va, q * 8 * ldq * 4, 0);
        st_a[q] = __builtin_bit_cast(f32x4, v);
        st_x[q] = __builtin_amdgcn_raw_buffer_load_b32(rx, vx, q * 8 * Rpad, 0);
    };
    auto store_piece = [&](int b, int q) {
        const unsigned int xw = st_x[q];
        *reinterpret_cast<f32x4 *>(&As[b][kk + 8 * q][c4]) = st_a[q];
        f32x4 bv = {(float)(xw & 0xffu), (float)((xw >> 8) & 0xffu), (float)((xw >> 16) & 0xffu), (float)(xw >> 24)};
        *reinterpret_cast<f32x4 *>(&Bs[b][kk + 8 * q][c4]) = bv;
    };
    unsigned int mw[2];
    auto load_mask = [&](int I) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q;
            mw[q] = Xm[(size_t)(I * kGemmTile + (e >> 2)) * W + r0 / 32 + (e & 3)];
        }
    };
    auto store_mask = [&](int mb) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int e = tid + 256 * q; Xs[mb][e >> 2][e & 3] = mw[q]; }
    };

    // prologue: chunk 0 into buffer 0, chunk 1 into registers, the mask of the first row
    loader_begin();
#pragma unroll
    for (int q = 0; q < 4; ++q) load_piece(q);
    load_mask(cI);
#pragma unroll
    for (int q = 0; q < 4; ++q) store_piece(0, q);
    store_mask(0);
    if (NT > 1) {
        loader_begin();
#pragma unroll
        for (int q = 0; q < 4; ++q) load_piece(q);
    }
    __syncthreads();

    f32x16 acc[2][2];
    double e_lo = 0.0, e_hi = 0.0;                               // states wr + col and wr + 32 + col
    int mb = 0, ccc = 0;
    bool mask_pending = false;
    for (int t = 0; t < NT; ++t) {
        const int b = t & 1;
        if (ccc == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) acc[a][bb] = f32x16{0};
        }
        const bool more1 = t + 1 < NT, more2 = t + 2 < NT;
        float a0[3], a1[3], b0[3], b1[3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a0[j] = As[b][2 * j + half][wi + col]; a1[j] = As[b][2 * j + half][wi + 32 + col];
            b0[j] = Bs[b][2 * j + half][wr + col]; b1[j] = Bs[b][2 * j + half][wr + 32 + col];
        }
#pragma unroll
        for (int j = 0; j < kGemmKC / 2; ++j) {
            const int cur = j % 3, nxt = (j + 2) % 3;
            if (j + 2 < kGemmKC / 2) {
                a0[nxt] = As[b][2 * (j + 2) + half][wi + col]; a1[nxt] = As[b][2 * (j + 2) + half][wi + 32 + col];
                b0[nxt] = Bs[b][2 * (j + 2) + half][wr + col]; b1[nxt] = Bs[b][2 * (j + 2) + half][wr + 32 + col];
            }
            if (j < 4) { if (more1) store_piece(b ^ 1, j); }
            else if (j < 8) {
                if (more2) { if (j == 4) loader_begin(); load_piece(j - 4); }
            }
            // the mask of the next row tile, one block ahead: loaded in the first chunk of the last block of a row, stored
            // in its third, both right after the chunk stores (no chunk load is in flight there, so neither waits for one);
            // the other mask buffer was last read by the folds of the row before this one
            if (j == 3 && ccc == 0 && cK == T - 1 && t + 4 < NT) { load_mask(cI + 1); mask_pending = true; }
            if (j == 3 && ccc == 2 && mask_pending) { store_mask(mb ^ 1); mask_pending = false; }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b0[cur], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b1[cur], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b0[cur], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b1[cur], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++ccc == 4) {
            // end of a block: the <= 128-term fp32 partial sums under the state mask -- the 16 masked values a lane holds
            // per accumulator are added in fp32 by a fixed balanced tree (4 more roundings on top of the chain), then ONE
            // fp64 add per accumulator; an off-diagonal block stands for its mirror image too
            ccc = 0;
            const double weight = (cK == cI) ? 1.0 : 2.0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float m0[16], m1[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = wi + 32 * a + (q & 3) + 8 * (q >> 2) + 4 * half;
                    m0[q] = ((Xs[mb][row][wr >> 5] >> col) & 1u) ? acc[a][0][q] : 0.0f;
                    m1[q] = ((Xs[mb][row][(wr >> 5) + 1] >> col) & 1u) ? acc[a][1][q] : 0.0f;
                }
#pragma unroll
                for (int wd = 8; wd >= 1; wd >>= 1)
#pragma unroll
                    for (int q = 0; q < wd; ++q) { m0[q] = m0[q] + m0[q + wd]; m1[q] = m1[q] + m1[q + wd]; }
                e_lo += weight * (double)m0[0];
                e_hi += weight * (double)m1[0];
            }
            if (++cK == T) { ++cI; cK = cI; mb ^= 1; }
        }
        __syncthreads();
    }
    e_lo += __shfl_xor(e_lo, 32, 64);
    e_hi += __shfl_xor(e_hi, 32, 64);
    if (half == 0) {
        if (r0 + wr + col < R) atomicAdd(&out[r0 + wr + col], e_lo);
        if (r0 + wr + 32 + col < R) atomicAdd(&out[r0 + wr + 32 + col], e_hi);
    }
}

}  // namespace

// scratch of the MFMA path behind dXt: n_pad * Rpad bytes of transposed states + n_pad * Rpad / 8 bytes of state bits
size_t mi_energy_dense_scratch_bytes(int n, int R)
{
    const size_t n_pad = ((size_t)n + kGemmTile - 1) / kGemmTile * kGemmTile, Rpad = ((size_t)R + kGemmTile - 1) / kGemmTile * kGemmTile;
    return n_pad * Rpad + n_pad * Rpad / 8;
}

// dQ: symmetric fp32, n rows of ldq floats; dX: R x n bytes, dE: R doubles -- all DEVICE pointers.  path 1 = VALU (any
// ldq >= n), 2 = MFMA: ldq = n_pad = n rounded up to 128, dQ holds n_pad rows, padding ZERO; dXt: scratch of
// mi_energy_dense_scratch_bytes(n, R).
int mi_launch_energy_dense(const float *dQ, int n, int ldq, const uint8_t *dX, int R, double offset, double *dE,
                           uint8_t *dXt, int path, hipStream_t st)
{
    if (path == 2) {
        if (!dXt) return fail(MI_EINVAL, "MFMA energy path needs the transposed-state scratch buffer");
        const int Rpad = ((R + kGemmTile - 1) / kGemmTile) * kGemmTile;
        const int T = (n + kGemmTile - 1) / kGemmTile;
        if (ldq != T * kGemmTile) return fail(MI_EINVAL, "MFMA energy path needs rows padded to %d floats", T * kGemmTile);
        unsigned int *dXm = reinterpret_cast<unsigned int *>(dXt + (size_t)ldq * Rpad);
        hipLaunchKernelGGL(k_transpose_states, dim3(Rpad / 64, ldq / 64), dim3(256), 0, st, dX, R, n, dXt, dXm, Rpad, offset, dE);
        // pieces per state tile: the units should fill whole rounds of the resident workgroups (68 KB of LDS each: two
        // per CU) with pieces of (nearly) equal length
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const long long rtiles = Rpad / kGemmTile, resident = 2LL * (cus > 0 ? cus : 256);
        const long long S = (long long)T * (T + 1) / 2;
        long long pieces = 1;
        double best = 0.0;
        for (long long rounds = 1; rounds <= 4; ++rounds) {
            long long p = rounds * resident / rtiles;
            if (p < 1) p = 1;
            if (p > S) p = S;
            const long long units = p * rtiles, longest = (S + p - 1) / p;
            const double fill = (double)(S * rtiles) / (double)(((units + resident - 1) / resident) * resident * longest);
            if (fill > best + 0.02) { best = fill; pieces = p; }
            if (p == S) break;
        }
        hipLaunchKernelGGL(k_energy_dense_mfma, dim3((unsigned)(pieces * rtiles)), dim3(256), 0, st, dQ, ldq, dXt, dXm, R, Rpad,
                           T, (int)pieces, dE);
    } else {
        hipLaunchKernelGGL(k_energy_dense_valu<float>, dim3((R + 3) / 4), dim3(256), 0, st, dQ, n, ldq, dX, R, offset, dE);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the same sum over an fp64 matrix (the caller's own coefficients): one wavefront per state
int mi_launch_energy_dense_f64(const double *dQ, int n, const uint8_t *dX, int R, double offset, double *dE, hipStream_t st)
{
    hipLaunchKernelGGL(k_energy_dense_valu<double>, dim3((R + 3) / 4), dim3(256), 0, st, dQ, n, n, dX, R, offset, dE);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace mi_sa_impl
