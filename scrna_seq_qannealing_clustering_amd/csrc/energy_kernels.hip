// energy_kernels.hip -- K4: batched energy evaluation  E_r = x_r^T Qs x_r + offset  (gfx950 only).
//
//   k_energy_dense_valu  one wavefront per state, every fp32 entry added once into fp64 (exact to 1e-15):
//                        for narrow batches.
//   k_energy_dense_mfma  Y = Qs * X on the matrix cores with the f32-INPUT MFMA (v_mfma_f32_32x32x2_f32:
//                        exact f32 products, f32 accumulate == an fmaf chain), LDS-tiled 128 x 128 x 32, fused
//                        with E_r = sum_i X_ir Y_ir.
//                        Used only when the batch is a true dense contraction (R >= 32 states).  Partial
//                        dot products are kept in fp32 for at most 32 terms (+ a 16-way fp32 tree) and then
//                        folded into fp64; the error is ~1e-7 of sum|terms| (tolerance in the tests: 2e-6).
//
// Serves SampleSet energy re-evaluation for the sampler surface (BQM_clustering.py:93-98 prints these
// energies; the "conf" rule :133-146 divides them).
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename QT>
__global__ void __launch_bounds__(256) k_energy_dense_valu(const QT *__restrict__ Qs, int n,
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
        const QT *row = Qs + (size_t)i * n;
        for (int j = lane; j < n; j += 64)
            if (x[j]) e += (double)row[j];
    }
    e = wave_sum_f64(e);
    if (lane == 0) out[r] = e + offset;
}

// X [R][n] (state-major bytes) -> Xt [n][Rpad] (variable-major, zero padded to a multiple of 128 states):
// 64 x 64 byte tiles through LDS, so both the reads (along n) and the writes (along R) are contiguous.
__global__ void __launch_bounds__(256) k_transpose_states(const uint8_t *__restrict__ X, int R, int n,
                                                          uint8_t *__restrict__ Xt, int Rpad)
{
    __shared__ unsigned char t[64][68];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int row = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int r = r0 + row, k = k0 + seg + b;
        t[row][seg + b] = (r < R && k < n) ? X[(size_t)r * n + k] : (uint8_t)0;
    }
    __syncthreads();
    if (k0 + row < n) {
#pragma unroll
        for (int b = 0; b < 16; ++b) Xt[(size_t)(k0 + row) * Rpad + r0 + seg + b] = t[seg + b][row];
    }
}

__global__ void __launch_bounds__(256) k_fill_f64(double *__restrict__ out, int R, double v)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) out[r] = v;
}

// Y = Qs * X as an LDS-tiled f32 GEMM on the matrix cores, fused with the masked reduction E_r = sum_i X_ir Y_ir.
// Workgroup = 4 wavefronts (2 x 2) = a 128 (rows i) x 128 (states r) tile of Y; each wave owns 64 x 64 as 2 x 2
// accumulators of v_mfma_f32_32x32x2_f32.  The k dimension is walked in chunks of 32: the chunk of Qs
// (A[i][k] = Qs[k][i], read as rows: Qs is symmetric) and the chunk of states (bytes widened to f32 once, at
// staging) are double-buffered in LDS as [k][128] floats, so the operand reads are conflict-free
// (lane l reads A[k0 + (l >> 5)][tile + (l & 31)]).  After every chunk the <= 32-term fp32 partial sums are
// folded into fp64 under the state mask of the tile (staged once per workgroup); one fp64 atomic per state
// per workgroup at the end.
// Operand maps of v_mfma_f32_32x32x2_f32:  A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds
// B[k = l >> 5][j = l & 31];  C/D: register q of lane l is C[row = (q & 3) + 8 (q >> 2) + 4 (l >> 5)][col = l & 31].
constexpr int kGemmTile = 128, kGemmKC = 32;
constexpr int kFoldChunks = 4;        // fp32 partial sums run over kFoldChunks * kGemmKC = 128 terms before they are folded into fp64

__global__ void __launch_bounds__(256) k_energy_dense_mfma(const float *__restrict__ Qs, int n,
                                                           const uint8_t *__restrict__ Xt, int R, int Rpad,
                                                           int row_tiles, int ksplit, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float As[2][kGemmKC][kGemmTile];
    __shared__ __attribute__((aligned(16))) float Bs[2][kGemmKC][kGemmTile];
    __shared__ unsigned int Xs[kGemmTile][kGemmTile / 32];        // X[i][r] of this tile as bits (the reduction mask)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // work unit = (tile, k range): E_r is linear in Y, so a tile's k dimension may be cut into `ksplit` pieces
    // that add their partial sums independently -- chosen by the host so that the units fill whole rounds of
    // the chip's 2-per-CU resident workgroups (a 21 x 32 tile grid alone leaves a third of the second round idle)
    const int ks = blockIdx.x % ksplit, tile = blockIdx.x / ksplit;
    const int i0 = (tile % row_tiles) * kGemmTile;
    const int r0 = (tile / row_tiles) * kGemmTile;
    const int wi = (wave & 1) * 64, wr = (wave >> 1) * 64;       // this wave's 64 x 64 corner inside the tile
    const int half = lane >> 5, col = lane & 31;

    // staging of chunk c (k = 32 c .. 32 c + 31) in two halves, so that the global loads of chunk c+1 are in
    // flight WHILE the MFMAs of chunk c run: load_chunk issues them into registers, store_chunk (after the
    // MFMAs) widens the state bytes and writes both operands to the other LDS buffer.
    f32x4 st_a[4];
    unsigned int st_x[4];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;                         // 1024 float4 slots: [k][32 float4]
            const int kk = e >> 5, c4 = (e & 31) * 4;
            const int k = c * kGemmKC + kk;
            f32x4 av = {0.0f, 0.0f, 0.0f, 0.0f};
            unsigned int xw = 0u;
            if (k < n) {
                const float *src = Qs + (size_t)k * n + i0 + c4;
                if (i0 + c4 + 3 < n && (((size_t)k * n + i0 + c4) & 3) == 0) av = *reinterpret_cast<const f32x4 *>(src);
                else {
                    av.x = (i0 + c4 + 0 < n) ? src[0] : 0.0f; av.y = (i0 + c4 + 1 < n) ? src[1] : 0.0f;
                    av.z = (i0 + c4 + 2 < n) ? src[2] : 0.0f; av.w = (i0 + c4 + 3 < n) ? src[3] : 0.0f;
                }
                xw = *reinterpret_cast<const unsigned int *>(Xt + (size_t)k * Rpad + r0 + c4);   // Rpad, r0: multiples of 128
            }
            st_a[q] = av;
            st_x[q] = xw;
        }
    };
    auto store_chunk = [&](int b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            const int kk = e >> 5, c4 = (e & 31) * 4;
            const unsigned int xw = st_x[q];
            *reinterpret_cast<f32x4 *>(&As[b][kk][c4]) = st_a[q];
            f32x4 bv = {(float)(xw & 0xffu), (float)((xw >> 8) & 0xffu), (float)((xw >> 16) & 0xffu), (float)(xw >> 24)};
            *reinterpret_cast<f32x4 *>(&Bs[b][kk][c4]) = bv;
        }
    };
    // the mask tile: bit (rr & 31) of Xs[ii][rr >> 5] = X[i0 + ii][r0 + rr]
    for (int e = tid; e < kGemmTile * (kGemmTile / 32); e += 256) {
        const int ii = e >> 2, w = e & 3;
        unsigned int bits = 0u;
        if (i0 + ii < n) {
            const uint8_t *src = Xt + (size_t)(i0 + ii) * Rpad + r0 + 32 * w;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned int xw = *reinterpret_cast<const unsigned int *>(src + 4 * q);
                bits |= ((xw & 1u) | ((xw >> 7) & 2u) | ((xw >> 14) & 4u) | ((xw >> 21) & 8u)) << (4 * q);
            }
        }
        Xs[ii][w] = bits;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    double e_lo = 0.0, e_hi = 0.0;                               // states wr + col and wr + 32 + col

    const int all_chunks = (n + kGemmKC - 1) / kGemmKC;
    const int c_begin = (int)((long long)all_chunks * ks / ksplit), chunks = (int)((long long)all_chunks * (ks + 1) / ksplit);
    load_chunk(c_begin);
    store_chunk(0);
    __syncthreads();
    for (int c = c_begin; c < chunks; ++c) {
        const int b = (c - c_begin) & 1;
        if (c + 1 < chunks) load_chunk(c + 1);                   // global loads in flight during the MFMAs below
#pragma unroll
        for (int k0 = 0; k0 < kGemmKC; k0 += 2) {
            const float a0 = As[b][k0 + half][wi + col], a1 = As[b][k0 + half][wi + 32 + col];
            const float b0 = Bs[b][k0 + half][wr + col], b1 = Bs[b][k0 + half][wr + 32 + col];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (c + 1 < chunks) store_chunk(b ^ 1);                  // (buffer b^1 was last read before the previous barrier)
        // every kFoldChunks chunks (and at the end) fold the <= 128-term fp32 partial sums under the state mask: the
        // 16 masked values a lane holds per accumulator are first added in fp32 by a fixed balanced tree (4 more
        // roundings on top of the chain), then ONE fp64 add per accumulator -- fp64 instructions are the expensive
        // ones here, and the fold (mask, tree, clearing 64 accumulator registers) is vector work the matrix pipe
        // waits for: folding after every 32-term chunk held the kernel at 0.48 of the f32 MFMA peak
        if (((c - c_begin) % kFoldChunks) == kFoldChunks - 1 || c + 1 == chunks)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float m0[16], m1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wi + 32 * a + (q & 3) + 8 * (q >> 2) + 4 * half;
                m0[q] = ((Xs[row][wr >> 5] >> col) & 1u) ? acc[a][0][q] : 0.0f;
                m1[q] = ((Xs[row][(wr >> 5) + 1] >> col) & 1u) ? acc[a][1][q] : 0.0f;
                acc[a][0][q] = 0.0f;
                acc[a][1][q] = 0.0f;
            }
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int q = 0; q < w; ++q) { m0[q] = m0[q] + m0[q + w]; m1[q] = m1[q] + m1[q + w]; }
            e_lo += (double)m0[0];
            e_hi += (double)m1[0];
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

// dQ: n x n symmetric fp32, dX: R x n bytes, dE: R doubles -- all DEVICE pointers; dXt: scratch of
// n * Rpad bytes (Rpad = R rounded up to 128) or nullptr for the VALU path.  path: 1 = VALU, 2 = MFMA.
int mi_launch_energy_dense(const float *dQ, int n, const uint8_t *dX, int R, double offset, double *dE,
                           uint8_t *dXt, int path, hipStream_t st)
{
    if (path == 2) {
        if (!dXt) return fail(MI_EINVAL, "MFMA energy path needs the transposed-state scratch buffer");
        const int Rpad = ((R + kGemmTile - 1) / kGemmTile) * kGemmTile;
        const int row_tiles = (n + kGemmTile - 1) / kGemmTile;
        hipLaunchKernelGGL(k_transpose_states, dim3(Rpad / 64, (n + 63) / 64), dim3(256), 0, st, dX, R, n, dXt, Rpad);
        hipLaunchKernelGGL(k_fill_f64, dim3((R + 255) / 256), dim3(256), 0, st, dE, R, offset);
        // split the k dimension so that the work units fill whole rounds of the resident workgroups
        // (66 KB of LDS each: two per CU)
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const long long tiles = (long long)row_tiles * (Rpad / kGemmTile), resident = 2LL * (cus > 0 ? cus : 256);
        const int all_chunks = (n + kGemmKC - 1) / kGemmKC;
        int ksplit = 1;
        double best = 0.0;
        for (int k = 1; k <= 4 && k * 8 <= all_chunks; ++k) {
            const long long units = tiles * k, rounds = (units + resident - 1) / resident;
            const double fill = (double)units / (double)(rounds * resident);
            if (fill > best + 0.02) { best = fill; ksplit = k; }
        }
        hipLaunchKernelGGL(k_energy_dense_mfma, dim3((unsigned)(tiles * ksplit)), dim3(256), 0, st, dQ, n, dXt, R, Rpad,
                           row_tiles, ksplit, dE);
    } else {
        hipLaunchKernelGGL(k_energy_dense_valu<float>, dim3((R + 3) / 4), dim3(256), 0, st, dQ, n, dX, R, offset, dE);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// the same sum over an fp64 matrix (the caller's own coefficients): one wavefront per state
int mi_launch_energy_dense_f64(const double *dQ, int n, const uint8_t *dX, int R, double offset, double *dE, hipStream_t st)
{
    hipLaunchKernelGGL(k_energy_dense_valu<double>, dim3((R + 3) / 4), dim3(256), 0, st, dQ, n, dX, R, offset, dE);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace mi_sa_impl
