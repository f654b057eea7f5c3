// energy_kernels.hip -- K4: batched energy evaluation  E_r = x_r^T Qs x_r + offset  (gfx950 only).
//
//   k_energy_dense_valu  one wavefront per state, every fp32 entry added once into fp64 (exact to 1e-15):
//                        for narrow batches.
//   k_energy_dense_mfma  Y = Qs * X on the matrix cores with the f32-INPUT MFMA (v_mfma_f32_32x32x2_f32:
//                        exact f32 products, f32 accumulate == an fmaf chain), then E_r = sum_i X_ir Y_ir.
//                        Used only when the batch is a true dense contraction (R >= 32 states).  Partial
//                        dot products are kept in fp32 for at most 32 terms and then folded into fp64; the
//                        error is ~1e-7 of sum|terms| (tolerance in the tests: 2e-6 of the energy).
//
// Serves SampleSet energy re-evaluation for the sampler surface (BQM_clustering.py:93-98 prints these
// energies; the "conf" rule :133-146 divides them).
#include "mi_sa_device.h"

namespace mi_sa_impl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_energy_dense_valu(const float *__restrict__ Qs, int n,
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
        const float *row = Qs + (size_t)i * n;
        for (int j = lane; j < n; j += 64)
            if (x[j]) e += (double)row[j];
    }
    e = wave_sum_f64(e);
    if (lane == 0) out[r] = e + offset;
}

// X [R][n] (state-major bytes) -> Xt [n][Rpad] (variable-major, zero padded to a multiple of 64 states)
__global__ void __launch_bounds__(256) k_transpose_states(const uint8_t *__restrict__ X, int R, int n,
                                                          uint8_t *__restrict__ Xt, int Rpad)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (r < Rpad) Xt[(size_t)k * Rpad + r] = (r < R) ? X[(size_t)r * n + k] : (uint8_t)0;
}

__global__ void __launch_bounds__(256) k_fill_f64(double *__restrict__ out, int R, double v)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < R) out[r] = v;
}

// One wavefront = one 32-row tile of Y for TWO adjacent 32-state tiles (the A operand, a column block
// of the symmetric Qs read as rows, is shared by both).  Operand maps of v_mfma_f32_32x32x2_f32:
//   A: lane l holds A[i = l & 31][k = l >> 5];  B: lane l holds B[k = l >> 5][j = l & 31];
//   C/D: register q of lane l is C[row = (q & 3) + 8 (q >> 2) + 4 (l >> 5)][col = l & 31].
__global__ void __launch_bounds__(256) k_energy_dense_mfma(const float *__restrict__ Qs, int n,
                                                           const uint8_t *__restrict__ Xt, int R, int Rpad,
                                                           int row_tiles, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pair_tiles = Rpad / 64;
    if (tile >= row_tiles * pair_tiles) return;            // wave-uniform
    const int i0 = (tile % row_tiles) * 32;
    const int r0 = (tile / row_tiles) * 64;
    const int half = lane >> 5, col = lane & 31;
    const bool row_ok = (i0 + col) < n;                     // the A row this lane feeds
    f32x16 acc0 = {0}, acc1 = {0};
    double e0 = 0.0, e1 = 0.0;

    auto flush = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = i0 + (q & 3) + 8 * (q >> 2) + 4 * half;
            if (row < n) {
                const uint8_t *xr = Xt + (size_t)row * Rpad + r0 + col;
                if (xr[0]) e0 += (double)acc0[q];
                if (xr[32]) e1 += (double)acc1[q];
            }
            acc0[q] = 0.0f;
            acc1[q] = 0.0f;
        }
    };

    int since_flush = 0;
#pragma unroll 4
    for (int k0 = 0; k0 < n; k0 += 2) {
        const int k = k0 + half;
        float a = 0.0f, b0 = 0.0f, b1 = 0.0f;
        if (k < n) {
            if (row_ok) a = Qs[(size_t)k * n + i0 + col];   // A[i][k] = Qs[k][i]: coalesced 128 B per half-wave
            const uint8_t *xk = Xt + (size_t)k * Rpad + r0 + col;
            b0 = (float)xk[0];
            b1 = (float)xk[32];
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        if (++since_flush == 16) { flush(); since_flush = 0; }    // <= 32 fp32 terms per partial sum
    }
    flush();
    e0 += __shfl_xor(e0, 32, 64);
    e1 += __shfl_xor(e1, 32, 64);
    if (half == 0) {
        if (r0 + col < R) atomicAdd(&out[r0 + col], e0);
        if (r0 + 32 + col < R) atomicAdd(&out[r0 + 32 + col], e1);
    }
}

}  // namespace

// dQ: n x n symmetric fp32, dX: R x n bytes, dE: R doubles -- all DEVICE pointers; dXt: scratch of
// n * Rpad bytes (Rpad = R rounded up to 64) or nullptr for the VALU path.  path: 1 = VALU, 2 = MFMA.
int mi_launch_energy_dense(const float *dQ, int n, const uint8_t *dX, int R, double offset, double *dE,
                           uint8_t *dXt, int path, hipStream_t st)
{
    if (path == 2) {
        if (!dXt) return fail(MI_EINVAL, "MFMA energy path needs the transposed-state scratch buffer");
        const int Rpad = ((R + 63) / 64) * 64;
        const int row_tiles = (n + 31) / 32;
        hipLaunchKernelGGL(k_transpose_states, dim3((Rpad + 255) / 256, n), dim3(256), 0, st, dX, R, n, dXt, Rpad);
        hipLaunchKernelGGL(k_fill_f64, dim3((R + 255) / 256), dim3(256), 0, st, dE, R, offset);
        const int tiles = row_tiles * (Rpad / 64);
        hipLaunchKernelGGL(k_energy_dense_mfma, dim3((tiles + 3) / 4), dim3(256), 0, st, dQ, n, dXt, R, Rpad,
                           row_tiles, dE);
    } else {
        hipLaunchKernelGGL(k_energy_dense_valu, dim3((R + 3) / 4), dim3(256), 0, st, dQ, n, dX, R, offset, dE);
    }
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

}  // namespace mi_sa_impl
