// ubench_mfma.hip -- what the f32-input matrix instructions sustain on the whole chip under its real clock: W waves per
// SIMD, each with four independent accumulators, nothing but MFMAs (operands constant).  Prints TFLOP/s from hipEvents
// and cycles per instruction from s_memtime (100 MHz) -- the achievable ceiling K4 / K1g / K1m are priced against.
//   hipcc -O2 --offload-arch=gfx950 -o scripts/ubench_mfma scripts/ubench_mfma.hip && scripts/ubench_mfma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kIter = 20000;

__global__ void __launch_bounds__(256) k_mfma_32x32x2(float *out, float a, float b)
{
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f32x16{0};
    const float av = a * threadIdx.x, bv = b + threadIdx.x;
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) s += acc[k][q];
    if (s == 12345.678f) out[0] = s;
}

__global__ void __launch_bounds__(256) k_mfma_16x16x4(float *out, float a, float b)
{
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0};
    const float av = a * threadIdx.x, bv = b + threadIdx.x;
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) s += acc[k][q];
    if (s == 12345.678f) out[0] = s;
}

// K4's inner loop without its memory side: operands of every k-step read from LDS (two steps ahead), optionally one
// workgroup barrier per 64 MFMAs
template <int BARRIER, int PREFETCH>
__global__ void __launch_bounds__(256) k_mfma_lds(float *out, int chunks)
{
    __shared__ float As[2][32][128];
    __shared__ float Bs[2][32][128];
    for (int e = threadIdx.x; e < 2 * 32 * 128; e += 256) { (&As[0][0][0])[e] = 1.0f; (&Bs[0][0][0])[e] = 0.5f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const int wi = (wave & 1) * 64, wr = (wave >> 1) * 64;
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f32x16{0};
    for (int c = 0; c < chunks; ++c) {
        const int b = c & 1;
        if (PREFETCH) {
            float a0[3], a1[3], b0[3], b1[3];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                a0[j] = As[b][2 * j + half][wi + col]; a1[j] = As[b][2 * j + half][wi + 32 + col];
                b0[j] = Bs[b][2 * j + half][wr + col]; b1[j] = Bs[b][2 * j + half][wr + 32 + col];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int cur = j % 3, nxt = (j + 2) % 3;
                if (j + 2 < 16) {
                    a0[nxt] = As[b][2 * (j + 2) + half][wi + col]; a1[nxt] = As[b][2 * (j + 2) + half][wi + 32 + col];
                    b0[nxt] = Bs[b][2 * (j + 2) + half][wr + col]; b1[nxt] = Bs[b][2 * (j + 2) + half][wr + 32 + col];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b0[cur], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[cur], b1[cur], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b0[cur], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cur], b1[cur], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float a0 = As[b][2 * j + half][wi + col], a1 = As[b][2 * j + half][wi + 32 + col];
                const float b0 = Bs[b][2 * j + half][wr + col], b1 = Bs[b][2 * j + half][wr + 32 + col];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.0f;
    for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) s += acc[k][q];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv)
{
    const int lds_chunks = argc > 1 ? atoi(argv[1]) : 300;
    float *out;
    CHECK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    for (int variant = 0; variant < 2; ++variant)
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (variant == 0) hipLaunchKernelGGL(k_mfma_32x32x2, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 1.0f, 2.0f);
                else hipLaunchKernelGGL(k_mfma_16x16x4, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 1.0f, 2.0f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double per = variant == 0 ? 4.0 * 4096.0 : 8.0 * 2048.0;          // flop per wave per iteration
            const double flop = per * kIter * 4.0 * cus * wgs_per_cu;
            printf("%s  %d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n",
                   variant == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", wgs_per_cu, best,
                   flop / best / 1e9, flop / best / 1e9 / 157.3);
        }
    for (int variant = 0; variant < 4; ++variant)
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            const int chunks = lds_chunks;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                const dim3 g(cus * wgs_per_cu), t(256);
                if (variant == 0) hipLaunchKernelGGL((k_mfma_lds<0, 0>), g, t, 0, 0, out, chunks);
                if (variant == 1) hipLaunchKernelGGL((k_mfma_lds<0, 1>), g, t, 0, 0, out, chunks);
                if (variant == 2) hipLaunchKernelGGL((k_mfma_lds<1, 0>), g, t, 0, 0, out, chunks);
                if (variant == 3) hipLaunchKernelGGL((k_mfma_lds<1, 1>), g, t, 0, 0, out, chunks);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double flop = 64.0 * 4096.0 * chunks * 4.0 * cus * wgs_per_cu;
            printf("32x32x2 from LDS, barrier %d, prefetch %d, %d workgroup(s) per CU: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n",
                   variant >> 1, variant & 1, wgs_per_cu, best, flop / best / 1e9, flop / best / 1e9 / 157.3);
        }
    return 0;
}
