// probe_valu_under_mfma.hip -- does a wavefront's vector/LDS work make progress while the other wavefronts of its SIMD keep
// the matrix pipe streaming?  One workgroup of 16 waves per CU (4 per SIMD, as K1m).  Wave 0 of the workgroup runs a chain
// of STEPS dependent steps (LDS read -> compare -> select -> quad broadcast -> 4 fma, the shape of K1m's decision steps);
// `mfma_waves` of the other waves of SIMD 0 (waves 4, 8, 12) issue 16x16x4 f32 MFMAs back to back until wave 0 is done.
// Prints the chain's cycles per step against the number of MFMA waves beside it, with and without s_setprio(3).
//   hipcc -O2 --offload-arch=gfx950 -o scripts/probe_valu_under_mfma scripts/probe_valu_under_mfma.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 4096;

// the same question per instruction class: wave 0 runs 4096 x 8 dependent instructions of ONE class (KIND 0 = v_fma_f32,
// 1 = ds_read_b32 whose result is the next address, 2 = s_add_u32 / s_xor_b32) beside `mfma_waves` streaming waves of its SIMD
template <int KIND>
__global__ void __launch_bounds__(1024) k_probe_class(unsigned long long *out, float *sink, int mfma_waves)
{
    __shared__ int ring[1024];
    __shared__ volatile int done;
    for (int e = threadIdx.x; e < 1024; e += 1024) ring[e] = ((e + 17) & 1023) * 4;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        float f = 0.001f * lane;
        int addr = lane * 4;
        unsigned int sc = (unsigned int)__builtin_amdgcn_readfirstlane(lane + 3);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < STEPS; ++k) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(f));
                if (KIND == 1) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(addr) :: "memory");
                if (KIND == 2) asm volatile("s_add_u32 %0, %0, 7\n\ts_xor_b32 %0, %0, 0x55" : "+s"(sc) :: "scc");
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x] = t1 - t0; done = 1; }
        sink[threadIdx.x] = f + (float)addr + (float)sc;
    } else if ((wave & 3) == 0 && (wave >> 2) <= mfma_waves) {
        f32x4 acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0, 0, 0, 0};
        const float av = 0.001f * lane, bv = 1.0f;
        while (!done) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
        }
        float s = 0.0f;
        for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) s += acc[k][q];
        sink[threadIdx.x] = s;
    }
}

// which waves share a SIMD with wave 0: a dependent v_fma chain on wave 0 beside ONE streaming wave `partner`
__global__ void __launch_bounds__(1024) k_probe_partner(unsigned long long *out, float *sink, int partner)
{
    __shared__ volatile int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        float f = 0.001f * lane;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < STEPS; ++k) {
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(f));
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x] = t1 - t0; done = 1; }
        sink[threadIdx.x] = f;
    } else if (wave == partner) {
        f32x4 acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = f32x4{0, 0, 0, 0};
        const float av = 0.001f * lane, bv = 1.0f;
        while (!done) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
        }
        float s = 0.0f;
        for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) s += acc[k][q];
        sink[threadIdx.x] = s;
    }
}

template <int PRIO>
__global__ void __launch_bounds__(1024) k_probe(unsigned long long *out, float *sink, int mfma_waves, int all_simds)
{
    __shared__ float tab[1024];
    __shared__ volatile int done;
    for (int e = threadIdx.x; e < 1024; e += 1024) tab[e] = 0.001f * (e & 15);
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float t[4] = {0.1f * lane, 0.2f, 0.3f, 0.4f};
        const float h = 0.37f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < STEPS; ++k) {
            const f32x4 c = *reinterpret_cast<f32x4 *>(&tab[((k * 4) & 1020)]);
            const bool d = t[0] < h;
            const float so = d ? 1.0f : 0.0f;
            const float sk = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(so), 0, 0xf, 0xf, false));
#pragma unroll
            for (int m = 0; m < 4; ++m) t[m] = __builtin_fmaf(c[m], sk, t[m]);
            t[0] = t[0] - t[1] * 0.5f;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { out[blockIdx.x] = t1 - t0; done = 1; }
        sink[threadIdx.x] = t[0] + t[1] + t[2] + t[3];
    } else {
        const bool on = all_simds ? (wave >= 4 && (wave >> 2) <= mfma_waves) : ((wave & 3) == 0 && (wave >> 2) <= mfma_waves);
        if (on) {
            f32x4 acc[8];
            for (int k = 0; k < 8; ++k) acc[k] = f32x4{0, 0, 0, 0};
            const float av = 0.001f * lane, bv = 1.0f;
            while (!done) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[k], 0, 0, 0);
            }
            float s = 0.0f;
            for (int k = 0; k < 8; ++k) for (int q = 0; q < 4; ++q) s += acc[k][q];
            sink[threadIdx.x] = s;
        }
    }
}

int main()
{
    unsigned long long *d_out; float *d_sink;
    const int wgs = 256;
    CHECK(hipMalloc(&d_out, wgs * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, 1024 * sizeof(float)));
    unsigned long long h[wgs];
    for (int all = 0; all < 2; ++all)
        for (int prio = 0; prio < 2; ++prio)
            for (int mw = 0; mw <= 3; ++mw) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (prio) hipLaunchKernelGGL(k_probe<1>, dim3(wgs), dim3(1024), 0, 0, d_out, d_sink, mw, all);
                    else hipLaunchKernelGGL(k_probe<0>, dim3(wgs), dim3(1024), 0, 0, d_out, d_sink, mw, all);
                    CHECK(hipDeviceSynchronize());
                }
                CHECK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
                double s = 0; for (int i = 0; i < wgs; ++i) s += (double)h[i];
                // (s_memtime ticks: on MI355X one tick is about one shader cycle -- K1m's per-unit tick sums equal its cycles per unit)
                printf("mfma waves beside the chain: %d (%s)  setprio %d   %.1f s_memtime ticks per step\n", mw,
                       all ? "on every SIMD" : "on its SIMD only", prio, s / wgs / STEPS);
            }
    const char *names[3] = {"v_fma_f32 (dependent)", "ds_read_b32 (dependent, waited)", "s_add_u32 + s_xor_b32 (dependent)"};
    for (int kind = 0; kind < 3; ++kind)
        for (int mw = 0; mw <= 3; ++mw) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k_probe_class<0>, dim3(wgs), dim3(1024), 0, 0, d_out, d_sink, mw);
                if (kind == 1) hipLaunchKernelGGL(k_probe_class<1>, dim3(wgs), dim3(1024), 0, 0, d_out, d_sink, mw);
                if (kind == 2) hipLaunchKernelGGL(k_probe_class<2>, dim3(wgs), dim3(1024), 0, 0, d_out, d_sink, mw);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
            double s = 0; for (int i = 0; i < wgs; ++i) s += (double)h[i];
            printf("%-36s beside %d streaming waves of its SIMD: %.1f ticks per instruction%s\n", names[kind], mw,
                   s / wgs / STEPS / 8, kind == 2 ? " pair" : "");
        }
    for (int partner = 1; partner < 16; ++partner) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_probe_partner, dim3(wgs), dim3(1024), 0, 0, d_out, d_sink, partner);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < wgs; ++i) s += (double)h[i];
        printf("v_fma_f32 chain on wave 0 beside ONE streaming wave, wave %2d of the workgroup: %.1f ticks per instruction\n",
               partner, s / wgs / STEPS / 8);
    }
    return 0;
}
