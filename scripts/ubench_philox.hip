// ubench_philox.hip -- what a Philox4x32-10 block, the threshold arithmetic and a rendezvous of a small workgroup cost a
// wavefront that is alone on its SIMD (the few-replica regime: 500 reads on 1024 SIMDs).  Cycles from s_memtime.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_philox scripts/ubench_philox.hip && scripts/ubench_philox
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../scrna_seq_qannealing_clustering_amd/csrc/mi_sa_device.h"
using namespace mi_sa_impl;
namespace mi_sa_impl { int fail(int, const char *, ...) { return -1; } void note_kernel(const char *, ...) {} }

constexpr int kIter = 2000;

template <int STREAMS>
__global__ void __launch_bounds__(64) k_philox(unsigned long long *out, uint32_t seed)
{
    uint32_t w[STREAMS][4];
    for (int q = 0; q < STREAMS; ++q) for (int k = 0; k < 4; ++k) w[q][k] = threadIdx.x + q * 77 + k;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int q = 0; q < STREAMS; ++q) philox4x32_10(w[q][0] + it, w[q][1], w[q][2], 0u, seed, seed ^ 0x55u, w[q]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int q = 0; q < STREAMS; ++q) for (int k = 0; k < 4; ++k) acc ^= w[q][k];
    if (acc == 0x7fffffffu) out[1] = acc;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

__global__ void __launch_bounds__(64) k_neglog(unsigned long long *out, uint32_t seed)
{
    uint32_t r = threadIdx.x * 2654435761u + seed;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) { const float l = neglog_u(r); acc += l; r = r * 1664525u + __float_as_uint(l); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 1.2345f) out[1] = 1;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

// NW waves: write one word, barrier, read NW words -- the exchange of the few-replica kernel
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_exchange(unsigned long long *out, uint32_t seed)
{
    __shared__ int comm[8][4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int acc = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) {
        if (lane == 0) comm[it & 7][w] = acc + it;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int s = 0;
        for (int k = 0; k < NW; ++k) s += __builtin_amdgcn_readfirstlane(comm[it & 7][k]);
        acc = s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x7fffffff) out[1] = acc;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

// where the wavefronts of a 256-thread workgroup land: HW_REG_HW_ID (wave_id [3:0], simd_id [5:4], cu_id [11:8], sh [12], se [15:13])
__global__ void __launch_bounds__(256) k_where(uint32_t *out)
{
    const uint32_t id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, bits 0..31
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = id; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}

// LDS bank structure: 16 ds_read_b32 per iteration, lane l reads dword (l * stride + k * 67) % span_dwords
__global__ void __launch_bounds__(64) k_lds_stride(unsigned long long *out, uint32_t stride)
{
    extern __shared__ char lds[];
    for (int k = threadIdx.x; k < 4096; k += 64) ((uint32_t *)lds)[k] = k;
    __syncthreads();
    uint32_t addr[16], d[16];
    for (int k = 0; k < 16; ++k) { addr[k] = ((threadIdx.x * stride + k * 67) % 4096) * 4; d[k] = 0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(d[k]) : "v"(addr[k]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int k = 0; k < 16; ++k) acc ^= d[k];
    if (acc == 0x7fffffffu) out[1] = acc;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

template <typename K>
static void run(const char *name, K kern, int threads, int blocks, double per)
{
    unsigned long long *d;
    hipMalloc(&d, 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
        hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.1f cycles\n", name, (double)h / blocks / kIter / per);
    hipFree(d);
}

int main()
{
    {
        uint32_t *d, h[64];
        hipMalloc(&d, sizeof h);
        hipLaunchKernelGGL(k_where, dim3(8), dim3(256), 0, 0, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int b = 0; b < 8; ++b) {
            printf("workgroup %d:", b);
            for (int w = 0; w < 4; ++w) { const uint32_t id = h[(b * 4 + w) * 2]; printf("  wave%d xcc %u se %u cu %u simd %u", w, h[(b * 4 + w) * 2 + 1] & 7, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3); }
            printf("\n");
        }
        hipFree(d);
    }
    for (uint32_t stride : {1u, 2u, 3u, 4u, 8u, 16u, 32u, 64u, 128u}) {
        unsigned long long *d;
        hipMalloc(&d, 16);
        for (int blocks : {256, 1024}) {
            hipMemset(d, 0, 16);
            hipLaunchKernelGGL(k_lds_stride, dim3(blocks), dim3(64), 16384 + (blocks == 256 ? 90000 : 20000), 0, d, stride);
            hipDeviceSynchronize();
            unsigned long long h = 0;
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("ds_read_b32 x16, dword stride %3u, %d wave(s)/CU: %6.1f cycles per read (wave time)\n", stride, blocks / 256, (double)h / blocks / kIter / 16);
        }
        hipFree(d);
    }
    run("philox4x32-10, one stream, lone wave", k_philox<1>, 64, 256, 1);
    run("philox4x32-10, two streams interleaved (each)", k_philox<2>, 64, 256, 2);
    run("philox4x32-10, four streams interleaved (each)", k_philox<4>, 64, 256, 4);
    run("philox4x32-10, one stream, 4 waves/SIMD", k_philox<1>, 64, 256 * 16, 1);
    run("neglog_u dependent chain, lone wave", k_neglog, 64, 256, 1);
    run("exchange (write, barrier, read) 2 waves", k_exchange<2>, 128, 256, 1);
    run("exchange (write, barrier, read) 4 waves", k_exchange<4>, 256, 256, 1);
    return 0;
}
