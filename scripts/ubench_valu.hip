// ubench_valu.hip -- issue cost of the vector instructions the anneal kernels are made of, measured the way
// the kernels run them: W wavefronts per SIMD (64-thread workgroups, W*4 per CU), independent instructions,
// cycles from s_memtime inside the wave.  Prints SIMD cycles per wave-instruction (= wave cycles / W).
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/ubench_valu scripts/ubench_valu.hip && /tmp/ubench_valu
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIter = 2000;

#define R8(X, a) X(a##0) X(a##1) X(a##2) X(a##3) X(a##4) X(a##5) X(a##6) X(a##7)
// 16 independent destinations v[d0..d15]; BODY is the instruction text with %0 = dst (also src), %1 %2 = other sources
#define KERNEL(NAME, TEXT)                                                                                   \
    __global__ void __launch_bounds__(64) NAME(unsigned long long *out, uint32_t seed)                       \
    {                                                                                                        \
        uint32_t d[16];                                                                                      \
        for (int k = 0; k < 16; ++k) d[k] = seed * (threadIdx.x + 1) + k * 0x9E3779B9u;                      \
        uint32_t a = seed ^ 0x12345u, b = threadIdx.x | 0x3f800000u;                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < kIter; ++it) {                                                                 \
            _Pragma("unroll") for (int k = 0; k < 16; ++k)                                                   \
                asm volatile(TEXT : "+v"(d[k]) : "v"(a), "v"(b), "s"(seed));                                 \
        }                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        uint32_t acc = 0;                                                                                    \
        for (int k = 0; k < 16; ++k) acc ^= d[k];                                                            \
        if (acc == 0x7fffffffu) out[1] = acc;                                                                \
        if (threadIdx.x == 0) atomicAdd(out, t1 - t0);                                                       \
    }

KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_mul_hi_s, "v_mul_hi_u32 %0, %0, %3")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %1, %2, %0")
KERNEL(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(k_cvt_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %3, %0")
KERNEL(k_bfe, "v_bfe_i32 %0, %0, %1, 1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %2, %1, %0 op_sel_hi:[0,1,0]")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_exp, "v_exp_f32 %0, %0")
KERNEL(k_log, "v_log_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")

// 64-bit destination forms
__global__ void __launch_bounds__(64) k_mad_u64(unsigned long long *out, uint32_t seed)
{
    unsigned long long d[8];
    for (int k = 0; k < 8; ++k) d[k] = seed * (threadIdx.x + 1) + k * 0x9E3779B9u;
    uint32_t a = seed ^ 0x12345u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(d[k]) : "v"(a), "v"((uint32_t)d[k]) : "vcc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long acc = 0;
    for (int k = 0; k < 8; ++k) acc ^= d[k];
    if (acc == 0x7fffffffu) out[1] = acc;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

__global__ void __launch_bounds__(64) k_pk_fma2(unsigned long long *out, uint32_t seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 d[8];
    for (int k = 0; k < 8; ++k) d[k] = f2{(float)(threadIdx.x + k), 1.0f};
    f2 a = {1.0001f, 0.9999f}, b = {0.5f, 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[k]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int k = 0; k < 8; ++k) acc += d[k].x + d[k].y;
    if (acc == 1.2345f) out[1] = 1;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

// LDS gathers: 16 independent reads per iteration at pseudo-random addresses inside `span` bytes
template <int BYTES>
__global__ void __launch_bounds__(64) k_lds_gather(unsigned long long *out, uint32_t seed, int span, int random)
{
    extern __shared__ char lds[];
    for (int k = threadIdx.x; k < span / 4 + 1; k += 64) ((uint32_t *)lds)[k] = k;
    __syncthreads();
    uint32_t addr[16], d[16];
    for (int k = 0; k < 16; ++k) {
        uint32_t h = (threadIdx.x * 2654435761u + k * 40503u + seed) * 2246822519u;
        h ^= h >> 15;
        addr[k] = random ? ((h % (uint32_t)span) & ~(uint32_t)(BYTES - 1)) : ((threadIdx.x * BYTES + k * 64 * BYTES) % span);
        d[k] = 0;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (BYTES == 1) asm volatile("ds_read_u8 %0, %1" : "=v"(d[k]) : "v"(addr[k]));
            else if (BYTES == 2) asm volatile("ds_read_u16 %0, %1" : "=v"(d[k]) : "v"(addr[k]));
            else asm volatile("ds_read_b32 %0, %1" : "=v"(d[k]) : "v"(addr[k]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    for (int k = 0; k < 16; ++k) acc ^= d[k];
    if (acc == 0x7fffffffu) out[1] = acc;
    if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
}

template <typename K, typename... A>
static int run(const char *name, K kern, int waves_per_simd, int inst_per_iter, size_t lds, A... args)
{
    unsigned long long *d_out;
    CHECK(hipMalloc(&d_out, 16));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * 4 * waves_per_simd;
    std::vector<double> res;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(d_out, 0, 16));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d_out, args...);
        CHECK(hipDeviceSynchronize());
        unsigned long long h = 0;
        CHECK(hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost));
        res.push_back((double)h / blocks / ((double)kIter * inst_per_iter) / waves_per_simd);
    }
    std::sort(res.begin(), res.end());
    printf("%-28s W=%d  %.2f SIMD-cycles per wave-instruction (min %.2f)\n", name, waves_per_simd, res[1], res[0]);
    CHECK(hipFree(d_out));
    return 0;
}

int main()
{
    for (int W : {1, 2, 3, 4, 8}) {
        const size_t lds = 160 * 1024 / (4 * W) - 64;     // forces exactly W waves per SIMD... at most
        run("v_xor_b32", k_xor, W, 16, lds, 12345u);
        run("v_add_u32", k_add_u32, W, 16, lds, 12345u);
        run("v_fma_f32", k_fma, W, 16, lds, 12345u);
        run("v_fma_mix_f32", k_fma_mix, W, 16, lds, 12345u);
        run("v_pk_fma_f32", k_pk_fma2, W, 16, lds, 12345u);
        run("v_mul_lo_u32", k_mul_lo, W, 16, lds, 12345u);
        run("v_mul_hi_u32", k_mul_hi, W, 16, lds, 12345u);
        run("v_mul_hi_u32 (sgpr)", k_mul_hi_s, W, 16, lds, 12345u);
        run("v_mad_u64_u32", k_mad_u64, W, 16, lds, 12345u);
        run("v_mul_u32_u24", k_mul_u24, W, 16, lds, 12345u);
        run("v_mad_i32_i24", k_mad_i24, W, 16, lds, 12345u);
        run("v_cvt_f32_ubyte0", k_cvt_ub0, W, 16, lds, 12345u);
        run("v_cvt_f32_i32", k_cvt_i32, W, 16, lds, 12345u);
        run("v_mbcnt_lo_u32_b32", k_mbcnt, W, 16, lds, 12345u);
        run("v_bfe_i32", k_bfe, W, 16, lds, 12345u);
        run("v_cndmask_b32", k_cndmask, W, 16, lds, 12345u);
        run("v_lshl_add_u32", k_lshl_add, W, 16, lds, 12345u);
        run("v_xad_u32", k_xad, W, 16, lds, 12345u);
        run("v_cmp_lt_f32", k_cmp, W, 16, lds, 12345u);
        run("v_exp_f32", k_exp, W, 16, lds, 12345u);
        run("v_log_f32", k_log, W, 16, lds, 12345u);
        run("v_rcp_f32", k_rcp, W, 16, lds, 12345u);
        for (int random : {0, 1}) {
            if (W == 8 || W == 2 || W == 3) break;        // (8: 5 KB of LDS per wave: the gather spans do not fit)
            char nm[64];
            snprintf(nm, sizeof nm, "ds_read_u8  %s 2688 B", random ? "random" : "linear");
            run(nm, k_lds_gather<1>, W, 16, lds, 777u, 2688, random);
            snprintf(nm, sizeof nm, "ds_read_u16 %s 5376 B", random ? "random" : "linear");
            run(nm, k_lds_gather<2>, W, 16, lds, 777u, 5376, random);
            snprintf(nm, sizeof nm, "ds_read_b32 %s 336 B", random ? "random" : "linear");
            run(nm, k_lds_gather<4>, W, 16, lds, 777u, 336, random);
        }
    }
    return 0;
}
