// probe_placement.hip -- where do the wavefronts of small workgroups land?  For a grid that fills the chip in one round
// (as the pair kernel's does) every wavefront records (XCC, SE, CU, SIMD, wave slot) from HW_ID / XCC_ID; the host prints
// how the waves of a workgroup spread over SIMDs and how many wave-0 / wave-1 each SIMD of a CU hosts.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/probe scripts/probe_placement.hip && /tmp/probe 128 2048 16384
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <array>

__global__ void probe(uint32_t *out, int spin)
{
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));     // XCC_ID
    // keep the wave resident for a while so that the whole grid is on the chip together
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * nw + wave) * 2] = hw;
        out[(blockIdx.x * nw + wave) * 2 + 1] = xcc;
        lds[wave] = 1;
    }
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 128, blocks = argc > 2 ? atoi(argv[2]) : 2048, ldsb = argc > 3 ? atoi(argv[3]) : 16384;
    const int nw = threads / 64;
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * nw * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), ldsb, 0, d, 2000000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h((size_t)blocks * nw * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13 (gfx90a+: 3 bits), ...
    std::map<uint32_t, std::array<int, 16>> per_cu;   // key (xcc, se, sh, cu) -> count[simd * 4 + wave-in-wg (<4)]
    std::map<std::string, int> patterns;
    for (int b = 0; b < blocks; ++b) {
        std::string pat;
        for (int w = 0; w < nw; ++w) {
            const uint32_t hw = h[(b * nw + w) * 2], xcc = h[(b * nw + w) * 2 + 1] & 15u;
            const uint32_t simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            const uint32_t key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_cu[key][simd * 4 + (w & 3)]++;
            pat += char('0' + simd);
        }
        patterns[pat]++;
    }
    printf("threads %d blocks %d lds %d: %zu CUs seen\n", threads, blocks, ldsb, per_cu.size());
    printf("SIMD pattern of a workgroup's waves (wave 0, 1, ...): count\n");
    for (auto &p : patterns) printf("  %s : %d\n", p.first.c_str(), p.second);
    std::map<std::string, int> cu_shapes;
    for (auto &c : per_cu) {
        char buf[256]; int o = 0;
        for (int s = 0; s < 4; ++s) { o += snprintf(buf + o, sizeof buf - o, "S%d[", s); for (int w = 0; w < (nw < 4 ? nw : 4); ++w) o += snprintf(buf + o, sizeof buf - o, "%d%s", c.second[s * 4 + w], w + 1 < (nw < 4 ? nw : 4) ? "," : ""); o += snprintf(buf + o, sizeof buf - o, "] "); }
        cu_shapes[buf]++;
    }
    printf("per CU: waves per SIMD split by wave index inside the workgroup: count of CUs\n");
    int shown = 0;
    for (auto &s : cu_shapes) { if (shown++ < 24) printf("  %s: %d\n", s.first.c_str(), s.second); }
    printf("  (%zu distinct shapes)\n", cu_shapes.size());
    // the first workgroups' placements in order
    printf("first 24 workgroups: (xcc se cu | simd per wave)\n");
    for (int b = 0; b < 24 && b < blocks; ++b) {
        const uint32_t hw = h[(b * nw) * 2], xcc = h[(b * nw) * 2 + 1] & 15u;
        printf("  wg %2d: xcc %u se %u cu %2u |", b, xcc, (hw >> 13) & 7u, (hw >> 8) & 15u);
        for (int w = 0; w < nw; ++w) printf(" %u", (h[(b * nw + w) * 2] >> 4) & 3u);
        printf("\n");
    }
    return 0;
}
