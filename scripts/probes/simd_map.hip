// Probe: on which SIMD of its CU does each wave of a 128-thread workgroup land when 4 such
// workgroups share a CU (the persistent rollout kernel's shape)?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(128) void k(int* out, int spin)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    const int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
    const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;     // stay resident while the grid fills
    lds[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 2 + wave) * 2] = hw; out[(blockIdx.x * 2 + wave) * 2 + 1] = xcc + (lds[threadIdx.x] == 12345.f); }
}
int main()
{
    const int nb = 1024;
    int* d; hipMalloc(&d, nb * 4 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(nb), dim3(128), 17 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<int> h(nb * 4); hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    std::map<long, std::vector<int>> per_cu;   // (xcc, se, sh, cu) -> list of simd*2+wave_role
    int same_simd = 0;
    for (int b = 0; b < nb; ++b) {
        int s[2]; long key = 0;
        for (int w = 0; w < 2; ++w) {
            const int hw = h[(b * 2 + w) * 2], xcc = h[(b * 2 + w) * 2 + 1];
            s[w] = (hw >> 4) & 3;
            key = ((long)xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
            per_cu[key].push_back(s[w] * 2 + w);
        }
        same_simd += s[0] == s[1];
    }
    printf("CUs used: %zu; workgroups whose two waves share a SIMD: %d / %d\n", per_cu.size(), same_simd, nb);
    int hist[5][5] = {};   // [#wave0 on a simd][#wave1 on that simd]
    for (auto& kv : per_cu) {
        int c[4][2] = {};
        for (int v : kv.second) c[v >> 1][v & 1]++;
        for (int s = 0; s < 4; ++s) hist[c[s][0] > 4 ? 4 : c[s][0]][c[s][1] > 4 ? 4 : c[s][1]]++;
    }
    printf("SIMDs by (#wave0, #wave1) resident:\n");
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) if (hist[a][b]) printf("  (%d wave0, %d wave1): %d SIMDs\n", a, b, hist[a][b]);
    int n = 0;
    for (auto& kv : per_cu) { if (n++ >= 3) break; printf("  CU %lx:", kv.first); for (int v : kv.second) printf(" s%d%c", v >> 1, (v & 1) ? 'b' : 'a'); printf("\n"); }
    return 0;
}
