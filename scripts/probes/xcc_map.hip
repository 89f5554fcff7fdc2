// Probe: is the workgroup -> XCD placement the same from launch to launch? (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
}
int main() {
    const int nb = 1024, nl = 12;
    int* d; hipMalloc(&d, nb * nl * sizeof(int));
    for (int l = 0; l < nl; ++l) hipLaunchKernelGGL(k, dim3(nb), dim3(128), 0, 0, d + l * nb);
    hipDeviceSynchronize();
    std::vector<int> h(nb * nl); hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    for (int l = 0; l < nl; ++l) {
        int same = 0, rr = 0;
        for (int b = 0; b < nb; ++b) { same += h[l * nb + b] == h[b]; rr += h[l * nb + b] == (h[l * nb] + b) % 8; }
        printf("launch %2d: block0 on XCC %d, blocks on the same XCC as in launch 0: %4d/1024, round-robin from block0: %4d/1024\n",
               l, h[l * nb], same, rr);
    }
    int cnt[8] = {0}; for (int b = 0; b < nb; ++b) cnt[h[b]]++;
    printf("blocks per XCC in launch 0:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]); printf("\n");
    return 0;
}
