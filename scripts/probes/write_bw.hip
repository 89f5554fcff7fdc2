// Probe: what write bandwidth does a streaming store pattern like the tile flush reach?
// Each wave owns `tile_bytes` of contiguous output and rewrites it `steps` times with
// 1 KiB buffer_store_dwordx4 bursts (the persistent rollout kernel's stream wave without
// any of its compute).  Varies waves per block / blocks, cache policy.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ void k(float* out, int tile_bytes, int steps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tile = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
    float* dst = out + tile * (tile_bytes / 4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, tile_bytes, 0x00020000);
    v4f_t val = { (float)lane, 1.f, 2.f, 3.f };
    for (int t = 0; t < steps; ++t) {
        val.y = (float)t;
        for (int off = lane * 16; off + 16 <= tile_bytes; off += 1024)
            __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, off, 0, AUX);
    }
}
int main(int argc, char** argv)
{
    const int steps = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* d; const size_t cap = 512ull << 20; hipMalloc(&d, cap);
    struct Cfg { int blocks, waves, tile_bytes; };
    const Cfg cfgs[] = { {1024, 1, 38400}, {1024, 2, 19200}, {1024, 4, 9600}, {2048, 1, 19200}, {4096, 1, 9600},
                         {1024, 1, 153600}, {4096, 1, 38400}, {4096, 4, 38400} };
    for (const Cfg& c : cfgs) {
        for (int aux : {0, 2}) {
            const double bytes = (double)c.blocks * c.waves * c.tile_bytes;
            if (bytes > cap) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (aux == 0) hipLaunchKernelGGL(k<0>, dim3(c.blocks), dim3(64 * c.waves), 0, 0, d, c.tile_bytes, steps);
                else hipLaunchKernelGGL(k<2>, dim3(c.blocks), dim3(64 * c.waves), 0, 0, d, c.tile_bytes, steps);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("blocks %5d x %d waves, %6d B/wave/step (%.1f MB/step), %s: %.2f us/step, %.2f TB/s\n", c.blocks, c.waves,
                   c.tile_bytes, bytes / 1e6, aux ? "nt   " : "plain", best * 1e3 / steps, bytes * steps / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
