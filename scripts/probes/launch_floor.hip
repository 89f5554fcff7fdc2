// Probe: what does ONE launch of the step kernel's shape cost before it does any env work?
//   empty        1024 workgroups x 128 threads, 32 KB of LDS, no memory traffic
//   kernarg      the same with a 640-byte by-value parameter block of which every wave reads the LAST word
//   stream R/W   the same grid moving the step kernel's bytes and nothing else: each workgroup loads r_bytes and
//                stores w_bytes of its own contiguous slice with 16-byte-per-lane accesses (wave 0 loads, then stores,
//                like the zone wave: load burst -> barrier -> store burst), for the byte counts of
//                PointTSP-25 (330 B in / 714 B out per env) and ColourMatch-6 (170 B in / 290 B out per env)
// Times are the dispatch's own begin/end timestamps (hipExtLaunchKernelGGL events = what rocprofv3 reports), min and
// median of 200 back-to-back launches.  Diagnostic only; build: hipcc --offload-arch=gfx950 -O3 launch_floor.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Big { unsigned long long w[80]; };   // 640 B, like DevParams by value

__global__ __launch_bounds__(128) void k_empty(int *sink)
{
    extern __shared__ float lds[];
    if (sink && threadIdx.x == 0 && blockIdx.x == 0xFFFFFF) sink[0] = (int)lds[0];
}

__global__ __launch_bounds__(128) void k_kernarg(Big b, unsigned long long *sink)
{
    if (b.w[79] == 0x1234 && threadIdx.x == 0) sink[blockIdx.x] = b.w[3];
}

typedef float v4f_t __attribute__((ext_vector_type(4)));
// wave 0 of each workgroup: load r_bytes (16 B per lane per instruction, all issued up front), then store w_bytes
template <bool SPLIT>
__global__ __launch_bounds__(128) void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, int r16, int w16)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave != 0) { __syncthreads(); return; }
    const float4 *src = in + (size_t)blockIdx.x * r16 * 64;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < r16; ++i) {
        const float4 v = src[i * 64 + lane];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    __syncthreads();
    float4 *dst = out + (size_t)blockIdx.x * w16 * 64;
    for (int i = 0; i < w16; ++i) dst[i * 64 + lane] = make_float4(acc.x + i, acc.y, acc.z, acc.w);
}

template <typename F>
static void time_it(const char *name, F launch, double mb)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int i = 0; i < 220; ++i) {
        launch(e0, e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 20) us.push_back(ms * 1e3f);
    }
    // back to back (no host sync in between): the loop time per launch
    hipEvent_t b0, b1; hipEventCreate(&b0); hipEventCreate(&b1);
    hipEventRecord(b0, 0);
    for (int i = 0; i < 500; ++i) launch(nullptr, nullptr);
    hipEventRecord(b1, 0); hipEventSynchronize(b1);
    float loop_ms; hipEventElapsedTime(&loop_ms, b0, b1);
    std::sort(us.begin(), us.end());
    printf("%-44s dispatch min %6.2f  median %6.2f  p90 %6.2f us | back-to-back loop %6.2f us/launch", name, us.front(),
           us[us.size() / 2], us[us.size() * 9 / 10], loop_ms * 1e3 / 500);
    if (mb > 0) printf(" | %.1f MB -> %.2f TB/s at the median", mb, mb / us[us.size() / 2]);   // MB / us = TB/s
    printf("\n");
}

int main()
{
    const int blocks = 1024;
    int *sink; hipMalloc(&sink, 1 << 20);
    float4 *in, *out;
    const size_t cap = 256ull << 20;
    hipMalloc(&in, cap); hipMalloc(&out, cap);
    hipMemset(in, 0, cap); hipMemset(out, 0, cap);
    Big big{}; big.w[79] = 1;
    time_it("empty, 1024 x 128, 32 KB LDS", [&](hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_empty, dim3(blocks), dim3(128), 32768, 0, a, b, 0, sink); }, 0);
    time_it("empty, 1024 x 128, no LDS", [&](hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_empty, dim3(blocks), dim3(128), 0, 0, a, b, 0, sink); }, 0);
    time_it("empty, 256 x 512, no LDS", [&](hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(128), 0, 0, a, b, 0, sink); }, 0);
    time_it("640-B by-value kernarg, last word read", [&](hipEvent_t a, hipEvent_t b) {
        hipExtLaunchKernelGGL(k_kernarg, dim3(blocks), dim3(128), 0, 0, a, b, 0, big, (unsigned long long *)sink); }, 0);
    struct Shape { const char *name; int r_env, w_env; };
    const Shape shapes[] = { { "stream PointTSP-25 bytes (330 in / 714 out)", 330, 714 },
                             { "stream TimedTSP-25 bytes (430 in / 814 out)", 430, 814 },
                             { "stream ColourMatch-6 bytes (170 in / 290 out)", 170, 290 },
                             { "stream PointTSP-15 bytes (230 in / 470 out)", 230, 470 },
                             { "stream stores only (0 in / 714 out)", 0, 714 },
                             { "stream loads only (330 in / 16 out)", 330, 16 } };
    for (const Shape &s : shapes) {
        const int r16 = (s.r_env * 64 + 1023) / 1024, w16 = (s.w_env * 64 + 1023) / 1024;
        const double mb = (double)blocks * (r16 + w16) * 1024 / 1e6;
        time_it(s.name, [&](hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(k_stream<false>, dim3(blocks), dim3(128), 32768, 0, a, b, 0, in, out, r16, w16); }, mb);
    }
    return 0;
}
