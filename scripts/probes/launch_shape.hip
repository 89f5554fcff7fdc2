// Probe: does the SHAPE of the launch matter for a kernel of K1's size?  The same work -- 1 024 tiles, per tile one wave
// that loads r KiB and stores w KiB (ColourMatch-6's and PointTSP-25's bytes) beside an idle partner wave -- launched as
// 1024 x 128 (K1 today: one tile per workgroup), 512 x 256, 256 x 512, 128 x 1024 threads (2, 4, 8 tiles per workgroup)
// and 1024 x 64 (no partner wave).  Dispatch begin/end (what rocprofv3 reports) and back-to-back loop time per launch.
// Diagnostic only; build: hipcc --offload-arch=gfx950 -O3 launch_shape.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_stream(const float4 *__restrict__ in, float4 *__restrict__ out, int r16, int w16)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TILES = THREADS >= 128 ? THREADS / 128 : 1;
    const int tile = blockIdx.x * TILES + (THREADS >= 128 ? (wave >> 1) : 0);
    if (THREADS >= 128 && (wave & 1)) { __syncthreads(); return; }
    const float4 *src = in + (size_t)tile * r16 * 64;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < r16; ++i) {
        const float4 v = src[i * 64 + lane];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (THREADS >= 128) __syncthreads();
    float4 *dst = out + (size_t)tile * w16 * 64;
    for (int i = 0; i < w16; ++i) dst[i * 64 + lane] = make_float4(acc.x + i, acc.y, acc.z, acc.w);
}

template <typename F>
static void time_it(const char *name, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int i = 0; i < 320; ++i) {
        launch(e0, e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i >= 20) us.push_back(ms * 1e3f);
    }
    hipEvent_t b0, b1; hipEventCreate(&b0); hipEventCreate(&b1);
    hipEventRecord(b0, 0);
    for (int i = 0; i < 1000; ++i) launch(nullptr, nullptr);
    hipEventRecord(b1, 0); hipEventSynchronize(b1);
    float loop_ms; hipEventElapsedTime(&loop_ms, b0, b1);
    std::sort(us.begin(), us.end());
    printf("%-40s dispatch min %6.2f median %6.2f p90 %6.2f us | loop %6.2f us/launch\n", name, us.front(), us[us.size() / 2],
           us[us.size() * 9 / 10], loop_ms);
}

int main()
{
    float4 *in, *out;
    const size_t cap = 256ull << 20;
    hipMalloc(&in, cap); hipMalloc(&out, cap);
    hipMemset(in, 0, cap); hipMemset(out, 0, cap);
    struct Shape { const char *name; int r_env, w_env; };
    const Shape shapes[] = { { "ColourMatch-6 (170 in / 290 out)", 170, 290 }, { "PointTSP-25 (330 in / 714 out)", 330, 714 }, { "nothing", 0, 0 } };
    for (const Shape &s : shapes) {
        const int r16 = (s.r_env * 64 + 1023) / 1024, w16 = (s.w_env * 64 + 1023) / 1024;
        printf("-- %s\n", s.name);
#define RUN(T, LDS)                                                                                                   \
        { char nm[64]; snprintf(nm, sizeof nm, "%4d x %4d threads, %3d KB LDS", 1024 * 128 / (T >= 128 ? T : 128), T, LDS / 1024); \
          time_it(nm, [&](hipEvent_t a, hipEvent_t b) {                                                                \
              hipExtLaunchKernelGGL((k_stream<T>), dim3(1024 * 128 / (T >= 128 ? T : 128)), dim3(T), LDS, 0, a, b, 0, in, out, r16, w16); }); }
        RUN(128, 32768) RUN(256, 65536) RUN(512, 65536) RUN(1024, 65536) RUN(64, 32768) RUN(128, 0) RUN(256, 0)
    }
    return 0;
}
