// Probe: is a dense bf16 MFMA stream on every SIMD clock (power) limited, and does it depend on the data?
// 256 workgroups x 4 waves, each wave a dependent chain of v_mfma_f32_32x32x16_bf16 with operands in registers.
// Reports, for all-ones and for random operands: s_memtime ticks per MFMA, s_memrealtime (100 MHz) time per
// MFMA, their ratio (= the frequency s_memtime counts at), and the achieved TFLOP/s.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ __launch_bounds__(256) void k(const uint4 *ops, unsigned long long *out, float *sink, int groups)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ops[(i * 64 + lane)]);
        b[i] = __builtin_bit_cast(bf16x8, ops[((4 + i) * 64 + lane)]);
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc, 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float keep = 0.f;
    for (int i = 0; i < 16; ++i) keep += acc[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int main()
{
    const int groups = 20000;     // 320 k MFMAs per wave
    uint4 *d_ops; unsigned long long *d_out; float *sink;
    hipMalloc(&d_ops, 8 * 64 * 16); hipMalloc(&d_out, 16); hipMalloc(&sink, 256 * 256 * 4);
    uint4 host[8 * 64];
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        for (auto &u : host) {
            auto bf = [&]() -> unsigned {
                if (mode == 0) return 0x3F80u;                                        // 1.0
                if (mode == 1) return (unsigned)(0x3C00 + (rand() & 0x3FF)) | ((rand() & 1) << 15);   // small randoms, both signs
                return 0u;                                                            // zeros
            };
            u = make_uint4(bf() | bf() << 16, bf() | bf() << 16, bf() | bf() << 16, bf() | bf() << 16);
        }
        hipMemcpy(d_ops, host, sizeof(host), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_ops, d_out, sink, groups / 10);   // warm
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_ops, d_out, sink, groups);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long o[2]; hipMemcpy(o, d_out, 16, hipMemcpyDeviceToHost);
        const double n = groups * 16.0;
        printf("%-7s: %.1f s_memtime ticks per MFMA, %.2f ns per MFMA (s_memrealtime), s_memtime rate %.3f GHz, kernel %.2f ms, %.0f TFLOP/s\n",
               mode == 0 ? "ones" : mode == 1 ? "random" : "zeros", o[0] / n, o[1] * 10.0 / n, (double)o[0] / (o[1] * 10.0), ms,
               1024.0 * n * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
