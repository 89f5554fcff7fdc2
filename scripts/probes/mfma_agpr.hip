// Probe: does it cost anything to feed v_mfma_f32_32x32x16_bf16 its A or B operand from AGPRs (result in VGPRs)?
// 256 workgroups x 4 waves, dependent chain, 16 distinct operands each side.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
__device__ inline bf16x8 pin_a(bf16x8 f) { asm("" : "+a"(f)); return f; }
__device__ inline bf16x8 pin_v(bf16x8 f) { asm("" : "+v"(f)); return f; }

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *ops, unsigned long long *out, float *sink, int groups)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a[12], b[12];
    for (int i = 0; i < 12; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, ops[(i * 64 + lane)]);
        b[i] = __builtin_bit_cast(bf16x8, ops[((12 + i) * 64 + lane)]);
        a[i] = (MODE & 2) ? pin_a(a[i]) : pin_v(a[i]);
        b[i] = (MODE & 1) ? pin_a(b[i]) : pin_v(b[i]);
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < groups; ++g) {
#pragma unroll
        for (int i = 0; i < 12; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float keep = 0.f;
    for (int i = 0; i < 16; ++i) keep += acc[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(const uint4 *ops, unsigned long long *d, float *sink)
{
    const int groups = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, ops, d, sink, groups);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    printf("A in %s, B in %s: %.1f cycles per MFMA\n", (MODE & 2) ? "AGPR" : "VGPR", (MODE & 1) ? "AGPR" : "VGPR", (double)c / (groups * 12.0));
}

int main()
{
    uint4 *ops; unsigned long long *d; float *sink;
    hipMalloc(&ops, 24 * 64 * 16); hipMemset(ops, 0x3c, 24 * 64 * 16); hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4);
    run<0>(ops, d, sink); run<1>(ops, d, sink); run<2>(ops, d, sink); run<3>(ops, d, sink);
    return 0;
}
