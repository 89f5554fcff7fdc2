// Does v_mfma_f32_32x32x16_f16 on gfx950 keep subnormal f16 inputs?  (k_mlp_zone_h3's lo operands live there.)
// hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_f16_denorm.hip -o scripts/probes/mfma_f16_denorm_bin && ./scripts/probes/mfma_f16_denorm_bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(unsigned short abits, unsigned short bbits, float *out)
{
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = __builtin_bit_cast(_Float16, abits);
        b[j] = __builtin_bit_cast(_Float16, bbits);
    }
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main()
{
    float *d, h;
    hipMalloc(&d, 4);
    const struct { unsigned short a, b; const char *what; double expect; } cases[] = {
        { 0x0010, 0x3C00, "A = 2^-20 (subnormal), B = 1", 16 * 9.5367431640625e-07 },
        { 0x3C00, 0x0010, "A = 1, B = 2^-20 (subnormal)", 16 * 9.5367431640625e-07 },
        { 0x0001, 0x3C00, "A = 2^-24 (smallest), B = 1", 16 * 5.9604644775390625e-08 },
        { 0x0001, 0x0001, "A = B = 2^-24: products 2^-48", 16 * 3.552713678800501e-15 },
        { 0x0400, 0x3C00, "A = 2^-14 (smallest normal), B = 1", 16 * 6.103515625e-05 },
    };
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c.a, c.b, d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("%-40s -> %.10e (exact %.10e) %s\n", c.what, h, c.expect, h == (float)c.expect ? "kept" : "NOT exact");
    }
    return 0;
}
