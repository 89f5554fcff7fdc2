// Probe: the K4 tile region rebuilt piece by piece, one wave per SIMD on every CU, cycles per region.
//   mode 0: 12-MFMA dependent chain + 1 independent MFMA                                  (13 MFMA = 416 cycles)
//   mode 1: + the next chain's 12 B fragments read from LDS into AGPRs at the top
//   mode 2: + conversion (8 cvt_pk + 8 pk_max) of the PREVIOUS chain and 2 pooling MFMAs   (15 MFMA = 480 cycles)
//   mode 3: + conversion of the previous independent MFMA's result into live registers
// Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 h2 __attribute__((ext_vector_type(2)));
__device__ inline bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ inline bf16x8 pin_a(bf16x8 f) { asm("" : "+a"(f)); return f; }
__device__ inline unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){ a, b }, h2)); }
__device__ inline void conv(const f32x16 &c, bf16x8 &f0, bf16x8 &f1)
{
    const s16x8 z = { 0, 0, 0, 0, 0, 0, 0, 0 };
    uint4 u = make_uint4(pk(c[0], c[1]), pk(c[2], c[3]), pk(c[4], c[5]), pk(c[6], c[7]));
    uint4 v = make_uint4(pk(c[8], c[9]), pk(c[10], c[11]), pk(c[12], c[13]), pk(c[14], c[15]));
    f0 = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, u), z));
    f1 = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}
__device__ inline void pool_mfma(f32x16 &p, bf16x8 a, bf16x8 b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(p) : "v"(a), "v"(b));
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(unsigned long long *out, float *sink, int iters)
{
    extern __shared__ uint4 lds[];
#ifdef RANDOM_DATA      // bf16 values of magnitude ~1/8..1/2 with random mantissas and signs: every operand bit toggles
    for (int i = threadIdx.x; i < 24 * 64; i += blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + 12345u;
        auto nx = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 8) & 0x80FF80FFu) | 0x3E003E00u; };
        lds[i] = make_uint4(nx(), nx(), nx(), nx());
    }
#else
    for (int i = threadIdx.x; i < 24 * 64; i += blockDim.x) lds[i] = make_uint4(0x3C003C00u + i, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
#endif
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 xa[12], wf[2][12], xb[12], w1 = pin_a(as_frag(lds[lane])), x0 = as_frag(lds[64 + lane]), ind0 = x0, ind1 = x0;
    for (int i = 0; i < 12; ++i) { xa[i] = as_frag(lds[(i * 64 + lane)]); asm volatile("" : "+v"(xa[i])); xb[i] = xa[i]; }
    for (int i = 0; i < 12; ++i) wf[0][i] = pin_a(as_frag(lds[((12 + i) * 64 + lane)]));
    f32x16 acc[2], a1, pool;
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; a1[i] = 0.f; pool[i] = 0.f; }
    asm volatile("" : "+a"(pool));
    const f32x16 zero = acc[0];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 12; ++i) wf[(n + 1) & 1][i] = as_frag(lds[((12 * ((n + 1) & 1) + i) * 64 + ((lane + it) & 63))]);
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i) wf[(n + 1) & 1][i] = wf[n & 1][i];
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x16 prev = acc[(n + 1) & 1];
            acc[n & 1] = zero;
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i], pin_a(wf[n & 1][i]), acc[n & 1], 0, 0, 0);
            if (MODE >= 2) {
                bf16x8 f0, f1;
                conv(prev, f0, f1);
                pool_mfma(pool, ind0, f0);
                pool_mfma(pool, ind1, f1);
            } else {
                asm volatile("" ::"v"(prev));
            }
            if (MODE >= 3) {
                conv(a1, xb[2 * n], xb[2 * n + 1]);
                asm volatile("" : "+v"(xb[2 * n]), "+v"(xb[2 * n + 1]));
            } else {
                asm volatile("" ::"v"(a1));
            }
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float keep = 0.f;
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    for (int i = 0; i < 16; ++i) keep += acc[0][i] + acc[1][i] + a1[i] + pool[i];
    for (int i = 0; i < 12; ++i) keep += (float)xb[i][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(unsigned long long *d, float *sink)
{
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 64 * 16);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 24 * 64 * 16, 0, d, sink, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    const int n_mfma = MODE >= 2 ? 15 : 13;
#ifdef RANDOM_DATA
    printf("[random operands] ");
#endif
    printf("mode %d: %.0f cycles per region (%d MFMA = %d cycles of matrix pipe)\n", MODE, (double)c / iters, n_mfma, 32 * n_mfma);
}

int main()
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4);
    // (mode 0 is not run: with identical operands in both regions of an iteration the compiler merges the two chains)
    run<1>(d, sink); run<2>(d, sink); run<3>(d, sink);
    return 0;
}
