// Probe: would K4's layer-2 region run faster on v_mfma_f32_16x16x32_bf16 than on the shipped
// v_mfma_f32_32x32x16_bf16?  MI355X_MICROARCH.md (DVFS, item 7) measures 1.12-1.15 x the FLOP/s for BARE loops of the
// 16x16x32 shape on random data (higher sustained clock).  A K4 region is not a bare loop: this rebuilds ONE region --
// the 32-feature x 32-env output block of zone_net_.2 with everything the shipped kernel does around it -- in both
// shapes, one wave per SIMD on every CU, random bf16 operands, and reports wall time, cycles and clock per region.
//   shape 32: 12 chain MFMAs (K = 192 in steps of 16; weight fragments LDS -> AGPR, one ds_read_b128 each) + 1 layer-1 MFMA
//             + 1 pooling MFMA (the shipped sparse product costs the same 32 cycles) + ReLU / convert of the previous
//             block (8 v_cvt_pk_bf16_f32 + 8 v_pk_max_i16)                                   = 14 MFMA = 448 matrix cycles
//   shape 16: the same block as 2 x 2 tiles of 16 x 16: 24 chain MFMAs (K = 192 in steps of 32, FOUR independent chains;
//             12 weight fragments from LDS, each used for both env halves) + 4 layer-1 MFMAs (its K = 16 operand fills
//             half of the instruction's K = 32) + 4 pooling MFMAs (a 16 x 16 result per instruction: [I|0] and [0|I]
//             per env half) + the same 16 conversions                                       = 32 MFMA = 512 matrix cycles
// So the 16-shape starts 14 % behind in matrix cycles and needs more than that from the clock.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 h2 __attribute__((ext_vector_type(2)));
__device__ inline bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ inline bf16x8 pin_a(bf16x8 f) { asm("" : "+a"(f)); return f; }
__device__ inline unsigned pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){ a, b }, h2)); }
__device__ inline bf16x8 relu8(uint4 u)
{
    const s16x8 z = { 0, 0, 0, 0, 0, 0, 0, 0 };
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, u), z));
}

template <int SHAPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(unsigned long long *out, float *sink, int iters)
{
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 24 * 64; i += blockDim.x) {   // random bf16, magnitude 1/8 .. 1/2, both signs
        unsigned h = (unsigned)i * 2654435761u + 12345u;
        auto nx = [&]() { h = h * 1664525u + 1013904223u; return ((h >> 8) & 0x80FF80FFu) | 0x3E003E00u; };
        lds[i] = make_uint4(nx(), nx(), nx(), nx());
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 xa[12], wf[2][12], w1 = pin_a(as_frag(lds[lane])), x0 = as_frag(lds[64 + lane]), ind = x0;
    for (int i = 0; i < 12; ++i) { xa[i] = as_frag(lds[(i * 64 + lane)]); asm volatile("" : "+v"(xa[i])); }
    for (int i = 0; i < 12; ++i) wf[0][i] = pin_a(as_frag(lds[((12 + i) * 64 + lane)]));
    unsigned long long t0, r0, t1, r1;
    float keep = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2], a1, pool;
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; a1[i] = 0.f; pool[i] = 0.f; }
        const f32x16 zero = acc[0];
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int i = 0; i < 12; ++i) wf[(n + 1) & 1][i] = as_frag(lds[((12 * ((n + 1) & 1) + i) * 64 + ((lane + it) & 63))]);
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 prev = acc[(n + 1) & 1];
                acc[n & 1] = zero;
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    acc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[i], pin_a(wf[n & 1][i]), acc[n & 1], 0, 0, 0);
                const bf16x8 f0 = relu8(make_uint4(pk(prev[0], prev[1]), pk(prev[2], prev[3]), pk(prev[4], prev[5]), pk(prev[6], prev[7])));
                const bf16x8 f1 = relu8(make_uint4(pk(prev[8], prev[9]), pk(prev[10], prev[11]), pk(prev[12], prev[13]), pk(prev[14], prev[15])));
                pool = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ind, f0, pool, 0, 0, 0);
                asm volatile("" ::"v"(f1));
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, zero, 0, 0, 0);
#ifdef INTERLEAVE
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
#pragma unroll
                for (int g = 12; g < 14; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        for (int i = 0; i < 16; ++i) keep += acc[0][i] + acc[1][i] + a1[i] + pool[i];
    } else {
        f32x4 acc[2][4], a1[4], pool[4];
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 4; ++i) { acc[0][q][i] = 0.f; acc[1][q][i] = 0.f; a1[q][i] = 0.f; pool[q][i] = 0.f; }
        const f32x4 zero = acc[0][0];
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
#pragma unroll
                for (int i = 0; i < 12; ++i) wf[(n + 1) & 1][i] = as_frag(lds[((12 * ((n + 1) & 1) + i) * 64 + ((lane + it) & 63))]);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 prev[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { prev[q] = acc[(n + 1) & 1][q]; acc[n & 1][q] = zero; }
                // 6 k-steps of 32: weight fragments 2 per k-step (feature halves), activation fragments 2 (env halves)
#pragma unroll
                for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[n & 1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pin_a(wf[n & 1][2 * ks + (q >> 1)]), xa[2 * ks + (q & 1)],
                                                                                acc[n & 1][q], 0, 0, 0);
                // ReLU + convert the previous block: per env half, two feature tiles -> one fragment
                const bf16x8 f0 = relu8(make_uint4(pk(prev[0][0], prev[0][1]), pk(prev[0][2], prev[0][3]), pk(prev[2][0], prev[2][1]), pk(prev[2][2], prev[2][3])));
                const bf16x8 f1 = relu8(make_uint4(pk(prev[1][0], prev[1][1]), pk(prev[1][2], prev[1][3]), pk(prev[3][0], prev[3][1]), pk(prev[3][2], prev[3][3])));
                pool[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ind, f0, pool[0], 0, 0, 0);
                pool[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f0, pool[1], 0, 0, 0);
                pool[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ind, f1, pool[2], 0, 0, 0);
                pool[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f1, pool[3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) a1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q & 2 ? w1 : ind, q & 1 ? x0 : xa[0], zero, 0, 0, 0);
#ifdef INTERLEAVE
                // the region's 32 MFMAs with its other instructions spread over their gaps: 1 MFMA, 1 LDS read (the first
                // 12 gaps), 1-2 vector instructions -- what the shipped kernel's hand pipelining does for the 32-shape
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
#pragma unroll
                for (int g = 12; g < 32; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < 4; ++i) keep += acc[0][q][i] + acc[1][q][i] + a1[q][i] + pool[q][i];
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    sink[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <int SHAPE>
static void run(unsigned long long *d, float *sink, int iters, const char *tag)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 64 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(256), 24 * 64 * 16, 0, d, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2] = { 0, 0 };
    hipMemcpy(c, d, 16, hipMemcpyDeviceToHost);
    const int n_mfma = SHAPE == 32 ? 14 : 32, cyc = SHAPE == 32 ? 32 : 16;
    printf("%s shape %2d: %7.1f ns per region (wall %.0f ms), %6.1f cycles per region (%d MFMA = %d matrix cycles), in-kernel clock %.3f GHz\n",
           tag, SHAPE, ms * 1e6 / iters, ms, (double)c[0] / iters, n_mfma, n_mfma * cyc, (double)c[0] / (c[1] * 10.0));
}

int main()
{
    unsigned long long *d; float *sink;
    hipMalloc(&d, 16); hipMalloc(&sink, 256 * 256 * 4);
    const int warm = 400000, iters = 8000000;       // ~0.1 s warm-up, then ~2 s per timed launch (one region ~ 250 ns)
    run<32>(d, sink, warm, "[warm]"); run<16>(d, sink, warm, "[warm]");
    for (int rep = 0; rep < 2; ++rep) { run<32>(d, sink, iters, "[timed]"); run<16>(d, sink, iters, "[timed]"); }
    return 0;
}
