// Probe: what does one v_mfma_f32_32x32x16_bf16 cost a wave in the situations the actor-network kernel puts
// it in?  One workgroup per CU; waves per SIMD = 1 or 2; per wave a loop of 12-MFMA groups:
//   mode 0: one dependent accumulation chain, operands in registers
//   mode 1: two alternating independent chains
//   mode 2: one chain, every MFMA's B operand read from LDS just before it (ds_read_b128 + s_waitcnt)
//   mode 3: one chain, the 12 B operands of the NEXT group read from LDS at the start of a group
//   mode 4: mode 3 + 8 VALU conversions (cvt_pk / pk_max) after every group on the finished accumulator
// Prints cycles per MFMA (s_memtime, wave 0 of block 0) -- 32 is the matrix pipe's pace.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;
__device__ inline bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int groups)
{
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 12 * 64 * 4; i += blockDim.x) lds[i] = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* w = lds + (wave & 3) * 12 * 64;
    bf16x8 a = as_frag(lds[lane]), b[12], nb[12];
    for (int k2 = 0; k2 < 12; ++k2) b[k2] = as_frag(w[k2 * 64 + lane]);
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float keep = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < groups; ++g) {
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2) nb[k2] = as_frag(w[k2 * 64 + ((lane + g) & 63)]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 12; ++k2) {
            if (MODE == 2) b[k2] = as_frag(w[k2 * 64 + ((lane + g) & 63)]);
            if (MODE == 1 && (k2 & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[k2], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[k2], acc0, 0, 0, 0);
        }
        if (MODE == 4) {
            uint4 u;
            typedef float f2 __attribute__((ext_vector_type(2)));
            typedef __bf16 h2 __attribute__((ext_vector_type(2)));
            u.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){ acc0[0], acc0[1] }, h2));
            u.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){ acc0[2], acc0[3] }, h2));
            u.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){ acc0[4], acc0[5] }, h2));
            u.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){ acc0[6], acc0[7] }, h2));
            const s16x8 z = { 0, 0, 0, 0, 0, 0, 0, 0 };
            a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, u), z));
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int k2 = 0; k2 < 12; ++k2) b[k2] = nb[k2];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) keep += acc0[i] + acc1[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(int waves_per_simd, unsigned long long* d, float* sink)
{
    const int groups = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * waves_per_simd), 12 * 64 * 4 * 16, 0, d, sink, groups);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    printf("mode %d, %d wave(s)/SIMD: %.1f cycles per MFMA of ONE wave (%.1f per MFMA on the SIMD)\n", MODE, waves_per_simd,
           (double)c / (groups * 12.0), (double)c / (groups * 12.0 * waves_per_simd));
}

int main()
{
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 256 * 512 * 4);
    for (int w = 1; w <= 2; ++w) { run<0>(w, d, sink); run<1>(w, d, sink); run<2>(w, d, sink); run<3>(w, d, sink); run<4>(w, d, sink); }
    return 0;
}
