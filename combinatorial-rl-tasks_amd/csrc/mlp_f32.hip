// mlp_f32.hip -- the reference's actor-critic network in float32 (ZENV_MLP_F32), gfx950.
//
// Same network as mlp_policy.hip (ZoneEnvModel main/src/env_model.py:48-79, ACModel flat_model.py:21-68,
// PolicyNetwork policy_network.py:9-52), evaluated in the reference's own arithmetic type: float32 operands, float32
// FMA accumulation, precise expf / log1pf.  It exists so that evaluate() with a trained checkpoint -- and any test that
// asks for it -- sees actions within 1e-5 of what the reference's torch modules produce; the bf16 MFMA kernels are the
// fast mode (their operands carry 8 significant bits).  Only the summation order differs from torch's GEMM, and the
// mean over an env's zone rows is taken BEFORE zone_net_'s third (activation-free) layer, with which it commutes.
//
// Two kernels, chosen by batch size (launch_mlp_forward_f32):
//
//  * k_mlp_zone_f32m -- the WHOLE network on the float32 MATRIX instruction (v_mfma_f32_32x32x2_f32: bit for bit a
//    k-ordered fmaf chain at the vector rate, one VGPR per operand).  A 32x32 result has its column (zone row / env
//    slot) on the lane and its rows (features) in the 16 registers, so relu(layer n) IS layer n + 1's B operand
//    register by register when that layer's weights are packed in accumulator k order; the two zone layers' images
//    (147 KB + 12 KB) sit in LDS, one workgroup per CU; the per-env head layers chain on in the same registers with
//    their images read from L2.  1.12 ms per step at N = 65 536 = 0.76 of the float32 MFMA peak; 0.56 ms from
//    N = 4 096 to 32 768 (one 32-env group per wave).
//  * k_mlp_f32 -- vector ALU, for small batches (evaluate(): 500 envs): a workgroup of kMlpHP = 192 threads owns EB
//    consecutive envs, thread j owns hidden feature j.  An input row's activations sit in LDS ([k][row],
//    row-contiguous: one ds_read_b128 = 4 rows of a feature, the same address in every lane = a broadcast), the weights
//    are read from memory transposed ([k][j]: 192 consecutive floats per k, served by L2 -- every workgroup streams
//    the same 0.6 MB), each thread keeps RP = 32 row accumulators in registers.  (pooled_in: a diagnostic entry that
//    takes the per-env zone sums from a float32 array instead of computing them.)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "mlp_head_out.hpp"
#include "mlp_policy.hpp"

namespace zenvk {
namespace {

constexpr int HP = kMlpHP;     // 192 threads = hidden features (padded)
constexpr int RP = 32;         // rows (zone rows of consecutive envs) per pass
constexpr int EB = 4;          // envs per workgroup
constexpr int KIN = 16;        // padded input width of zone_net_.0 (8 obs + F <= 7 zone features)

// acc[r] += w * x[k][r] for the RP rows of one pass
__device__ __forceinline__ void fma_rows(float (&acc)[RP], float w, const float *__restrict__ xk)
{
    const float4 *x4 = reinterpret_cast<const float4 *>(xk);
#pragma unroll
    for (int q = 0; q < RP / 4; ++q) {
        const float4 v = x4[q];
        acc[4 * q + 0] = __builtin_fmaf(w, v.x, acc[4 * q + 0]);
        acc[4 * q + 1] = __builtin_fmaf(w, v.y, acc[4 * q + 1]);
        acc[4 * q + 2] = __builtin_fmaf(w, v.z, acc[4 * q + 2]);
        acc[4 * q + 3] = __builtin_fmaf(w, v.w, acc[4 * q + 3]);
    }
}

// out[e] = b[j] + sum_k wt[k][j] * x[e][k]   for the EB env vectors in LDS (x: [EB][stride])
__device__ __forceinline__ void matvec(float (&out)[EB], const float *__restrict__ wt, const float *__restrict__ b,
                                       const float *__restrict__ x, int stride, int n_in, int j)
{
#pragma unroll
    for (int e = 0; e < EB; ++e) out[e] = b[j];
    for (int k = 0; k < n_in; ++k) {
        const float w = wt[(size_t)k * HP + j];
#pragma unroll
        for (int e = 0; e < EB; ++e) out[e] = __builtin_fmaf(w, x[e * stride + k], out[e]);
    }
}

__global__ __launch_bounds__(HP) void k_mlp_f32(MlpF32 w, int N, int Z, int F, const float *__restrict__ obs,
                                                const float *__restrict__ zone_obs, float *__restrict__ mu,
                                                float *__restrict__ stdv, float *__restrict__ value,
                                                float *__restrict__ value_sigma, MlpAction act,
                                                const float *__restrict__ pooled_in)
{
    __shared__ __align__(16) float x0[KIN * RP];        // zone_net_.0 input of the pass      [k][row]
    __shared__ __align__(16) float y1[HP * RP];         // relu(zone_net_.0) of the pass      [k][row]
    __shared__ float va[EB * (8 + HP)];                 // head vectors: [obs (8); features]  per env
    __shared__ float vb[EB * HP];
    const int j = threadIdx.x;
    const int h = w.h;
    const bool live = j < h;                             // padded features stay exactly 0
    const int env0 = blockIdx.x * EB;
    const int n_env = min(EB, N - env0);
    const int n_rows = n_env * Z;
    const float b1 = w.b1[j], b2 = w.b2[j];

    // ---- zone_net_.0, ReLU, zone_net_.2, ReLU on every [obs, zone row]; rows summed per env
    float psum[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) psum[e] = 0.f;
    if (pooled_in) {
        // the per-env sums over the zone rows come from k_mlp_zone_f32m
#pragma unroll
        for (int e = 0; e < EB; ++e) psum[e] = e < n_env ? pooled_in[(size_t)(env0 + e) * HP + j] : 0.f;
    }
    for (int r0 = 0; r0 < (pooled_in ? 0 : n_rows); r0 += RP) {
        __syncthreads();                                  // the previous pass is done with x0 / y1
        for (int i = j; i < KIN * RP; i += HP) {
            const int k = i / RP, r = i % RP, row = r0 + r;
            float v = 0.f;
            if (row < n_rows) {
                const int e = row / Z, z = row - e * Z;
                if (k < 8) v = obs[(size_t)(env0 + e) * 8 + k];
                else if (k < 8 + F) v = zone_obs[((size_t)(env0 + e) * Z + z) * F + (k - 8)];
            }
            x0[k * RP + r] = v;
        }
        __syncthreads();
        float acc[RP];
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[r] = b1;
        for (int k = 0; k < 8 + F; ++k) fma_rows(acc, w.w1t[(size_t)k * HP + j], x0 + k * RP);
#pragma unroll
        for (int q = 0; q < RP / 4; ++q)
            reinterpret_cast<float4 *>(y1 + j * RP)[q] =
                live ? make_float4(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f), fmaxf(acc[4 * q + 2], 0.f),
                                   fmaxf(acc[4 * q + 3], 0.f))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RP; ++r) acc[r] = b2;
        for (int k = 0; k < h; ++k) fma_rows(acc, w.w2t[(size_t)k * HP + j], y1 + k * RP);
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const int row = r0 + r;
            const int e = row / Z;
            const float v = fmaxf(acc[r], 0.f);
#pragma unroll
            for (int ee = 0; ee < EB; ++ee)
                if (row < n_rows && e == ee) psum[ee] += v;
        }
    }

    // ---- per env: zone_emb = zone_net_.4(mean) ; c = combine_net_([obs, zone_emb]) ; actor / critic heads
    __syncthreads();
    const float inv_z = 1.0f / (float)Z;
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        vb[e * HP + j] = live ? psum[e] * inv_z : 0.f;
        if (j < 8) va[e * (8 + HP) + j] = e < n_env ? obs[(size_t)(env0 + e) * 8 + j] : 0.f;
    }
    __syncthreads();
    float t[EB];
    // zone_net_.4 (no activation, applied to the mean) is folded into combine_net_ by pack_f32: [obs; mean] -> c
#pragma unroll
    for (int e = 0; e < EB; ++e) va[e * (8 + HP) + 8 + j] = vb[e * HP + j];
    __syncthreads();
    matvec(t, w.wct, w.bc, va, 8 + HP, 8 + h, j);                          // combine_net_ -> embedding c
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EB; ++e) vb[e * HP + j] = live ? t[e] : 0.f;       // c
    __syncthreads();
    float hv[EB];                                                           // relu(critic.0(c))
#pragma unroll
    for (int e = 0; e < EB; ++e) hv[e] = 0.f;
    if (w.has_critic) matvec(hv, w.wv1t, w.bv1, vb, HP, h, j);
    matvec(t, w.wat, w.ba, vb, HP, h, j);                                  // actor.enc_.0.0
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        va[e * (8 + HP) + 8 + j] = live ? fmaxf(t[e], 0.f) : 0.f;          // a = relu(enc(c))
        vb[e * HP + j] = live ? fmaxf(hv[e], 0.f) : 0.f;                   // relu(critic.0(c))
    }
    __syncthreads();
    // ---- heads: 8 rows of (h + 1) floats -- mu_ 0-1, std_ 0-1 on a; critic.2 / critic_mu, critic_sigma on the critic
    // hidden; one (env, row) dot product per thread
    __shared__ float hd[EB * 8];
    if (j < EB * 8) {
        const int e = j >> 3, row = j & 7;
        const float *wr = w.heads + (size_t)row * (HP + 1);
        const float *x = row < 4 ? va + e * (8 + HP) + 8 : vb + e * HP;
        float s = wr[HP];
        for (int k = 0; k < h; ++k) s = __builtin_fmaf(wr[k], x[k], s);
        hd[j] = s;
    }
    __syncthreads();
    if (j < n_env) {
        const int env = env0 + j;
        const float *o = hd + 8 * j;
        const float v = w.has_critic ? o[4] : 0.f;
        if (w.has_critic) {
            value[env] = v;
            if (w.distributional && value_sigma) value_sigma[env] = softplus03(o[5]) + 1e-3f;
        }
        head_outputs(env, o[0], o[1], o[2], o[3], v, mu, stdv, act);
    }
}


// ------------------------------------------------------------------------------------------ zone part on the f32 MFMA
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = HP / 32;        // 6 output tiles of 32 features
constexpr int KS1 = KIN / 2;       // 8 k-steps of zone_net_.0 (K = 2 per v_mfma_f32_32x32x2_f32)
constexpr int KS2 = HP / 2;        // 96 k-steps of zone_net_.2
constexpr int kZoneWaves = 4;
constexpr int kMfmaMinEnvs = 10240;
constexpr int kSplitMinEnvs = 2048;     // k_mlp_zone_s3
constexpr size_t kZoneLds = (size_t)(NT * KS2 + NT * KS1) * 64 * sizeof(float);   // 159 744 B of the CU's 163 840

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Y = W [obs; X] for a 32-env group, all on registers: X (NT accumulator tiles, feature in registers, env on the lane) is
// the B operand in accumulator order, OBS_STEPS leading k-steps take the lane's obs values (natural order); fragments
// from memory, 16 per chunk, double-buffered like the W2 reads.  The constant-1 feature carries the bias through.
typedef __attribute__((address_space(1))) float gfloat;
template <int OBS_STEPS, bool RELU>
__device__ __forceinline__ void head_layer_f32m(const float *__restrict__ img, int lane, const f32x16 (&x)[NT],
                                                const float (&xo)[4], f32x16 (&y)[NT])
{
    static_assert(OBS_STEPS == 0 || OBS_STEPS == 4, "obs rides in 4 k-steps or not at all");
    // a running pointer, opaque to the compiler: a chunk's 16 fragments sit at immediate offsets 0 .. 3840 B from it.
    // (Indexed from the image base, every chunk wants its own 64-bit offset in SGPRs -- hundreds of them, hoisted
    // out of the loops, spilled into VGPR lanes, which then spill in turn: 4 800 scratch instructions.)
    const gfloat *pf = (const gfloat *)(img + lane);     // address space 1: global_load, not flat_load
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        f32x16 acc;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(pf));
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (OBS_STEPS) {
            float wo[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) wo[s] = pf[s * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma32(wo[s], xo[s], acc);
            pf += OBS_STEPS * 64;
            asm volatile("" : "+v"(pf));
        }
        float wa[2][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wa[0][i] = pf[i * 64];
#pragma unroll
        for (int c = 0; c < KS2 / 16; ++c) {
            pf += 16 * 64;
            asm volatile("" : "+v"(pf));
            if (c + 1 < KS2 / 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i) wa[(c + 1) & 1][i] = pf[i * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = mfma32(wa[c & 1][i], x[c][i], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) y[n][i] = RELU ? fmaxf(acc[i], 0.f) : acc[i];
    }
}
// the heads: one output tile whose first rows are the few outputs (register i of lane half 0 = row i for i < 4)
__device__ __forceinline__ f32x16 head_rows_f32m(const float *__restrict__ img, int lane, const f32x16 (&x)[NT])
{
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const gfloat *pf = (const gfloat *)(img + lane);
    float wa[2][16];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(pf));
#pragma unroll
    for (int i = 0; i < 16; ++i) wa[0][i] = pf[i * 64];
#pragma unroll
    for (int c = 0; c < KS2 / 16; ++c) {
        pf += 16 * 64;
        asm volatile("" : "+v"(pf));
        if (c + 1 < KS2 / 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) wa[(c + 1) & 1][i] = pf[i * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = mfma32(wa[c & 1][i], x[c][i], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// One wave per 64 envs, two groups of 32; a tile = zone t of the group's 32 envs (zone-major, like k_mlp_zone1), so
// column n of every tile is env slot n and the per-env sum over the zone rows is a register-wise add.
//   Y1[f][row] = sum_k W1[f][k] X0[k][row]        A = W1 image (natural k), B = the lane's input values
//   Y2[f][row] = sum_k W2[f][k] relu(Y1)[k][row]  A = W2 image (k in accumulator order), B = relu(Y1) registers
// Biases ride in a constant-1 slot (input k = 15, hidden feature h), as in the bf16 kernels.
__global__ __launch_bounds__(kZoneWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_zone_f32m(MlpF32 w, int N, int Z, int F, const float *__restrict__ obs, const float *__restrict__ zone_obs,
                     float *__restrict__ mu, float *__restrict__ stdv, float *__restrict__ value,
                     float *__restrict__ value_sigma, MlpAction act, int envs_per_wave)
{
    extern __shared__ __align__(16) float zl[];
    float *w2s = zl, *w1s = zl + NT * KS2 * 64;
    {
        const float4 *s2 = reinterpret_cast<const float4 *>(w.w2m), *s1 = reinterpret_cast<const float4 *>(w.w1m);
        float4 *d2 = reinterpret_cast<float4 *>(w2s), *d1 = reinterpret_cast<float4 *>(w1s);
        for (int i = threadIdx.x; i < NT * KS2 * 16; i += kZoneWaves * 64) d2[i] = s2[i];
        for (int i = threadIdx.x; i < NT * KS1 * 16; i += kZoneWaves * 64) d1[i] = s1[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int env0 = (blockIdx.x * kZoneWaves + wave) * envs_per_wave;     // 64, or 32 when that still fills the chip
    for (int e_base = 0; e_base < envs_per_wave && env0 + e_base < N; e_base += 32) {
        const bool valid = env0 + e_base + r < N;
        const int env = valid ? env0 + e_base + r : env0;
        // B operand of k-step s: X0[k = 2 s + h][row]; k 0..7 = the env's obs, 8..8+F-1 = the zone row, 15 = 1
        float xo[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xo[s] = valid ? obs[(size_t)env * 8 + 2 * s + h] : 0.f;
        f32x16 P[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[n][i] = 0.f;
        for (int t = 0; t < Z; ++t) {
            const float *row = zone_obs + ((size_t)env * Z + t) * F;
            float xz[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int jf = 2 * s + h;                       // zone feature index of this lane half
                xz[s] = (valid && jf < F) ? row[jf] : 0.f;
            }
            if (h == 1) xz[3] = 1.0f;                           // k = 15: the constant that carries the biases
            // ---- zone_net_.0 + ReLU: the results stay in registers as layer 2's B operands
            f32x16 a1[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int s = 0; s < KS1; ++s)
                    acc = mfma32(w1s[(n * KS1 + s) * 64 + lane], s < 4 ? xo[s] : xz[s - 4], acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) a1[n][i] = fmaxf(acc[i], 0.f);
            }
            // ---- zone_net_.2 + ReLU, summed over the tiles (= over the env's zone rows).  36 chunks of 16 k-steps; the
            // 16 weight fragments of chunk c + 1 are read from LDS before the 16 MFMAs of chunk c go out (explicit
            // double buffer between scheduling barriers: left alone, the scheduler hoists hundreds of the 576
            // independent reads and spills)
            float wa[2][16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wa[0][i] = w2s[i * 64 + lane];
            f32x16 acc;
#pragma unroll
            for (int c = 0; c < NT * (KS2 / 16); ++c) {
                const int n2 = c / (KS2 / 16), sc = c % (KS2 / 16);
                if (c + 1 < NT * (KS2 / 16)) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) wa[(c + 1) & 1][i] = w2s[((c + 1) * 16 + i) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (sc == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = mfma32(wa[c & 1][i], a1[sc][i], acc);
                if (sc == KS2 / 16 - 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) P[n2][i] += fmaxf(acc[i], 0.f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the per-env head, still on the matrix instruction: register i of lane (slot r, half h) of tile n is
        // feature 32 n + (i & 3) + 8 (i >> 2) + 4 h of env slot r -- the B operand of the next product as it stands.
        // These layers' images do not fit beside W2 in LDS: their fragments come from L2 (256 B per wave and MFMA).
        const float inv_z = 1.0f / (float)Z;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[n][i] *= inv_z;              // the mean; its feature h_dim is the constant 1
        f32x16 e3[NT], cc[NT];
        // zone_net_.4 (no activation, applied to the mean) is folded into combine_net_ by pack_f32: one layer
        head_layer_f32m<4, false>(w.wcm, lane, P, xo, cc);               // combine_net_([obs, zone_net_.4(mean)]) -> c
        float v_mu = 0.f, v_sigma = 0.f;
        if (w.has_critic) {
            head_layer_f32m<0, true>(w.wv1m, lane, cc, xo, e3);          // relu(critic.0(c))
            f32x16 hv = head_rows_f32m(w.whvm, lane, e3);                // rows 0 / 1: critic.2 or critic_mu / critic_sigma
            v_mu = hv[0];
            v_sigma = hv[1];
        }
        head_layer_f32m<0, true>(w.wam, lane, cc, xo, e3);               // a = relu(actor.enc_(c))
        const f32x16 hd = head_rows_f32m(w.whm, lane, e3);               // rows 0-1 mu_, 2-3 std_
        if (h == 0 && valid) {
            if (w.has_critic) {
                value[env] = v_mu;
                if (w.distributional && value_sigma) value_sigma[env] = softplus03(v_sigma) + 1e-3f;
            }
            head_outputs(env, hd[0], hd[1], hd[2], hd[3], v_mu, mu, stdv, act);
        }
    }
}

// ------------------------------------------------------------------------------------------ the network on split operands
// ZENV_MLP_BF16X3 / ZENV_MLP_F16X3: the 16-bit matrix instruction with every operand written as hi + lo
// (x_hi = round16(x), x_lo = round16(x - x_hi); the weights split the same way on the host) and three products per
// k-step, hi*hi + hi*lo + lo*hi, accumulated in float32 by the instruction.  v_mfma_f32_32x32x16_{bf16,f16} does 8 times
// the k depth of v_mfma_f32_32x32x2_f32 in half its cycles: 3 products cost 3/16 of the float32 kernel's matrix time.
//   bf16 halves: 8 + 8 = 16 significant bits, float32's exponent range.  scripts/split_bf16_frontier.py (torch
//     emulation against the float32 restatement of the reference's modules) with ALL layers split: max |d mu| 3.7e-6,
//     |d std| 2.0e-6, |d value| 7.8e-6 -- right at the 1e-5 the float32 mode is held to, so in this mode only the two
//     zone layers (96 % of the arithmetic) are split and the per-env head stays on the float32 matrix instruction
//     (head_layer_f32m); on the device: up to 1.1e-5 on the test matrix, stated as 2e-5.
//   f16 halves: 11 + 11 = 22 bits (subnormal lo halves are kept by the matrix pipe: scripts/probes/mfma_f16_denorm.hip),
//     the same emulation: 2.4e-7 / 1.8e-7 / 4.8e-7, the float32 accumulation's own noise -- so here the head is split
//     too.  The price is float16's range: an operand of 65 520 or more becomes inf.  Weights are checked when they are
//     loaded; activations are watched by the kernel (the running v_pk_max_f16 over every hi half it makes costs one
//     instruction per two values) and a hit sets MlpF32::range_flag, which the next synchronising call reports as
//     ZENV_E_RANGE.
//
// Same shape as k_mlp_zone_f32m: one wave per 64 envs (two groups of 32), tile t = zone t of the group's 32 envs,
// Y = W X with the feature in the accumulator registers and the env on the lane, so relu(Y1) -- split into hi / lo
// fragments, registers 0-7 -> k-step 2m, 8-15 -> k-step 2m + 1 of output tile m -- IS layer 2's B operand when W2's
// fragments are packed in that k order (pack_f32: k_b3), and the per-env sum over the zone rows is a register-wise add
// into P[].  LDS: W2 hi + lo (2 x 72 KiB) and W1 hi + lo (2 x 6 KiB) = 159 744 B, one workgroup per CU.
//
// The zone loop is software-pipelined by hand (zone_tile): while the matrix pipe works through one k-step pair of the
// second layer, the vector ALU -- idle otherwise -- makes the next pair from the first layer's next output tile (ReLU,
// hi / lo split) and folds finished accumulators into P; the zone row of tile t + 2 is already on its way from memory.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16v __attribute__((__vector_size__(16 * sizeof(float))));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gu32x4;
constexpr int KSB = HP / 16;                       // 12 k-steps (K = 16) per hidden layer
constexpr size_t kZoneLdsS3 = (size_t)(NT * KSB + NT) * 2 * 64 * 16;   // (72 + 6) fragments x {hi, lo} x 1 KiB = 159 744 B

template <bool F16>
__device__ __forceinline__ f32x16v mfma_s(const u32x4 a, const u32x4 b, const f32x16v c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// max(x, 0) as an integer maximum of the bit patterns (negative floats are negative integers; -0 is the most negative).
// fmaxf() -- and the median with 0 and +inf, which the compiler folds back into it -- first quiets its argument: a second
// v_max_f32 per value, 200 per zone tile on a vector ALU that has to keep up with the matrix pipe.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ f32x16v zero16()
{
    f32x16v z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// (a, b) -> one dword of their 16-bit roundings (hi) and one of the roundings of what is left (lo).  F16: mx keeps the
// largest |hi| seen (ABS false: the values are known to be >= 0)
template <bool F16, bool ABS>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t &hi, uint32_t &lo, uint32_t &mx)
{
    const f32x2v v = { a, b };
    if constexpr (F16) {
        const f16x2v hv = __builtin_convertvector(v, f16x2v);
        hi = __builtin_bit_cast(uint32_t, hv);
        const f32x2v r = v - __builtin_convertvector(hv, f32x2v);          // one v_pk_add_f32
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2v));
        const uint32_t mag = ABS ? (hi & 0x7FFF7FFFu) : hi;
        mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2v, mx), __builtin_bit_cast(f16x2v, mag)));
    } else {
        hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
        const f32x2v hf = { __uint_as_float(hi << 16), __uint_as_float(hi & 0xFFFF0000u) };
        const f32x2v r = v - hf;
        lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2v));
    }
}
template <bool F16, bool ABS>
__device__ __forceinline__ void split8(const float *v, u32x4 &hi, u32x4 &lo, uint32_t &mx)
{
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair<F16, ABS>(v[0], v[1], h0, l0, mx);
    split_pair<F16, ABS>(v[2], v[3], h1, l1, mx);
    split_pair<F16, ABS>(v[4], v[5], h2, l2, mx);
    split_pair<F16, ABS>(v[6], v[7], h3, l3, mx);
    hi = u32x4{ h0, h1, h2, h3 };
    lo = u32x4{ l0, l1, l2, l3 };
}
// relu (or not) of a finished accumulator tile -> its two hi / lo fragment pairs (registers 0-7, 8-15)
template <bool F16, bool RELU>
__device__ __forceinline__ void split_tile(const f32x16v acc, u32x4 &h0, u32x4 &l0, u32x4 &h1, u32x4 &l1, uint32_t &mx)
{
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = RELU ? relu1(acc[i]) : acc[i];
    split8<F16, !RELU>(v, h0, l0, mx);
    split8<F16, !RELU>(v + 8, h1, l1, mx);
}

// One zone tile, k-step-major: relu(Y1) is never held whole.  Step n turns output tile n of zone_net_.0 into the two
// hi / lo fragment pairs X = (h[2], l[2]) -- k-steps 2 n, 2 n + 1 of zone_net_.2 -- which go straight into all six of
// its output accumulators (6 chunks of 6 matrix instructions), while the vector ALU prepares step n + 1: zone_net_.0's
// tile n + 1 (for n = 5: tile 0 of the NEXT zone, from its input pair x0n) in chunk 0, its ReLU and split spread over
// chunks 1-5.  acc[] is finished at the end of the call; P[j] += relu(acc[j]) is done in chunk j - 1 of the next call,
// right before chunk j starts acc[j] afresh (P[0]: in the last two chunks of this one; the caller does the very last
// P[1..5]).  One wave per SIMD: nothing else hides the vector work, so every chunk's instructions are laid out
// explicitly -- per matrix instruction one LDS read and up to five vector instructions (sched_group_barrier) -- instead
// of the scheduler's choice, all vector work first and the matrix instructions back to back after it.
struct XFrag {
    u32x4 h[2], l[2];
};
typedef __attribute__((address_space(3))) u32x4 lu32x4;
// fragment pair f of an LDS image, half i (0 hi, 1 lo): 1 KiB each; three bases 60 KiB apart keep every offset inside
// ds_read_b128's 16-bit immediate
__device__ __forceinline__ u32x4 lds_frag(const lu32x4 *const (&lb)[3], int f, int i)
{
    const int off = (f * 2 + i) * 64, k = off / 3840;      // in 16-byte units: 3840 = 60 KiB
    return lb[k][off - k * 3840];
}
template <int MFMAS>
__device__ __forceinline__ void interleave()
{
#pragma unroll
    for (int m = 0; m < MFMAS; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one matrix instruction,
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read,
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // five vector instructions
    }
}
template <bool F16>
__device__ __forceinline__ void zone_tile(const lu32x4 *const (&lb)[3], int lane, XFrag &x, const u32x4 x0h,
                                          const u32x4 x0l, const u32x4 x0nh, const u32x4 x0nl, f32x16 (&P)[NT],
                                          f32x16v (&acc)[NT], u32x4 (&wa)[2][2][2], u32x4 (&w1)[2], uint32_t &mx)
{
    constexpr int W1F = NT * KSB;       // zone_net_.0's fragment pairs follow zone_net_.2's
    f32x16v a1 = zero16();
    XFrag xn;
#pragma unroll
    for (int c = 0; c < NT * NT; ++c) {
        const int n = c / NT, n2 = c % NT;          // step (= k-step pair) n, output tile n2
        {                                           // the next chunk's fragments (chunk 36 = chunk 0 of the next call)
            const int cn = (c + 1) % (NT * NT), nn = cn / NT, nn2 = cn % NT;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wa[(c + 1) & 1][q][0] = lds_frag(lb, nn2 * KSB + 2 * nn + q, 0);
                wa[(c + 1) & 1][q][1] = lds_frag(lb, nn2 * KSB + 2 * nn + q, 1);
            }
            if (nn2 == 0) {                         // ... and zone_net_.0's pair for the tile made in that chunk
                w1[0] = lds_frag(lb, W1F + (nn + 1) % NT, 0);
                w1[1] = lds_frag(lb, W1F + (nn + 1) % NT, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (constant trip counts with the chunk's share picked by a condition: bounds that depend on c would keep these
        // loops from unrolling before the chunk loop does, and the vectors they index would go to scratch)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const bool mine = j == 0 ? (n == NT - 1 && ((n2 == NT - 2 && i < 8) || (n2 == NT - 1 && i >= 8)))
                                         : (n == 0 && n2 == j - 1);
                if (mine) {
                    const f32x2v sum = f32x2v{ P[j][i], P[j][i + 1] } + f32x2v{ relu1(acc[j][i]), relu1(acc[j][i + 1]) };
                    P[j][i] = sum[0];
                    P[j][i + 1] = sum[1];
                }
            }
        if (n2 == 0) {                              // zone_net_.0's next output tile
            const u32x4 ah = w1[0], al = w1[1];
            const u32x4 bh = n + 1 < NT ? x0h : x0nh, bl = n + 1 < NT ? x0l : x0nl;
            a1 = mfma_s<F16>(ah, bh, zero16());
            a1 = mfma_s<F16>(ah, bl, a1);
            a1 = mfma_s<F16>(al, bh, a1);
        }
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {            // its ReLU and split: value pairs 0-1, 2-3, 4-5, 6, 7 in chunks 1-5
            const int at = pr < 6 ? 1 + pr / 2 : pr - 2;
            if (n2 == at) {
                uint32_t hi, lo;
                split_pair<F16, false>(relu1(a1[2 * pr]), relu1(a1[2 * pr + 1]), hi, lo, mx);
                xn.h[pr >> 2][pr & 3] = hi;
                xn.l[pr >> 2][pr & 3] = lo;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            acc[n2] = mfma_s<F16>(wa[c & 1][q][0], x.h[q], (n == 0 && q == 0) ? zero16() : acc[n2]);
            acc[n2] = mfma_s<F16>(wa[c & 1][q][0], x.l[q], acc[n2]);
            acc[n2] = mfma_s<F16>(wa[c & 1][q][1], x.h[q], acc[n2]);
        }
        if (n2 == 0) interleave<9>(); else interleave<6>();
        if (n2 == NT - 1) x = xn;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Y = W [obs; X] for a 32-env group with X given as hi / lo fragments and Y returned the same way: TILES output tiles
// of KSB (+ 1 leading, the obs in natural order) k-steps, three matrix instructions each.  The fragments come from L2
// through a ring of D k-steps (2 KiB each per wave), loaded D steps ahead across tile boundaries; done(n, acc) takes
// every finished tile.
template <bool F16, bool OBS, int TILES, typename Done>
__device__ __forceinline__ void head_product(const void *img, int lane, const u32x4 (&xh)[KSB], const u32x4 (&xl)[KSB],
                                             const u32x4 xoh, const u32x4 xol, Done done)
{
    constexpr int KT = KSB + (OBS ? 1 : 0), S = TILES * KT, D = 12;
    const gu32x4 *pl = (const gu32x4 *)img + lane;       // runs D steps ahead of the arithmetic; opaque to the compiler so
    u32x4 ring[D][2];                                    // that every load is (pointer, immediate offset)
#pragma unroll
    for (int g = 0; g < D && g < S; ++g) {
        ring[g][0] = pl[0];
        ring[g][1] = pl[64];
        pl += 128;
        asm volatile("" : "+v"(pl));
    }
    f32x16v acc = zero16();
#pragma unroll
    for (int g = 0; g < S; ++g) {
        const int n = g / KT, s = g % KT;
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 bh = (OBS && s == 0) ? xoh : xh[s - (OBS ? 1 : 0)], bl = (OBS && s == 0) ? xol : xl[s - (OBS ? 1 : 0)];
        acc = mfma_s<F16>(ring[g % D][0], bh, s == 0 ? zero16() : acc);
        acc = mfma_s<F16>(ring[g % D][0], bl, acc);
        acc = mfma_s<F16>(ring[g % D][1], bh, acc);
        if (g + D < S) {
            ring[g % D][0] = pl[0];
            ring[g % D][1] = pl[64];
            pl += 128;
            asm volatile("" : "+v"(pl));
        }
        if (s == KT - 1) done(n, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool F16, bool OBS, bool RELU>
__device__ __forceinline__ void head_layer_s3(const void *img, int lane, const u32x4 (&xh)[KSB], const u32x4 (&xl)[KSB],
                                              const u32x4 xoh, const u32x4 xol, u32x4 (&yh)[KSB], u32x4 (&yl)[KSB],
                                              uint32_t &mx)
{
    head_product<F16, OBS, NT>(img, lane, xh, xl, xoh, xol, [&](int n, const f32x16v acc) __attribute__((always_inline)) {
        split_tile<F16, RELU>(acc, yh[2 * n], yl[2 * n], yh[2 * n + 1], yl[2 * n + 1], mx);
    });
}
template <bool F16>
__device__ __forceinline__ f32x16v head_rows_s3(const void *img, int lane, const u32x4 (&xh)[KSB], const u32x4 (&xl)[KSB])
{
    f32x16v out = zero16();
    const u32x4 none = { 0u, 0u, 0u, 0u };
    head_product<F16, false, 1>(img, lane, xh, xl, none, none, [&](int, const f32x16v acc) __attribute__((always_inline)) { out = acc; });
    return out;
}

template <bool F16>
__global__ __launch_bounds__(kZoneWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_zone_s3(MlpF32 w, int N, int Z, int F, const float *__restrict__ obs, const float *__restrict__ zone_obs,
                   float *__restrict__ mu, float *__restrict__ stdv, float *__restrict__ value,
                   float *__restrict__ value_sigma, MlpAction act, int envs_per_wave)
{
    extern __shared__ __align__(16) u32x4 zs[];
    // zs: zone_net_.2's fragment pairs [NT][KSB][{hi, lo}][64], then zone_net_.0's [NT][{hi, lo}][64]
    {
        // the images sit back to back in memory in the order they have in LDS (pack_f32): one copy of 39 x 4 KiB, 13 loads
        // in flight per thread (one after the other, each round trip to L2 would cost as much as eight zone tiles)
        const gu32x4 *src = (const gu32x4 *)(F16 ? w.w2h : w.w2b) + threadIdx.x;
        constexpr int PER_THREAD = (NT * KSB + NT) * 2 * 64 / (kZoneWaves * 64), BATCH = 13;
        static_assert(PER_THREAD % BATCH == 0, "39 = 3 x 13");
#pragma unroll 1
        for (int b0 = 0; b0 < PER_THREAD; b0 += BATCH) {
            u32x4 v[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) v[k] = src[(b0 + k) * kZoneWaves * 64];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) zs[(b0 + k) * kZoneWaves * 64 + threadIdx.x] = v[k];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int env0 = (blockIdx.x * kZoneWaves + wave) * envs_per_wave;
    // the lane's window on the images: three bases, opaque to the compiler so that it keeps them (see lds_frag)
    const lu32x4 *lb0 = (const lu32x4 *)zs + lane, *lb1 = lb0 + 3840, *lb2 = lb0 + 7680;
    asm volatile("" : "+v"(lb0), "+v"(lb1), "+v"(lb2));
    const lu32x4 *const lb[3] = { lb0, lb1, lb2 };
    constexpr int W1F = NT * KSB;
    uint32_t mx = 0u;                             // F16: the largest |hi half| this lane has made, as two f16
    for (int e_base = 0; e_base < envs_per_wave && env0 + e_base < N; e_base += 32) {
        const bool valid = env0 + e_base + r < N;
        const int env = valid ? env0 + e_base + r : env0;
        // the float32 head's obs operand (natural order, K = 2 per step) ...
        float xo[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xo[s] = valid ? obs[(size_t)env * 8 + 2 * s + h] : 0.f;
        // ... and the split products': lane half 0 carries k = 0..7 = the env's obs, half 1 nothing
        u32x4 xoh = { 0u, 0u, 0u, 0u }, xol = { 0u, 0u, 0u, 0u };
        if (h == 0 && valid) {
            const float4 *o4 = reinterpret_cast<const float4 *>(obs + (size_t)env * 8);
            const float4 a = o4[0], b = o4[1];
            const float v8[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
            split8<F16, true>(v8, xoh, xol, mx);
        }
        // layer 1's B operand of tile t: k = 8 h + j; half 0 = the obs pair above, half 1 = the zone row (k = 8 ..
        // 8 + F - 1), zeros, the constant 1 (k = 15).  Rows are fetched two tiles ahead of their use.
        const float *rows = zone_obs + (size_t)env * Z * F;
        auto fetch_row = [&](int t, float (&v)[8]) {
            const float *row = rows + (size_t)(t < Z ? t : Z - 1) * F;
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = (h == 1 && valid && j < F) ? row[j] : 0.f;
            v[7] = 1.0f;
        };
        auto input_pair = [&](const float (&v)[8], u32x4 &x0h, u32x4 &x0l) {
            u32x4 zh, zl;
            split8<F16, true>(v, zh, zl, mx);
            x0h = h == 1 ? zh : xoh;
            x0l = h == 1 ? zl : xol;
        };
        f32x16 P[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[n][i] = 0.f;
        float rowv[8];
        u32x4 x0h, x0l, x0nh, x0nl;
        fetch_row(0, rowv);
        input_pair(rowv, x0h, x0l);
        fetch_row(1, rowv);
        XFrag x;
        {
            const u32x4 wh = lds_frag(lb, W1F, 0), wl = lds_frag(lb, W1F, 1);
            f32x16v a = mfma_s<F16>(wh, x0h, zero16());
            a = mfma_s<F16>(wh, x0l, a);
            split_tile<F16, true>(mfma_s<F16>(wl, x0h, a), x.h[0], x.l[0], x.h[1], x.l[1], mx);
        }
        f32x16v acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = zero16();
        u32x4 wa[2][2][2];                 // [buffer][k-step of the pair][hi, lo], chunk 0 to start with
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wa[0][q][0] = lds_frag(lb, q, 0);
            wa[0][q][1] = lds_frag(lb, q, 1);
        }
        u32x4 w1[2] = { lds_frag(lb, W1F + 1, 0), lds_frag(lb, W1F + 1, 1) };   // chunk 0 makes zone_net_.0's tile 1
        for (int t = 0; t < Z; ++t) {
            input_pair(rowv, x0nh, x0nl);                // zone t + 1's (the last one: a clamped row, never used)
            fetch_row(t + 2, rowv);
            zone_tile<F16>(lb, lane, x, x0h, x0l, x0nh, x0nl, P, acc, wa, w1, mx);
            x0h = x0nh;
            x0l = x0nl;
        }
        const float inv_z = 1.0f / (float)Z;
#pragma unroll
        for (int j = 1; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[j][i] += relu1(acc[j][i]);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[n][i] *= inv_z;              // the mean; its feature h_dim is the constant 1
        u32x4 ah[KSB], al[KSB], bh[KSB], bl[KSB];
        float v_mu = 0.f, v_sigma = 0.f, m0, m1, s0, s1;
        if constexpr (F16) {
            // ---- the per-env head on split operands as well: zone_net_.4 folded into combine_net_ (pack_f32), critic.0 +
            // critic heads, actor.enc_ + actor heads; (ah, al) / (bh, bl) are free again
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x16v pv;
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[i] = P[n][i];
                split_tile<true, true>(pv, ah[2 * n], al[2 * n], ah[2 * n + 1], al[2 * n + 1], mx);    // means of ReLUs: >= 0
            }
            head_layer_s3<true, true, false>(w.wch, lane, ah, al, xoh, xol, bh, bl, mx);              // c
            if (w.has_critic) {
                head_layer_s3<true, false, true>(w.wv1h, lane, bh, bl, xoh, xol, ah, al, mx);         // relu(critic.0(c))
                const f32x16v hv = head_rows_s3<true>(w.whvh, lane, ah, al);
                v_mu = hv[0];
                v_sigma = hv[1];
            }
            head_layer_s3<true, false, true>(w.wah, lane, bh, bl, xoh, xol, ah, al, mx);              // relu(actor.enc_(c))
            const f32x16v hd = head_rows_s3<true>(w.whh, lane, ah, al);
            m0 = hd[0], m1 = hd[1], s0 = hd[2], s1 = hd[3];
        } else {
            // ---- the per-env head on the float32 matrix instruction, exactly as in k_mlp_zone_f32m
            f32x16 e3[NT], cc[NT];
            head_layer_f32m<4, false>(w.wcm, lane, P, xo, cc);
            if (w.has_critic) {
                head_layer_f32m<0, true>(w.wv1m, lane, cc, xo, e3);
                const f32x16 hv = head_rows_f32m(w.whvm, lane, e3);
                v_mu = hv[0];
                v_sigma = hv[1];
            }
            head_layer_f32m<0, true>(w.wam, lane, cc, xo, e3);
            const f32x16 hd = head_rows_f32m(w.whm, lane, e3);
            m0 = hd[0], m1 = hd[1], s0 = hd[2], s1 = hd[3];
        }
        if (h == 0 && valid) {
            if (w.has_critic) {
                value[env] = v_mu;
                if (w.distributional && value_sigma) value_sigma[env] = softplus03(v_sigma) + 1e-3f;
            }
            head_outputs(env, m0, m1, s0, s1, v_mu, mu, stdv, act);
        }
    }
    if constexpr (F16) {
        // an operand of 65 520 or more became inf on its way to float16: tell the host (pinned memory; read at its next sync)
        if (((mx & 0x7FFFu) >= 0x7C00u || (mx >> 16 & 0x7FFFu) >= 0x7C00u) && w.range_flag) *w.range_flag = 1;
    }
}
}  // namespace

size_t pack_f32(const zenv_mlp_weights &w, int F, std::vector<float> &out, size_t offs[30])
{
    const int h = w.h_dim;
    out.clear();
    auto transposed = [&](const float *W, int n_out, int n_in, int in_rows, int in_offset) {
        // W [n_out][n_in] row-major -> [in_rows][HP], input k placed at row in_offset + k
        const size_t at = out.size();
        out.resize(at + (size_t)in_rows * HP, 0.f);
        for (int o = 0; o < n_out; ++o)
            for (int k = 0; k < n_in; ++k) out[at + (size_t)(in_offset + k) * HP + o] = W[(size_t)o * n_in + k];
        return at;
    };
    auto bias = [&](const float *b, int n) {
        const size_t at = out.size();
        out.resize(at + HP, 0.f);
        for (int o = 0; o < n; ++o) out[at + o] = b[o];
        return at;
    };
    // zone_net_.4 folded into combine_net_ (no activation between them; the mean commutes with zone_net_.4):
    //   c = Wc [obs; W3 m + b3] + bc = Wc_obs obs + (Wc_emb W3) m + (Wc_emb b3 + bc), products in float64
    std::vector<float> combf((size_t)h * (8 + h)), bcf(h);
    for (int i = 0; i < h; ++i) {
        const float *ci = w.comb_w + (size_t)i * (8 + h);
        for (int k = 0; k < 8; ++k) combf[(size_t)i * (8 + h) + k] = ci[k];
        for (int j = 0; j < h; ++j) {
            double a = 0.0;
            for (int k = 0; k < h; ++k) a += (double)ci[8 + k] * (double)w.zone_w3[(size_t)k * h + j];
            combf[(size_t)i * (8 + h) + 8 + j] = (float)a;
        }
        double b = (double)w.comb_b[i];
        for (int k = 0; k < h; ++k) b += (double)ci[8 + k] * (double)w.zone_b3[k];
        bcf[i] = (float)b;
    }
    offs[0] = transposed(w.zone_w1, h, 8 + F, KIN, 0);   offs[1] = bias(w.zone_b1, h);
    offs[2] = transposed(w.zone_w2, h, h, HP, 0);        offs[3] = bias(w.zone_b2, h);
    offs[4] = transposed(w.zone_w3, h, h, HP, 0);        offs[5] = bias(w.zone_b3, h);
    offs[6] = transposed(combf.data(), h, 8 + h, 8 + HP, 0); offs[7] = bias(bcf.data(), h);   // [obs (8), zone mean (h)]
    offs[8] = transposed(w.enc_w, h, h, HP, 0);          offs[9] = bias(w.enc_b, h);
    if (w.critic_w1) {
        offs[10] = transposed(w.critic_w1, h, h, HP, 0);
        offs[11] = bias(w.critic_b1, h);
    } else {
        offs[10] = offs[11] = 0;
    }
    offs[12] = out.size();
    out.resize(out.size() + 8 * (size_t)(HP + 1), 0.f);
    auto head_row = [&](int row, const float *W, const float *b) {
        for (int k = 0; k < h; ++k) out[offs[12] + (size_t)row * (HP + 1) + k] = W[k];
        out[offs[12] + (size_t)row * (HP + 1) + HP] = b[0];
    };
    head_row(0, w.mu_w, w.mu_b);          head_row(1, w.mu_w + h, w.mu_b + 1);
    head_row(2, w.std_w, w.std_b);        head_row(3, w.std_w + h, w.std_b + 1);
    if (w.critic_w1) head_row(4, w.critic_w2, w.critic_b2);
    if (w.critic_w1 && w.critic_sigma_w) head_row(5, w.critic_sigma_w, w.critic_sigma_b);
    // ---- images of the two zone layers for k_mlp_zone_f32m: fragment (tile n, k-step s) = 64 floats, lane (r, hh) holds
    // W[32 n + r][k(s, hh)].  zone_net_.0: natural k = 2 s + hh (k 15 = the bias slot; row h_dim keeps the constant 1);
    // zone_net_.2: k in the accumulator order of a 32x32 result, k(s, hh) = 32 (s / 16) + (i & 3) + 8 (i >> 2) + 4 hh
    // with i = s % 16, column h_dim = the bias.
    auto w1ext = [&](int o, int k) -> float {
        if (o == h) return k == 15 ? 1.f : 0.f;
        if (o > h) return 0.f;
        if (k == 15) return w.zone_b1[o];
        return k < 8 + F ? w.zone_w1[(size_t)o * (8 + F) + k] : 0.f;
    };
    auto w2ext = [&](int o, int k) -> float {
        if (o == h) return k == h ? 1.f : 0.f;
        if (o > h || k > h) return 0.f;
        return k == h ? w.zone_b2[o] : w.zone_w2[(size_t)o * h + k];
    };
    offs[13] = out.size();
    for (int n = 0; n < NT; ++n)
        for (int s = 0; s < KS1; ++s)
            for (int lane = 0; lane < 64; ++lane) out.push_back(w1ext(32 * n + (lane & 31), 2 * s + (lane >> 5)));
    auto k_acc = [](int s, int hh) { const int i = s % 16; return 32 * (s / 16) + (i & 3) + 8 * (i >> 2) + 4 * hh; };
    offs[14] = out.size();
    for (int n = 0; n < NT; ++n)
        for (int s = 0; s < KS2; ++s)
            for (int lane = 0; lane < 64; ++lane) out.push_back(w2ext(32 * n + (lane & 31), k_acc(s, lane >> 5)));
    // ---- the head layers as images of the same kind (input k in accumulator order; the constant-1 feature h_dim of
    // every hidden vector carries the bias and is passed on by row h_dim): zone_net_.4, combine_net_ (4 leading
    // k-steps = obs in natural order), critic.0, actor.enc_.0.0; the two head images have 32 rows of which the first
    // few are outputs
    auto hidden = [&](const float *W, const float *b, int in_off, int in_stride) {
        return [=](int o, int k) -> float {
            if (o == h) return k == h ? 1.f : 0.f;
            if (o > h || k > h) return 0.f;
            return k == h ? b[o] : W[(size_t)o * in_stride + in_off + k];
        };
    };
    auto pack_hidden = [&](auto ext, const float *Wobs, int in_stride) {
        const size_t at = out.size();
        for (int n = 0; n < NT; ++n) {
            if (Wobs)
                for (int s = 0; s < 4; ++s)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int o = 32 * n + (lane & 31), k = 2 * s + (lane >> 5);
                        out.push_back(o < h ? Wobs[(size_t)o * in_stride + k] : 0.f);
                    }
            for (int s = 0; s < KS2; ++s)
                for (int lane = 0; lane < 64; ++lane) out.push_back(ext(32 * n + (lane & 31), k_acc(s, lane >> 5)));
        }
        return at;
    };
    auto pack_rows = [&](std::initializer_list<std::pair<const float *, const float *>> rows) {
        const size_t at = out.size();
        std::vector<std::pair<const float *, const float *>> rv(rows);
        for (int s = 0; s < KS2; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = lane & 31, k = k_acc(s, lane >> 5);
                float v = 0.f;
                if (o < (int)rv.size() && rv[o].first && k <= h) v = k == h ? rv[o].second[0] : rv[o].first[k];
                out.push_back(v);
            }
        return at;
    };
    offs[15] = pack_hidden(hidden(w.zone_w3, w.zone_b3, 0, h), nullptr, 0);
    offs[16] = pack_hidden(hidden(combf.data(), bcf.data(), 8, 8 + h), combf.data(), 8 + h);
    offs[17] = pack_hidden(hidden(w.enc_w, w.enc_b, 0, h), nullptr, 0);
    offs[18] = pack_rows({ { w.mu_w, w.mu_b }, { w.mu_w + h, w.mu_b + 1 }, { w.std_w, w.std_b }, { w.std_w + h, w.std_b + 1 } });
    offs[19] = offs[20] = 0;
    if (w.critic_w1) {
        offs[19] = pack_hidden(hidden(w.critic_w1, w.critic_b1, 0, h), nullptr, 0);
        offs[20] = pack_rows({ { w.critic_w2, w.critic_b2 }, { w.critic_sigma_w, w.critic_sigma_b } });
    }
    // ---- ZENV_MLP_BF16X3 / ZENV_MLP_F16X3: hi / lo fragment pairs for k_mlp_zone_s3 (8 halves = 16 bytes per lane and
    // fragment, stored here as 4 float-sized words; a pair = 64 x 16 B of hi, then 64 x 16 B of lo).  A operand of
    // v_mfma_f32_32x32x16_*: lane (m, hh) element j = W[32 n + m][k]; zone_net_.0 and the obs step of combine_net_:
    // k = 8 hh + j (natural); everything else: the k order in which the previous layer's accumulator registers arrive --
    // k-step s = 2 t + q takes registers 8 q .. 8 q + 7 of tile t:
    //   k_b3(s, hh, j) = 32 (s / 2) + 16 (s % 2) + (j & 3) + 8 (j >> 2) + 4 hh
    auto bf16_rne = [](float x) -> uint16_t {
        uint32_t u;
        std::memcpy(&u, &x, 4);
        if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);       // inf / nan: truncate
        return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    };
    auto bf16_val = [](uint16_t b) -> float {
        const uint32_t u = (uint32_t)b << 16;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    auto f16_rne = [](float x) -> uint16_t {        // round to nearest even, subnormals kept, 65 520 and up -> inf
        uint32_t u;
        std::memcpy(&u, &x, 4);
        const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
        u &= 0x7FFFFFFFu;
        if (u >= 0x47800000u) return (uint16_t)(sign | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));
        if (u < 0x38800000u) {                                                  // below 2^-14: a multiple of 2^-24
            float f;
            std::memcpy(&f, &u, 4);
            return (uint16_t)(sign | (uint16_t)std::lrintf(f * 16777216.0f));   // 1024 = the smallest normal's encoding
        }
        u += 0xC8000FFFu + ((u >> 13) & 1u);                                    // exponent bias 127 -> 15, half-ulp - 1 + odd
        return (uint16_t)(sign | (u >> 13));
    };
    auto f16_val = [](uint16_t b) -> float {
        const int e = (b >> 10) & 31, m = b & 1023;
        const float mag = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : std::ldexp((float)(1024 + m), e - 25);
        return (b & 0x8000u) ? -mag : mag;
    };
    auto push_frag = [&](bool f16, auto ext, int n, auto kof) {       // one fragment pair (hi, lo): 2 x 64 lanes x 8 halves
        std::vector<uint16_t> hi(64 * 8), lo(64 * 8);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const float v = ext(32 * n + (lane & 31), kof(lane >> 5, j));
                const uint16_t hb = f16 ? f16_rne(v) : bf16_rne(v);
                hi[lane * 8 + j] = hb;
                lo[lane * 8 + j] = f16 ? f16_rne(v - f16_val(hb)) : bf16_rne(v - bf16_val(hb));
            }
        const size_t at = out.size();
        out.resize(at + 2 * 64 * 4);
        std::memcpy(&out[at], hi.data(), 64 * 16);
        std::memcpy(&out[at + 64 * 4], lo.data(), 64 * 16);
    };
    auto k_nat = [](int hh, int j) { return 8 * hh + j; };
    auto k_b3 = [](int s2) {
        return [s2](int hh, int j) { return 32 * (s2 / 2) + 16 * (s2 % 2) + (j & 3) + 8 * (j >> 2) + 4 * hh; };
    };
    for (int f16 = 0; f16 < 2; ++f16) {         // zone_net_.2 then zone_net_.0, back to back: the kernel's LDS image in one copy
        offs[22 + 2 * f16] = out.size();
        for (int n = 0; n < NT; ++n)
            for (int s2 = 0; s2 < HP / 16; ++s2) push_frag(f16, w2ext, n, k_b3(s2));
        offs[21 + 2 * f16] = out.size();
        for (int n = 0; n < NT; ++n) push_frag(f16, w1ext, n, k_nat);
    }
    // the head layers (ZENV_MLP_F16X3 only): [tile][k-step]; combine_net_ (zone_net_.4 folded in) has one leading k-step
    // whose lane half 0 is the obs in natural order (half 1: nothing); the two head images are one tile whose first rows
    // are the outputs
    auto pack_hidden16 = [&](auto ext, const float *Wobs, int in_stride) {
        const size_t at = out.size();
        for (int n = 0; n < NT; ++n) {
            if (Wobs)
                push_frag(true, [&](int o, int k) { return (o < h && k < 8) ? Wobs[(size_t)o * in_stride + k] : 0.f; }, n, k_nat);
            for (int s2 = 0; s2 < HP / 16; ++s2) push_frag(true, ext, n, k_b3(s2));
        }
        return at;
    };
    auto pack_rows16 = [&](std::initializer_list<std::pair<const float *, const float *>> rows) {
        const size_t at = out.size();
        std::vector<std::pair<const float *, const float *>> rv(rows);
        auto ext = [&](int o, int k) -> float {
            if (o < (int)rv.size() && rv[o].first && k <= h) return k == h ? rv[o].second[0] : rv[o].first[k];
            return 0.f;
        };
        for (int s2 = 0; s2 < HP / 16; ++s2) push_frag(true, ext, 0, k_b3(s2));
        return at;
    };
    offs[25] = pack_hidden16(hidden(combf.data(), bcf.data(), 8, 8 + h), combf.data(), 8 + h);
    offs[27] = pack_hidden16(hidden(w.enc_w, w.enc_b, 0, h), nullptr, 0);
    offs[28] = pack_rows16({ { w.mu_w, w.mu_b }, { w.mu_w + h, w.mu_b + 1 }, { w.std_w, w.std_b }, { w.std_w + h, w.std_b + 1 } });
    offs[26] = offs[29] = 0;
    if (w.critic_w1) {
        offs[26] = pack_hidden16(hidden(w.critic_w1, w.critic_b1, 0, h), nullptr, 0);
        offs[29] = pack_rows16({ { w.critic_w2, w.critic_b2 }, { w.critic_sigma_w, w.critic_sigma_b } });
    }
    return out.size();
}

hipError_t launch_mlp_forward_f32(const MlpF32 &w, int N, int Z, int F, const float *obs, const float *zone_obs, float *mu,
                                  float *stdv, float *value, float *value_sigma, const MlpAction &act, hipStream_t s)
{
    // A wave of the MFMA kernel works through its envs' zone tiles one group of 32 after the other: ~0.55 ms per group
    // whatever N is, as long as there is at most one wave per SIMD -- one group per wave up to N = 32 768, two above.
    // Small batches (evaluate(): 500 envs) are faster on the vector-ALU kernel, which spreads 4 envs per workgroup over
    // the chip (0.50 ms at N = 8 192, 0.86 ms at 16 384): crossover N ~ 10 000.
    // (a wave of the split-operand kernel is through its 25 tiles in ~0.16 ms: it overtakes the vector kernel near 2 000 envs)
    if (w.split3 && (w.on_mfma == 2 || (w.on_mfma == 1 && N >= kSplitMinEnvs))) {
        // ZENV_MLP_BF16X3 / ZENV_MLP_F16X3: three 16-bit products per k-step on hi / lo operands
        const int epw = N <= 32768 ? 32 : 64;
        const dim3 grid((N + kZoneWaves * epw - 1) / (kZoneWaves * epw)), block(kZoneWaves * 64);
        if (w.split3 == 2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_zone_s3<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZoneLdsS3);
            hipLaunchKernelGGL(k_mlp_zone_s3<true>, grid, block, kZoneLdsS3, s, w, N, Z, F, obs, zone_obs, mu, stdv, value,
                               value_sigma, act, epw);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_zone_s3<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZoneLdsS3);
            hipLaunchKernelGGL(k_mlp_zone_s3<false>, grid, block, kZoneLdsS3, s, w, N, Z, F, obs, zone_obs, mu, stdv, value,
                               value_sigma, act, epw);
        }
        return hipGetLastError();
    }
    if (w.on_mfma == 2 || (w.on_mfma == 1 && N >= kMfmaMinEnvs)) {
        // the whole network on the float32 matrix instruction (one workgroup of 4 waves per CU, the two zone layers'
        // images in LDS)
        const int epw = N <= 32768 ? 32 : 64;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_zone_f32m),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZoneLds);
        hipLaunchKernelGGL(k_mlp_zone_f32m, dim3((N + kZoneWaves * epw - 1) / (kZoneWaves * epw)), dim3(kZoneWaves * 64),
                           kZoneLds, s, w, N, Z, F, obs, zone_obs, mu, stdv, value, value_sigma, act, epw);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_mlp_f32, dim3((N + EB - 1) / EB), dim3(HP), 0, s, w, N, Z, F, obs, zone_obs, mu, stdv, value,
                       value_sigma, act, nullptr);
    return hipGetLastError();
}

}  // namespace zenvk
