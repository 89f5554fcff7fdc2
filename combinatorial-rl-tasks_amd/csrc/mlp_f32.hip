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

// ------------------------------------------------------------------------------------------ zone part on bf16 x 3
// ZENV_MLP_BF16X3: the two zone layers -- 96 % of the network's arithmetic -- on the bf16 matrix instruction with every
// operand written as hi + lo (x_hi = bf16(x), x_lo = bf16(x - x_hi); the weights split the same way on the host) and
// three products per k-step, hi*hi + hi*lo + lo*hi, accumulated in float32 by the instruction: 16 significant bits per
// operand instead of 8.  scripts/split_bf16_frontier.py (torch emulation against the float32 restatement of the
// reference's modules): max |d mu| 3.7e-6, |d std| 2.0e-6, |d value| 7.8e-6 with ALL layers split -- inside the 1e-5 the
// float32 mode is held to -- and the per-env head here stays on the float32 matrix instruction (head_layer_f32m), so
// only the zone layers carry the split's error.  v_mfma_f32_32x32x16_bf16 does 8 times the k depth of
// v_mfma_f32_32x32x2_f32 in half its cycles: 3 products cost 3/16 of the float32 kernel's matrix time.
//
// Same shape as k_mlp_zone_f32m: one wave per 64 envs (two groups of 32), tile t = zone t of the group's 32 envs,
// Y = W X with the feature in the accumulator registers and the env on the lane, so relu(Y1) -- split into hi / lo
// fragments, registers 0-7 -> k-step 2m, 8-15 -> k-step 2m + 1 of output tile m -- IS layer 2's B operand when W2's
// fragments are packed in that k order (pack_f32: k_b3), and the per-env sum over the zone rows is a register-wise add
// into P[], which the float32 head takes over as it stands.  LDS: W2 hi + lo (2 x 72 KiB) and W1 hi + lo (2 x 6 KiB) =
// 159 744 B, one workgroup per CU.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x16v __attribute__((__vector_size__(16 * sizeof(float))));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
constexpr int KSB = HP / 16;                       // 12 k-steps (K = 16) per hidden layer
constexpr size_t kZoneLdsB3 = (size_t)(NT * KSB + NT) * 2 * 64 * 16;   // (72 + 6) fragments x {hi, lo} x 1 KiB = 159 744 B

__device__ __forceinline__ f32x16v mfma_bf(const uint4 a, const uint4 b, const f32x16v c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// (a, b) -> one dword of their bf16 roundings (hi) and one of the roundings of what is left (lo)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t &hi, uint32_t &lo)
{
    const f32x2v v = { a, b };
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xFFFF0000u);
    const f32x2v r = { a - ah, b - bh };
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2v));
}
__device__ __forceinline__ void split8(const float *v, uint4 &hi, uint4 &lo)
{
    split_pair(v[0], v[1], hi.x, lo.x);
    split_pair(v[2], v[3], hi.y, lo.y);
    split_pair(v[4], v[5], hi.z, lo.z);
    split_pair(v[6], v[7], hi.w, lo.w);
}

__global__ __launch_bounds__(kZoneWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_zone_b3(MlpF32 w, int N, int Z, int F, const float *__restrict__ obs, const float *__restrict__ zone_obs,
                   float *__restrict__ mu, float *__restrict__ stdv, float *__restrict__ value,
                   float *__restrict__ value_sigma, MlpAction act, int envs_per_wave)
{
    extern __shared__ __align__(16) uint4 zb[];
    uint4 *w2s = zb;                              // [NT][KSB][2][64]   fragment (n, s): hi, then lo
    uint4 *w1s = zb + NT * KSB * 2 * 64;          // [NT][2][64]
    {
        const uint4 *s2 = reinterpret_cast<const uint4 *>(w.w2b), *s1 = reinterpret_cast<const uint4 *>(w.w1b);
        for (int i = threadIdx.x; i < NT * KSB * 2 * 64; i += kZoneWaves * 64) w2s[i] = s2[i];
        for (int i = threadIdx.x; i < NT * 2 * 64; i += kZoneWaves * 64) w1s[i] = s1[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int env0 = (blockIdx.x * kZoneWaves + wave) * envs_per_wave;
    for (int e_base = 0; e_base < envs_per_wave && env0 + e_base < N; e_base += 32) {
        const bool valid = env0 + e_base + r < N;
        const int env = valid ? env0 + e_base + r : env0;
        // the float32 head's obs operand (natural order, K = 2 per step) ...
        float xo[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xo[s] = valid ? obs[(size_t)env * 8 + 2 * s + h] : 0.f;
        // ... and layer 1's: lane half 0 carries k = 0..7 = the env's obs (the same for every tile of the group)
        uint4 xoh = make_uint4(0u, 0u, 0u, 0u), xol = make_uint4(0u, 0u, 0u, 0u);
        if (h == 0 && valid) {
            const float4 *o4 = reinterpret_cast<const float4 *>(obs + (size_t)env * 8);
            const float4 a = o4[0], b = o4[1];
            const float v8[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
            split8(v8, xoh, xol);
        }
        f32x16 P[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[n][i] = 0.f;
        for (int t = 0; t < Z; ++t) {
            // ---- layer 1's B operand: k = 8 h + j; half 1 = the zone row (k = 8 .. 8 + F - 1), zeros, the constant 1 (k = 15)
            uint4 x0h = xoh, x0l = xol;
            if (h == 1) {
                const float *row = zone_obs + ((size_t)env * Z + t) * F;
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = (valid && j < F) ? row[j] : 0.f;
                v8[7] = 1.0f;
                split8(v8, x0h, x0l);
            }
            // ---- zone_net_.0 + ReLU, split into layer 2's hi / lo B fragments
            uint4 xh[KSB], xl[KSB];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const uint4 ah = w1s[(n * 2 + 0) * 64 + lane], al = w1s[(n * 2 + 1) * 64 + lane];
                f32x16v acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                acc = mfma_bf(ah, x0h, acc);
                acc = mfma_bf(ah, x0l, acc);
                acc = mfma_bf(al, x0h, acc);
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[i], 0.f);
                split8(v, xh[2 * n], xl[2 * n]);
                split8(v + 8, xh[2 * n + 1], xl[2 * n + 1]);
            }
            // ---- zone_net_.2 + ReLU, summed over the tiles.  Output tile n2: 12 k-steps x 3 products on one accumulator;
            // the fragments of k-steps s + 2, s + 3 are read from LDS while those of s, s + 1 are in the matrix pipe
            // (explicit double buffer between scheduling barriers, as in k_mlp_zone_f32m)
            uint4 wa[2][2][2];          // [buffer][k-step of the pair][hi, lo]
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wa[0][q][0] = w2s[((0 * KSB + q) * 2 + 0) * 64 + lane];
                wa[0][q][1] = w2s[((0 * KSB + q) * 2 + 1) * 64 + lane];
            }
            f32x16v acc;
#pragma unroll
            for (int c = 0; c < NT * (KSB / 2); ++c) {
                const int n2 = c / (KSB / 2), sc = c % (KSB / 2);
                if (c + 1 < NT * (KSB / 2)) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        wa[(c + 1) & 1][q][0] = w2s[(((c + 1) * 2 + q) * 2 + 0) * 64 + lane];
                        wa[(c + 1) & 1][q][1] = w2s[(((c + 1) * 2 + q) * 2 + 1) * 64 + lane];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (sc == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int s = 2 * sc + q;
                    acc = mfma_bf(wa[c & 1][q][0], xh[s], acc);
                    acc = mfma_bf(wa[c & 1][q][0], xl[s], acc);
                    acc = mfma_bf(wa[c & 1][q][1], xh[s], acc);
                }
                if (sc == KSB / 2 - 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) P[n2][i] += fmaxf(acc[i], 0.f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the per-env head on the float32 matrix instruction, exactly as in k_mlp_zone_f32m
        const float inv_z = 1.0f / (float)Z;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) P[n][i] *= inv_z;
        f32x16 e3[NT], cc[NT];
        head_layer_f32m<4, false>(w.wcm, lane, P, xo, cc);
        float v_mu = 0.f, v_sigma = 0.f;
        if (w.has_critic) {
            head_layer_f32m<0, true>(w.wv1m, lane, cc, xo, e3);
            f32x16 hv = head_rows_f32m(w.whvm, lane, e3);
            v_mu = hv[0];
            v_sigma = hv[1];
        }
        head_layer_f32m<0, true>(w.wam, lane, cc, xo, e3);
        const f32x16 hd = head_rows_f32m(w.whm, lane, e3);
        if (h == 0 && valid) {
            if (w.has_critic) {
                value[env] = v_mu;
                if (w.distributional && value_sigma) value_sigma[env] = softplus03(v_sigma) + 1e-3f;
            }
            head_outputs(env, hd[0], hd[1], hd[2], hd[3], v_mu, mu, stdv, act);
        }
    }
}
}  // namespace

size_t pack_f32(const zenv_mlp_weights &w, int F, std::vector<float> &out, size_t offs[23])
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
    // ---- ZENV_MLP_BF16X3: the two zone layers as bf16 hi / lo fragments for k_mlp_zone_b3 (8 bf16 = 16 bytes per lane and
    // fragment, stored here as 4 float-sized words).  A operand of v_mfma_f32_32x32x16_bf16: lane (m, hh) element j =
    // W[32 n + m][k], zone_net_.0: k = 8 hh + j (natural); zone_net_.2: the k order in which relu(Y1)'s accumulator
    // registers arrive -- k-step s = 2 t + q takes registers 8 q .. 8 q + 7 of tile t:
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
    auto push_frag = [&](auto ext, int n, auto kof) {       // one fragment pair (hi, lo): 2 x 64 lanes x 8 bf16
        std::vector<uint16_t> hi(64 * 8), lo(64 * 8);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const float v = ext(32 * n + (lane & 31), kof(lane >> 5, j));
                const uint16_t hb = bf16_rne(v);
                hi[lane * 8 + j] = hb;
                lo[lane * 8 + j] = bf16_rne(v - bf16_val(hb));
            }
        const size_t at = out.size();
        out.resize(at + 2 * 64 * 4);
        std::memcpy(&out[at], hi.data(), 64 * 16);
        std::memcpy(&out[at + 64 * 4], lo.data(), 64 * 16);
    };
    offs[21] = out.size();
    for (int n = 0; n < NT; ++n) push_frag(w1ext, n, [](int hh, int j) { return 8 * hh + j; });
    offs[22] = out.size();
    for (int n = 0; n < NT; ++n)
        for (int s2 = 0; s2 < HP / 16; ++s2)
            push_frag(w2ext, n, [s2](int hh, int j) { return 32 * (s2 / 2) + 16 * (s2 % 2) + (j & 3) + 8 * (j >> 2) + 4 * hh; });
    return out.size();
}

hipError_t launch_mlp_forward_f32(const MlpF32 &w, int N, int Z, int F, const float *obs, const float *zone_obs, float *mu,
                                  float *stdv, float *value, float *value_sigma, const MlpAction &act, hipStream_t s)
{
    // A wave of the MFMA kernel works through its envs' zone tiles one group of 32 after the other: ~0.55 ms per group
    // whatever N is, as long as there is at most one wave per SIMD -- one group per wave up to N = 32 768, two above.
    // Small batches (evaluate(): 500 envs) are faster on the vector-ALU kernel, which spreads 4 envs per workgroup over
    // the chip (0.50 ms at N = 8 192, 0.86 ms at 16 384): crossover N ~ 10 000.
    if (w.split3 && (w.on_mfma == 2 || (w.on_mfma == 1 && N >= kMfmaMinEnvs))) {
        // ZENV_MLP_BF16X3: the zone layers as three bf16 products per k-step, the head on the float32 matrix instruction
        const int epw = N <= 32768 ? 32 : 64;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_zone_b3),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZoneLdsB3);
        hipLaunchKernelGGL(k_mlp_zone_b3, dim3((N + kZoneWaves * epw - 1) / (kZoneWaves * epw)), dim3(kZoneWaves * 64),
                           kZoneLdsB3, s, w, N, Z, F, obs, zone_obs, mu, stdv, value, value_sigma, act, epw);
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
