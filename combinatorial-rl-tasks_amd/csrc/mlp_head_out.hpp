// mlp_head_out.hpp -- what both network paths (bf16 MFMA: mlp_policy.hip, float32: mlp_f32.hip) do with the head's
// pre-activations: the actor's Normal(mu, std) (policy_network.py:47-50), the action, and -- inside zenv_collect -- frame
// t of the experience buffers (collect_experiences, main/src/torch_ac/algos/base.py:146-160).
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_policy.hpp"

namespace zenvk {

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// a = mu, or Normal(mu, std).sample(): Box-Muller on two Philox uniforms keyed by (seed, global env, step)
__device__ __forceinline__ float2 mlp_action(const MlpAction &act, int env, float2 m, float2 sd)
{
    if (act.mode != 1) return m;
    const uint64_t g = act.env_index0 + (uint64_t)env;
    uint32_t c[4] = { (uint32_t)g, (uint32_t)(g >> 32), act.step_index, 0x4D4C50u };
    philox4x32_10(c, (uint32_t)act.seed, (uint32_t)(act.seed >> 32));
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * 5.9604644775390625e-08f;     // (0, 1)
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * logf(u1));
    return make_float2(m.x + sd.x * rad * cosf(6.283185307179586f * u2), m.y + sd.y * rad * sinf(6.283185307179586f * u2));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// nn.Softplus(beta = 0.3) of the distributional critic's sigma head (flat_model.py:29, :59): log(1 + exp(beta x)) / beta,
// the identity beyond beta x > 20 (torch's threshold)
__device__ __forceinline__ float softplus03(float x)
{
    const float bx = 0.3f * x;
    return bx > 20.f ? x : log1pf(expf(bx)) / 0.3f;
}

// mu_a / mu_b: mu_(x) of the two action dimensions, sd_a / sd_b: std_(x); value: the critic's (mean) value
__device__ __forceinline__ void head_outputs(int env, float mu_a, float mu_b, float sd_a, float sd_b, float value_out,
                                             float *__restrict__ mu, float *__restrict__ stdv, const MlpAction &act)
{
    const float2 m = make_float2(2.0f * (sigmoidf_(mu_a) - 0.5f), 2.0f * (sigmoidf_(mu_b) - 0.5f));
    const float2 sd = make_float2(sigmoidf_(sd_a) + 1e-3f, sigmoidf_(sd_b) + 1e-3f);
    reinterpret_cast<float2 *>(mu)[env] = m;
    reinterpret_cast<float2 *>(stdv)[env] = sd;
    if (act.mode < 0) return;
    const float2 a = mlp_action(act, env, m, sd);
    reinterpret_cast<float2 *>(act.actions)[env] = a;
    const MlpRecord &rc = act.rec;
    if (!rc.action) return;
    const size_t slot = (size_t)rc.t * rc.N + env;      // time-major [T][N]
    reinterpret_cast<float2 *>(rc.action)[slot] = a;
    // Normal(mu, std).log_prob(a), per action dimension (base.py:160)
    const float z0 = (a.x - m.x) / sd.x, z1 = (a.y - m.y) / sd.y;
    reinterpret_cast<float2 *>(rc.log_prob)[slot] =
        make_float2(-0.5f * z0 * z0 - logf(sd.x) - 0.91893853320467274178f,
                    -0.5f * z1 * z1 - logf(sd.y) - 0.91893853320467274178f);
    rc.value[slot] = value_out;
    if (rc.t == 0) {
        rc.mask[slot] = rc.cur_mask[env];            // self.masks[i] = self.mask (:149), BEFORE this step
    } else {
        rc.mask[slot] = rc.prev_done[env] ? 0.f : 1.f;                 // self.mask = 1 - done (:150)
        rc.reward[slot - rc.N] = rc.prev_shaped ? (float)rc.prev_shaped[env] : rc.prev_reward[env];
    }
}

}  // namespace zenvk
