// mlp_policy.hpp -- the reference's actor network on the device.
//
// ZoneEnvModel (main/src/env_model.py:48-79) + the actor of ACModel (flat_model.py:24-37,
// policy_network.py:12-53) as two bf16-MFMA kernels that read the env's own obs / zone_obs
// buffers: no host round trip between env.step and the next action.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

#include "../../include/zenv.h"

namespace zenvk {

constexpr int kMlpHP = 192;              // hidden width padded to whole 32-wide MFMA tiles
constexpr int kMlpNT = kMlpHP / 32;      // output tiles per hidden layer
constexpr int kMlpKS = kMlpHP / 16;      // k-steps (K = 16 each) per hidden layer

// Device images of the weights, already in MFMA fragment order (one 16-byte piece per lane and
// fragment; see pack_images()).  Counts are in fragments of 64 lanes x 8 bf16 = 1 KiB.
struct MlpImages {
    const void *w1;     // [NT][1]      zone_net_.0   A operand, natural k order (obs, zone row, bias slot)
    const void *w2;     // [NT][KS]     zone_net_.2   B operand of H2^T = X1^T W2^T (k in accumulator order)
    const void *w3;     // [NT][KS]     zone_net_.4   A operand, natural k order (input read from memory)
    const void *wc;     // [NT][KS+1]   combine_net_  A operand: KS steps accumulator order + 1 step obs
    const void *wa;     // [NT][KS]     actor.enc_.0.0  A operand, accumulator order
    const void *wh;     // [1][KS]      actor.mu_ / actor.std_ (rows 0-1 / 2-3), accumulator order
    const void *wv1;    // [NT][KS]     critic.0 (null: no value head), accumulator order
    const void *wv2;    // [1][KS]      critic.2 or critic_mu (row 0) and critic_sigma (row 1), accumulator order
    const struct MlpF32 *f32;   // non-null: the float32 path (mlp_f32.hip) runs instead of the bf16 MFMA kernels
    int distributional;         // critic_mu / critic_sigma heads (flat_model.py:35-41) instead of critic.2
    int elem_f16;               // the images hold float16, not bf16: the float16 build of the same kernels runs (ZENV_MLP_F16)
    int *range_flag;            // float16 only: pinned host word set when an observation or an activation left the range
};
// ZENV_MLP_F16: zenv_mlp_load bounds the zone layers' activations for observations of at most this magnitude (the robot
// 190 m from the arena's centre); k_mlp_zone1 reports an env beyond it
constexpr float kMlpF16ObsBound = 64.0f;

// The float32 path's weights: every matrix TRANSPOSED ([in][kMlpHP], so that consecutive threads = consecutive output
// features read consecutive floats) and zero-padded to kMlpHP columns; biases [kMlpHP].  Device pointers.
struct MlpF32 {
    int h, distributional, has_critic, pad;
    const float *w1t, *b1;      // zone_net_.0   [16][HP]: k = obs 0..7, zone row 0..F-1
    const float *w2t, *b2;      // zone_net_.2   [HP][HP]
    const float *w3t, *b3;      // zone_net_.4
    const float *wct, *bc;      // combine_net_  [8 + HP][HP]: k = obs 0..7, then zone_emb
    const float *wat, *ba;      // actor.enc_.0.0
    const float *wv1t, *bv1;    // critic.0
    const float *heads;         // [8][HP + 1] rows: mu_ 0-1, std_ 0-1, critic.2 / critic_mu, critic_sigma, 0, 0; bias last
    // MFMA fragment images (k_mlp_zone_f32m; [tiles][k-steps][64 lanes]): zone_net_.0 [6][8], zone_net_.2 [6][96],
    // zone_net_.4 [6][96], combine_net_ [6][100], actor.enc_ [6][96], actor heads [1][96], critic.0 [6][96], critic
    // heads [1][96]
    const float *w1m, *w2m, *w3m, *wcm, *wam, *whm, *wv1m, *whvm;
    int on_mfma, split3;        // on_mfma 0: always the vector-ALU kernel k_mlp_f32 (ZENV_MLP_F32_VALU=1); 2: always the MFMA
                                // kernel (ZENV_MLP_F32_MFMA=1); 1: by batch size.  split3: the MFMA kernel is k_mlp_zone_s3 on
                                // hi / lo split operands, three products per k-step -- 1: bf16 halves, zone layers only
                                // (ZENV_MLP_BF16X3); 2: f16 halves, every layer (ZENV_MLP_F16X3)
    // k_mlp_zone_s3's images, fragment pairs of 64 x 16 B hi + 64 x 16 B lo: zone_net_.0 [6], zone_net_.2 [6][12] in bf16
    // (w1b, w2b) and f16 (w1h, w2h); f16 only: combine_net_ [6][1 + 12], critic.0 [6][12], actor.enc_ [6][12], actor
    // heads [12], critic heads [12]
    const void *w1b, *w2b, *w1h, *w2h, *wch, *wv1h, *wah, *whh, *whvh;
    int *range_flag;            // pinned host word the f16 kernel sets when an operand left float16's range
};
// floats needed for the device image and the packer (offsets of the arrays above, in floats, in `offs[30]`)
size_t pack_f32(const zenv_mlp_weights &w, int F, std::vector<float> &out, size_t offs[30]);
hipError_t launch_mlp_forward_f32(const MlpF32 &w, int N, int Z, int F, const float *obs, const float *zone_obs, float *mu,
                                  float *stdv, float *value, float *value_sigma, const struct MlpAction &act,
                                  hipStream_t s);

// Packs the float32 state_dict tensors into one host buffer of fragments; offsets (in bytes) of the
// images are returned in `offs` (the last two only when the critic tensors are given).  h = hidden width (<= 191), F = zone features (6 or 7).
int pack_images(const zenv_mlp_weights &w, int F, std::vector<uint16_t> &out, size_t offs[8], bool f16 = false);

// What the head kernel does with (mu, std) besides storing them: nothing (mode < 0), actions = mu (0), or
// actions = mu + std * eps with eps ~ N(0,1) from Philox4x32-10 keyed by (seed, global env, step) (1) -- the
// reference's dist.sample() (utils/agent.py:41-44).
// rec (zenv_collect only): the head kernel also writes frame t of the experience buffers -- action, per-dimension
// log_prob, value, mask -- and the reward of frame t-1, which the step kernel left behind in the env's own
// reward / done buffers (collect_experiences, base.py:146-160; time-major [T][N] arrays).
struct MlpRecord {
    float *action, *log_prob, *value, *mask, *reward;   // null action: nothing is recorded
    const float *cur_mask;                               // mask of frame 0 (self.mask of the previous call)
    const float *prev_reward;                            // frame t-1 ...
    const double *prev_shaped;                           // ... or its shaped_reward (goal-conditioned envs), else null
    const uint8_t *prev_done;
    int T, t, N;
};
struct MlpAction {
    int mode;
    uint32_t step_index;
    uint64_t seed, env_index0;
    float *actions;
    MlpRecord rec;
};
inline MlpAction no_mlp_action() { return MlpAction{ -1, 0u, 0ull, 0ull, nullptr, MlpRecord{} }; }

// obs [N,8], zone_obs [N,Z,F] float32 (device) -> mu, std [N,2] float32 (device).
// pooled: scratch [N][kMlpHP] bf16 (device).
hipError_t launch_mlp_forward(const MlpImages &img, int N, int Z, int F, const float *obs, const float *zone_obs,
                              void *pooled, float *mu, float *stdv, float *value, float *value_sigma,
                              const MlpAction &act, hipStream_t s);

// the float16 build of the same two kernels (mlp_policy_f16.hip); called by launch_mlp_forward when img.elem_f16
hipError_t launch_mlp_forward_f16(const MlpImages &img, int N, int Z, int F, const float *obs, const float *zone_obs,
                                  void *pooled, float *mu, float *stdv, float *value, float *value_sigma,
                                  const MlpAction &act, hipStream_t s);

// Experience buffers of one collect_experiences() call (base.py:131-216), all time-major [T][N][...] (the
// observations are written in place by the step kernel)
struct ExpBuffers {
    int T;
    float *obs;        // [T][N][8]
    float *zone_obs;   // [T][N][Z*F]
    float *action;     // [T][N][2]
    float *log_prob;   // [T][N][2]
    float *value, *reward, *mask, *advantage, *returnn;   // [T][N]
    float *cur_mask;   // [N]  self.mask, carried from one call to the next
};
hipError_t launch_exp_reward(const ExpBuffers &x, int N, int t, const float *reward, const double *shaped,
                             const uint8_t *done, hipStream_t s);
hipError_t launch_exp_gae(const ExpBuffers &x, int N, const float *next_value, float discount, float gae_lambda,
                          hipStream_t s);

}  // namespace zenvk
