// kernels.hpp -- host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "../../include/zenv.h"
#include "dev_params.hpp"

namespace zenvk {

// Scripted action source: policy < 0 = none.  As an argument of launch_step it asks the step
// kernel to also write the action of the NEXT step (index step_index) into `out` (fused K3).
struct StepPolicy {
    int32_t policy;
    uint32_t step_index;
    uint64_t seed;
    uint64_t env_index0;
    float *out;
};
inline StepPolicy no_policy() { return StepPolicy{ -1, 0u, 0ull, 0ull, nullptr }; }

// ev_start/ev_stop (may be null) receive the step dispatch's own begin/end timestamps
hipError_t launch_step(const DevParams &p, const float *actions, int auto_reset, const StepPolicy &pol,
                       hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// K closed-loop steps in one launch (persistent kernel); a_0 must be in p.actions, a_K is left there.
// pol.step_index = index of the FIRST action the launch computes (= step count before + 1).
bool rollout_kernel_available(const DevParams &p);
// tile0 / n_tiles: the slice of the batch's 64-env tiles this launch covers (n_tiles < 0: all from tile0 on)
int rollout_tiles(const DevParams &p);
hipError_t launch_rollout(const DevParams &p, int n_steps, int auto_reset, const StepPolicy &pol, hipStream_t s,
                          hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int tile0 = 0, int n_tiles = -1);
// The action-chunk form of the persistent kernel (zenv_step_many): n_steps steps of caller-supplied actions
// io.actions [n_steps][N] (device), every step's reward / done recorded time-major in io.reward / io.done.
// reset_mode: 0 never, 1 every step, 2 only in the launch's last step.
struct ChunkIO {
    const float2 *actions;
    float *reward;
    uint8_t *done;
};
hipError_t launch_rollout_actions(const DevParams &p, int n_steps, int reset_mode, const ChunkIO &io, hipStream_t s,
                                  hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int tile0 = 0, int n_tiles = -1);
// next_slot[env] = seq_slot(slot_first[env], episode_idx[env]) after the host changed the schedule or the bank's size
// restart != 0: the schedule also starts over at episode 0
hipError_t launch_sched_sync(const DevParams &p, int restart, hipStream_t s);
// HotA -> plain arrays (ZENV_F_EP_RETURN float64 [N], ZENV_F_EP_LEN int32 [N])
hipError_t launch_unpack_hot(const DevParams &p, double *ep_return, int32_t *steps, hipStream_t s);
hipError_t launch_reset(const DevParams &p, const uint8_t *mask, hipStream_t s);
// goal-conditioned variant: set goals (new_goal[N], -1 = keep; *bad counts rejected ones) / per-step shaping
hipError_t launch_goal_set(const DevParams &p, const int32_t *new_goal, int32_t *bad, hipStream_t s);
hipError_t launch_goal_step(const DevParams &p, hipStream_t s);
hipError_t launch_goal_clear(const DevParams &p, const uint8_t *mask, hipStream_t s);
// ColourMatchSolverEnv.solver_get_next_goal for every env -> out[N] (device)
hipError_t launch_solver_goal(const DevParams &p, int32_t *out, hipStream_t s);
// solver-ordered variant (TSP_order_env.py): routes = the bank's aux column
hipError_t launch_order_reset(const DevParams &p, const uint8_t *mask, hipStream_t s);
hipError_t launch_order_step(const DevParams &p, hipStream_t s);
hipError_t launch_policy(const DevParams &p, const StepPolicy &pol, hipStream_t s);
// first obs + first greedy action of every bank slot (slots == null: all `count` = bank_size slots) -> p.bank_first
hipError_t launch_bank_derive(const DevParams &p, const int32_t *slots, int count, hipStream_t s);
// zenv_bank_update: scatter `count` packed layout records (rec_bytes each) into the bank slots slots[count]
hipError_t launch_bank_scatter(const DevParams &p, const int32_t *slots, const void *staging, int rec_bytes, int count,
                               hipStream_t s);
// send buffer of zenv_allgather: n elements of 4 (copied) or 8 bytes (float64 -> float32) into 4-byte slots
hipError_t launch_gather_prep(const void *src, int elem_bytes, void *dst, int n, hipStream_t s);
// measurement utility: n_tiles waves each rewrite their contiguous tile_bytes `steps` times (aux: 0 plain, 2 nt, 16 sc1)
hipError_t launch_probe_store(float *out, long long n_tiles, int tile_bytes, int steps, int aux, hipStream_t s,
                              hipEvent_t ev_start, hipEvent_t ev_stop);

}  // namespace zenvk
