// kernels.hpp -- host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "../../include/zenv.h"
#include "dev_params.hpp"

namespace zenvk {

// ev_start/ev_stop (may be null) receive the step dispatch's own begin/end timestamps
hipError_t launch_step(const DevParams &p, const float *actions, int auto_reset, hipStream_t s,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
hipError_t launch_reset(const DevParams &p, const uint8_t *mask, hipStream_t s);
hipError_t launch_policy(const DevParams &p, int policy, uint64_t policy_seed, uint64_t env_index0,
                         uint32_t step_index, float *out, hipStream_t s);

}  // namespace zenvk
