"""MI355X-native batched PointTSP / TimedTSP / ColourMatch env.step() path.

Layout of this package (only what the hot path needs):
  csrc/        HIP kernels (gfx950) + the C ABI of include/zenv.h + the host layout sampler
  _native.py   ctypes binding (no PyTorch, no CPU fallback)
  vec_env.py   ZoneVecEnv: N device-resident envs, struct-of-arrays results
  envs/        host-side mirror of the reference interface (main/envs/*): registry ids,
               TSPEnv/TimedTSPEnv/ColourMatchEnv, FixedSeedsWrapper/ZoneWrapper, make_*_env
  penv.py      ParallelEnv-shaped vector env over one batched handle
  sharding.py  one-process-per-GPU env sharding + gather of episodic returns
"""
from ._native import (Config, ZenvError, E_ARG, E_HIP, E_STATE, E_LAYOUT, E_DONE, E_RANGE, TASK_TSP, TASK_TIMED_TSP, TASK_COLOUR_MATCH,
                      POLICY_UNIFORM, POLICY_GREEDY, POLICY_MLP_MEAN, POLICY_MLP_SAMPLE, F_OBS, F_ZONE_OBS, F_REWARD, F_DONE,
                      F_GOAL_MET, F_EP_RETURN, F_EP_LEN, F_LAST_RETURN, F_LAST_LEN, F_EPISODES,
                      F_VISIT_COUNT, F_SEED, F_ACTIONS, F_POLICY_MU, F_POLICY_STD, F_POLICY_VALUE,
                      F_SHAPED_REWARD, F_NEED_GOAL, F_AVAILABLE_GOALS, F_GOAL, F_ORDER_VAL, F_EXCEPTION, F_POLICY_VALUE_SIGMA)
from .vec_env import (ZoneVecEnv, config_for_id, default_config, sample_layout,
                      fixed_seed_sequence, route_ranks, zone_feat)

__version__ = "0.1.0"
