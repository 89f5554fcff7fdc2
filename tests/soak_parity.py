"""Long parity soak (run by hand on the GPU box: python tests/soak_parity.py [N] [T]; not collected by pytest): persistent rollouts of many envs over many steps and episodes against the
oracle's batch driver, all three tasks, both scripted policies, wrapping map bank."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import combinatorial_rl_tasks_amd as Z
from oracle import oracle as O
from tests.helpers import oracle_config_from
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
mode = sys.argv[3] if len(sys.argv) > 3 else "persistent"     # persistent | per_step | unfused | chunk
slice_envs = int(sys.argv[4]) if len(sys.argv) > 4 else None   # zenv_set_rollout_slice (0: one launch over the batch)
depth = 6
bad = 0
for task, zones, keep in ((0, 25, 0.40), (1, 25, 0.40), (2, 6, 0.55), (0, 15, 0.55), (1, 15, 0.55), (0, 5, 0.55)):
    for pol_d, pol_o in ((Z.POLICY_GREEDY, O.POLICY_GREEDY), (Z.POLICY_UNIFORM, O.POLICY_UNIFORM)):
        cfg = Z.default_config(task, zones, zones_keepout=keep, num_steps=700)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(1, depth * n, n_threads=16)
        env.schedule_sequential(stride=n)
        env.reset()
        if slice_envs is not None:
            env.set_rollout_slice(slice_envs)
        t0 = time.time()
        if mode == "chunk":
            # the scripted policy's own actions, recorded step by step from the reset state, then the state restored and
            # the whole [T][n][2] buffer replayed through zenv_step_many (launches of <= 256 steps, auto-reset every step)
            from combinatorial_rl_tasks_amd import _native as nat
            s0 = env.get_state()
            env.step_many(np.zeros((T, n, 2), np.float32), reset="every")      # sizes the chunk buffers
            ptr = env.device_ptr(nat.F_CHUNK_ACTIONS)
            env.set_state(s0)
            env.policy(pol_d, policy_seed=77, env_index0=5)
            for t in range(T):
                env.get_into_device(nat.F_ACTIONS, ptr + 8 * n * t)
                env.rollout(1, pol_d, policy_seed=77, env_index0=5, mode="per_step")
            env.set_state(s0)
            env.step_many(None, reset="every", actions_ptr=(ptr, T))
        else:
            env.rollout(T, pol_d, policy_seed=77, env_index0=5, mode=mode)
        ref = O.rollout(oracle_config_from(O, cfg), 1 + np.arange(n), T, pol_o, seed_stride=n, policy_seed=77,
                        env_index0=5, n_threads=16, seed_period=depth)
        ok = (np.array_equal(env.get(Z.F_OBS), ref["obs"]) and np.array_equal(env.get(Z.F_ZONE_OBS), ref["zone_obs"])
              and np.array_equal(env.get(Z.F_EPISODES), ref["episodes"])
              and np.array_equal(env.get(Z.F_LAST_RETURN), ref["last_return"])
              and np.array_equal(env.get(Z.F_LAST_LEN), ref["last_len"]))
        bad += not ok
        print("task %d Z %2d policy %d: %s  (%d env-steps, %d episodes, %.1fs)" % (
            task, zones, pol_d, "bit-identical" if ok else "MISMATCH", n * T, int(ref["episodes"].sum()), time.time() - t0),
            flush=True)
        env.close()
print("soak (%s%s):" % (mode, "" if slice_envs is None else ", slice %d" % slice_envs), "all bit-identical" if not bad else f"{bad} MISMATCHES")
