"""Diagnostic: what does the action-buffer source of k_rollout_lane<EXT> cost?  Same envs, statistically the same actions
(uniform on [-1, 1]^2): scripted pi_uniform inside the persistent kernel (Philox per step) vs zenv_step_many replaying a
host-drawn uniform buffer from HBM (larger than the last-level cache).  us per step, N = 65 536, per workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import combinatorial_rl_tasks_amd as Z
from combinatorial_rl_tasks_amd import _native as nat
WL = {"PointTSP-25": (0, 25, .4), "TimedTSP-25": (1, 25, .4), "ColourMatch-6": (2, 6, .55), "PointTSP-15": (0, 15, .55)}
n, K = 65536, 1024      # 512 MiB of actions: twice the Infinity Cache, so every replay streams the rows from HBM
rs = np.random.RandomState(0)
a = rs.uniform(-1, 1, (K, n, 2)).astype(np.float32)
for w, (task, zones, keep) in WL.items():
    cfg = Z.default_config(task, zones, zones_keepout=keep)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, 4 * n, n_threads=16)
    env.schedule_sequential(stride=n)
    env.reset()
    env.rollout(6000, Z.POLICY_UNIFORM)
    ep0 = int(env.get(Z.F_EPISODES).sum())
    ms, _ = env.rollout(6 * K, Z.POLICY_UNIFORM)
    ep1 = int(env.get(Z.F_EPISODES).sum())
    env.step_many(a, reset="every")
    ptr = (env.device_ptr(nat.F_CHUNK_ACTIONS), K)
    for _ in range(3):
        env.step_many(None, reset="every", actions_ptr=ptr)
    env.sync()
    ep2 = int(env.get(Z.F_EPISODES).sum())
    t0 = time.perf_counter()
    for _ in range(6):
        env.step_many(None, reset="every", actions_ptr=ptr)
    env.sync()
    ext = (time.perf_counter() - t0) / (6 * K) * 1e6
    ep3 = int(env.get(Z.F_EPISODES).sum())
    ms2, _ = env.rollout(6 * K, Z.POLICY_GREEDY)
    ms2, _ = env.rollout(6 * K, Z.POLICY_GREEDY)
    print(f"{w:14s} scripted uniform {ms / (6 * K) * 1e3:6.3f} us/step | action buffer {ext:6.3f} us/step | scripted greedy {ms2 / (6 * K) * 1e3:6.3f} | episodes ended: uniform {ep1 - ep0}, buffer {ep3 - ep2}", flush=True)
    env.close()
