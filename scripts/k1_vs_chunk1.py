"""Diagnostic: one externally driven step as k_step_lane (zenv_step) vs as a ONE-step launch of the persistent kernel's
action-buffer form (zenv_step_many, K = 1 .. 4), us per step, back-to-back launches, device-resident actions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import combinatorial_rl_tasks_amd as Z
from combinatorial_rl_tasks_amd import _native as nat
WL = {"PointTSP-25": (0, 25, .4), "TimedTSP-25": (1, 25, .4), "ColourMatch-6": (2, 6, .55), "PointTSP-15": (0, 15, .55)}
n = 65536
rs = np.random.RandomState(0)
for w, (task, zones, keep) in WL.items():
    cfg = Z.default_config(task, zones, zones_keepout=keep)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, 4 * n, n_threads=16)
    env.schedule_sequential(stride=n)
    env.reset()
    env.rollout(6000, Z.POLICY_GREEDY)
    a = rs.uniform(-1, 1, (4, n, 2)).astype(np.float32)
    env.step_many(a, reset="every")
    ptr = env.device_ptr(nat.F_CHUNK_ACTIONS)
    out = []
    for K in (1, 2, 4):
        for _ in range(200):
            env.step_many(None, reset="every", actions_ptr=(ptr, K))
        env.sync(); t0 = time.perf_counter()
        for _ in range(2000):
            env.step_many(None, reset="every", actions_ptr=(ptr, K))
        env.sync(); out.append((time.perf_counter() - t0) / (2000 * K) * 1e6)
    env.policy(Z.POLICY_UNIFORM)
    for _ in range(200):
        env.step(None, auto_reset=True)
    env.sync(); t0 = time.perf_counter()
    for _ in range(3000):
        env.step(None, auto_reset=True)
    env.sync(); k1 = (time.perf_counter() - t0) / 3000 * 1e6
    print(f"{w:14s} zenv_step (k_step_lane) {k1:6.2f} us/step | step_many K=1 {out[0]:6.2f}  K=2 {out[1]:6.2f}  K=4 {out[2]:6.2f} us/step", flush=True)
    env.close()
