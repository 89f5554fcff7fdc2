"""Diagnostic: where does the action-chunk replay of bench.py's aux.action_chunk spend its time?  Replays of the recorded
greedy actions, timed several times over, at chunk lengths 2048 and 256, with and without the snapshot restore."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import combinatorial_rl_tasks_amd as Z
from combinatorial_rl_tasks_amd import _native as nat
n, K = 65536, 2048
cfg = Z.default_config(0, 25, zones_keepout=0.4)
env = Z.ZoneVecEnv(cfg, n)
env.build_bank(1, 4 * n, n_threads=16)
env.schedule_sequential(stride=n)
env.reset()
env.step_many(np.zeros((K, n, 2), np.float32), reset="every")
ptr = env.device_ptr(nat.F_CHUNK_ACTIONS)
env.rollout(6000, Z.POLICY_GREEDY)
env.policy(Z.POLICY_GREEDY)
s0 = env.get_state()
for t in range(K):
    env.get_into_device(nat.F_ACTIONS, ptr + 8 * n * t)
    env.rollout(1, Z.POLICY_GREEDY, mode="per_step")
def timed(k, reps, restore):
    out = []
    for r in range(reps):
        if restore:
            env.set_state(s0)
            env.rollout(0, Z.POLICY_GREEDY)
        env.sync()
        t0 = time.perf_counter()
        env.step_many(None, reset="every", actions_ptr=(ptr, k))
        env.sync()
        out.append((time.perf_counter() - t0) / k * 1e6)
    return " ".join("%.3f" % x for x in out)
print("K=2048 with restore:", timed(2048, 4, True))
print("K=2048 back to back:", timed(2048, 4, False))
print("K=256  back to back:", timed(256, 8, False))
env.set_state(s0)
ms, _ = env.rollout(2048, Z.POLICY_GREEDY); print("scripted greedy 2048 from S0: %.3f" % (ms / 2048 * 1e3))
ms, _ = env.rollout(2048, Z.POLICY_GREEDY); print("scripted greedy 2048 again  : %.3f" % (ms / 2048 * 1e3))
