import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import combinatorial_rl_tasks_amd as Z
task, zones, keep = bench.WORKLOADS["PointTSP-25"]
cfg = Z.default_config(task, zones, zones_keepout=keep)
def mk(n):
    env = Z.ZoneVecEnv(cfg, n); bench.replay_bank(env, n, 65536); env.reset(); env.rollout(3000, Z.POLICY_GREEDY); return env
# A: one handle of 131072
a = mk(131072)
for _ in range(3):
    ms, k = a.rollout(1024, Z.POLICY_GREEDY, time_step_kernel=True)
    print("one handle 131072: %.2f us/step (%.2f per 65536)" % (k * 1e3, k * 1e3 / 2), flush=True)
a.close()
# B: two handles of 65536, launches alternating on their own streams (enqueue both, then wait)
b1, b2 = mk(65536), mk(65536)
for _ in range(3):
    t0 = time.perf_counter()
    b1.rollout(1024, Z.POLICY_GREEDY, wait=False); b2.rollout(1024, Z.POLICY_GREEDY, wait=False)
    b1.sync(); b2.sync()
    dt = (time.perf_counter() - t0) / 1024 * 1e6
    print("two handles 65536 concurrently (two streams): %.2f us per step of both (%.2f per 65536)" % (dt, dt / 2), flush=True)
for _ in range(3):
    t0 = time.perf_counter()
    b1.rollout(1024, Z.POLICY_GREEDY); b2.rollout(1024, Z.POLICY_GREEDY)
    dt = (time.perf_counter() - t0) / 1024 * 1e6
    print("two handles 65536 one after the other: %.2f us per step of both (%.2f per 65536)" % (dt, dt / 2), flush=True)
b1.close(); b2.close()
