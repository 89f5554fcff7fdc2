"""Diagnostic: persistent-kernel time per step and per 65 536 envs over the batch size (one handle, env i replays map
1 + (i mod 65536)).  usage: python scripts/nsweep.py [workload] [sizes...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import combinatorial_rl_tasks_amd as Z
w = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "PointTSP-25"
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [65536, 131072, 196608, 262144, 393216, 524288, 1048576]
task, zones, keep = bench.WORKLOADS[w]
for n in sizes:
    cfg = Z.default_config(task, zones, zones_keepout=keep)
    env = Z.ZoneVecEnv(cfg, n)
    bench.replay_bank(env, n, 65536)
    env.reset()
    env.rollout(max(256, int(6000 * 65536 / n)), Z.POLICY_GREEDY)
    per_launch = []
    for _ in range(6):
        ms, k = env.rollout(256, Z.POLICY_GREEDY, time_step_kernel=True)
        per_launch.append(k * 1e3)
    us = float(np.median(per_launch))
    alg = bench.algorithmic_bytes(task, zones, 256)
    print("%-14s N %8d: %8.2f us/step  %6.3f us per 65536 envs  %.2f TB/s algorithmic  (launches: %s)" %
          (w, n, us, us * 65536 / n, alg * n / us / 1e6, " ".join("%.2f" % x for x in per_launch)), flush=True)
    env.close()
