"""Diagnostic: time per step of successive 256-step persistent launches over a 2000-step episode."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
n = 65536
task, zones, keep = {"tsp": (0, 25, .4), "timed": (1, 25, .4), "colour": (2, 6, .55), "tsp15": (0, 15, .55)}[sys.argv[1] if len(sys.argv) > 1 else "tsp"]
cfg = Z.default_config(task, zones, zones_keepout=keep)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
if len(sys.argv) > 3:      # pre-warm the GPU, then start over from a fresh reset
    env.rollout(int(sys.argv[3]), Z.POLICY_GREEDY)
    env.schedule_sequential(stride=n)
    env.reset()
env.rollout(50, Z.POLICY_GREEDY)
prev = 0
for c in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    tot, k = env.rollout(256, Z.POLICY_GREEDY, time_step_kernel=True)
    ep = int(env.get(Z.F_EPISODES).sum())
    print("steps %4d..%4d: %.2f us/step (kernel %.2f), resets in window %6d, mean visited %.1f" % (
        50 + 256 * c, 50 + 256 * (c + 1), tot / 256 * 1e3, k * 1e3, ep - prev, env.get(Z.F_VISIT_COUNT).mean()))
    prev = ep
