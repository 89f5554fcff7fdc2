"""Diagnostic: steady-state time per step of the persistent kernel, per workload and scripted policy (N = 65 536).
Uses only the rollout API, so it runs unchanged in an older checkout (scripts/ab_trees.py-style A/B):
    python <tree>/scripts/k1p_rate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combinatorial_rl_tasks_amd as Z
WL = {"PointTSP-25": (0, 25, .4), "TimedTSP-25": (1, 25, .4), "ColourMatch-6": (2, 6, .55), "PointTSP-15": (0, 15, .55)}
n = 65536
for w, (task, zones, keep) in WL.items():
    cfg = Z.default_config(task, zones, zones_keepout=keep)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, 4 * n, n_threads=16)
    env.schedule_sequential(stride=n)
    env.reset()
    env.rollout(6000, Z.POLICY_GREEDY)
    out = []
    for pol in (Z.POLICY_GREEDY, Z.POLICY_UNIFORM, Z.POLICY_GREEDY):
        env.rollout(512, pol)
        ms, k = env.rollout(8192, pol, time_step_kernel=True)
        out.append("%6.3f (kernel %6.3f)" % (ms / 8192 * 1e3, k * 1e3))
    print(f"{os.path.basename(ROOT) or ROOT:8s} {w:14s} greedy {out[0]} | uniform {out[1]} | greedy {out[2]}", flush=True)
    env.close()
