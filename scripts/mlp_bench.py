"""Closed-loop rollout with the on-device actor network (POLICY_MLP_MEAN): time per step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
from oracle import policy_ref as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = Z.default_config(0, 25, zones_keepout=0.40)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n, n_threads=16); env.reset()
env.load_mlp(P.random_tensors(6, seed=0))
env.rollout(300, Z.POLICY_MLP_MEAN)
T = 300
tot, _ = env.rollout(T, Z.POLICY_MLP_MEAN)
h, F, Zn = 185, 6, 25
flop = n * (Zn * 2 * ((8 + F) * h + h * h) + 2 * (h * h + (8 + h) * h + h * h + 4 * h))
print("N %d: %.1f us per step (policy forward + action + env step), %.2f M env-steps/s; network %.1f GFLOP per step"
      % (n, tot / T * 1e3, n * T / tot / 1e3, flop / 1e9))
