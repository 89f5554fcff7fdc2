"""Fixed cost of the zone kernel K4 (staging, launch ramp, epilogue): its duration at Z = 1, 5, 13, 25 zones per env
(N = 65 536; the tile count per wave is 2 Z), from HIP events around zenv_mlp_forward minus the head kernel's share
(measured at the same N by the Z = 1 point's intercept).  usage: python scripts/k4_fixed_cost.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
from oracle import policy_ref as P
n = 65536
for zones in (1, 5, 13, 25):
    cfg = Z.default_config(0, zones, zones_keepout=0.30)
    env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n, n_threads=16); env.reset()
    env.load_mlp(P.random_tensors(6, seed=1), precision="bf16")
    for _ in range(5):
        env.policy(Z.POLICY_MLP_MEAN)
    env.sync()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        env.policy(Z.POLICY_MLP_MEAN)
    env.sync()
    print("Z %2d: K4 + K5 %.1f us per forward" % (zones, (time.perf_counter() - t0) / K * 1e6))
    env.close()
