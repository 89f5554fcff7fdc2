"""Regression vectors of the ORACLE itself (NOT reference output -- the reference cannot run here).

They freeze what oracle/zenv_oracle.c computes today for the reference-faithful configurations: 300 steps of three
evaluation maps per task under the scripted greedy policy (and 100 under the Philox-uniform one), every step's 8-float
obs, reward, done and goal flag, plus the final zone_obs.  tests/test_oracle_cpu.py replays them, so a later change of the
oracle's arithmetic -- a reformulated substep, another evaluation order -- cannot go unnoticed: it has to regenerate this
file on purpose (python tests/golden/make_oracle_regression.py) and say so in DESIGN.md.  Round 2's physics
reformulation (one half-angle sincos per step, turned (s, k) pair, constant Schur complement) is what is frozen here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O      # noqa: E402

SEEDS = (1000000, 1000001, 1000002)      # main/scripts/evaluate.py:47
CASES = (("tsp", 0, 15), ("timed", 1, 15), ("colour", 2, 6))


def trajectories():
    out = {}
    for tag, task, zones in CASES:
        cfg = O.default_config(task, zones)
        for pol_tag, policy, T in (("greedy", O.POLICY_GREEDY, 300), ("uniform", O.POLICY_UNIFORM, 100)):
            obs = np.zeros((len(SEEDS), T, 8), np.float32)
            rew = np.zeros((len(SEEDS), T), np.float64)
            flags = np.zeros((len(SEEDS), T), np.uint8)          # bit 0 done, bit 1 goal_met
            last_zo = []
            for i, s in enumerate(SEEDS):
                e = O.OracleEnv(cfg)
                e.reset(s)
                o, zo = e.obs()
                for t in range(T):
                    a = e.policy(policy, o, zo, i, t, 0x5EED)
                    r, d, g = e.step(a)
                    if d:
                        e.reset(s)
                    o, zo = e.obs()
                    obs[i, t], rew[i, t], flags[i, t] = o, r, int(d) | (int(g) << 1)
                last_zo.append(zo.copy())
            out[f"{tag}_{pol_tag}_obs"] = obs
            out[f"{tag}_{pol_tag}_reward"] = rew
            out[f"{tag}_{pol_tag}_flags"] = flags
            out[f"{tag}_{pol_tag}_zone_obs"] = np.stack(last_zo)
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz")
    np.savez_compressed(path, seeds=np.array(SEEDS), **trajectories())
    print("wrote", path, os.path.getsize(path), "bytes")
