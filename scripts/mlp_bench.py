"""Closed-loop rollout with the on-device actor network (POLICY_MLP_MEAN): time per step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"      # "bf16" | "f16" | "f32" | "f16x3" | "bf16x3"
short = len(sys.argv) > 3                                      # a few steps only (under the profiler's counters)
cfg = Z.default_config(0, 25, zones_keepout=0.40)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n, n_threads=16); env.reset()
rs = np.random.RandomState(0)
h_, F_ = 185, 6


def lin(n_out, n_in):   # rows of N(0,1) normalised to unit norm (flat_model.py:13-19), small biases
    w = rs.standard_normal((n_out, n_in)).astype(np.float32)
    return w / np.sqrt((w * w).sum(1, keepdims=True)), (0.1 * rs.standard_normal(n_out)).astype(np.float32)


t = {}
for (kw, kb), shape in ((("zone_w1", "zone_b1"), (h_, 8 + F_)), (("zone_w2", "zone_b2"), (h_, h_)),
                        (("zone_w3", "zone_b3"), (h_, h_)), (("comb_w", "comb_b"), (h_, 8 + h_)),
                        (("enc_w", "enc_b"), (h_, h_)), (("mu_w", "mu_b"), (2, h_)), (("std_w", "std_b"), (2, h_))):
    t[kw], t[kb] = lin(*shape)
env.load_mlp(t, precision=precision)
T = 6 if short else 300 if precision in ("bf16", "f16") else 40
env.rollout(T, Z.POLICY_MLP_MEAN)
tot, _ = env.rollout(T, Z.POLICY_MLP_MEAN)
h, F, Zn = 185, 6, 25
flop = n * (Zn * 2 * ((8 + F) * h + h * h) + 2 * (h * h + (8 + h) * h + h * h + 4 * h))
print(precision, "N %d: %.1f us per step (policy forward + action + env step), %.2f M env-steps/s; network %.1f GFLOP per step"
      % (n, tot / T * 1e3, n * T / tot / 1e3, flop / 1e9))
