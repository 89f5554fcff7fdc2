"""Throughput of zenv_collect (one PPO rollout on the device: actor-critic forward, sampling, record, env step,
GAE).  usage: python scripts/collect_bench.py [T = 64] [N = 65536]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
import combinatorial_rl_tasks_amd._native as nat
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cfg = Z.default_config(0, 25, zones_keepout=0.40)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n, n_threads=16); env.reset()
rs = np.random.RandomState(0)
h, F = 185, 6


def lin(o, i):
    w = rs.standard_normal((o, i)).astype(np.float32)
    return w / np.sqrt((w * w).sum(1, keepdims=True)), (0.1 * rs.standard_normal(o)).astype(np.float32)


t = {}
for (kw, kb), shape in ((("zone_w1", "zone_b1"), (h, 8 + F)), (("zone_w2", "zone_b2"), (h, h)), (("zone_w3", "zone_b3"), (h, h)),
                        (("comb_w", "comb_b"), (h, 8 + h)), (("enc_w", "enc_b"), (h, h)), (("mu_w", "mu_b"), (2, h)),
                        (("std_w", "std_b"), (2, h)), (("critic_w1", "critic_b1"), (h, h)), (("critic_w2", "critic_b2"), (1, h))):
    t[kw], t[kb] = lin(*shape)
env.load_mlp(t, precision="bf16")
L = nat.lib()
for _ in range(2):
    nat.check(L.zenv_collect(env._h, T, 1, 0, 0.99, 0.95)); env.sync()
t0 = time.perf_counter(); K = 5
for _ in range(K):
    nat.check(L.zenv_collect(env._h, T, 1, 0, 0.99, 0.95))
env.sync()
dt = (time.perf_counter() - t0) / K
print("zenv_collect N=%d T=%d: %.2f ms per rollout = %.1f us per frame, %.1f M env-steps/s (buffers %.1f GB)" % (
    n, T, dt * 1e3, dt / T * 1e6, n * T / dt / 1e6, n * T * (8 + 150 + 9) * 4 / 1e9))
