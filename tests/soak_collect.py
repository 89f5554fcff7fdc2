"""Longer run of the zenv_collect checks (by hand on the GPU box: python tests/soak_collect.py [N] [T]; not collected
by pytest): the recorded actions replayed through the oracle must reproduce every recorded observation, reward and
mask bit for bit; log_prob and GAE are recomputed from the recorded values."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first HIP runtime loaded)
import combinatorial_rl_tasks_amd as Z
from oracle import oracle as O
from oracle import policy_ref as P
from tests.helpers import OracleBatch, oracle_config_from

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
O.build()
bad = 0
for env_id, num_steps in (("PointTSP-v0", 90), ("PointTTSP-v0", 90), ("ColourMatch-v0", 60), ("PointTSP-v1", 50)):
    t0 = time.time()
    cfg = Z.config_for_id(env_id, num_steps=num_steps)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(31, n)
    env.schedule_sequential()                 # a reset replays the env's own map, like OracleBatch
    env.reset()
    w = P.random_tensors(env.zone_feat, seed=4, critic=True)
    env.load_mlp(w, precision="bf16")
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(31, 31 + n))
    ob.reset()
    prev_mask = np.ones(n, np.float32)
    ok = True
    episodes = 0
    for call in range(2):
        x = env.collect(T, policy_seed=8 + call, discount=0.99, gae_lambda=0.95)
        for k in range(T):
            o_ref, zo_ref = ob.obs()
            ok &= np.array_equal(x["obs"][:, k], o_ref) and np.array_equal(x["zone_obs"][:, k], zo_ref)
            ok &= np.array_equal(x["mask"][:, k], prev_mask)
            r, d, _ = ob.step(x["action"][:, k])
            ok &= np.array_equal(x["reward"][:, k], r.astype(np.float32))
            prev_mask = np.where(d, 0.0, 1.0).astype(np.float32)
            episodes += int(d.sum())
        mu, std, val = P.forward_bf16_emulated(w, x["obs"].reshape(-1, 8), x["zone_obs"].reshape(n * T, cfg.num_zones, -1))
        ok &= np.abs(val.reshape(n, T) - x["value"]).max() < 4e-3
        mu, std = mu.reshape(n, T, 2), std.reshape(n, T, 2)
        lp = -0.5 * ((x["action"] - mu) / std) ** 2 - np.log(std) - 0.5 * np.log(2 * np.pi)
        ok &= np.abs(lp - x["log_prob"]).max() < 0.15
        _, _, next_value = env.mlp_forward(with_value=True)
        nv, nm, na = next_value.astype(np.float32), prev_mask.copy(), np.zeros(n, np.float32)
        adv = np.zeros((n, T), np.float32)
        for k in reversed(range(T)):
            delta = x["reward"][:, k] + np.float32(0.99) * nv * nm - x["value"][:, k]
            adv[:, k] = delta + np.float32(0.99) * np.float32(0.95) * na * nm
            nv, nm, na = x["value"][:, k], x["mask"][:, k], adv[:, k]
        ok &= np.abs(adv - x["advantage"]).max() < 1e-5 and np.abs(x["value"] + adv - x["returnn"]).max() < 1e-5
    bad += not ok
    print("%-15s collect x2: %s  (%d frames, %d episodes ended, %.1fs)" % (
        env_id, "env half bit-identical, bookkeeping within tolerance" if ok else "MISMATCH", 2 * n * T, episodes,
        time.time() - t0), flush=True)
    env.close()
print("collect soak:", "all ok" if not bad else "%d configuration(s) differ" % bad)
sys.exit(1 if bad else 0)
