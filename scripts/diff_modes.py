"""Diagnostic: where does the state blob of two rollout modes differ (no auto-reset)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
n = 130
cfg = Z.default_config(1, 25, zones_keepout=0.40, num_steps=90)
blobs = []
for mode in ("persistent", "per_step"):
    env = Z.ZoneVecEnv(cfg, n); env.build_bank(3, n); env.reset()
    env.rollout(20, Z.POLICY_GREEDY, auto_reset=False, mode=mode)
    env.rollout(130, Z.POLICY_GREEDY, auto_reset=False, mode=mode)
    blobs.append(np.frombuffer(env.get_state(), np.uint8).copy()); env.close()
a, b = blobs
d = np.nonzero(a != b)[0]
print("blob bytes", a.size, "differing", d.size)
Zn, F, ZH = 25, 7, 13
sizes = [("qa",16),("qb",16),("qc",16),("fa",16),("fb",16),("zxy",16*Zn),("zpf",16*ZH),("vis",4),("tmax",4*Zn),("colpack",8),
 ("cooldown",Zn),("goal_dist",4),("steps",4),("done_state",1),("ep_return",8),("last_return",8),("last_len",4),("episodes",4),
 ("visit_count",4),("seed",8),("slot_first",4),("episode_idx",4),("pcg",32),("pcg_buf",8),("obs",32),("zone_obs",4*Zn*F),
 ("reward",4),("actions",8),("done_out",1),("goal_met",1)]
off = 16
for name, per in sizes:
    sz = per * n
    k = ((d >= off) & (d < off + sz)).sum()
    if k: print(name, "differs in", k, "bytes; first rel offset", (d[(d >= off)][0] - off), "per-env", per)
    off += (sz + 255) // 256 * 256 if False else sz
print("end offset", off, "(blob", a.size, ")")
acts = []
for mode in ("persistent", "per_step"):
    env = Z.ZoneVecEnv(cfg, n); env.build_bank(3, n); env.reset()
    env.rollout(20, Z.POLICY_GREEDY, auto_reset=False, mode=mode)
    a1 = env.get(Z.F_ACTIONS).copy(); d1 = env.get(Z.F_DONE).copy()
    env.rollout(130, Z.POLICY_GREEDY, auto_reset=False, mode=mode)
    acts.append((a1, d1, env.get(Z.F_ACTIONS).copy(), env.get(Z.F_LAST_LEN).copy())); env.close()
print("after 20: equal", np.array_equal(acts[0][0], acts[1][0]))
print("after 150: equal", np.array_equal(acts[0][2], acts[1][2]))
for i in range(8):
    print(i, "len", acts[0][3][i], "done@20", acts[0][1][i], "pers", acts[0][0][i], acts[0][2][i], "step", acts[1][0][i], acts[1][2][i])
