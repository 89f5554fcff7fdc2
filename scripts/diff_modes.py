"""Diagnostic: which per-env fields / state arrays differ between two rollout modes (auto-reset on)?
usage: python scripts/diff_modes.py [modeA modeB] [n=65536] [task=1] [zones=25] [steps=150]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
modes = [a for a in sys.argv[1:] if "=" not in a] or ["persistent", "per_step"]
n, task, Zn, steps = int(kw.get("n", 65536)), int(kw.get("task", 1)), int(kw.get("zones", 25)), int(kw.get("steps", 150))
cfg = Z.default_config(task, Zn, zones_keepout=0.40 if Zn > 15 else 0.55)
fields = {k: getattr(Z, k) for k in ("F_OBS", "F_ZONE_OBS", "F_REWARD", "F_DONE", "F_GOAL_MET", "F_EP_RETURN", "F_EP_LEN",
                                      "F_LAST_RETURN", "F_LAST_LEN", "F_EPISODES", "F_VISIT_COUNT", "F_SEED", "F_ACTIONS")}
runs = []
for mode in modes[:2]:
    env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n, n_threads=16); env.reset()
    env.rollout(steps, Z.POLICY_GREEDY, mode=mode)
    got = {k: env.get(f) for k, f in fields.items()}
    got.update({"dbg_" + k: v for k, v in env.debug_state().items()})
    got["blob"] = np.frombuffer(env.get_state(), np.uint8).copy()
    runs.append(got); env.close()
a, b = runs
for k in a:
    if not np.array_equal(a[k], b[k], equal_nan=True):
        bad = np.flatnonzero((a[k] != b[k]).reshape(len(a[k]), -1).any(1)) if k != "blob" else np.flatnonzero(a[k] != b[k])
        print(k, "differs at", bad.size, "envs/bytes; first", bad[:8], "episodes there", a["F_EPISODES"][bad[:8]] if k != "blob" else "")
        if k == "F_ACTIONS":
            print("  ", a[k][bad[:4]], b[k][bad[:4]], "ep_len", a["F_EP_LEN"][bad[:4]])
print(modes, "n", n, "compared", len(a), "arrays")
