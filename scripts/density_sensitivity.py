"""How much the unverifiable geom density of point.xml (1, could be 5) moves the results on
the evaluation seeds 1000000-1000099 (CPU, oracle only; run by hand: python scripts/density_sensitivity.py).
Per task: the scripted pi_greedy episode of every seed under both densities -- visit order, step index of every
visit, termination step, return.  Printed as the table of DESIGN.md section 0."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def constants(density):
    m_s = density * 4.0 / 3.0 * math.pi * 0.1 ** 3
    m_b = density * 0.1 ** 3
    mass = m_s + m_b
    return dict(mass=mass, com_x=0.1 * m_b / mass,
                inertia_zz=0.4 * m_s * 0.01 + m_b * (0.05 ** 2 + 0.05 ** 2) / 3.0 + m_b * 0.01)


def episode(cfg, seed):
    env = O.OracleEnv(cfg)
    o, zo = env.reset(seed)
    visits, ret, t = [], 0.0, 0
    while True:
        a = env.policy(O.POLICY_GREEDY, o, zo, 0, t)
        r, done, goal = env.step(a)
        t += 1
        ret += r
        if env.e.last_visit >= 0:
            visits.append((t, env.e.last_visit))
        if done:
            return visits, t, ret, goal
        o, zo = env.obs()


rows = []
for name, task, Z in (("PointTSP-v0", 0, 15), ("PointTTSP-v0", 1, 15), ("ColourMatch-v0", 2, 6)):
    res = {}
    for d in (1.0, 5.0):
        cfg = O.default_config(task, Z, **constants(d))
        res[d] = [episode(cfg, s) for s in range(1000000, 1000100)]
    same_order = sum([z for _, z in a[0]] == [z for _, z in b[0]] for a, b in zip(res[1.0], res[5.0]))
    same_steps = sum(a[0] == b[0] for a, b in zip(res[1.0], res[5.0]))
    dt = np.array([b[1] - a[1] for a, b in zip(res[1.0], res[5.0])])
    dr = np.array([b[2] - a[2] for a, b in zip(res[1.0], res[5.0])])
    first = np.array([b[0][0][0] - a[0][0][0] for a, b in zip(res[1.0], res[5.0]) if a[0] and b[0]])
    rows.append((name, same_order, same_steps, dt, dr, first,
                 np.mean([a[2] for a in res[1.0]]), np.mean([b[2] for b in res[5.0]])))
print("| task | same visit order | same visit steps | termination step (d5 - d1): mean / max abs | first visit step: mean shift | mean return d1 / d5 |")
print("|---|---|---|---|---|---|")
for name, so, ss, dt, dr, first, r1, r5 in rows:
    print(f"| {name} | {so}/100 | {ss}/100 | {dt.mean():+.1f} / {np.abs(dt).max()} | {first.mean():+.1f} | {r1:.3f} / {r5:.3f} |")
