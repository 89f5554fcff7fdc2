"""Diagnostic: what a SHORT launch of the persistent kernel costs (the driver's bench command is --steps 20: one launch
of 20 steps).  For K steps per launch: mean wall time of back-to-back launches and of a launch + synchronisation,
PointTSP-25 (or argv[1]: timed / colour / tsp15), N = 65 536.  A least-squares line gives fixed + per-step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import combinatorial_rl_tasks_amd as Z
task, zones, keep = {"tsp": (0, 25, .4), "timed": (1, 25, .4), "colour": (2, 6, .55), "tsp15": (0, 15, .55)}[sys.argv[1] if len(sys.argv) > 1 else "tsp"]
n = 65536
cfg = Z.default_config(task, zones, zones_keepout=keep)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
env.rollout(6000, Z.POLICY_GREEDY)
ks, b2b, synced, waited = [1, 2, 4, 8, 16, 20, 32, 64, 128, 256], [], [], []
for k in ks:
    reps = max(8, 2048 // k)
    env.rollout(k, Z.POLICY_GREEDY); env.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        env.rollout(k, Z.POLICY_GREEDY, wait=False)
    env.sync()
    b2b.append((time.perf_counter() - t0) / reps * 1e6)
    ts = []
    for _ in range(reps):
        env.sync()
        t0 = time.perf_counter()
        env.rollout(k, Z.POLICY_GREEDY, wait=False)
        env.sync()
        ts.append((time.perf_counter() - t0) * 1e6)
    synced.append(float(np.median(ts)))
    ts = []
    for _ in range(reps):               # what bench.py's timed region does: the waiting form of the call, then a sync
        env.sync()
        t0 = time.perf_counter()
        env.rollout(k, Z.POLICY_GREEDY)
        env.sync()
        ts.append((time.perf_counter() - t0) * 1e6)
    waited.append(float(np.median(ts)))
    print("K = %3d: back to back %8.1f us per launch (%.2f per step) | launch + sync %8.1f us (%.2f per step) | waiting call + sync %8.1f us" % (
        k, b2b[-1], b2b[-1] / k, synced[-1], synced[-1] / k, waited[-1]), flush=True)
for name, y in (("back to back", b2b), ("launch + sync", synced), ("waiting call + sync", waited)):
    A = np.vstack([np.ones(len(ks)), ks]).T
    (a, b), *_ = np.linalg.lstsq(A, np.array(y), rcond=None)
    print("%s: %.1f us fixed + %.3f us per step" % (name, a, b))
