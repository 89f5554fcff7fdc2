"""Where the wall time of ONE short persistent launch goes (the driver's `--steps 20` call): the rollout call
(launch + event wait), the stream sync and torch.cuda.synchronize, with and without dispatch events.
usage: python scripts/launch_overhead.py [steps] [spin]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
import numpy as np
import torch
import combinatorial_rl_tasks_amd as Z
torch.cuda.init(); torch.zeros(1, device="cuda")
n = 65536
cfg = Z.default_config(0, 25, zones_keepout=0.40)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
env.rollout(512, Z.POLICY_GREEDY)
env.sync(); torch.cuda.synchronize()
for ev in (True, False):
    rows = []
    for rep in range(30):
        env.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot, k = env.rollout(steps, Z.POLICY_GREEDY, time_step_kernel=ev)
        t1 = time.perf_counter()
        env.sync()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, tot * 1e3, (k or 0) * steps * 1e3))
    r = np.median(np.array(rows[5:]), axis=0)
    print("events=%d steps=%d: rollout call %.1f us, env.sync %.1f, torch sync %.1f  | total %.1f; event-bracket %.1f us, kernel %.1f us"
          % (ev, steps, r[0], r[1], r[2], r[0] + r[1] + r[2], r[3], r[4]))
# an empty sync pair, for scale
t0 = time.perf_counter()
for _ in range(100):
    env.sync(); torch.cuda.synchronize()
print("idle env.sync + torch sync: %.1f us" % ((time.perf_counter() - t0) * 1e4))
# bench.py's own sequence: a long settle rollout, W warm-up steps, fence, ONE timed call of `steps` steps, fence
for rep in range(8):
    env.rollout(6000 - 5, Z.POLICY_GREEDY)
    env.rollout(5, Z.POLICY_GREEDY)
    env.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot, _ = env.rollout(steps, Z.POLICY_GREEDY, policy_seed=0x5EED, env_index0=0, auto_reset=True, time_step_kernel=False,
                         mode="persistent", event_stride=16)
    t1 = time.perf_counter()
    env.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("bench sequence: wall %.1f us (rollout call %.1f), bracket events %.1f us" % ((t2 - t0) * 1e6, (t1 - t0) * 1e6, tot * 1e3))
