"""Diagnostic: where a SHORT persistent launch spends its time (the driver's bench command is one launch of 20 steps).
In-kernel s_memrealtime stamps of every workgroup (separate -DZENV_STAMPS build, never the shipped library), relative to
the first workgroup's entry: entry spread, prologue (state in, static entries, barrier), step loop, write-back, and the
stream wave's first and last flush.  usage: python scripts/stamps_launch.py [tsp|timed|colour|tsp15] [steps=20]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combinatorial_rl_tasks_amd.build as B
if B.under_profiler():
    raise SystemExit("this script compiles a variant library: run it without rocprofv3")
so = os.path.join(ROOT, "gpurun_out", "libzenv_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run([B._hipcc()] + B.FLAGS + ["-DZENV_STAMPS"] + flags + ["-o", so] + [os.path.join(B.CSRC, s) for s in B.SOURCES], check=True)
import combinatorial_rl_tasks_amd._native as nat
nat.LIB_PATH = so
import combinatorial_rl_tasks_amd as Z
wl = [a for a in sys.argv[1:] if not a.startswith("-D")]
task, zones, keep = {"tsp": (0, 25, .4), "timed": (1, 25, .4), "colour": (2, 6, .55), "tsp15": (0, 15, .55)}[wl[0] if wl else "tsp"]
K = int(wl[1]) if len(wl) > 1 else 20
n = 65536
cfg = Z.default_config(task, zones, zones_keepout=keep)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
env.rollout(6000, Z.POLICY_GREEDY)
L = nat.lib(); L.zenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rows = []
for it in range(24):
    ms, kern = env.rollout(K, Z.POLICY_GREEDY, time_step_kernel=True)
    buf = np.zeros((n // 64, 16), np.uint64)
    nat.check(L.zenv_debug_stamps(env._h, buf.ctypes.data, buf.size))
    a = buf.astype(np.int64) * 0.01            # us
    t0 = a[:, 12].min()
    rows.append([np.median(a[:, 12] - t0), (a[:, 12] - t0).max(),
                 np.median(a[:, 13] - a[:, 12]), np.median(a[:, 6] - a[:, 13]), np.median(a[:, 14] - a[:, 13]),
                 np.median(a[:, 7] - a[:, 14]), np.median(a[:, 15] - a[:, 14]),
                 max(a[:, 15].max(), a[:, 7].max()) - t0, kern * K * 1e3])
r = np.median(np.array(rows[4:]), axis=0)
print("%s, %d steps per launch, medians over 20 launches (us)" % (wl[0] if wl else "tsp", K))
for nm, v in zip(("entry: median workgroup after the first", "entry: last workgroup after the first",
                  "env wave: prologue (entry -> loop)", "stream wave: first flush done after the loop began",
                  "env wave: step loop", "stream wave: last flush issued after the env wave's loop ended",
                  "env wave: write-back", "first entry -> last stamp of the launch", "dispatch duration (begin/end events)"), r):
    print("  %-62s %7.2f" % (nm, v))
