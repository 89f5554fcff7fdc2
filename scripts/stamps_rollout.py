"""Diagnostic: timeline of the MIDDLE step of a persistent launch from in-kernel s_memrealtime stamps
(separate -DZENV_STAMPS build, never the shipped library)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combinatorial_rl_tasks_amd.build as B
if B.under_profiler():
    raise SystemExit("this script compiles a variant library: run it without rocprofv3, or build the variant first "
                     "(scripts/build_variant.py) and profile a script that loads it through ZENV_LIB_PATH")
so = os.path.join(ROOT, "gpurun_out", "libzenv_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run([B._hipcc()] + B.FLAGS + ["-DZENV_STAMPS"] + flags + ["-o", so] + [os.path.join(B.CSRC, s) for s in B.SOURCES], check=True)
import combinatorial_rl_tasks_amd._native as nat
nat.LIB_PATH = so
import combinatorial_rl_tasks_amd as Z
wl = [a for a in sys.argv[1:] if not a.startswith("-D")]
task, zones, keep = {"tsp": (0, 25, .4), "timed": (1, 25, .4), "colour": (2, 6, .55), "tsp15": (0, 15, .55)}[wl[0] if wl else "tsp"]
n = 65536
cfg = Z.default_config(task, zones, zones_keepout=keep)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
env.rollout(3000, Z.POLICY_GREEDY)      # past the clock transient, envs desynchronised
L = nat.lib(); L.zenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
acc = []
for it in range(12):
    env.rollout(64, Z.POLICY_GREEDY)
    buf = np.zeros((n // 64, 16), np.uint64)
    nat.check(L.zenv_debug_stamps(env._h, buf.ctypes.data, buf.size))
    acc.append(buf.astype(np.int64))
a = np.stack(acc) * 0.01                  # us
def d(x, y): return np.median(a[:, :, x] - a[:, :, y]), np.percentile(a[:, :, x] - a[:, :, y], 90)
print(flags, wl)
for nm, x, y in (("E zone pass + reward", 1, 0), ("E physics", 2, 1), ("E obs + policy (+reset)", 3, 2), ("E wait flushed(t-2)", 4, 3),
                 ("E publish + stores", 5, 4), ("E whole step", 5, 0), ("S wait published", 9, 8), ("S flush", 10, 9),
                 ("S period (flush end t-1 -> flush end t)", 10, 11), ("S lag: E step start -> S flush end", 10, 0)):
    m, p90 = d(x, y)
    print("%-44s median %6.2f us   p90 %6.2f" % (nm, m, p90))
