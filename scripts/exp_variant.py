"""Diagnostic: time the step kernel of a -D<flags> variant build (never the shipped library)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combinatorial_rl_tasks_amd.build as B
if B.under_profiler() and os.environ.get("EXP_SHIPPED") != "1":
    raise SystemExit("this script compiles a variant library: run it without rocprofv3, or build the variant first "
                     "(scripts/build_variant.py) and profile a script that loads it through ZENV_LIB_PATH")
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
wl = [a for a in sys.argv[1:] if not a.startswith("-D")]
so = os.path.join(ROOT, "gpurun_out", "libzenv_exp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
import combinatorial_rl_tasks_amd._native as nat
if os.environ.get("EXP_SHIPPED") == "1":       # the library as it was last built in-tree (for an A/B on one box)
    flags = ["<shipped library>"]
else:
    subprocess.run([B._hipcc()] + B.FLAGS + flags + ["-o", so] + [os.path.join(B.CSRC, s) for s in B.SOURCES], check=True)
    nat.LIB_PATH = so
import combinatorial_rl_tasks_amd as Z
wl0 = [w for w in wl if w not in ("steady", "perstep")]
mode = "per_step" if "perstep" in wl else "persistent"
task, zones, keep = {"tsp": (0, 25, .4), "timed": (1, 25, .4), "colour": (2, 6, .55), "tsp15": (0, 15, .55)}[wl0[0] if wl0 else "tsp"]
n = 65536
cfg = Z.default_config(task, zones, zones_keepout=keep)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n); env.reset()
if "steady" in wl:      # past the power controller's transient, envs desynchronised
    env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
    env.rollout(6000, Z.POLICY_GREEDY, mode=mode)
    T = 4096
else:
    env.rollout(30, Z.POLICY_GREEDY, mode=mode)
    T = 300
tot, k = env.rollout(T, Z.POLICY_GREEDY, time_step_kernel=True, mode=mode, event_stride=16)
print(flags, wl, "kernel avg us %.2f  loop us/step %.2f" % (k * 1e3, tot / T * 1e3))
