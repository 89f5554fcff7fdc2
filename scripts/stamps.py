"""Diagnostic: per-phase timeline of the step kernel from in-kernel s_memrealtime stamps.
Builds a separate -DZENV_STAMPS library (never the shipped one) and prints medians."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combinatorial_rl_tasks_amd.build as B
if B.under_profiler():
    raise SystemExit("this script compiles a variant library: run it without rocprofv3, or build the variant first "
                     "(scripts/build_variant.py) and profile a script that loads it through ZENV_LIB_PATH")
so = os.path.join(B.LIB_DIR, "variants", "libzenv_stamps.so")      # travels to the GPU box when built here (cross-compiled)
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(os.path.join(B.CSRC, f)) for f in os.listdir(B.CSRC)):
    subprocess.run([B._hipcc()] + B.FLAGS + ["-DZENV_STAMPS"] + os.environ.get("ZENV_EXTRA_FLAGS", "").split() + ["-o", so] + [os.path.join(B.CSRC, s) for s in B.SOURCES], check=True)
import combinatorial_rl_tasks_amd._native as nat
nat.LIB_PATH = so
import combinatorial_rl_tasks_amd as Z
task, zones, keep = {"tsp": (0, 25, .4), "timed": (1, 25, .4), "colour": (2, 6, .55)}[sys.argv[1] if len(sys.argv) > 1 else "tsp"]
fused = "unfused" not in sys.argv
n = 65536
cfg = Z.default_config(task, zones, zones_keepout=keep)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n); env.reset()
warm = int([a for a in sys.argv if a.startswith('warm=')][0][5:]) if any(a.startswith('warm=') for a in sys.argv) else 30
env.build_bank_seeds(np.concatenate([1 + np.arange(n) + k * n for k in range(3)])); env.schedule_sequential(first=np.arange(n, dtype=np.int32), stride=n); env.reset()
mode = "per_step" if fused else "unfused"      # K1 either way: with its fused action source, or behind a K3 launch
env.rollout(warm, Z.POLICY_GREEDY, mode=mode)
L = nat.lib(); L.zenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
acc, kus = [], []
burst = int([a for a in sys.argv if a.startswith('burst=')][0][6:]) if any(a.startswith('burst=') for a in sys.argv) else 1
for it in range(20):
    if burst > 1:       # the LAST launch of a back-to-back burst (no launch ramp of an idle queue); its duration: the burst's mean
        ms, _ = env.rollout(burst, Z.POLICY_GREEDY, mode=mode)
        k_ms = ms / burst
    else:
        _, k_ms = env.rollout(1, Z.POLICY_GREEDY, mode=mode, time_step_kernel=True)
    kus.append(k_ms * 1e3)
    buf = np.zeros((n // 64, 16), np.uint64)
    nat.check(L.zenv_debug_stamps(env._h, buf.ctypes.data, buf.size))
    acc.append(buf.astype(np.int64))
a = np.stack(acc)                       # [it, block, slot], 10 ns ticks
t0 = a[:, :, [0, 8]].min(axis=(1, 2), keepdims=True)   # first wave start of the launch
rel = (a - t0) * 0.01                   # us
names = {0: "Z start", 1: "Z pose ready", 14: "Z zone loop done", 6: "Z finalize math done", 15: "Z prefetch issued", 2: "Z at barrier (finalize+reset done)", 3: "Z flush issued",
         4: "Z past barrier", 8: "P start", 9: "P loads landed", 10: "P physics done", 11: "P past barrier",
         12: "P obs8 stored", 13: "P end"}
print(sys.argv[1:], "fused" if fused else "unfused", "episodes so far", int(env.get(Z.F_EPISODES).sum()))
ends = np.maximum(rel[:, :, 3], rel[:, :, 13])
span = (a[:, :, [2, 3, 4, 10, 11, 12, 13]].max(axis=(1, 2)) - a[:, :, [0, 8]].min(axis=(1, 2))) * 0.01
print("dispatch (begin/end events) median %.2f us | first stamp -> last stamp median %.2f us | difference (launch ramp before the "
      "first wave + drain after the last) %.2f us" % (np.median(kus), np.median(span), np.median(np.array(kus) - span)))
print("block end: median %.2f p90 %.2f p99 %.2f max %.2f" % (np.median(ends), np.percentile(ends, 90), np.percentile(ends, 99), ends.max()))
slow = ends > np.percentile(ends, 99)
rs = a[:, :, 5] > 0
print("blocks with a reset this step: %.1f per launch" % (rs.sum() / a.shape[0]))
if rs.any():
    for k, nm in ((1, "pose"), (5, "reset loop entry"), (6, "bank rows landed, entries written"), (7, "reset done (last)"), (2, "zone wave at barrier")):
        print("   reset blocks: %-34s median %.2f p90 %.2f" % (nm, np.median(rel[:, :, k][rs]), np.percentile(rel[:, :, k][rs], 90)))
for k in (1, 2, 4, 3, 10, 12, 13):
    print("  slot", k, "all median %.2f   slowest-1%% median %.2f" % (np.median(rel[:, :, k]), np.median(rel[:, :, k][slow])))
for k, v in names.items():
    x = rel[:, :, k]
    print(f"{v:32s} median {np.median(x):7.2f}  p10 {np.percentile(x,10):7.2f}  p90 {np.percentile(x,90):7.2f}  max {x.max():7.2f} us")
if any(a.startswith("json=") for a in sys.argv):
    import json
    path = [a for a in sys.argv if a.startswith("json=")][0][5:]
    def q(x):
        return {"median": round(float(np.median(x)), 3), "p10": round(float(np.percentile(x, 10)), 3),
                "p90": round(float(np.percentile(x, 90)), 3), "p99": round(float(np.percentile(x, 99)), 3),
                "max": round(float(x.max()), 3)}
    rec = {"kernel": "k_step_lane", "workload": sys.argv[1], "n_env": n, "launches_sampled": len(acc),
           "action_source": "fused scripted policy (rollout mode per_step)" if fused else "stand-alone policy kernel before the step kernel",
           "unit": "us since the first wave of the launch started (s_memrealtime, 10 ns ticks); medians over all 1024 workgroups x sampled launches",
           "note": "isolated launches (host synchronises between them): the launch ramp is longer than in a back-to-back loop",
           "dispatch_us": q(np.array(kus)), "first_to_last_stamp_us": q(span),
           "ramp_plus_drain_us": q(np.array(kus) - span), "block_end_us": q(ends),
           "blocks_with_a_reset_per_launch": round(float(rs.sum() / a.shape[0]), 1),
           "phases": {v: q(rel[:, :, k]) for k, v in names.items()}}
    if rs.any():
        rec["reset_blocks"] = {nm: q(rel[:, :, k][rs]) for k, nm in ((1, "pose ready"), (5, "reset loop entry"), (6, "bank rows landed (last reset)"), (7, "reset done (last)"), (2, "zone wave at barrier"))}
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    json.dump(rec, open(path, "w"), indent=1)
if "blocks" in sys.argv:
    late_mask = rel[:, :, 2] > 12
    for k in (0, 1, 5, 6, 7, 2):
        print("late blocks: slot", k, "median", round(float(np.median(rel[:, :, k][late_mask])), 2),
              " normal median", round(float(np.median(rel[:, :, k][~late_mask])), 2))
    zp = rel[:, :, 2] - rel[:, :, 1]          # zone pass duration per block
    m = np.median(zp, axis=0)
    order = np.argsort(-m)
    print("slowest blocks (zone pass us):", [(int(b), round(float(m[b]), 2)) for b in order[:24]])
    print("zone pass by block%8:", [round(float(np.median(m[i::8])), 2) for i in range(8)])
    print("zone pass by (block//8)%32 first 8:", [round(float(np.median(m[(np.arange(len(m)) // 8) % 32 == i])), 2) for i in range(8)])
    late = rel[:, :, 2].max(axis=0)
    print("count blocks with zone-pass-done > 12us in any iter:", int((late > 12).sum()), "of", len(late))
    it_late = (rel[:, :, 2] > 12).sum(axis=1)
    print("per-iteration count of late blocks:", it_late.tolist())
