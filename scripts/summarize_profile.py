"""Condense the rocprofv3 outputs of scripts/profile_round.sh into <out>/summary.json and <out>/traffic.json.

traffic.json is what bench.py reads (copy it to profiles/traffic.json): per workload and launch mode the HBM bytes of the
dominant kernel as `fixed per launch + per step` and the VALU-issue counters, per SIMD and step.

  * FETCH_SIZE / WRITE_SIZE are in KiB and come from separate passes; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM:
    gfx950 reports half the bytes of wide coalesced reads), WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
  * persistent kernel: a launch covers 256 steps or, at the end of a rollout call, fewer; the step count of every
    dispatch follows from the bench arguments (settle, warm-up, timed), so bytes = fixed + per_step * steps is fitted
    over ALL dispatches.  per-step kernel: the mean over ALL dispatches (round 1 kept only dispatches > 0.9 * max,
    which for a one-step kernel selected the reset-burst outliers).
"""
import collections, csv, glob, json, os, sys

out = sys.argv[1]
tag = os.path.basename(os.path.normpath(out))
KERNELS = ("k_rollout_lane_ext", "k_rollout_lane", "k_step_lane", "k_policy_lane", "k_reset_lane", "k_mlp_zone1", "k_mlp_head")
N_SIMD = 256 * 4
CHUNK = 256
SETTLE, WARMUP, STEPS = 6000 - 256, 256, 512      # bench.py --steps 512 --warmup 256 (profile_round.sh)
BIG_WARMUP, BIG_STEPS = 128, 512                   # the batch-size sweep: --no-settle --warmup 128 --steps 512
BIG_PS_WARMUP, BIG_PS_STEPS = 16, 64               # ... and its per-step passes


def short(name):
    # the action-chunk form of the persistent kernel is its own row (k_rollout_lane<TASK, ZT, true>)
    if "k_rollout_lane" in name and ("ELb1E" in name or ", true>" in name or ",true>" in name):
        return "k_rollout_lane_ext"
    for k in KERNELS:
        if k in name:
            return k
    return name[:40]


def launch_steps(no_settle, warmup=WARMUP, timed=STEPS):
    seq = []
    for k in ([] if no_settle else [SETTLE]) + [warmup, timed]:
        seq += [CHUNK] * (k // CHUNK) + ([k % CHUNK] if k % CHUNK else [])
    return seq


def kernel_sources_sha():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return list(bench.KERNEL_SOURCES), bench.kernel_sources_sha()


def counter_rows(dirname):
    fs = glob.glob(os.path.join(out, dirname, "*", "*counter_collection.csv"))
    return list(csv.DictReader(open(fs[0]))) if fs else []


def per_dispatch(rows, kern, counter):
    """Counter value of each dispatch of `kern`, in dispatch order (values of one dispatch summed over dimensions)."""
    agg = collections.OrderedDict()
    for r in rows:
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            agg[int(r["Dispatch_Id"])] = agg.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [agg[k] for k in sorted(agg)]


def fit(values, steps):
    """Least squares values = a + b * steps; (a, b).  One distinct step count -> a = 0."""
    n = min(len(values), len(steps))
    values, steps = values[:n], steps[:n]
    if len(set(steps)) < 2:
        return 0.0, sum(values) / max(sum(steps), 1)
    mx, my = sum(steps) / n, sum(values) / n
    b = sum((x - mx) * (y - my) for x, y in zip(steps, values)) / sum((x - mx) ** 2 for x in steps)
    if b < 0:           # a counter that does not grow with the step count (state reads): all of it is per launch
        return my, 0.0
    return my - b * mx, b


summary, traffic = {}, {}
for sub in ("stats", "stats_per_step"):
    for f in glob.glob(os.path.join(out, sub, "*", "*kernel_stats.csv")):
        with open(os.path.join(out, f"{sub}_kernel_stats.csv"), "w") as g:
            g.write(open(f).read())
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k in KERNELS:
                summary.setdefault(sub, {})[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                  "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                                  "pct": float(r["Percentage"])}

workloads = sorted({os.path.basename(d)[len("pmc_fetch_"):].rsplit("_", 1)[0].replace("_per", "")
                    for d in glob.glob(os.path.join(out, "pmc_fetch_*")) if os.path.isdir(d)})
for w in workloads:                       # "PointTSP-25" (N = 65536) or "PointTSP-25@1048576" (the batch-size sweep)
    wname, n_env = (w.split("@")[0], int(w.split("@")[1])) if "@" in w else (w, 65536)
    big = "@" in w
    for mode, kern in (("persistent", "k_rollout_lane"), ("per_step", "k_step_lane")):
        steps = (launch_steps(True, BIG_WARMUP, BIG_STEPS) if big else launch_steps(no_settle=False)) \
            if mode == "persistent" else None
        ent = {"kernel": kern, "source": f"profiles/{tag}/summary.json", "n_env": n_env,
               "method": "rocprofv3 --pmc, one counter group per run; FETCH_SIZE (KiB) x2 per MI355X_MICROARCH.md, "
                         "WRITE_SIZE (KiB) as is; persistent: least-squares fixed + per-step over all dispatches of "
                         "`bench.py --steps 512 --warmup 256`; per_step: mean over all dispatches"}
        for name, key, scale in ((f"pmc_fetch_{w}_{mode}", "FETCH_SIZE", 2048.0), (f"pmc_write_{w}_{mode}", "WRITE_SIZE", 1024.0)):
            rows = counter_rows(name)
            vals = [v * scale for v in per_dispatch(rows, kern, key)]
            if not vals:
                continue
            keep = [r for r in rows if kern in r["Kernel_Name"]][:4]
            with open(os.path.join(out, f"{name}_head.csv"), "w") as g:   # a few raw rows as evidence
                wr = csv.DictWriter(g, fieldnames=list(keep[0].keys()))
                wr.writeheader()
                wr.writerows(keep)
            if mode == "persistent":
                a, b = fit(vals, steps)
            else:
                a, b = 0.0, sum(vals) / len(vals)
            ent[key.lower() + "_fixed"] = a
            ent[key.lower() + "_per_step"] = b
            ent[key.lower() + "_dispatches"] = len(vals)
            if mode == "per_step":
                ent[key.lower() + "_min_max"] = [min(vals), max(vals)]
        if "fetch_size_per_step" in ent and "write_size_per_step" in ent:
            ent["hbm_bytes_per_step"] = ent["fetch_size_per_step"] + ent["write_size_per_step"]
            ent["hbm_bytes_fixed_per_launch"] = ent["fetch_size_fixed"] + ent["write_size_fixed"]
        rows = counter_rows(f"pmc_sq_{w}_{mode}")
        if rows:
            def per_step_of(counter):
                vals = per_dispatch(rows, kern, counter)
                if not vals:
                    return None
                if mode == "persistent":
                    full = [v / CHUNK for v, s in zip(vals, steps) if s == CHUNK]
                    return sum(full) / len(full)
                return sum(vals) / len(vals)
            valu, salu, gui = per_step_of("SQ_INSTS_VALU"), per_step_of("SQ_INSTS_SALU"), per_step_of("GRBM_GUI_ACTIVE")
            if valu is not None:
                ent["valu_insts_per_simd_step"] = valu / N_SIMD
            if salu is not None:
                ent["salu_insts_per_simd_step"] = salu / N_SIMD
            if gui is not None:
                ent["gpu_cycles_per_step"] = gui / 8.0       # the counter sums the 8 XCDs
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
                v = per_step_of(c)
                if v is not None:
                    ent[c.lower() + "_per_step"] = v
            # clock: GRBM_GUI_ACTIVE / 8 / duration of the same dispatches (kernel trace of the same run)
            kt = glob.glob(os.path.join(out, f"pmc_sq_{w}_{mode}", "*", "*kernel_trace.csv"))
            if kt and gui is not None:
                durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))
                        if kern in r["Kernel_Name"]]
                if mode == "persistent":
                    durs = [d / CHUNK for d, s in zip(durs, steps) if s == CHUNK]
                if durs:
                    ent["pmc_run_ns_per_step"] = sum(durs) / len(durs)
                    ent["gpu_clock_ghz"] = round(ent["gpu_cycles_per_step"] / ent["pmc_run_ns_per_step"], 3)
        traffic.setdefault(f"{wname}@{n_env}", {})[mode] = ent
    # GRBM_GUI_ACTIVE / duration reads high on dispatches far below 0.3 ms (MI355X_MICROARCH.md, DVFS give-back): the
    # per-step kernel's cycles are its measured duration at the clock of the same workload's persistent launches
    ps, pe = traffic[f"{wname}@{n_env}"].get("per_step"), traffic[f"{wname}@{n_env}"].get("persistent")
    if ps and pe and "gpu_clock_ghz" in pe and "pmc_run_ns_per_step" in ps:
        ps["gpu_clock_ghz"] = pe["gpu_clock_ghz"]
        ps["gpu_cycles_per_step"] = ps["pmc_run_ns_per_step"] * pe["gpu_clock_ghz"]
        ps["clock_note"] = "clock of the persistent launches of the same workload (short dispatches read high)"
# the action-chunk form (scripts/chunk_pmc.py: 256-step launches of k_rollout_lane<.., EXT> on PointTSP-25, fresh actions)
ext = {}
for name, key, scale in (("pmc_fetch_chunk", "FETCH_SIZE", 2048.0), ("pmc_write_chunk", "WRITE_SIZE", 1024.0)):
    rows = [r for r in counter_rows(name) if short(r["Kernel_Name"]) == "k_rollout_lane_ext"]
    vals = [v * scale / CHUNK for v in per_dispatch(rows, "k_rollout_lane", key)]
    if vals:
        ext[key.lower() + "_per_step"] = sum(vals) / len(vals)
        ext[key.lower() + "_dispatches"] = len(vals)
        with open(os.path.join(out, f"{name}_head.csv"), "w") as g:
            wr = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
            wr.writeheader()
            wr.writerows(rows[:4])
if len(ext) == 4:
    ext.update({"kernel": "k_rollout_lane<EXT>", "n_env": 65536, "source": f"profiles/{tag}/summary.json",
                "hbm_bytes_per_step": ext["fetch_size_per_step"] + ext["write_size_per_step"],
                "method": "scripts/chunk_pmc.py under rocprofv3 --pmc (one counter per run): every 256-step launch / 256"})
    traffic.setdefault("PointTSP-25@65536", {})["action_chunk"] = ext
names, sha = kernel_sources_sha()
traffic["_meta"] = {"kernel_sources": names, "kernel_sources_sha256": sha, "measured_in": f"profiles/{tag}",
                    "note": "bench.py marks the record stale (aux.traffic_stale) and tests/test_bench_contract.py fails "
                            "when the kernel sources no longer hash to this"}
summary["pmc"] = traffic
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
