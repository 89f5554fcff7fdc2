"""Step-kernel time of the two layouts on the bench workloads: the lane-per-env K1 and the
wave-per-env K1w (the layout north_star spells out), same rollout loop (policy kernel + step kernel per
step), HIP events around the step kernel only.  Run on the GPU box:  python scripts/layout_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z  # noqa: E402

N = int(os.environ.get("LAYOUT_N", 65536))
STEPS = int(os.environ.get("LAYOUT_STEPS", 600))
WORK = [("PointTSP-25", 0, 25, 0.40), ("TimedTSP-25", 1, 25, 0.40), ("ColourMatch-6", 2, 6, 0.55),
        ("PointTSP-15", 0, 15, 0.55)]
for name, task, zones, keepout in WORK:
    row = {"workload": name, "num_envs": N, "steps": STEPS}
    for label, kernel in (("lane_per_env_us", 0), ("wave_per_env_us", 1)):
        cfg = Z.default_config(task, zones, zones_keepout=keepout, kernel=kernel)
        env = Z.ZoneVecEnv(cfg, N)
        env.build_bank(1, 4 * N)
        env.schedule_sequential(stride=N)
        env.reset()
        env.rollout(200, Z.POLICY_GREEDY, mode="unfused")
        _, k_ms = env.rollout(STEPS, Z.POLICY_GREEDY, mode="unfused", time_step_kernel=True, event_stride=4)
        row[label] = round(1e3 * k_ms, 2)
        env.close()
    row["ratio"] = round(row["wave_per_env_us"] / row["lane_per_env_us"], 1)
    print(json.dumps(row), flush=True)
