"""Diagnostic: the persistent kernel's launch slice (envs per launch, zenv_set_rollout_slice) against throughput for one
workload at one batch size.  usage: python scripts/slice_sweep.py ColourMatch-6 1048576 0 65536 131072 262144"""
import json
import subprocess
import sys

w, n = sys.argv[1], sys.argv[2]
for sl in sys.argv[3:]:
    out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--envs-per-gpu", n, "--rollout-slice", sl, "--no-sweep",
                          "--no-mlp", "--no-cpu-baseline", "--no-per-step", "--steps", "2048", "--warmup", "512"],
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    r = d["roofline"]
    print(f"{w} N {n} slice {sl}: {d['value'] / 1e9:.2f} G env-steps/s, {r['kernel_us_per_step']} us per step, frac {r['frac']}, "
          f"spot check {d['aux']['parity_spot_check']}", flush=True)
