"""Diagnostic: same-box A/B of the per-step kernel (k_step_lane) between library builds.
usage: python scripts/k1_ab.py <libA.so> <libB.so> ...   (each timed in its own child process, A B A B ...)
Prints, per workload and library: dispatch time (begin/end events of every 4th launch), back-to-back loop time per
step, with the fused scripted policy (rollout mode per_step) and with external actions (zenv_step, no policy)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = {"PointTSP-25": (0, 25, .4), "TimedTSP-25": (1, 25, .4), "ColourMatch-6": (2, 6, .55), "PointTSP-15": (0, 15, .55)}


def child(lib):
    sys.path.insert(0, ROOT)
    import numpy as np
    import combinatorial_rl_tasks_amd._native as nat
    nat.LIB_PATH = lib
    import combinatorial_rl_tasks_amd as Z
    n = 65536
    for w, (task, zones, keep) in WL.items():
        cfg = Z.default_config(task, zones, zones_keepout=keep)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(1, 4 * n, n_threads=16)
        env.schedule_sequential(stride=n)
        env.reset()
        env.rollout(6000, Z.POLICY_GREEDY, mode="persistent")          # settle the clock, desynchronise the envs
        env.rollout(500, Z.POLICY_GREEDY, mode="per_step")
        tot, k = env.rollout(4000, Z.POLICY_GREEDY, time_step_kernel=True, mode="per_step", event_stride=4)
        tot2, _ = env.rollout(4000, Z.POLICY_GREEDY, mode="per_step")
        # external actions: the zenv_step path (no fused policy), device-resident action buffer
        import time
        env.sync()
        t0 = time.perf_counter()
        for _ in range(3000):
            env.step(None, auto_reset=True)
        env.sync()
        ext = (time.perf_counter() - t0) / 3000 * 1e6
        print("%-14s %-26s fused: dispatch %6.2f us, loop %6.2f us/step | zenv_step loop %6.2f us/step" %
              (w, os.path.basename(lib), k * 1e3, tot2 / 4000 * 1e3, ext), flush=True)
        env.close()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for rep in range(2):
            for lib in sys.argv[1:]:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib)], check=True)
