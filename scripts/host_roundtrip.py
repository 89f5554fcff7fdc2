"""PCIe-inclusive rate of the host-policy surface: actions H2D + step + obs/zone_obs/reward/done D2H
per step (what a CPU-resident policy pays), with pageable numpy buffers and with page-locked ones
(ZoneVecEnv.pinned_array + results_into).  Never the bench `value`; quoted in DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = Z.default_config(0, 25, zones_keepout=0.40)
env = Z.ZoneVecEnv(cfg, n); env.build_bank(1, n); env.reset()
fields = (Z.F_OBS, Z.F_ZONE_OBS, Z.F_REWARD, Z.F_DONE, Z.F_GOAL_MET)
dtypes = (np.float32, np.float32, np.float32, np.uint8, np.uint8)
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        a = np.zeros((n, 2), np.float32)
        bufs = [np.empty(env._shape(f), t) for f, t in zip(fields, dtypes)]
    else:
        a = env.pinned_array((n, 2), np.float32); a[:] = 0
        bufs = [env.pinned_array(env._shape(f), t) for f, t in zip(fields, dtypes)]

    def step():
        env.step(a, auto_reset=True)
        env.results_into(fields, bufs)
    for _ in range(5): step()
    t0 = time.perf_counter(); K = 50
    for _ in range(K): step()
    dt = (time.perf_counter() - t0) / K
    mb = (a.nbytes + sum(b.nbytes for b in bufs)) / 1e6
    print(f"N={n}, {kind} host memory: {dt*1e3:.3f} ms/step host round trip ({mb:.1f} MB over PCIe, {mb/dt/1e3:.1f} GB/s) = {n/dt/1e6:.1f} M env-steps/s")
