"""Experiment: the per-step kernel (k_step_lane) over 65 536 envs as ONE handle vs TWO handles of 32 768 envs (and FOUR
of 16 384) on their own streams, driven by host threads -- do the phases of independent launches (load, compute, store)
interleave better than the lock-step phases of one launch?  Fused scripted policy, rollout mode per_step."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z
WL = {"PointTSP-25": (0, 25, .4), "ColourMatch-6": (2, 6, .55)}
T = 4000
for w, (task, zones, keep) in WL.items():
    def make(n, first):
        cfg = Z.default_config(task, zones, zones_keepout=keep)
        env = Z.ZoneVecEnv(cfg, n); env.build_bank(1 + first, 4 * n, n_threads=16); env.schedule_sequential(stride=n); env.reset()
        env.rollout(3000, Z.POLICY_GREEDY); env.rollout(500, Z.POLICY_GREEDY, mode="per_step")
        return env
    one = make(65536, 0)
    ms, _ = one.rollout(T, Z.POLICY_GREEDY, mode="per_step")
    print("%-14s one handle, 65536 envs: %.2f us per step" % (w, ms / T * 1e3), flush=True)
    one.close()
    for parts in (2, 4):
        n = 65536 // parts
        hs = [make(n, i * n) for i in range(parts)]
        def run(e): e.rollout(T, Z.POLICY_GREEDY, mode="per_step")
        for rep in range(2):
            th = [threading.Thread(target=run, args=(e,)) for e in hs]
            t0 = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            dt = time.perf_counter() - t0
            print("%-14s %d handles x %d envs, %d streams: %.2f us per step of the 65536" % (w, parts, n, parts, dt / T * 1e6), flush=True)
        for e in hs: e.close()
