"""Per-step latency of the reference-shaped surface at the batch sizes the reference itself uses (train_ppo.py: 16
procs): ParallelEnv.step (P tuples of dicts), step_arrays (struct of arrays), and the raw pieces.
usage: python scripts/small_batch_latency.py [env_id]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import combinatorial_rl_tasks_amd as Z
from combinatorial_rl_tasks_amd.envs.make_env import make_train_env
from combinatorial_rl_tasks_amd.penv import ParallelEnv
from combinatorial_rl_tasks_amd import _native as nat

env_id = sys.argv[1] if len(sys.argv) > 1 else "PointTSP-v0"
for P in (1, 16, 256, 4096):
    envs = [make_train_env(env_id, rng_seed=1 + 1000 * i) for i in range(P)]
    pe = ParallelEnv(envs)
    pe.reset()
    rs = np.random.RandomState(0)
    a = rs.uniform(-1, 1, (P, 2)).astype(np.float32)
    T = 300 if P <= 256 else 60
    for name, fn in (("ParallelEnv.step (tuples of dicts)", lambda: list(pe.step(a))),
                     ("step_arrays (struct of arrays)", lambda: pe.step_arrays(a)),
                     ("vec.step only (H2D + launch, no download)", lambda: pe.vec.step(a)),
                     ("vec.step + sync", lambda: (pe.vec.step(a), pe.vec.sync())),
                     # the platform's floor for "launch one kernel and wait for it": the uniform-action kernel writes 8 bytes per env
                     ("smallest kernel (uniform actions) + sync", lambda: (pe.vec.policy(Z.POLICY_UNIFORM), pe.vec.sync())),
                     ("sync of an idle stream", lambda: pe.vec.sync())):
        for _ in range(20):
            fn()
        pe.vec.sync()
        t0 = time.perf_counter()
        for _ in range(T):
            fn()
        pe.vec.sync()
        dt = (time.perf_counter() - t0) / T
        print("P %5d  %-44s %8.1f us/step  %10.0f env-steps/s" % (P, name, dt * 1e6, P / dt))
    pe.close()
# the single-env gym surface (what one `worker` process of the reference runs)
e = make_train_env(env_id, rng_seed=1)
e.reset()
a1 = np.array([0.5, 0.1], np.float32)
for _ in range(20):
    e.step(a1)
t0 = time.perf_counter()
for _ in range(300):
    _, _, done, _ = e.step(a1)
    if done:
        e.reset()
print("single env  env.step (ZoneWrapper dict)                %8.1f us/step" % ((time.perf_counter() - t0) / 300 * 1e6))
