import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import combinatorial_rl_tasks_amd as Z
from combinatorial_rl_tasks_amd import sharding
task, zones, keep = bench.WORKLOADS["ColourMatch-6"]
n = 65536
for style in ("replay", "shard", "replay_depth4"):
    cfg = Z.default_config(task, zones, zones_keepout=keep)
    env = Z.ZoneVecEnv(cfg, n)
    if style == "replay":
        bench.replay_bank(env, n, 65536)
    elif style == "shard":
        sharding.EnvShard(0, 1, n).build_bank(env, 4, n_threads=16)
    else:
        env.build_bank(1, 4 * n, n_threads=16); env.schedule_sequential(first=np.arange(n, dtype=np.int32), stride=n)
    env.reset()
    for rep in range(3):
        env.rollout(3000, Z.POLICY_GREEDY)
        ms, k = env.rollout(2048, Z.POLICY_GREEDY, time_step_kernel=True)
        print(style, "us/step %.3f" % (k * 1e3), "episodes", int(env.get(Z.F_EPISODES).sum()), "mean len", float(env.get(Z.F_LAST_LEN).mean()), flush=True)
    env.close()
