"""PMC target: the action-chunk form of the persistent kernel (k_rollout_lane<.., EXT>) on PointTSP-25, N = 65 536:
1 024 steps of fresh host-drawn actions (512 MiB, uploaded once) replayed as four 256-step launches, twice.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (scripts/profile_round.sh); summarize_profile.py divides
each dispatch by its 256 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import combinatorial_rl_tasks_amd as Z
from combinatorial_rl_tasks_amd import _native as nat
n, K = 65536, 1024
cfg = Z.default_config(0, 25, zones_keepout=0.4)
env = Z.ZoneVecEnv(cfg, n)
env.build_bank(1, 4 * n, n_threads=16)
env.schedule_sequential(stride=n)
env.reset()
env.rollout(2048, Z.POLICY_GREEDY)
a = np.random.RandomState(0).uniform(-1, 1, (K, n, 2)).astype(np.float32)
env.step_many(a, reset="every")
ptr = (env.device_ptr(nat.F_CHUNK_ACTIONS), K)
env.step_many(None, reset="every", actions_ptr=ptr)
env.sync()
print("episodes", int(env.get(Z.F_EPISODES).sum()))
