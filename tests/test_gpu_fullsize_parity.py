"""Full-size ORACLE parity: BASELINE.json configs[1..3] at their real size (N = 65 536 envs on one MI355X), every env
compared with the CPU oracle's batch driver -- final obs, zone_obs, episode counts, last return and last length of
ALL envs, bit for bit -- for the persistent rollout kernel (K1p) and for one launch per step (K1).

Seed protocol as in bench.py / main/scripts/evaluate.py:47-72 restated for a batch: env i plays map seeds 1+i,
1+i+N, ... from a bank of `DEPTH` episodes per env that wraps (the oracle's driver wraps the same way), closed-loop
pi_greedy on the device, auto-reset on (penv.py:8-11).  Steps are chosen so that every config sees thousands of
episode ends (time limit shortened for PointTSP, whose greedy episodes otherwise last ~1000 steps); the oracle
does ~3 M env-steps/s per host core, so each case costs a few seconds of CPU.
"""
import os

import numpy as np
import pytest

from tests.helpers import oracle_config_from

pytestmark = pytest.mark.gpu
N = 65536
DEPTH = 3
_REF = {}


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


# (task, zones, zones_keepout, num_steps, T): BASELINE configs[1], [2], [3] (+ the reference-faithful 15-zone map)
CASES = [
    pytest.param(0, 25, 0.40, 2000, 300, id="PointTSP-25-num_steps2000"),
    pytest.param(0, 25, 0.40, 250, 600, id="PointTSP-25-num_steps250-many-resets"),
    pytest.param(1, 25, 0.40, 2000, 400, id="TimedTSP-25"),
    pytest.param(2, 6, 0.55, 2000, 500, id="ColourMatch-6"),
    pytest.param(0, 15, 0.55, 300, 400, id="PointTSP-15-reference-keepout"),
]


@pytest.mark.parametrize("task,zones,keepout,num_steps,T", CASES)
@pytest.mark.parametrize("mode", ["persistent", "per_step"])
def test_every_env_of_the_full_batch_matches_the_oracle(zenv_mod, oracle_mod, task, zones, keepout, num_steps, T, mode):
    Z, O = zenv_mod, oracle_mod
    cfg = Z.default_config(task, zones, zones_keepout=keepout, num_steps=num_steps)
    env = Z.ZoneVecEnv(cfg, N)
    env.build_bank(1, DEPTH * N, n_threads=_threads())
    env.schedule_sequential(stride=N)
    env.reset()
    env.rollout(T, Z.POLICY_GREEDY, policy_seed=0x5EED, env_index0=0, mode=mode)
    key = (task, zones, keepout, num_steps, T)
    if key not in _REF:        # the two launch modes of a case share one oracle run (<= 50 MB each)
        _REF[key] = O.rollout(oracle_config_from(O, cfg), 1 + np.arange(N), T, O.POLICY_GREEDY, seed_stride=N,
                              policy_seed=0x5EED, env_index0=0, n_threads=_threads(), seed_period=DEPTH)
    ref = _REF[key]
    got = {"obs": env.get(Z.F_OBS), "zone_obs": env.get(Z.F_ZONE_OBS), "episodes": env.get(Z.F_EPISODES),
           "last_return": env.get(Z.F_LAST_RETURN), "last_len": env.get(Z.F_LAST_LEN)}
    env.close()
    assert int(ref["total_steps"]) == N * T
    for name, a in got.items():
        b = ref[name]
        same = (a == b) | ((a != a) & (b != b))
        assert same.all(), (f"{name}: {int((~same).sum())} of {same.size} values differ; first env "
                            f"{int(np.argwhere(~same.reshape(N, -1).all(1))[0, 0])}")
    # the case must have exercised what it is there for: episode ends inside the batch
    if num_steps <= 300 or task != 0:
        assert int(ref["episodes"].sum()) > N // 8, "too few episode ends to test the auto-reset path at scale"
