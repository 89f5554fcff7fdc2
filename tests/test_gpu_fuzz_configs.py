"""Randomised configurations: every launch mode and both kernel layouts of the step path against the oracle's batch driver on configurations no
other test names -- zone counts 1..30 (compiled-in and runtime-Z kernels), short and long episodes, cooldowns, zone radii,
reward constants, frameskips other than 10 (the looped substep path), robot constants within validate_config's bounds,
both scripted policies.  Bit-exact, as everywhere (obs, zone_obs, episode counters, returns, lengths)."""
import numpy as np
import pytest

from tests.helpers import oracle_config_from

pytestmark = pytest.mark.gpu


def _draw(rs, Z, case):
    task = int(rs.randint(0, 3))
    zones = int(rs.choice([1, 2, 3, 5, 6, 7, 10, 12, 15, 20, 25, 28, 30]))
    if task == 2:
        zones = min(zones, 15)
    keepout = 0.55 if zones <= 12 else 0.4 if zones <= 20 else 0.3
    over = dict(zones_keepout=keepout,
                num_steps=int(rs.choice([40, 97, 250, 600])),
                zones_size=float(rs.choice([0.2, 0.15, 0.3, 0.45])),
                time_saved_reward=float(rs.choice([0.01, 0.0, 0.5])),
                frameskip=int(rs.choice([10, 10, 10, 4, 1, 13])))
    if task == 2:
        over["max_cd"] = int(rs.choice([150, 1, 7, 40]))
    if task == 1:
        over["beta_a"], over["beta_b"] = [(3.0, 1.5), (1.2, 1.1), (5.0, 2.0)][int(rs.randint(0, 3))]
    if rs.rand() < 0.3:                       # another robot: heavier, other gear / damping (finite, well inside the bounds)
        over.update(mass=0.0052 * float(rs.uniform(0.8, 3.0)), gear=float(rs.uniform(0.2, 0.4)),
                    vel_kv=float(rs.uniform(0.7, 1.2)))
    cfg = Z.default_config(task, zones, **over)
    if rs.rand() < 0.3:
        cfg.damping[0] = cfg.damping[1] = 0.01 * float(rs.uniform(0.5, 2.0))    # iso-damping keeps the constant Schur path
    elif rs.rand() < 0.3:
        cfg.damping[1] = cfg.damping[0] * 1.5                                    # the divide path
    return cfg


@pytest.mark.parametrize("case", range(24))
def test_random_configuration_all_modes(zenv_mod, oracle_mod, case):
    Z, O = zenv_mod, oracle_mod
    rs = np.random.RandomState(9000 + case)
    cfg = _draw(rs, Z, case)
    n = int(rs.choice([1, 63, 64, 65, 200, 333]))
    T = int(rs.choice([60, 150, 400]))
    depth = 5
    policy = (Z.POLICY_GREEDY, O.POLICY_GREEDY) if rs.rand() < 0.6 else (Z.POLICY_UNIFORM, O.POLICY_UNIFORM)
    ref = O.rollout(oracle_config_from(O, cfg), 7 + np.arange(n), T, policy[1], seed_stride=n, policy_seed=31 + case,
                    env_index0=3 * case, n_threads=8, seed_period=depth)
    assert ref["episodes"].sum() > 0 or T < cfg.num_steps
    for mode in ("persistent", "per_step", "unfused", "wave_per_env"):
        cfg.kernel = Z._native.KERNEL_WAVE_PER_ENV if mode == "wave_per_env" else Z._native.KERNEL_LANE_PER_ENV
        if mode == "wave_per_env":
            if n > 200:
                continue                   # (one wave per env: keep the slow layout's share of the suite small)
            mode = "unfused"               # K1w has no fused action source
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(7, depth * n)
        env.schedule_sequential(stride=n)
        env.reset()
        env.rollout(T, policy[0], policy_seed=31 + case, env_index0=3 * case, mode=mode)
        what = (case, mode, cfg.task, cfg.num_zones, cfg.frameskip, n, T)
        for f, name in ((Z.F_OBS, "obs"), (Z.F_ZONE_OBS, "zone_obs"), (Z.F_EPISODES, "episodes"),
                        (Z.F_LAST_RETURN, "last_return"), (Z.F_LAST_LEN, "last_len")):
            assert np.array_equal(env.get(f), ref[name]), (name,) + what
        env.close()
