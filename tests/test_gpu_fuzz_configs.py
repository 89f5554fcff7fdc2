"""Randomised configurations: every launch mode and both kernel layouts of the step path against the oracle's batch driver on configurations no
other test names -- zone counts 1..30 (compiled-in and runtime-Z kernels), short and long episodes, cooldowns, zone radii,
reward constants, frameskips other than 10 (the looped substep path), robot constants within validate_config's bounds,
both scripted policies.  Bit-exact, as everywhere (obs, zone_obs, episode counters, returns, lengths)."""
import os

import numpy as np
import pytest

from tests.helpers import oracle_config_from

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("ZENV_FUZZ_CASES", "24"))        # a longer hunt: ZENV_FUZZ_CASES=400 pytest ... -k random


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


@pytest.mark.parametrize("case", range(N_CASES))
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


@pytest.mark.parametrize("case", range(8))
def test_lockstep_with_adversarial_actions_and_reset_patterns(zenv_mod, oracle_mod, case):
    """External actions, one launch per step, compared after EVERY step: greedy actions mixed with out-of-range values,
    exact +-1 / 0, +-inf (clipped by Engine.step) and the occasional NaN (Engine.step's exception branch); auto-reset
    switched on and off from step to step (ParallelEnv.step / step_no_reset: finished envs become masked no-ops) and
    masked resets of random subsets in between (zenv_reset with a mask)."""
    from tests.helpers import OracleBatch
    Z, O = zenv_mod, oracle_mod
    rs = np.random.RandomState(500 + case)
    task = case % 3
    zones = [15, 25, 6, 5, 10, 9, 1, 20][case]
    if task == 2:
        zones = min(zones, 12)
    cfg = Z.default_config(task, zones, zones_keepout=0.3 if zones > 15 else 0.5, num_steps=int(rs.choice([60, 120, 200])),
                           max_cd=int(rs.choice([150, 5])))
    n = int(rs.choice([37, 64, 101]))
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(300, n)
    env.schedule_sequential()
    env.reset()
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(300, 300 + n))
    o_ref, zo_ref = ob.reset()
    n_exc = n_masked = 0
    for t in range(260):
        a = ob.policy(O.POLICY_GREEDY, o_ref, zo_ref, t)
        kind = rs.rand(n)
        wild = rs.uniform(-3, 3, (n, 2)).astype(np.float32)
        exact = rs.choice(np.array([-1.0, 0.0, 1.0], np.float32), (n, 2))
        a[kind < 0.15] = wild[kind < 0.15]
        a[(kind >= 0.15) & (kind < 0.25)] = exact[(kind >= 0.15) & (kind < 0.25)]
        a[(kind >= 0.25) & (kind < 0.27), 0] = np.inf
        a[(kind >= 0.27) & (kind < 0.29), 1] = -np.inf
        nan = kind > 0.997
        a[nan, int(rs.randint(0, 2))] = np.nan
        auto = bool(rs.rand() < 0.75)
        live = np.array([not e.e.done for e in ob.envs])
        n_exc += int((nan & live).sum())
        n_masked += int((~live).sum())
        got = env.step_results(a, auto_reset=auto)
        r_ref, d_ref, g_ref = ob.step(a, auto_reset=auto)
        o_ref, zo_ref = ob.obs()
        if not auto:                           # an env that was already finished: WaitWrapper's noop_obs (zeros), reward 0;
            o_ref[~live] = 0                   # under auto-reset the worker resets it and returns the new first obs
            zo_ref[~live] = 0
        assert np.array_equal(got[3], d_ref) and np.array_equal(got[4], g_ref), (case, t)
        assert np.array_equal(got[2], r_ref.astype(np.float32)), (case, t)
        assert np.array_equal(got[0], o_ref) and np.array_equal(got[1], zo_ref), (case, t)
        assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all() and np.isfinite(got[2]).all()
        if t % 37 == 36:                       # masked reset of a random subset, finished or not
            m = rs.rand(n) < 0.2
            env.reset(mask=m)
            for i in np.flatnonzero(m):
                ob.envs[i].reset(ob.seeds[i])
            o_new, zo_new = ob.obs()
            o_ref[m], zo_ref[m] = o_new[m], zo_new[m]          # the other envs' buffers stay as the last step left them
            first = env.step_results(None)
            assert np.array_equal(first[0], o_ref) and np.array_equal(first[1], zo_ref), (case, t, "after masked reset")
    q, v, steps = ob.state()
    dbg = env.debug_state()
    assert np.array_equal(dbg["qpos"], q) and np.array_equal(dbg["qvel"], v) and np.array_equal(dbg["steps"], steps)
    assert n_masked > 0                         # frozen envs were stepped over
    env.close()
