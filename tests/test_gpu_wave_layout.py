"""GPU parity of the wave-per-env layout (cfg.kernel = ZENV_KERNEL_WAVE_PER_ENV, K1w): one wave64 per env,
lane z owns zone z -- the layout BASELINE.json's north_star spells out.  Same oracle, same bar (everything
bit-identical), plus: both layouts leave byte-identical device state behind."""
import numpy as np
import pytest

from tests.helpers import OracleBatch, oracle_config_from

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task,zones,keepout,n", [(0, 25, 0.40, 101), (1, 25, 0.40, 66), (2, 6, 0.55, 130),
                                                 (2, 25, 0.40, 37), (0, 1, 0.55, 5), (1, 32, 0.30, 9)])
def test_wave_layout_lockstep(zenv_mod, oracle_mod, task, zones, keepout, n):
    Z, O = zenv_mod, oracle_mod
    E = Z._native
    cfg = Z.default_config(task, zones, zones_keepout=keepout, num_steps=120, kernel=E.KERNEL_WAVE_PER_ENV)
    if zones == 32:
        cfg.zones_size = 0.2
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(77, n)
    env.schedule_sequential()
    env.reset()
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(77, 77 + n))
    o_ref, zo_ref = ob.reset()
    o, zo = env.observations()
    assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref)
    rs = np.random.RandomState(3)
    n_done = 0
    for t in range(300):
        a = ob.policy(O.POLICY_GREEDY, o_ref, zo_ref, t)
        if t % 7 == 0:
            a = rs.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)      # also outside ctrlrange
        env.step(a, auto_reset=True)
        r_ref, d_ref, g_ref = ob.step(a)
        o_ref, zo_ref = ob.obs()
        o, zo, r, d, g = env.results()
        assert np.array_equal(d, d_ref) and np.array_equal(g, g_ref), f"flags differ at step {t}"
        assert np.array_equal(r, r_ref.astype(np.float32)), f"reward differs at step {t}"
        assert np.array_equal(o, o_ref), f"obs differs at step {t}"
        assert np.array_equal(zo, zo_ref), f"zone_obs differs at step {t}"
        n_done += int(d.sum())
    st = env.debug_state()
    q, v, steps = ob.state()
    assert np.array_equal(st["qpos"], q) and np.array_equal(st["qvel"], v) and np.array_equal(st["steps"], steps)
    assert n_done > 0
    env.close()


@pytest.mark.parametrize("task,zones", [(0, 15), (1, 25), (2, 6)])
def test_wave_layout_equals_lane_layout(zenv_mod, task, zones):
    """Random schedule, policy on the device, frozen envs in between: the two layouts agree on every field."""
    Z = zenv_mod
    E = Z._native
    n = 300
    envs = []
    for kernel in (E.KERNEL_LANE_PER_ENV, E.KERNEL_WAVE_PER_ENV):
        cfg = Z.default_config(task, zones, zones_keepout=0.45, num_steps=50, kernel=kernel)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(5, 64)
        env.schedule_fixed_seeds(np.arange(n, dtype=np.uint64) + 99, 5, 68)
        env.reset()
        envs.append(env)
    for phase, auto in ((0, True), (1, False), (2, True)):
        for env in envs:
            env.rollout(90, Z.POLICY_GREEDY if phase != 1 else Z.POLICY_UNIFORM, policy_seed=phase, auto_reset=auto)
        if phase == 1:
            assert envs[0].get(Z.F_DONE).all()          # everything froze without auto-reset
            for env in envs:
                env.reset(mask=np.ones(n, np.uint8))
        for f in (Z.F_OBS, Z.F_ZONE_OBS, Z.F_REWARD, Z.F_DONE, Z.F_GOAL_MET, Z.F_EPISODES, Z.F_LAST_RETURN,
                  Z.F_LAST_LEN, Z.F_VISIT_COUNT):      # (actions: the fused paths leave a_{t+1} behind, K3 a_t)
            assert np.array_equal(envs[0].get(f), envs[1].get(f), equal_nan=True), (phase, f)
        a, b = envs[0].debug_state(), envs[1].debug_state()
        for key in a:
            assert np.array_equal(a[key], b[key]), (phase, key)
    for env in envs:
        env.close()


def test_wave_layout_goal_conditioned(zenv_mod):
    """The goal-conditioned post-step kernel reads visit_zone / term_xy that K1w leaves behind."""
    Z = zenv_mod
    E = Z._native
    n = 64
    out = []
    for kernel in (E.KERNEL_LANE_PER_ENV, E.KERNEL_WAVE_PER_ENV):
        cfg = Z.default_config(0, 15, num_steps=200, kernel=kernel)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(3, n)
        env.schedule_sequential()
        env.enable_goals()
        env.reset()
        rs = np.random.RandomState(0)
        hist = []
        for t in range(260):
            _, need, avail, _ = env.goal_info()
            if need.any():
                g = np.full(n, -1, np.int32)
                for i in np.nonzero(need)[0]:
                    opts = [z for z in range(15) if (int(avail[i]) >> z) & 1]
                    g[i] = opts[rs.randint(len(opts))] if opts else -1
                env.set_goals(g)
            env.policy(Z.POLICY_GREEDY)
            env.step(None, auto_reset=True)
            hist.append((env.get(Z.F_SHAPED_REWARD).copy(), env.get(Z.F_DONE).copy(), env.get(Z.F_GOAL).copy()))
        out.append(hist)
        env.close()
    for (s0, d0, g0), (s1, d1, g1) in zip(*out):
        assert np.array_equal(s0, s1) and np.array_equal(d0, d1) and np.array_equal(g0, g1)


@pytest.mark.parametrize("task,zones,seed", [(0, 25, 0), (1, 15, 1), (2, 6, 2), (1, 25, 3), (2, 25, 4)])
def test_random_api_sequences_agree_between_layouts(zenv_mod, task, zones, seed):
    """Differential test: K1w shares no step / reset / rollout code with K1 + K1p.  Random sequences of API calls
    (rollouts in every launch mode, host actions, masked resets, frozen envs, snapshots) must leave the same
    results behind in both layouts after every call."""
    Z = zenv_mod
    E = Z._native
    n = 200
    rs = np.random.RandomState(100 + seed)
    envs = []
    for kernel in (E.KERNEL_LANE_PER_ENV, E.KERNEL_WAVE_PER_ENV):
        cfg = Z.default_config(task, zones, zones_keepout=0.40 if zones == 25 else 0.55, num_steps=40, kernel=kernel)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(11, 96)
        env.schedule_fixed_seeds(np.arange(n, dtype=np.uint64) * 7 + seed, 11, 106)
        env.reset()
        envs.append(env)
    fields = (Z.F_OBS, Z.F_ZONE_OBS, Z.F_REWARD, Z.F_DONE, Z.F_GOAL_MET, Z.F_EPISODES, Z.F_LAST_RETURN, Z.F_LAST_LEN,
              Z.F_VISIT_COUNT, Z.F_EP_RETURN, Z.F_EP_LEN, Z.F_SEED)
    snap = None
    for it in range(40):
        op = rs.randint(6)
        if op == 0:
            k, pol, auto = int(rs.randint(1, 70)), int(rs.randint(2)), bool(rs.randint(2))
            mode = ("persistent", "per_step", "unfused")[rs.randint(3)]
            for env in envs:
                env.rollout(k, pol, policy_seed=it, env_index0=3, auto_reset=auto, mode=mode)
        elif op == 1:
            auto = bool(rs.randint(2))
            for _ in range(int(rs.randint(1, 8))):
                a = rs.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)
                for env in envs:
                    env.step(a, auto_reset=auto)
        elif op == 2:
            m = (rs.uniform(size=n) < 0.3).astype(np.uint8)
            for env in envs:
                env.reset(mask=m)
        elif op == 3:
            for env in envs:
                env.policy(Z.POLICY_GREEDY)
                env.step(None, auto_reset=True)
        elif op == 4:
            snap = [env.get_state() for env in envs]
        elif op == 5 and snap is not None:
            for env, s in zip(envs, snap):
                env.set_state(s)
        for f in fields:
            assert np.array_equal(envs[0].get(f), envs[1].get(f), equal_nan=True), (it, op, f)
        a, b = envs[0].debug_state(), envs[1].debug_state()
        for key in a:
            assert np.array_equal(a[key], b[key]), (it, op, key)
    for env in envs:
        env.close()
