"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): visit counts / termination step indices bit-identical,
rewards within 1e-5.  What is asserted here is stricter: float32 observations, rewards,
done/goal flags and the float64 joint state are all bit-identical to the oracle.
"""
import numpy as np
import pytest

from tests.helpers import TASK_IDS, OracleBatch, oracle_config_from

pytestmark = pytest.mark.gpu

EVAL_SEED0 = 1000000   # main/scripts/evaluate.py:47


def _make(Z, env_id, n, seed0, **overrides):
    cfg = Z.config_for_id(env_id, **overrides)
    env = Z.ZoneVecEnv(cfg, n, device=0)
    env.build_bank(seed0, n)
    env.schedule_sequential()      # env i replays bank slot i (evaluate.py: 5 runs per map)
    return cfg, env


@pytest.mark.parametrize("task", [0, 1, 2])
def test_lockstep_eval_seeds(zenv_mod, oracle_mod, task):
    """100 eval seeds, greedy actions computed by the oracle, stepped in lock step."""
    Z, O = zenv_mod, oracle_mod
    n = 100
    T = 2100 if task == 0 else 700
    cfg, env = _make(Z, TASK_IDS[task], n, EVAL_SEED0)
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(EVAL_SEED0, EVAL_SEED0 + n))
    env.reset()
    o_ref, zo_ref = ob.reset()
    o, zo = env.observations()
    assert np.array_equal(o, o_ref)
    assert np.array_equal(zo, zo_ref)
    n_done = n_goal = 0
    for t in range(T):
        a = ob.policy(O.POLICY_GREEDY, o_ref, zo_ref, t)
        env.step(a, auto_reset=True)
        r_ref, d_ref, g_ref = ob.step(a)
        o_ref, zo_ref = ob.obs()
        o, zo, r, d, g = env.results()
        assert np.array_equal(d, d_ref), f"done mismatch at step {t}"
        assert np.array_equal(g, g_ref), f"goal_met mismatch at step {t}"
        assert np.array_equal(r, r_ref.astype(np.float32)), f"reward mismatch at step {t}"
        assert np.max(np.abs(r.astype(np.float64) - r_ref)) <= 1e-5
        assert np.array_equal(o, o_ref), f"obs mismatch at step {t}"
        assert np.array_equal(zo, zo_ref), f"zone_obs mismatch at step {t}"
        n_done += int(d.sum())
        n_goal += int(g.sum())
        if t % 97 == 0:
            st = env.debug_state()
            q, v, steps = ob.state()
            assert np.array_equal(st["qpos"], q) and np.array_equal(st["qvel"], v)
            assert np.array_equal(st["steps"], steps)
    assert n_done > 0, "the scripted policy never finished an episode: test is too weak"
    env.close()


@pytest.mark.parametrize("task", [0, 1, 2])
def test_policy_kernels_match_oracle(zenv_mod, oracle_mod, task):
    Z, O = zenv_mod, oracle_mod
    n = 257   # ragged: not a multiple of the 64-env tile
    cfg, env = _make(Z, TASK_IDS[task], n, 5000)
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(5000, 5000 + n))
    env.reset()
    o_ref, zo_ref = ob.reset()
    for t in range(120):
        pol = O.POLICY_GREEDY if t % 3 else O.POLICY_UNIFORM
        env.policy(pol, policy_seed=0x5EED, env_index0=7)
        a_dev = env.get(Z.F_ACTIONS)
        a_ref = ob.policy(pol, o_ref, zo_ref, t, env_index0=7)
        assert np.array_equal(a_dev, a_ref), f"policy {pol} mismatch at step {t}"
        env.step(None, auto_reset=True)
        ob.step(a_ref)
        o_ref, zo_ref = ob.obs()
        o, zo = env.observations()
        assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref)
    env.close()


@pytest.mark.parametrize("mode", ["persistent", "per_step", "unfused"])
@pytest.mark.parametrize("task,zones,keepout", [(0, 25, 0.40), (1, 25, 0.40), (2, 6, 0.55), (0, 5, 0.55),
                                                (1, 7, 0.55), (0, 15, 0.55), (2, 25, 0.40), (1, 6, 0.55),
                                                (2, 10, 0.55), (1, 20, 0.45)])
def test_closed_loop_rollout_matches_oracle(zenv_mod, oracle_mod, task, zones, keepout, mode):
    """Device-resident closed loop (persistent rollout kernel / step kernel with fused action
    source / policy kernel + step kernel; auto-reset onto fresh seeds) against the oracle's
    batch driver: BASELINE.json configs 1-3 at reduced N.  (Z = 7 has no persistent kernel.)"""
    Z, O = zenv_mod, oracle_mod
    n, T, stride = 1000, 400, 1000
    cfg = Z.default_config(task, zones, zones_keepout=keepout)
    if (task, zones) != (2, 6):
        cfg.num_steps = 400 if task != 2 else 150      # force time-limit resets inside the window
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, 16 * stride)
    env.schedule_sequential(stride=stride)
    env.reset()
    env.rollout(T, Z.POLICY_GREEDY, policy_seed=11, env_index0=0, auto_reset=True, mode=mode)
    ocfg = oracle_config_from(O, cfg)
    ref = O.rollout(ocfg, np.arange(1, 1 + n), T, O.POLICY_GREEDY, seed_stride=stride,
                    policy_seed=11, n_threads=8)
    assert np.array_equal(env.get(Z.F_EPISODES), ref["episodes"])
    assert np.array_equal(env.get(Z.F_LAST_LEN), ref["last_len"])
    assert np.array_equal(env.get(Z.F_LAST_RETURN), ref["last_return"])
    assert np.array_equal(env.get(Z.F_OBS), ref["obs"])
    assert np.array_equal(env.get(Z.F_ZONE_OBS), ref["zone_obs"])
    assert ref["episodes"].max() < 16 and ref["episodes"].sum() > 0
    assert env.step_count == T
    env.close()


@pytest.mark.parametrize("task,zones", [(0, 15), (1, 25), (2, 6)])
def test_persistent_rollout_interleaves_with_steps(zenv_mod, oracle_mod, task, zones):
    """A persistent launch starts from and leaves behind exactly the state the step API sees:
    rollout(70) ; step(a) x 5 ; rollout(3) ; snapshot ; rollout(130) == restore ; per-step x 130,
    all in lock-step with the oracle."""
    Z, O = zenv_mod, oracle_mod
    n = 193                                            # ragged last tile
    cfg = Z.default_config(task, zones, zones_keepout=0.45, num_steps=60)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(9, n)
    env.schedule_sequential()                          # every reset replays the env's own map
    env.reset()
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(9, 9 + n))
    o_ref, zo_ref = ob.reset()
    t = 0

    def oracle_greedy(k):
        nonlocal o_ref, zo_ref, t
        for _ in range(k):
            ob.step(ob.policy(O.POLICY_GREEDY, o_ref, zo_ref, t))
            o_ref, zo_ref = ob.obs()
            t += 1

    env.rollout(70, Z.POLICY_GREEDY)
    oracle_greedy(70)
    o, zo = env.observations()
    assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref)
    rs = np.random.RandomState(4)
    for _ in range(5):
        a = rs.uniform(-1, 1, (n, 2)).astype(np.float32)
        env.step(a, auto_reset=True)
        r_ref, d_ref, g_ref = ob.step(a)
        o_ref, zo_ref = ob.obs()
        t += 1
        o, zo, r, d, g = env.results()
        assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref)
        assert np.array_equal(r, r_ref.astype(np.float32)) and np.array_equal(d, d_ref)
    env.rollout(3, Z.POLICY_GREEDY)
    oracle_greedy(3)
    blob = env.get_state()
    env.rollout(130, Z.POLICY_GREEDY)
    oracle_greedy(130)
    res_a = env.results()
    assert np.array_equal(res_a[0], o_ref) and np.array_equal(res_a[1], zo_ref)
    epi_a = env.get(Z.F_EPISODES)
    env.set_state(blob)
    env.rollout(130, Z.POLICY_GREEDY, mode="per_step")
    for x, y in zip(res_a, env.results()):
        assert np.array_equal(x, y)
    assert np.array_equal(epi_a, env.get(Z.F_EPISODES)) and epi_a.sum() > 0
    env.close()


@pytest.mark.parametrize("mode", ["persistent", "per_step"])
def test_rollout_wraps_around_the_map_bank(zenv_mod, oracle_mod, mode):
    """More episodes than the bank holds per env: the schedule replays the env's maps in order."""
    Z, O = zenv_mod, oracle_mod
    n, depth, T = 150, 3, 700
    cfg = Z.default_config(1, 15, zones_keepout=0.55, num_steps=60)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, depth * n)
    env.schedule_sequential(stride=n)
    env.reset()
    env.rollout(T, Z.POLICY_GREEDY, mode=mode)
    ref = O.rollout(oracle_config_from(O, cfg), 1 + np.arange(n), T, O.POLICY_GREEDY, seed_stride=n,
                    seed_period=depth, n_threads=8)
    assert ref["episodes"].min() > 2 * depth
    assert np.array_equal(env.get(Z.F_EPISODES), ref["episodes"])
    assert np.array_equal(env.get(Z.F_OBS), ref["obs"])
    assert np.array_equal(env.get(Z.F_ZONE_OBS), ref["zone_obs"])
    assert np.array_equal(env.get(Z.F_LAST_RETURN), ref["last_return"])
    env.close()


def test_persistent_rollout_without_auto_reset(zenv_mod):
    """auto_reset = 0 inside a persistent launch: a finished env freezes (WaitWrapper masking),
    its joint state stays where the episode ended -- same as per-step launches."""
    Z = zenv_mod
    n = 130
    cfg = Z.default_config(1, 25, zones_keepout=0.40, num_steps=90)
    outs = []
    for mode in ("persistent", "per_step"):
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(3, n)
        env.reset()
        env.rollout(20, Z.POLICY_GREEDY, auto_reset=False, mode=mode)
        mid = env.get(Z.F_DONE).copy()
        env.rollout(130, Z.POLICY_GREEDY, auto_reset=False, mode=mode)
        outs.append((mid, env.results(), env.get(Z.F_LAST_LEN), env.get(Z.F_LAST_RETURN), env.get_state()))
        env.close()
    a, b = outs
    assert 0 < a[0].sum() < n                      # some envs timed out early (TTSP_env.py:67), not all
    assert a[1][3].all() and not a[1][0].any()     # everyone is done after 150 >= num_steps; zero obs
    assert np.array_equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert np.array_equal(a[4], b[4])              # the whole state blob, joint state included


def test_persistent_rollout_fixed_seed_schedule(zenv_mod):
    """FixedSeedsWrapper draws (device PCG64) made from inside the persistent kernel."""
    Z = zenv_mod
    n, lo, hi = 100, 1, 100
    cfg = Z.config_for_id("PointTSP-v1", num_steps=17)
    outs = []
    for mode in ("persistent", "unfused"):
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(lo, hi - lo + 1)
        env.schedule_fixed_seeds(np.arange(n, dtype=np.uint64) * 10000, lo, hi)
        env.reset()
        env.rollout(200, Z.POLICY_UNIFORM, policy_seed=5, mode=mode)
        outs.append((env.get(Z.F_SEED), env.get(Z.F_EPISODES), env.results()))
        env.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1].min() == 11
    assert len(np.unique(outs[0][0])) > 30
    for x, y in zip(outs[0][2], outs[1][2]):
        assert np.array_equal(x, y)


def test_uniform_policy_rollout(zenv_mod, oracle_mod):
    Z, O = zenv_mod, oracle_mod
    n, T = 300, 250
    cfg = Z.config_for_id("PointTSP-v0")
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(EVAL_SEED0, n)
    env.schedule_sequential()
    env.reset()
    env.rollout(T // 2, Z.POLICY_UNIFORM, policy_seed=0x5EED, env_index0=123, auto_reset=True, mode="persistent")
    env.rollout(T - T // 2, Z.POLICY_UNIFORM, policy_seed=0x5EED, env_index0=123, auto_reset=True, fused=False)
    ref = O.rollout(oracle_config_from(O, cfg), np.arange(EVAL_SEED0, EVAL_SEED0 + n), T,
                    O.POLICY_UNIFORM, policy_seed=0x5EED, env_index0=123, n_threads=8)
    assert np.array_equal(env.get(Z.F_OBS), ref["obs"])
    assert np.array_equal(env.get(Z.F_ZONE_OBS), ref["zone_obs"])
    assert np.array_equal(env.get(Z.F_EP_RETURN), ref["reward_sum"])
    env.close()


def test_no_auto_reset_masks_finished_envs(zenv_mod, oracle_mod):
    """step_no_reset (penv.py:61-66) + WaitWrapper semantics (wrappers.py:34-45)."""
    Z, O = zenv_mod, oracle_mod
    n = 64
    cfg = Z.config_for_id("PointTSP-v1", num_steps=30)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, n)
    env.reset()
    a = np.zeros((n, 2), np.float32)
    for t in range(29):
        env.step(a, auto_reset=False)
        assert not env.get(Z.F_DONE).any()
    env.step(a, auto_reset=False)
    assert env.get(Z.F_DONE).all()
    assert np.array_equal(env.get(Z.F_LAST_LEN), np.full(n, 30))
    env.step(a, auto_reset=False)       # finished: zero obs, zero reward, done stays set
    o, zo, r, d, g = env.results()
    assert d.all() and not g.any() and not r.any() and not o.any() and not zo.any()
    mask = np.zeros(n, np.uint8)
    mask[::2] = 1
    env.reset(mask)
    env.step(a, auto_reset=False)
    d = env.get(Z.F_DONE).astype(bool)
    assert not d[::2].any() and d[1::2].all()
    env.close()


def test_state_snapshot_roundtrip(zenv_mod):
    Z = zenv_mod
    cfg = Z.config_for_id("ColourMatch-v0")
    env = Z.ZoneVecEnv(cfg, 130)
    env.build_bank(1, 130)
    env.reset()
    env.rollout(40, Z.POLICY_GREEDY)
    blob = env.get_state()
    env.rollout(25, Z.POLICY_GREEDY)
    want = env.observations()
    env.set_state(blob)
    env.rollout(25, Z.POLICY_GREEDY)
    got = env.observations()
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
    env.close()


def test_fixed_seeds_schedule_on_device(zenv_mod):
    """FixedSeedsWrapper (wrappers.py:10-23) seed draws done by the device PCG64."""
    Z = zenv_mod
    n, lo, hi = 96, 1, 100
    cfg = Z.config_for_id("PointTSP-v1", num_steps=5)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(lo, hi - lo + 1)
    rng_seeds = np.arange(n, dtype=np.uint64) * 10000      # train_ppo.py:112 rng_seed=seed+10000*i
    env.schedule_fixed_seeds(rng_seeds, lo, hi)
    env.reset()
    a = np.zeros((n, 2), np.float32)
    seen = [env.get(Z.F_SEED).copy()]
    for ep in range(4):
        for _ in range(5):
            env.step(a, auto_reset=True)
        seen.append(env.get(Z.F_SEED).copy())
    seen = np.stack(seen, 1)
    for i in range(n):
        g = np.random.default_rng(int(rng_seeds[i]))
        want = [int(g.integers(low=lo, high=hi + 1, size=1)[0]) for _ in range(5)]
        assert seen[i].tolist() == want
    env.close()


@pytest.mark.parametrize("task", [0, 1, 2])
def test_rim_zones_take_the_exact_path(zenv_mod, oracle_mod, task):
    """Zones placed within +-3e-6 of the 0.2 rim (inside the float32 prefilter's ambiguous shell,
    and just outside it on both sides): the device verdicts must be the oracle's float64
    sqrt(dx^2+dy^2) <= 0.2 verdicts, zone by zone, step by step."""
    Z, O = zenv_mod, oracle_mod
    n_zone, n = 15, 70
    cfg = Z.default_config(task, n_zone)
    ocfg = oracle_config_from(O, cfg)
    rng = np.random.RandomState(123)
    robots, zones, auxs, refs = [], [], [], []
    offsets = np.array([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 3e-8, -3e-8, 2e-7, -2e-7, 9e-7, -9e-7,
                        1.9e-6, -1.9e-6, 3e-6, -3e-6])
    for i in range(n):
        ref = O.OracleEnv(ocfg)
        ref.reset(500 + i)                      # gives rot / colours / tmax of a real reset
        rx, ry = rng.uniform(-2.5, 2.5, 2)
        ang = rng.uniform(0, 2 * np.pi, n_zone)
        zxy = np.stack([rx + (0.2 + offsets) * np.cos(ang), ry + (0.2 + offsets) * np.sin(ang)], 1)
        ref.e.x0, ref.e.y0 = rx, ry
        ref.e.xpos[0], ref.e.xpos[1] = rx, ry
        for z in range(n_zone):
            ref.e.zone_xy[z][0], ref.e.zone_xy[z][1] = zxy[z]
        robots.append([rx, ry, ref.e.rot]); zones.append(zxy)
        auxs.append(ref.state()["tmax"] if task == 1 else ref.state()["colour"])
        refs.append(ref)
    env = Z.ZoneVecEnv(cfg, n)
    env.set_bank(np.array(robots), np.array(zones), np.array(auxs, np.int32), np.arange(n))
    env.schedule_sequential()
    env.reset()
    a = np.zeros((n, 2), np.float32)
    seen_inside = 0
    for t in range(n_zone + 2):
        env.step(a, auto_reset=False)
        st = env.debug_state()
        r = env.get(Z.F_REWARD)
        for i, ref in enumerate(refs):
            if ref.e.done:
                continue
            rr, d, g = ref.step(a[i])
            key = "colour" if task == 2 else "visited"
            assert np.array_equal(st["zone_state"][i], ref.state()[key]), (i, t)
            assert r[i] == np.float32(rr)
            seen_inside += int(rr != 0)
        o, zo = env.observations()
        for i in (0, n // 2, n - 1):
            if not refs[i].e.done:
                o_ref, zo_ref = refs[i].obs()
                assert np.array_equal(zo[i], zo_ref) and np.array_equal(o[i], o_ref)
    assert seen_inside > 4 * n        # about half of the rim zones are inside
    env.close()


@pytest.mark.parametrize("env_id", ["PointTSP-v0", "PointTTSP-v0", "ColourMatch-v0"])
def test_goal_conditioned_variant_lockstep(zenv_mod, oracle_mod, env_id):
    """TSPNextCityEnv / TimedTSPNextCityEnv (TSP_next_city_env.py:41-109): goal zone per
    env, shaped_reward, need_next_goal, available goals -- in lock step with the oracle, with auto-reset and
    a high-level 'policy' that picks a random available zone whenever one is needed."""
    Z, O = zenv_mod, oracle_mod
    n, T = 150, 260
    cfg = Z.config_for_id(env_id, num_steps=120)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(21, n)
    env.schedule_sequential()
    env.enable_goals()
    env.reset()
    refs = [O.OracleEnv(oracle_config_from(O, cfg)) for _ in range(n)]
    for i, e in enumerate(refs):
        e.reset(21 + i)
    rs = np.random.RandomState(8)
    sh0, need, avail, goal = env.goal_info()
    nz = cfg.num_zones
    assert need.all() and (goal == -1).all() and (avail == (1 << nz) - 1).all()
    n_reached = n_resets = 0
    for t in range(T):
        # high level: a goal for every env that needs one (penv.py:76-99)
        goals = np.full(n, -1, np.int32)
        for i in np.nonzero(need)[0]:
            options = np.nonzero(refs[i].available_goals())[0]
            assert np.array_equal(options, np.nonzero([(avail[i] >> z) & 1 for z in range(nz)])[0])
            goals[i] = rs.choice(options)
            refs[i].set_goal(goals[i])
        env.set_goals(goals)
        # low level: steer towards the goal zone, with noise
        o, zo = env.observations()
        g_now = env.get(Z.F_GOAL)
        assert (g_now >= 0).all()
        tgt = zo[np.arange(n), g_now, :2] * 3.0
        d = tgt - o[:, 1:3] * 3.0
        ang = np.arctan2(d[:, 1], d[:, 0]) - np.arctan2(o[:, 4], o[:, 3])
        ang = (ang + np.pi) % (2 * np.pi) - np.pi
        a = np.stack([np.where(np.abs(ang) < 0.6, 1.0, 0.0), np.clip(2 * ang, -1, 1)], 1).astype(np.float32)
        a += rs.normal(0, 0.05, a.shape).astype(np.float32)
        env.step(a, auto_reset=True)
        o, zo, r, dn, gm = env.results()
        sh, need, avail, goal = env.goal_info()
        for i, e in enumerate(refs):
            r_ref, d_ref, g_ref, sh_ref, need_ref = e.step_goal(a[i])
            assert (r[i], dn[i], gm[i]) == (np.float32(r_ref), d_ref, g_ref), (t, i)
            assert sh[i] == sh_ref and need[i] == need_ref, (t, i, sh[i], sh_ref)
            n_reached += need_ref and not d_ref
            if d_ref:
                e.reset(21 + i)
                n_resets += 1
        o_ref = np.stack([e.obs()[0] for e in refs])
        assert np.array_equal(o, o_ref), t
    assert n_reached > 50 or env_id == "PointTTSP-v0"      # TimedTSP mostly ends by timeout
    assert n_resets > n or env_id == "ColourMatch-v0"
    with pytest.raises(Z.ZenvError):
        env.set_goals(np.full(n, 99, np.int32))
    with pytest.raises(Z.ZenvError):
        env.rollout(3, Z.POLICY_GREEDY)
    env.close()


@pytest.mark.parametrize("fresh", [False, True])
def test_solver_ordered_variant_lockstep(zenv_mod, oracle_mod, fresh):
    """TSPOrderEnv (TSP_order_env.py:13-113) -- route per episode (the bank's aux column: the
    built-in tour for even envs, an arbitrary caller-supplied permutation for odd ones), order feature
    0.5^(position in the remaining route), shaped reward towards the route's first zone; auto-reset on.
    fresh=False is the reference's reset(): an episode's first observation -- from reset() and from the auto-reset inside
    a step -- is built before generate_route() (:108-113) and carries the feature of the route the env was left with
    (zeros at first, zeros after a finished episode, the leftover after a time-limit end: the short horizon here makes
    most episodes end that way).  fresh=True: the build's opt-out."""
    Z, O = zenv_mod, oracle_mod
    n, T, nz = 120, 300, 15
    cfg = Z.config_for_id("PointTSP-v0", num_steps=140)
    env = Z.ZoneVecEnv(cfg, n)
    env.enable_order(fresh_route_in_first_obs=fresh)
    rs = np.random.RandomState(3)
    robots, zones, ranks = [], [], []
    for i in range(n):
        robot, zxy, _, _ = Z.sample_layout(cfg, 50 + i)
        robots.append(robot); zones.append(zxy)
        ranks.append(Z.route_ranks(robot, zxy) if i % 2 == 0 else rs.permutation(nz).astype(np.int32))
    env.set_bank(np.array(robots), np.array(zones), aux=np.array(ranks), seeds=50 + np.arange(n))
    env.schedule_sequential()
    env.reset()
    refs = [O.OracleEnv(oracle_config_from(O, cfg)) for _ in range(n)]
    for i, e in enumerate(refs):
        e.reset_order(50 + i, ranks[i], fresh_first_obs=fresh)
        assert np.allclose(e.layout[1], zones[i], atol=0, rtol=0)
    sh, val = env.order_info()
    assert np.array_equal(val, np.stack([e.order_vals() for e in refs])) and not sh.any()
    assert val.any() == fresh                                  # the reference's very first obs: self.route = [] (:27)
    pos = env.order_routes()
    assert all([int(z) for z in np.argsort(pos[i])] == refs[i].route for i in range(n))
    # the built-in tour visits every zone once and is never longer than the plain nearest-neighbour tour
    r0 = ranks[0]
    assert sorted(r0) == list(range(nz))
    n_visits = n_resets = n_stale = 0
    for t in range(T):
        o, zo = env.observations()
        # steer to the first zone of the remaining route, with noise
        pos = env.order_routes()
        tgt_idx = np.argmax(pos == 0, axis=1)
        d = zo[np.arange(n), tgt_idx, :2] * 3.0 - o[:, 1:3] * 3.0
        ang = np.arctan2(d[:, 1], d[:, 0]) - np.arctan2(o[:, 4], o[:, 3])
        ang = (ang + np.pi) % (2 * np.pi) - np.pi
        a = np.stack([np.where(np.abs(ang) < 0.6, 1.0, 0.0), np.clip(2 * ang, -1, 1)], 1).astype(np.float32)
        a += rs.normal(0, 0.05, a.shape).astype(np.float32)
        env.step(a, auto_reset=True)
        _, _, r, dn, _ = env.results()
        sh, val = env.order_info()
        pos = env.order_routes()
        for i, e in enumerate(refs):
            r_ref, d_ref, _, sh_ref = e.step_order(a[i])
            assert (r[i], dn[i]) == (np.float32(r_ref), d_ref) and sh[i] == sh_ref, (t, i, sh[i], sh_ref)
            n_visits += r_ref >= 1.0
            if d_ref:
                left = e.route
                e.reset_order(50 + i, ranks[i], fresh_first_obs=fresh)     # penv.py:8-11: `if done: obs = env.reset()`
                n_resets += 1
                if not fresh and left:
                    n_stale += 1
                    assert e.order_vals()[left[0]] == 1.0 and len(e.route) == nz
            assert [int(z) for z in np.argsort(pos[i]) if pos[i][z] >= 0] == e.route, (t, i)
        assert np.array_equal(val, np.stack([e.order_vals() for e in refs])), t
    assert n_visits > n and n_resets > n and (fresh or n_stale > n // 2)
    with pytest.raises(Z.ZenvError):
        env.enable_goals()                                    # one variant per handle
    env.close()
    env = Z.ZoneVecEnv(cfg, 4)
    env.build_bank(1, 4)
    with pytest.raises(Z.ZenvError):
        env.enable_order()                                    # routes ride in the bank: enable first
    env.close()


@pytest.mark.parametrize("task,zones", [(0, 25), (1, 15), (2, 6)])
def test_persistent_launch_slices_do_not_change_results(zenv_mod, task, zones):
    """zenv_set_rollout_slice: a batch stepped in launches over 64, 128 or 200 of its envs, or in one launch over all of
    them (0), ends in the same state bit for bit (ragged last slice and ragged last tile included)."""
    Z = zenv_mod
    n, T = 333, 300
    blobs = []
    for sl in (0, 64, 128, 200, 65536):
        cfg = Z.default_config(task, zones, zones_keepout=0.40 if zones == 25 else 0.55, num_steps=120)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(5, 6 * n)
        env.schedule_sequential(stride=n)
        env.reset()
        env.set_rollout_slice(sl)
        env.rollout(T, Z.POLICY_GREEDY, policy_seed=9)
        env.rollout(70, Z.POLICY_UNIFORM, policy_seed=9, env_index0=17)
        blobs.append(env.get_state())
        assert env.get(Z.F_EPISODES).sum() > n
        env.close()
    for b in blobs[1:]:
        assert np.array_equal(b, blobs[0])
    with pytest.raises(Z.ZenvError):
        env = Z.ZoneVecEnv(Z.default_config(0, 5), 8)
        try:
            env.set_rollout_slice(-1)
        finally:
            env.close()
