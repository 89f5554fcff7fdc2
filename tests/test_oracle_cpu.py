"""CPU tests of the oracle: RNG/layout half pinned to numpy's RandomState and to the committed
golden vectors; step semantics pinned by truth tables derived from the reference sources and
by the analytic known-answers of the Point model."""
import ctypes as C
import itertools
import math
import os

import numpy as np
import pytest

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "reset_vectors.npz"))


# ------------------------------------------------------------------ RNG / layout
@pytest.mark.parametrize("seed", [0, 1, 7, 1000000, 1000001, 2 ** 32 - 1])
def test_randomstate_matches_numpy(oracle_mod, seed):
    O = oracle_mod
    rs, ns = O.RandomState(seed), np.random.RandomState(seed)
    assert [rs.u32() for _ in range(700)] == [int(x) for x in ns.randint(0, 2 ** 32, 700, dtype=np.uint32)]
    rs, ns = O.RandomState(seed), np.random.RandomState(seed)
    assert [rs.uniform(-2.45, 2.45) for _ in range(200)] == [ns.uniform(-2.45, 2.45) for _ in range(200)]
    assert [rs.choice(3) for _ in range(200)] == [int(ns.choice(3)) for _ in range(200)]
    assert [rs.beta(3, 1.5) for _ in range(200)] == [ns.beta(3, 1.5) for _ in range(200)]
    # interleaved, as the gaussian cache makes order matter
    for _ in range(50):
        assert rs.beta(3, 1.5) == ns.beta(3, 1.5)
        assert rs.uniform(0, 1) == ns.uniform(0, 1)
        assert rs.choice(7) == int(ns.choice(7))


def test_survey_spot_vectors(oracle_mod):
    """The three known-answer vectors recorded in SURVEY.md section 7."""
    O = oracle_mod
    rs = O.RandomState(1000001)
    assert [rs.uniform(-2.6, 2.6) for _ in range(4)] == [
        -1.8412163667869015, 1.6651534802335397, -1.6850701404184583, -1.3067369472388906]
    rs = O.RandomState(1000000)
    assert [int(rs.beta(3, 1.5) * 2000) for _ in range(15)] == [
        1690, 1579, 1406, 1714, 1374, 767, 1702, 694, 1639, 1482, 1882, 1751, 717, 1825, 1678]
    rs = O.RandomState(1000000)
    assert [rs.choice(3) for _ in range(6)] == [2, 2, 1, 0, 2, 0]


@pytest.mark.parametrize("tag,task,Z,keepout", [("z15", 0, 15, 0.55), ("z6", 2, 6, 0.55),
                                                ("z5", 0, 5, 0.55), ("z25k40", 0, 25, 0.40)])
def test_layouts_match_golden(oracle_mod, tag, task, Z, keepout):
    O = oracle_mod
    cfg = O.default_config(task, Z, zones_keepout=keepout)
    env = O.OracleEnv(cfg)
    for i, s in enumerate(GOLD["seeds"]):
        env.reset(int(s))
        robot, zones = env.layout
        assert np.array_equal(robot, GOLD[f"robot_{tag}"][i])
        assert np.array_equal(zones, GOLD[f"zones_{tag}"][i])
        assert env.e.layout_restarts == GOLD[f"restarts_{tag}"][i]


def test_task_randomness_matches_golden(oracle_mod):
    O = oracle_mod
    timed = O.OracleEnv(O.default_config(O.TASK_TIMED, 15))
    timed5 = O.OracleEnv(O.default_config(O.TASK_TIMED, 5, num_steps=1000))
    colour = O.OracleEnv(O.default_config(O.TASK_COLOUR, 6))
    for i, s in enumerate(GOLD["seeds"]):
        timed.reset(int(s)); timed5.reset(int(s)); colour.reset(int(s))
        assert np.array_equal(timed.state()["tmax"], GOLD["tmax_z15"][i])
        assert np.array_equal(timed5.state()["tmax"], GOLD["tmax_z5"][i])
        assert np.array_equal(colour.state()["colour"], GOLD["colours_z6"][i])


def test_layout_is_feasible_and_25_zones_need_the_synthetic_keepout(oracle_mod):
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TSP, 15))
    env.reset(3)
    robot, zones = env.layout
    pts = np.vstack([robot[:2], zones])
    keep = np.r_[0.4, np.full(15, 0.55)]
    for a, b in itertools.combinations(range(16), 2):
        assert np.hypot(*(pts[a] - pts[b])) >= keep[a] + keep[b]
    assert (np.abs(zones) <= 3 - 0.55).all() and (np.abs(robot[:2]) <= 3 - 0.4).all()
    # the reference keepout cannot place 25 zones -> ResamplingError
    with pytest.raises(RuntimeError):
        O.OracleEnv(O.default_config(O.TASK_TSP, 25)).reset(1)


# ------------------------------------------------------------------ deterministic sin/cos
def test_sincos_accuracy_and_exact_points(oracle_mod):
    O = oracle_mod
    assert O.sincos(0.0) == (0.0, 1.0)
    xs = np.concatenate([np.linspace(-130, 130, 20001), np.random.RandomState(0).uniform(-1e4, 1e4, 20000)])
    err = max(max(abs(O.sincos(x)[0] - math.sin(x)), abs(O.sincos(x)[1] - math.cos(x))) for x in xs)
    assert err <= 2.3e-16
    for x in xs[:2000]:
        s, c = O.sincos(x)
        assert abs(s * s + c * c - 1.0) < 5e-16


# ------------------------------------------------------------------ dynamics known answers
def test_terminal_speeds_are_the_obs_normalisers(oracle_mod):
    """ZoneEnvBase.py:223-224 divides by 1.5 and 3: the Point robot's terminal speeds."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TSP, 5))
    env.reset(11)
    for _ in range(600):
        env.step([1.0, 0.0])
    o, _ = env.obs()
    assert abs(math.hypot(o[5], o[6]) - 1.0) < 1e-4 and o[7] == 0.0
    env.reset(11)
    for _ in range(600):
        env.step([0.0, 1.0])
    o, _ = env.obs()
    assert abs(o[7] - 1.0) < 2e-3     # the offset COM bleeds a little energy into translation
    env.reset(11)
    for _ in range(100):
        env.step([0.03, 0.0])          # below the force clamp: v_inf = 0.3 * 0.03 / 0.01 = 0.9
    for _ in range(500):
        env.step([0.03, 0.0])
    o, _ = env.obs()
    assert abs(math.hypot(o[5], o[6]) * 1.5 - 0.9) < 1e-4


def test_free_decay_is_implicit_euler(oracle_mod):
    """a = 0, omega = 0: v_{n+1} = v_n * m / (m + h*b) per substep (Appendix A.5)."""
    O = oracle_mod
    cfg = O.default_config(O.TASK_TSP, 5)
    env = O.OracleEnv(cfg)
    env.reset(5)
    for _ in range(50):
        env.step([1.0, 0.0])
    v0 = np.array(env.state()["qvel"])
    assert v0[2] == 0.0
    env.step([0.0, 0.0])
    v1 = np.array(env.state()["qvel"])
    ratio = (cfg.mass / (cfg.mass + cfg.timestep * cfg.damping[0])) ** 10
    assert np.allclose(v1[:2], v0[:2] * ratio, rtol=1e-13, atol=0)


def test_action_is_clipped_and_throttle_saturates(oracle_mod):
    O = oracle_mod
    a, b = O.OracleEnv(O.default_config(0, 5)), O.OracleEnv(O.default_config(0, 5))
    a.reset(9); b.reset(9)
    for _ in range(20):
        a.step([7.5, -3.0])
        b.step([0.05, -1.0])           # |a0| >= forcerange already saturates the motor
    assert np.array_equal(a.state()["qpos"], b.state()["qpos"])


# ------------------------------------------------------------------ task truth tables
def _place(env, z, xy):
    env.e.zone_xy[z][0], env.e.zone_xy[z][1] = float(xy[0]), float(xy[1])


def _far(env):
    for z in range(env.Z):
        _place(env, z, (50.0 + z, 50.0))


def test_tsp_visit_rules(oracle_mod):
    """TSP_env.py:54-69: pre-physics pose, radius <= 0.2, lowest index wins, one per step."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TSP, 5))
    env.reset(2)
    _far(env)
    rx, ry = env.e.xpos[0], env.e.xpos[1]
    _place(env, 3, (rx + 0.2, ry))          # exactly on the rim: dist <= size counts
    _place(env, 1, (rx, ry - 0.1))
    _place(env, 4, (rx + 0.2000001, ry))    # just outside
    r, d, g = env.step([0, 0])
    assert (r, d, g) == (1.0, False, False)
    assert list(env.state()["visited"]) == [0, 1, 0, 0, 0]      # index 1 before index 3
    r, d, g = env.step([0, 0])
    assert r == 1.0 and list(env.state()["visited"]) == [0, 1, 0, 1, 0]
    r, d, g = env.step([0, 0])
    assert r == 0.0 and list(env.state()["visited"]) == [0, 1, 0, 1, 0]
    _, zo = env.obs()
    assert zo[1].tolist()[2:] == [1.0, 1.0, 0.0, 0.25] and zo[0].tolist()[2:] == [0.0, 1.0, 1.0, 0.25]


def test_tsp_visit_lags_physics_by_one_step(oracle_mod):
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TSP, 5))
    env.reset(2)
    _far(env)
    probe = O.OracleEnv(O.default_config(O.TASK_TSP, 5))
    probe.reset(2)
    for _ in range(40):
        probe.step([1.0, 0.0])
    target = (probe.e.xpos[0], probe.e.xpos[1])      # where the robot is after 40 steps
    _place(env, 0, target)
    rewards = [env.step([1.0, 0.0])[0] for _ in range(60)]
    first = rewards.index(1.0)
    # the robot enters the 0.2 disc several steps before step 40; whatever that step is, the
    # detection happens one env.step() after the pose first lies inside
    probe.reset(2)
    inside_at = None
    for t in range(60):
        if math.hypot(probe.e.xpos[0] - target[0], probe.e.xpos[1] - target[1]) <= 0.2:
            inside_at = t
            break
        probe.step([1.0, 0.0])
    assert first == inside_at          # step index (0-based) that sees the pose of step `inside_at`


def test_tsp_goal_bonus_and_time_limit(oracle_mod):
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TSP, 2, num_steps=50))
    env.reset(4)
    rx, ry = env.e.xpos[0], env.e.xpos[1]
    _place(env, 0, (rx, ry)); _place(env, 1, (rx, ry))
    assert env.step([0, 0]) == (1.0, False, False)
    r, d, g = env.step([0, 0])
    assert (d, g) == (True, True)
    assert r == 1 + (50 - 1) * 0.01      # TSP_env.py:37-39 with the pre-increment step count
    with pytest.raises(AssertionError):
        env.step([0, 0])                  # 'Environment must be reset before stepping'
    env.reset(4)
    _far(env)
    for t in range(49):
        assert env.step([0, 0]) == (0.0, False, False)
    assert env.step([0, 0]) == (0.0, True, False)
    o, _ = env.obs()
    assert o[0] == 0.0                    # remaining = 1 - 50/50


def test_timed_tsp_timeout(oracle_mod):
    """TTSP_env.py:62-71: after the base step, any unvisited zone with tmax - steps <= 0 ends it."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TIMED, 3))
    env.reset(6)
    _far(env)
    env.e.tmax[0], env.e.tmax[1], env.e.tmax[2] = 7, 3, 900
    rx, ry = env.e.xpos[0], env.e.xpos[1]
    _place(env, 1, (rx, ry))              # visited at step 1: its timeout no longer matters
    assert env.step([0, 0]) == (1.0, False, False)
    _, zo = env.obs()
    assert zo[1][6] == 1.0 and zo[0][6] == np.float32((7 - 1) / 2000) and zo[2][6] == np.float32(899 / 2000)
    for t in range(2, 7):
        assert env.step([0, 0]) == (0.0, False, False)
    r, d, g = env.step([0, 0])            # step 7: tmax[0] - 7 = 0 -> done, no bonus, no goal
    assert (r, d, g) == (0.0, True, False)
    _, zo = env.obs()
    assert zo[0][6] == 0.0


def test_colour_match_rules(oracle_mod):
    """colour_match_env.py:26-36,86-120: cooldown tick first, cycle B->G->R->B, cd = 150,
    reward = old - new Hamming distance (may be negative), bonus when it reaches 0."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_COLOUR, 3))
    env.reset(8)
    _far(env)
    for z, c in enumerate([0, 0, 1]):     # Blue Blue Green -> dist to blue = 2
        env.e.colour[z] = c
    env.e.goal_dist = 2
    rx, ry = env.e.xpos[0], env.e.xpos[1]
    _place(env, 2, (rx, ry))
    r, d, g = env.step([0, 0])            # Green -> Red: B B R, dist to blue = 1
    assert (r, d, g) == (1.0, False, False)
    st = env.state()
    assert list(st["colour"]) == [0, 0, 2] and list(st["cooldown"]) == [0, 0, 150] and st["goal_dist"] == 1
    _, zo = env.obs()
    assert zo[2].tolist() == [zo[2][0], zo[2][1], 1.0, 0.0, 0.0, 0.25, 1.0]
    for t in range(149):                  # cooling down: no change, cd ticks 150 -> 1
        assert env.step([0, 0])[0] == 0.0
    assert env.state()["cooldown"][2] == 1
    r, d, g = env.step([0, 0])            # tick to 0 first, then eligible: Red -> Blue: goal
    assert g and d and r == 1 + (2000 - 150) * 0.01
    env.reset(8)
    _far(env)
    for z, c in enumerate([0, 0, 2]):     # B B R (dist 1); cycling a Blue makes it worse
        env.e.colour[z] = c
    env.e.goal_dist = 1
    _place(env, 0, (env.e.xpos[0], env.e.xpos[1]))
    assert env.step([0, 0])[0] == -2.0    # -> G B R: every target colour is now 3 cycles away, 1 - 3 = -2


def test_hamming_table(oracle_mod):
    """All 3^6 colourings against the reference formula (colour_match_env.py:38-55)."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_COLOUR, 6))
    env.reset(1)
    for z in range(6):
        _place(env, z, (60.0 + z, 60.0))
    _place(env, 0, (env.e.xpos[0], env.e.xpos[1]))
    for cols in itertools.product(range(3), repeat=6):
        for z, c in enumerate(cols):
            env.e.colour[z] = c
            env.e.cooldown[z] = 0
        env.e.goal_dist = 100
        env.e.done = 0
        env.e.steps = 0
        new = list(cols)
        new[0] = (new[0] + 1) % 3
        nb, ng, nr = new.count(0), new.count(1), new.count(2)
        want = min(ng * 2 + nr, nr * 2 + nb, nb * 2 + ng)
        r, d, g = env.step([0, 0])
        assert env.state()["goal_dist"] == want
        assert r == (100 - want) + (20.0 if want == 0 else 0.0)


def test_colour_degenerate_start(oracle_mod):
    """an all-equal start has goal_dist 0; the first step pays the bonus."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_COLOUR, 2))
    seed = next(s for s in range(1, 200) if len(set(np.random.RandomState(s).choice(3, 1)[0:1].tolist()
                                                     + [int(np.random.RandomState(s).choice(3, 2)[1])])) == 1)
    env.reset(seed)
    assert env.state()["goal_dist"] == 0
    r, d, g = env.step([0, 0])
    assert (r, d, g) == (20.0, True, True)


# ------------------------------------------------------------------ observation layout
def test_obs_layout_and_dtype_semantics(oracle_mod):
    O = oracle_mod
    env = O.OracleEnv(O.default_config(O.TASK_TSP, 15))
    o, zo = env.reset(1000000)
    e = env.e
    assert o[0] == 1.0 and o[5] == o[6] == o[7] == 0.0
    assert o[1] == np.float32(e.x0 / 3.0) and o[2] == np.float32(e.y0 / 3.0)
    q0, q3 = np.float32(math.cos(e.rot / 2)), np.float32(math.sin(e.rot / 2))
    # ZoneEnvBase.py:221-222 under numpy 1.21 promotion: float32 components, float64 arithmetic
    assert abs(o[3] - np.float32(float(q0) ** 2 - float(q3) ** 2)) <= 1.2e-7
    assert abs(o[4] - np.float32(2 * float(q0) * float(q3))) <= 1.2e-7
    assert abs(o[3] - math.cos(e.rot)) < 1e-6 and abs(o[4] - math.sin(e.rot)) < 1e-6
    for z in range(15):
        assert zo[z].tolist() == [np.float32(e.zone_xy[z][0] / 3.0), np.float32(e.zone_xy[z][1] / 3.0),
                                  0.0, 1.0, 1.0, 0.25]
    for _ in range(30):
        env.step([1.0, 0.5])
    o, _ = env.obs()
    assert o[0] == np.float32(1.0 - 30 / 2000)
    assert abs(math.hypot(o[3], o[4]) - 1.0) < 1e-6


def test_cooldown_feature_is_promotion_independent():
    """np.float32(cd)/150 is float64 under numpy 1.21 and float32 under NEP 50: same float32."""
    for cd in range(151):
        assert np.float32(cd / 150.0) == np.float32(np.float32(cd) / np.float32(150))


# ------------------------------------------------------------------ scripted policies
def test_philox_known_answer(oracle_mod):
    """Philox4x32-10, counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8 (Random123 KAT)."""
    O = oracle_mod
    env = O.OracleEnv(O.default_config(0, 5))
    o, zo = env.reset(1)
    a = env.policy(O.POLICY_UNIFORM, o, zo, env_index=0, step_index=0, policy_seed=0)
    want0 = np.float32(2.0) * (np.float32(0x6627e8d5 >> 8) * np.float32(2.0 ** -24)) - np.float32(1.0)
    want1 = np.float32(2.0) * (np.float32(0xe169c58d >> 8) * np.float32(2.0 ** -24)) - np.float32(1.0)
    assert a[0] == want0 and a[1] == want1


def test_greedy_policy_solves_pointtsp(oracle_mod):
    """The scripted controller is strong enough to exercise every visit / goal / reset branch."""
    O = oracle_mod
    cfg = O.default_config(O.TASK_TSP, 15)
    out = O.rollout(cfg, np.arange(1000000, 1000016), 2000, O.POLICY_GREEDY, n_threads=4)
    assert (out["episodes"] >= 1).all()
    assert out["last_return"].mean() > 15.0        # all 15 zones + a time bonus on most maps


def test_rollout_wrapping_map_bank(oracle_mod):
    """orc_rollout_wrapped: a bank of `seed_period` maps per env replayed in order."""
    O = oracle_mod
    cfg = O.default_config(O.TASK_TIMED, 6, zones_keepout=0.55, num_steps=40)
    seeds = np.arange(11, 19)
    a = O.rollout(cfg, seeds, 300, O.POLICY_GREEDY, seed_stride=7, seed_period=1, n_threads=2)
    b = O.rollout(cfg, seeds, 300, O.POLICY_GREEDY, seed_stride=0, n_threads=2)
    assert a["episodes"].min() >= 7
    for k in ("obs", "zone_obs", "episodes", "last_return", "reward_sum"):
        assert np.array_equal(a[k], b[k])
    c = O.rollout(cfg, seeds, 300, O.POLICY_GREEDY, seed_stride=7, seed_period=3, n_threads=2)
    d = O.rollout(cfg, seeds, 300, O.POLICY_GREEDY, seed_stride=7, n_threads=2)       # no wrap
    assert not np.array_equal(c["zone_obs"], d["zone_obs"])
    # the first three episodes are the same maps in both
    e = O.rollout(cfg, seeds, 10, O.POLICY_GREEDY, seed_stride=7, seed_period=3, n_threads=2)
    f = O.rollout(cfg, seeds, 10, O.POLICY_GREEDY, seed_stride=7, n_threads=2)
    assert e["episodes"].max() < 3 and np.array_equal(e["zone_obs"], f["zone_obs"])


def test_goal_conditioned_semantics(oracle_mod):
    """TSPNextCityEnv truth table (TSP_next_city_env.py:53-100), written from the reference source:
    step asserts a goal; set_goal asserts unvisited; shaped_reward = last_dist - dist, 0 in the step that reaches
    the goal; need_next_goal when the goal is reached or the episode ends; the goal is cleared then."""
    O = oracle_mod
    cfg = O.default_config(O.TASK_TSP, 5, num_steps=1000)
    e = O.OracleEnv(cfg)
    e.reset(1000000)
    with pytest.raises(AssertionError):
        e.step_goal([0.0, 0.0])                               # :54 assert goal_zone is not None
    with pytest.raises(AssertionError):
        e.set_goal(5)
    robot, zones = e.layout
    e.set_goal(2)
    d0 = math.hypot(zones[2][0] - robot[0], zones[2][1] - robot[1])
    assert e.e.goal_zone == 2 and e.e.last_dist == d0          # :87-88
    total, reached_at = 0.0, None
    for t in range(1000):
        o, zo = e.obs()
        g = zo[2, :2] * 3.0 - o[1:3] * 3.0
        ang = (math.atan2(g[1], g[0]) - math.atan2(o[4], o[3]) + math.pi) % (2 * math.pi) - math.pi
        r, d, gm, sh, need = e.step_goal([1.0 if abs(ang) < 0.6 else 0.0, max(-1.0, min(1.0, 2 * ang))])
        if need:
            reached_at = t
            assert r == 1.0 and sh == 0.0 and e.e.goal_zone == -1 and e.e.visited[2]     # :60-61, :69-72
            break
        total += sh
    assert reached_at is not None and reached_at > 10
    # the shaped rewards telescope to (initial distance - distance one step before arrival), i.e. ~ d0 - 0.2
    assert abs(total - (d0 - 0.2)) < 0.05
    with pytest.raises(AssertionError):
        e.set_goal(2)                                         # :86 visited zones are not valid goals
    assert list(e.available_goals()) == [not e.e.visited[z] for z in range(5)]
    # a timeout also asks for the next goal (TTSP_next_city_env.py:46-49)
    cfg = O.default_config(O.TASK_TIMED, 5, num_steps=50)
    e = O.OracleEnv(cfg)
    e.reset(7)
    e.set_goal(0)
    need = False
    for t in range(50):
        r, d, gm, sh, need = e.step_goal([0.0, 0.0])
        if d:
            break
    assert d and need and e.e.goal_zone == -1
    # ColourMatchNextCityEnv: any zone may be the goal; cycling another zone costs 1
    cfg = O.default_config(O.TASK_COLOUR, 6)
    e = O.OracleEnv(cfg)
    e.reset(3)
    assert e.available_goals().all()
    e.set_goal(5)
    assert e.e.goal_zone == 5


def test_actor_network_restatement_is_self_consistent():
    """oracle/policy_ref.py: the bf16-emulated forward stays within bf16 tolerance of the float32 one, and
    moving the mean in front of the (linear) third zone layer -- what the kernels do -- is exact algebra."""
    torch = pytest.importorskip("torch")
    from oracle import policy_ref as P
    t = P.random_tensors(7, h=185, seed=4, critic=True)
    rs = np.random.RandomState(1)
    obs = rs.uniform(-1, 1, (40, 8)).astype(np.float32)
    zo = rs.uniform(-1, 1, (40, 15, 7)).astype(np.float32)
    mu, std, val = P.forward_fp32(t, obs, zo)
    mu_e, std_e, val_e = P.forward_bf16_emulated(t, obs, zo)
    assert mu.shape == (40, 2) and val.shape == (40,) and (std > 1e-3).all() and (np.abs(mu) < 1).all()
    assert np.abs(mu - mu_e).max() < 2e-2 and np.abs(std - std_e).max() < 2e-2 and np.abs(val - val_e).max() < 3e-2
    td = {k: torch.as_tensor(v, dtype=torch.float64) for k, v in t.items()}
    x = torch.cat([torch.as_tensor(obs, dtype=torch.float64)[:, None, :].expand(40, 15, 8),
                   torch.as_tensor(zo, dtype=torch.float64)], -1)
    h2 = torch.relu(torch.relu(x @ td["zone_w1"].T + td["zone_b1"]) @ td["zone_w2"].T + td["zone_b2"])
    after = (h2 @ td["zone_w3"].T + td["zone_b3"]).sum(1) / 15            # env_model.py:78
    before = (h2.sum(1) / 15) @ td["zone_w3"].T + td["zone_b3"]            # mlp_policy.hip
    assert (after - before).abs().max() < 1e-12


def test_solver_ordered_semantics_and_route_heuristic(oracle_mod, zenv_mod):
    """TSPOrderEnv truth table (TSP_order_env.py:37-75, written from the source) on the oracle, and the host
    route heuristic of the product (no GPU needed): a permutation, never longer than nearest neighbour."""
    O, Z = oracle_mod, zenv_mod
    cfg = O.default_config(O.TASK_TSP, 5, num_steps=400)
    e = O.OracleEnv(cfg)
    rank = np.array([2, 0, 4, 1, 3], np.int32)                # route = [1, 3, 0, 4, 2]
    e.reset_order(1000003, rank, fresh_first_obs=True)        # (the reference's own first obs: next test)
    robot, zones = e.layout
    assert list(e.order_vals()) == [0.25, 1.0, 0.0625, 0.5, 0.125]            # :41-45: 0.5 ** route.index(i)
    assert e.route == [1, 3, 0, 4, 2]
    assert e.e.last_dist == math.hypot(zones[1][0] - robot[0], zones[1][1] - robot[1])   # :112
    total, visits = 0.0, []
    for t in range(400):
        o, zo = e.obs()
        tgt = int(np.argmax(e.order_vals()))
        g = zo[tgt, :2] * 3.0 - o[1:3] * 3.0
        ang = (math.atan2(g[1], g[0]) - math.atan2(o[4], o[3]) + math.pi) % (2 * math.pi) - math.pi
        r, d, gm, sh = e.step_order([1.0 if abs(ang) < 0.6 else 0.0, max(-1.0, min(1.0, 2 * ang))])
        if r >= 1.0:
            visits.append(e.e.last_visit)
            assert sh == 0.0                                   # :64-66: re-based on the new first zone, reward 0
            vals = e.order_vals()
            assert vals[e.e.last_visit] == 0.0 and (e.e.route_len == 0 or vals.max() == 1.0)
        if d:
            break
    assert visits[:2] == [1, 3]                                # the agent follows the route
    # the product's host heuristic: TSP_Solver.get_optim_route's problem (closed tour from the robot, int64(10 x distance)
    # arcs, PATH_CHEAPEST_ARC + local search) -- a permutation, never costlier than its own first solution, and on small
    # maps (brute force) the optimum nearly always
    import itertools

    def int_costs(robot, zxy):
        pts = np.vstack([robot[:2], zxy])
        return (np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1)) * 10.0).astype(np.int64)

    def tour_cost(c, order):
        seq = [0] + [int(o) + 1 for o in order] + [0]
        return sum(c[a, b] for a, b in zip(seq[:-1], seq[1:]))
    for seed in range(5):
        robot, zxy, _, _ = Z.sample_layout(Z.default_config(0, 15), 100 + seed)
        rk = Z.route_ranks(robot, zxy)
        assert sorted(rk) == list(range(15))
        c = int_costs(robot, zxy)
        first, cur, left = [], 0, list(range(15))
        while left:                                            # PATH_CHEAPEST_ARC, lowest index on ties
            j = min(left, key=lambda z: (c[cur, z + 1], z))
            first.append(j); left.remove(j); cur = j + 1
        assert tour_cost(c, np.argsort(rk)) <= tour_cost(c, first)
    hits = 0
    for seed in range(12):
        robot, zxy, _, _ = Z.sample_layout(Z.default_config(0, 7), 100 + seed)
        c = int_costs(robot, zxy)
        best = min(tour_cost(c, p) for p in itertools.permutations(range(7)))
        got = tour_cost(c, np.argsort(Z.route_ranks(robot, zxy)))
        assert got <= 1.05 * best
        hits += got == best
    assert hits >= 10


def _steer(o, zo, tgt):
    g = zo[tgt, :2] * 3.0 - o[1:3] * 3.0
    ang = (math.atan2(g[1], g[0]) - math.atan2(o[4], o[3]) + math.pi) % (2 * math.pi) - math.pi
    return [1.0 if abs(ang) < 0.6 else 0.0, max(-1.0, min(1.0, 2 * ang))]


def test_order_env_first_observation_is_built_before_generate_route(oracle_mod):
    """TSPOrderEnv.reset() (TSP_order_env.py:108-113) returns `init_obs = super().reset()`, built BEFORE
    `self.generate_route()`: obs_zones (:37-47) sees self.route as the env object was left with it.  Truth table written
    from the source, with a plain Python list playing self.route:
      (0) before the first episode self.route = [] (:27)                      -> first obs: all zeros
      (a) episode 1 finished (every city visited, route emptied by :90)         -> episode 2's first obs: all zeros
      (b) episode 1 ended by the time limit with cities left                    -> episode 2's first obs: 0.5 ** index in
          the LEFTOVER route, by zone number, on the new map's rows
    and in every case the route proper, last_dist_to_goal (:112) and every later observation are the new episode's."""
    O = oracle_mod
    Zn = 5

    def feature(route):                                        # :41-45
        return [0.5 ** route.index(i) if i in route else 0.0 for i in range(Zn)]

    rank1 = np.array([2, 0, 4, 1, 3], np.int32)
    rank2 = np.array([4, 3, 2, 1, 0], np.int32)
    # ---- (0) + (a): a finished episode
    cfg = O.default_config(O.TASK_TSP, Zn, num_steps=2000)
    e = O.OracleEnv(cfg)
    e.reset_order(1000003, rank1)
    route = []                                                 # self.route = [] (:27) is what init_obs saw ...
    assert list(e.order_vals()) == feature(route) == [0.0] * Zn
    route = list(np.argsort(rank1))                            # ... then generate_route() (:111)
    assert e.route == route == [1, 3, 0, 4, 2]
    robot, zones = e.layout
    assert e.e.last_dist == math.hypot(zones[1][0] - robot[0], zones[1][1] - robot[1])   # :112, the NEW route's first city
    done, first_step = False, True
    while not done:
        o, zo = e.obs()
        r, done, gm, sh = e.step_order(_steer(o, zo, route[0]))
        if r >= 1.0:
            route.remove(e.e.last_visit)                       # :90
        assert list(e.order_vals()) == feature(route)          # from the first step on: the route proper
        assert first_step is False or e.order_vals().max() == 1.0
        first_step = False
    assert gm and route == [] and e.route == []
    e.reset_order(1000004, rank2)                              # the worker's `if done: obs = env.reset()` (penv.py:8-11)
    assert list(e.order_vals()) == [0.0] * Zn                  # (a)
    assert e.route == [4, 3, 2, 1, 0]
    e.step_order([0.0, 0.0])
    assert list(e.order_vals()) == feature([4, 3, 2, 1, 0])
    # ---- (b): the time limit ends episode 1 after two visits
    cfg = O.default_config(O.TASK_TSP, Zn, num_steps=2000)
    e = O.OracleEnv(cfg)
    e.reset_order(1000003, rank1)
    route = list(np.argsort(rank1))
    n_visits = 0
    for t in range(2000):
        o, zo = e.obs()
        a = _steer(o, zo, route[0]) if n_visits < 2 else [0.0, 0.0]    # two cities, then stand still until the limit
        r, done, gm, sh = e.step_order(a)
        if r >= 1.0:
            route.remove(e.e.last_visit)
            n_visits += 1
        if done:
            break
    assert done and not gm and e.e.steps == 2000 and route == [0, 4, 2]
    e.reset_order(1000004, rank2)
    assert list(e.order_vals()) == feature([0, 4, 2]) == [1.0, 0.0, 0.25, 0.0, 0.5]     # (b): the leftover, by zone number
    assert e.route == [4, 3, 2, 1, 0]
    robot, zones = e.layout                                    # the new map
    assert e.e.last_dist == math.hypot(zones[4][0] - robot[0], zones[4][1] - robot[1])
    o, zo = e.obs()
    _, _, _, sh = e.step_order(_steer(o, zo, 4))
    assert list(e.order_vals()) == feature([4, 3, 2, 1, 0])
    # shaped_reward of the first step is progress towards the NEW route's first city from the reset pose (:68-71)
    o2, _ = e.obs()
    d_after = math.hypot(zones[4][0] - 3.0 * float(o2[1]), zones[4][1] - 3.0 * float(o2[2]))
    assert abs(sh - (math.hypot(zones[4][0] - robot[0], zones[4][1] - robot[1]) - d_after)) < 1e-6
    # ---- the build's opt-out: the first observation already shows the new route
    e.reset_order(1000005, rank1, fresh_first_obs=True)
    assert list(e.order_vals()) == feature([1, 3, 0, 4, 2])


def test_oracle_regression_vectors(oracle_mod):
    """The oracle against its own frozen trajectories (tests/golden/oracle_regression.npz: NOT reference output, a pin
    against silent drift of the checker -- see make_oracle_regression.py)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_oracle_regression", os.path.join(here, "golden", "make_oracle_regression.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    gold = np.load(os.path.join(here, "golden", "oracle_regression.npz"))
    now = m.trajectories()
    assert sorted(now) == sorted(k for k in gold.files if k != "seeds")
    for k, v in now.items():
        assert v.dtype == gold[k].dtype and np.array_equal(v, gold[k]), k
    # the frozen runs contain visits, a TimedTSP episode end and colour changes
    assert (gold["tsp_greedy_reward"] > 0).sum() > 5 and gold["timed_greedy_flags"].any()
    assert (gold["colour_greedy_reward"] != 0).any()
