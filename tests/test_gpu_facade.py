"""GPU tests of the host-side mirror of the reference interface (envs/, penv.py, evaluate.py):
they read like the reference's own call sites (make_fixed_env -> reset -> step loop of
main/scripts/evaluate.py:47-72; ParallelEnv of torch_ac/torch_utils/penv.py) and check every
value against the oracle."""
import numpy as np
import pytest

from tests.helpers import OracleBatch, oracle_config_from

pytestmark = pytest.mark.gpu


def _oracle_for(O, env_id, Zm, **over):
    return oracle_config_from(O, Zm.config_for_id(env_id, **over))


@pytest.mark.parametrize("env_id", ["PointTSP-v0", "PointTTSP-v0", "ColourMatch-v0"])
def test_make_fixed_env_episode_matches_oracle(zenv_mod, oracle_mod, env_id):
    """evaluate.py:47-72 for one map, 2 runs: same obs / reward / done / info as the oracle."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import make_fixed_env
    env_seed = 1000007
    env = make_fixed_env(env_id, hier=False, seed=0, env_seed=env_seed)
    ref = O.OracleEnv(_oracle_for(O, env_id, Zm))
    assert set(env.observation_space.spaces) == {"zone_obs", "obs"}
    assert env.observation_space.spaces["obs"].shape == (8,)
    assert env.action_space.shape == (2,) and (env.action_space.low == -1).all() and (env.action_space.high == 1).all()
    for run in range(2):
        obs = env.reset()
        o_ref, zo_ref = ref.reset(env_seed)          # every run replays the same map
        assert obs["obs"].dtype == np.float64 and obs["zone_obs"].shape == zo_ref.shape
        total, t = 0.0, 0
        while True:
            assert np.array_equal(obs["obs"].astype(np.float32), o_ref)
            assert np.array_equal(obs["zone_obs"].astype(np.float32), zo_ref)
            a = ref.policy(O.POLICY_GREEDY, o_ref, zo_ref, 0, t)
            obs, reward, done, info = env.step(a)
            r_ref, d_ref, g_ref = ref.step(a)
            o_ref, zo_ref = ref.obs()
            assert abs(reward - r_ref) <= 1e-5 and done == d_ref
            assert info.get("goal_met", False) == g_ref and info["cost"] == 0
            total += reward
            t += 1
            if done:
                break
            assert t < 2001
        assert t > 10
        with pytest.raises(AssertionError, match="reset before stepping"):
            env.step(a)
    env.close()


def test_engine_seed_increment_semantics(zenv_mod, oracle_mod):
    """Engine.reset does _seed += 1: make_test_env(seed=s) plays maps s, s+1, s+2, ..."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import make_test_env
    env = make_test_env("PointTTSP-v1", seed=77)
    ref = O.OracleEnv(_oracle_for(O, "PointTTSP-v1", Zm))
    for k in range(3):
        obs = env.reset()
        o_ref, zo_ref = ref.reset(77 + k)
        assert np.array_equal(obs["obs"].astype(np.float32), o_ref)
        assert np.array_equal(obs["zone_obs"].astype(np.float32), zo_ref)
    base = env.unwrapped
    assert base.num_cities == 5 and len(base.zones) == 5 and str(base.zones[0]) == "C"
    env.close()


def test_raw_env_dict_keys_and_zone_views(zenv_mod):
    from combinatorial_rl_tasks_amd.envs import make
    env = make("ColourMatch-v0")
    env.seed(5)
    obs = env.reset()
    assert list(obs) == ["remaining"] + [f"zones_lidar_{i}" for i in range(6)] + [
        "robot_pos", "robot_dir", "robot_velp", "robot_velr"]          # ZoneEnvBase.obs() order
    assert obs["zones_lidar_0"].shape == (7,) and obs["remaining"][0] == 1.0
    rs = np.random.RandomState(5)
    want = [["Blue", "Green", "Red"][int(rs.choice(3))] for _ in range(6)]
    assert [repr(z) for z in env.zones] == want                       # colour_match_env.py:60-62
    assert env.zone_cooldowns == [0] * 6 and env.goal_dist > 0
    env.close()


def test_parallel_env_train_semantics(zenv_mod, oracle_mod):
    """train_ppo.py:110-112 + penv.py: P envs with rng_seed = seed + 10000*i, auto-reset."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import make_train_env
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    P, tasks = 6, 20
    envs = [make_train_env("PointTSP-v1", num_training_tasks=tasks, rng_seed=1 + 10000 * i) for i in range(P)]
    penv = ParallelEnv(envs)
    cfg = oracle_config_from(O, Zm.config_for_id("PointTSP-v1"))
    rngs = [np.random.default_rng(1 + 10000 * i) for i in range(P)]
    draw = lambda i: int(rngs[i].integers(low=1, high=tasks + 1, size=1)[0])   # noqa: E731
    refs = [O.OracleEnv(cfg) for _ in range(P)]
    obs = penv.reset()
    ref_obs = [refs[i].reset(draw(i)) for i in range(P)]
    n_done = 0
    for t in range(1300):
        for i in range(P):
            assert np.array_equal(obs[i]["obs"].astype(np.float32), ref_obs[i][0])
            assert np.array_equal(obs[i]["zone_obs"].astype(np.float32), ref_obs[i][1])
        acts = np.stack([refs[i].policy(O.POLICY_GREEDY, ref_obs[i][0], ref_obs[i][1], i, t) for i in range(P)])
        obs, reward, done, info = penv.step(acts)
        assert len(obs) == len(reward) == len(done) == len(info) == P
        for i in range(P):
            r, d, g = refs[i].step(acts[i])
            assert abs(reward[i] - r) <= 1e-5 and done[i] == d and info[i].get("goal_met", False) == g
            if d:                      # worker(): obs = env.reset() with a freshly drawn seed
                ref_obs[i] = refs[i].reset(draw(i))
                n_done += 1
            else:
                ref_obs[i] = refs[i].obs()
    assert n_done >= P
    penv.close()


def test_parallel_env_step_no_reset(zenv_mod):
    from combinatorial_rl_tasks_amd.envs import make_train_env
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    envs = [make_train_env("PointTSP-v1", hier=True, num_training_tasks=5, rng_seed=i) for i in range(3)]
    penv = ParallelEnv(envs)
    penv.vec.cfg.num_steps  # noqa: B018  (config is shared)
    penv.reset()
    acts = np.zeros((3, 2), np.float32)
    for t in range(1000):
        obs, rew, done, info = penv.step_no_reset(acts)
    assert all(done) and all(i == {"cost": 0} for i in info)
    obs, rew, done, info = penv.step_no_reset(acts)        # WaitWrapper no-op
    assert all(done) and rew == (0.0, 0.0, 0.0) and all(i == {} for i in info)
    assert not obs[0]["obs"].any() and not obs[0]["zone_obs"].any()
    penv.close()


def test_evaluate_protocol(zenv_mod, oracle_mod):
    """evaluate.py: maps x runs batch == sequential oracle episodes with the scripted policy."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.evaluate import evaluate
    out = evaluate("PointTSP-v1", Zm.POLICY_GREEDY, n_maps=12, n_runs_per_map=2, env_seed0=1000000)
    ref = O.rollout(_oracle_for(O, "PointTSP-v1", Zm), np.arange(1000000, 1000012), 1000, O.POLICY_GREEDY,
                    seed_stride=10 ** 6, n_threads=4)        # first episode of each map
    got = np.array(out["return"])
    assert got.shape == (12, 2) and np.array_equal(got[:, 0], got[:, 1])
    # the oracle driver restarts after the first episode; compare through a lock-step replay
    cfg = _oracle_for(O, "PointTSP-v1", Zm)
    for m in range(12):
        e = O.OracleEnv(cfg)
        o, zo = e.reset(1000000 + m)
        total, t, d = 0.0, 0, False
        while not d:
            a = e.policy(O.POLICY_GREEDY, o, zo, 2 * m, t)
            r, d, g = e.step(a)
            o, zo = e.obs()
            total += r
            t += 1
        assert got[m, 0] == total and out["length"][m][0] == t and out["goal_met"][m][0] == g
    assert ref["episodes"].min() >= 1


def test_torch_policy_closed_loop_without_host_copies(zenv_mod, oracle_mod):
    """SURVEY 8(f) row 1, the plumbing half: a device-resident torch policy consumes the env's obs
    buffers as aliasing tensors and feeds its action tensor straight to zenv_step on torch's stream
    (replaces base.py:139-145's .cpu().numpy() round trip).  The same actions replayed through the
    host path and through the oracle give the same observations."""
    torch = pytest.importorskip("torch")
    from combinatorial_rl_tasks_amd.torch_interop import TorchZoneEnv
    Z, O = zenv_mod, oracle_mod
    n, T = 257, 60
    cfg = Z.config_for_id("PointTTSP-v0", num_steps=40)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(7, n)
    tenv = TorchZoneEnv(env)
    obs = tenv.reset()
    assert obs["obs"].data_ptr() == env.device_ptr(Z.F_OBS) and obs["zone_obs"].shape == (n, 15, 7)
    g = torch.Generator(device="cuda").manual_seed(3)
    w1 = torch.randn(8 + 7, 32, device="cuda", generator=g) * 0.5
    w2 = torch.randn(8 + 32, 2, device="cuda", generator=g) * 0.5
    ob = OracleBatch(O, _oracle_for(O, "PointTTSP-v0", Z, num_steps=40), range(7, 7 + n))
    ob.reset()
    actions_log = []
    for t in range(T):
        # a ZoneEnvModel-shaped policy (env_model.py:70-79): per-zone MLP on [obs, zone row], mean-pool, head
        x = torch.cat([obs["obs"][:, None, :].expand(n, 15, 8), obs["zone_obs"]], dim=-1)
        emb = torch.relu(x @ w1).mean(dim=1)
        a = torch.tanh(torch.cat([obs["obs"], emb], dim=-1) @ w2).contiguous()
        actions_log.append(a)                       # stays on the device; no sync in the loop
        obs, reward, done, goal = tenv.step(a)
    torch.cuda.synchronize()
    o_dev, zo_dev = obs["obs"].cpu().numpy(), obs["zone_obs"].cpu().numpy()
    for a in actions_log:
        ob.step(a.cpu().numpy())
    o_ref, zo_ref = ob.obs()
    assert np.array_equal(o_dev, o_ref) and np.array_equal(zo_dev, zo_ref)
    assert tenv.episodes.sum().item() > n          # TimedTSP with num_steps 40: everyone restarted
    env.set_stream(None)
    o_host, zo_host = env.observations()
    assert np.array_equal(o_host, o_ref) and np.array_equal(zo_host, zo_ref)
    env.close()


def test_next_city_env_facade(zenv_mod, oracle_mod):
    """PointTSP-v3 (TSPNextCityEnv): gym surface + set_goal / get_available_goals / info keys, single env and
    folded into a ParallelEnv (zone-goals penv.py:76-99)."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import TSPNextCityEnv, make
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    env = make("PointTSP-v3")
    assert isinstance(env, TSPNextCityEnv)
    env.seed(77)
    env.reset()
    ref = O.OracleEnv(_oracle_for(O, "PointTSP-v0", Zm))
    ref.reset(77)
    with pytest.raises(AssertionError):
        env.step(np.zeros(2, np.float32))                    # no goal yet
    assert env.get_available_goals().all()
    env.set_goal(4)
    ref.set_goal(4)
    assert np.allclose(env.get_goal() * 3.0, ref.layout[1][4], atol=1e-6)
    for t in range(40):
        a = np.array([1.0, 0.3 * np.sin(t / 3)], np.float32)
        _, r, d, info = env.step(a)
        r_ref, d_ref, _, sh_ref, need_ref = ref.step_goal(a)
        assert (r, d, info["shaped_reward"], info["need_next_goal"]) == (r_ref, d_ref, sh_ref, need_ref)
    env.close()
    envs = [make("PointTSP-v3") for _ in range(3)]
    for i, e in enumerate(envs):
        e.seed(100 + i)
    penv = ParallelEnv(envs)
    penv.reset()
    assert penv.needs_goal() == [True] * 3 and penv.available_goals(1).all()
    penv.set_goal(0, 2); penv.set_goal(1, 0); penv.set_goal(2, 14)
    assert penv.needs_goal() == [False] * 3
    obs, rew, done, info = penv.step(np.zeros((3, 2), np.float32))
    assert all("shaped_reward" in i and i["need_next_goal"] is False for i in info)
    assert penv.get_goal(2).shape == (2,)
    penv.close()
    cm = make("ColourMatch-v3")
    cm.seed(5)
    cm.reset()
    assert cm.get_available_goals().all()
    cm.set_goal(3)
    _, _, _, info = cm.step(np.array([1.0, 0.0], np.float32))
    assert set(info) >= {"shaped_reward", "need_next_goal"} and info["need_next_goal"] is False
    cm.close()


def test_order_env_facade(zenv_mod, oracle_mod):
    """PointTSP-v2 (TSPOrderEnv): (Z,7) rows with the order feature, info['shaped_reward'], the route property;
    a caller-supplied route function replaces the built-in tour.  reset() returns the reference's first observation
    (built before generate_route(), TSP_order_env.py:108-113): zeros on a new object, the leftover route's feature
    after an episode the time limit ended."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import TSPOrderEnv, make
    from combinatorial_rl_tasks_amd.envs.registry import config_point
    env = make("PointTSP-v2")
    assert isinstance(env, TSPOrderEnv) and env.observation_space.spaces["zones_lidar_0"].shape == (7,)
    env.seed(9)
    obs = env.reset()
    ref = O.OracleEnv(_oracle_for(O, "PointTSP-v0", Zm))
    ref.reset(9)
    robot, zxy = ref.layout
    rank = Zm.route_ranks(robot, zxy)
    ref = O.OracleEnv(_oracle_for(O, "PointTSP-v0", Zm))
    ref.reset_order(9, rank)
    assert env.route == list(np.argsort(rank)) == ref.route
    feat = lambda ob: np.array([ob[f"zones_lidar_{i}"][6] for i in range(15)], np.float32)
    assert np.array_equal(feat(obs), ref.order_vals()) and not feat(obs).any()       # self.route = [] (:27)
    for t in range(30):
        a = np.array([1.0, 0.2], np.float32)
        obs, r, d, info = env.step(a)
        r_ref, d_ref, _, sh_ref = ref.step_order(a)
        assert (r, d, info["shaped_reward"]) == (r_ref, d_ref, sh_ref)
        assert np.array_equal(feat(obs), ref.order_vals()) and feat(obs).max() == 1.0
    env.close()
    # a short horizon: the time limit ends episode 1 with cities left; episode 2's first obs shows that leftover
    cfg = dict(config_point, num_steps=40)
    env = TSPOrderEnv(cfg)
    ocfg = _oracle_for(O, "PointTSP-v0", Zm)
    ocfg.num_steps = 40
    ref = O.OracleEnv(ocfg)
    env.seed(20)
    env.reset()
    ref.reset(20)
    rank1 = Zm.route_ranks(*ref.layout)
    ref = O.OracleEnv(ocfg)
    ref.reset_order(20, rank1)
    done = False
    while not done:
        o, zo = ref.obs()
        tgt = ref.route[0]
        g = zo[tgt, :2] * 3.0 - o[1:3] * 3.0
        ang = (np.arctan2(g[1], g[0]) - np.arctan2(o[4], o[3]) + np.pi) % (2 * np.pi) - np.pi
        a = np.array([1.0 if abs(ang) < 0.6 else 0.0, np.clip(2 * ang, -1, 1)], np.float32)
        obs, r, done, info = env.step(a)
        r_ref, d_ref, _, sh_ref = ref.step_order(a)
        assert (r, done, info["shaped_reward"]) == (r_ref, d_ref, sh_ref)
    left = ref.route
    assert 0 < len(left) and env.route == left
    probe = O.OracleEnv(ocfg)
    probe.reset(21)
    rank2 = Zm.route_ranks(*probe.layout)
    obs = env.reset()                                                     # env.seed advanced to 21 (Engine.reset)
    ref.reset_order(21, rank2)
    assert np.array_equal(feat(obs), ref.order_vals()) and feat(obs)[left[0]] == 1.0
    assert env.route == list(np.argsort(rank2)) == ref.route
    obs, _, _, info = env.step(np.zeros(2, np.float32))
    _, _, _, sh_ref = ref.step_order(np.zeros(2, np.float32))
    assert np.array_equal(feat(obs), ref.order_vals()) and info["shaped_reward"] == sh_ref
    env.close()
    rev = make("PointTSP-v2", route_fn=lambda robot, zones: np.arange(15)[::-1], fresh_route_in_first_obs=True)
    rev.seed(9)
    first = rev.reset()
    assert rev.route == list(range(14, -1, -1)) and feat(first)[14] == 1.0
    rev.close()


@pytest.mark.gpu
def test_parallel_env_of_order_envs_auto_reset_first_obs(zenv_mod, oracle_mod):
    """ParallelEnv over TSPOrderEnv workers: the observation an auto-reset returns is `env.reset()`'s (penv.py:8-11),
    i.e. TSPOrderEnv's init_obs built before generate_route() (TSP_order_env.py:108-113) -- rows (Z,7) whose order
    feature belongs to the route the finished episode left behind, info['shaped_reward'] on every step."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import TSPOrderEnv
    from combinatorial_rl_tasks_amd.envs.registry import config_point
    from combinatorial_rl_tasks_amd.envs.wrappers import ZoneWrapper
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    P, H = 6, 60
    envs = []
    for i in range(P):
        e = TSPOrderEnv(dict(config_point, num_steps=H))
        e.seed(700 + 10 * i)
        envs.append(ZoneWrapper(e))
    pe = ParallelEnv(envs)
    obs = pe.reset()
    ocfg = _oracle_for(O, "PointTSP-v0", Zm)
    ocfg.num_steps = H

    def rank_of(seed):
        probe = O.OracleEnv(ocfg)
        probe.reset(seed)
        return Zm.route_ranks(*probe.layout)
    refs, seeds = [], [700 + 10 * i for i in range(P)]
    for i in range(P):
        r = O.OracleEnv(ocfg)
        r.reset_order(seeds[i], rank_of(seeds[i]))
        refs.append(r)
    for i in range(P):
        assert obs[i]["zone_obs"].shape == (15, 7) and not obs[i]["zone_obs"][:, 6].any()
        assert np.array_equal(obs[i]["zone_obs"][:, :6].astype(np.float32), refs[i].obs()[1])
    n_stale = 0
    for t in range(3 * H + 5):
        a = np.zeros((P, 2), np.float32)
        for i, r in enumerate(refs):
            o, zo = r.obs()
            g = zo[r.route[0], :2] * 3.0 - o[1:3] * 3.0
            ang = (np.arctan2(g[1], g[0]) - np.arctan2(o[4], o[3]) + np.pi) % (2 * np.pi) - np.pi
            a[i] = [1.0 if abs(ang) < 0.6 else 0.0, np.clip(2 * ang, -1, 1)]
        obs, rew, done, info = pe.step(a)
        for i, r in enumerate(refs):
            r_ref, d_ref, _, sh_ref = r.step_order(a[i])
            assert (rew[i], done[i], info[i]["shaped_reward"]) == (r_ref, d_ref, sh_ref), (t, i)
            if d_ref:
                left = r.route
                seeds[i] += 1                                             # Engine.reset: _seed += 1
                r.reset_order(seeds[i], rank_of(seeds[i]))
                n_stale += bool(left)
                if left:
                    assert obs[i]["zone_obs"][left[0], 6] == 1.0
            assert np.array_equal(obs[i]["zone_obs"][:, 6].astype(np.float32), r.order_vals()), (t, i)
            assert np.array_equal(obs[i]["zone_obs"][:, :6].astype(np.float32), r.obs()[1]), (t, i)
    assert n_stale >= P
    pe.close()


@pytest.mark.gpu
def test_step_results_is_step_plus_every_download(zenv_mod):
    """zenv_step_results (one upload, one launch, ONE download of the results slab, one synchronisation) against
    zenv_step + zenv_get of each field; the slab's pieces are 256-byte aligned, in ZENV_RESULT_* order."""
    import ctypes as C
    Z = zenv_mod
    nat = Z._native
    for env_id, n in (("PointTSP-v0", 16), ("ColourMatch-v0", 1), ("PointTTSP-v0", 333)):
        cfg = Z.config_for_id(env_id)
        a_env, b_env = Z.ZoneVecEnv(cfg, n), Z.ZoneVecEnv(cfg, n)
        off = (C.c_int64 * nat.N_RESULTS)()
        total = nat.lib().zenv_results_layout(a_env._h, off)
        sizes = [n * 8 * 4, n * 4, n, n, n, n * cfg.num_zones * a_env.zone_feat * 4]
        assert list(off) == sorted(off) and all(o % 256 == 0 for o in off) and total % 256 == 0
        assert all(off[i] + sizes[i] <= (off[i + 1] if i + 1 < nat.N_RESULTS else total) for i in range(nat.N_RESULTS))
        for e in (a_env, b_env):
            e.build_bank(1, 4 * n)
            e.schedule_sequential(stride=n)
            e.reset()
        first = a_env.step_results(None)
        assert np.array_equal(first[0], b_env.get(Z.F_OBS)) and np.array_equal(first[1], b_env.get(Z.F_ZONE_OBS))
        rs = np.random.RandomState(3)
        for t in range(40):
            act = rs.uniform(-1, 1, (n, 2)).astype(np.float32)
            got = a_env.step_results(act, auto_reset=True, copy=(t % 2 == 0))
            b_env.step(act, auto_reset=True)
            for arr, f in zip(got, (Z.F_OBS, Z.F_ZONE_OBS, Z.F_REWARD, Z.F_DONE, Z.F_GOAL_MET, Z.F_EXCEPTION)):
                assert np.array_equal(np.asarray(arr).astype(b_env.get(f).dtype), b_env.get(f)), (env_id, t, f)
        with pytest.raises(ValueError):
            a_env.step_results(np.zeros((n + 1, 2), np.float32))
        a_env.close()
        b_env.close()


def test_plain_seeded_parallel_env_never_runs_out_of_maps(zenv_mod, oracle_mod):
    """[not vendored] Engine.reset: `_seed += 1` at every reset, for ever -- make_test_env (make_env.py:20-35) envs in a
    ParallelEnv play seed, seed + 1, seed + 2, ...  16 PointTSP-v1 envs, >= 1000 episodes EACH (a ring of 3 maps per env
    on the device, refilled by the host sampler behind every reset: zenv_schedule_ring + zenv_bank_update), every obs /
    reward / done in lock-step with the oracle playing the same seed stream; two explicit reset() calls in between
    consume a seed each, as Engine.reset does."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import make_test_env
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    P, base = 16, 4000
    penv = ParallelEnv([make_test_env("PointTSP-v1", seed=base + 100000 * i) for i in range(P)], episodes_per_env=3)
    cfg = oracle_config_from(O, Zm.config_for_id("PointTSP-v1"))
    refs = [O.OracleEnv(cfg) for _ in range(P)]
    nxt = [base + 100000 * i for i in range(P)]           # the seed the next reset of env i plays

    def ref_reset(i):
        out = refs[i].reset(nxt[i])
        nxt[i] += 1
        return out
    penv.reset()
    ref_obs = [ref_reset(i) for i in range(P)]
    episodes = np.zeros(P, np.int64)
    t = 0
    while episodes.min() < 1000:
        if t in (5000, 5001):                               # Engine.reset consumes a seed whenever it is called
            obs = penv.reset()
            ref_obs = [ref_reset(i) for i in range(P)]
            for i in range(P):
                assert np.array_equal(obs[i]["obs"].astype(np.float32), ref_obs[i][0]), (t, i)
        acts = np.stack([refs[i].policy(O.POLICY_GREEDY, ref_obs[i][0], ref_obs[i][1], i, t) for i in range(P)])
        o, zo, r, d, g = penv.step_arrays(acts)
        for i in range(P):
            rr, dd, gg = refs[i].step(acts[i])
            assert np.float32(rr) == r[i] and dd == d[i] and gg == g[i], (t, i)
            if dd:
                ref_obs[i] = ref_reset(i)
                episodes[i] += 1
            else:
                ref_obs[i] = refs[i].obs()
            if dd or t % 64 == 0:                           # every episode boundary + a sample of ordinary steps
                assert np.array_equal(o[i], ref_obs[i][0]) and np.array_equal(zo[i], ref_obs[i][1]), (t, i, episodes[i])
        t += 1
        assert t < 2_000_000
    assert episodes.min() >= 1000 and np.array_equal(penv.vec.get(Zm.F_EPISODES), episodes)
    assert np.array_equal(penv.vec.get(Zm.F_SEED), np.array(nxt) - 1)      # each env is on the last seed it drew
    penv.close()


def test_skill_boundary_step_resets_envs_left_finished(zenv_mod, oracle_mod):
    """The fixed-length-skill loop (torch_ac/algos/hier_base.py:179-183): skill_len - 1 `step_no_reset` calls, then one
    `step`, on make_train_env(hier=True) envs (WaitWrapper(ZoneWrapper(FixedSeedsWrapper)), make_env.py:12-14).  An env
    that finished inside the skill idles as WaitWrapper's no-op (zero obs, reward 0, done, info {}) and the boundary
    `step` brings it back: the worker's `if done: obs = env.reset()` (penv.py:8-11) returns the next episode's first obs
    -- a freshly drawn seed -- with reward 0 and done."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import make_train_env
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    P, skill_len, tasks = 8, 25, 20
    penv = ParallelEnv([make_train_env("PointTSP-v1", hier=True, num_training_tasks=tasks, rng_seed=3 + 10000 * i)
                        for i in range(P)])
    cfg = oracle_config_from(O, Zm.config_for_id("PointTSP-v1"))
    refs = [O.OracleEnv(cfg) for _ in range(P)]
    rngs = [np.random.default_rng(3 + 10000 * i) for i in range(P)]
    draw = lambda i: int(rngs[i].integers(low=1, high=tasks + 1, size=1)[0])   # noqa: E731
    obs = penv.reset()
    ref_obs = [refs[i].reset(draw(i)) for i in range(P)]
    revived = 0
    for t in range(1500):
        acts = np.stack([refs[i].policy(O.POLICY_GREEDY, ref_obs[i][0], ref_obs[i][1], i, t) for i in range(P)])
        boundary = (t + 1) % skill_len == 0
        obs, rew, done, info = (penv.step if boundary else penv.step_no_reset)(acts)
        for i in range(P):
            if refs[i].e.done:                              # WaitWrapper: the inner env is already done
                assert rew[i] == 0.0 and done[i] and info[i] == {}, (t, i)
                if boundary:
                    ref_obs[i] = refs[i].reset(draw(i))
                    revived += 1
                else:
                    ref_obs[i] = (np.zeros(8, np.float32), np.zeros_like(ref_obs[i][1]))
            else:
                r, d, g = refs[i].step(acts[i])
                assert abs(rew[i] - r) <= 1e-5 and done[i] == d and info[i].get("goal_met", False) == g, (t, i)
                # the terminal step's own obs under step_no_reset, the next episode's first obs under step
                ref_obs[i] = refs[i].reset(draw(i)) if (d and boundary) else refs[i].obs()
            assert np.array_equal(obs[i]["obs"].astype(np.float32), ref_obs[i][0]), (t, i)
            assert np.array_equal(obs[i]["zone_obs"].astype(np.float32), ref_obs[i][1]), (t, i)
    assert revived >= P
    penv.close()


@pytest.mark.parametrize("env_id,n_free", [("PointTSP-v4", 5), ("PointTSP-v5", 3)])
def test_goal_conditioned_hard_instances_of_the_zone_goals_tree(zenv_mod, oracle_mod, env_id, n_free):
    """zone-goals/envs/TSP_hard_env.py:11: there TSPHardEnv derives from TSPNextCityEnv -- `make(id, tree="zone-goals")`.
    Pre-visited cities are not available goals; a greedy walk over the free cities, every step against the oracle's
    goal-conditioned step (orc_step_goal) on the same hard config."""
    Zm, O = zenv_mod, oracle_mod
    from combinatorial_rl_tasks_amd.envs import TSPHardEnv, TSPNextCityEnv, make
    env = make(env_id, tree="zone-goals")
    assert isinstance(env, TSPHardEnv) and isinstance(env, TSPNextCityEnv)
    assert type(make(env_id)).__name__ == "TSPHardEnv"          # main/ registers the plain TSPEnv subclass
    env.seed(31)
    obs = env.reset()
    ref = O.OracleEnv(_oracle_for(O, env_id, Zm))
    o_ref, zo_ref = ref.reset(31)
    avail = env.get_available_goals()
    assert avail.sum() == n_free and avail[:n_free].all() and np.array_equal(avail, ref.available_goals())
    with pytest.raises(AssertionError):
        env.set_goal(n_free)                                     # a city that starts visited
    t = 0
    done = False
    while not done and t < 1000:
        if env.goal_zone is None:
            g = int(np.flatnonzero(env.get_available_goals())[0])
            env.set_goal(g)
            ref.set_goal(g)
        a = ref.policy(O.POLICY_GREEDY, o_ref, zo_ref, 0, t)
        obs, r, done, info = env.step(a)
        r_ref, d_ref, g_ref, sh_ref, need_ref = ref.step_goal(a)
        o_ref, zo_ref = ref.obs()
        assert abs(r - r_ref) <= 1e-5 and (done, info["shaped_reward"], info["need_next_goal"]) == (d_ref, sh_ref, need_ref), t
        # the raw env's observation dict (ZoneEnvBase.obs): one row per city + the robot's own entries
        rows = np.stack([obs[f"zones_lidar_{i}"] for i in range(15)]).astype(np.float32)
        rest = np.concatenate([obs[k] for k in ("remaining", "robot_pos", "robot_dir", "robot_velp", "robot_velr")])
        assert np.array_equal(rows, zo_ref) and np.array_equal(rest.astype(np.float32), o_ref), t
        t += 1
    assert done and t > 20
    env.close()
