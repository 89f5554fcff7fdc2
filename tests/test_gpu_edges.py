"""GPU edge cases: smallest / largest zone counts (runtime-Z kernel path), ragged and tiny batches,
argument validation, out-of-range actions, a large batch."""
import numpy as np
import pytest

from tests.helpers import OracleBatch, oracle_config_from

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task,zones,keepout,n", [(0, 1, 0.55, 1), (0, 32, 0.30, 67), (1, 32, 0.30, 63),
                                                  (2, 32, 0.30, 65), (2, 1, 0.55, 5), (1, 2, 0.55, 129)])
def test_zone_count_extremes_lockstep(zenv_mod, oracle_mod, task, zones, keepout, n):
    Z, O = zenv_mod, oracle_mod
    cfg = Z.default_config(task, zones, zones_keepout=keepout, num_steps=300)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(77, n)
    env.schedule_sequential()
    env.reset()
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(77, 77 + n))
    o_ref, zo_ref = ob.reset()
    o, zo = env.observations()
    assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref)
    for t in range(420):
        a = ob.policy(O.POLICY_GREEDY if t % 5 else O.POLICY_UNIFORM, o_ref, zo_ref, t)
        env.step(a, auto_reset=True)
        r_ref, d_ref, g_ref = ob.step(a)
        o_ref, zo_ref = ob.obs()
        o, zo, r, d, g = env.results()
        assert np.array_equal(d, d_ref) and np.array_equal(g, g_ref), t
        assert np.array_equal(r, r_ref.astype(np.float32)), t
        assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref), t
    env.close()


def test_fused_rollout_runtime_zone_count(zenv_mod, oracle_mod):
    """Z = 32 / 9 have no compile-time instantiation: the generic kernel + fused policy path."""
    Z, O = zenv_mod, oracle_mod
    for task, zones in ((0, 32), (2, 9), (1, 11)):
        n, T = 200, 350
        cfg = Z.default_config(task, zones, zones_keepout=0.30, num_steps=150)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(1, 24 * n)
        env.schedule_sequential(stride=n)
        env.reset()
        env.rollout(T, Z.POLICY_GREEDY, policy_seed=3)
        ref = O.rollout(oracle_config_from(O, cfg), 1 + np.arange(n), T, O.POLICY_GREEDY, seed_stride=n,
                        policy_seed=3, n_threads=8)
        assert 0 < ref["episodes"].max() < 24
        assert np.array_equal(env.get(Z.F_EPISODES), ref["episodes"])
        assert np.array_equal(env.get(Z.F_OBS), ref["obs"])
        assert np.array_equal(env.get(Z.F_ZONE_OBS), ref["zone_obs"])
        assert np.array_equal(env.get(Z.F_LAST_RETURN), ref["last_return"])
        env.close()


def test_out_of_range_and_nonfinite_actions(zenv_mod, oracle_mod):
    """Engine.step clips to the actuator ctrlrange (+-inf included); a NaN survives np.clip and takes Engine.step's
    MujocoException branch: done, reward_exception, state reset -- identically on both sides, observations finite."""
    Z, O = zenv_mod, oracle_mod
    cfg = Z.config_for_id("PointTSP-v1")
    n = 8
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(5, n)
    env.reset()
    ob = OracleBatch(O, oracle_config_from(O, cfg), range(5, 5 + n))
    ob.reset()
    a = np.array([[5.0, -7.0], [-np.inf, np.inf], [1e-30, -1e-30], [0.049, 0.9], [0.051, -0.9],
                  [np.nan, 0.5], [1.0, np.nan], [-0.0, 0.0]], np.float32)
    for t in range(25):
        env.step(a, auto_reset=True)
        ob.step(a)
        o, zo = env.observations()
        o_ref, zo_ref = ob.obs()
        assert np.array_equal(o, o_ref) and np.array_equal(zo, zo_ref)
        assert np.isfinite(o).all()
        r, d = env.get(Z.F_REWARD), env.get(Z.F_DONE).astype(bool)
        assert (r[5:7] == -10.0).all() and d[5:7].all() and env.get(Z.F_EXCEPTION)[5:7].all()
        assert not d[:5].any() and not d[7]
    env.close()


def test_argument_validation(zenv_mod):
    Z = zenv_mod
    E = Z._native
    with pytest.raises(Z.ZenvError) as ei:
        Z.ZoneVecEnv(Z.default_config(0, 15), 0)
    assert ei.value.code == E.E_ARG
    for bad in (dict(num_zones=33), dict(num_zones=0), dict(num_steps=0), dict(max_cd=256), dict(frameskip=0),
                dict(kernel=2), dict(task=3)):
        cfg = Z.default_config(0, 15)
        for k, v in bad.items():
            setattr(cfg, k, v)
        with pytest.raises(Z.ZenvError):
            Z.ZoneVecEnv(cfg, 4)
    with pytest.raises(Z.ZenvError):
        Z.ZoneVecEnv(Z.default_config(0, 15), 4, device=99)
    env = Z.ZoneVecEnv("PointTSP-v1", 4)
    with pytest.raises(Z.ZenvError) as ei:
        env.reset()                                   # no bank yet
    assert ei.value.code == E.E_STATE
    with pytest.raises(Z.ZenvError) as ei:
        env.step(np.zeros((4, 2), np.float32))        # 'Environment must be reset before stepping'
    assert ei.value.code == E.E_STATE
    env.build_bank(1, 3)
    with pytest.raises(Z.ZenvError):
        env.schedule_sequential(first=np.array([0, 1, 2, 3], np.int32))   # slot 3 outside the bank
    with pytest.raises(Z.ZenvError):
        env.schedule_fixed_seeds(np.zeros(4, np.uint64), 1, 100)          # bank holds 3 seeds, not 100
    with pytest.raises(ValueError):
        env.reset(np.zeros(3, np.uint8))
    env.reset()
    with pytest.raises(ValueError):
        env.step(np.zeros((5, 2), np.float32))
    with pytest.raises(Z.ZenvError):
        env.set_state(np.zeros(10, np.uint8))         # wrong blob size
    env.close()
    env.close()                                       # idempotent


def test_large_batch_runs(zenv_mod, oracle_mod):
    """1 M envs in one handle (the 8-GPU config's per-node total): state + bank stay under 2 GB."""
    Z, O = zenv_mod, oracle_mod
    n = 1 << 20
    cfg = Z.default_config(0, 25, zones_keepout=0.40)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(1, 4096, n_threads=16)
    env.schedule_sequential()
    env.reset()
    env.rollout(30, Z.POLICY_GREEDY)
    ref = O.rollout(oracle_config_from(O, cfg), 1 + (np.arange(4096 * 3, 4096 * 3 + 64) % 4096), 30,
                    O.POLICY_GREEDY, env_index0=4096 * 3)
    assert np.array_equal(env.get(Z.F_OBS)[4096 * 3:4096 * 3 + 64], ref["obs"])
    assert np.array_equal(env.get(Z.F_ZONE_OBS)[-64:], env.get(Z.F_ZONE_OBS)[4096 - 64:4096])   # same maps
    env.close()


def test_pinned_host_buffers_and_batched_download(zenv_mod):
    """zenv_host_alloc / zenv_get_many: the host-policy surface through page-locked buffers gives the same
    arrays as zenv_get into pageable ones."""
    Z = zenv_mod
    n = 300
    env = Z.ZoneVecEnv(Z.config_for_id("PointTTSP-v0"), n)
    env.build_bank(2, n)
    env.reset()
    a = env.pinned_array((n, 2), np.float32)
    a[:] = np.random.RandomState(0).uniform(-1, 1, (n, 2))
    fields = (Z.F_OBS, Z.F_ZONE_OBS, Z.F_REWARD, Z.F_DONE)
    bufs = [env.pinned_array(env._shape(f), t) for f, t in zip(fields, (np.float32, np.float32, np.float32, np.uint8))]
    for _ in range(3):
        env.step(a, auto_reset=True)
        env.results_into(fields, bufs)
        for f, b in zip(fields, bufs):
            assert np.array_equal(b, env.get(f))
    with pytest.raises(Z.ZenvError):
        env.results_into((99,), [bufs[0]])
    env.close()
    assert np.isfinite(bufs[0]).all()          # pinned arrays outlive the env


@pytest.mark.parametrize("n", [1, 63, 64, 65])
def test_persistent_rollout_tiny_and_ragged_batches(zenv_mod, oracle_mod, n):
    Z, O = zenv_mod, oracle_mod
    for task, zones in ((0, 25), (2, 6)):
        cfg = Z.default_config(task, zones, zones_keepout=0.40 if zones == 25 else 0.55, num_steps=90)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(31, 8 * n)
        env.schedule_sequential(stride=n)
        env.reset()
        env.rollout(300, Z.POLICY_GREEDY, policy_seed=5)        # 256 + 44 steps: two launches
        ref = O.rollout(oracle_config_from(O, cfg), 31 + np.arange(n), 300, O.POLICY_GREEDY, seed_stride=n,
                        policy_seed=5, seed_period=8, n_threads=2)
        assert np.array_equal(env.get(Z.F_OBS), ref["obs"]) and np.array_equal(env.get(Z.F_ZONE_OBS), ref["zone_obs"])
        assert np.array_equal(env.get(Z.F_EPISODES), ref["episodes"]) and ref["episodes"].min() >= 3
        env.close()


def test_c_abi_demo_matches_python_path(zenv_mod, tmp_path):
    """examples/c_abi_demo.c (plain C caller of include/zenv.h, no Python in the loop) produces the same numbers
    as the same sequence of calls through the ctypes facade."""
    import os
    import subprocess
    Z = zenv_mod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "combinatorial-rl-tasks_amd", "lib")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_demo.c"),
                    "-o", exe, "-L", lib_dir, "-lzenv_hip", "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined"],
                   check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout.split()
    got = {out[i]: float(out[i + 1]) for i in range(0, len(out), 2)}
    n = 256
    env = Z.ZoneVecEnv(Z.config_for_id("PointTSP-v0"), n)
    env.build_bank(1000000, n, n_threads=4)
    env.schedule_sequential()
    env.reset()
    a = np.zeros((n, 2), np.float32)
    a[:, 0] = 1.0
    a[:, 1] = (np.arange(n) % 3 - 1) * 0.5
    host_return, host_dones = 0.0, 0
    for _ in range(40):
        env.step(a, auto_reset=True)
        host_return += float(env.get(Z.F_REWARD).astype(np.float64).sum())
        host_dones += int(env.get(Z.F_DONE).sum())
    env.rollout(100, Z.POLICY_GREEDY, policy_seed=7)
    o, zo, r, _, _ = env.results()
    # the demo's fixed-length skill: 10 pre-computed steps as one zenv_step_many(ZENV_CHUNK_RESET_LAST)
    chunk = np.zeros((10, n, 2), np.float32)
    chunk[:, :, 0] = 1.0
    chunk[:, :, 1] = ((np.arange(n)[None, :] + np.arange(10)[:, None]) % 3 - 1) * 0.5
    env.step_many(chunk, reset="last")
    c_rew, c_done = env.chunk_results()
    c_obs = env.get(Z.F_OBS)
    assert got["steps"] == 150
    assert abs(got["chunk_return"] - float(c_rew.astype(np.float64).sum())) < 1e-3
    assert got["chunk_dones"] == int(c_done.sum())
    assert abs(got["chunk_obs_sum"] - float(c_obs.astype(np.float64).sum())) < 1e-6
    assert abs(got["obs_sum"] - float(o.astype(np.float64).sum())) < 1e-6
    assert abs(got["zone_obs_sum"] - float(zo.astype(np.float64).sum())) < 1e-6
    assert abs(got["reward_sum"] - float(r.astype(np.float64).sum())) < 1e-6
    assert abs(got["host_return"] - host_return) < 1e-3 and got["host_dones"] == host_dones and got["slab_ok"] == 1
    env.close()


def test_state_blob_with_a_nan_is_refused(zenv_mod):
    """MuJoCo resets a state that holds a NaN (mj_checkPos / mj_checkVel); the kernels rely on a finite one, so
    zenv_set_state refuses such a blob and leaves the handle untouched."""
    Z = zenv_mod
    env = Z.ZoneVecEnv("PointTSP-v1", 130)
    env.build_bank(3, 130)
    env.reset()
    env.rollout(20, Z.POLICY_GREEDY)
    want = env.results()
    blob = env.get_state()
    bad = blob.copy()
    bad.view(np.uint8)[16 + 8 * 5: 16 + 8 * 6] = np.frombuffer(np.float64(np.nan).tobytes(), np.uint8)   # qa[2].y
    with pytest.raises(Z.ZenvError) as ei:
        env.set_state(bad)
    assert ei.value.code == Z._native.E_ARG
    for x, y in zip(want, env.results()):
        assert np.array_equal(x, y)
    env.set_state(blob)
    env.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_plain_seeded_parallel_env_with_an_episode_ending_at_every_step(zenv_mod, depth):
    """The hardest case for the ring of pre-sampled maps (penv.py, zenv_schedule_ring + zenv_bank_update): every step
    ends every env's episode (a NaN action: Engine.step's exception branch), so every step takes a map from every ring
    and the host refills it before the next step -- with a ring of ONE map too.  Env i keeps playing Engine.reset's
    seed stream s_i, s_i + 1, s_i + 2, ...; explicit reset() calls consume a seed each as well."""
    Z = zenv_mod
    from combinatorial_rl_tasks_amd import envs
    from combinatorial_rl_tasks_amd.penv import ParallelEnv
    es = []
    for i in range(3):
        e = envs.make("PointTSP-v1")
        e.seed(100 * i)
        es.append(envs.ZoneWrapper(e))
    penv = ParallelEnv(es, episodes_per_env=depth)
    cfg = Z.config_for_id("PointTSP-v1")
    a = np.full((3, 2), np.nan, np.float32)            # a NaN action ends the episode at once (exception branch)
    k = 0                                              # resets so far: env i is on seed 100 i + k - 1
    for rnd in range(3):
        obs = penv.reset()
        k += 1
        for t in range(40):
            if t:
                obs, r, d, info = penv.step(a)
                k += 1
                assert all(d) and r == (-10.0,) * 3 and all(i == {"exception": True} for i in info)
            assert np.array_equal(penv.vec.get(Z.F_SEED), 100 * np.arange(3) + k - 1), (rnd, t)
            for i in range(3):                         # the first obs of that seed's map: zone rows = its layout / 3
                _, zones, _, _ = Z.sample_layout(cfg, 100 * i + k - 1)
                assert np.array_equal(obs[i]["zone_obs"][:, :2].astype(np.float32), (zones / 3.0).astype(np.float32))
    penv.close()


def test_integration_md_ctypes_stub_runs(zenv_mod):
    """The reference-side binding INTEGRATION.md shows (section 2) is executed as written -- only the library name is
    replaced by the in-tree path -- and its arrays are checked against the facade on the same calls."""
    import os
    import re
    Z = zenv_mod
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, numpy as np.*?)```", text, re.S).group(1)
    lib_path = os.path.join(root, "combinatorial-rl-tasks_amd", "lib", "libzenv_hip.so")
    assert 'C.CDLL("libzenv_hip.so")' in code
    ns = {}
    exec(compile(code.replace('C.CDLL("libzenv_hip.so")', f'C.CDLL({lib_path!r})'), "INTEGRATION.md", "exec"), ns)
    n = ns["N"]
    env = Z.ZoneVecEnv(Z.config_for_id("PointTSP-v0"), n)
    env.build_bank(1, 100, n_threads=8)
    env.schedule_fixed_seeds(np.arange(n, dtype=np.uint64) * 10000, 1, 100)
    env.reset()
    a = np.zeros((n, 2), np.float32)
    a[:, 0] = 1.0
    for _ in range(100):
        env.step(a, auto_reset=True)
    o, zo, r, d, _ = env.results()
    assert np.array_equal(ns["obs"], o) and np.array_equal(ns["zone_obs"], zo)
    assert np.array_equal(ns["reward"], r) and np.array_equal(ns["done"].astype(bool), d)
    env.close()


def test_set_state_rejects_a_hinge_velocity_beyond_the_small_angle_update(zenv_mod):
    """The kernels advance sin / cos of the hinge angle by h * omega per substep with Taylor kernels good to 0.1 rad;
    validate_config bounds what the model can reach, zenv_set_state applies the same bound (0.05) to a blob."""
    Z = zenv_mod
    env = Z.ZoneVecEnv("PointTSP-v1", 70)
    env.build_bank(1, 70)
    env.reset()
    env.rollout(20, Z.POLICY_GREEDY)
    blob = env.get_state()
    env.set_state(blob)                                       # its own state passes
    qc_off = 16 + 70 * 16 * 2                                 # head, qa, qb; then qc = (v1, v2) per env
    bad = blob.copy()
    bad[qc_off + 16 * 33 + 8: qc_off + 16 * 33 + 16] = np.frombuffer(np.float64(1e3).tobytes(), np.uint8)   # env 33: omega
    with pytest.raises(Z.ZenvError, match="small-angle"):
        env.set_state(bad)
    ok = blob.copy()
    ok[qc_off + 16 * 33 + 8: qc_off + 16 * 33 + 16] = np.frombuffer(np.float64(-20.0).tobytes(), np.uint8)  # 0.04 rad
    env.set_state(ok)
    env.close()
