"""Full-size (BASELINE.json configs 1-3: N = 65 536) checks through size-independent properties, since the
oracle cannot replay that many envs inside a test: invariants of the task logic on every env, invariance of
an env's trajectory under the batch it is stepped in, determinism, snapshot idempotence."""
import numpy as np
import pytest

from tests.helpers import oracle_config_from

pytestmark = pytest.mark.gpu
N = 65536


def _run(Z, task, zones, keepout, steps, mode, n=N, first_seed=1, env_index0=0):
    cfg = Z.default_config(task, zones, zones_keepout=keepout)
    env = Z.ZoneVecEnv(cfg, n)
    env.build_bank(first_seed, n, n_threads=16)
    env.reset()
    env.rollout(steps, Z.POLICY_GREEDY, env_index0=env_index0, mode=mode)
    return cfg, env


@pytest.mark.parametrize("task,zones,keepout", [(0, 25, 0.40), (1, 25, 0.40), (2, 6, 0.55)])
def test_full_size_invariants(zenv_mod, task, zones, keepout):
    Z = zenv_mod
    cfg, env = _run(Z, task, zones, keepout, 400, "persistent")
    o, zo, r, d, g = env.results()
    steps = env.get(Z.F_EP_LEN)
    # obs: remaining = 1 - steps/num_steps; heading is a unit vector; positions near the arena
    assert np.array_equal(o[:, 0], (1.0 - steps.astype(np.float64) / cfg.num_steps).astype(np.float32))
    assert np.abs(o[:, 3] ** 2 + o[:, 4] ** 2 - 1.0).max() < 1e-5
    assert np.abs(o[:, 1:3]).max() < 2.0 and np.isfinite(o).all() and np.isfinite(zo).all()   # no walls: may overshoot
    # zone rows: positions inside the placement extents, alpha 0.25, a valid colour
    assert np.abs(zo[:, :, :2]).max() <= 1.0 and (zo[:, :, 5] == 0.25).all()
    rgb = zo[:, :, 2:5]
    if task == 2:
        assert ((rgb == 0) | (rgb == 1)).all() and (rgb.sum(-1) == 1).all()               # Blue / Green / Red
        assert ((zo[:, :, 6] >= 0) & (zo[:, :, 6] <= 1)).all()                            # cooldown / max_cd
        ng, nr, nb = rgb[:, :, 1].sum(1), rgb[:, :, 0].sum(1), rgb[:, :, 2].sum(1)
        ham = np.minimum(np.minimum(ng * 2 + nr, nr * 2 + nb), nb * 2 + ng)               # colour_match_env.py:38-55
        # (an env auto-reset in this step reports the finished episode's count, its rows the new episode)
        assert np.array_equal(env.get(Z.F_VISIT_COUNT)[~d], ham.astype(np.int32)[~d])
    else:
        visited = (rgb == np.array([1, 1, 0], np.float32)).all(-1)                        # Yellow
        unvisited = (rgb == np.array([0, 1, 1], np.float32)).all(-1)                      # Cyan
        assert (visited ^ unvisited).all()
        assert np.array_equal(env.get(Z.F_VISIT_COUNT)[~d], visited.sum(1).astype(np.int32)[~d])
        running = steps > 0
        # TSP_env.py:41-42: while an episode runs its return is the number of cities visited
        assert np.array_equal(env.get(Z.F_EP_RETURN)[running], visited.sum(1)[running].astype(np.float64))
        if task == 1:
            t = zo[:, :, 6]
            assert (t[visited] == 1.0).all() and (t[unvisited] > 0).all()                 # else the episode had ended
    assert set(np.unique(d)) <= {False, True} and not (g & ~d).any()                      # goal_met implies done
    env.close()


def test_trajectory_does_not_depend_on_the_batch(zenv_mod):
    """Env g stepped inside the 65 536-env batch == env g stepped in a 70-env batch of its own (same map seeds,
    same global env index for the policy's counters), persistent and per-step launches alike."""
    Z = zenv_mod
    T, g0, m = 300, 40000, 70
    _, big = _run(Z, 0, 25, 0.40, T, "persistent")
    want = [x[g0:g0 + m] for x in big.results()]
    big.close()
    for mode in ("persistent", "per_step"):
        _, small = _run(Z, 0, 25, 0.40, T, mode, n=m, first_seed=1 + g0, env_index0=g0)
        for x, y in zip(want, small.results()):
            assert np.array_equal(x, y), mode
        small.close()


def test_determinism_and_snapshot_idempotence(zenv_mod):
    Z = zenv_mod
    _, a = _run(Z, 1, 25, 0.40, 150, "persistent")
    _, b = _run(Z, 1, 25, 0.40, 150, "per_step")
    sa, sb = a.get_state(), b.get_state()
    # the whole state blob, every array, from one launch per 150 steps and from 150 launches (an unfused
    # rollout differs in the action buffer only: it holds a_T-1, not the next action a_T)
    assert np.array_equal(sa, sb)
    a.set_state(sa)                                        # restoring a snapshot of itself changes nothing
    assert np.array_equal(a.get_state(), sa)
    a.rollout(50, Z.POLICY_GREEDY)
    b.rollout(50, Z.POLICY_GREEDY, mode="per_step")
    assert np.array_equal(a.get_state(), b.get_state())
    a.close(); b.close()
