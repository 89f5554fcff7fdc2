"""ParallelEnv -- drop-in for main/src/torch_ac/torch_utils/penv.py:26-66.

The reference runs one OS process per env and ships pickled ``(obs, reward, done, info)``
tuples over Pipes.  Here the same constructor argument (a list of wrapped single envs, as
built by ``make_train_env`` / ``make_fixed_env``) is folded into ONE device handle of
``len(envs)`` envs: ``step`` is one kernel launch, auto-reset (penv.py:7-11) happens inside
it, and each env keeps its own FixedSeedsWrapper seed stream (wrappers.py:10-23) on the device.
``reset/step/step_no_reset`` return exactly the reference's shapes; ``step_arrays`` returns the
struct-of-arrays results without building P Python dicts.
"""
import numpy as np

from . import _native as nat
from .envs.wrappers import FixedSeedsWrapper, WaitWrapper, ZoneWrapper
from .envs.zone_envs import ColourMatchNextCityEnv, TSPNextCityEnv, TSPOrderEnv, TSPOrderTestEnv, ZoneEnvBase
from .vec_env import ZoneVecEnv

_RING_DEPTH = 4   # default ring of pre-sampled maps per env for envs that are not behind a FixedSeedsWrapper


def _unwrap(env):
    fixed, chain = None, env
    while not isinstance(chain, ZoneEnvBase):
        if isinstance(chain, FixedSeedsWrapper):
            fixed = chain
        elif not isinstance(chain, (ZoneWrapper, WaitWrapper)):
            raise TypeError(f"ParallelEnv cannot batch {type(chain).__name__}")
        chain = chain.env
    return chain, fixed


class ParallelEnv:
    """A batch of zone envs stepped by one MI355X kernel launch."""

    def __init__(self, envs, device=0, episodes_per_env=_RING_DEPTH):
        """episodes_per_env: only for plain seeded envs (no FixedSeedsWrapper, e.g. make_test_env, make_env.py:20-35) --
        the depth of the ring of maps kept ahead of each env on the device.  Env i plays Engine.reset's unbounded
        stream _seed, _seed + 1, ... ([not vendored] Engine.reset: `_seed += 1` at every reset): every reset -- an
        explicit reset() or the auto-reset inside step() -- takes the next map from the env's ring and the host puts
        the map `depth` episodes further on into the slot it came from (zenv_bank_update), so the stream never ends."""
        assert len(envs) >= 1, "No environment given."
        self._plain_depth = None
        self.envs = envs
        self.observation_space = envs[0].observation_space
        self.action_space = envs[0].action_space
        bases, fixed = zip(*[_unwrap(e) for e in envs])
        cfg = bases[0]._cfg
        for b in bases[1:]:
            if bytes(b._cfg) != bytes(cfg):
                raise ValueError("all envs of a ParallelEnv must share one configuration")
        self.num_envs = len(envs)
        self._vec = ZoneVecEnv(cfg, self.num_envs, device=device)
        self._vec.host_io("if small")     # 16 procs: the kernel writes the results into host memory itself, no copies
        # solver-ordered envs (TSP_order_env.py; zone-goals' TSPOrderTestEnv): routes ride in the bank, so before it
        self._order = all(isinstance(b, TSPOrderEnv) for b in bases)
        if not self._order and any(isinstance(b, TSPOrderEnv) for b in bases):
            raise ValueError("mixing solver-ordered and plain envs is not supported")
        if self._order:
            if any(b._route_fn is not None for b in bases):
                raise NotImplementedError("ParallelEnv batches the built-in tour; a route_fn needs ZoneVecEnv.set_bank(aux=ranks)")
            if len({b._fresh_first_obs for b in bases}) != 1:
                raise ValueError("all envs of a ParallelEnv must share fresh_route_in_first_obs")
            self._vec.enable_order(fresh_route_in_first_obs=bases[0]._fresh_first_obs)
            self._order_shaped = not any(isinstance(b, TSPOrderTestEnv) for b in bases)   # TSP_order_test_env.py:72-74
        if all(f is not None for f in fixed):
            lo, hi = fixed[0].min_seed, fixed[0].max_seed
            if any((f.min_seed, f.max_seed) != (lo, hi) for f in fixed):
                raise ValueError("all FixedSeedsWrappers must share [min_seed, max_seed]")
            for f in fixed:   # the device stream starts where default_rng(rng_seed) starts
                fresh = np.random.default_rng(seed=f.rng_seed).bit_generator.state
                if f.rng.bit_generator.state != fresh:
                    raise NotImplementedError("wrap fresh envs: a FixedSeedsWrapper already drew seeds")
            self._vec.build_bank(int(lo), int(hi) - int(lo) + 1)
            self._vec.schedule_fixed_seeds(np.array([f.rng_seed for f in fixed], np.uint64), int(lo), int(hi))
        elif all(f is None for f in fixed):
            # Engine semantics: reset k of env i plays seed _seed_i + k (reset() does _seed += 1)
            depth = max(1, int(episodes_per_env))
            seeds = []
            for b in bases:
                if b._seed is None:
                    b.seed(None)
                seeds.append(int(b._seed) + np.arange(depth, dtype=np.int64))
            self._seed0 = np.array([s[0] for s in seeds], np.int64)     # seed of env i's episode 0
            self._vec.build_bank_seeds(np.concatenate(seeds))
            self._ring_first = np.arange(self.num_envs, dtype=np.int32) * depth
            self._vec.schedule_ring(self._ring_first, depth)
            self._consumed = np.zeros(self.num_envs, np.int64)           # maps env i has taken from its ring
            self._plain_depth = depth
        else:
            raise ValueError("mixing seeded and FixedSeedsWrapper envs is not supported")
        self._goals = all(isinstance(b, (TSPNextCityEnv, ColourMatchNextCityEnv)) for b in bases)
        if self._goals:
            self._vec.enable_goals()

    # ------------------------------------------------------------------ reference surface
    def reset(self):
        self._vec.reset()
        self._finished = None
        if self._plain_depth is not None:
            self._refill(np.arange(self.num_envs))
        return self._obs_list()

    def _refill(self, envs_idx):
        """envs_idx just took a map each (episode k = _consumed[i]) from their rings: put episode k + depth into the
        slot it came from.  The device never runs dry: a step ends at most one episode per env, and this runs before
        the next step is enqueued."""
        k = self._consumed[envs_idx]
        slots = self._ring_first[envs_idx] + (k % self._plain_depth)
        self._vec.update_bank(slots, self._seed0[envs_idx] + k + self._plain_depth)
        self._consumed[envs_idx] = k + 1

    def step(self, actions):
        return self._step(actions, True)

    def step_no_reset(self, actions):
        return self._step(actions, False)

    def step_chunk(self, actions, reset="last"):
        """K steps of pre-computed actions (K, P, 2) in ONE persistent launch (ZoneVecEnv.step_many).  reset="last" is the
        fixed-length-skill loop of _hier_policy_opt.py:68-71 -- K - 1 x step_no_reset, then one step; "never" / "every" are
        K x step_no_reset / K x step.  Returns (obs, rewards (K,P) float64, dones (K,P) bool, infos): obs and infos are
        what the chunk's LAST call returned, rewards / dones every call's."""
        a = np.asarray(actions, np.float32)
        K = a.shape[0]
        a = a.reshape(K, self.num_envs, 2)
        if self._goals or self._order:
            raise NotImplementedError("goal-conditioned / solver-ordered envs report per-step info: use step()")
        if reset == "every" and self._plain_depth is not None and K > self._plain_depth:
            raise ValueError(f"a ring of {self._plain_depth} maps per env allows {self._plain_depth} auto-resetting steps per "
                             "chunk (episodes_per_env); use reset='last' or a deeper ring")
        was_finished = getattr(self, "_finished", None)
        self._vec.step_many(a, reset=reset)
        rew, done = self._vec.chunk_results()
        o, zo, r, d, g, exc = self._vec.step_results(None, copy=False)
        o = o.astype(np.float64)
        zo = zo.astype(np.float64)
        if self._plain_depth is not None and reset != "never":
            # maps taken from the rings: one per done flag of an auto-resetting call (every call / the last one)
            taken = done.sum(0) if reset == "every" else done[-1].astype(np.int64)
            for _ in range(int(taken.max()) if taken.size else 0):
                idx = np.flatnonzero(taken > 0)
                self._refill(idx)
                taken[idx] -= 1
        infos = [{"cost": 0} for _ in range(self.num_envs)]
        finished_before_last = done[:-1].any(0) if K > 1 and reset != "every" else np.zeros(self.num_envs, bool)
        if was_finished is not None:
            finished_before_last |= was_finished
        for i in np.flatnonzero(finished_before_last):
            infos[i] = {}                                     # WaitWrapper no-op: info = {}
        for i in np.flatnonzero(g):
            infos[i]["goal_met"] = True
        for i in np.flatnonzero(exc & d & ~finished_before_last):
            infos[i] = {"exception": True}
        self._finished = None if reset != "never" else (done.any(0) if was_finished is None else (was_finished | done.any(0)))
        obs = tuple({"zone_obs": z, "obs": x} for z, x in zip(zo, o))
        return obs, rew.astype(np.float64), done, tuple(infos)

    def render(self):
        raise NotImplementedError

    # goal-conditioned envs: zone-goals/src/torch_ac/torch_utils/penv.py:76-99
    def set_goal(self, env_idx, goal):
        g = np.full(self.num_envs, -1, np.int32)
        g[env_idx] = goal
        self._vec.set_goals(g)

    def set_goals(self, goals):
        """Batched set_goal: int32 [P], -1 = leave that env alone (one call instead of P pipes)."""
        self._vec.set_goals(goals)

    def get_goal(self, env_idx):
        g = int(self._vec.get(nat.F_GOAL)[env_idx])
        assert g >= 0
        return self._vec.get(nat.F_ZONE_OBS)[env_idx][g, :2].astype(np.float64)

    def needs_goal(self):
        return [bool(x) for x in self._vec.get(nat.F_NEED_GOAL)]

    def available_goals(self, env_idx):
        mask = int(self._vec.get(nat.F_AVAILABLE_GOALS)[env_idx])
        return np.array([(mask >> i) & 1 for i in range(self._vec.num_zones)], bool)

    def close(self):
        self._vec.close()

    # ------------------------------------------------------------------ array surface
    def step_arrays(self, actions, auto_reset=True):
        """(obs (P,8), zone_obs (P,Z,F), reward (P,), done (P,), goal_met (P,)) float32/bool."""
        res = self._vec.step_results(np.asarray(actions, np.float32).reshape(self.num_envs, 2),
                                     auto_reset=auto_reset)[:5]
        if self._plain_depth is not None and auto_reset and res[3].any():
            self._refill(np.flatnonzero(res[3]))
        return res

    @property
    def vec(self):
        """The underlying ZoneVecEnv (device pointers, counters, rollouts)."""
        return self._vec

    # ------------------------------------------------------------------ helpers
    def _with_order_feature(self, zo):
        """TSPOrderEnv.obs_zones (:37-47): every row gains the order feature of the route the observation saw."""
        val = self._vec.get(nat.F_ORDER_VAL).astype(np.float64)
        return np.concatenate([zo, val[:, :, None]], axis=2)

    def _obs_list(self):
        o, zo = self._vec.step_results(None, copy=False)[:2]
        o = o.astype(np.float64)
        zo = zo.astype(np.float64)
        if self._order:
            zo = self._with_order_feature(zo)
        return [{"zone_obs": z, "obs": x} for z, x in zip(zo, o)]

    def _step(self, actions, auto_reset):
        a = np.asarray(actions, np.float32).reshape(self.num_envs, 2)
        o, zo, r, d, g, exc = self._vec.step_results(a, auto_reset=auto_reset, copy=False)
        o = o.astype(np.float64)            # the reference's obs are float64 (ZoneWrapper concatenates float64)
        zo = zo.astype(np.float64)
        if self._order:
            zo = self._with_order_feature(zo)
        P = self.num_envs
        was_finished = getattr(self, "_finished", None)
        any_done = bool(d.any())
        if self._plain_depth is not None and auto_reset and any_done:
            # every env that reports done was reset inside the step (also one left finished by step_no_reset: the
            # worker resets it after WaitWrapper's no-op, penv.py:8-11): each took a map from its ring
            self._refill(np.flatnonzero(d))
        # info dicts: {'cost': 0} for everybody, then the few envs with something to say
        infos = [{"cost": 0} for _ in range(P)]
        if was_finished is not None and was_finished.any():
            for i in np.flatnonzero(was_finished):
                infos[i] = {}                                 # WaitWrapper no-op: info = {}
        if any_done:
            for i in np.flatnonzero(g):
                infos[i]["goal_met"] = True
            for i in np.flatnonzero(exc & d):
                if was_finished is None or not was_finished[i]:
                    infos[i] = {"exception": True}            # Engine.step's MujocoException branch
        if getattr(self, "_goals", False):
            shaped, need = self._vec.goal_info()[:2]
            for i, (sr, ng) in enumerate(zip(shaped.tolist(), need.tolist())):
                infos[i]["shaped_reward"] = sr
                infos[i]["need_next_goal"] = bool(ng)
        if self._order and self._order_shaped:
            for i, sr in enumerate(self._vec.get(nat.F_SHAPED_REWARD).tolist()):
                infos[i]["shaped_reward"] = sr                # TSP_order_env.py:77-81
        self._finished = None if auto_reset else (d.copy() if was_finished is None else (was_finished | d))
        obs = tuple({"zone_obs": z, "obs": x} for z, x in zip(zo, o))
        return obs, tuple(r.astype(np.float64).tolist()), tuple(d.tolist()), tuple(infos)
