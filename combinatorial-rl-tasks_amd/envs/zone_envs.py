"""Single-environment facades with the reference's gym.Env surface.

Mirrors main/envs/zone_envs/ZoneEnvBase.py (ZoneEnvBase, zone), main/envs/TSP_env.py
(TSPEnv), main/envs/TTSP_env.py (TimedTSPEnv) and main/envs/colour_match_env.py
(ColourMatchEnv): same constructor ``config`` dicts (main/envs/__init__.py:7-50), same
``seed / reset / step / observation_space / action_space`` behaviour, same raw observation
dict (keys and order of ZoneEnvBase.obs(), ZoneEnvBase.py:190-224) so that the reference's
ZoneWrapper logic applies unchanged.  All arithmetic runs in the HIP library through one
device handle of N = 1 env; nothing is computed in Python.
"""
import copy
import enum

import numpy as np

from .. import _native as nat
from ..vec_env import ZoneVecEnv, default_config
from .spaces import Box, Dict


class zone(enum.Enum):          # ZoneEnvBase.py:13-30
    JetBlack = 0
    White = 1
    Blue = 2
    Green = 3
    Red = 4
    Yellow = 5
    Cyan = 6
    Magenta = 7

    def __lt__(self, sth):
        return self.value < sth.value

    def __str__(self):
        return self.name[0]

    def __repr__(self):
        return self.name


visited = zone.Yellow       # TSP_env.py:9-10
unvisited = zone.Cyan
colours = [zone.Blue, zone.Green, zone.Red]    # colour_match_env.py:9


class ZoneEnvBase:
    """Common part of the three task envs (Engine + ZoneEnvBase of the reference)."""

    _TASK = None
    metadata = {"render.modes": []}
    reward_range = (-float("inf"), float("inf"))

    def __init__(self, config, device=0, **native_overrides):
        config = copy.deepcopy(config)
        self.time_saved_reward = config.pop("time_saved_reward", 0.01)   # TSP_env.py:15
        self.num_cities = config.pop("num_cities")                        # TSP_env.py:16
        if config.get("walled", False):
            raise NotImplementedError("walled arenas are outside the MI355X hot path")
        robot_base = config.get("robot_base", "xmls/point.xml")
        if robot_base != "xmls/point.xml":
            raise NotImplementedError(f"only the Point robot is implemented, not {robot_base}")
        if config.get("observation_flatten", False):
            raise NotImplementedError("observation_flatten=True is not used by the zone envs")
        self.num_steps = int(config.get("num_steps", 1000))               # ZoneEnvBase.py:86
        self.zones_num = self.num_cities
        self.zones_size = 0.2
        self.zones_keepout = 0.55
        self.config = config
        self._cfg = default_config(self._TASK, self.num_cities, num_steps=self.num_steps,
                                   time_saved_reward=float(self.time_saved_reward),
                                   **self._native_overrides(), **native_overrides)
        self._vec = ZoneVecEnv(self._cfg, 1, device=device)
        self._vec.host_io(True)           # one env: a step is one launch and one wait, actions / results in host memory
        self._seed = None
        self.done = True      # Engine: must reset before the first step
        self.steps = 0
        self.action_space = Box(-1, 1, (2,), dtype=np.float32)
        self.build_observation_space()

    def _native_overrides(self):
        return {}

    # ------------------------------------------------------------------ gym.Env surface
    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        """Engine.seed: remember the seed for the next reset()."""
        self._seed = np.random.randint(2 ** 32) if seed is None else seed

    def reset(self):
        """Engine.reset: task randomness from RandomState(_seed), then _seed += 1 and the
        layout from RandomState(_seed) (TTSP_env.py:73-76, colour_match_env.py:125-127)."""
        if self._seed is None:
            self.seed(None)
        s = int(self._seed)
        self._vec.build_bank(s, 1, n_threads=1)
        self._vec.schedule_sequential()
        self._vec.reset()
        self._seed = s + 1
        self.done = False
        self.steps = 0
        return self.obs()

    def step(self, action):
        assert not self.done, "Environment must be reset before stepping"   # Engine.step
        a = np.asarray(action, dtype=np.float32).reshape(1, 2)
        # one call: action upload, step kernel, ONE download of every result, one synchronisation
        o, zo, r, d, g, exc = self._vec.step_results(a, auto_reset=False, copy=False)
        reward = float(r[0])
        self.done = bool(d[0])
        self.steps += 1
        info = {"cost": 0}
        if g[0]:
            info["goal_met"] = True
        if self.done and exc[0]:
            info = {"exception": True}       # Engine.step's MujocoException branch: no cost(), no goal test
        self._fresh = (o[0].astype(np.float64), zo[0].astype(np.float64))
        try:
            return self.obs(), reward, self.done, info
        finally:
            self._fresh = None

    def close(self):
        self._vec.close()

    def render(self, *args, **kwargs):
        raise NotImplementedError("rendering (mujoco-py viewers) is outside the hot path")

    # ------------------------------------------------------------------ observations
    def build_observation_space(self):
        d = {"remaining": Box(0.0, 1.0, (1,), dtype=np.float32)}
        F = self._vec.zone_feat
        for i in range(self.num_cities):
            d[f"zones_lidar_{i}"] = Box(-np.inf, np.inf, (F,), dtype=np.float32)
        d["robot_pos"] = Box(-np.inf, np.inf, (2,), dtype=np.float32)
        d["robot_dir"] = Box(-np.inf, np.inf, (2,), dtype=np.float32)
        d["robot_velp"] = Box(-np.inf, np.inf, (2,), dtype=np.float32)
        d["robot_velr"] = Box(-np.inf, np.inf, (1,), dtype=np.float32)
        self.obs_space_dict = d
        self.observation_space = Dict(d)

    def obs(self):
        """ZoneEnvBase.obs(): the raw dict, float64 like the reference's numpy arrays."""
        if getattr(self, "_fresh", None) is not None:      # inside step(): the arrays that call just downloaded
            o, zo = self._fresh
        else:
            o, zo = self._vec.step_results(None, copy=False)[:2]
            o, zo = o[0].astype(np.float64), zo[0].astype(np.float64)
        out = {"remaining": o[0:1]}
        for i in range(self.num_cities):
            out[f"zones_lidar_{i}"] = zo[i]
        out["robot_pos"] = o[1:3]
        out["robot_dir"] = o[3:5]
        out["robot_velp"] = o[5:7]
        out["robot_velr"] = o[7:8]
        return out

    # ------------------------------------------------------------------ state views
    @property
    def zones(self):
        raise NotImplementedError

    def goal_met(self):
        return bool(self._vec.get(nat.F_GOAL_MET)[0])


class TSPEnv(ZoneEnvBase):
    """PointTSP (main/envs/TSP_env.py)."""
    _TASK = nat.TASK_TSP

    @property
    def zones(self):
        st = self._vec.debug_state()["zone_state"][0]
        return [visited if v else unvisited for v in st]


class TSPHardEnv(TSPEnv):
    """PointTSP-v4 / -v5 (main/envs/TSP_hard_env.py:11-29 with config_zone_fixed_1/_2, envs/__init__.py:52-81): the
    first cities and the robot sit at fixed locations (Engine 'zones_locations', 'robot_locations', 'robot_rot'), the
    other cities are sampled as usual but start the episode already visited ('zones_colours': zone.Yellow)."""

    def __init__(self, config, **kw):
        config = copy.deepcopy(config)
        self.zone_colours = config.pop("zones_colours", None)              # :14
        if self.zone_colours is None or len(self.zone_colours) != config["num_cities"]:
            raise ValueError("TSPHardEnv needs one zones_colours entry per city")
        bad = [c for c in self.zone_colours if zone(c) not in (visited, unvisited)]
        if bad:
            raise ValueError(f"zones_colours must be {unvisited.value} (Cyan) or {visited.value} (Yellow), got {bad}")
        self._hard = {"visited0": sum(1 << i for i, c in enumerate(self.zone_colours) if zone(c) == visited),
                      "zones_locations": config.get("zones_locations", []),
                      "robot_locations": config.get("robot_locations", []),
                      "robot_rot": config.get("robot_rot", None)}
        if config.get("zones_num", config["num_cities"]) != config["num_cities"]:
            raise ValueError("zones_num must equal num_cities")
        super().__init__(config, **kw)

    def _native_overrides(self):
        return dict(self._hard)


class TimedTSPEnv(TSPEnv):
    """TimedTSP (main/envs/TTSP_env.py)."""
    _TASK = nat.TASK_TIMED_TSP

    def __init__(self, config, beta_a=3, beta_b=1.5, **kw):
        self.beta_a, self.beta_b = beta_a, beta_b
        super().__init__(config, **kw)
        self.max_steps = self.num_steps

    def _native_overrides(self):
        return {"beta_a": float(self.beta_a), "beta_b": float(self.beta_b)}

    @property
    def zone_times(self):
        return self._vec.get(nat.F_ZONE_OBS)[0][:, 6].astype(np.float64)


class TSPNextCityEnv(TSPEnv):
    """PointTSP-v3 (main/envs/zone_envs/TSP_next_city_env.py:11-109): dense reward towards a goal city that the
    caller chooses with set_goal() whenever info['need_next_goal'] is set."""
    goal_dim = 2

    def __init__(self, config, **kw):
        super().__init__(config, **kw)
        self._vec.enable_goals()
        self.goal_zone = None

    def reset(self):
        obs = super().reset()
        self.goal_zone = None
        return obs

    def step(self, action):
        assert self.goal_zone is not None                                   # :54
        obs, reward, done, info = super().step(action)
        shaped, need, _, goal = self._vec.goal_info()
        info["shaped_reward"] = float(shaped[0])                           # :60-66
        info["need_next_goal"] = bool(need[0])                             # :69-75
        self.goal_zone = None if need[0] else int(goal[0])
        return obs, reward, done, info

    def set_goal(self, next_goal):
        assert self.zones[next_goal] == unvisited                           # :86
        self._vec.set_goals(np.array([next_goal], np.int32))
        self.goal_zone = int(next_goal)

    def get_goal(self):
        assert self.goal_zone is not None                                   # :90-91
        return self._vec.get(nat.F_ZONE_OBS)[0][self.goal_zone, :2].astype(np.float64)

    def get_available_goals(self):
        assert self.goal_zone is None                                       # :93-100
        mask = int(self._vec.get(nat.F_AVAILABLE_GOALS)[0])
        return np.array([(mask >> i) & 1 for i in range(self.num_cities)], bool)


class TSPHardNextCityEnv(TSPHardEnv, TSPNextCityEnv):
    """PointTSP-v4 / -v5 as the zone-goals tree registers them (zone-goals/envs/TSP_hard_env.py:11-34: TSPHardEnv derives
    from TSPNextCityEnv there): the hard instances -- fixed placements, cities that start visited -- with the
    goal-conditioned surface (set_goal / get_goal / get_available_goals, info['shaped_reward'], info['need_next_goal']).
    The cities that start visited are not available goals."""

    def __init__(self, config, **kw):
        TSPHardEnv.__init__(self, config, **kw)       # -> ... -> TSPNextCityEnv.__init__ -> TSPEnv.__init__ (MRO)


class TSPOrderEnv(TSPEnv):
    """PointTSP-v2 (main/envs/TSP_order_env.py:13-113): the observation carries the visiting order of the cities
    (7th row feature 0.5^i for the i-th city of the remaining route) and info['shaped_reward'] is the progress
    towards the next city of the route.  The reference gets the route from OR-tools (:49-50, not available
    here); this class uses the library's own solution of the same problem (PATH_CHEAPEST_ARC + local search), or ``route_fn(robot_xyrot, zone_xy)
    -> rank[Z]`` when the caller brings a solver.

    As in the reference, reset() builds the first observation BEFORE generate_route() (:108-113): an episode's first
    obs carries the order feature of the route this object was left with (all zeros before the first episode and after
    a finished one, the unvisited rest of the previous route after a time-limit end); from the first step on the
    feature follows the new route, and shaped_reward uses the new route from the start.
    ``fresh_route_in_first_obs=True`` opts out: the first obs already shows the new episode's route."""

    def __init__(self, config, route_fn=None, fresh_route_in_first_obs=False, **kw):
        self._route_fn = route_fn
        self._fresh_first_obs = bool(fresh_route_in_first_obs)
        super().__init__(config, **kw)
        self._vec.enable_order(fresh_route_in_first_obs=self._fresh_first_obs)

    def build_observation_space(self):
        super().build_observation_space()
        for i in range(self.num_cities):                                   # :31-35: 6 + the order feature
            self.obs_space_dict[f"zones_lidar_{i}"] = Box(-np.inf, np.inf, (7,), dtype=np.float32)
        self.observation_space = Dict(self.obs_space_dict)

    def reset(self):
        if self._seed is None:
            self.seed(None)
        s = int(self._seed)
        from ..vec_env import sample_layout
        robot, zxy, _, _ = sample_layout(self._cfg, s)
        rank = None if self._route_fn is None else np.asarray(self._route_fn(robot, zxy), np.int32)
        self._vec.set_bank(robot[None], zxy[None], aux=None if rank is None else rank[None], seeds=[s])
        self._vec.schedule_sequential()
        self._vec.reset()
        self._seed = s + 1
        self.done = False
        self.steps = 0
        return self.obs()

    def step(self, action):
        obs, reward, done, info = super().step(action)
        info["shaped_reward"] = float(self._vec.get(nat.F_SHAPED_REWARD)[0])   # :78
        return obs, reward, done, info

    def obs(self):
        out = super().obs()
        val = self._vec.get(nat.F_ORDER_VAL)[0].astype(np.float64)
        for i in range(self.num_cities):
            out[f"zones_lidar_{i}"] = np.concatenate((out[f"zones_lidar_{i}"], val[i:i + 1]))
        return out

    @property
    def route(self):
        """Indices of the cities still to visit, in order (TSP_order_env.py: self.route)."""
        pos = self._vec.get(nat.F_ORDER_POS)[0]
        return sorted((i for i in range(self.num_cities) if pos[i] >= 0), key=lambda i: pos[i])


class TimedTSPNextCityEnv(TSPNextCityEnv):
    """zone-goals/envs/TTSP_next_city_env.py:14-51: TSPNextCityEnv with per-city deadlines."""
    _TASK = nat.TASK_TIMED_TSP

    def __init__(self, config, beta_a=3, beta_b=1.5, **kw):
        self.beta_a, self.beta_b = beta_a, beta_b
        super().__init__(config, **kw)
        self.max_steps = self.num_steps

    def _native_overrides(self):
        return {"beta_a": float(self.beta_a), "beta_b": float(self.beta_b)}


class ColourMatchEnv(ZoneEnvBase):
    """ColourMatch (main/envs/colour_match_env.py)."""
    _TASK = nat.TASK_COLOUR_MATCH
    max_cd = 150

    @property
    def zones(self):
        st = self._vec.debug_state()["zone_state"][0]
        return [colours[int(c)] for c in st]

    @property
    def zone_cooldowns(self):
        return [int(c) for c in self._vec.debug_state()["cooldown"][0]]

    @property
    def goal_dist(self):
        return int(self._vec.get(nat.F_VISIT_COUNT)[0])


class ColourMatchNextCityEnv(ColourMatchEnv):
    """zone-goals/envs/colour_match_next_city_env.py: ColourMatch with a goal zone chosen by the caller; any zone
    may be a goal, and cycling a zone other than the goal costs 1 of shaped reward."""
    goal_dim = 2

    def __init__(self, config, **kw):
        super().__init__(config, **kw)
        self._vec.enable_goals()
        self.goal_zone = None

    def reset(self):
        obs = super().reset()
        self.goal_zone = None
        return obs

    def step(self, action):
        assert self.goal_zone is not None
        obs, reward, done, info = super().step(action)
        shaped, need, _, goal = self._vec.goal_info()
        info["shaped_reward"] = float(shaped[0])
        info["need_next_goal"] = bool(need[0])
        self.goal_zone = None if need[0] else int(goal[0])
        return obs, reward, done, info

    def set_goal(self, next_goal):
        assert 0 <= next_goal < self.num_cities
        self._vec.set_goals(np.array([next_goal], np.int32))
        self.goal_zone = int(next_goal)

    def get_goal(self):
        assert self.goal_zone is not None
        return self._vec.get(nat.F_ZONE_OBS)[0][self.goal_zone, :2].astype(np.float64)

    def get_available_goals(self):
        assert self.goal_zone is None
        return np.ones(self.num_cities, dtype=bool)


class ColourMatchSolverEnv(ColourMatchNextCityEnv):
    """ColourMatch-v2 (zone-goals/envs/colour_match_solver_env.py:11-220): ColourMatchNextCityEnv plus the scripted
    high-level policy ``solver_get_next_goal`` -- the nearest zone that a cheapest recolouring plan has to cycle."""

    def solver_get_next_goal(self):
        return int(self._vec.solver_goals()[0])                            # :57-97


class TSPOrderTestEnv(TSPOrderEnv):
    """PointTSP-v21 (zone-goals/envs/TSP_order_test_env.py:11-104): TSPOrderEnv's observation (route feature) without
    the shaped reward in info."""

    def step(self, action):
        obs, reward, done, info = super().step(action)
        info.pop("shaped_reward", None)                                    # :74: plain super().step()
        return obs, reward, done, info
