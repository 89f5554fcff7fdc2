"""The gym registry of main/envs/__init__.py:7-141 for the ids on the MI355X hot path."""
from .zone_envs import (ColourMatchEnv, ColourMatchNextCityEnv, ColourMatchSolverEnv, TimedTSPEnv,
                        TimedTSPNextCityEnv, TSPEnv, TSPHardEnv, TSPHardNextCityEnv, TSPNextCityEnv, TSPOrderEnv,
                        TSPOrderTestEnv)

config_point = {                      # __init__.py:7-14
    "robot_base": "xmls/point.xml", "num_cities": 15, "walled": False,
    "observe_remaining": True, "observation_flatten": False, "num_steps": 2000,
}
config_point_easy = {                 # __init__.py:16-23
    "robot_base": "xmls/point.xml", "num_cities": 5, "walled": False,
    "observe_remaining": True, "observation_flatten": False, "num_steps": 1000,
}
config_point_colour = {               # __init__.py:43-50
    "robot_base": "xmls/point.xml", "num_cities": 6, "walled": False,
    "observe_remaining": True, "observation_flatten": False, "num_steps": 2000,
}

zone_locations_1 = [(-2.6, -1.6), (-0., -0.5), (1., 0.5), (1.8, 1.5), (2.6, 2.6)]          # __init__.py:52
config_zone_fixed_1 = {               # __init__.py:54-66
    "robot_base": "xmls/point.xml", "num_cities": 15, "walled": False,
    "observe_remaining": True, "observation_flatten": False, "num_steps": 1000, "zones_num": 15,
    "zones_locations": zone_locations_1, "zones_colours": [6] * 5 + [5] * 10,
    "robot_locations": [(-0.9, -0.9)], "robot_rot": -1,
}
zone_locations_2 = [(-2.6, -2.6), (-2, -1.6), (2, 1)]                                       # __init__.py:68-69
config_zone_fixed_2 = {               # __init__.py:70-81
    "robot_base": "xmls/point.xml", "num_cities": 15, "walled": False,
    "observe_remaining": True, "observation_flatten": False, "num_steps": 250, "zones_num": 15,
    "zones_locations": zone_locations_2, "zones_colours": [6] * 3 + [5] * 12,
    "robot_locations": [(0.8, 0.8)],
}

REGISTRY = {
    "PointTSP-v0": (TSPEnv, config_point),                 # __init__.py:88-90
    "PointTSP-v1": (TSPEnv, config_point_easy),            # :94-96
    "PointTTSP-v0": (TimedTSPEnv, config_point),           # :127-129
    "PointTTSP-v1": (TimedTSPEnv, config_point_easy),      # :131-133
    "ColourMatch-v0": (ColourMatchEnv, config_point_colour),   # :136-138
    "PointTSP-v4": (TSPHardEnv, config_zone_fixed_1),      # :109-111 hard instance 1
    "PointTSP-v5": (TSPHardEnv, config_zone_fixed_2),      # :114-116 hard instance 2
    "PointTSP-v2": (TSPOrderEnv, config_point),            # :98-100 solver-ordered (own tour instead of OR-tools)
    "PointTSP-v3": (TSPNextCityEnv, config_point),         # :104-106 goal-conditioned
    "PointTTSP-v3": (TimedTSPNextCityEnv, config_point),   # zone-goals/envs/__init__.py:140-142
    "ColourMatch-v3": (ColourMatchNextCityEnv, config_point_colour),   # zone-goals/envs/__init__.py:151-153
    "ColourMatch-v2": (ColourMatchSolverEnv, config_point_colour),     # zone-goals/envs/__init__.py:148-150
    "PointTSP-v21": (TSPOrderTestEnv, config_point),                   # zone-goals/envs/__init__.py:102-104
}

# Ids that the reference's trees register on DIFFERENT classes.  REGISTRY above follows main/envs/__init__.py; the
# zone-goals tree (zone-goals/envs/__init__.py:112-119, TSP_hard_env.py:11) builds PointTSP-v4 / -v5 on the
# goal-conditioned TSPNextCityEnv: `make(id, tree="zone-goals")` selects that registration.
TREE_OVERRIDES = {
    "zone-goals": {
        "PointTSP-v4": (TSPHardNextCityEnv, config_zone_fixed_1),
        "PointTSP-v5": (TSPHardNextCityEnv, config_zone_fixed_2),
    },
}

# registered by the reference but outside this build (other robots)
OUT_OF_SCOPE = ("CarTSP-v0", "DoggoTSP-v0")


def make(env_id, tree="main", **kwargs):
    """gym.make for the registered zone envs.  tree: which of the reference's source trees' registrations to follow
    where they differ ("main", or "zone-goals" for the goal-conditioned hard instances)."""
    if tree != "main":
        if tree not in TREE_OVERRIDES:
            raise ValueError(f"unknown tree {tree!r}")
        if env_id in TREE_OVERRIDES[tree]:
            cls, config = TREE_OVERRIDES[tree][env_id]
            return cls(config, **kwargs)
    if env_id in REGISTRY:
        cls, config = REGISTRY[env_id]
        return cls(config, **kwargs)
    if env_id in OUT_OF_SCOPE:
        raise NotImplementedError(f"{env_id} is registered by the reference but outside the MI355X hot path")
    raise RuntimeError("Unknown environment")
