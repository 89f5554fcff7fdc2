"""The gym registry of main/envs/__init__.py:7-141 for the ids on the MI355X hot path."""
from .zone_envs import (ColourMatchEnv, ColourMatchNextCityEnv, TimedTSPEnv, TimedTSPNextCityEnv, TSPEnv,
                        TSPNextCityEnv, TSPOrderEnv)

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

REGISTRY = {
    "PointTSP-v0": (TSPEnv, config_point),                 # __init__.py:88-90
    "PointTSP-v1": (TSPEnv, config_point_easy),            # :94-96
    "PointTTSP-v0": (TimedTSPEnv, config_point),           # :127-129
    "PointTTSP-v1": (TimedTSPEnv, config_point_easy),      # :131-133
    "ColourMatch-v0": (ColourMatchEnv, config_point_colour),   # :136-138
    "PointTSP-v2": (TSPOrderEnv, config_point),            # :98-100 solver-ordered (own tour instead of OR-tools)
    "PointTSP-v3": (TSPNextCityEnv, config_point),         # :104-106 goal-conditioned
    "PointTTSP-v3": (TimedTSPNextCityEnv, config_point),   # zone-goals/envs/__init__.py:140-142
    "ColourMatch-v3": (ColourMatchNextCityEnv, config_point_colour),   # zone-goals/envs/__init__.py:151-153
}

# registered by the reference but outside this build (other robots, solver/goal variants)
OUT_OF_SCOPE = ("PointTSP-v4", "PointTSP-v5", "CarTSP-v0", "DoggoTSP-v0")


def make(env_id, **kwargs):
    """gym.make for the registered zone envs."""
    if env_id in REGISTRY:
        cls, config = REGISTRY[env_id]
        return cls(config, **kwargs)
    if env_id in OUT_OF_SCOPE:
        raise NotImplementedError(f"{env_id} is registered by the reference but outside the MI355X hot path")
    raise RuntimeError("Unknown environment")
