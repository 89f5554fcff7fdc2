"""Host-side mirror of the reference's ``envs`` package (main/envs/) for the hot path."""
from .make_env import make_fixed_env, make_test_env, make_train_env
from .registry import REGISTRY, config_point, config_point_colour, config_point_easy, make
from .wrappers import FixedSeedsWrapper, WaitWrapper, ZoneWrapper
from .zone_envs import (ColourMatchEnv, ColourMatchNextCityEnv, TimedTSPEnv, TimedTSPNextCityEnv, TSPEnv, TSPHardEnv,
                        TSPHardNextCityEnv, TSPNextCityEnv, TSPOrderEnv, ZoneEnvBase, zone)
