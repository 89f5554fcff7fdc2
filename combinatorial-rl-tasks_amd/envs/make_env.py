"""main/envs/make_env.py: make_train_env / make_test_env / make_fixed_env (same signatures).

The reference keeps one id list per factory and per tree (main/envs/make_env.py:14, :32, :49; zone-goals/envs/make_env.py):
e.g. main's make_train_env does not list PointTSP-v4 / -v5.  Here every factory accepts every id of the merged
registry -- a superset: whatever main/ constructs is constructed the same way, unknown ids raise the same
RuntimeError("Unknown environment"), Car / Doggo ids raise NotImplementedError (other robots: out of scope).
One id pair is registered on different classes by different trees: PointTSP-v4 / -v5 are TSPHardEnv(TSPEnv) in main/
(TSP_hard_env.py:11) and TSPHardEnv(TSPNextCityEnv) in zone-goals/ (zone-goals/envs/TSP_hard_env.py:11); the factories
here follow main/, `registry.make(id, tree="zone-goals")` builds the goal-conditioned variant (TSPHardNextCityEnv)."""
from .registry import OUT_OF_SCOPE, REGISTRY, make
from .wrappers import FixedSeedsWrapper, WaitWrapper, ZoneWrapper

_ZONE_IDS = tuple(REGISTRY) + OUT_OF_SCOPE


def make_train_env(env_name, hier=False, num_training_tasks=100, rng_seed=0):
    if env_name not in _ZONE_IDS:
        raise RuntimeError("Unknown environment")          # make_env.py:18
    env = make(env_name)
    env = ZoneWrapper(FixedSeedsWrapper(env, min_seed=1, max_seed=num_training_tasks, rng_seed=rng_seed))
    return WaitWrapper(env) if hier else env


def make_test_env(env_name, hier=False, seed=1000):
    if env_name not in _ZONE_IDS:
        raise RuntimeError("Unknown environment")          # make_env.py:34
    env = make(env_name)
    env.seed(seed)
    return ZoneWrapper(env)


def make_fixed_env(env_name, hier=False, seed=1000, env_seed=0):
    if env_name not in _ZONE_IDS:
        raise RuntimeError("Unknown environment")          # make_env.py:51
    env = make(env_name)
    env.seed(seed)
    return ZoneWrapper(FixedSeedsWrapper(env, min_seed=env_seed, max_seed=env_seed, rng_seed=seed))
