"""main/envs/wrappers.py: FixedSeedsWrapper, ZoneWrapper, WaitWrapper (same names, arguments
and behaviour).  They wrap any object with the gym.Env surface of zone_envs.py; nothing here
depends on gym."""
import numpy as np

from .spaces import Box, Dict


class Wrapper:
    """The slice of gym.Wrapper the reference relies on."""

    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def step(self, action):
        return self.env.step(action)

    def reset(self):
        return self.env.reset()

    def close(self):
        return self.env.close()


class FixedSeedsWrapper(Wrapper):
    """wrappers.py:10-23: every episode draws its map seed from [min_seed, max_seed]."""

    def __init__(self, env, min_seed, max_seed, rng_seed=0):
        super().__init__(env)
        self.min_seed = min_seed
        self.max_seed = max_seed
        self.rng_seed = rng_seed
        self.rng = np.random.default_rng(seed=rng_seed)

    def reset(self):
        new_seed = self.rng.integers(low=self.min_seed, high=self.max_seed + 1, size=1)[0]
        self.env.seed(new_seed)
        return self.env.reset()


class WaitWrapper(Wrapper):
    """wrappers.py:29-54: stepping a finished env is a no-op (zero obs, zero reward, done)."""

    def __init__(self, env):
        super().__init__(env)
        self.inner_done = False

    def step(self, action):
        if not self.inner_done:
            obs, rew, done, info = self.env.step(action)
            if done:
                self.inner_done = True
        else:
            obs, rew, done, info = self.noop_obs(), 0, True, {}
        return obs, rew, done, info

    def noop_obs(self):
        return {"zone_obs": np.zeros(self.observation_space.spaces["zone_obs"].shape),
                "obs": np.zeros(self.observation_space.spaces["obs"].shape)}

    def reset(self):
        self.inner_done = False
        return self.env.reset()


class ZoneWrapper(Wrapper):
    """wrappers.py:125-156: raw dict -> {'zone_obs': (Z,F), 'obs': (8,)} in key order."""

    def __init__(self, env):
        super().__init__(env)
        self.observation_space = self.split_zone_obs_space()

    def step(self, action):
        obs, rew, done, info = self.env.step(action)
        return self.split_zone_obs(obs), rew, done, info

    @staticmethod
    def split_zone_obs(obs):
        zone_obs = np.stack([np.asarray(obs[k]).flatten() for k in obs.keys() if "zones_lidar" in k])
        rest = np.concatenate([np.asarray(obs[k]).flatten() for k in obs.keys() if "zones_lidar" not in k])
        return {"zone_obs": zone_obs, "obs": rest}

    def split_zone_obs_space(self):
        spaces = self.env.observation_space.spaces
        n_zone = [s.shape for k, s in spaces.items() if "zones_lidar" in k]
        n_rest = sum(int(np.prod(s.shape)) for k, s in spaces.items() if "zones_lidar" not in k)
        return Dict({
            "zone_obs": Box(low=-np.inf, high=np.inf, shape=(len(n_zone), n_zone[0][0])),
            "obs": Box(low=-np.inf, high=np.inf, shape=(n_rest,)),
        })

    def reset(self):
        return self.split_zone_obs(self.env.reset())
