"""Shared helpers for the parity tests: config mirroring and lock-step drivers."""
import numpy as np

TASK_IDS = {0: "PointTSP-v0", 1: "PointTTSP-v0", 2: "ColourMatch-v0"}
_SHARED_FIELDS = ["task", "num_zones", "num_steps", "max_cd", "frameskip", "zones_size",
                  "zones_keepout", "robot_keepout", "extent", "placements_margin",
                  "time_saved_reward", "beta_a", "beta_b", "timestep", "mass", "com_x",
                  "inertia_zz", "gear", "forcerange", "vel_kv", "reward_exception", "n_zones_locations",
                  "n_robot_locations", "robot_rot_fixed", "visited0", "robot_rot"]


def oracle_config_from(O, zcfg):
    """Oracle config with every shared field copied from the product's zenv_config."""
    ocfg = O.default_config(zcfg.task, zcfg.num_zones)
    for f in _SHARED_FIELDS:
        setattr(ocfg, f, getattr(zcfg, f))
    for i in range(3):
        ocfg.damping[i] = zcfg.damping[i]
    for i in range(2):
        ocfg.robot_location[i] = zcfg.robot_location[i]
    for z in range(32):
        for i in range(2):
            ocfg.zones_locations[z][i] = zcfg.zones_locations[z][i]
    return ocfg


def configs_equal(O, zcfg):
    """The two independently written default tables must agree bit for bit."""
    ocfg = O.default_config(zcfg.task, zcfg.num_zones)
    bad = [f for f in _SHARED_FIELDS if f not in ("num_steps",) and getattr(ocfg, f) != getattr(zcfg, f)]
    bad += [f"damping[{i}]" for i in range(3) if ocfg.damping[i] != zcfg.damping[i]]
    return bad


class OracleBatch:
    """N oracle envs driven in lock step with the device batch (auto-reset like penv.py:7-11)."""

    def __init__(self, O, ocfg, seeds):
        self.O = O
        self.cfg = ocfg
        self.envs = [O.OracleEnv(ocfg) for _ in seeds]
        self.seeds = list(seeds)
        self.Z = ocfg.num_zones
        self.F = self.envs[0].F

    def reset(self):
        for e, s in zip(self.envs, self.seeds):
            e.reset(s)
        return self.obs()

    def obs(self):
        n = len(self.envs)
        o = np.empty((n, 8), np.float32)
        zo = np.empty((n, self.Z, self.F), np.float32)
        for i, e in enumerate(self.envs):
            o[i], zo[i] = e.obs()
        return o, zo

    def policy(self, policy, o, zo, step_index, env_index0=0, policy_seed=0x5EED):
        a = np.empty((len(self.envs), 2), np.float32)
        for i, e in enumerate(self.envs):
            a[i] = e.policy(policy, o[i], zo[i], env_index0 + i, step_index, policy_seed)
        return a

    def step(self, actions, auto_reset=True):
        n = len(self.envs)
        r = np.zeros(n, np.float64)
        d = np.zeros(n, bool)
        g = np.zeros(n, bool)
        for i, e in enumerate(self.envs):
            if e.e.done:
                # finished earlier under step_no_reset: WaitWrapper's no-op (zero obs, reward 0, done, wrappers.py:34-45);
                # under `step` the worker then resets it like after any done step (penv.py:8-11)
                d[i] = True
                if auto_reset:
                    e.reset(self.seeds[i])
                continue
            r[i], d[i], g[i] = e.step(actions[i])
            if d[i] and auto_reset:
                e.reset(self.seeds[i])
        return r, d, g

    def state(self):
        q = np.array([list(e.e.qpos) for e in self.envs])
        v = np.array([list(e.e.qvel) for e in self.envs])
        steps = np.array([e.e.steps for e in self.envs], np.int32)
        return q, v, steps
