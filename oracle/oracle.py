"""ctypes binding of the CPU ORACLE (oracle/zenv_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED for the dynamics half (the reference's
MuJoCo/Safety-Gym dependencies are absent, see zenv_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
shipped package (combinatorial-rl-tasks_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "zenv_oracle.c")
_HDR = os.path.join(_HERE, "zenv_oracle.h")
_OUT_DIR = os.path.join(_HERE, "build")
_SO = os.path.join(_OUT_DIR, "libzenv_oracle.so")

MAX_Z = 32
TASK_TSP, TASK_TIMED, TASK_COLOUR = 0, 1, 2
POLICY_UNIFORM, POLICY_GREEDY = 0, 1


def build(force=False):
    """gcc -O2 -ffp-contract=off (bit-reproducible IEEE double; OpenMP for the batch driver).
    -mfma only inlines the explicit __builtin_fma calls on hosts that have the instruction; without it
    they go to libm's fma(), which is correctly rounded too, so the results do not depend on the flag."""
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(_SRC), os.path.getmtime(_HDR))):
        return _SO
    os.makedirs(_OUT_DIR, exist_ok=True)
    tmp = _SO + f".tmp{os.getpid()}"       # concurrent builders (ranks of one job) each rename a complete file in
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared",
           "-Wall", "-Wextra", "-o", tmp, _SRC, "-lm"]
    try:
        with open("/proc/cpuinfo") as f:
            if " fma " in f.read().replace("\n", " "):
                cmd.insert(1, "-mfma")
    except OSError:
        pass
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, _SO)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return _SO


class Config(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("num_zones", C.c_int32), ("num_steps", C.c_int32),
        ("max_cd", C.c_int32), ("frameskip", C.c_int32), ("pad0", C.c_int32),
        ("zones_size", C.c_double), ("zones_keepout", C.c_double), ("robot_keepout", C.c_double),
        ("extent", C.c_double), ("placements_margin", C.c_double),
        ("time_saved_reward", C.c_double), ("beta_a", C.c_double), ("beta_b", C.c_double),
        ("timestep", C.c_double), ("mass", C.c_double), ("com_x", C.c_double),
        ("inertia_zz", C.c_double), ("damping", C.c_double * 3), ("gear", C.c_double),
        ("forcerange", C.c_double), ("vel_kv", C.c_double), ("reward_exception", C.c_double),
        ("n_zones_locations", C.c_int32), ("n_robot_locations", C.c_int32), ("robot_rot_fixed", C.c_int32),
        ("visited0", C.c_uint32), ("robot_rot", C.c_double), ("robot_location", C.c_double * 2),
        ("zones_locations", (C.c_double * 2) * MAX_Z),
    ]


class Env(C.Structure):
    _fields_ = [
        ("cfg", Config), ("seed", C.c_int64),
        ("x0", C.c_double), ("y0", C.c_double), ("rot", C.c_double),
        ("bq0", C.c_double), ("bq3", C.c_double),
        ("zone_xy", (C.c_double * 2) * MAX_Z), ("tmax", C.c_int32 * MAX_Z),
        ("qpos", C.c_double * 3), ("qvel", C.c_double * 3),
        ("xpos", C.c_double * 2), ("xvelp", C.c_double * 2), ("xvelr", C.c_double),
        ("xquat0", C.c_double), ("xquat3", C.c_double),
        ("visited", C.c_int32 * MAX_Z), ("colour", C.c_int32 * MAX_Z),
        ("cooldown", C.c_int32 * MAX_Z),
        ("goal_dist", C.c_int32), ("steps", C.c_int32), ("done", C.c_int32),
        ("layout_restarts", C.c_int32),
        ("goal_zone", C.c_int32), ("last_visit", C.c_int32), ("last_dist", C.c_double),
        ("route", C.c_int32 * MAX_Z), ("route_len", C.c_int32), ("exception", C.c_int32),
        ("obs_route", C.c_int32 * MAX_Z), ("obs_route_len", C.c_int32), ("pad_order", C.c_int32),
    ]


class RS(C.Structure):
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int), ("has_gauss", C.c_int),
                ("gauss", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_rs_seed.argtypes = [C.POINTER(RS), C.c_uint32]
        L.orc_rs_u32.argtypes = [C.POINTER(RS)]
        L.orc_rs_u32.restype = C.c_uint32
        for f in (L.orc_rs_double,):
            f.argtypes = [C.POINTER(RS)]
            f.restype = C.c_double
        L.orc_rs_uniform.argtypes = [C.POINTER(RS), C.c_double, C.c_double]
        L.orc_rs_uniform.restype = C.c_double
        L.orc_rs_choice.argtypes = [C.POINTER(RS), C.c_int64]
        L.orc_rs_choice.restype = C.c_int64
        L.orc_rs_beta.argtypes = [C.POINTER(RS), C.c_double, C.c_double]
        L.orc_rs_beta.restype = C.c_double
        L.orc_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_default_config.argtypes = [C.c_int, C.c_int, C.POINTER(Config)]
        L.orc_reset.argtypes = [C.POINTER(Env), C.POINTER(Config), C.c_int64]
        L.orc_reset.restype = C.c_int
        L.orc_step.argtypes = [C.POINTER(Env), C.POINTER(C.c_float), C.POINTER(C.c_double),
                               C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_step.restype = C.c_int
        L.orc_obs.argtypes = [C.POINTER(Env), C.c_void_p, C.c_void_p]
        L.orc_zone_feat.argtypes = [C.POINTER(Config)]
        L.orc_zone_feat.restype = C.c_int
        L.orc_policy.argtypes = [C.c_int, C.POINTER(Config), C.c_void_p, C.c_void_p,
                                 C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_float)]
        L.orc_rollout.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_int64, C.c_uint64, C.c_uint64, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
        L.orc_rollout.restype = C.c_int64
        L.orc_set_goal.argtypes = [C.POINTER(Env), C.c_int]
        L.orc_solver_next_goal.argtypes = [C.POINTER(Env)]
        L.orc_reset_order.argtypes = [C.POINTER(Env), C.POINTER(Config), C.c_int64, C.c_void_p, C.c_int]
        L.orc_order_route.argtypes = [C.POINTER(Env), C.c_void_p]
        L.orc_step_order.argtypes = [C.POINTER(Env), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_order_vals.argtypes = [C.POINTER(Env), C.c_void_p]
        L.orc_step_goal.argtypes = [C.POINTER(Env), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_rollout_wrapped.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_int64, C.c_int32, C.c_uint64, C.c_uint64, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        L.orc_rollout_wrapped.restype = C.c_int64
        _lib = L
    return _lib


def default_config(task, num_zones, **overrides):
    cfg = Config()
    lib().orc_default_config(task, num_zones, C.byref(cfg))
    for k, v in overrides.items():
        if k == "damping":
            for i in range(3):
                cfg.damping[i] = v[i]
        else:
            setattr(cfg, k, v)
    return cfg


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


class RandomState:
    """The oracle's numpy-legacy RandomState restatement (for pinning against numpy)."""

    def __init__(self, seed):
        self._rs = RS()
        lib().orc_rs_seed(C.byref(self._rs), int(seed) & 0xFFFFFFFF)

    def u32(self):
        return lib().orc_rs_u32(C.byref(self._rs))

    def random_sample(self):
        return lib().orc_rs_double(C.byref(self._rs))

    def uniform(self, lo, hi):
        return lib().orc_rs_uniform(C.byref(self._rs), lo, hi)

    def choice(self, n):
        return lib().orc_rs_choice(C.byref(self._rs), n)

    def beta(self, a, b):
        return lib().orc_rs_beta(C.byref(self._rs), a, b)


class OracleEnv:
    """One env instance of the oracle: reset(seed) / step(action) / obs()."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.e = Env()
        self.Z = cfg.num_zones
        self.F = lib().orc_zone_feat(C.byref(cfg))

    def reset(self, seed):
        rc = lib().orc_reset(C.byref(self.e), C.byref(self.cfg), int(seed))
        if rc != 0:
            raise RuntimeError(f"orc_reset failed rc={rc}")
        return self.obs()

    def step(self, action):
        a = (C.c_float * 2)(float(action[0]), float(action[1]))
        r, d, g = C.c_double(), C.c_int(), C.c_int()
        rc = lib().orc_step(C.byref(self.e), a, C.byref(r), C.byref(d), C.byref(g))
        if rc != 0:
            raise AssertionError("Environment must be reset before stepping")
        return r.value, bool(d.value), bool(g.value)

    # goal-conditioned variant (TSP_next_city_env.py)
    def set_goal(self, goal):
        if lib().orc_set_goal(C.byref(self.e), int(goal)) != 0:
            raise AssertionError("goal zone must be unvisited")

    def solver_next_goal(self):
        return lib().orc_solver_next_goal(C.byref(self.e))

    def step_goal(self, action):
        a = (C.c_float * 2)(float(action[0]), float(action[1]))
        r, d, g, sh, nd = C.c_double(), C.c_int(), C.c_int(), C.c_double(), C.c_int()
        rc = lib().orc_step_goal(C.byref(self.e), a, C.byref(r), C.byref(d), C.byref(g), C.byref(sh), C.byref(nd))
        if rc != 0:
            raise AssertionError("no goal set" if rc == -2 else "Environment must be reset before stepping")
        return r.value, bool(d.value), bool(g.value), sh.value, bool(nd.value)

    # solver-ordered variant (TSP_order_env.py)
    def reset_order(self, seed, rank, fresh_first_obs=False):
        """TSPOrderEnv.reset() (TSP_order_env.py:108-113) for env.seed(seed): the first observation carries the order
        feature of the route this object was left with (the reference builds it before generate_route()); `rank` is
        what generate_route() then produces."""
        r = np.ascontiguousarray(rank, np.int32)
        rc = lib().orc_reset_order(C.byref(self.e), C.byref(self.cfg), int(seed), r.ctypes.data, int(bool(fresh_first_obs)))
        if rc == -3:
            raise RuntimeError("ResamplingError: Failed to sample layout of objects")
        if rc != 0:
            raise ValueError("rank out of range")
        return self.obs()

    @property
    def route(self):
        r = np.empty(self.Z, np.int32)
        n = lib().orc_order_route(C.byref(self.e), r.ctypes.data)
        return [int(v) for v in r[:n]]

    def step_order(self, action):
        a = (C.c_float * 2)(float(action[0]), float(action[1]))
        r, d, g, sh = C.c_double(), C.c_int(), C.c_int(), C.c_double()
        if lib().orc_step_order(C.byref(self.e), a, C.byref(r), C.byref(d), C.byref(g), C.byref(sh)) != 0:
            raise AssertionError("Environment must be reset before stepping")
        return r.value, bool(d.value), bool(g.value), sh.value

    def order_vals(self):
        v = np.empty(self.Z, np.float32)
        lib().orc_order_vals(C.byref(self.e), v.ctypes.data)
        return v

    def available_goals(self):
        if self.cfg.task == TASK_COLOUR:
            return np.ones(self.Z, bool)
        return np.array([not self.e.visited[z] for z in range(self.Z)], bool)

    def obs(self):
        o = np.empty(8, np.float32)
        zo = np.empty((self.Z, self.F), np.float32)
        lib().orc_obs(C.byref(self.e), o.ctypes.data, zo.ctypes.data)
        return o, zo

    def policy(self, policy, obs, zone_obs, env_index, step_index, policy_seed=0x5EED):
        a = (C.c_float * 2)()
        o = np.ascontiguousarray(obs, np.float32)
        zo = np.ascontiguousarray(zone_obs, np.float32)
        lib().orc_policy(policy, C.byref(self.cfg), o.ctypes.data, zo.ctypes.data,
                         env_index, step_index, policy_seed, a)
        return np.array([a[0], a[1]], np.float32)

    # convenient views of the state
    @property
    def layout(self):
        e = self.e
        zone_xy = np.array([[e.zone_xy[z][0], e.zone_xy[z][1]] for z in range(self.Z)])
        return np.array([e.x0, e.y0, e.rot]), zone_xy

    def state(self):
        e = self.e
        return dict(
            qpos=np.array(e.qpos[:]), qvel=np.array(e.qvel[:]), xpos=np.array(e.xpos[:]),
            visited=np.array(e.visited[:self.Z]), colour=np.array(e.colour[:self.Z]),
            cooldown=np.array(e.cooldown[:self.Z]), tmax=np.array(e.tmax[:self.Z]),
            goal_dist=e.goal_dist, steps=e.steps, done=e.done)


def rollout(cfg, seeds0, n_steps, policy, seed_stride=0, policy_seed=0x5EED, env_index0=0,
            n_threads=1, seed_period=0):
    """Batch driver.  Returns dict of per-env arrays + total env-steps.  seed_period > 0: episode k
    of env i replays seed seeds0[i] + (k % seed_period) * seed_stride (a wrapping map bank)."""
    seeds0 = np.ascontiguousarray(seeds0, np.int64)
    n = len(seeds0)
    Z, F = cfg.num_zones, lib().orc_zone_feat(C.byref(cfg))
    out = dict(
        reward_sum=np.zeros(n, np.float64), episodes=np.zeros(n, np.int32),
        last_return=np.zeros(n, np.float64), last_len=np.zeros(n, np.int32),
        obs=np.zeros((n, 8), np.float32), zone_obs=np.zeros((n, Z, F), np.float32))
    total = lib().orc_rollout_wrapped(
        C.byref(cfg), n, int(n_steps), int(policy), seeds0.ctypes.data, int(seed_stride), int(seed_period),
        int(policy_seed), int(env_index0), int(n_threads),
        out["reward_sum"].ctypes.data, out["episodes"].ctypes.data,
        out["last_return"].ctypes.data, out["last_len"].ctypes.data,
        out["obs"].ctypes.data, out["zone_obs"].ctypes.data)
    out["total_steps"] = int(total)
    return out
