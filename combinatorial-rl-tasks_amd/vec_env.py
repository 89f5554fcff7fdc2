"""ZoneVecEnv -- the batched, device-resident surface of the zone-env hot path.

One object = N independent PointTSP / TimedTSP / ColourMatch environments stepped by one
HIP kernel launch on one MI355X.  This is the thing that replaces the reference's
``ParallelEnv`` process pool (main/src/torch_ac/torch_utils/penv.py:26-66): ``step`` takes a
(N, 2) float32 action array and returns struct-of-arrays results; finished envs are
re-reset inside the same launch and report the next episode's first observation with the
terminal reward/done, exactly like ``worker`` (penv.py:7-11).
"""
import ctypes as C

import numpy as np

from . import _native as nat
from ._native import Config, ZenvError, check, lib

_FIELD_DTYPES = {
    nat.F_OBS: np.float32, nat.F_ZONE_OBS: np.float32, nat.F_REWARD: np.float32,
    nat.F_DONE: np.uint8, nat.F_GOAL_MET: np.uint8, nat.F_EP_RETURN: np.float64,
    nat.F_EP_LEN: np.int32, nat.F_LAST_RETURN: np.float64, nat.F_LAST_LEN: np.int32,
    nat.F_EPISODES: np.int32, nat.F_VISIT_COUNT: np.int32, nat.F_SEED: np.int64,
    nat.F_ACTIONS: np.float32, nat.F_POLICY_MU: np.float32, nat.F_POLICY_STD: np.float32,
    nat.F_POLICY_VALUE: np.float32, nat.F_SHAPED_REWARD: np.float64, nat.F_NEED_GOAL: np.uint8,
    nat.F_AVAILABLE_GOALS: np.uint32, nat.F_GOAL: np.int32, nat.F_ORDER_VAL: np.float32,
    nat.F_EXCEPTION: np.uint8, nat.F_POLICY_VALUE_SIGMA: np.float32, nat.F_ORDER_POS: np.int8,
}


def config_for_id(env_id, **overrides):
    """Config of a registry id (envs/__init__.py:88-141); unknown id -> RuntimeError."""
    cfg = Config()
    rc = lib().zenv_config_for_id(env_id.encode(), C.byref(cfg))
    if rc != 0:
        raise RuntimeError("Unknown environment")   # make_env.py:18,34,51
    return apply_overrides(cfg, overrides)


def default_config(task, num_zones, **overrides):
    cfg = Config()
    check(lib().zenv_default_config(int(task), int(num_zones), C.byref(cfg)))
    return apply_overrides(cfg, overrides)


def apply_overrides(cfg, overrides):
    for k, v in overrides.items():
        if k == "damping":
            for i in range(3):
                cfg.damping[i] = float(v[i])
        elif k == "robot_locations":              # Engine 'robot_locations': [] or [(x, y)]
            cfg.n_robot_locations = len(v)
            for i, c in enumerate(v[:1]):
                cfg.robot_location[0], cfg.robot_location[1] = float(c[0]), float(c[1])
            if len(v) > 1:
                raise ValueError("at most one robot location")
        elif k == "zones_locations":              # Engine 'zones_locations': the first len(v) zones are fixed
            if len(v) > nat.MAX_ZONES:
                raise ValueError("too many zones_locations")
            cfg.n_zones_locations = len(v)
            for i, c in enumerate(v):
                cfg.zones_locations[i][0], cfg.zones_locations[i][1] = float(c[0]), float(c[1])
        elif k == "robot_rot":                    # Engine 'robot_rot': None = random
            cfg.robot_rot_fixed = 0 if v is None else 1
            cfg.robot_rot = 0.0 if v is None else float(v)
        elif not hasattr(cfg, k):
            raise KeyError(f"unknown config key {k!r}")
        else:
            setattr(cfg, k, v)
    return cfg


def mlp_tensors_from_state_dict(sd):
    """ACModel.state_dict() (flat_model.py:24-52) -> the tensors load_mlp wants (numpy float32)."""
    names = {"zone_w1": "env_model.zone_net_.0.weight", "zone_b1": "env_model.zone_net_.0.bias",
             "zone_w2": "env_model.zone_net_.2.weight", "zone_b2": "env_model.zone_net_.2.bias",
             "zone_w3": "env_model.zone_net_.4.weight", "zone_b3": "env_model.zone_net_.4.bias",
             "comb_w": "env_model.combine_net_.weight", "comb_b": "env_model.combine_net_.bias",
             "enc_w": "actor.enc_.0.0.weight", "enc_b": "actor.enc_.0.0.bias",
             "mu_w": "actor.mu_.weight", "mu_b": "actor.mu_.bias",
             "std_w": "actor.std_.weight", "std_b": "actor.std_.bias"}
    if "critic.0.weight" in sd and "critic.2.weight" in sd:      # non-distributional critic, flat_model.py:43-47
        names.update({"critic_w1": "critic.0.weight", "critic_b1": "critic.0.bias",
                      "critic_w2": "critic.2.weight", "critic_b2": "critic.2.bias"})
    elif "critic.0.weight" in sd and "critic_mu.weight" in sd:   # distributional_value=True, flat_model.py:35-41
        names.update({"critic_w1": "critic.0.weight", "critic_b1": "critic.0.bias",
                      "critic_w2": "critic_mu.weight", "critic_b2": "critic_mu.bias",
                      "critic_sigma_w": "critic_sigma.weight", "critic_sigma_b": "critic_sigma.bias"})
    return {k: np.asarray(sd[v].detach().cpu().numpy() if hasattr(sd[v], "detach") else sd[v], np.float32)
            for k, v in names.items()}


def zone_feat(cfg):
    return lib().zenv_zone_feat(C.byref(cfg))


def comm_unique_id():
    """rank 0 of a sharded job: the RCCL unique id (128 bytes) every rank passes to ZoneVecEnv.comm_init."""
    buf = (C.c_char * nat.COMM_ID_BYTES)()
    check(lib().zenv_comm_unique_id(buf))
    return bytes(buf.raw)


def probe_store_stream(n_tiles, tile_bytes, steps=64, cache_policy=0, reps=3, device=0):
    """us per step of a bare write-only row stream of the step kernels' shape (zenv_probe_store_stream)."""
    us = C.c_float(0)
    check(lib().zenv_probe_store_stream(int(device), int(n_tiles), int(tile_bytes), int(steps), int(cache_policy),
                                        int(reps), C.byref(us)))
    return us.value


def sample_layout(cfg, seed):
    """Host half of reset() for ``env.seed(seed); env.reset()``.

    Returns (robot_xyrot[3], zone_xy[Z,2], aux[Z], restarts)."""
    Z = cfg.num_zones
    robot = np.zeros(3, np.float64)
    zones = np.zeros((Z, 2), np.float64)
    aux = np.zeros(Z, np.int32)
    restarts = C.c_int32(0)
    check(lib().zenv_sample_layout(C.byref(cfg), int(seed), robot.ctypes.data, zones.ctypes.data,
                                   aux.ctypes.data, C.byref(restarts)))
    return robot, zones, aux, restarts.value


def route_ranks(robot_xy, zone_xy):
    """The built-in visiting order of a layout (TSPOrderEnv without OR-tools): rank[z] = position of zone z in a
    tour: TSP_Solver.get_optim_route's problem (closed tour from the robot, int64(10 x distance) arcs,
    PATH_CHEAPEST_ARC + local search) solved without OR-tools."""
    r = np.ascontiguousarray(robot_xy, np.float64)[:2].copy()
    z = np.ascontiguousarray(zone_xy, np.float64)
    rank = np.zeros(z.shape[0], np.int32)
    check(lib().zenv_route_ranks(r.ctypes.data, z.ctypes.data, z.shape[0], rank.ctypes.data))
    return rank


def fixed_seed_sequence(rng_seed, min_seed, max_seed, count):
    """Seeds FixedSeedsWrapper.reset (wrappers.py:20-23) would draw, without numpy's Generator."""
    out = np.zeros(count, np.int64)
    check(lib().zenv_fixed_seed_sequence(int(rng_seed), int(min_seed), int(max_seed), int(count),
                                         out.ctypes.data))
    return out


class ZoneVecEnv:
    """N device-resident zone envs.

    Parameters
    ----------
    cfg : Config or registry id (str)
    num_envs : N
    device : HIP device ordinal
    """

    def __init__(self, cfg, num_envs, device=0, **overrides):
        if isinstance(cfg, str):
            cfg = config_for_id(cfg, **overrides)
        elif overrides:
            cfg = apply_overrides(cfg.copy(), overrides)
        self.cfg = cfg
        self.num_envs = int(num_envs)
        self.num_zones = cfg.num_zones
        self.zone_feat = zone_feat(cfg)
        self.device = int(device)
        self._h = C.c_void_p()
        check(lib().zenv_create(C.byref(cfg), self.num_envs, self.device, C.byref(self._h)))

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().zenv_destroy(self._h)
            self._h = C.c_void_p()
            # the handle's own page-locked images (copy=False views of step_results() die with the env)
            ptrs = [a.ctypes.data for a in getattr(self, "_own_pinned", [])]
            self._own_pinned, self._slab, self._slab_views, self._goal_host = [], None, None, None
            self._host_actions, self._host_io = None, False      # (zenv_host_io's buffers went with the handle)
            for ptr in ptrs:
                lib().zenv_host_free(C.c_void_p(ptr))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ bank / schedule
    def build_bank(self, seed_first, count, n_threads=8):
        check(lib().zenv_bank_build(self._h, int(seed_first), int(count), int(n_threads)))

    def build_bank_seeds(self, seeds, n_threads=8):
        s = np.ascontiguousarray(seeds, np.int64).reshape(-1)
        check(lib().zenv_bank_build_seeds(self._h, s.ctypes.data, s.size, int(n_threads)))

    def set_bank(self, robot_xyrot, zone_xy, aux=None, seeds=None):
        robot = np.ascontiguousarray(robot_xyrot, np.float64).reshape(-1, 3)
        S = robot.shape[0]
        zones = np.ascontiguousarray(zone_xy, np.float64).reshape(S, self.num_zones, 2)
        aux_a = None if aux is None else np.ascontiguousarray(aux, np.int32).reshape(S, self.num_zones)
        seeds_a = None if seeds is None else np.ascontiguousarray(seeds, np.int64).reshape(S)
        check(lib().zenv_bank_set(self._h, robot.ctypes.data, zones.ctypes.data,
                                  None if aux_a is None else aux_a.ctypes.data,
                                  None if seeds_a is None else seeds_a.ctypes.data, S))

    @property
    def bank_size(self):
        return lib().zenv_bank_size(self._h)

    def schedule_sequential(self, first=None, stride=0):
        f = None if first is None else np.ascontiguousarray(first, np.int32)
        if f is not None and f.shape != (self.num_envs,):
            raise ValueError("first must have shape (num_envs,)")
        check(lib().zenv_schedule_sequential(self._h, None if f is None else f.ctypes.data, int(stride)))

    def schedule_ring(self, first, depth):
        """Env i walks round the slots first[i] .. first[i] + depth - 1 (zenv_schedule_ring); ``update_bank`` keeps
        the ring ahead of it."""
        f = np.ascontiguousarray(first, np.int32)
        if f.shape != (self.num_envs,):
            raise ValueError("first must have shape (num_envs,)")
        check(lib().zenv_schedule_ring(self._h, f.ctypes.data, int(depth)))

    def update_bank(self, slots, seeds, n_threads=4):
        """Refill bank slots in place with the layouts of `seeds` (zenv_bank_update; stream-ordered)."""
        sl = np.ascontiguousarray(slots, np.int32).reshape(-1)
        sd = np.ascontiguousarray(seeds, np.int64).reshape(-1)
        if sl.shape != sd.shape:
            raise ValueError("one seed per slot")
        check(lib().zenv_bank_update(self._h, sl.ctypes.data, sd.ctypes.data, sl.size, int(n_threads)))

    def schedule_fixed_seeds(self, rng_seeds, min_seed, max_seed):
        s = np.ascontiguousarray(rng_seeds, np.uint64)
        if s.shape != (self.num_envs,):
            raise ValueError("rng_seeds must have shape (num_envs,)")
        check(lib().zenv_schedule_fixed_seeds(self._h, s.ctypes.data, int(min_seed), int(max_seed)))

    # ------------------------------------------------------------------ hot path
    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, np.uint8)
            if m.shape != (self.num_envs,):
                raise ValueError("mask must have shape (num_envs,)")
        check(lib().zenv_reset(self._h, None if m is None else m.ctypes.data))

    def step(self, actions=None, auto_reset=True):
        """actions: (N,2) float32 ndarray (host), or None to use the device action buffer."""
        if actions is None:
            check(lib().zenv_step(self._h, None, 0, int(bool(auto_reset))))
            return
        a = np.ascontiguousarray(actions, np.float32)
        if a.shape != (self.num_envs, 2):
            raise ValueError(f"actions must have shape ({self.num_envs}, 2)")
        check(lib().zenv_step(self._h, a.ctypes.data, 0, int(bool(auto_reset))))

    _RESET_MODES = {"never": nat.CHUNK_NO_RESET, "every": nat.CHUNK_RESET_EVERY, "last": nat.CHUNK_RESET_LAST}

    def step_many(self, actions, reset="every", actions_ptr=None):
        """An action chunk (zenv_step_many): K steps of caller-supplied actions, one launch of the persistent kernel per
        256 steps.  actions: (K, N, 2) float32 (host), or actions_ptr = (device address, K) for a device buffer.
        reset: "every" (K x step), "never" (K x step_no_reset) or "last" (K - 1 x step_no_reset, then one step: the
        fixed-length-skill loop, _hier_policy_opt.py:68-71).  Asynchronous like step(); afterwards observations() /
        results() hold the last step, chunk_results() every step's reward and done flag."""
        mode = self._RESET_MODES[reset]
        if actions_ptr is not None:
            ptr, k = actions_ptr
            check(lib().zenv_step_many(self._h, C.c_void_p(int(ptr)), 1, int(k), mode))
            return
        a = np.ascontiguousarray(actions, np.float32)
        if a.ndim != 3 or a.shape[1:] != (self.num_envs, 2) or a.shape[0] < 1:
            raise ValueError(f"actions must have shape (K, {self.num_envs}, 2)")
        check(lib().zenv_step_many(self._h, a.ctypes.data, 0, int(a.shape[0]), mode))

    def chunk_results(self):
        """(reward float32 (K,N), done bool (K,N)) of the last step_many(), time-major."""
        k = lib().zenv_field_bytes(self._h, nat.F_CHUNK_DONE) // self.num_envs
        r = np.empty((k, self.num_envs), np.float32)
        d = np.empty((k, self.num_envs), np.uint8)
        check(lib().zenv_get(self._h, nat.F_CHUNK_REWARD, r.ctypes.data, 0))
        check(lib().zenv_get(self._h, nat.F_CHUNK_DONE, d.ctypes.data, 0))
        return r, d.astype(bool)

    def step_device(self, actions_ptr, auto_reset=True):
        """actions_ptr: integer device address of a float32 [N,2] buffer (zero-copy policies)."""
        check(lib().zenv_step(self._h, C.c_void_p(int(actions_ptr)), 1, int(bool(auto_reset))))

    def policy(self, policy, policy_seed=0x5EED, env_index0=0, dst_ptr=None):
        check(lib().zenv_policy(self._h, int(policy), int(policy_seed), int(env_index0),
                                None if dst_ptr is None else C.c_void_p(int(dst_ptr))))

    def rollout(self, steps, policy, policy_seed=0x5EED, env_index0=0, auto_reset=True,
                time_step_kernel=False, fused=True, event_stride=1, mode=None, wait=True):
        """K closed-loop steps a_t = policy(obs_t, t); step(a_t) on the handle's stream.

        mode "persistent" (default): one launch advances every env by up to 256 steps, state in
        registers, all per-step outputs still written on every step; "per_step": one step-kernel
        launch per step which also emits the next action; "unfused" (or fused=False): per-step
        launches with a stand-alone policy kernel before each.  Same results in all three.

        Returns (ms_total, ms_step_kernel_avg or None), both from HIP events on that stream;
        the kernel figure is kernel time per step (see include/zenv.h)."""
        if mode is None:
            mode = "persistent" if fused else "unfused"
        flags = {"persistent": 0, "per_step": nat.ROLLOUT_PER_STEP, "unfused": nat.ROLLOUT_UNFUSED}[mode]
        if not wait:            # enqueue only (ZENV_ROLLOUT_ASYNC): collect with done() / sync(); no times
            flags |= nat.ROLLOUT_ASYNC
        total = C.c_float(0)
        kern = C.c_float(0)
        check(lib().zenv_rollout(self._h, int(steps), int(policy), int(policy_seed),
                                 int(env_index0), int(bool(auto_reset)),
                                 flags, int(event_stride), C.byref(total),
                                 C.byref(kern) if time_step_kernel else None))
        return total.value, (kern.value if time_step_kernel else None)

    def set_rollout_slice(self, envs_per_launch):
        """Envs one persistent launch covers (default 65 536; 0 = the whole batch in one launch): zenv_set_rollout_slice."""
        check(lib().zenv_set_rollout_slice(self._h, int(envs_per_launch)))

    # ------------------------------------------------------------------ solver-ordered variant (8(f) row 3)
    def enable_order(self, fresh_route_in_first_obs=False):
        """TSPOrderEnv semantics (TSP_order_env.py:13-113); call before build_bank / set_bank -- an episode's
        route is the bank's aux column (built-in PATH_CHEAPEST_ARC + local-search tour, or the caller's ranks).
        Default = the reference's reset(): the first observation of an episode is built before generate_route()
        (:108-113) and carries the order feature of the route the env was left with.  fresh_route_in_first_obs=True:
        the first observation shows the new episode's route (the build's opt-out, not reference behaviour)."""
        check(lib().zenv_order_enable(self._h))
        check(lib().zenv_order_configure(self._h, nat.ORDER_FRESH_FIRST_OBS if fresh_route_in_first_obs else 0))

    def order_info(self):
        """(shaped_reward float64 [N], order feature float32 [N,Z] of the last observation: 0.5^(position in the route
        that observation saw))."""
        return self.get(nat.F_SHAPED_REWARD), self.get(nat.F_ORDER_VAL)

    def order_routes(self):
        """self.route of every env as positions: int8 [N,Z], position of zone z in the remaining route, -1 = not in it."""
        return self.get(nat.F_ORDER_POS)

    # ------------------------------------------------------------------ goal-conditioned variant (8(f) row 3)
    def enable_goals(self):
        """TSPNextCityEnv / TimedTSPNextCityEnv semantics (TSP_next_city_env.py:41-109): after this every
        ``step`` also yields shaped_reward / need_next_goal / available goals, see ``goal_info``."""
        check(lib().zenv_goal_enable(self._h))

    def set_goals(self, goals):
        """goals: int32 [N], -1 = leave that env's goal alone (penv.py:76-80 set_goal, batched)."""
        g = np.ascontiguousarray(goals, np.int32)
        if g.shape != (self.num_envs,):
            raise ValueError(f"goals must have shape ({self.num_envs},)")
        check(lib().zenv_set_goals(self._h, g.ctypes.data))

    def solver_goals(self):
        """ColourMatchSolverEnv.solver_get_next_goal (zone-goals/envs/colour_match_solver_env.py:57-97) of every env:
        int32 [N], the nearest zone a cheapest recolouring plan has to cycle."""
        g = np.empty(self.num_envs, np.int32)
        check(lib().zenv_solver_goals(self._h, g.ctypes.data))
        return g

    def goal_info(self):
        """(shaped_reward float64 [N], need_next_goal bool [N], available uint32 bit masks [N], goal int32 [N])."""
        if getattr(self, "_goal_host", None) is None:      # page-locked images: four downloads, one synchronisation
            fields = (nat.F_SHAPED_REWARD, nat.F_NEED_GOAL, nat.F_AVAILABLE_GOALS, nat.F_GOAL)
            self._goal_host = (fields, [self._own_pinned_array(self._shape(f), _FIELD_DTYPES[f]) for f in fields])
        fields, arrays = self._goal_host
        self.results_into(fields, arrays)
        return (arrays[0].copy(), arrays[1].astype(bool), arrays[2].copy(), arrays[3].copy())

    # ------------------------------------------------------------------ actor network
    def load_mlp(self, tensors, precision="auto"):
        """The reference's ZoneEnvModel + actor (env_model.py:48-79, policy_network.py:12-53) for the
        device policies POLICY_MLP_MEAN / POLICY_MLP_SAMPLE.  tensors: dict of float32 arrays named as in
        ``_native.MLP_TENSORS`` (see ``mlp_tensors_from_state_dict``), state_dict layout [out][in].

        precision -- the default keeps the reference's float32 arithmetic (its modules are torch float32):
          "auto" (default)  "f16x3", or "f32" when the weights leave float16's range (ZENV_E_RANGE at load); the mode
                            taken is in ``self.mlp_precision``
          "f16x3"           the 16-bit matrix instruction on hi / lo split float16 operands, three products per k-step:
                            mu / std / value within 3e-6 of torch float32 at 0.30 of "f32"'s time.  float16's range applies:
                            a weight >= 32 768 is refused here, an input / activation >= 65 520 raises
                            ZenvError(E_RANGE) at the next call that waits for the device (then load with "f32")
          "f32"             float32 throughout (f32 MFMA / FMA): within 1e-5 of torch float32
          "bf16x3"          the split with bfloat16 halves: within 2e-5 at 0.35 of "f32"'s time, float32's RANGE -- for
                            networks "f16x3" refuses when 1e-5 is not needed
        Reduced precision, opt-in (narrower arithmetic than the reference's -- for throughput experiments, not parity):
          "bf16"            bf16 MFMA kernels, 8x faster than "f32"; mu / std within 4e-2 of torch float32 by contract
          "f16"             the same kernels on float16 operands: within 1e-3; float16's range guaranteed by a bound on the
                            zone layers at load plus run-time checks (E_RANGE otherwise)."""
        if precision == "auto":
            try:
                self.load_mlp(tensors, precision="f16x3")
            except ZenvError as e:
                if e.code != nat.E_RANGE:
                    raise
                self.load_mlp(tensors, precision="f32")
            return
        F = self.zone_feat
        h = int(np.asarray(tensors["zone_b1"]).shape[0])
        want = {"zone_w1": (h, 8 + F), "zone_b1": (h,), "zone_w2": (h, h), "zone_b2": (h,), "zone_w3": (h, h),
                "zone_b3": (h,), "comb_w": (h, 8 + h), "comb_b": (h,), "enc_w": (h, h), "enc_b": (h,),
                "mu_w": (2, h), "mu_b": (2,), "std_w": (2, h), "std_b": (2,),
                "critic_w1": (h, h), "critic_b1": (h,), "critic_w2": (1, h), "critic_b2": (1,),
                "critic_sigma_w": (1, h), "critic_sigma_b": (1,)}
        keep = {}
        w = nat.MlpWeights(h_dim=h, precision={"bf16": nat.MLP_BF16, "f32": nat.MLP_F32, "bf16x3": nat.MLP_BF16X3,
                                                   "f16x3": nat.MLP_F16X3, "f16": nat.MLP_F16}[precision])
        names = nat.MLP_TENSORS + (nat.MLP_CRITIC_TENSORS if "critic_w1" in tensors else ()) + (
            nat.MLP_SIGMA_TENSORS if "critic_sigma_w" in tensors else ())
        self._mlp_has_critic = "critic_w1" in tensors
        self._mlp_distributional = "critic_sigma_w" in tensors
        for name in names:
            a = np.ascontiguousarray(tensors[name], np.float32)
            if a.shape != want[name]:
                raise ValueError(f"{name}: shape {a.shape}, expected {want[name]}")
            keep[name] = a
            setattr(w, name, a.ctypes.data)
        check(lib().zenv_mlp_load(self._h, C.byref(w)))
        self.mlp_precision = precision

    def mlp_forward(self, with_value=False):
        """(mu, std) float32 [N,2] of the actor's Normal for the current observations; with_value: also
        the critic's value float32 [N] (needs the critic tensors in load_mlp)."""
        check(lib().zenv_mlp_forward(self._h))
        out = (self.get(nat.F_POLICY_MU), self.get(nat.F_POLICY_STD))
        if with_value:
            if not getattr(self, "_mlp_has_critic", False):
                raise ValueError("load_mlp was called without critic tensors")
            out += (self.get(nat.F_POLICY_VALUE),)
            if getattr(self, "_mlp_distributional", False):     # value = (mu, sigma), flat_model.py:57-60
                out += (self.get(nat.F_POLICY_VALUE_SIGMA),)
        return out

    # ------------------------------------------------------------------ one PPO rollout
    def collect(self, frames_per_proc, policy_seed=1, env_index0=0, discount=0.99, gae_lambda=0.95):
        """BaseAlgo.collect_experiences (main/src/torch_ac/algos/base.py:131-227) on the device with the loaded
        actor-critic.  Returns a dict of env-major arrays [N, T, ...] -- reshape(N*T, ...) gives exps.* of the
        reference (:211-227): obs, zone_obs, action, log_prob, value, reward, mask, advantage, returnn (transposed
        views of time-major buffers: reshape copies them once)."""
        T = int(frames_per_proc)
        self.collect_on_device(T, policy_seed, env_index0, discount, gae_lambda)
        out = {}
        for name, (field, shape, time_major) in self.experience_layout(T).items():
            a = np.empty(shape, np.float32)
            assert a.nbytes == lib().zenv_field_bytes(self._h, field)
            check(lib().zenv_get(self._h, field, a.ctypes.data, 0))
            out[name] = a.swapaxes(0, 1) if time_major else a
        return out

    def collect_on_device(self, frames_per_proc, policy_seed=1, env_index0=0, discount=0.99, gae_lambda=0.95):
        """The same rollout, results left in the handle's device buffers (``experience_layout`` names them)."""
        check(lib().zenv_collect(self._h, int(frames_per_proc), int(policy_seed), int(env_index0),
                                 float(discount), float(gae_lambda)))

    def experience_layout(self, frames_per_proc):
        """name -> (field id, shape in memory, time_major) of the float32 buffers one collect of T frames per env
        fills.  Everything is time-major [T, N, ...] in memory (the step kernel writes the observations in place, the
        head kernel and the GAE scan touch whole lines); ``collect`` hands out the [N, T, ...] views."""
        N, Z, F, T = self.num_envs, self.num_zones, self.zone_feat, int(frames_per_proc)
        return {"obs": (nat.F_EXP_OBS, (T, N, 8), True), "zone_obs": (nat.F_EXP_ZONE_OBS, (T, N, Z, F), True),
                "action": (nat.F_EXP_ACTION, (T, N, 2), True), "log_prob": (nat.F_EXP_LOG_PROB, (T, N, 2), True),
                "value": (nat.F_EXP_VALUE, (T, N), True), "reward": (nat.F_EXP_REWARD, (T, N), True),
                "mask": (nat.F_EXP_MASK, (T, N), True), "advantage": (nat.F_EXP_ADVANTAGE, (T, N), True),
                "returnn": (nat.F_EXP_RETURN, (T, N), True)}

    def sync(self):
        check(lib().zenv_sync(self._h))

    def done(self):
        """Non-blocking: has everything enqueued on the handle's stream finished (zenv_query)?"""
        rc = lib().zenv_query(self._h)
        if rc < 0:
            check(rc)
        return rc == 1

    # ------------------------------------------------------------------ multi-GPU: the one collective (native RCCL)
    def comm_init(self, rank, world, unique_id):
        """Join the job's RCCL communicator (collective: every rank calls it).  unique_id: the 128 bytes rank 0 got
        from ``comm_unique_id()``, handed over by the host (sharding.FileRendezvous)."""
        uid = bytes(unique_id)
        if len(uid) != nat.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {nat.COMM_ID_BYTES} bytes")
        check(lib().zenv_comm_init(self._h, int(rank), int(world), uid))
        self.comm_rank, self.comm_world = int(rank), int(world)

    @property
    def comm_library(self):
        path = C.c_char_p()
        check(lib().zenv_comm_info(self._h, None, None, C.byref(path)))
        return (path.value or b"").decode()

    def allgather(self, field):
        """ncclAllGather of one per-env figure over the job's ranks -> [world * N] on the host, ordered by global env
        index (float64 fields arrive as float32, int32 fields as int32)."""
        world = getattr(self, "comm_world", 0)
        if not world:
            raise nat.ZenvError(nat.E_STATE, "comm_init first")
        dt = np.int32 if _FIELD_DTYPES[field] == np.int32 else np.float32
        out = np.empty(world * self.num_envs, dt)
        check(lib().zenv_allgather(self._h, int(field), out.ctypes.data, 0))
        return out

    def comm_barrier(self):
        check(lib().zenv_comm_barrier(self._h))

    def comm_max(self, value):
        v = C.c_double(float(value))
        check(lib().zenv_comm_allreduce_max(self._h, C.byref(v)))
        return v.value

    def set_stream(self, hip_stream=None):
        """Enqueue all further work on the caller's HIP stream (an integer hipStream_t, e.g.
        ``torch.cuda.current_stream().cuda_stream``); None = the handle's own stream again.  0 is the null
        stream (torch's default stream): it goes down as hipStreamLegacy, since NULL means "own stream"
        in the C ABI."""
        if hip_stream is None:
            ptr = None
        else:
            ptr = C.c_void_p(int(hip_stream) if int(hip_stream) else nat.HIP_STREAM_LEGACY)
        check(lib().zenv_set_stream(self._h, ptr))

    @property
    def step_count(self):
        return lib().zenv_step_count(self._h)

    # ------------------------------------------------------------------ results
    def _shape(self, field):
        N = self.num_envs
        if field == nat.F_OBS:
            return (N, nat.OBS_DIM)
        if field in (nat.F_ORDER_VAL, nat.F_ORDER_POS):
            return (N, self.num_zones)
        if field == nat.F_ZONE_OBS:
            return (N, self.num_zones, self.zone_feat)
        if field in (nat.F_ACTIONS, nat.F_POLICY_MU, nat.F_POLICY_STD):
            return (N, 2)
        return (N,)

    def get(self, field, out=None):
        if out is None:
            out = np.empty(self._shape(field), _FIELD_DTYPES[field])
        assert out.nbytes == lib().zenv_field_bytes(self._h, field)
        check(lib().zenv_get(self._h, field, out.ctypes.data, 0))
        return out

    def get_head(self, field, count, first_env=0):
        """Rows [first_env, first_env + count) of an env-major field (zenv_get_rows)."""
        shape = (int(count),) + tuple(self._shape(field)[1:])
        out = np.empty(shape, _FIELD_DTYPES[field])
        check(lib().zenv_get_rows(self._h, int(field), int(first_env), int(count), out.ctypes.data))
        return out

    def get_into_device(self, field, dst_ptr):
        check(lib().zenv_get(self._h, field, C.c_void_p(int(dst_ptr)), 1))

    def device_ptr(self, field):
        p = C.c_void_p()
        check(lib().zenv_device_ptr(self._h, field, C.byref(p)))
        return p.value

    def field_bytes(self, field):
        return lib().zenv_field_bytes(self._h, field)

    def observations(self):
        return self.get(nat.F_OBS), self.get(nat.F_ZONE_OBS)

    # ------------------------------------------------------------------ host-policy surface at PCIe rate
    def pinned_array(self, shape, dtype):
        """A numpy array in page-locked host memory (zenv_host_alloc): uploads from it / downloads into it are
        plain DMA.  The memory stays allocated until the process ends (arrays may outlive the env)."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        ptr = lib().zenv_host_alloc(max(nbytes, 1))
        if not ptr:
            raise nat.ZenvError(nat.E_HIP, "zenv_host_alloc failed")
        buf = (C.c_char * max(nbytes, 1)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def _own_pinned_array(self, shape, dtype):
        a = self.pinned_array(shape, dtype)
        if getattr(self, "_own_pinned", None) is None:
            self._own_pinned = []
        self._own_pinned.append(a)
        return a

    def results_into(self, fields, arrays):
        """Downloads of several fields, one synchronisation (zenv_get_many); arrays should be pinned_array()s."""
        f = (C.c_int * len(fields))(*[int(x) for x in fields])
        d = (C.c_void_p * len(fields))(*[a.ctypes.data for a in arrays])
        check(lib().zenv_get_many(self._h, len(fields), f, d))

    # results of up to this many bytes per step (16 envs of 25 zones: 10 KB) are worth keeping in host memory: two copy
    # enqueues cost more than the kernel writing them over the bus itself
    HOST_IO_MAX_BYTES = 256 << 10

    def host_io(self, enable=True):
        """zenv_host_io: the results slab and the action buffer in page-locked HOST memory that the kernels write / read
        themselves -- step_results() is then one launch and one wait, no upload, no download (for small batches driven
        by a host policy: ParallelEnv and the single-env gym surface switch it on by themselves, enable="if small":
        results of at most HOST_IO_MAX_BYTES per step)."""
        if enable == "if small":
            enable = lib().zenv_results_layout(self._h, (C.c_int64 * nat.N_RESULTS)()) <= self.HOST_IO_MAX_BYTES
        res, act = C.c_void_p(), C.c_void_p()
        check(lib().zenv_host_io(self._h, int(bool(enable)), C.byref(res), C.byref(act)))
        self._slab = self._slab_views = self._host_actions = None
        self._host_io = bool(enable)
        if enable:
            off = (C.c_int64 * nat.N_RESULTS)()
            total = lib().zenv_results_layout(self._h, off)
            self._host_actions = np.ctypeslib.as_array((C.c_float * (2 * self.num_envs)).from_address(act.value)).reshape(
                self.num_envs, 2)
            self._results_slab(raw=np.ctypeslib.as_array((C.c_uint8 * total).from_address(res.value)))
        return self

    def _results_slab(self, raw=None):
        """One page-locked host image of the handle's results slab, with typed views of its pieces."""
        if getattr(self, "_slab", None) is None:
            off = (C.c_int64 * nat.N_RESULTS)()
            total = lib().zenv_results_layout(self._h, off)
            if raw is None:
                raw = self._own_pinned_array((total,), np.uint8)
            N, Zn, F = self.num_envs, self.num_zones, self.zone_feat

            def view(i, count, dtype, shape):
                return raw[off[i]:off[i] + count * np.dtype(dtype).itemsize].view(dtype).reshape(shape)
            self._slab = raw
            self._slab_views = (view(nat.RESULT_OBS, N * 8, np.float32, (N, 8)),
                                view(nat.RESULT_ZONE_OBS, N * Zn * F, np.float32, (N, Zn, F)),
                                view(nat.RESULT_REWARD, N, np.float32, (N,)),
                                view(nat.RESULT_DONE, N, np.uint8, (N,)).view(bool),
                                view(nat.RESULT_GOAL_MET, N, np.uint8, (N,)).view(bool),
                                view(nat.RESULT_EXCEPTION, N, np.uint8, (N,)).view(bool))
        return self._slab

    def step_results(self, actions=None, auto_reset=True, copy=True):
        """One env.step() of a host policy in ONE call (zenv_step_results): action upload, step, download of every
        per-step result, one synchronisation.  actions None: just the download (after reset()).  Returns
        (obs, zone_obs, reward, done, goal_met, exception); with copy=False the arrays are views of one page-locked
        buffer that the next call overwrites."""
        a = None
        if actions is not None:
            a = np.ascontiguousarray(actions, np.float32)
            if a.shape != (self.num_envs, 2):
                raise ValueError(f"actions must have shape ({self.num_envs}, 2)")
        if copy and self.num_envs * self.num_zones * self.zone_feat * 4 > (8 << 20):
            # a big batch whose caller wants its own arrays: download each field straight into them (a second pass
            # over tens of MB through the page-locked image would cost more than the extra synchronisations)
            if a is not None:
                self.step(a, auto_reset=auto_reset)
            return (self.get(nat.F_OBS), self.get(nat.F_ZONE_OBS), self.get(nat.F_REWARD),
                    self.get(nat.F_DONE).view(bool), self.get(nat.F_GOAL_MET).view(bool),
                    self.get(nat.F_EXCEPTION).view(bool))
        if getattr(self, "_host_io", False):      # the kernel reads the actions from, and writes the results to, host memory
            if a is not None:
                self._host_actions[...] = a
                check(lib().zenv_step_host(self._h, int(bool(auto_reset))))
            else:
                self.sync()
            return tuple(v.copy() for v in self._slab_views) if copy else self._slab_views
        slab = self._results_slab()
        check(lib().zenv_step_results(self._h, None if a is None else a.ctypes.data, int(bool(auto_reset)),
                                      slab.ctypes.data))
        return tuple(v.copy() for v in self._slab_views) if copy else self._slab_views

    def results(self):
        """(obs, zone_obs, reward, done, goal_met) of the last step, as host arrays."""
        return self.step_results(None)[:5]

    # ------------------------------------------------------------------ snapshots / debug
    def get_state(self):
        n = lib().zenv_state_bytes(self._h)
        buf = np.empty(n, np.uint8)
        check(lib().zenv_get_state(self._h, buf.ctypes.data, n))
        return buf

    def set_state(self, blob):
        b = np.ascontiguousarray(blob, np.uint8)
        check(lib().zenv_set_state(self._h, b.ctypes.data, b.nbytes))

    def debug_state(self):
        N, Z = self.num_envs, self.num_zones
        out = dict(qpos=np.empty((N, 3)), qvel=np.empty((N, 3)),
                   zone_state=np.empty((N, Z), np.int32), cooldown=np.empty((N, Z), np.int32),
                   steps=np.empty(N, np.int32))
        check(lib().zenv_debug_state(self._h, out["qpos"].ctypes.data, out["qvel"].ctypes.data,
                                     out["zone_state"].ctypes.data, out["cooldown"].ctypes.data,
                                     out["steps"].ctypes.data))
        return out


__all__ = ["ZoneVecEnv", "Config", "ZenvError", "config_for_id", "default_config",
           "sample_layout", "fixed_seed_sequence", "zone_feat"]
