"""Zero-copy hand-off between the env buffers and a device-resident PyTorch policy.

The reference moves every observation to the host and every action back
(``main/src/torch_ac/algos/base.py:139-145``: ``preprocess_obss(self.obs, device)`` ...
``action.cpu().numpy()`` ... ``self.env.step``).  Here the env's output buffers are exposed as
``torch`` tensors that ALIAS device memory (no copy, through ``__cuda_array_interface__``), the
env is put on torch's current stream, and ``step`` takes the action tensor's device address: a
closed loop policy -> step -> policy never leaves the GPU and needs no synchronisation.

torch is plumbing here (device memory, streams); no env arithmetic runs through it.
"""
import numpy as np

from . import _native as nat
from .vec_env import _FIELD_DTYPES

_TYPESTR = {np.dtype(np.float32): "<f4", np.dtype(np.float64): "<f8", np.dtype(np.uint8): "|u1",
            np.dtype(np.int32): "<i4", np.dtype(np.int64): "<i8"}


class _DeviceView:
    """Minimal __cuda_array_interface__ (v2) carrier for a buffer owned by the env handle."""

    def __init__(self, ptr, shape, dtype):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": _TYPESTR[np.dtype(dtype)],
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class TorchZoneEnv:
    """Tensor view of a ``ZoneVecEnv``: ``obs (N,8)``, ``zone_obs (N,Z,F)``, ``reward (N,)`` float32,
    ``done``/``goal_met (N,)`` uint8, ``ep_return``/``last_return (N,)`` float64 -- all aliases of the
    env's device buffers, valid until ``env.close()``; they change in place on every ``step``."""

    def __init__(self, env, use_current_stream=True):
        import torch
        self._torch = torch
        self.env = env
        self.device = torch.device("cuda", env.device)
        with torch.cuda.device(self.device):
            if use_current_stream:
                env.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
            self.obs = self._alias(nat.F_OBS)
            self.zone_obs = self._alias(nat.F_ZONE_OBS)
            self.reward = self._alias(nat.F_REWARD)
            self.done = self._alias(nat.F_DONE)
            self.goal_met = self._alias(nat.F_GOAL_MET)
            self.ep_return = self._alias(nat.F_EP_RETURN)
            self.last_return = self._alias(nat.F_LAST_RETURN)
            self.episodes = self._alias(nat.F_EPISODES)

    def _alias(self, field):
        t = self._torch.as_tensor(_DeviceView(self.env.device_ptr(field), self.env._shape(field),
                                              _FIELD_DTYPES[field]), device=self.device)
        assert t.data_ptr() == self.env.device_ptr(field), "torch copied instead of aliasing"
        return t

    def _alias_raw(self, field, shape):
        t = self._torch.as_tensor(_DeviceView(self.env.device_ptr(field), shape, np.float32), device=self.device)
        assert t.data_ptr() == self.env.device_ptr(field), "torch copied instead of aliasing"
        return t

    def reset(self, mask=None):
        self.env.reset(mask)
        return {"obs": self.obs, "zone_obs": self.zone_obs}

    def load_state_dict(self, state_dict, precision="auto"):
        """Put an ACModel state_dict (main/src/flat_model.py:24-52 names; torch tensors on any device) into the
        device actor-critic that ``collect`` and the ``POLICY_MLP_*`` action sources run.  precision: see
        ``ZoneVecEnv.load_mlp`` -- the default is float32-grade (the reference's modules are float32); "bf16" is the
        fast reduced-precision mode."""
        from .vec_env import mlp_tensors_from_state_dict
        self.env.load_mlp(mlp_tensors_from_state_dict(state_dict), precision=precision)

    def collect(self, frames_per_proc, policy_seed=1, env_index0=0, discount=0.99, gae_lambda=0.95):
        """collect_experiences (torch_ac/algos/base.py:131-227) on the device; returns exps.* as float32 CUDA
        tensors [N, T, ...] ALIASING the handle's experience buffers (overwritten by the next collect;
        transposed views of time-major memory, so ``reshape(N*T, ...)`` copies them once).  Enqueued
        on the shared stream like everything else: no synchronisation, no host copy."""
        self.env.collect_on_device(frames_per_proc, policy_seed, env_index0, discount, gae_lambda)
        out = {}
        for name, (field, shape, time_major) in self.env.experience_layout(frames_per_proc).items():
            t = self._alias_raw(field, shape)
            out[name] = t.transpose(0, 1) if time_major else t        # [N, T, ...] views either way
        return out

    def step(self, actions, auto_reset=True):
        """actions: float32 CUDA tensor (N, 2) on the env's device (contiguous).  Asynchronous: the
        step kernel is enqueued behind whatever produced ``actions`` on the shared stream."""
        torch = self._torch
        if not (actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
                and tuple(actions.shape) == (self.env.num_envs, 2) and actions.device == self.device):
            raise ValueError("actions must be a contiguous float32 CUDA tensor of shape (N, 2) on the env's device")
        self.env.step_device(actions.data_ptr(), auto_reset=auto_reset)
        return {"obs": self.obs, "zone_obs": self.zone_obs}, self.reward, self.done, self.goal_met
