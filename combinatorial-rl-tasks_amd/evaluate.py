"""The reference's evaluation protocol (main/scripts/evaluate.py:22-25,47-78) on one device.

100 maps (env seeds 1000000..1000099) x 5 runs per map, undiscounted episodic return,
result layout ``{"return": [[r_run0..r_run4] for each map]}`` -- but all maps and runs are
stepped together as one batch of n_maps*n_runs envs instead of 500 sequential episodes.
"""
import pickle

import numpy as np

from . import _native as nat
from .vec_env import ZoneVecEnv, config_for_id

EVAL_SEED0 = 1000000     # evaluate.py:47


def load_model_state(model_dir):
    """``utils.get_model_state(model_dir)`` (main/src/utils/storage.py:36-52): the ``model_state`` entry of the
    ``status.pt`` a reference training run leaves in its model directory (train_ppo.py:201-208).  ``model_dir`` may also
    name the file itself.  The checkpoint holds only tensors and plain Python objects, so torch loads it without any of
    the reference's modules."""
    import os
    import torch
    path = model_dir if os.path.isfile(model_dir) else os.path.join(model_dir, "status.pt")
    status = torch.load(path, map_location="cpu", weights_only=True)
    if "model_state" not in status:
        raise KeyError(f"{path} has no 'model_state' (keys: {sorted(status)})")
    return status["model_state"]


def evaluate(env_id, policy, n_maps=100, n_runs_per_map=5, env_seed0=EVAL_SEED0, device=0,
             policy_seed=0, pkl_path=None, max_steps=None, argmax=False, precision="f32"):
    """policy: ZENV_POLICY_* (on-device scripted policy), a callable
    ``policy(obs (B,8) float32, zone_obs (B,Z,F) float32) -> actions (B,2)`` running on the host
    (e.g. the reference's ``Agent.get_actions`` behind a small adapter), an ACModel ``state_dict``
    (main/src/flat_model.py:24-52 names; what ``utils.Agent`` loads, main/src/utils/agent.py:15-28) or the path of
    a reference model directory / ``status.pt`` (``load_model_state``): the actor
    then runs on the device (``csrc/mlp_policy.hip``), ``dist.sample()`` per step as ``Agent.get_actions`` does
    (agent.py:41-44), or the mean with ``argmax=True``; ``precision`` "f32" (default: the reference's own arithmetic,
    actions within 1e-5 of its torch float32 modules), "f16x3" (within 3e-6 as well, a third of the time on batches of
    2 048 envs and more -- smaller ones run the float32 vector kernel either way) "bf16" (the MFMA kernels, 8x faster) or "f16" (the same kernels on
    float16 operands: within 1e-3, float16's range guarded).

    Returns ``{"return": [[...]], "length": [[...]], "goal_met": [[...]]}``."""
    cfg = config_for_id(env_id) if isinstance(env_id, str) else env_id
    n = n_maps * n_runs_per_map
    env = ZoneVecEnv(cfg, n, device=device)
    env.build_bank(env_seed0, n_maps)
    env.schedule_sequential(first=np.repeat(np.arange(n_maps, dtype=np.int32), n_runs_per_map), stride=0)
    env.reset()
    if isinstance(policy, str):
        policy = load_model_state(policy)
    if isinstance(policy, dict):
        from .vec_env import mlp_tensors_from_state_dict
        env.load_mlp(mlp_tensors_from_state_dict(policy), precision=precision)
        policy = nat.POLICY_MLP_MEAN if argmax else nat.POLICY_MLP_SAMPLE
    goal = np.zeros(n, bool)
    horizon = cfg.num_steps if max_steps is None else max_steps
    if callable(policy):
        o, zo = env.step_results(None, copy=False)[:2]
    for t in range(horizon):
        if callable(policy):               # one upload, one launch, one download, one synchronisation per step
            o, zo, _, d, g, _ = env.step_results(np.asarray(policy(o, zo), np.float32), auto_reset=False, copy=False)
        else:
            env.policy(int(policy), policy_seed=policy_seed)
            env.step(None, auto_reset=False)
            _, _, _, d, g, _ = env.step_results(None, copy=False)
        goal |= g
        if d.all():                        # every episode finished (evaluate.py:64-72)
            break
    out = {
        "return": env.get(nat.F_LAST_RETURN).reshape(n_maps, n_runs_per_map).tolist(),
        "length": env.get(nat.F_LAST_LEN).reshape(n_maps, n_runs_per_map).tolist(),
        "goal_met": goal.reshape(n_maps, n_runs_per_map).tolist(),
    }
    env.close()
    if pkl_path:
        with open(pkl_path, "wb") as f:       # evaluate.py:76-78 writes {"return": record_returns}
            pickle.dump({"return": out["return"]}, f)
    return out
