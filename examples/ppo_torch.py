"""PPO on the device env: experience collection by ``zenv_collect`` (env step + actor-critic forward + GAE in
HIP kernels, nothing leaves the GPU), parameter update by plain PyTorch autograd on the same device and stream.

This is the loop of the reference's ``train_ppo.py`` (main/scripts/train_ppo.py:127-190 with
torch_ac/algos/ppo.py:32-140, recurrence 1) with ``ParallelEnv`` + ``collect_experiences`` replaced by
``TorchZoneEnv.collect``; the update itself is the caller's side of the boundary and stays ordinary torch.
``ActorCritic`` has the reference ACModel's parameter names (flat_model.py:24-52), so its checkpoints load
into either.

    python examples/ppo_torch.py --env PointTSP-v0 --procs 4096 --frames-per-proc 64 --updates 20
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z  # noqa: E402
from combinatorial_rl_tasks_amd.torch_interop import TorchZoneEnv  # noqa: E402


class _ZoneEncoder(nn.Module):          # ZoneEnvModel: shared MLP over (obs, zone row), mean over zones, combine
    def __init__(self, zone_feat, h):
        super().__init__()
        self.zone_net_ = nn.Sequential(nn.Linear(8 + zone_feat, h), nn.ReLU(), nn.Linear(h, h), nn.ReLU(),
                                       nn.Linear(h, h))
        self.combine_net_ = nn.Linear(8 + h, h)

    def forward(self, obs, zone_obs):
        n_zones = zone_obs.shape[1]
        x = torch.cat([obs.unsqueeze(1).expand(-1, n_zones, -1), zone_obs], dim=-1)
        return self.combine_net_(torch.cat([obs, self.zone_net_(x).mean(dim=1)], dim=-1))


class _GaussianHead(nn.Module):         # PolicyNetwork, Box branch
    def __init__(self, h):
        super().__init__()
        self.enc_ = nn.Sequential(nn.Sequential(nn.Linear(h, h), nn.ReLU()))
        self.mu_ = nn.Linear(h, 2)
        self.std_ = nn.Linear(h, 2)

    def forward(self, emb):
        a = self.enc_(emb)
        return torch.distributions.Normal(2.0 * (torch.sigmoid(self.mu_(a)) - 0.5), torch.sigmoid(self.std_(a)) + 1e-3)


class ActorCritic(nn.Module):
    def __init__(self, zone_feat, h=185):
        super().__init__()
        self.env_model = _ZoneEncoder(zone_feat, h)
        self.actor = _GaussianHead(h)
        self.critic = nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, 1))
        for m in self.modules():        # init_params: unit-norm rows, zero bias
            if isinstance(m, nn.Linear):
                with torch.no_grad():
                    m.weight.normal_(0, 1)
                    m.weight /= m.weight.pow(2).sum(1, keepdim=True).sqrt()
                    m.bias.zero_()

    def forward(self, obs, zone_obs):
        emb = self.env_model(obs, zone_obs)
        return self.actor(emb), self.critic(emb).squeeze(1)


def ppo_update(model, opt, exps, epochs, batch_size, clip_eps, entropy_coef, value_loss_coef, max_grad_norm, gen):
    """One update_parameters(): `epochs` passes over shuffled minibatches of the flattened experiences."""
    flat = {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in exps.items()}
    total = flat["obs"].shape[0]
    stats = {}
    for _ in range(epochs):
        order = torch.randperm(total, device=flat["obs"].device, generator=gen)
        for lo in range(0, total, batch_size):
            idx = order[lo:lo + batch_size]
            dist, value = model(flat["obs"][idx], flat["zone_obs"][idx])
            adv, ret, old_v = flat["advantage"][idx], flat["returnn"][idx], flat["value"][idx]
            ratio = torch.exp((dist.log_prob(flat["action"][idx]) - flat["log_prob"][idx]).sum(dim=1))
            policy_loss = -torch.min(ratio * adv, ratio.clamp(1.0 - clip_eps, 1.0 + clip_eps) * adv).mean()
            v_clip = old_v + (value - old_v).clamp(-clip_eps, clip_eps)
            value_loss = torch.max((value - ret).pow(2), (v_clip - ret).pow(2)).mean()
            entropy = dist.entropy().mean()
            loss = policy_loss - entropy_coef * entropy + value_loss_coef * value_loss
            opt.zero_grad(set_to_none=True)
            loss.backward()
            grad_norm = nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
            opt.step()
            stats = {"policy_loss": policy_loss, "value_loss": value_loss, "entropy": entropy, "grad_norm": grad_norm}
    return {k: float(v.detach()) for k, v in stats.items()}


def train(env_id="PointTSP-v0", procs=4096, frames_per_proc=64, updates=10, epochs=4, batch_size=16384, lr=3e-4,
          discount=0.99, gae_lambda=0.95, clip_eps=0.2, entropy_coef=0.003, value_loss_coef=0.5, max_grad_norm=0.5,
          hidden=185, seed=1, log=print):
    torch.manual_seed(seed)
    dev = torch.device("cuda", 0)
    env = Z.ZoneVecEnv(env_id, procs)
    env.build_bank(seed, 4 * procs)             # fresh layouts on every reset, as FixedSeedWrapper draws them
    env.schedule_sequential(stride=procs)
    tenv = TorchZoneEnv(env)
    tenv.reset()
    model = ActorCritic(env.zone_feat, hidden).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr, eps=1e-8)
    gen = torch.Generator(device=dev).manual_seed(seed)
    history = []
    for u in range(updates):
        t0 = time.perf_counter()
        tenv.load_state_dict(model.state_dict())
        exps = tenv.collect(frames_per_proc, policy_seed=seed * 1000003 + u, discount=discount, gae_lambda=gae_lambda)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = ppo_update(model, opt, exps, epochs, batch_size, clip_eps, entropy_coef, value_loss_coef, max_grad_norm, gen)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        frames = procs * frames_per_proc
        st.update(update=u, frames=frames * (u + 1), reward_per_frame=float(exps["reward"].mean()),
                  collect_fps=frames / (t1 - t0), update_fps=frames * epochs / (t2 - t1))
        history.append(st)
        log({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()})
    env.close()
    return model, history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="PointTSP-v0")
    ap.add_argument("--procs", type=int, default=4096)
    ap.add_argument("--frames-per-proc", type=int, default=64)
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=16384)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--hidden-size", type=int, default=185)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    train(a.env, a.procs, a.frames_per_proc, a.updates, a.epochs, a.batch_size, a.lr, hidden=a.hidden_size, seed=a.seed)
