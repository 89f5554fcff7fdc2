"""One process per GPU: how the N_env envs, their map seeds and their returns are sharded.

The step path has no exchange (every env is independent, like the reference's one-process-
per-env ``ParallelEnv``, main/src/torch_ac/torch_utils/penv.py:26-40), so ranks never talk
during a rollout.  The only collective of a job is one all-gather of the per-env episodic
returns (RCCL over xGMI on GPUs; gloo in the CPU tests).  Global env g = rank*n + i plays map
seeds seed0 + g, seed0 + g + G, seed0 + g + 2G, ... with G = world*n, so results do not
depend on how the envs are split over ranks (shard invariance).
"""
import numpy as np


class EnvShard:
    def __init__(self, rank, world, envs_per_rank, seed0=1):
        if not (0 <= rank < world):
            raise ValueError("rank outside [0, world)")
        self.rank, self.world, self.n = int(rank), int(world), int(envs_per_rank)
        self.seed0 = int(seed0)
        self.env_index0 = self.rank * self.n          # global index of local env 0
        self.seed_stride = self.world * self.n        # G

    def first_seeds(self):
        return self.seed0 + self.env_index0 + np.arange(self.n, dtype=np.int64)

    def bank_seeds(self, episodes_per_env):
        """Slot i + k*n holds the seed of local env i's k-th episode."""
        k = np.arange(int(episodes_per_env), dtype=np.int64)[:, None]
        return (self.first_seeds()[None, :] + k * self.seed_stride).reshape(-1)

    def build_bank(self, env, episodes_per_env, n_threads=8):
        env.build_bank_seeds(self.bank_seeds(episodes_per_env), n_threads=n_threads)
        env.schedule_sequential(first=np.arange(self.n, dtype=np.int32), stride=self.n)

    # ------------------------------------------------------------------ the one collective
    def all_gather(self, local):
        """local: 1-D torch tensor [n] (cuda for nccl, cpu for gloo) -> [world*n] on every rank,
        ordered by global env index."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            return local.clone()
        out = torch.empty(self.world * self.n, dtype=local.dtype, device=local.device)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, local.contiguous())
        else:
            parts = list(out.chunk(self.world))
            dist.all_gather(parts, local.contiguous())
        return out

    def gather_returns(self, env):
        """Episodic return of each env's last finished episode, float32, all ranks."""
        from . import _native as nat
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            return env.get(nat.F_LAST_RETURN).astype(np.float32)
        buf = torch.empty(self.n, dtype=torch.float64, device=f"cuda:{env.device}")
        env.get_into_device(nat.F_LAST_RETURN, buf.data_ptr())
        local = buf.to(torch.float32)
        if dist.get_backend() != "nccl":          # gloo (CPU tests, rehearsals): gather host tensors
            local = local.cpu()
        return self.all_gather(local).cpu().numpy()
