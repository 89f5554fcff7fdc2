"""One process per GPU: how the N_env envs, their map seeds and their returns are sharded.

The step path has no exchange (every env is independent, like the reference's one-process-
per-env ``ParallelEnv``, main/src/torch_ac/torch_utils/penv.py:26-40), so ranks never talk
during a rollout.  The only collective of a job is one all-gather of the per-env episodic
returns: ``ncclAllGather`` over RCCL / xGMI through the C ABI (``zenv_allgather``), no PyTorch.
Global env g = rank*n + i plays map seeds seed0 + g, seed0 + g + G, seed0 + g + 2G, ... with
G = world*n, so results do not depend on how the envs are split over ranks (shard invariance).

What the host has to do for that collective is hand RCCL's unique id from rank 0 to the other
ranks: ``FileRendezvous`` does it through a directory every rank of one node can see (the ranks
of a launcher are children of one agent process; its pid + start time name the directory).  The
same object carries the CPU-only fallbacks used by rehearsals on a box with fewer GPUs than
ranks (RCCL refuses two ranks per device) and by the CPU tests.  ``EnvShard.all_gather`` keeps
the ``torch.distributed`` form for callers that already run a process group (gloo in tests).
"""
import os
import time

import numpy as np


class FileRendezvous:
    """Rank-0-to-all broadcast, all-gather and barrier of small host byte strings through files of one directory.

    Not a data path: it moves the 128-byte RCCL unique id (and, in rehearsals without RCCL, the gathered returns).
    Every exchange has a name; a name is used once per job.  File `<name>.<r>` is written by rank r only.

    One directory per ATTEMPT of a launch: the launcher agent's pid + start time + MASTER_PORT name the launch, a nonce
    names the attempt (ZENV_RDZV_NONCE from a launcher that sets one -- bench.py's own spawner does --, else
    TORCHELASTIC_RUN_ID / TORCHELASTIC_RESTART_COUNT, which torchrun changes when it restarts its workers): a rank of a
    restarted job never reads the RCCL unique id a crashed attempt left behind.  On open every rank also removes the
    files it owns itself from whatever the directory held."""

    def __init__(self, rank, world, directory=None, timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.dir = directory or self.default_directory()
        os.makedirs(self.dir, exist_ok=True)
        suffix = f".{self.rank}"
        for f in os.listdir(self.dir):                      # leftovers of an earlier user of this directory
            if f.endswith(suffix) or f.endswith(suffix + ".tmp"):
                try:
                    os.remove(os.path.join(self.dir, f))
                except OSError:
                    pass

    @staticmethod
    def attempt_nonce():
        e = os.environ
        if e.get("ZENV_RDZV_NONCE"):
            return e["ZENV_RDZV_NONCE"]
        parts = [e.get("TORCHELASTIC_RUN_ID", ""), e.get("TORCHELASTIC_RESTART_COUNT", "")]
        return "-".join("".join(c if c.isalnum() else "_" for c in p) for p in parts if p)

    @staticmethod
    def default_directory():
        """The launch: the launcher agent (the parent of every rank) by pid and start time, + the port; the attempt: a
        nonce (see the class docstring)."""
        ppid = os.getppid()
        start = "0"
        try:
            with open(f"/proc/{ppid}/stat") as f:
                start = f.read().rsplit(")", 1)[1].split()[19]      # starttime, clock ticks since boot
        except (OSError, IndexError):
            pass
        port = os.environ.get("MASTER_PORT", "0")
        nonce = FileRendezvous.attempt_nonce()
        return os.path.join(os.environ.get("ZENV_RDZV_DIR", "/tmp"),
                            f"zenv_rdzv_{ppid}_{start}_{port}" + (f"_{nonce}" if nonce else ""))

    def _path(self, name, rank):
        return os.path.join(self.dir, f"{name}.{rank}")

    def _put(self, name, data):
        tmp = self._path(name, self.rank) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, self._path(name, self.rank))        # appears whole or not at all

    def _get(self, name, rank):
        path, t0 = self._path(name, rank), time.monotonic()
        while True:
            try:
                with open(path, "rb") as f:                 # (no exists()-then-open window)
                    return f.read()
            except FileNotFoundError:
                pass
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rendezvous: rank {rank} never wrote {name!r} in {self.dir} "
                                   f"(waited {self.timeout:.0f} s; is every rank of the job running?)")
            time.sleep(0.002)

    def broadcast(self, name, data=None):
        """rank 0 passes `data` (bytes); every rank returns it."""
        if self.rank == 0:
            self._put(name, bytes(data))
            return bytes(data)
        return self._get(name, 0)

    def all_gather(self, name, data):
        self._put(name, bytes(data))
        return [self._get(name, r) for r in range(self.world)]

    def barrier(self, name):
        self.all_gather(name, b"1")

    def close(self, name="close", linger=10.0):
        """Everybody is done with the directory; rank 0 removes it -- once every other rank has said that it has read
        all it is going to read (a `left` marker behind its own barrier), or after `linger` seconds: a rank that is
        still polling never finds its files gone, and a rank that died cannot keep rank 0 here."""
        self.barrier(name)
        if self.rank != 0:
            self._put(name + "_left", b"1")
            return
        t0 = time.monotonic()
        for r in range(1, self.world):
            while not os.path.exists(self._path(name + "_left", r)) and time.monotonic() - t0 < linger:
                time.sleep(0.002)
        for f in os.listdir(self.dir):
            try:
                os.remove(os.path.join(self.dir, f))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
        except OSError:
            pass


class EnvShard:
    def __init__(self, rank, world, envs_per_rank, seed0=1):
        if not (0 <= rank < world):
            raise ValueError("rank outside [0, world)")
        self.rank, self.world, self.n = int(rank), int(world), int(envs_per_rank)
        self.seed0 = int(seed0)
        self.env_index0 = self.rank * self.n          # global index of local env 0
        self.seed_stride = self.world * self.n        # G
        self.host_comm = None                         # FileRendezvous standing in for RCCL (rehearsals)

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
    def comm_init(self, env, rdzv):
        """Native RCCL communicator over the job's ranks: rank 0 draws the unique id, the rendezvous hands it round."""
        from .vec_env import comm_unique_id
        uid = rdzv.broadcast("rccl_unique_id", comm_unique_id() if self.rank == 0 else None)
        env.comm_init(self.rank, self.world, uid)

    def all_gather(self, local):
        """local: 1-D torch tensor [n] (cuda for nccl, cpu for gloo) -> [world*n] on every rank,
        ordered by global env index.  (The torch.distributed form, for callers that run a process group.)"""
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
        """Episodic return of each env's last finished episode, float32 [world * n] on every rank."""
        from . import _native as nat
        if getattr(env, "comm_world", 0):                       # native RCCL (zenv_allgather)
            return env.allgather(nat.F_LAST_RETURN)
        local = env.get(nat.F_LAST_RETURN).astype(np.float32)
        if self.host_comm is not None:                          # rehearsal: ranks share a GPU, RCCL unavailable
            parts = self.host_comm.all_gather("returns", local.tobytes())
            return np.concatenate([np.frombuffer(p, np.float32) for p in parts])
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                import torch
                t = torch.from_numpy(local)
                if dist.get_backend() == "nccl":            # RCCL gathers device tensors only
                    t = t.to(f"cuda:{getattr(env, 'device', 0)}")
                return self.all_gather(t).cpu().numpy()
        except ImportError:
            pass
        return local
