"""The RCCL branch of the sharded job (sharding.py: init_process_group("nccl") -> get_into_device ->
all_gather_into_tensor) on the one GPU a test box has: bench.py started as a FRESH child process (never a re-exec of
the pytest process, which has already initialised the GPU) with a 1-rank process group, checked against the same
command without a process group.  The N > 1 layout of the gather is covered by tests/test_sharding_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29641", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "512", "--warmup", "8", "--workload", "ColourMatch-6",
           "--no-cpu-baseline", "--no-mlp", "--no-steady", "--no-settle", "--envs-per-gpu", "8192", *args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"rc {r.returncode}\nstdout: {r.stdout[-2000:]}\nstderr: {r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_one_rank_nccl_gather_matches_the_plain_run(zenv_mod):
    plain = _bench({})
    dist = _bench({"ZENV_BENCH_FORCE_DIST": "1"})
    assert plain["aux"]["collective"] == "none (single process)"
    assert dist["aux"]["collective"] == "nccl all_gather_into_tensor"
    assert "all-gather" in dist["config"]["parallelism"]
    assert dist["n_gpus"] == 1 and dist["steps"] == 512
    # same envs, same seeds, same steps: the gathered returns are the local ones
    assert dist["aux"]["mean_last_return_all_ranks"] == plain["aux"]["mean_last_return_all_ranks"]
    assert dist["aux"]["episodes_finished_rank0"] == plain["aux"]["episodes_finished_rank0"] > 0
    assert dist["aux"]["mean_last_return_all_ranks"] != 0.0
    assert dist["aux"]["parity_spot_check"] == "bit-identical" == plain["aux"]["parity_spot_check"]


def test_two_rank_launch_line_on_one_gpu(zenv_mod):
    """The driver's N > 1 command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2`
    -- on this box's one GPU (ZENV_BENCH_REHEARSAL=gloo: both ranks on cuda:0, the process group over gloo): build lock,
    rendezvous, per-rank shards, barrier-bracketed timing, MAX over ranks, the gather, one JSON line from rank 0.  Shard
    invariance on the device: 2 ranks x 4096 envs return what one process with 8192 envs returns."""
    env = dict(os.environ)
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "ZENV_BENCH_REHEARSAL": "gloo"})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    tail = ["--steps", "512", "--warmup", "8", "--workload", "ColourMatch-6", "--no-cpu-baseline", "--no-mlp", "--no-steady",
            "--no-settle"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29643", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs-per-gpu", "4096", *tail]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"rc {r.returncode}\nstdout: {r.stdout[-2000:]}\nstderr: {r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    two = json.loads(lines[0])
    one = _bench({}, "--envs-per-gpu", "8192")
    assert two["n_gpus"] == 2 and two["config"]["n_env_total"] == 8192 == one["config"]["n_env_total"]
    assert two["aux"]["collective"] == "gloo" and two["scaling"] == "weak"
    assert two["value"] > 0 and two["roofline"]["kernel"] == "k_rollout_lane"
    assert two["cpu_baseline"] is None                      # an N = 1 figure
    assert two["aux"]["mean_last_return_all_ranks"] == one["aux"]["mean_last_return_all_ranks"] != 0.0
    assert two["aux"]["parity_spot_check"] == "bit-identical"
