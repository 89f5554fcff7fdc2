#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched PointTSP-25 env.step() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every one of the N_env = 65 536 envs
takes one env.step() (10 MuJoCo substeps, visit logic, reward/termination, fused auto-reset,
obs emit) and all of its outputs (obs, zone_obs, reward, done, goal_met) are written to HBM.
The scripted closed-loop policy pi_greedy(obs) that produces the next action runs on the
device.  --mode persistent (default): one launch of k_rollout_lane covers up to 256 steps with
the env state in registers; --mode per_step: one k_step_lane launch per step (the kernel of
step t also emits a_{t+1}); --mode unfused: a policy kernel + a step kernel per step.  The
results are bit-identical in all modes.  All inputs (env state, layout bank) are resident in
HBM before the timed region.

The host side is Python over the C ABI (ctypes) only: no PyTorch is imported at any N.
N > 1: one process per GPU, launched as the driver does (`python -m torch.distributed.run ...`
starts the ranks; RANK / LOCAL_RANK / WORLD_SIZE come from its environment).  Envs shard
trivially: rank r owns global envs [r*65536, (r+1)*65536); there is no collective on the step
path; after the rollout the per-env episodic returns are all-gathered with ncclAllGather (RCCL
over xGMI) through the C ABI (zenv_allgather); the barrier and the max-over-ranks of the step
time are RCCL calls too (zenv_comm_barrier / zenv_comm_allreduce_max), each followed by a
synchronisation of the handle's stream -- the only stream this process enqueues work on.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (k_rollout_lane, or k_step_lane with --mode per_step) and `cpu_baseline` (the
float64 C oracle on the host cores).  Side blocks: `aux.workloads` (BASELINE configs 2-3 and the
15-zone map, steady state + per-step launches + spot check each), `aux.beyond_llc` (PointTSP-25 at
batches whose per-step output exceeds the 256 MiB Infinity Cache, with the bare store stream of
the same footprint beside each), `aux.env_overrides` (every switch that changes what is measured).
"""
import argparse
import hashlib
import json
import os
import sys
import time

# multi-process GPU work on this pool: the host driver only supports dmabuf IPC, and without this RCCL's
# hipIpcGetMemHandle fails.  The launcher exports it; kept here for a launch that does not (read when HSA initialises,
# i.e. at the first HIP call, long after this line)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LLC_BYTES = 256 << 20          # Infinity Cache (MI355X_MICROARCH.md: 256 MiB, die-level)

WORKLOADS = {
    # name: (task, zones, zones_keepout)  -- SURVEY.md 8(d) configs 1-3 (+ the reference-faithful 15-zone map)
    "PointTSP-25": (0, 25, 0.40),
    "TimedTSP-25": (1, 25, 0.40),
    "ColourMatch-6": (2, 6, 0.55),
    "PointTSP-15": (0, 15, 0.55),
}

SETTLE_STEPS = 6000     # untimed steps (settle + warmup) before the timed region, ~40 ms of GPU time
EPISODES_PER_ENV = 4    # depth of the map bank per env; the schedule wraps around it (the oracle too)
DEFAULT_SLICE = 65536   # envs per persistent launch (zenv_set_rollout_slice's default): one env wave per SIMD
BEYOND_LLC_SIZES = (262144, 1048576, 3145728)   # 160 MB, 640 MB, 1.9 GB of outputs per step (LLC = 268 MB)
BEYOND_LLC_BANK = 262144                        # maps of the sweep: env i replays map 1 + (i mod this)

# Environment variables that change WHAT is measured (a diagnostic variant of the library, another launch shape,
# another kernel): recorded in aux.env_overrides; the run is refused unless --experiment / --override is given.
# (ZENV_BENCH_*, ZENV_CPU_THREADS, ZENV_RDZV_*, ZENV_RCCL_PATH, ZENV_FUZZ_CASES steer the harness, not the kernels.)
HARNESS_ENV = ("ZENV_BENCH_", "ZENV_CPU_THREADS", "ZENV_RDZV_", "ZENV_RCCL_PATH", "ZENV_FUZZ_CASES")
KERNEL_SOURCES = ("kernels.hip", "kernels.hpp", "dev_params.hpp", "det_math.hpp")   # what the PMC record depends on


def algorithmic_bytes(task, Z, steps_per_launch=1):
    """HBM bytes one env-step must move: 1247 B for PointTSP-25 when every step
    is its own launch (state in + state out + outputs).  A persistent launch of K steps keeps the
    state in registers: per env-step it must still publish the outputs (obs, zone_obs, reward,
    done = 637 B for PointTSP-25) and moves the state once per launch, i.e. (state in + out) / K."""
    F = 6 if task == 0 else 7
    reads = 8 + 48 + 24 + 16 * Z + Z + 16
    state_out = 48 + Z + 16
    outputs = 4 + 1 + 32 + 4 * Z * F
    if task == 1:
        reads += 4 * Z
    if task == 2:
        reads += Z
        state_out += Z
    return outputs + (reads + state_out) / steps_per_launch


def output_bytes_per_step(task, Z, n_env):
    """What one step of the whole batch writes (the stream that either fits the Infinity Cache or does not)."""
    F = 6 if task == 0 else 7
    return n_env * (4 + 1 + 32 + 4 * Z * F)


def kernel_sources_sha():
    """sha256 over the sources the env kernels are compiled from: the committed PMC record names the one it was
    measured on, so a kernel change without a new PMC pass shows (traffic_stale; tests/test_bench_contract.py)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "combinatorial-rl-tasks_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


_TRAFFIC = None


def traffic_record():
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                _TRAFFIC = json.load(f)
        except (OSError, ValueError):
            _TRAFFIC = {}
    return _TRAFFIC


def traffic_is_stale():
    meta = traffic_record().get("_meta", {})
    return meta.get("kernel_sources_sha256") != kernel_sources_sha()


def load_pmc(workload, n_env, mode):
    """The committed rocprofv3 PMC record of this workload's dominant kernel (profiles/traffic.json, written by
    scripts/summarize_profile.py from separate --pmc passes): HBM bytes as `fixed per launch + per step` (FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is) and the VALU-issue counters."""
    return traffic_record().get(f"{workload}@{n_env}", {}).get("persistent" if mode == "persistent" else "per_step")


def traffic_for_launch(pmc, steps_per_launch):
    """PMC HBM bytes of ONE launch of `steps_per_launch` steps (None when no PMC record is committed)."""
    if not pmc or "hbm_bytes_per_step" not in pmc:
        return None
    return int(round(pmc.get("hbm_bytes_fixed_per_launch", 0.0) + pmc["hbm_bytes_per_step"] * steps_per_launch))


def valu_issue(pmc, us_per_step):
    """How busy the SIMDs' vector issue is (the bound of the small-row workloads, where HBM bytes are not):
    VALU instructions per SIMD and step (committed PMC, a property of the binary and the workload) x 4 issue cycles
    (a float64 or a lone wave's float32 instruction holds the SIMD-32's issue for 4 cycles, MI355X_MICROARCH.md
    cycle-constants table) / the cycles a step takes.  Both the PMC pass's own cycle count and -- with the live
    time per step at the committed clock -- the live figure are given."""
    if not pmc or "valu_insts_per_simd_step" not in pmc:
        return None
    insts, cyc = pmc["valu_insts_per_simd_step"], pmc.get("gpu_cycles_per_step")
    out = {"valu_insts_per_simd_step": round(insts, 1), "issue_cycles_per_inst": 4,
           "pmc_gpu_cycles_per_step": None if cyc is None else round(cyc, 1),
           "frac_pmc": None if not cyc else round(4.0 * insts / cyc, 4),
           "salu_insts_per_simd_step": pmc.get("salu_insts_per_simd_step"),
           "source": pmc.get("source")}
    clk = pmc.get("gpu_clock_ghz")
    if clk and us_per_step:
        out["frac_live"] = round(4.0 * insts / (us_per_step * 1e3 * clk), 4)
        out["clock_ghz_assumed"] = clk
    return out


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def experiment_switches(native, args):
    """Everything that makes this run something else than the shipped library at its shipped launch shape."""
    L = native.lib()
    env = {k: v for k, v in sorted(os.environ.items())
           if k.startswith("ZENV_") and not k.startswith(HARNESS_ENV)}
    flags = (L.zenv_build_flags() or b"").decode()
    chunk = int(L.zenv_rollout_chunk())
    out = {"build_flags": flags, "env": env, "rollout_chunk": chunk, "config_overrides": list(args.override),
           "library": (L.zenv_version() or b"").decode()}
    out["active"] = bool(flags or env or chunk != native.ROLLOUT_CHUNK or args.override)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--warmup", type=int, default=1000, help="untimed steps right before the timed ones")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the untimed clock-settling steps that precede the warm-up (see SETTLE_STEPS)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--bank-maps", type=int, default=0,
                    help="0 (default): env g plays map seeds 1+g, 1+g+G, ... from a 4-episode-deep bank.  M > 0: env i "
                         "replays map 1 + (i mod M) in every episode -- the batch-size sweep's setting, where a bank of "
                         "4 x 3 M layouts would only cost host time")
    ap.add_argument("--rollout-slice", type=int, default=None,
                    help="envs one persistent launch covers (zenv_set_rollout_slice; library default 65536, 0 = the whole "
                         "batch in one launch -- the setting in which every step's outputs really stream to HBM)")
    ap.add_argument("--workload", default="PointTSP-25", choices=sorted(WORKLOADS))
    ap.add_argument("--policy", default="greedy", choices=["greedy", "uniform"])
    ap.add_argument("--override", action="append", default=[],
                    help="experiment only: config key=value (e.g. frameskip=1); marks the run invalid")
    ap.add_argument("--experiment", action="store_true",
                    help="allow a diagnostic library variant / ZENV_* kernel switch (recorded in aux.env_overrides and "
                         "in config.workload); without it such a run is refused")
    ap.add_argument("--mode", choices=["persistent", "per_step", "unfused"], default="persistent",
                    help="persistent: one launch per 256 steps, env state in registers, every step's outputs "
                         "still written; per_step: one step-kernel launch per step (also emits the next "
                         "action); unfused: per-step launches + a policy kernel before each")
    ap.add_argument("--unfused", action="store_true",
                    help="run the stand-alone policy kernel before every step instead of the fused action source")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady", action="store_true",
                    help="skip the 8192-step steady-state side measurement of the same kernel (aux.steady_state)")
    ap.add_argument("--no-mlp", action="store_true",
                    help="skip the side measurement with the reference's actor network as the on-device policy")
    ap.add_argument("--no-per-step", action="store_true",
                    help="skip the per-step-kernel side measurement (aux.per_step_launch_mode)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip aux.workloads (the other BASELINE configs) and aux.beyond_llc (the batch-size sweep)")
    ap.add_argument("--event-stride", type=int, default=16,
                    help="time every k-th step-kernel dispatch with begin/end HIP events (each pair costs "
                         "~6 us of launch path, so timing all of them would distort `value`)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="skip the per-launch HIP events around the step kernel")
    ap.add_argument("--dispatch-events", action="store_true",
                    help="persistent mode: begin/end events on every dispatch of the TIMED region (hipExtLaunchKernel; "
                         "costs ~17 us of wall per call -- scripts/launch_overhead.py) instead of the two events recorded "
                         "on the stream right before and after its launches; the steady-state side run always has them")
    ap.add_argument("--print-rank-env", action="store_true",
                    help="diagnostic: print this rank's launch environment (RANK, WORLD_SIZE, rendezvous directory ...) as "
                         "one JSON line and exit -- what tests/test_bench_contract.py checks the self-spawn path with")
    args = ap.parse_args()
    if args.unfused:
        args.mode = "unfused"
    args.unfused = args.mode == "unfused"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Build (or find up to date) the native library BEFORE this process touches the GPU: a compiler child must never be
    # forked from a GPU-initialised process.  Every rank calls it; build_library() serialises concurrent callers with
    # a file lock and is a no-op when lib/ is newer than csrc/ and was built with the same flags.
    import __graft_entry__ as entry
    entry.build()
    if args.print_rank_env and (world > 1 or args.gpus == 1):
        from combinatorial_rl_tasks_amd.sharding import FileRendezvous
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "ppid": os.getppid(),
                          "master": [os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")],
                          "rdzv_dir": FileRendezvous.default_directory(),
                          "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
        return None
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # a bare `python bench.py --gpus N`: be the launcher -- N fresh rank processes (never a re-exec: this
            # process has not touched the GPU and does not from here on), one rendezvous directory, one nonce
            raise SystemExit(spawn_ranks(args.gpus))
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py: RANK is set but WORLD_SIZE is 1: launch with `python bench.py --gpus N` (it starts "
                             "its own ranks) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
        args.gpus = world

    import combinatorial_rl_tasks_amd as Z
    from combinatorial_rl_tasks_amd import sharding
    native = Z._native

    exp = experiment_switches(native, args)
    if exp["active"] and not (args.experiment or args.override):
        raise SystemExit(f"bench.py: refusing to time a diagnostic configuration without --experiment: {json.dumps(exp)}")

    # ZENV_BENCH_REHEARSAL=host (or gloo, its old name): the driver's N-rank launch line on a box with fewer GPUs than
    # ranks (tests only: ranks share the cards; RCCL refuses two ranks on one device, so the gather, the barrier and
    # the max-over-ranks go through the host rendezvous instead)
    rehearsal = os.environ.get("ZENV_BENCH_REHEARSAL") in ("gloo", "host")
    distributed = world > 1 or os.environ.get("ZENV_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank rehearsal
    n_dev = native.lib().zenv_device_count()
    if rehearsal and n_dev > 0:
        local_rank = local_rank % n_dev
    elif distributed and local_rank >= n_dev:
        raise SystemExit(f"bench.py: rank {rank} (local rank {local_rank}) has no device: {n_dev} HIP device(s) visible, "
                         f"{world} rank(s) asked for -- one process per GPU.  (ZENV_BENCH_REHEARSAL=host rehearses the "
                         "launch line on fewer GPUs than ranks, ranks sharing the cards.)")

    task, zones, keepout = WORKLOADS[args.workload]
    n_env = args.envs_per_gpu
    cfg = Z.default_config(task, zones, zones_keepout=keepout)
    for kv in args.override:
        key, val = kv.split("=")
        setattr(cfg, key, type(getattr(cfg, key))(float(val)))
    policy = Z.POLICY_GREEDY if args.policy == "greedy" else Z.POLICY_UNIFORM
    shard = sharding.EnvShard(rank=rank, world=world, envs_per_rank=n_env)

    # env g (global index) plays map seeds 1+g, 1+g+G, 1+g+2G, ... (G = global env count)
    env = Z.ZoneVecEnv(cfg, n_env, device=local_rank)
    slice_envs = DEFAULT_SLICE if args.rollout_slice is None else args.rollout_slice
    if args.rollout_slice is not None:
        env.set_rollout_slice(args.rollout_slice)
    rdzv = None
    rccl_error = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rdzv = sharding.FileRendezvous(rank, world)
        if rehearsal:
            shard.host_comm = rdzv
        else:
            # ncclCommInitRank on this rank's device (collective).  The collective is NOT on the timed path (one gather of
            # returns after the rollout, one barrier on either side of the region): when RCCL cannot be brought up -- a
            # library that does not load, a fabric the ranks cannot agree on -- the ranks agree on that through the
            # rendezvous and the job still measures its shards, with the gather / barrier / max over the host, and says
            # so in the line (`collective`, `aux.rccl_ranks` = 0, `aux.rccl_error`).
            try:
                shard.comm_init(env, rdzv)
            except Exception as ex:
                rccl_error = f"rank {rank}: {ex}"
            oks = rdzv.all_gather("rccl_ok", (rccl_error or "").encode())
            failed = [b.decode() for b in oks if b]
            if failed:
                rccl_error = failed[0]
                if getattr(env, "comm_world", 0):
                    native.lib().zenv_comm_destroy(env._h)
                    env.comm_world = 0
                shard.host_comm = rdzv
                rehearsal = True
    t_bank = time.perf_counter()
    if args.bank_maps > 0:
        spot_seeds, spot_stride, spot_period = replay_bank(env, n_env, args.bank_maps, shard.env_index0)
    else:
        shard.build_bank(env, EPISODES_PER_ENV, n_threads=min(32, usable_cores()))
        spot_seeds, spot_stride, spot_period = shard.first_seeds(), shard.seed_stride, EPISODES_PER_ENV
    t_bank = time.perf_counter() - t_bank
    env.reset()
    # Untimed: first let the power controller settle (it dips to ~1.5 GHz 4-15 ms after load arrives and
    # is steady after ~20 ms, profiles/r01/clock_per_launch.txt) -- enough steps that settle + warmup
    # is at least SETTLE_STEPS -- then the W warm-up steps, then the K timed steps, back to back.
    args.settle = 0 if args.no_settle else max(0, SETTLE_STEPS - args.warmup)
    for k in (args.settle, args.warmup):
        if k > 0:
            env.rollout(k, policy, policy_seed=0x5EED, env_index0=shard.env_index0, mode=args.mode)

    fences = [0]

    def fence():
        """barrier + device synchronisation: this rank's stream drained, then every rank's."""
        env.sync()
        if distributed:
            fences[0] += 1
            if rehearsal:
                rdzv.barrier(f"fence{fences[0]}")
            else:
                env.comm_barrier()              # RCCL all-reduce on the stream + hipStreamSynchronize
            env.sync()

    # The K timed steps, bracketed by barrier + synchronize on both sides.  A rank's clock stops when ITS device has
    # finished (after its own synchronize, before it enters the closing barrier): the figure reported is the MAX over
    # ranks, i.e. the time until the slowest rank was done -- the barrier's own latency (an RCCL round, tens of
    # microseconds) is not part of anybody's K steps.
    # persistent mode: the launches of the region sit between two hipEventRecord()s on the kernel's stream (on one launch
    # that bracket reads within 0.2 us of the dispatch's own begin/end events, which cost 17 us of host path per call)
    timed_dispatch_events = (not args.no_kernel_events) and (args.mode != "persistent" or args.dispatch_events)
    fence()
    t0 = time.perf_counter()
    ms_total, ms_kernel = env.rollout(args.steps, policy, policy_seed=0x5EED,
                                      env_index0=shard.env_index0, auto_reset=True,
                                      time_step_kernel=timed_dispatch_events,
                                      mode=args.mode, event_stride=args.event_stride)
    env.sync()
    elapsed = time.perf_counter() - t0
    fence()

    # rank-local results, then the one collective of the job: all-gather of episodic returns
    returns = shard.gather_returns(env)          # float32 [world * n_env] on every rank
    if distributed:
        if rehearsal:
            elapsed = max(float(np.frombuffer(b, np.float64)[0])
                          for b in rdzv.all_gather("elapsed", np.float64(elapsed).tobytes()))
        else:
            elapsed = env.comm_max(elapsed)
    rccl_ranks = int(getattr(env, "comm_world", 0)) if distributed and not rehearsal else 0
    collective = ("none (single process)" if not distributed else
                  "host rendezvous (RCCL could not be initialised)" if rccl_error else
                  "host rendezvous (rehearsal: ranks share a GPU)" if rehearsal else
                  f"rccl ncclAllGather (native), {env.comm_library}")

    out = None
    if rank == 0:
        lib_chunk = exp["rollout_chunk"]
        total_env_steps = world * n_env * args.steps
        value = total_env_steps / elapsed
        persistent = args.mode == "persistent" and zones in (5, 6, 10, 15, 20, 25)
        chunk = min(lib_chunk, max(args.steps, 1)) if persistent else 1
        n_launches = (args.steps + lib_chunk - 1) // lib_chunk if persistent else args.steps
        pmc = load_pmc(args.workload, n_env, "persistent" if persistent else "per_step")
        roofline = None
        if args.steps > 0 and (ms_kernel is not None or not args.unfused):
            # Duration of the dominant kernel per step, from HIP events on the kernel's stream over the timed
            # region.  persistent: two events recorded on the stream right before and after the region's launches
            # (one launch for --steps <= 256: the bracket is that dispatch + the sub-microsecond gaps around it), or,
            # with --dispatch-events, begin/end events of every dispatch (hipExtLaunchKernel) summed / steps -- what
            # rocprofv3's kernel trace reports for the same dispatches; the steady-state block below always uses those.
            # per_step: the events that bracket the back-to-back loop / steps (an upper bound: it
            # includes the ~0.5 us gaps between launches; timing every dispatch would slow the loop), with the
            # begin/end events of every event_stride-th dispatch beside it.
            loop_s = ms_total / 1e3 / args.steps
            if persistent and ms_kernel is not None:
                k_step_s = ms_kernel / 1e3
            elif args.unfused:
                k_step_s = ms_kernel / 1e3
            else:
                k_step_s = loop_s
            roofline = roofline_block(task, zones, n_env, k_step_s, chunk if persistent else 1, persistent, pmc, slice_envs)
            roofline.update({
                "kernel_launches_timed": n_launches if not args.unfused else
                (args.steps + args.event_stride - 1) // args.event_stride,
                "timing": ("begin/end HIP events of every dispatch of the timed region" if persistent and
                           ms_kernel is not None else
                           "hipEventRecord on the kernel's stream right before and after the region's launch(es)"
                           if persistent else "HIP events around the back-to-back launch loop / steps"),
                "loop_us_per_step": round(loop_s * 1e6, 3),
                "sampled_dispatch_avg_us": None if ms_kernel is None else round(ms_kernel * 1e3, 3),
            })
        # side measurements, GPU ones first and back to back (each settles the clock itself); the CPU baseline last
        ep = env.get(Z.F_EPISODES)
        spot = parity_spot_check(env, cfg, spot_seeds, spot_stride, shard.env_index0,
                                 args.settle + args.warmup + args.steps, policy, spot_period)
        side = not distributed and not args.override
        steady = None if (args.no_steady or args.override) else \
            steady_state(env, task, zones, policy, shard.env_index0, args.mode, pmc, lib_chunk, slice_envs=slice_envs)
        per_step = per_step_rate(env, task, zones, policy, shard.env_index0, args.workload) \
            if (args.mode == "persistent" and side and not args.no_per_step) else None
        chunked = action_chunk_rate(env, task, zones, lib_chunk, workload=args.workload) \
            if (args.mode == "persistent" and side and not args.no_per_step) else None
        mlp = None if (args.no_mlp or not side) else mlp_policy_rate(env, zones)
        host_rt = None if (args.no_mlp or not side) else host_roundtrip_rate(env)
        sweep = side and not args.no_sweep and args.workload == "PointTSP-25" and n_env == 65536 and \
            args.mode == "persistent"
        env.close()
        workloads = other_workloads(Z, policy, lib_chunk, local_rank) if sweep else None
        beyond = beyond_llc(Z, policy, lib_chunk, local_rank) if sweep else None
        ceiling = store_stream_ceiling(Z, task, zones, n_env, local_rank) if side else None
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is an N = 1 figure (rank 0 only)
            cpu = cpu_baseline(cfg, task, zones, keepout, policy)
        # the figures to compare rounds by ride inside `roofline` as well (compact), beside the timed region's own
        if roofline is not None:
            roofline["steady_state"] = compact(steady)
            roofline["per_step_kernel"] = compact(per_step)
            roofline["store_stream_ceiling"] = ceiling
            # the number to read as "fraction of HBM when the bytes really go to HBM": the same kernel on the largest
            # batch of the sweep, whose per-step output is several times the Infinity Cache
            big = [b for b in (beyond or []) if isinstance(b, dict) and isinstance(b.get("persistent"), dict)
                   and not b["persistent"].get("llc_resident", True)]
            if big:
                b = big[-1]
                roofline["frac_hbm_resident"] = b["persistent"]["frac"]
                roofline["hbm_resident_case"] = {
                    "n_env": b["n_env"], "launch": "one launch over the whole batch (rollout slice 0)",
                    "output_bytes_per_step": b["persistent"]["output_bytes_per_step"],
                    "achieved": b["persistent"]["achieved"], "kernel_us_per_step": b["persistent"]["kernel_us_per_step"],
                    "us_per_65536_envs": b["persistent"].get("us_per_65536_envs"), "traffic": b["persistent"]["traffic"],
                    "store_stream_GBps": (b.get("store_stream") or {}).get("best_GBps")}
            else:
                roofline["frac_hbm_resident"] = None
        tag = " EXPERIMENT " + json.dumps({k: exp[k] for k in ("build_flags", "env", "config_overrides")
                                           if exp[k]}) if exp["active"] else ""
        out = {
            "metric": "env-steps/sec", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}, N_env={n_env} per GPU, num_steps=2000, "
                                   f"zones_keepout={keepout}, policy=pi_{args.policy} (on-device, "
                                   f"launch mode {args.mode}), auto-reset on" +
                                   (f", env i replays map 1 + (i mod {args.bank_maps})" if args.bank_maps else "") +
                                   (f", rollout slice {args.rollout_slice}" if args.rollout_slice is not None else "") + tag,
                       "n_env_total": world * n_env, "zones": zones,
                       "parallelism": f"env-shard x{world}, all-gather(ep_return) after rollout"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "aux": {"collective": collective, "rccl_ranks": rccl_ranks, "rccl_error": rccl_error, "host_side": "python + ctypes over the C ABI (no PyTorch imported)"
                    if sys.modules.get("torch") is None else "python + ctypes over the C ABI (torch present in the process)",
                    "env_overrides": exp, "traffic_stale": traffic_is_stale(),
                    "settle_steps_untimed": args.settle, "hip_event_ms_total": round(ms_total, 3),
                    "bank_build_s": round(t_bank, 2), "episodes_finished_rank0": int(ep.sum()),
                    "mean_last_return_all_ranks": float(np.mean(returns[returns != 0]))
                    if (returns != 0).any() else 0.0,
                    "parity_spot_check": spot, "steady_state": steady, "per_step_launch_mode": per_step,
                    "action_chunk": chunked,
                    "workloads": workloads, "beyond_llc": beyond,
                    "mlp_policy": mlp, "host_policy_roundtrip_pcie_inclusive": host_rt},
        }
        print(json.dumps(out), flush=True)
    else:
        env.close()
    if distributed:
        rdzv.close()
    return out


def spawn_ranks(n, timeout=None):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same command line -- fresh
    children (subprocess: fork + exec from a process that never initialised the GPU), RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment as torch.distributed.run would set them, ONE rendezvous directory and attempt nonce for
    the RCCL unique id.  Returns the exit code: 0 when every rank exited 0; when one fails the others are ended (they
    would wait for it in ncclCommInitRank or at the rendezvous)."""
    import secrets
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rdzv_root = tempfile.mkdtemp(prefix="zenv_bench_")
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                ZENV_RDZV_DIR=rdzv_root, ZENV_RDZV_NONCE=secrets.token_hex(8), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                          env=dict(base, RANK=str(r), LOCAL_RANK=str(r))))
        t0, rc = time.monotonic(), 0
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad or (timeout and time.monotonic() - t0 > timeout):
                rc = bad[0] if bad else 124
                break
            time.sleep(0.05)
        rc = rc or next((p.returncode for p in procs if p.returncode), 0)
        return rc
    finally:
        for p in procs:                      # exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        import shutil
        shutil.rmtree(rdzv_root, ignore_errors=True)


def replay_bank(env, n_env, maps, env_index0=0):
    """Bank of min(n_env, maps) layouts (seeds 1 ..), env i replaying map 1 + ((env_index0 + i) mod maps) in every
    episode.  Returns the oracle's (first seeds, seed stride, seed period) for the spot check."""
    S = min(n_env, maps) if env_index0 == 0 else maps
    env.build_bank(1, S, n_threads=min(32, usable_cores()))
    first = ((env_index0 + np.arange(n_env, dtype=np.int64)) % S).astype(np.int32)
    env.schedule_sequential(first=first, stride=0)
    return 1 + first.astype(np.int64), 0, 1


def compact(b):
    return None if not isinstance(b, dict) else {
        k: b.get(k) for k in ("kernel", "steps_per_launch", "kernel_us_per_step", "kernel_avg_us", "achieved",
                              "frac", "algorithmic_bytes_per_env_step", "algorithmic_bytes_per_launch",
                              "traffic", "llc_resident", "envs_per_launch", "steps", "launches", "env_steps_per_s")}


def roofline_block(task, zones, n_env, k_step_s, steps_per_launch, persistent, pmc, envs_per_launch=None):
    """`roofline` of the bench line for a kernel that takes k_step_s per step in launches of steps_per_launch.

    Two byte bases, both algorithmic: `outputs_only` -- what a launch of K steps must move when the
    state stays in registers: K x the step's outputs + the state once (for K = 1 this IS the second base) --
    and SURVEY.md 8(d)'s per-step figure, which charges the state round trip to every step.  `achieved`/`frac`
    use the first, the one that describes the kernel measured; for the persistent kernel the second is given
    for reference only (it exceeds 1: the kernel does not do that traffic, by design).  `llc_resident`: the batch's
    per-step output fits the 256 MiB Infinity Cache, i.e. the stream is rewritten in place on chip and the HBM peak
    is not what physically bounds it (aux.beyond_llc holds the batches where it does not fit)."""
    alg = algorithmic_bytes(task, zones, steps_per_launch)
    alg1 = algorithmic_bytes(task, zones, 1)
    achieved = alg * n_env / k_step_s / 1e9
    achieved1 = alg1 * n_env / k_step_s / 1e9
    traffic = traffic_for_launch(pmc, steps_per_launch)
    out_bytes = output_bytes_per_step(task, zones, n_env)
    # what is rewritten in place while a launch runs: the outputs of the envs ONE launch covers (a persistent launch
    # covers a slice of the batch, a per-step launch all of it -- but then every step is a new launch over everything)
    per_launch = n_env if (not persistent or not envs_per_launch) else min(n_env, envs_per_launch)
    launch_out_bytes = output_bytes_per_step(task, zones, per_launch) if persistent else out_bytes
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_note": (None if traffic is None else
                         f"committed PMC bytes (FETCH_SIZE x2 + WRITE_SIZE) scaled to this launch of "
                         f"{steps_per_launch} step(s); {pmc.get('source')}"),
        "llc_resident": bool(launch_out_bytes <= LLC_BYTES), "output_bytes_per_step": int(out_bytes),
        "envs_per_launch": int(per_launch), "output_bytes_per_step_per_launch": int(launch_out_bytes), "llc_bytes": LLC_BYTES,
        "kernel": "k_rollout_lane" if persistent else "k_step_lane",
        "kernel_avg_us": round(k_step_s * 1e6 * steps_per_launch, 2),
        "steps_per_launch": steps_per_launch,
        "kernel_us_per_step": round(k_step_s * 1e6, 3),
        "env_steps_per_launch": n_env * steps_per_launch,
        "algorithmic_bytes_per_env_step": round(alg, 1),
        "algorithmic_bytes_per_launch": int(round(alg * n_env * steps_per_launch)),
        "byte_bases": {
            "outputs_only": {"bytes_per_env_step": round(alg, 1), "achieved": round(achieved, 1),
                             "frac": round(achieved / HBM_PEAK_GBS, 4),
                             "what": "outputs of every step + state in/out once per launch"},
            "survey_8d": {"bytes_per_env_step": alg1, "achieved": round(achieved1, 1),
                          "frac": round(achieved1 / HBM_PEAK_GBS, 4),
                          "what": "state in + state out + outputs on EVERY step (one launch per step)",
                          "applies": steps_per_launch == 1},
        },
        "valu_issue": valu_issue(pmc, k_step_s * 1e6),
    }


def store_stream_ceiling(Z, task, zones, n_env, device, steps=64):
    """The bare row stream of this batch's footprint on THIS box (zenv_probe_store_stream): one wave per 64-env tile
    rewrites the tile's Z*F*4*64 contiguous bytes with 1 KiB dwordx4 bursts under each cache policy -- the write-only
    ceiling the kernels' row stores are held against (a float4 COPY reads and writes, it is not that ceiling)."""
    try:
        from combinatorial_rl_tasks_amd.vec_env import probe_store_stream
        F = 6 if task == 0 else 7
        tile_bytes, n_tiles = 64 * zones * F * 4, (n_env + 63) // 64
        mb = n_tiles * tile_bytes / 1e6
        # fewer sweeps over a multi-GB footprint, more over a small one: ~20 ms of stores per launch
        steps = int(max(4, min(steps, 160000 // max(mb, 1))))
        res = {}
        for name, pol in (("plain", 0), ("nt", 2), ("sc1", 16)):
            us = probe_store_stream(n_tiles, tile_bytes, steps=steps, cache_policy=pol, reps=3, device=device)
            res[name] = {"us_per_step": round(us, 3), "GBps": round(mb / us * 1e3, 1)}
        best = max(res, key=lambda k: res[k]["GBps"])
        return {"footprint_mb": round(mb, 1), "tile_bytes": tile_bytes, "steps_per_launch": steps, "policies": res,
                "best_policy": best, "best_GBps": res[best]["GBps"], "best_us_per_step": res[best]["us_per_step"],
                "llc_resident": bool(n_tiles * tile_bytes <= LLC_BYTES)}
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def steady_state(env, task, zones, policy, env_index0, mode, pmc, lib_chunk, steps=8192, settle=SETTLE_STEPS,
                 slice_envs=DEFAULT_SLICE):
    """Side measurement (never `value`): the SAME kernel, same envs, right after the timed region, over enough
    steps that launch overheads and the clock transient are out of the picture -- every dispatch timed with its
    own begin/end HIP events.  This is the figure to compare rounds by."""
    try:
        persistent = mode == "persistent" and zones in (5, 6, 10, 15, 20, 25)
        env.rollout(settle, policy, policy_seed=0x5EED, env_index0=env_index0, mode=mode)   # untimed
        ms, ms_k = env.rollout(steps, policy, policy_seed=0x5EED, env_index0=env_index0, mode=mode,
                               time_step_kernel=persistent)
        chunk = min(lib_chunk, steps) if persistent else 1
        k_step_s = (ms_k if persistent else ms / steps) / 1e3
        blk = roofline_block(task, zones, env.num_envs, k_step_s, chunk, persistent, pmc, slice_envs)
        n_slices = 1 if (not persistent or not slice_envs) else -(-env.num_envs // slice_envs)
        blk.update({"steps": steps, "launches": n_slices * ((steps + chunk - 1) // chunk),
                    "env_steps_per_s": round(env.num_envs * steps / (ms * 1e-3), 1),
                    "loop_us_per_step": round(ms / steps * 1e3, 3)})
        return blk
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("ZENV_CPU_THREADS", "64"))))


def cpu_baseline(cfg, task, zones, keepout, policy):
    """The float64 C oracle (the build's own port; the mujoco-py reference cannot run here)
    timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import oracle as O
    from tests.helpers import oracle_config_from
    ocfg = oracle_config_from(O, cfg)
    cores = usable_cores()
    n, T = 65536, 3000
    seeds = np.arange(1, 1 + n)
    O.rollout(ocfg, seeds[:256], 20, policy, n_threads=cores)      # warm the pages/threads
    t0 = time.perf_counter()
    r = O.rollout(ocfg, seeds, T, policy, seed_stride=65536, policy_seed=0x5EED, n_threads=cores)
    dt = time.perf_counter() - t0
    # one core (the single-env loop evaluate.py runs; the mujoco-py original is absent)
    t1 = time.perf_counter()
    r1 = O.rollout(ocfg, seeds[:1024], 1000, policy, seed_stride=65536, policy_seed=0x5EED, n_threads=1)
    dt1 = time.perf_counter() - t1
    return {"value": round(r["total_steps"] / dt, 1), "unit": "env-steps/s", "cores": cores,
            "cpu_model": cpu_model(), "nproc": os.cpu_count(),
            "kind": "port",
            "sample": f"first {n} envs x {T} steps of the same workload, OpenMP over envs, "
                      f"{dt:.2f}s wall",
            "one_core_value": round(r1["total_steps"] / dt1, 1),
            "one_core_sample": f"1024 envs x 1000 steps on 1 thread, {dt1:.2f}s wall"}


def action_chunk_rate(env, task, zones, lib_chunk, n_steps=2048, skill_len=10, workload=None):
    """Side measurement (never `value`): the same envs driven by EXTERNALLY supplied actions in chunks (zenv_step_many:
    k_rollout_lane's action-buffer form) -- what a consumer with pre-computed or open-loop action sequences gets instead
    of one k_step_lane launch per step.

    Like for like with the headline kernel: the actions are the scripted greedy policy's own.  From a snapshot S0 the
    envs are stepped `n_steps` times with one k_step_lane launch per step, every a_t copied (device to device) into an
    [n_steps][N][2] buffer resident in HBM; the snapshot is restored and the buffer replayed through zenv_step_many -- the
    same trajectories, auto-resets included, as ceil(n_steps / 256) launches -- and the replay's observations, episode
    counts and returns must equal the recorded run's bit for bit (`replay_matches`).  Timed: the replay between two stream
    synchronisations (>= 10 ms of GPU time; three times from the same snapshot, the median reported), right behind ~20 ms of a bare store stream so that it does not start
    on the boost clock of an idle GPU.  Algorithmic bytes per env-step = the persistent kernel's (outputs of every step,
    state once per launch) + the action read (8) + the time-major reward / done records (5).  Beside it the
    fixed-length-skill shape: the same buffer in chunks of `skill_len` steps that reset at the boundary
    (_hier_policy_opt.py:68-71), a launch every 10 steps."""
    try:
        import combinatorial_rl_tasks_amd as Z
        from combinatorial_rl_tasks_amd import _native as nat
        from combinatorial_rl_tasks_amd.vec_env import probe_store_stream
        n, K = env.num_envs, int(n_steps)
        F = 6 if task == 0 else 7
        env.step_many(np.zeros((K, n, 2), np.float32), reset="every")      # allocates the chunk buffers (untimed)
        ptr = env.device_ptr(nat.F_CHUNK_ACTIONS)
        env.rollout(SETTLE_STEPS // 2, Z.POLICY_GREEDY)
        env.policy(Z.POLICY_GREEDY)                                         # a_0 of the recording
        s0 = env.get_state()
        for t in range(K):
            env.get_into_device(nat.F_ACTIONS, ptr + 8 * n * t)
            env.rollout(1, Z.POLICY_GREEDY, mode="per_step")               # leaves a_{t+1} in the action buffer
        want = [env.get(f).copy() for f in (nat.F_OBS, nat.F_ZONE_OBS, nat.F_EPISODES, nat.F_LAST_RETURN, nat.F_EP_LEN)]
        ep_rec = int(want[2].sum())

        def warm():
            probe_store_stream((n + 63) // 64, 64 * zones * F * 4, steps=max(64, int(20e3 / 6)), cache_policy=16, reps=1,
                               device=env.device)
        out = {}
        runs, match = [], True
        for _ in range(3):                      # the same replay three times from the same snapshot; the median is reported
            env.set_state(s0)
            ep0 = int(env.get(nat.F_EPISODES).sum())
            warm()
            env.sync()
            t0 = time.perf_counter()
            env.step_many(None, reset="every", actions_ptr=(ptr, K))
            env.sync()
            runs.append(time.perf_counter() - t0)
            got = [env.get(f) for f in (nat.F_OBS, nat.F_ZONE_OBS, nat.F_EPISODES, nat.F_LAST_RETURN, nat.F_EP_LEN)]
            match = match and all(np.array_equal(a, b) for a, b in zip(got, want))
        dt = sorted(runs)[1]
        env.set_state(s0)                       # like for like: the scripted kernel over the same 2 048 steps from the same snapshot
        warm()
        ms_scripted, _ = env.rollout(K, Z.POLICY_GREEDY)
        rec = traffic_record().get(f"{workload}@{n}", {}).get("action_chunk") if workload else None

        def block(k_launch, steps, dt, reset):
            alg = algorithmic_bytes(task, zones, k_launch) + 8 + 5
            ach = alg * n * steps / dt / 1e9
            return {"kernel": "k_rollout_lane<EXT>", "steps_per_launch": k_launch, "reset": reset, "steps_timed": steps,
                    "us_per_step": round(dt / steps * 1e6, 3), "env_steps_per_s": round(n * steps / dt, 1),
                    "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(ach / HBM_PEAK_GBS, 4),
                                 "traffic": None if not rec or k_launch != lib_chunk
                                 else int(round(rec["hbm_bytes_per_step"] * k_launch)),
                                 "algorithmic_bytes_per_env_step": round(alg, 1),
                                 "llc_resident": bool(output_bytes_per_step(task, zones, n) <= LLC_BYTES)}}
        out["full_launch"] = block(lib_chunk, K, dt, "every")
        out["full_launch"].update({"us_per_step_runs": [round(d / K * 1e6, 3) for d in runs],
                                   "scripted_kernel_same_snapshot_us_per_step": round(ms_scripted / K * 1e3, 3),
                                   "replay_matches": "bit-identical" if match else "MISMATCH",
                                   "episodes_ended_in_replay": ep_rec - ep0,
                                   "actions": "the scripted greedy policy's, recorded step by step from the same snapshot"})
        # the fixed-length-skill shape on the same buffer (trajectories differ from the recording: finished envs wait)
        env.set_state(s0)
        warm()
        env.sync()
        n_chunks = K // skill_len
        t0 = time.perf_counter()
        for c in range(n_chunks):
            env.step_many(None, reset="last", actions_ptr=(ptr + 8 * n * skill_len * c, skill_len))
        env.sync()
        dt10 = time.perf_counter() - t0
        out[f"skill_len_{skill_len}"] = block(skill_len, n_chunks * skill_len, dt10, "last")
        out["timing"] = "host wall clock between two stream synchronisations around the replay"
        return out
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def per_step_rate(env, task, zones, policy, env_index0, workload, steps=2000, warm=2000):
    """Side measurement (never `value`): the same envs with ONE kernel launch per step (k_step_lane, the
    path an externally supplied action takes), right after the timed region, clocks still settled."""
    try:
        env.rollout(warm, policy, policy_seed=0x5EED, env_index0=env_index0, mode="per_step")
        ms, _ = env.rollout(steps, policy, policy_seed=0x5EED, env_index0=env_index0, mode="per_step")
        blk = roofline_block(task, zones, env.num_envs, ms / steps / 1e3, 1, False,
                             load_pmc(workload, env.num_envs, "per_step"))
        blk.update({"us_per_step": round(ms / steps * 1e3, 2), "steps": steps,
                    "env_steps_per_s": round(env.num_envs * steps / (ms * 1e-3), 1),
                    "frac_of_hbm_peak": blk["frac"]})
        return blk
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def other_workloads(Z, policy, lib_chunk, device, n_env=65536):
    """aux.workloads: BASELINE.json configs 2-3 (TimedTSP-25, ColourMatch-6) and the reference-faithful 15-zone map,
    each as the headline is measured -- steady state of the persistent kernel (every dispatch timed), the per-step
    kernel, and the 16-env spot check against the oracle over everything that ran."""
    from combinatorial_rl_tasks_amd import sharding
    out = {}
    for w in ("TimedTSP-25", "ColourMatch-6", "PointTSP-15"):
        try:
            task, zones, keepout = WORKLOADS[w]
            cfg = Z.default_config(task, zones, zones_keepout=keepout)
            shard = sharding.EnvShard(rank=0, world=1, envs_per_rank=n_env)
            env = Z.ZoneVecEnv(cfg, n_env, device=device)
            shard.build_bank(env, EPISODES_PER_ENV, n_threads=min(32, usable_cores()))
            env.reset()
            steady = steady_state(env, task, zones, policy, 0, "persistent", load_pmc(w, n_env, "persistent"), lib_chunk)
            per = per_step_rate(env, task, zones, policy, 0, w)
            total = SETTLE_STEPS + 8192 + 2000 + 2000
            spot = parity_spot_check(env, cfg, shard.first_seeds(), shard.seed_stride, 0, total, policy, EPISODES_PER_ENV)
            env.close()
            out[w] = {"n_env": n_env, "steady_state": compact(steady) if isinstance(steady, dict) else steady,
                      "valu_issue": steady.get("valu_issue") if isinstance(steady, dict) else None,
                      "per_step_launch_mode": compact(per) if isinstance(per, dict) else per,
                      "parity_spot_check": spot, "steps_checked": total}
        except Exception as ex:  # the bench line must still print
            out[w] = f"error: {ex}"
    return out


def beyond_llc(Z, policy, lib_chunk, device, sizes=BEYOND_LLC_SIZES):
    """aux.beyond_llc: PointTSP-25 at batches whose per-step output (N x 637 B) goes from 0.6x to 7x the 256 MiB
    Infinity Cache -- nothing a step writes is still on chip when the next step rewrites it, so the stores do reach
    HBM.  Per size: the persistent kernel (512 steps, every dispatch timed) and the per-step kernel, each with the
    committed PMC bytes of THAT size (profiles/traffic.json, `PointTSP-25@N`), the bare store stream of the same
    footprint on this box, and the oracle spot check.  env i replays map 1 + (i mod 262144).
    `persistent` is ONE launch over the whole batch (zenv_set_rollout_slice 0): the HBM-streaming case the PMC bytes
    belong to; `persistent_sliced` is the library's default for such a batch -- launches over 65 536 envs each, whose
    outputs stay in the Infinity Cache while they are rewritten."""
    task, zones, keepout = WORKLOADS["PointTSP-25"]
    out = []
    for n in sizes:
        try:
            cfg = Z.default_config(task, zones, zones_keepout=keepout)
            env = Z.ZoneVecEnv(cfg, n, device=device)
            seeds, stride, period = replay_bank(env, n, BEYOND_LLC_BANK)
            env.reset()
            scale = n / 65536.0
            settle = max(64, int(SETTLE_STEPS / scale))
            env.set_rollout_slice(0)           # ONE launch over the whole batch: every step's outputs stream to HBM
            steady = steady_state(env, task, zones, policy, 0, "persistent", load_pmc("PointTSP-25", n, "persistent"),
                                  lib_chunk, steps=512, settle=settle, slice_envs=0)
            env.set_rollout_slice(DEFAULT_SLICE)   # the library's default: slices of 65 536 envs (a slice's outputs stay on chip)
            sliced = steady_state(env, task, zones, policy, 0, "persistent", None, lib_chunk, steps=512, settle=64,
                                  slice_envs=DEFAULT_SLICE)
            per = per_step_rate(env, task, zones, policy, 0, "PointTSP-25", steps=200, warm=50)
            total = settle + 512 + 64 + 512 + 50 + 200
            spot = parity_spot_check(env, cfg, seeds, stride, 0, total, policy, period)
            env.close()
            ent = {"n_env": n, "parity_spot_check": spot, "steps_checked": total,
                   "store_stream": store_stream_ceiling(Z, task, zones, n, device)}
            for key, blk in (("persistent", steady), ("persistent_sliced", sliced), ("per_step", per)):
                if isinstance(blk, dict):
                    c = compact(blk)
                    c["output_bytes_per_step"] = blk["output_bytes_per_step"]
                    c["us_per_65536_envs"] = round(blk["kernel_us_per_step"] / scale, 3)
                    ss = ent["store_stream"]
                    if isinstance(ss, dict) and ss.get("best_GBps"):
                        c["frac_of_store_stream"] = round(blk["achieved"] / ss["best_GBps"], 4)
                    ent[key] = c
                else:
                    ent[key] = blk
            out.append(ent)
        except Exception as ex:  # the bench line must still print
            out.append({"n_env": n, "error": str(ex)})
    return out


def host_roundtrip_rate(env, steps=24):
    """Side measurement (never `value`): the PCIe-inclusive rate of the legacy host-policy surface -- what
    ParallelEnv.step costs a CPU-resident policy: actions H2D + one step-kernel launch + obs / zone_obs / reward /
    done / goal_met D2H, per step, from and into page-locked host memory (zenv_host_alloc / zenv_get_many)."""
    try:
        import combinatorial_rl_tasks_amd as Z
        n = env.num_envs
        fields = (Z.F_OBS, Z.F_ZONE_OBS, Z.F_REWARD, Z.F_DONE, Z.F_GOAL_MET)
        dtypes = (np.float32, np.float32, np.float32, np.uint8, np.uint8)
        a = env.pinned_array((n, 2), np.float32)
        a[:] = 0
        bufs = [env.pinned_array(env._shape(f), t) for f, t in zip(fields, dtypes)]
        for _ in range(4):
            env.step(a, auto_reset=True)
            env.results_into(fields, bufs)
        t0 = time.perf_counter()
        for _ in range(steps):
            env.step(a, auto_reset=True)
            env.results_into(fields, bufs)
        dt = (time.perf_counter() - t0) / steps
        mb = (a.nbytes + sum(b.nbytes for b in bufs)) / 1e6
        return {"ms_per_step": round(dt * 1e3, 3), "env_steps_per_s": round(n / dt, 1), "pcie_mb_per_step": round(mb, 1),
                "pcie_gb_per_s": round(mb / dt / 1e3, 1), "host_memory": "page-locked", "steps": steps}
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def mlp_policy_rate(env, zones, steps=200):
    """Side measurement (never `value`): the same envs stepped with the reference's actor network (ZoneEnvModel +
    PolicyNetwork, h = 185, random weights) as the on-device policy: network kernels + the
    per-step env kernel per step.  The reference's modules are torch float32, so the figure that stands for "the
    reference's network" is the REFERENCE-GRADE one -- ZENV_MLP_F16X3, mu / std / value within 3e-6 of torch float32
    (tests/test_gpu_mlp.py), what load_mlp() takes by default -- with the plain float32 mode beside it.  The bf16 / f16
    kernels compute in narrower arithmetic than the reference (within 4e-2 / 1e-3): reported as reduced precision."""
    try:
        import combinatorial_rl_tasks_amd as Z
        F, h = env.zone_feat, 185
        rs = np.random.RandomState(0)

        def lin(n_out, n_in):
            w = rs.standard_normal((n_out, n_in)).astype(np.float32)
            return w / np.sqrt((w * w).sum(1, keepdims=True)), (0.1 * rs.standard_normal(n_out)).astype(np.float32)
        t = {}
        for name, shape in (("zone", (h, 8 + F)), ("zone2", (h, h)), ("zone3", (h, h)), ("comb", (h, 8 + h)),
                            ("enc", (h, h)), ("mu", (2, h)), ("std", (2, h))):
            w, b = lin(*shape)
            key = {"zone": ("zone_w1", "zone_b1"), "zone2": ("zone_w2", "zone_b2"), "zone3": ("zone_w3", "zone_b3")}.get(
                name, (name + "_w", name + "_b"))
            t[key[0]], t[key[1]] = w, b
        n = env.num_envs
        flop = n * (zones * 2 * ((8 + F) * h + h * h) + 2 * (h * h + (8 + h) * h + h * h + 4 * h))

        def rate(prec, k):
            env.load_mlp(t, precision=prec)
            env.rollout(max(8, k // 4), Z.POLICY_MLP_SAMPLE, policy_seed=1)
            ms, _ = env.rollout(k, Z.POLICY_MLP_SAMPLE, policy_seed=1)
            return ms / k * 1e3
        env.load_mlp(t)                                   # the default: reference grade
        default_mode = env.mlp_precision
        us_ref = rate("f16x3", steps)
        us_f32 = rate("f32", 24)
        us_bf16x3 = rate("bf16x3", 40)
        us_bf16 = rate("bf16", 300)
        us_f16 = rate("f16", 200)
        env.load_mlp(t)
        return {"reference_grade_us_per_step": round(us_ref, 1),
                "reference_grade_env_steps_per_s": round(n / (us_ref * 1e-6), 1),
                "reference_grade_mode": "ZENV_MLP_F16X3: hi/lo split float16 operands, 3 products per k-step, float32 "
                                        "accumulation -- mu / std / value within 3e-6 of torch float32",
                "load_mlp_default": default_mode,
                "f32_mode_us_per_step": round(us_f32, 1), "f32_mode_env_steps_per_s": round(n / (us_f32 * 1e-6), 1),
                "bf16x3_mode_us_per_step": round(us_bf16x3, 1),
                "network_gflop_per_step": round(flop / 1e9, 1),
                "reference_grade_tflops_3_products": round(3 * flop / (us_ref * 1e-6) / 1e12, 1),
                "reduced_precision": {
                    "note": "narrower arithmetic than the reference's float32 modules (opt-in; not the reference's network)",
                    "bf16_us_per_step": round(us_bf16, 1), "bf16_env_steps_per_s": round(n / (us_bf16 * 1e-6), 1),
                    "bf16_tolerance_vs_torch_f32": 4e-2, "f16_us_per_step": round(us_f16, 1),
                    "f16_tolerance_vs_torch_f32": 1e-3,
                    "bf16_network_tflops_incl_env_step": round(flop / (us_bf16 * 1e-6) / 1e12, 1)},
                "mfma_peak_tflops": 2500.0,
                # a bare v_mfma_f32_32x32x16_bf16 chain on every SIMD with random operands: the power controller
                # holds 1.71 GHz (scripts/probes/mfma_clock.hip), i.e. this, not the spec figure, is reachable
                "mfma_sustained_random_operands_tflops": 1647.0}
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def parity_spot_check(env, cfg, first_seeds, seed_stride, env_index0, total_steps, policy, seed_period, n=16):
    """First 16 envs of this rank vs the oracle over every step the handle has executed."""
    try:
        import combinatorial_rl_tasks_amd as Z
        from oracle import oracle as O
        from tests.helpers import oracle_config_from
        ref = O.rollout(oracle_config_from(O, cfg), np.asarray(first_seeds)[:n], total_steps, policy,
                        seed_stride=seed_stride, policy_seed=0x5EED,
                        env_index0=env_index0, n_threads=4, seed_period=seed_period)
        if env.num_envs <= (1 << 18):
            o, zo, ep = env.get(Z.F_OBS)[:n], env.get(Z.F_ZONE_OBS)[:n], env.get(Z.F_EPISODES)[:n]
        else:           # a multi-GB batch: fetch the first rows only (device pointers + a small copy)
            o, zo, ep = (env.get_head(f, n) for f in (Z.F_OBS, Z.F_ZONE_OBS, Z.F_EPISODES))
        ok = (np.array_equal(o, ref["obs"]) and np.array_equal(zo, ref["zone_obs"])
              and np.array_equal(ep, ref["episodes"]))
        return "bit-identical" if ok else "MISMATCH"
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


if __name__ == "__main__":
    main()
