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

N > 1: one process per GPU (torch.distributed, backend "nccl" == RCCL).  Envs shard
trivially: rank r owns global envs [r*65536, (r+1)*65536); there is no collective on the
step path; after the rollout the per-env episodic returns are all-gathered over xGMI.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (k_rollout_lane, or k_step_lane with --mode per_step) and `cpu_baseline` (the float64 C oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch first: its bundled libamdhip64 must be the only HIP runtime in the process
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (task, zones, zones_keepout)  -- SURVEY.md 8(d) configs 1-3
    "PointTSP-25": (0, 25, 0.40),
    "TimedTSP-25": (1, 25, 0.40),
    "ColourMatch-6": (2, 6, 0.55),
    "PointTSP-15": (0, 15, 0.55),
}


SETTLE_STEPS = 6000     # untimed steps (settle + warmup) before the timed region, ~40 ms of GPU time
EPISODES_PER_ENV = 4   # depth of the map bank per env; the schedule wraps around it (the oracle too)


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


def load_pmc(workload, n_env, mode):
    """The committed rocprofv3 PMC record of this workload's dominant kernel (profiles/traffic.json, written by
    scripts/summarize_profile.py from separate --pmc passes): HBM bytes as `fixed per launch + per step` (FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is) and the VALU-issue counters."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get(f"{workload}@{n_env}", {}).get("persistent" if mode == "persistent" else "per_step")
    except (OSError, ValueError):
        return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8000)
    ap.add_argument("--warmup", type=int, default=1000, help="untimed steps right before the timed ones")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the untimed clock-settling steps that precede the warm-up (see SETTLE_STEPS)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--workload", default="PointTSP-25", choices=sorted(WORKLOADS))
    ap.add_argument("--policy", default="greedy", choices=["greedy", "uniform"])
    ap.add_argument("--override", action="append", default=[],
                    help="experiment only: config key=value (e.g. frameskip=1); marks the run invalid")
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
    ap.add_argument("--event-stride", type=int, default=16,
                    help="time every k-th step-kernel dispatch with begin/end HIP events (each pair costs "
                         "~6 us of launch path, so timing all of them would distort `value`)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="skip the per-launch HIP events around the step kernel")
    ap.add_argument("--dispatch-events", action="store_true",
                    help="persistent mode: begin/end events on every dispatch of the TIMED region (hipExtLaunchKernel; "
                         "costs ~17 us of wall per call -- scripts/launch_overhead.py) instead of the two events recorded "
                         "on the stream right before and after its launches; the steady-state side run always has them")
    args = ap.parse_args()
    if args.unfused:
        args.mode = "unfused"
    args.unfused = args.mode == "unfused"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Build (or find up to date) the native library BEFORE this process touches the GPU or the process group: a
    # compiler child must never be forked from a GPU-initialised process.  Every rank calls it; build_library()
    # serialises concurrent callers with a file lock and is a no-op when lib/ is newer than csrc/.
    import __graft_entry__ as entry
    entry.build()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    # ZENV_BENCH_REHEARSAL=gloo: the driver's N-rank launch line on a box with fewer GPUs than ranks (tests only: ranks
    # share the cards, the process group runs over gloo because RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("ZENV_BENCH_REHEARSAL") == "gloo"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get("ZENV_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank rehearsal
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import combinatorial_rl_tasks_amd as Z
    from combinatorial_rl_tasks_amd import sharding

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
    episodes_per_env = EPISODES_PER_ENV
    t_bank = time.perf_counter()
    shard.build_bank(env, episodes_per_env, n_threads=min(32, usable_cores()))
    t_bank = time.perf_counter() - t_bank
    env.reset()
    # Untimed: first let the power controller settle (it dips to ~1.5 GHz 4-15 ms after load arrives and
    # is steady after ~20 ms, profiles/r01/clock_per_launch.txt) -- enough steps that settle + warmup
    # is at least SETTLE_STEPS -- then the W warm-up steps, then the K timed steps, back to back.
    args.settle = 0 if args.no_settle else max(0, SETTLE_STEPS - args.warmup)
    for k in (args.settle, args.warmup):
        if k > 0:
            env.rollout(k, policy, policy_seed=0x5EED, env_index0=shard.env_index0, mode=args.mode)

    def local_sync():
        env.sync()
        torch.cuda.synchronize()

    def fence():
        local_sync()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

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
    local_sync()
    elapsed = time.perf_counter() - t0
    fence()

    # rank-local results, then the one collective of the job: all-gather of episodic returns
    returns = shard.gather_returns(env)          # float32 [world * n_env] on every rank
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        total_env_steps = world * n_env * args.steps
        value = total_env_steps / elapsed
        persistent = args.mode == "persistent" and zones in (5, 6, 10, 15, 20, 25)
        chunk = min(Z._native.ROLLOUT_CHUNK, max(args.steps, 1)) if persistent else 1
        n_launches = (args.steps + chunk - 1) // chunk if persistent else args.steps
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
            roofline = roofline_block(task, zones, n_env, k_step_s, chunk if persistent else 1, persistent, pmc)
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
        spot = parity_spot_check(env, cfg, shard, args, policy)
        steady = None if (args.no_steady or args.override) else \
            steady_state(env, task, zones, policy, shard, args.mode, pmc)
        per_step = per_step_rate(env, task, zones, policy, shard, args.workload) \
            if (args.mode == "persistent" and not distributed) else None
        mlp = None if (args.no_mlp or distributed) else mlp_policy_rate(env, zones)
        host_rt = None if (args.no_mlp or distributed) else host_roundtrip_rate(env)
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is an N = 1 figure (rank 0 only)
            cpu = cpu_baseline(cfg, task, zones, keepout, policy)
        # the figures to compare rounds by ride inside `roofline` as well (compact), beside the timed region's own
        if roofline is not None:
            def compact(b):
                return None if not isinstance(b, dict) else {
                    k: b.get(k) for k in ("kernel", "steps_per_launch", "kernel_us_per_step", "kernel_avg_us", "achieved",
                                          "frac", "algorithmic_bytes_per_env_step", "algorithmic_bytes_per_launch",
                                          "traffic", "steps", "launches", "env_steps_per_s")}
            roofline["steady_state"] = compact(steady)
            roofline["per_step_kernel"] = compact(per_step)
        out = {
            "metric": "env-steps/sec", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}, N_env={n_env} per GPU, num_steps=2000, "
                                   f"zones_keepout={keepout}, policy=pi_{args.policy} (on-device, "
                                   f"launch mode {args.mode}), "
                                   "auto-reset on" + (f" EXPERIMENT {args.override}" if args.override else ""),
                       "n_env_total": world * n_env, "zones": zones,
                       "parallelism": f"env-shard x{world}, all-gather(ep_return) after rollout"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "aux": {"collective": ("nccl all_gather_into_tensor" if distributed and dist.get_backend() == "nccl"
                                   else "none (single process)" if not distributed else dist.get_backend()),
                    "settle_steps_untimed": args.settle, "hip_event_ms_total": round(ms_total, 3), "bank_build_s": round(t_bank, 2),
                    "episodes_finished_rank0": int(ep.sum()),
                    "mean_last_return_all_ranks": float(np.mean(returns[returns != 0]))
                    if (returns != 0).any() else 0.0,
                    "parity_spot_check": spot, "steady_state": steady, "per_step_launch_mode": per_step,
                    "mlp_policy": mlp, "host_policy_roundtrip_pcie_inclusive": host_rt},
        }
        print(json.dumps(out), flush=True)
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return out


def roofline_block(task, zones, n_env, k_step_s, steps_per_launch, persistent, pmc):
    """`roofline` of the bench line for a kernel that takes k_step_s per step in launches of steps_per_launch.

    Two byte bases, both algorithmic: `outputs_only` -- what a launch of K steps must move when the
    state stays in registers: K x the step's outputs + the state once (for K = 1 this IS the second base) --
    and SURVEY.md 8(d)'s per-step figure, which charges the state round trip to every step.  `achieved`/`frac`
    use the first, the one that describes the kernel measured; for the persistent kernel the second is given
    for reference only (it exceeds 1: the kernel does not do that traffic, by design)."""
    alg = algorithmic_bytes(task, zones, steps_per_launch)
    alg1 = algorithmic_bytes(task, zones, 1)
    achieved = alg * n_env / k_step_s / 1e9
    achieved1 = alg1 * n_env / k_step_s / 1e9
    traffic = traffic_for_launch(pmc, steps_per_launch)
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_note": (None if traffic is None else
                         f"committed PMC bytes (FETCH_SIZE x2 + WRITE_SIZE) scaled to this launch of "
                         f"{steps_per_launch} step(s); {pmc.get('source')}"),
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
        "frac_of_measured_copy_ceiling": round(achieved / 6290.0, 4),   # 6.29 TB/s float4 copy, MI355X_MICROARCH.md
        "valu_issue": valu_issue(pmc, k_step_s * 1e6),
    }


def steady_state(env, task, zones, policy, shard, mode, pmc, steps=8192):
    """Side measurement (never `value`): the SAME kernel, same envs, right after the timed region, over enough
    steps that launch overheads and the clock transient are out of the picture -- every dispatch timed with its
    own begin/end HIP events.  This is the figure to compare rounds by."""
    try:
        persistent = mode == "persistent" and zones in (5, 6, 10, 15, 20, 25)
        env.rollout(SETTLE_STEPS, policy, policy_seed=0x5EED, env_index0=shard.env_index0, mode=mode)   # untimed
        ms, ms_k = env.rollout(steps, policy, policy_seed=0x5EED, env_index0=shard.env_index0, mode=mode,
                               time_step_kernel=persistent)
        import combinatorial_rl_tasks_amd as Z
        chunk = Z._native.ROLLOUT_CHUNK if persistent else 1
        k_step_s = (ms_k if persistent else ms / steps) / 1e3
        blk = roofline_block(task, zones, env.num_envs, k_step_s, chunk, persistent, pmc)
        blk.update({"steps": steps, "launches": (steps + chunk - 1) // chunk,
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


def per_step_rate(env, task, zones, policy, shard, workload, steps=2000):
    """Side measurement (never `value`): the same envs with ONE kernel launch per step (k_step_lane, the
    path an externally supplied action takes), right after the timed region, clocks still settled."""
    try:
        env.rollout(2000, policy, policy_seed=0x5EED, env_index0=shard.env_index0, mode="per_step")
        ms, _ = env.rollout(steps, policy, policy_seed=0x5EED, env_index0=shard.env_index0, mode="per_step")
        blk = roofline_block(task, zones, env.num_envs, ms / steps / 1e3, 1, False,
                             load_pmc(workload, env.num_envs, "per_step"))
        blk.update({"us_per_step": round(ms / steps * 1e3, 2), "steps": steps,
                    "env_steps_per_s": round(env.num_envs * steps / (ms * 1e-3), 1),
                    "frac_of_hbm_peak": blk["frac"]})
        return blk
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


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


def mlp_policy_rate(env, zones, steps=300):
    """Side measurement (never `value`): the same envs stepped with the reference's actor network
    (ZoneEnvModel + PolicyNetwork, h = 185, random weights) as the on-device policy -- two bf16 MFMA
    kernels + the per-step env kernel per step."""
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
        env.load_mlp(t)
        env.rollout(steps, Z.POLICY_MLP_SAMPLE, policy_seed=1)
        ms, _ = env.rollout(steps, Z.POLICY_MLP_SAMPLE, policy_seed=1)
        n = env.num_envs
        flop = n * (zones * 2 * ((8 + F) * h + h * h) + 2 * (h * h + (8 + h) * h + h * h + 4 * h))
        us = ms / steps * 1e3
        # the float32 mode of the same network (ZENV_MLP_F32: the reference's arithmetic, for evaluation)
        env.load_mlp(t, precision="f32")
        env.rollout(5, Z.POLICY_MLP_SAMPLE, policy_seed=1)
        ms32, _ = env.rollout(20, Z.POLICY_MLP_SAMPLE, policy_seed=1)
        return {"us_per_step": round(us, 1), "f32_mode_us_per_step": round(ms32 / 20 * 1e3, 1), "env_steps_per_s": round(n * steps / (ms * 1e-3), 1),
                "network_gflop_per_step": round(flop / 1e9, 1), "dtype": "bf16 MFMA, f32 accumulate",
                "network_tflops_incl_env_step": round(flop / (us * 1e-6) / 1e12, 1), "mfma_peak_tflops": 2500.0,
                # a bare v_mfma_f32_32x32x16_bf16 chain on every SIMD with random operands: the power controller
                # holds 1.71 GHz (scripts/probes/mfma_clock.hip), i.e. this, not the spec figure, is reachable
                "mfma_sustained_random_operands_tflops": 1647.0}
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


def parity_spot_check(env, cfg, shard, args, policy):
    """First 16 envs of rank 0 vs the oracle over the whole warmup+timed rollout."""
    try:
        import combinatorial_rl_tasks_amd as Z
        from oracle import oracle as O
        from tests.helpers import oracle_config_from
        n = 16
        T = args.settle + args.warmup + args.steps
        ref = O.rollout(oracle_config_from(O, cfg), shard.first_seeds()[:n], T, policy,
                        seed_stride=shard.seed_stride, policy_seed=0x5EED,
                        env_index0=shard.env_index0, n_threads=4, seed_period=EPISODES_PER_ENV)
        ok = (np.array_equal(env.get(Z.F_OBS)[:n], ref["obs"])
              and np.array_equal(env.get(Z.F_ZONE_OBS)[:n], ref["zone_obs"])
              and np.array_equal(env.get(Z.F_EPISODES)[:n], ref["episodes"]))
        return "bit-identical" if ok else "MISMATCH"
    except Exception as ex:  # the bench line must still print
        return f"error: {ex}"


if __name__ == "__main__":
    main()
