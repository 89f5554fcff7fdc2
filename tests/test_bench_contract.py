"""bench.py's arithmetic, on the CPU: the algorithmic byte counts of SURVEY.md 8(d), the two byte bases of the
roofline block, the PMC record scaled to the steps of a launch, the VALU-issue fraction -- and that the committed
profiles/traffic.json carries a record for every workload and launch mode the bench can be asked for."""
import json
import os

import pytest

import bench

pytest.importorskip("numpy")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_are_survey_8d():
    assert bench.algorithmic_bytes(0, 25, 1) == 1247          # PointTSP-25:   521 + 726
    assert bench.algorithmic_bytes(1, 25, 1) == 1447          # TimedTSP-25:   621 + 826
    assert bench.algorithmic_bytes(2, 6, 1) == 485            # ColourMatch-6: 204 + 281
    assert bench.algorithmic_bytes(0, 15, 1) == 827           # PointTSP-15:   351 + 476
    # a launch of K steps publishes the outputs K times and moves the state once
    assert bench.algorithmic_bytes(0, 25, 256) == pytest.approx(637 + (521 + 89) / 256)
    assert round(bench.algorithmic_bytes(0, 25, 256), 1) == 639.4


def test_roofline_block_is_self_consistent():
    pmc = {"hbm_bytes_per_step": 39.27e6, "hbm_bytes_fixed_per_launch": 81.0e6, "valu_insts_per_simd_step": 1767.2,
           "gpu_cycles_per_step": 11440.1, "gpu_clock_ghz": 1.989, "source": "test"}
    b = bench.roofline_block(0, 25, 65536, 5.24e-6, 256, True, pmc)
    assert b["peak"] == 8000.0 and b["bound"] == "hbm" and b["unit"] == "GB/s"
    assert b["achieved"] == pytest.approx(639.4 * 65536 / 5.24e-6 / 1e9, rel=1e-3)
    assert b["frac"] == pytest.approx(b["achieved"] / 8000.0, abs=1e-4)
    assert b["kernel_avg_us"] == pytest.approx(5.24 * 256, rel=1e-6) and b["env_steps_per_launch"] == 65536 * 256
    assert b["algorithmic_bytes_per_launch"] == pytest.approx(639.4 * 65536 * 256, rel=1e-4)
    # the traffic is that of THIS launch length, and traffic / duration stays a bandwidth below the peak
    assert b["traffic"] == pytest.approx(81.0e6 + 256 * 39.27e6, rel=1e-9)
    assert b["traffic"] / (b["kernel_avg_us"] * 1e-6) < 8.0e12
    short = bench.roofline_block(0, 25, 65536, 5.9e-6, 20, True, pmc)
    assert short["traffic"] == pytest.approx(81.0e6 + 20 * 39.27e6, rel=1e-9)
    assert short["traffic"] / (short["kernel_avg_us"] * 1e-6) < 8.0e12
    # both byte bases: the per-step formula does not apply to a persistent launch, and is THE base of a one-step launch
    assert b["byte_bases"]["survey_8d"]["bytes_per_env_step"] == 1247 and not b["byte_bases"]["survey_8d"]["applies"]
    one = bench.roofline_block(0, 25, 65536, 13.6e-6, 1, False, None)
    assert one["byte_bases"]["survey_8d"]["applies"] and one["byte_bases"]["outputs_only"]["bytes_per_env_step"] == 1247
    assert one["traffic"] is None and one["valu_issue"] is None and one["kernel"] == "k_step_lane"
    v = b["valu_issue"]
    assert v["frac_pmc"] == pytest.approx(4 * 1767.2 / 11440.1, abs=1e-4)
    assert v["frac_live"] == pytest.approx(4 * 1767.2 / (5.24e3 * 1.989), abs=1e-3)


def test_committed_pmc_record_covers_every_workload_and_mode():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        t = json.load(f)
    for w, (task, zones, _) in bench.WORKLOADS.items():
        for mode in ("persistent", "per_step"):
            pmc = bench.load_pmc(w, 65536, mode)
            assert pmc is not None and pmc == t[f"{w}@65536"][mode], (w, mode)
            for key in ("hbm_bytes_per_step", "hbm_bytes_fixed_per_launch", "valu_insts_per_simd_step",
                        "gpu_cycles_per_step", "gpu_clock_ghz", "source", "method"):
                assert key in pmc, (w, mode, key)
            assert os.path.exists(os.path.join(ROOT, os.path.dirname(pmc["source"]))), pmc["source"]
            # measured traffic stays near the algorithmic bytes (no wasted re-reads).  The per-step kernel of the
            # smallest workload reads 1.28x: its two waves both load the pose and the action (56 B per env), and
            # FETCH_SIZE's gfx950 doubling is calibrated for 16-byte-per-lane loads only, not for its byte-wide
            # cooldown loads (MI355X_MICROARCH.md, HBM: other widths are uncalibrated)
            alg = bench.algorithmic_bytes(task, zones, 256 if mode == "persistent" else 1) * 65536
            # (round 3: 1.37x for ColourMatch-6 per step -- + the reset hint and the derived bank rows; its FETCH_SIZE is
            # 11.9 MB raw against 15.5 MB of loads the kernel issues, i.e. the x2 overstates these narrow loads)
            assert pmc["hbm_bytes_per_step"] < (1.05 if mode == "persistent" else 1.40) * alg, (w, mode)
            # ... and a VALU-issue fraction is a fraction
            assert 0.05 < 4 * pmc["valu_insts_per_simd_step"] / pmc["gpu_cycles_per_step"] < 1.0
    assert bench.load_pmc("PointTSP-25", 12345, "persistent") is None


def test_committed_pmc_record_was_measured_on_the_kernels_in_the_tree():
    """profiles/traffic.json names the sha256 of the kernel sources it was measured on: a kernel change without a
    new PMC pass (scripts/profile_round.sh + summarize_profile.py) fails here instead of silently keeping old bytes."""
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        meta = json.load(f).get("_meta", {})
    assert meta.get("kernel_sources") == list(bench.KERNEL_SOURCES)
    assert meta.get("kernel_sources_sha256") == bench.kernel_sources_sha(), \
        "csrc/ changed since the PMC passes of profiles/traffic.json: re-run scripts/profile_round.sh on the GPU box"
    assert not bench.traffic_is_stale()


def test_beyond_llc_sizes_have_committed_counters():
    """Every batch of aux.beyond_llc carries PMC bytes measured AT THAT SIZE."""
    task, zones, _ = bench.WORKLOADS["PointTSP-25"]
    for n in bench.BEYOND_LLC_SIZES:
        assert bench.output_bytes_per_step(task, zones, n) > 0.5 * bench.LLC_BYTES
        for mode in ("persistent", "per_step"):
            pmc = bench.load_pmc("PointTSP-25", n, mode)
            assert pmc is not None and pmc.get("hbm_bytes_per_step"), (n, mode)
            # real traffic per step: at least the row stream, at most 10 % over the algorithmic bytes (the per-step
            # kernel reads float32 zone positions where SURVEY 8(d) counts the float64 centres: 0.84 of the figure)
            alg = bench.algorithmic_bytes(task, zones, 256 if mode == "persistent" else 1) * n
            assert 0.80 * alg < pmc["hbm_bytes_per_step"] < 1.10 * alg, (n, mode, pmc["hbm_bytes_per_step"] / alg)
    assert bench.output_bytes_per_step(task, zones, bench.BEYOND_LLC_SIZES[-1]) > 4 * bench.LLC_BYTES


def test_llc_residency_flag():
    b = bench.roofline_block(0, 25, 65536, 5.24e-6, 256, True, None)
    assert b["llc_resident"] and b["output_bytes_per_step"] == 65536 * 637 and b["llc_bytes"] == 256 << 20
    big = bench.roofline_block(0, 25, 1048576, 88e-6, 256, True, None)
    assert not big["llc_resident"] and big["output_bytes_per_step"] == 1048576 * 637


def test_diagnostic_switches_are_recorded_and_refused(monkeypatch, zenv_mod):
    """a ZENV_* kernel switch or a variant build must show in the line, and bench.py refuses to
    time it unless asked (--experiment)."""
    import argparse
    nat = zenv_mod._native
    args = argparse.Namespace(override=[])
    for k in [k for k in os.environ if k.startswith("ZENV_")]:
        monkeypatch.delenv(k)
    clean = bench.experiment_switches(nat, args)
    assert clean["active"] is False and clean["build_flags"] == "" and clean["rollout_chunk"] == nat.ROLLOUT_CHUNK
    assert clean["env"] == {}
    monkeypatch.setenv("ZENV_MLP_F32_VALU", "1")
    monkeypatch.setenv("ZENV_BENCH_FORCE_DIST", "1")        # harness switch: not an experiment
    monkeypatch.setenv("ZENV_CPU_THREADS", "4")
    sw = bench.experiment_switches(nat, args)
    assert sw["active"] and sw["env"] == {"ZENV_MLP_F32_VALU": "1"}
    monkeypatch.delenv("ZENV_MLP_F32_VALU")
    assert bench.experiment_switches(nat, argparse.Namespace(override=["frameskip=1"]))["active"]


def test_bench_refuses_an_experiment_without_the_flag(zenv_mod):
    import subprocess
    import sys
    env = dict(os.environ, ZENV_ROLLOUT_CHUNK_EXP="16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing to time a diagnostic configuration" in (r.stderr + r.stdout)
    assert "ZENV_ROLLOUT_CHUNK_EXP" in (r.stderr + r.stdout)


def test_build_stamp_tracks_the_flag_set(monkeypatch, zenv_mod):
    """A diagnostic .so (ZENV_EXTRA_FLAGS) does not pass for the shipped build, and the other way round."""
    import combinatorial_rl_tasks_amd.build as B
    monkeypatch.delenv("ZENV_EXTRA_FLAGS", raising=False)
    assert B._up_to_date()                                    # the session fixture built the plain library
    monkeypatch.setenv("ZENV_EXTRA_FLAGS", "-DZENV_EXP=1")
    assert B.extra_flags() == "-DZENV_EXP=1" and not B._up_to_date()


def test_bare_gpus_n_spawns_its_own_ranks(zenv_mod):
    """`python bench.py --gpus N` without a launcher starts N fresh rank processes itself -- RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them, ONE rendezvous directory (with an attempt
    nonce) for all of them -- instead of dying at argument parsing.  (CPU: the ranks only print their launch environment.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "ZENV_RDZV_DIR",
                                                             "ZENV_RDZV_NONCE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--print-rank-env"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ranks = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == [0, 1, 2] == [d["local_rank"] for d in ranks]
    assert all(d["world"] == 3 and d["ipc_legacy"] == "0" and d["master"][0] == "127.0.0.1" for d in ranks)
    assert len({d["rdzv_dir"] for d in ranks}) == 1 and len({d["ppid"] for d in ranks}) == 1
    assert len({tuple(d["master"]) for d in ranks}) == 1
    assert not os.path.exists(os.path.dirname(ranks[0]["rdzv_dir"]))      # the launcher cleaned up behind its ranks
    # without devices for its ranks (this container has none) the job ends at once, with a reason, and a non-zero code
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    if zenv_mod._native.lib().zenv_device_count() < 2:
        assert r.returncode != 0 and "has no device" in r.stderr and "one process per GPU" in r.stderr
