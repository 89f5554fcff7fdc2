#!/bin/bash
# usage: scripts/profile_round.sh <tag> [workloads...]   (run on the GPU box through gpurun)
# Produces under gpurun_out/<tag>/: the bench JSON line, rocprofv3 kernel stats of the same command, and -- in
# SEPARATE runs, as MI355X_MICROARCH.md prescribes -- the FETCH_SIZE / WRITE_SIZE / SQ+GRBM PMC passes of the
# persistent rollout kernel and of the per-step kernel for each workload.  summarize_profile.py condenses them
# into <tag>/summary.json and profiles-ready traffic.json.
tag=${1:-r02}; shift
workloads=${@:-PointTSP-25 TimedTSP-25 ColourMatch-6 PointTSP-15}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the profiler's preload initialises the GPU in every child: compile BEFORE it starts (build.py refuses under it)
python -c 'import __graft_entry__ as g; g.build()' || exit 1
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline --no-mlp --no-sweep > $out/stats_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_per_step -- python bench.py --no-cpu-baseline --no-mlp --no-steady --no-sweep --mode per_step --steps 2000 --warmup 2000 > $out/stats_per_step_bench.json 2>/dev/null
echo "stats done"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for w in $workloads; do
  for mode in persistent per_step; do
    common="--workload $w --mode $mode --steps 512 --warmup 256 --no-cpu-baseline --no-kernel-events --no-mlp --no-steady --no-sweep --no-per-step"
    [ $mode = per_step ] && common="$common --no-settle"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_${w}_$mode -- python bench.py $common > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_${w}_$mode -- python bench.py $common > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $out/pmc_sq_${w}_$mode -- python bench.py $common > /dev/null 2>&1
    echo "pmc $w $mode done"
  done
done
# the action-chunk form of the persistent kernel (zenv_step_many), PointTSP-25
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_chunk -- python scripts/chunk_pmc.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_chunk -- python scripts/chunk_pmc.py > /dev/null 2>&1
echo "pmc action chunk done"
# the batch-size sweep of aux.beyond_llc: HBM bytes of both kernels AT each size (per-step output 0.6x .. 7x the LLC)
for n in ${SWEEP_SIZES:-262144 1048576 3145728}; do
  w="PointTSP-25@$n"
  base="--workload PointTSP-25 --envs-per-gpu $n --bank-maps 262144 --rollout-slice 0 --no-settle --no-cpu-baseline --no-kernel-events --no-mlp --no-steady --no-sweep --no-per-step"
  for mode in persistent per_step; do
    [ $mode = persistent ] && common="$base --mode persistent --warmup 128 --steps 512" || common="$base --mode per_step --warmup 16 --steps 64"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_${w}_$mode -- python bench.py $common > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_${w}_$mode -- python bench.py $common > $out/pmc_write_${w}_$mode.json 2>/dev/null
    echo "pmc $w $mode done"
  done
done
python scripts/summarize_profile.py $out
