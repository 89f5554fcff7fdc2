#!/bin/bash
# usage: scripts/stats_other_workloads.sh <tag>   (on the GPU box) -- rocprofv3 kernel stats of the per-step kernel for
# the workloads profile_round.sh's stats runs do not cover (they profile the headline workload).
tag=${1:-r04}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in TimedTSP-25 ColourMatch-6 PointTSP-15; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_per_step_$w -- python bench.py --workload $w --no-cpu-baseline --no-mlp --no-steady --no-sweep --mode per_step --steps 2000 --warmup 2000 > $out/stats_per_step_${w}_bench.json 2>/dev/null
  cp $out/stats_per_step_$w/*/*kernel_stats.csv $out/stats_per_step_${w}_kernel_stats.csv
  grep k_step_lane $out/stats_per_step_${w}_kernel_stats.csv | cut -c1-60,200-400
done
