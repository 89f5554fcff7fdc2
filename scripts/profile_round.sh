#!/bin/bash
# usage: scripts/profile_round.sh <tag>   (run on the GPU box through gpurun)
# Produces under gpurun_out/<tag>/: the bench JSON line, rocprofv3 kernel stats of the same command,
# and the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the guide prescribes) for both the
# persistent rollout kernel (default mode) and the per-step kernel.
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline > $out/stats_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_per_step -- python bench.py --no-cpu-baseline --mode per_step --steps 2000 --warmup 2000 > $out/stats_per_step_bench.json 2>/dev/null
for mode in persistent per_step; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$mode -- python bench.py --mode $mode --steps 512 --warmup 256 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$mode -- python bench.py --mode $mode --steps 512 --warmup 256 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
done
python scripts/summarize_profile.py $out
