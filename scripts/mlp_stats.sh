#!/bin/bash
# usage: scripts/mlp_stats.sh <tag>   (on the GPU box) -- rocprofv3 kernel stats of the closed loop with the network as the
# policy (scripts/mlp_bench.py), reference-grade split mode and the reduced-precision bf16 mode.
tag=${1:-r04}; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for prec in f16x3 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/mlp_$prec -- python scripts/mlp_bench.py 65536 $prec > $out/mlp_${prec}_bench.log 2>&1
  cp $out/mlp_$prec/*/*kernel_stats.csv $out/mlp_${prec}_rollout_kernel_stats.csv
  head -6 $out/mlp_${prec}_rollout_kernel_stats.csv | cut -c1-150
done
