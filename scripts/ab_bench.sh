#!/bin/bash
# same-box A/B of two builds of the library: scripts/ab_bench.sh <libA.so> <libB.so> [workloads...]
# (a library built from another commit's csrc/ may not match this tree's oracle: ignore its parity spot check)
A=$1; B=$2; shift 2
wl=${@:-PointTSP-25 TimedTSP-25 ColourMatch-6 PointTSP-15}
for w in $wl; do
  for lib in $A $B $A $B; do
    ZENV_LIB_PATH=$lib python bench.py --workload $w --no-cpu-baseline --no-mlp --steps 2048 --warmup 512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['aux']['steady_state']; p=d['aux']['per_step_launch_mode']
print('%-14s %-28s persistent %.3f us/step (frac %.3f)  per-step %.2f us (frac %.3f)  %s' % ('$w', '$lib'.split('/')[-1], s['kernel_us_per_step'], s['frac'], p['us_per_step'], p['frac'], d['aux']['parity_spot_check']))"
  done
done
