#!/bin/bash
# diagnostic: the persistent kernel at 1, 2, 4, ... steps per launch (ZENV_ROLLOUT_CHUNK_EXP) vs the per-step kernel
for w in ${@:-PointTSP-25 ColourMatch-6}; do
  for c in 1 2 4 16 256; do
    ZENV_ROLLOUT_CHUNK_EXP=$c python bench.py --workload $w --no-cpu-baseline --no-mlp --no-steady --steps 2048 --warmup 512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['aux']['per_step_launch_mode']
print('%-14s chunk %3d: loop %.2f us/step, dispatch %.2f us/step | per-step kernel %.2f us' % ('$w', $c, r['loop_us_per_step'], r['kernel_us_per_step'], p['us_per_step']))"
  done
done
