set -o pipefail
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/final/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/final/pytest_gpu.log
{ echo "# python tests/soak_parity.py 8192 6000 (round 2, final binaries)"; python tests/soak_parity.py 8192 6000; echo; echo "# python tests/soak_parity.py 65536 3000"; python tests/soak_parity.py 65536 3000; echo; echo "# python tests/soak_collect.py"; python tests/soak_collect.py; } > gpurun_out/final/parity_soak.log 2>&1 || { tail -20 gpurun_out/final/parity_soak.log; exit 1; }
grep -c "bit-identical" gpurun_out/final/parity_soak.log; grep -i "mismatch" gpurun_out/final/parity_soak.log | head -3
: > gpurun_out/final/bench_all_workloads.jsonl
for w in PointTSP-25 TimedTSP-25 ColourMatch-6 PointTSP-15; do python bench.py --workload $w --no-cpu-baseline --no-mlp 2>/dev/null | tail -1 >> gpurun_out/final/bench_all_workloads.jsonl; done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/final/bench_steps20.json
python - <<'PY'
import json
for l in open('gpurun_out/final/bench_all_workloads.jsonl'):
    d=json.loads(l); s=d['aux']['steady_state']; p=d['aux']['per_step_launch_mode']
    print(d['config']['workload'].split(',')[0], 'value %.2f G'%(d['value']/1e9), 'steady %.3f us (%.3f)'%(s['kernel_us_per_step'], s['frac']), 'per-step %.2f us (%.3f)'%(p['us_per_step'], p['frac']), d['aux']['parity_spot_check'])
d=json.loads(open('gpurun_out/final/bench_steps20.json').read()); print('steps20', d['value']/1e9, d['roofline']['kernel_us_per_step'], d['roofline']['frac'], d['aux'].get('mlp_policy',{}).get('us_per_step'), d['aux'].get('mlp_policy',{}).get('f32_mode_us_per_step'))
PY
