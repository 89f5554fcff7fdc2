for ov in "" "--unfused" "--workload PointTSP-15" "--workload ColourMatch-6" "--workload TimedTSP-25"; do
python bench.py --steps 500 --warmup 20 --no-cpu-baseline $ov 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ov', d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_us'], d['roofline']['frac'], d['aux']['parity_spot_check'])"
done
