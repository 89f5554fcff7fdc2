#!/bin/bash
# L2 hit rate of the step kernel (rocprofv3 PMC pass).  usage: scripts/pmc_l2.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.build()' || exit 1   # never compile under the profiler's preload
rm -rf gpurun_out/pmc_l2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_l2 -- python bench.py --steps 100 --no-cpu-baseline --no-kernel-events "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_l2/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "k_step_lane" in r["Kernel_Name"]:
        agg["step"][r["Counter_Name"]].append(float(r["Counter_Value"]))
h = sum(agg["step"]["TCC_HIT_sum"]) / len(agg["step"]["TCC_HIT_sum"]); m = sum(agg["step"]["TCC_MISS_sum"]) / len(agg["step"]["TCC_MISS_sum"])
print("k_step_lane: TCC_HIT %.0f  TCC_MISS %.0f  hit rate %.3f" % (h, m, h / (h + m)))
PY
