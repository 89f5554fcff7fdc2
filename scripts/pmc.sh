#!/bin/bash
# one rocprofv3 PMC pass over a short bench run, averaged per kernel.
# usage: scripts/pmc.sh "<COUNTER ...>" <kernel-name-substring> [bench args]
ctrs=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.build()' || exit 1   # never compile under the profiler's preload
rm -rf gpurun_out/pmc_tmp
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_tmp -- python bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-kernel-events "$@" > /dev/null 2>&1
python - "$kern" <<'PY'
import csv, glob, collections, sys
f = glob.glob("gpurun_out/pmc_tmp/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[1] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-28s mean %16.1f  over %d dispatches" % (k, sum(v) / len(v), len(v)))
PY
