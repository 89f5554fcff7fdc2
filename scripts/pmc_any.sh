#!/bin/bash
# one rocprofv3 PMC pass over any python script of this repo, averaged per kernel.
# usage: scripts/pmc_any.sh "<COUNTER ...>" <kernel-name-substring> <script.py> [args]
ctrs=$1; kern=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.build()' || exit 1   # never compile under the profiler's preload
rm -rf gpurun_out/pmc_tmp
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_tmp -- python "$@" > /dev/null 2>&1
python - "$kern" <<'PY'
import csv, glob, collections, sys
fs = glob.glob("gpurun_out/pmc_tmp/*/*counter_collection.csv")
if not fs:
    print("no counter file (unknown counter name?)"); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r: agg["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print("%-32s mean %16.1f  over %d dispatches" % (k, sum(v) / len(v), len(v)))
PY
