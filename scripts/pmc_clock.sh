#!/bin/bash
# clock of the GPU during each persistent launch: GRBM_GUI_ACTIVE cycles / dispatch duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c 'import __graft_entry__ as g; g.build()' || exit 1   # never compile under the profiler's preload
rm -rf gpurun_out/pmc_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_clk -- python scripts/chunk_times.py "$@" > gpurun_out/pmc_clk.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_clk/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_rollout_lane" in r["Kernel_Name"]]
print("columns:", list(rows[0].keys()))
for r in rows:
    cyc = float(r["Counter_Value"])
    s, e = r.get("Start_Timestamp"), r.get("End_Timestamp")
    if s and e:
        dur = (int(e) - int(s)) * 1e-9
        print("dispatch %s: %.0f cycles (sum over XCDs), %.1f us, %.2f GHz if /8" % (r["Dispatch_Id"], cyc, dur * 1e6, cyc / 8 / dur / 1e9))
    else:
        print("dispatch %s: %.0f cycles" % (r["Dispatch_Id"], cyc))
PY
