#!/bin/bash
# PMC passes over the policy-forward kernel of one precision (run on the GPU box): scripts/mlp_pmc.sh f16x3 [N]
# Writes gpurun_out/mlp_pmc_<precision>/pass*/...counter_collection.csv and prints per-kernel sums for the network kernel.
set -e
P=${1:-f16x3}; N=${2:-65536}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/mlp_pmc_$P
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $OUT/pass$i -o p -- python scripts/mlp_bench.py $N $P short > $OUT/pass$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "mlp" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_COEXEC_CYCLES"):
            seen.add(key)
    # dispatch count per file
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "mlp" in k: ids[k].add(r["Dispatch_Id"])
    for k, v in ids.items(): cnt[k] = max(cnt[k], len(v))
for k in tot:
    print(k, "dispatches", cnt[k])
    for c, v in sorted(tot[k].items()): print("   %-32s %16.0f  per dispatch %14.0f" % (c, v, v / max(cnt[k], 1)))
PY
