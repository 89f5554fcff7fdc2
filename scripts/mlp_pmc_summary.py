"""Per-dispatch sums of the counters scripts/mlp_pmc.sh collected for the policy-network kernel (rocprofv3 writes one
sqlite database per pass): python scripts/mlp_pmc_summary.py gpurun_out/mlp_pmc_f16x3 > profiles/rNN/mlp_f16x3_pmc.txt"""
import glob
import sqlite3
import sys

out = sys.argv[1]
vals, name, dur = {}, None, None
for f in sorted(glob.glob(out + "/pass*/*_results.db")):
    db = sqlite3.connect(f)
    for kernel, counter, total, n in db.execute(
            "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by 1, 2"):
        if "mlp_zone" in kernel:
            vals[counter], name = total / n, kernel.split("(")[0]
    if dur is None:
        row = db.execute("select avg(duration), count(*) from kernels where name like '%mlp_zone%'").fetchone()
        if row and row[0]:
            dur = row[0]
print("# rocprofv3 --pmc passes (scripts/mlp_pmc.sh) over `python scripts/mlp_bench.py N PRECISION short`, kernel", name)
print("# values per dispatch, summed over the chip's SEs / XCDs")
for k, v in vals.items():
    print("%-32s %16.0f" % (k, v))
need = ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
        "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE")
if all(k in vals for k in need):
    waves = 1024      # one wave per SIMD at N = 65 536
    w = vals["SQ_WAVE_CYCLES"] * 4 / waves
    print("# derived (%d waves; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles):" % waves)
    print("#   cycles per wave %.0f = issuing %.0f + issue-stalled (matrix pipe busy / dependencies) %.0f + parked in s_waitcnt %.0f"
          % (w, vals["SQ_ACTIVE_INST_ANY"] * 4 / waves, vals["SQ_WAIT_INST_ANY"] * 4 / waves, vals["SQ_WAIT_ANY"] * 4 / waves))
    print("#   matrix instructions per SIMD %.0f x 32 cycles = %.0f cycles = %.2f of the wave's lifetime"
          % (vals["SQ_INSTS_MFMA"] / waves, vals["SQ_VALU_MFMA_BUSY_CYCLES"] / waves, vals["SQ_VALU_MFMA_BUSY_CYCLES"] / waves / w))
    print("#   GRBM_GUI_ACTIVE / 8 XCDs = %.0f cycles per dispatch%s" % (
        vals["GRBM_GUI_ACTIVE"] / 8, "" if not dur else "; kernel duration under the counters %.1f us -> %.2f GHz" % (
            dur / 1e3, vals["GRBM_GUI_ACTIVE"] / 8 / dur)))
    print("#   vector instructions per matrix instruction %.2f; LDS array busy %.2f of the kernel" % (
        (vals["SQ_INSTS_VALU"] - vals["SQ_INSTS_MFMA"]) / vals["SQ_INSTS_MFMA"],
        vals["SQ_LDS_IDX_ACTIVE"] / 256 / (vals["GRBM_GUI_ACTIVE"] / 8)))
