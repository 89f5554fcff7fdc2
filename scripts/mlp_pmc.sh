#!/bin/bash
# PMC passes over the policy-forward kernel of one precision (run on the GPU box): scripts/mlp_pmc.sh f16x3 [N]
# Writes gpurun_out/mlp_pmc_<precision>/pass*/p_results.db (one sqlite database per pass) and prints per-dispatch sums
# for the network kernel (scripts/mlp_pmc_summary.py).
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
python scripts/mlp_pmc_summary.py "$OUT"
