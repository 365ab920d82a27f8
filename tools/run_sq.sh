#!/bin/bash
# SQ stall breakdown of single conv layers (tools/conv_bench.py), two counter passes each; condensed on the box.
set -u
R=$(pwd)/${1:-gpurun_out/sq}
REPO=$(pwd)
mkdir -p "$R"
export TMPDIR=/tmp
cd /tmp
for cfg in "256 256 56" "512 512 14" "64 64 224"; do
  set -- $cfg
  tag="c$1_$2_$3"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$R/a_$tag" -- python3 "$REPO/tools/conv_bench.py" --cin $1 --cout $2 --hw $3 --perm --iters 3 > "$R/a_$tag.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$R/b_$tag" -- python3 "$REPO/tools/conv_bench.py" --cin $1 --cout $2 --hw $3 --perm --iters 3 > "$R/b_$tag.log" 2>&1
  echo "== $tag" >> "$R/sq.txt"
  python3 "$REPO/tools/pmc_dump.py" "$R/a_$tag" | grep convtaps | tail -2 >> "$R/sq.txt"
  python3 "$REPO/tools/pmc_dump.py" "$R/b_$tag" | grep convtaps | tail -2 >> "$R/sq.txt"
done
find "$R" -name '*.csv' -delete
cat "$R/sq.txt"
