#!/bin/bash
set -u
R=$(pwd)/${1:-gpurun_out/sqx}
REPO=$(pwd)
mkdir -p "$R"
export TMPDIR=/tmp
cd /tmp
for cfg in "256 256 56" "512 512 14"; do
  set -- $cfg
  tag="x$1_$2_$3"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d "$R/a_$tag" -- python3 "$REPO/tools/conv_bench.py" --cin $1 --cout $2 --hw $3 --perm --iters 3 --exact > "$R/a_$tag.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$R/b_$tag" -- python3 "$REPO/tools/conv_bench.py" --cin $1 --cout $2 --hw $3 --perm --iters 3 --exact > "$R/b_$tag.log" 2>&1
  echo "== $tag" >> "$R/sq.txt"
  python3 "$REPO/tools/pmc_dump.py" "$R/a_$tag" | grep exact | tail -1 >> "$R/sq.txt"
  python3 "$REPO/tools/pmc_dump.py" "$R/b_$tag" | grep exact | tail -1 >> "$R/sq.txt"
done
find "$R" -name '*.csv' -delete
cat "$R/sq.txt"
