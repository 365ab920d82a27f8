#!/bin/bash
# PMC passes (separate runs, --kernel-trace only; no TA_* / TCP_* counters: a pass with them aborted rocprofv3 on this pool) over the LeNet whole-net kernel:  gpurun -- 'bash tools/chain_pmc.sh gpurun_out/chainpmc 1024'
set -u
R=${1:-gpurun_out/chainpmc}
N=${2:-1024}
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$R/p$i" -- python3 "$REPO/tools/chain_run.py" $N > "$R/p$i.log" 2>&1
  python3 "$REPO/tools/pmc_dump.py" "$R/p$i" | grep chain_kernel | tail -2
done
find "$R" -name '*.csv' -size +8M -delete
