#!/bin/bash
# Where the filled-in order-preserving kernel's cycles go, per kernel form (tools/fill_bench.py's synthetic operator): rate table, then counter passes per form
# (separate rocprofv3 runs, --kernel-trace + --pmc only).   gpurun -- 'bash tools/fill_pmc.sh gpurun_out/r06/fill 256'
set -u
R=${1:-gpurun_out/fill}
N=${2:-256}
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
python3 -c "from keynet_amd import build; build.build(out='/tmp/libkn_abl.so', defines=('KN_ABLATION',))" || exit 1
export KEYNET_HIP_LIB=/tmp/libkn_abl.so
python3 tools/fill_bench.py $N 2>&1 | grep -v amdgpu.ids | tee "$R/rates_$N.txt"
cd /tmp
for FORM in 2 4 3; do
  i=0
  for SET in "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_MFMA"; do
    i=$((i+1))
    KN_FILL_FORM=$FORM timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$R/f${FORM}p$i" -- python3 "$REPO/tools/fill_bench.py" $N once > "$R/f${FORM}p$i.log" 2>&1
    echo "form $FORM pass $i:"; python3 "$REPO/tools/pmc_dump.py" "$R/f${FORM}p$i" | grep -E "exact_fill|^kernel" | tail -3
  done
done
find "$R" -name '*.csv' -size +8M -delete
