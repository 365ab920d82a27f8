#!/bin/bash
# Per-LAYER PMC passes (separate runs, --kernel-trace only) over one forward of the bench workload, condensed on the box:
#   gpurun --timeout 1500 -- 'bash tools/run_pmc_layers.sh gpurun_out/pl0'
set -u
R=${1:-gpurun_out/pl}
shift || true
EXTRA="$*"
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
cd /tmp
PMC_BENCH="--steps 1 --warmup 0 --layer-iters 1 --no-cpu-baseline --no-exact-leg $EXTRA"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/pmc_fetch" -- python3 "$REPO/bench.py" $PMC_BENCH > "$R/pmc_fetch.json" 2> "$R/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/pmc_write" -- python3 "$REPO/bench.py" $PMC_BENCH > "$R/pmc_write.json" 2> "$R/pmc_write.log"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/pmc_mfma" -- python3 "$REPO/bench.py" $PMC_BENCH > "$R/pmc_mfma.json" 2> "$R/pmc_mfma.log"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/pmc_l2" -- python3 "$REPO/bench.py" $PMC_BENCH > "$R/pmc_l2.json" 2> "$R/pmc_l2.log"
cd "$REPO"
python3 tools/pmc_layers.py "$R" > "$R/per_layer.csv"
find "$R" -name '*.csv' -size +8M -delete
tail -n 80 "$R/per_layer.csv"
