#!/bin/bash
# rocprofv3 evidence for one bench workload (kernel stats + separate PMC passes, --kernel-trace only):
#   gpurun --timeout 1500 -- 'bash tools/run_profiles_wl.sh gpurun_out/p_lenet lenet'
#   python3 tools/pmc_table.py gpurun_out/p_lenet > profiles/r03_lenet_pmc.csv
set -u
R=${1:?out dir}
WL=${2:?workload}
shift 2
EXTRA="$*"
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
# (LeNet: one forward is 37 us -- 25 steps are 1 ms, the GPU has not left its idle clock by then: 2 000 warm-up steps = 75 ms, as the default bench's secondary leg does)
STEPS=20; WARM=5; SSTEPS=5; SWARM=2
if [ "$WL" = lenet ]; then STEPS=2000; WARM=2000; SSTEPS=500; SWARM=500; fi
python3 bench.py --workload $WL --steps $STEPS --warmup $WARM --no-secondary --detail "$R/bench_detail.json" $EXTRA > "$R/bench.json" 2> "$R/bench.log"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/stats" -- python3 "$REPO/bench.py" --workload $WL --steps $SSTEPS --warmup $SWARM --no-cpu-baseline --no-secondary --no-exact-leg $EXTRA > "$R/stats_bench.json" 2> "$R/stats_bench.log"
rocprofv3 --kernel-trace --output-format csv -d "$R/trace" -- python3 "$REPO/bench.py" --workload $WL --no-cpu-baseline --no-secondary --trace-layers "$R/layers.json" $EXTRA > /dev/null 2> "$R/trace.log"
PMC_BENCH="--workload $WL --steps 1 --warmup 0 --layer-iters 1 --no-cpu-baseline --no-exact-leg --no-secondary $EXTRA"
# (passes 1 and 2 -- FETCH_SIZE, WRITE_SIZE -- run ONE marked forward: bench.py --pmc-forward, condensed by tools/pmc_forward.py into roofline.traffic)
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  MARK=""; if [ $i -le 2 ]; then MARK="--pmc-forward $R/forward$i.json"; fi
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$R/p$i" -- python3 "$REPO/bench.py" $PMC_BENCH $MARK > "$R/p$i.json" 2> "$R/p$i.log"
done
cd "$REPO"
python3 tools/trace_layers.py "$R/trace" "$R/layers.json" > "$R/per_layer_trace.csv"
python3 tools/pmc_table.py "$R" > "$R/pmc.csv"
find "$R" -name '*.csv' -size +20M -delete
ls "$R"; head -30 "$R/per_layer_trace.csv"
