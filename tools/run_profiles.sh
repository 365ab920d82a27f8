#!/bin/bash
# Produces the raw material of profiles/rNN_* on a GPU box (run through gpurun), then tools/make_profiles.py condenses it:
#   gpurun --timeout 2400 -- 'bash tools/run_profiles.sh gpurun_out/r1e'
#   python3 tools/make_profiles.py gpurun_out/r1e r01
# Counter passes are separate runs with --kernel-trace only (never combined with sys/runtime traces).
set -u
R=${1:-gpurun_out/prof}
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --detail "$R/bench_detail.json" > "$R/bench.json" 2> "$R/bench.log"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/stats" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$R/stats_bench.json" 2> "$R/stats_bench.log"
rocprofv3 --kernel-trace --output-format csv -d "$R/trace" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-secondary --trace-layers "$R/layers.json" > /dev/null 2> "$R/trace.log"
PMC_BENCH="--steps 1 --warmup 0 --layer-iters 1 --no-cpu-baseline --no-exact-leg --no-secondary"
# roofline.traffic, ONE definition (round 6): the launches of exactly one forward between two marker kernels (bench.py --pmc-forward), FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/pmc_fetch" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-secondary --pmc-forward "$R/forward.json" > /dev/null 2> "$R/pmc_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/pmc_write" -- python3 "$REPO/bench.py" --no-cpu-baseline --no-secondary --pmc-forward "$R/forward_w.json" > /dev/null 2> "$R/pmc_write.log"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d "$R/pmc_mfma" -- python3 "$REPO/bench.py" $PMC_BENCH > "$R/pmc_mfma.json" 2> "$R/pmc_mfma.log"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/pmc_l2" -- python3 "$REPO/bench.py" $PMC_BENCH > "$R/pmc_l2.json" 2> "$R/pmc_l2.log"
cd "$REPO"
python3 tools/pmc_layers.py "$R" > "$R/per_layer.csv"
python3 tools/trace_layers.py "$R/trace" "$R/layers.json" > "$R/per_layer_trace.csv"
# keep the merge-back small: the per-dispatch traces are large, the condensed CSVs are what make_profiles reads
find "$R" -name '*.csv' -size +40M -delete
ls -la "$R" "$R"/*/* | head -60
