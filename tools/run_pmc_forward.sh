#!/bin/bash
# roofline.traffic, one definition (tools/pmc_forward.py): two separate counter passes over ONE marked forward of the bench workload.
#   gpurun --timeout 1500 -- 'bash tools/run_pmc_forward.sh gpurun_out/r06/pmc vgg16'
#   python3 tools/pmc_forward.py gpurun_out/r06/pmc/fetch gpurun_out/r06/pmc/write gpurun_out/r06/pmc/forward.json > profiles/r06_vgg16_b256_traffic.json
set -u
R=${1:-gpurun_out/pmc}
WL=${2:-vgg16}
EXTRA=${3:-}
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
cd /tmp
B="--workload $WL --no-cpu-baseline --no-secondary $EXTRA"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/fetch" -- python3 "$REPO/bench.py" $B --pmc-forward "$R/forward.json" > /dev/null 2> "$R/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/write" -- python3 "$REPO/bench.py" $B --pmc-forward "$R/forward_w.json" > /dev/null 2> "$R/write.log"
cd "$REPO"
python3 tools/pmc_forward.py "$R/fetch" "$R/write" "$R/forward.json" > "$R/traffic.json" 2> "$R/traffic.err"
find "$R" -name '*.csv' -size +30M -delete
tail -3 "$R/fetch.log" "$R/traffic.err"; head -c 1500 "$R/traffic.json"
