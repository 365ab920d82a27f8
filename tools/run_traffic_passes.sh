#!/bin/bash
# roofline.traffic of the two workloads that quote it, on the CURRENT kernel sources (bench.py quotes a committed pass only when its csrc sha256 matches):
#   gpurun --timeout 1500 -- 'bash tools/run_traffic_passes.sh gpurun_out/r06/traffic'
#   cp gpurun_out/r06/traffic/vgg16.json profiles/r06_vgg16_b256_traffic.json; cp gpurun_out/r06/traffic/allconv.json profiles/r06_allconv_b4096_traffic.json
set -u
R=${1:-gpurun_out/traffic}
REPO=$(pwd)
mkdir -p "$REPO/$R"
R="$REPO/$R"
export TMPDIR=/tmp
cd /tmp
for WL in vgg16 allconv; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$R/$WL-$C" -- python3 "$REPO/bench.py" --workload $WL --no-cpu-baseline --no-secondary --pmc-forward "$R/$WL-$C.json" > /dev/null 2> "$R/$WL-$C.log"
  done
done
cd "$REPO"
python3 tools/pmc_forward.py "$R/vgg16-FETCH_SIZE" "$R/vgg16-WRITE_SIZE" "$R/vgg16-FETCH_SIZE.json" > "$R/vgg16.json"
python3 tools/pmc_forward.py "$R/allconv-FETCH_SIZE" "$R/allconv-WRITE_SIZE" "$R/allconv-FETCH_SIZE.json" csr_group_mfma_kernel convexact > "$R/allconv.json"
find "$R" -name '*.csv' -size +30M -delete
python3 -c "
import json
for w in ('vgg16','allconv'):
    t=json.load(open('$R/%s.json' % w)); print(w, {k:t[k] for k in ('dominant_launches','fetch_x2','write','traffic_ratio','csrc_sha256','mode')})"
