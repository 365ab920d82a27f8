#!/bin/bash
# HBM traffic / speed trade of the conv-taps matrix-core kernel on the keyed VGG-16 forward (256 images), one variant per run:
#   ball  = output pixels per breadth-first ball of the processing order (default 64: what one XCD's resident workgroups cover)
#   occ   = workgroups per CU cap of the 128 x 128 launches (default: the rule of launch_conv, 4 or 3)
# Per variant: rocprofv3 --kernel-trace --pmc FETCH_SIZE (KiB; doubled per the guide's gfx950 note) and, separately, WRITE_SIZE over one warm forward;
# the conv-taps time is the sum of the kernel-trace durations of the same launches.  Diagnostic build (-DKN_ABLATION) under /tmp.
#   gpurun --timeout 2400 -- 'bash tools/conv_traffic_ablation.sh gpurun_out/r05_abl > gpurun_out/r05_conv_traffic_ablation.txt 2>&1'
set -u
OUT=${1:-gpurun_out/conv_abl}
REPO=$(pwd)
mkdir -p "$REPO/$OUT"
export TMPDIR=/tmp
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from keynet_amd import build
print(build.build(out='/tmp/libkeynet_hip_abl.so', defines=('KN_ABLATION',)))
PY
export KEYNET_HIP_LIB=/tmp/libkeynet_hip_abl.so
B="--steps 2 --warmup 1 --layer-iters 1 --no-cpu-baseline --no-exact-leg --no-secondary"
cd /tmp
# VARIANTS="base" restricts the table to the shipped setting (the like-for-like row that roofline.traffic -- tools/pmc_forward.py -- must agree with)
for V in ${VARIANTS:-base KN_CONV_BALL=16 KN_CONV_BALL=32 KN_CONV_BALL=128 KN_CONV_BALL=256 KN_OCC=3 KN_OCC=2}; do
  T=$(echo $V | tr '=' '_')
  if [ "$V" != "base" ]; then export $V; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$REPO/$OUT/$T-$C" -- python3 "$REPO/bench.py" $B > "$REPO/$OUT/$T-$C.json" 2> "$REPO/$OUT/$T-$C.log"
  done
  if [ "$V" != "base" ]; then unset ${V%%=*}; fi
done
cd "$REPO"
python3 tools/conv_traffic_table.py "$OUT"
find "$OUT" -name '*.csv' -size +20M -delete
