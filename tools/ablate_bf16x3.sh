#!/bin/bash
# Loop-piece ablation of convtaps_bf16x3_kernel (timing only: ablated launches compute garbage): separate -DKN_ABLATION library.
#   gpurun --timeout 600 -- 'bash tools/ablate_bf16x3.sh > gpurun_out/abl_bf16.txt 2>&1'
set -eu
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from keynet_amd import build
print(build.build(out='/tmp/libkeynet_hip_abl.so', defines=('KN_ABLATION',)))
PY
export KEYNET_HIP_LIB=/tmp/libkeynet_hip_abl.so
for A in 0 256 512 1024 2048 2304 2816 3840; do echo "== KN_ABL=$A"; KN_ABL=$A python3 tools/bf16x3_bench.py 2>&1 | grep "Cin 512 Cout 512 28x28\|Cin 256" | cut -c 95-200; done
