#!/bin/bash
# Loop-piece ablation of the conv-taps MFMA kernel on the real VGG-16 layers (timing only: the ablated launches compute garbage).
# Builds a SEPARATE library with -DKN_ABLATION (the product library carries none of these tests) and times, in one process per layer
# set, the variants KN_ABL = bit 0 no chunk barrier | 1 no LDS stores | 2 no global loads | 4 no pointer walk | 5 no tap loads |
# 6 no activation loads.      gpurun --timeout 900 -- 'bash tools/ablate_conv.sh > gpurun_out/ablate.txt 2>&1'
set -eu
REPO=$(pwd)
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from keynet_amd import build
print(build.build(out='/tmp/libkeynet_hip_abl.so', defines=('KN_ABLATION',)))
PY
export KEYNET_HIP_LIB=/tmp/libkeynet_hip_abl.so
python3 tools/ab_layers.py --layers ${1:-conv1_2,conv3_2,conv4_2} --rounds 3 \
  --variants "base;KN_ABL=1;KN_ABL=2;KN_ABL=4;KN_ABL=16;KN_ABL=32;KN_ABL=64;KN_ABL=96;KN_ABL=7" 2>&1 | grep -v "amdgpu.ids"
