#!/bin/bash
# Phase timestamps of the LeNet whole-net kernel (diagnostic build with -DKN_ABLATION, built on demand under /tmp: never left in the tree).
#   bash tools/chain_stamps.sh [n_images ...]   ->  gpurun_out/chain_stamps_<n>.txt
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "from keynet_amd import build; build.build(out='/tmp/libkeynet_hip_abl.so', defines=('KN_ABLATION',))"
for n in "${@:-1024}"; do
    KEYNET_HIP_LIB=/tmp/libkeynet_hip_abl.so KN_CHAIN_STAMPS=/tmp/chain_stamps.bin python tools/chain_run.py $n > /dev/null
    python tools/chain_stamps_report.py /tmp/chain_stamps.bin | tee gpurun_out/chain_stamps_$n.txt
    KN_CHAIN_WSTAMPS=1 KEYNET_HIP_LIB=/tmp/libkeynet_hip_abl.so KN_CHAIN_STAMPS=/tmp/chain_stamps.bin python tools/chain_run.py $n > /dev/null
    python tools/chain_stamps_report.py /tmp/chain_stamps.bin > gpurun_out/chain_wstamps_$n.txt
done
