#!/bin/bash
# A/B of diagnostic builds of the LeNet whole-net kernel on one box: each argument is a comma-separated list of extra -D defines ("base" = none beyond KN_ABLATION).
#   gpurun -- 'bash tools/chain_variants.sh base KN_CHAIN_THIN_SEQ > gpurun_out/r06/chain_variants.txt 2>&1'
# Variant libraries are built under /tmp and never left in the tree; with KN_STAMPS=1 the per-phase stamp table of every variant is printed too.
set -u
cd "$(dirname "$0")/.."
for V in "$@"; do
  D="'KN_ABLATION'"
  if [ "$V" != "base" ]; then for d in ${V//,/ }; do D="$D, '$d'"; done; fi
  python3 -c "from keynet_amd import build; build.build(out='/tmp/libkn_$V.so', defines=($D,))" || continue
  for rep in 1 2; do KEYNET_HIP_LIB=/tmp/libkn_$V.so python3 tools/chain_time.py 1024 "$V"; done
  if [ "${KN_STAMPS:-0}" = "1" ]; then
    KEYNET_HIP_LIB=/tmp/libkn_$V.so KN_CHAIN_STAMPS=/tmp/chain_stamps.bin python3 tools/chain_run.py 1024 > /dev/null
    python3 tools/chain_stamps_report.py /tmp/chain_stamps.bin
  fi
done
