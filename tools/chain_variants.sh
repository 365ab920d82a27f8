#!/bin/bash
# A/B of the LeNet whole-net kernel on one box with the diagnostic build (-DKN_ABLATION, built on demand under /tmp, never left in the tree).  Each argument is a variant:
#   base                       the shipped configuration
#   env:KN_CHAIN_NO_SEQ=1      a diagnostic option read at operator create (comma-separated list of assignments)
#   def:KN_CHAIN_DV=12         extra -D defines (comma-separated): a separate build
#   gpurun -- 'KN_STAMPS=1 bash tools/chain_variants.sh base env:KN_CHAIN_NO_SEQ=1 > gpurun_out/r06/chain_variants.txt 2>&1'
# With KN_STAMPS=1 the per-phase stamp table of every variant is printed too.
set -u
cd "$(dirname "$0")/.."
python3 -c "from keynet_amd import build; build.build(out='/tmp/libkn_abl.so', defines=('KN_ABLATION',))" || exit 1
for V in "$@"; do
  LIB=/tmp/libkn_abl.so
  ENVS=""
  case "$V" in
    def:*) D="'KN_ABLATION'"; for d in $(echo "${V#def:}" | tr ',' ' '); do D="$D, '$d'"; done
           LIB="/tmp/libkn_$(echo "$V" | tr -c 'A-Za-z0-9' '_').so"
           python3 -c "from keynet_amd import build; build.build(out='$LIB', defines=($D,))" || continue ;;
    env:*) ENVS=$(echo "${V#env:}" | tr ',' ' ') ;;
  esac
  for rep in 1 2; do env $ENVS KEYNET_HIP_LIB=$LIB python3 tools/chain_time.py 1024 "$V"; done
  if [ "${KN_STAMPS:-0}" = "1" ]; then
    env $ENVS KEYNET_HIP_LIB=$LIB KN_CHAIN_STAMPS=/tmp/chain_stamps.bin python3 tools/chain_run.py 1024 > /dev/null
    python3 tools/chain_stamps_report.py /tmp/chain_stamps.bin
  fi
done
