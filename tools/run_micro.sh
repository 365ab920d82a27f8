#!/bin/bash
# Raw output of every micro-benchmark DESIGN.md quotes (tools/micro/*.hip), with the box's clock under each load:
#   gpurun --timeout 600 -- 'bash tools/run_micro.sh gpurun_out/r05_micro'   ->  one .txt per program; copy them to profiles/r05_micro_*.txt
set -u
OUT=${1:-gpurun_out/micro}
mkdir -p "$OUT"
for src in tools/micro/*.hip; do
    name=$(basename "$src" .hip)
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off "$src" -o "/tmp/micro_$name" 2> "$OUT/$name.build.log" || { echo "build failed: $name"; continue; }
    {
        echo "# $src  ($(date -u +%Y-%m-%dT%H:%MZ), $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 'Marketing Name' | sed 's/.*: *//'))"
        echo "# sclk before: $(/opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -m1 -i sclk | sed 's/.*: *//')"
        timeout 120 "/tmp/micro_$name"
        echo "# exit code $?"
    } > "$OUT/$name.txt" 2>&1
    rm -f "$OUT/$name.build.log"
    echo "== $name"; tail -n 30 "$OUT/$name.txt"
done
