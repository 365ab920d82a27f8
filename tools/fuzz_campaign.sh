#!/bin/bash
# A longer fuzzing run than the test tier's, on seeds of its own:   gpurun --timeout 3000 -- 'bash tools/fuzz_campaign.sh gpurun_out/fuzz 7'   (second argument: first seed)
OUT=${1:-gpurun_out/fuzz}; S0=${2:-1}
mkdir -p "$OUT"
for spec in "chain 600" "csr 400" "convtaps 900" "tiled 500" "models 200" "floatmodels 120" "dense 200" "factored 80"; do
  set -- $spec
  for k in 0 1; do
    seed=$((S0 + k))
    timeout 1200 python3 tests/test_fuzz_gpu.py $1 $2 $seed > "$OUT/$1.$seed.log" 2>&1
    echo "$1 seed $seed: rc=$? $(tail -1 "$OUT/$1.$seed.log" | cut -c1-160)"
    grep -h "MISMATCH\|fault\|Traceback\|PATH off\|SPLIT APPLICATION\|PLANES" "$OUT/$1.$seed.log" | head -5
  done
done
