for mode in 16 8 0 16; do
  echo "== KN_EXACT_PIPE=$mode"
  for cfg in "64 64 112" "256 256 56" "512 512 28" "512 512 14"; do set -- $cfg
    KN_EXACT_PIPE=$mode timeout 300 python3 tools/conv_bench.py --cin $1 --cout $2 --hw $3 --perm --exact --iters 5 2>&1 | grep "EXACT"
  done
done
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
