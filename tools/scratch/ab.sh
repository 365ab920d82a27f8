KN_NO_ROW_ORDER=1 timeout 600 python3 bench.py --no-cpu-baseline 2>&1 | grep -E "pool|\"value\"" | cut -c1-200
timeout 600 python3 bench.py --no-cpu-baseline 2>&1 | grep -E "layer|\"value\"" | cut -c1-330
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
