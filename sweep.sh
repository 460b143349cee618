timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|mismatch" | tail -8
timeout 300 python tools/bench_configs.py 5 w 2>&1 | head -40
