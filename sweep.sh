timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 300 python tools/bench_configs.py 2 3 2>&1 | head -40
