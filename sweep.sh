timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do timeout 120 python bench.py --steps 20 --brief 2>&1 | tail -1; done
WT_FUSED_DEBUG=3 timeout 120 python bench.py --steps 20 --brief 2>&1 | tail -1
