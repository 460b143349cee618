for i in 1 2 3; do timeout 120 python bench.py --steps 20 --brief 2>&1 | tail -1; done
