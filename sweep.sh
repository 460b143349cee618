for sg in 512 2048 16384 65536 1000000; do echo -n "sumgrid=$sg : "; WT_SUM_GRID=$sg WT_FUSED_NW=4 timeout 120 python bench.py --steps 20 --brief 2>&1 | tail -1; done
