# cfg5 with the wow updates on the side stream: priority and occupancy variants (float32 line of bench_wow64)
mkdir -p gpurun_out/ab
for v in "A=1" "WT_NO_WOW_OVERLAP=1" "WT_SIDE_PRIORITY=0" "WT_SIDE_PRIORITY=-1" "WT_BIL_LDS_PAD=43000" "WT_BIL_LDS_PAD=43000 WT_SIDE_PRIORITY=0" "WT_BIL_LDS_PAD=43000 WT_NO_WOW_OVERLAP=1"; do
  echo "== $v: $(env $v timeout -k 10 200 python tools/bench_wow64.py 8192 5 2>&1 | grep 'ms/step' | grep -v launches | grep -v profiled | tr '\n' ' ')"
done
