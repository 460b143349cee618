set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s25
for n in 4096 8192 16384; do
  for sc in 0 1; do
    echo "nrows=$n WT_SCATTER_STRIPS=$sc" >> gpurun_out/s25/split.txt
    WT_SCATTER_STRIPS=$sc timeout -k 10 200 python tools/bench_split.py $n 32768 2>&1 | grep -E "whole|reserve=16" >> gpurun_out/s25/split.txt
  done
done
cat gpurun_out/s25/split.txt
