mkdir -p gpurun_out/q
timeout -k 10 900 python -m pytest tests/test_gpu_strips.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/q/pytest2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/q/pytest2.log
R="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
for ov in 0 1; do
  if [ $ov = 0 ]; then export WT_NO_OVERLAP=1; else unset WT_NO_OVERLAP; fi
  timeout -k 10 300 $R --master-port 2954$ov bench.py --gpus 2 --shared-gpu --size 8192 --steps 10 --warmup 2 --spinup 0 > gpurun_out/q/b2_$ov.json 2> gpurun_out/q/b2_$ov.err; echo "overlap=$ov rc=$?"
  python - <<PY
import json
j=json.load(open("gpurun_out/q/b2_$ov.json")); print(j["value"], j["ms_per_step"], {k:(v["calls_per_step"],v["avg_ms"]) for k,v in j["kernels"].items()})
PY
done
unset WT_NO_OVERLAP
echo "1-GPU: $(python bench.py --no-cpu --brief --steps 20)"
