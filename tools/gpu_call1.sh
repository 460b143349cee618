# round-1 re-entry: full GPU suite + bench + REAL multi-rank RCCL check on the one GPU
mkdir -p gpurun_out/c1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c1/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/c1/bench.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/check_rccl_ranks.py > gpurun_out/c1/rccl2.log 2>&1; echo "rccl2 rc=$?"; tail -5 gpurun_out/c1/rccl2.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 tools/check_rccl_ranks.py --shape 2048 1300 > gpurun_out/c1/rccl4.log 2>&1; echo "rccl4 rc=$?"; tail -5 gpurun_out/c1/rccl4.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 2 --shared-gpu --size 8192 --steps 5 --warmup 2 --spinup 0 > gpurun_out/c1/bench2.json 2> gpurun_out/c1/bench2.err; echo "bench2 rc=$?"; cut -c1-300 gpurun_out/c1/bench2.json; tail -3 gpurun_out/c1/bench2.err
