# quick parity + bench for fused-kernel work
mkdir -p gpurun_out/q
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_strips.py -m gpu -x -q -k "not rccl_ranks and not large_geometry" > gpurun_out/q/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/q/pytest.log
B="python bench.py --no-cpu --brief --steps 20"
echo "8192: $($B)"
echo "8192: $($B)"
echo "4096: $($B --size 4096)"
echo "32768x4096: $($B --size 32768 --rows 4096)"
