# cfg5 (float32) under different chunk geometries of the marching kernels (WT_CHAIN_SMAX / WT_CHAIN_LANES), one box, three rounds: no effect (5.77-5.87 ms)
for rep in 1 2 3; do
  for cfg in "default" "WT_CHAIN_SMAX=128 WT_CHAIN_LANES=262144" "WT_CHAIN_SMAX=128 WT_CHAIN_LANES=524288" "WT_CHAIN_SMAX=32 WT_CHAIN_LANES=1048576" "WT_CHAIN_SMAX=256 WT_CHAIN_LANES=131072"; do
    if [ "$cfg" = default ]; then r=$(python bench.py --config cfg5 --no-cpu --brief --steps 10 | cut -c1-120); else r=$(env $cfg python bench.py --config cfg5 --no-cpu --brief --steps 10 | cut -c1-120); fi
    echo "$cfg: $r"
  done
done
