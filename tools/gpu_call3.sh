B="python bench.py --no-cpu --brief --steps 20"
for hw in "8192 8192" "16384 8192" "32768 8192" "65536 8192" "4096 16384" "2048 32768" "8192 7424" "8192 8352" "8192 9280"; do set -- $hw; echo "H=$1 W=$2: $($B --rows $1 --size $2)"; done
for d in 1 2 3 4; do echo "debug $d: $(WT_FUSED_DEBUG=$d $B)"; done
