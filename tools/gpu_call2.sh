mkdir -p gpurun_out/c2
B="python bench.py --no-cpu --brief --steps 20"
echo "base: $($B)"
for pad in 32 64 128 256 512 1024 2048; do echo "pad $pad: $(WT_PITCH_PAD=$pad $B)"; done
echo "base again: $($B)"
for skew in 0 256 4096 65536; do echo "skew $skew: $(WT_PLANE_SKEW=$skew $B)"; done
for r in 4096 8192 16384; do echo "32768 x rows $r: $($B --size 32768 --rows $r)"; done
echo "4096: $($B --size 4096)"
