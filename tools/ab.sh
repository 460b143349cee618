# A/B of library builds on ONE box (hosts differ by +-10 %): bash tools/ab.sh libA.so libB.so ... [-- bench args]
libs=(); args=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; args=("$@"); break; fi; libs+=("$1"); shift; done
for rep in 1 2 3; do for l in "${libs[@]}"; do
  echo "$(basename $l): $(WATROO_HIP_LIB=$PWD/$l python bench.py --no-cpu --brief --steps 30 "${args[@]}")"
done; done
