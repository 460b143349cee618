#!/bin/bash
# What kind of box is this?  Streaming ceilings (tools/membench.hip), the bench's two passes, and
# the clocks / power the SMU reports while the bench runs.  Output: gpurun_out/probe/probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/probe; O=gpurun_out/probe/probe.txt; : > $O
rocm-smi --showuniqueid 2>/dev/null | grep -i unique | head -1 >> $O
hipcc -O3 --offload-arch=gfx950 -o /tmp/membench tools/membench.hip 2>> $O
/tmp/membench 8192 2>&1 | sort -t: -k2 -n | awk '{k=$1" "$2; if (!(k in seen)) {seen[k]=1; print}}' >> $O
for i in 1 2; do python bench.py --brief --steps 30 --no-build --no-cpu >> $O 2>&1; done
D=$(ls -d /sys/class/drm/card*/device 2>/dev/null | head -1)
python bench.py --steps 20000 --no-build --no-cpu --brief > /tmp/long.txt 2>&1 &
BP=$!
for i in $(seq 60); do
  sleep 0.25
  s=$(grep '\*' $D/pp_dpm_sclk 2>/dev/null | tr -d '\n'); m=$(grep '\*' $D/pp_dpm_mclk 2>/dev/null | tr -d '\n'); f=$(grep '\*' $D/pp_dpm_fclk 2>/dev/null | tr -d '\n')
  p=$(cat $D/hwmon/hwmon*/power1_average 2>/dev/null | head -1)
  echo "t=$i sclk[$s] mclk[$m] fclk[$f] power_uW[$p]" >> /tmp/clk.txt
  kill -0 $BP 2>/dev/null || break
done
wait $BP
sort -t'[' -k2 /tmp/clk.txt | uniq -c -f1 | sort -rn | head -8 >> $O
cat /tmp/long.txt >> $O
cat $O
