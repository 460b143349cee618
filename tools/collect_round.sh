#!/bin/bash
# After `gpurun -- bash tools/profile_round.sh` has merged gpurun_out/final back: copy the summaries
# the judge reads into profiles/<tag>_* and rebuild profiles/traffic.json (PMC bytes per launch, tied
# to the kernel sources' digest).     usage: bash tools/collect_round.sh r04_a
set -e
cd "$(dirname "$0")/.."
tag=${1:?tag}
F=gpurun_out/final
cp $F/bench.json profiles/${tag}_bench.json
cp $F/pytest_gpu.log profiles/${tag}_pytest_gpu.log
for c in headline cfg2 cfg3 cfg5; do
  cp $(ls $F/prof_$c/*/*_kernel_stats.csv | head -1) profiles/${tag}_kernel_stats_$c.csv
  side=8192; [ $c = cfg2 ] && side=4096
  cfgarg=$c
  python tools/pmc_summary.py $F profiles/${tag}_$c $side $cfgarg pmc_fetch_$c pmc_write_$c > /dev/null
done
cp $(ls $F/prof_cfg5_f64/*/*_kernel_stats.csv | head -1) profiles/${tag}_kernel_stats_cfg5_f32_f64.csv
python tools/pmc_table.py $F/pmc_cfg5_f64 > profiles/${tag}_cfg5_f32_f64_pmc_fetch_write.csv      # KiB per dispatch; HBM read = 2 x FETCH_SIZE on gfx950
cp $F/pmc_cfg5_sq/clocks.csv profiles/${tag}_cfg5_clocks.csv
cp $F/pmc_cfg5_sq/summary.csv profiles/${tag}_cfg5_sq_counters.csv
cp gpurun_out/parity_errors.log profiles/${tag}_parity_errors.log 2>/dev/null || true
python - <<'PY'
import json
d = json.load(open("profiles/traffic.json"))
print("traffic.json:", d["_meta"]["source_digest"][:12], d["_meta"]["image"], len(d) - 1, "entries")
PY
