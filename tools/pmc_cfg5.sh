#!/bin/bash
# SQ counters + kernel trace of the cfg5 flow in float32 and float64 (tools/bench_wow64.py, serial order):
# effective clock of the VALU-bound kernels (GRBM_GUI_ACTIVE / 8 / duration, MI355X_MICROARCH.md "DVFS
# give-back") and their issue utilisation at that clock.  Usage: tools/pmc_cfg5.sh OUTDIR
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/pmc_cfg5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WT_NO_WOW_OVERLAP=1
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/g$i -- \
      python3 $R/tools/bench_wow64.py 8192 2 > $OUT/g$i.log 2>&1 || echo "group $i failed" >> $OUT/errors.log
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
GROUPS
cd $R
python3 tools/pmc_table.py $OUT > $OUT/summary.csv 2>> $OUT/errors.log
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "g1", "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
cnt = defaultdict(dict)
for r in csv.DictReader(open(os.path.join(out, "summary.csv"))):
    cnt[r["kernel"]][r["counter"]] = float(r["avg_per_dispatch"])
with open(os.path.join(out, "clocks.csv"), "w") as fo:
    fo.write("kernel,launches,avg_ms,eff_clock_ghz,valu_inst_quadcycles_per_simd_share,wave_cycles\n")
    for k, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if k not in cnt or "GRBM_GUI_ACTIVE" not in cnt[k]:
            continue
        t = sum(d) / len(d)
        c = cnt[k]
        clk = c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip: / (1024 SIMDs) -> quad-cycles per SIMD
        share = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (clk * 1e9 * t) if clk > 0 else 0
        fo.write(f'"{k[:70]}",{len(d)},{t * 1e3:.4f},{clk:.3f},{share:.3f},{c.get("SQ_WAVE_CYCLES", 0):.0f}\n')
print(open(os.path.join(out, "clocks.csv")).read())
PY
