#!/bin/bash
# Counter passes over the float32 bilateral march alone (tools/bench_bil.py) for library variants: what does the
# kernel wait for?  Usage: tools/pmc_bil.sh OUTDIR [variants/x.so ...]   (the in-tree library is always first)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/pmc_bil}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$R/$l; fi
  tag=$(basename $l .so)
  i=0
  while read -r group; do
    [ -z "$group" ] && continue
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $OUT/$tag/g$i -- \
        python3 $R/tools/bench_bil.py 8192 2 > $OUT/$tag.g$i.log 2>&1 || echo "$tag group $i failed" >> $OUT/errors.log
  done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr
GROUPS
  (cd $R && python3 tools/pmc_table.py $OUT/$tag > $OUT/$tag.summary.csv 2>> $OUT/errors.log)
  rm -rf $OUT/$tag/g*/*/*agent_info.csv
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
tabs = {}
for f in sorted(glob.glob(os.path.join(out, "*.summary.csv"))):
    tag = os.path.basename(f)[:-12]
    tabs[tag] = {r["counter"]: float(r["avg_per_dispatch"]) for r in csv.DictReader(open(f)) if "bilateral2" in r["kernel"]}
names = sorted(set().union(*[set(t) for t in tabs.values()]))
with open(os.path.join(out, "compare.csv"), "w") as fo:
    fo.write("counter," + ",".join(tabs) + "\n")
    for n in names:
        fo.write(n + "," + ",".join(f"{tabs[t].get(n, float('nan')):.0f}" for t in tabs) + "\n")
print(open(os.path.join(out, "compare.csv")).read())
PY
