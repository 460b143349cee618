#!/bin/bash
# Calibration of the "VALU issue share" metric of tools/pmc_cfg5.sh: the same counters over tools/valubench (pure
# register-to-register instruction streams: what does a kernel that does NOTHING but issue read?) and over the
# float32 cfg5 flow in serial order.  Usage: tools/pmc_valu.sh OUTDIR
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/${1:-gpurun_out/pmc_valu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WT_NO_WOW_OVERLAP=1
G="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g1 -- $R/tools/valubench 20000 20 > $OUT/valubench.log 2>&1 || echo "valubench failed" >> $OUT/errors.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g2 -- python3 $R/tools/bench_wow64.py 8192 2 f32only > $OUT/cfg5.log 2>&1 || echo "cfg5 failed" >> $OUT/errors.log
cd $R
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
with open(os.path.join(out, "calibration.csv"), "w") as fo:
    fo.write("kernel,grid_waves,launches,avg_ms,eff_clock_ghz,valu_busy_share,valu_insts_per_wave,cycles_per_valu_inst_per_simd,wait_inst_any_share\n")
    for g in ("g1", "g2"):
        dur = defaultdict(list); grid = {}
        for f in glob.glob(os.path.join(out, g, "**", "*_kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = (r["Kernel_Name"], str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
        cnt = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(out, g, "**", "*_counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            for r in csv.DictReader(open(f)):
                per[(r["Kernel_Name"], r.get("Grid_Size", ""), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (k, gs, d, c), v in per.items():
                cnt[(k, gs)][c].append(v)
        for key, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            if key not in cnt or "GRBM_GUI_ACTIVE" not in cnt[key] or not ("valubench" in key[0] or "bilateral" in key[0]):
                continue
            c = {k: sum(v) / len(v) for k, v in cnt[key].items()}
            t = sum(d) / len(d)
            clk = c["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
            waves = int(key[1]) // 64 if key[1] else 0
            busy = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (clk * 1e9 * t)
            ipw = c.get("SQ_INSTS_VALU", 0) / max(1, waves)
            cpi = clk * 1e9 * t * 1024 / max(1.0, c.get("SQ_INSTS_VALU", 1))
            wia = c.get("SQ_WAIT_INST_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1))
            fo.write(f'"{key[0][:60]}",{waves},{len(d)},{t * 1e3:.4f},{clk:.3f},{busy:.3f},{ipw:.0f},{cpi:.2f},{wia:.3f}\n')
print(open(os.path.join(out, "calibration.csv")).read())
PY
