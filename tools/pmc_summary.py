#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/profile_round.sh.

    python tools/pmc_summary.py gpurun_out/final profiles/r01_e [side] [config] [fetch_dir write_dir]

Reads <dir>/pmc_fetch and <dir>/pmc_write (one counter per pass, as MI355X_MICROARCH.md
prescribes), writes <prefix>_pmc_summary.csv (KiB per dispatch, averaged) and updates
profiles/traffic.json with HBM bytes per launch: 2 * FETCH_SIZE (gfx950 correction) +
WRITE_SIZE, keyed by the names bench.py's live profiler uses."""
import csv
import glob
import json
import os
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
side = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
config = sys.argv[4] if len(sys.argv) > 4 else None      # cfg2 / cfg3 / cfg5: keys "<kernel>@<config>"
subdirs = (sys.argv[5], sys.argv[6]) if len(sys.argv) > 6 else ("pmc_fetch", "pmc_write")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scope_name(kernel):
    """rocprof kernel name -> ProfScope name of the host units / wt_fused.h"""
    m = re.match(r"void wt_fused_kernel<(?:float, )?(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))?>", kernel)
    if m:
        _, ns, d, _, _, acc = map(int, m.groups())
        return f"{('wt_fused', 'wt_fused_acc', 'wt_fused_sum', 'wt_fused_hist')[acc]}<d{d}x{ns}>"
    m = re.match(r"void (wt_row_kernel|wt_lattice_kernel|wt_chain_kernel)<(\d+), (\d+)", kernel)
    if m:       # the single-scale operators are named by their mode (wt_stencil_launch.h wt_row_name / wt_lattice_name)
        mode = int(m.group(3))
        names = {"wt_row_kernel": ("smooth", "smooth_sq", "decomp", "variance", "wow"),
                 "wt_lattice_kernel": ("smooth", "smooth_sq", "decomp", "var", "wow"),
                 "wt_chain_kernel": ("smooth", "smooth_sq", "decomp", "variance", "wow")}[m.group(1)]
        return f"{m.group(1)}<{names[min(mode, 4)]}>"
    return re.sub(r"^void ", "", kernel).split("(")[0].split("<")[0]


rows, traffic = [], {}
for counter, sub in (("FETCH_SIZE", subdirs[0]), ("WRITE_SIZE", subdirs[1])):
    acc, per_scope = {}, {}
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    for k, (n, tot) in sorted(acc.items()):
        rows.append((counter, k, n, tot / n))
        if k.startswith("void wt_") or k.startswith("wt_"):
            # several instantiations can share one profiling name (wt_row_kernel<wow> for d < 4 and
            # d >= 4, the two histogram variants): average over ALL their dispatches
            a = per_scope.setdefault(f"{scope_name(k)}@{config or side}", [0, 0.0])
            a[0] += n
            a[1] += tot
    for key, (n, tot) in per_scope.items():
        traffic[key] = traffic.get(key, 0.0) + (2 if counter == "FETCH_SIZE" else 1) * tot / n * 1024

with open(prefix + "_pmc_summary.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py "
            f"--steps 5 --warmup 1 --no-cpu --brief{' --config ' + config if config else ''}; {side}x{side} f32\n"
            "# KiB per dispatch, averaged. gfx950 correction (MI355X_MICROARCH.md, HBM): hbm_read = 2*FETCH_SIZE*1024\n"
            "counter,kernel,dispatches,avg_value_KiB\n")
    for c, k, n, v in rows:
        f.write(f'{c},"{k}",{n},{v:.1f}\n')
tpath = os.path.join(ROOT, "profiles", "traffic.json")
old = json.load(open(tpath)) if os.path.exists(tpath) else {}
# The table is tied to the kernel sources it was measured on (bench.source_digest: csrc + header,
# comments and whitespace removed) and to the image size of every configuration: bench.py reports
# `traffic` / `hbm_achieved` as null with `traffic_stale: true` when either differs, and
# tests/test_abi_cpu.py fails on a committed table that no longer matches the sources.  Entries
# measured on other sources are dropped when the digest changes.
sys.path.insert(0, ROOT)
from bench import source_digest, CONFIGS  # noqa: E402
digest = source_digest()
meta = old.get("_meta") or {}
if meta.get("source_digest") != digest:
    old = {}
    meta = {"source_digest": digest, "image": {}}
meta.setdefault("image", {})[config or "headline"] = [side, side]
if not config:
    meta["image"][str(side)] = [side, side]
meta["note"] = ("HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, one counter per pass; "
                "tools/profile_round.sh), keys '<profiling name>@<config>'")
old["_meta"] = meta
old.update({k: round(v) for k, v in traffic.items()})
json.dump(old, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps({k: round(v) for k, v in traffic.items()}, indent=1))
