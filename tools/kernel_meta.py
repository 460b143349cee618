#!/usr/bin/env python3
"""Kernel resource table (VGPRs, spills, scratch, LDS) of a built library's gfx950 code objects.

    python tools/kernel_meta.py [lib.so] [substring-of-mangled-name ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "wavelets_amd", "libwatroo_hip.so")
pats = sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    for obj in sorted(f for f in os.listdir(d) if "amdgcn" in f):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, obj)], check=True, capture_output=True,
                               text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if pats and not any(p in name for p in pats):
                continue
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))   # noqa: E731
            print(f"{name}: vgpr {g('vgpr_count')} sgpr {g('sgpr_count')} spill {g('vgpr_spill_count')} "
                  f"scratch {g('private_segment_fixed_size')} lds {g('group_segment_fixed_size')}")
