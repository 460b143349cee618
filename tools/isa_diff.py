#!/usr/bin/env python3
"""Compare the gfx950 code of the library's kernels between two builds, kernel by kernel.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o before.s wavelets_amd/csrc/wt_transform.hip
    ... edit ...
    hipcc ... -o after.s ...
    python tools/isa_diff.py before.s after.s [name-filter]

A refactor of a tuned kernel (e.g. templating wt_fused_kernel on the element type) must leave the
float32 instantiations' instruction streams unchanged; this prints which kernels differ.  Kernel
names are demangled and normalised (`float, ` template arguments that the refactor adds are
dropped), basic-block labels are renumbered per function."""
import re
import subprocess
import sys


def functions(path):
    out, name, body = {}, None, []
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                out[name] = body
                name = None
                continue
            s = line.split(";")[0].rstrip()
            if not s.strip() or s.lstrip().startswith("."):
                if not re.match(r"^\.LBB\d+_\d+:", s):
                    continue
            body.append(s)
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def norm_name(n):
    n = re.sub(r"^void ", "", n).split("(")[0]
    n = n.replace("<float, ", "<").replace("FusedArgsT<float>", "FusedArgs")
    return n


def norm_body(body):
    labels = {}
    out = []
    for s in body:
        for lab in re.findall(r"\.LBB\d+_\d+", s):
            labels.setdefault(lab, f".L{len(labels)}")
    for s in body:
        out.append(re.sub(r"\.LBB\d+_\d+", lambda m: labels[m.group(0)], s))
    return out


a, b = functions(sys.argv[1]), functions(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
da, db = demangle(list(a)), demangle(list(b))
A = {norm_name(da[k]): norm_body(v) for k, v in a.items()}
B = {norm_name(db[k]): norm_body(v) for k, v in b.items()}
same = diff = 0
for k in sorted(A):
    if flt and flt not in k:
        continue
    if k not in B:
        print(f"ONLY BEFORE  {k}")
        continue
    if A[k] == B[k]:
        same += 1
    else:
        diff += 1
        n = next((i for i, (x, y) in enumerate(zip(A[k], B[k])) if x != y), min(len(A[k]), len(B[k])))
        print(f"DIFFERS      {k}: {len(A[k])} vs {len(B[k])} lines, first difference at line {n}")
for k in sorted(B):
    if (not flt or flt in k) and k not in A:
        print(f"ONLY AFTER   {k}")
print(f"{same} kernels identical, {diff} differ")
