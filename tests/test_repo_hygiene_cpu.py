"""Repository hygiene: the history stays source-only.

Round 5 committed 42 unbundled gfx950 code objects next to the library by accident (names that slipped
past `*.so`); these checks make that impossible to repeat."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tracked():
    try:
        out = subprocess.run(["git", "ls-files", "-z"], cwd=ROOT, check=True, capture_output=True).stdout
    except (OSError, subprocess.CalledProcessError):
        pytest.skip("not a git checkout (the GPU box receives a snapshot without .git)")
    return [f for f in out.decode().split("\0") if f and os.path.exists(os.path.join(ROOT, f))]


def test_no_large_tracked_file_outside_the_fixtures():
    big = [(f, os.path.getsize(os.path.join(ROOT, f))) for f in tracked()
           if not f.startswith("tests/golden/") and os.path.getsize(os.path.join(ROOT, f)) > (1 << 20)]
    assert not big, f"tracked files over 1 MiB outside tests/golden/: {big}"


def test_no_binary_in_the_package_directory():
    bad = []
    for f in tracked():
        if not f.startswith("wavelets_amd/"):
            continue
        with open(os.path.join(ROOT, f), "rb") as fh:
            head = fh.read(8)
        # ELF (host objects, shared libraries, gfx950 code objects) and clang offload bundles
        if head[:4] == b"\x7fELF" or head.startswith(b"__CLANG") or ".hipv4-" in f or ".host-x86_64" in f:
            bad.append(f)
        elif not f.endswith((".py", ".h", ".hip", ".md", ".c", ".cpp")):
            bad.append(f)
    assert not bad, f"non-source files tracked under wavelets_amd/: {bad}"


def test_ignore_lists_cover_unbundled_code_objects():
    gi = open(os.path.join(ROOT, ".gitignore")).read().split()
    gp = open(os.path.join(ROOT, ".gpurunignore")).read().split()
    for pat in ("*.hipv4-*", "*.host-x86_64-*"):
        assert pat in gi and pat in gp
    assert "*.so.*" in gi
    assert "oracle/_ref/" not in gp and "*.so" not in gp          # built libraries must travel to the GPU box
    assert "wavelets_amd/csrc/_build*" in gp                      # ... their object files need not (13 MB per build directory)
