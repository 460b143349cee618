"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports
every symbol include/watroo_hip.h declares, the host-logic entry points work, and the product
package fails loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as entry
from wavelets_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    entry.build()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "watroo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wt(?:64)?_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"libwatroo_hip.so lacks {n}"
    assert sorted(_lib.SIGNATURES) == names       # python binding covers the whole header
    assert _lib.load().wt_abi_version() == 8


def test_schedule_host_logic():
    # B3 (hw=2): 6 scales fuse into two 3-scale passes with cumulative halos 14 and 112 rows
    assert _lib.schedule(_lib.B3SPLINE, 6, True) == [(0, 3, 14), (3, 3, 112)]
    assert _lib.schedule(_lib.B3SPLINE, 4, True) == [(0, 3, 14), (3, 1, 16)]
    # unfused: one pass per scale, halo hw * 2^s
    assert _lib.schedule(_lib.TRIANGLE, 4, False) == [(0, 1, 1), (1, 1, 2), (2, 1, 4), (3, 1, 8)]
    sch = _lib.schedule(_lib.B3SPLINE, 11, True)
    assert sum(n for _, n, _ in sch) == 11 and [s for s, _, _ in sch][:3] == [0, 3, 6]
    # scales 6-7 fuse too (D = 64, two scales): Triangle L = 8 is three fused passes
    # ... and since round 2 the 3-tap family fuses FOUR scales per pass from 8 scales on
    assert _lib.schedule(_lib.TRIANGLE, 8, True) == [(0, 4, 15), (4, 4, 240)]
    assert _lib.schedule(_lib.TRIANGLE, 7, True) == [(0, 3, 7), (3, 3, 56), (6, 1, 64)]
    assert _lib.schedule(_lib.TRIANGLE, 4, True) == [(0, 4, 15)]
    assert _lib.schedule(_lib.TRIANGLE, 5, True) == [(0, 3, 7), (3, 2, 24)]
    assert _lib.schedule(_lib.TRIANGLE, 6, True) == [(0, 3, 7), (3, 3, 56)]
    assert _lib.schedule(_lib.TRIANGLE, 10, True)[:3] == [(0, 4, 15), (4, 4, 240), (8, 1, 256)]
    assert sch[2] == (6, 2, 2 * (256 - 64)) and sch[3][:2] == (8, 1)
    total = sum(h for _, _, h in _lib.schedule(_lib.B3SPLINE, 6, False))
    assert total == 2 * 63 == sum(h for _, _, h in _lib.schedule(_lib.B3SPLINE, 6, True))
    with pytest.raises(_lib.WatrooHipError):
        _lib.schedule(7, 3)


def test_fails_loudly_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    import wavelets_amd as W
    with pytest.raises(_lib.WatrooHipError, match="no CPU fallback"):
        W.AtrousTransform()(np.ones((8, 8), np.float32), 2)
    with pytest.raises(_lib.WatrooHipError):
        W.denoise(np.ones((8, 8), np.float32), [3])


def test_argument_errors_match_reference():
    import wavelets_amd as W
    with pytest.raises(ValueError, match="Unsupported number of dimensions"):
        W.AtrousTransform()(np.ones((2, 2, 2, 2)), 1)          # ref wavelets.py:316-317
    with pytest.raises(ValueError, match="Unknown input type"):
        W.wow([1, 2, 3])                                        # ref utils.py:133
    with pytest.raises(ValueError, match="Unsupported number of dimensions"):
        W.B3spline(4)                                           # ref wavelets.py:189


def test_scaling_function_objects_match_oracle_constants():
    import wavelets_amd as W
    from oracle import atrous_numpy as O
    for cls, fam in ((W.B3spline, "b3spline"), (W.Triangle, "triangle")):
        sf = cls(2)
        assert sf.name == fam and sf.n_dim == 2
        np.testing.assert_array_equal(sf.coefficients_1d, O.TAPS[fam])
        np.testing.assert_array_equal(sf.kernel, O.kernel_2d(fam, np.float64))
        np.testing.assert_array_equal(sf.sigma_e(), O.SIGMA_E_2D[fam])
        np.testing.assert_array_equal(sf.sigma_e(bilateral=1), O.SIGMA_E_2D_BILATERAL[fam])
        k = sf.atrous_kernel(2)
        assert k.shape == ((len(O.TAPS[fam]) - 1) * 4 + 1,) * 2
        np.testing.assert_array_equal(k[::4, ::4], sf.kernel)
        assert k.sum() == pytest.approx(1.0)
        assert cls(3).kernel.shape == (len(O.TAPS[fam]),) * 3
        assert cls(1).sigma_e(bilateral=1) is None              # ref: 1-D bilateral table absent


def test_bench_algorithmic_byte_shares_add_up():
    """bench.py attributes SURVEY 8(d)'s 64 B/pixel (L = 6) to the launches of a step: the shares
    of the accumulate passes must add up to 8*(L+2), those of the plain passes to 4*(L+2)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.algorithmic_bytes_per_pixel
    assert f("wt_fused_acc<d1x3>", 6) + f("wt_fused_sum<d8x3>", 6) == 64
    assert f("wt_fused<d1x3>", 6) + f("wt_fused<d8x3>", 6) == 32
    assert f("wt_plane_sum_kernel", 6) == 32
    # Triangle L = 8: three passes
    assert f("wt_fused_acc<d1x3>", 8) + f("wt_fused_acc<d8x3>", 8) + f("wt_fused_sum<d64x2>", 8) == 80
    assert f("wt_fused_sum<d1x3>", 3) == 8 * 5       # single pass, L = 3


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_traffic_table_is_tied_to_the_kernel_sources(tmp_path):
    """profiles/traffic.json (PMC bytes per launch) describes the kernels it was measured on:
    bench.load_traffic refuses a table whose source digest differs from the sources the library is
    built from (bench.py then reports traffic / hbm_achieved as null with traffic_stale: true), a
    table without a digest, and any table when WATROO_HIP_LIB puts another library in place."""
    import json
    bench = _bench_module()
    d = bench.source_digest()
    assert len(d) == 40 and d == bench.source_digest()
    good = tmp_path / "good.json"
    good.write_text(json.dumps({"_meta": {"source_digest": d, "image": {"headline": [8192, 8192]}}, "k@headline": 5}))
    db, stale = bench.load_traffic(str(good), lib_overridden=False)
    assert stale is None and db["k@headline"] == 5
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"_meta": {"source_digest": "0" * 40}, "k@headline": 5}))
    assert "other kernel sources" in bench.load_traffic(str(bad), lib_overridden=False)[1]
    old = tmp_path / "old.json"
    old.write_text(json.dumps({"k@headline": 5}))
    assert "no source digest" in bench.load_traffic(str(old), lib_overridden=False)[1]
    assert "WATROO_HIP_LIB" in bench.load_traffic(str(good), lib_overridden=True)[1]
    assert "missing" in bench.load_traffic(str(tmp_path / "none.json"), lib_overridden=False)[1]


def test_source_digest_ignores_comments_and_layout_only(tmp_path, monkeypatch):
    bench = _bench_module()
    import shutil
    root = tmp_path / "r"
    shutil.copytree(os.path.join(ROOT, "wavelets_amd", "csrc"), root / "wavelets_amd" / "csrc")
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    monkeypatch.setattr(bench, "ROOT", str(root))
    d0 = bench.source_digest()
    f = root / "wavelets_amd" / "csrc" / "wt_internal.h"
    f.write_text("// a new comment\n/* and a block\n comment */\n" + f.read_text().replace("    ", "\t"))
    assert bench.source_digest() == d0
    f.write_text(f.read_text().replace("int num_cus = 256;", "int num_cus = 255;"))
    assert bench.source_digest() != d0


def test_committed_traffic_table_matches_the_current_sources():
    """The committed table must have been measured on the committed kernels: after a kernel edit,
    re-run tools/profile_round.sh on a GPU box and commit profiles/traffic.json with it."""
    bench = _bench_module()
    db, stale = bench.load_traffic(lib_overridden=False)
    assert stale is None, stale
    sizes = db["_meta"]["image"]
    for cfg, (side, _, _, _) in bench.CONFIGS.items():
        assert sizes.get(cfg) == [side, side], (cfg, sizes.get(cfg))


def _build_abi_demo():
    import subprocess
    import __graft_entry__ as entry
    entry.build()
    exe = os.path.join(ROOT, "examples", "abi_demo")
    libdir = os.path.join(ROOT, "wavelets_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "abi_demo.c"), "-o", exe, "-L" + libdir,
                           "-lwatroo_hip", "-Wl,-rpath," + libdir, "-lm"])
    return exe


def test_header_is_plain_c_and_a_c_client_links():
    """include/watroo_hip.h compiles as C (gcc, -Wall -Werror) and a C client links against the
    shared library; without a GPU it must fail loudly (exit code 3, 'no HIP device')."""
    import subprocess
    exe = _build_abi_demo()
    r = subprocess.run([exe, "64", "80", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 3), r.stdout + r.stderr
    if r.returncode == 3:
        assert "no HIP device" in r.stderr


def test_no_kernel_uses_scratch_memory_or_spills(tmp_path):
    """Every kernel of libwatroo_hip.so keeps its working set in registers: no private (scratch)
    segment, no spilled VGPRs.  The marching kernels hold sliding windows in register arrays indexed
    by fully unrolled loops; an innocent edit (a run-time switch around a window update, a loop the
    compiler declines to unroll) silently moves a window to scratch memory and costs 10x - this
    happened twice in round 3.  Reads the kernel metadata of the embedded gfx950 code object."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("ROCm LLVM tools not installed")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", so], check=True, capture_output=True)
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    import __graft_entry__ as entry
    assert len(objs) == len(entry._units()), f"expected one device code object per translation unit, found {len(objs)}"
    kernels = []
    for obj in objs:
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / obj)],
                               check=True, capture_output=True, text=True).stdout
        kernels += re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)",
                              notes, flags=re.S)
    # (the kernels of wt_kernels_common.h that are not templates are `static`: one copy per unit that includes them,
    #  and a template instantiated in two units appears in both - a few dozen duplicates, no more)
    assert len(kernels) > 200 and len(kernels) - len(set(k[0] for k in kernels)) <= 40
    bad = [(n, int(sc), int(sp)) for n, sc, sp in kernels if int(sc) or int(sp)]
    assert not bad, f"kernels with scratch / spills: {bad[:5]}"


def test_warmup_unit_list_is_the_list_of_units_the_build_compiles():
    """wt_unit_probe.h's WT_UNITS (what the warm-up threads of a context can load: wt_unit_count / wt_unit_name,
    host logic) names exactly the translation units __graft_entry__._units() compiles, each of which defines its
    probe kernel and loader."""
    import __graft_entry__ as G
    from wavelets_amd import _lib as L
    built = sorted(name[:-2] for name, _, _ in G._units())
    assert sorted(L.unit_names()) == built
    for name, _, flags in G._units():
        assert f"-DWT_TU_NAME={name[:-2]}" in flags
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "wavelets_amd", "libwatroo_hip.so")],
                         capture_output=True, text=True).stdout + \
        subprocess.run(["nm", "--defined-only", os.path.join(ROOT, "wavelets_amd", "libwatroo_hip.so")],
                       capture_output=True, text=True).stdout
    for u in built:
        assert f"wt_unit_load_{u}" in out, u
