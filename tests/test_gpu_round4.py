"""Round-4 GPU tests (run with -m gpu on an MI355X): the torch-free multi-rank plumbing of the bench
on the driver's own command form, the per-pass / overlap / placement instrumentation of the
multi-GPU line, Coefficients of generic-tap scaling functions that lost their plan."""
import copy
import json
import os
import pickle
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as entry
    entry.build()
    from wavelets_amd import _lib
    return _lib


def rnd(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def _clean_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "WT_RDZV"):
        env.pop(k, None)
    return env


def _one_json_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_multi_gpu_fields(out, nranks, size):
    assert out["n_gpus"] == nranks and out["rccl_ranks"] == nranks
    # round 5: who ran where, the RCCL that was loaded, and what the line should read
    assert [r["rank"] for r in out["ranks"]] == list(range(nranks))
    assert all(r["pci"] not in ("", "?") and r["name"] and r["cus"] >= 64 for r in out["ranks"]), out["ranks"]
    from wavelets_amd import _lib as _L               # (a one-card box: every rank shares it, and the line says so)
    assert out["distinct_gpus"] == min(nranks, max(1, _L.device_count()))
    assert out["rccl_version"] >= 20000, out["rccl_version"]
    sm = out["scaling_model"]
    assert "error" not in sm, sm
    assert len(sm["strip_compute_ms_per_rank"]) == nranks and all(v > 0 for v in sm["strip_compute_ms_per_rank"])
    assert sm["exchange_bytes_per_neighbour_per_step"] == sum(h for _, _, h in out["config"]["schedule"]) * size * 4
    assert sm["order"] in ("overlapped", "serial") and sm["predicted_ms_per_step"][sm["order"]] > 0
    assert abs(sm["measured_over_predicted_time"] - sm["measured_ms_per_step"] / sm["predicted_ms_per_step"][sm["order"]]) < 2e-3
    assert sm["predicted_value_mpix_s"] > 0
    assert out["config"]["image"] == [size, size] and out["config"]["parallelism"] == f"strips{nranks}"
    assert out["halo_selfcheck"]["ok"], out["halo_selfcheck"]
    # per pass: the exchange on the communication stream, the interior launch beside it, the edge rows
    sp = out["strip_passes"]
    assert len(sp["rank0"]) == len(out["config"]["schedule"]) == len(sp["max_over_ranks"])
    overlapped = out["overlap"]["chosen"] == "on"
    for e, (s0, ns, halo) in zip(sp["rank0"], out["config"]["schedule"]):
        assert e["scales"] == [s0, s0 + ns] and e["halo_rows"] == halo and e["kernel"].startswith("wt_fused")
        if overlapped:
            assert e["exchange_ms"] > 0 and e["interior_ms"] > 0 and e["edge_ms"] > 0, e
        else:
            assert e["whole_ms"] > 0, e
    if not overlapped:
        assert sp["serial_exchanges_rank0"]["exchanges_per_step"] == len(out["config"]["schedule"])
    for k, v in out["kernels"].items():
        if k.startswith("wt_fused"):
            assert v["calls_per_step"] == 1, (k, v)
            if overlapped:
                assert set(v["parts_ms"]) == {"interior", "edge"}, (k, v)
    # the same steps with the exchanges in serial order, and the placement A/B of the strip planes
    ov = out["overlap"]
    assert ov["default"] == "on" and ov["chosen"] in ("on", "off") and ov["ms_per_step_on"] > 0 and ov["ms_per_step_off"] > 0
    if ov["chosen"] == "off":             # the reported value is the faster order's full timed region
        assert ov["ms_per_step_off"] < ov["ms_per_step_on"] and ov["steps_off"] == out["steps"]
    else:                                 # round 5: the compute units left to RCCL beside the interior launch, swept
        assert "reserve_sweep_error" not in ov, ov
        assert set(ov["reserve_sweep_ms_per_step"]) == {"0", "8", "16", "32", "64"} and all(v > 0 for v in ov["reserve_sweep_ms_per_step"].values())
        assert ov["reserve_chosen"] in (0, 8, 16, 32, 64)
        best = min(ov["reserve_sweep_ms_per_step"].values())
        if ov["reserve_chosen"] != 16:
            assert ov["reserve_sweep_ms_per_step"][str(ov["reserve_chosen"])] < 0.98 * ov["reserve_sweep_ms_per_step"]["16"]
        else:
            assert best >= 0.98 * ov["reserve_sweep_ms_per_step"]["16"] * 0.9      # (nothing much faster was passed over)
    pl = out["strip_planes"]
    assert pl["chosen"] in ("hipMalloc", "scattered") and pl["hipMalloc_ms_per_step"] > 0
    if size * (size // nranks) * 4 >= (8 << 20):          # planes of 8 MiB and more can be scattered
        assert pl.get("scattered_halo_selfcheck_ok") is True, pl
        assert pl["scattered_ms_per_step"] > 0
    chosen_ms = pl["scattered_ms_per_step"] if pl["chosen"] == "scattered" else pl["hipMalloc_ms_per_step"]
    assert abs(out["ms_per_step"] - chosen_ms) < 1e-3
    if pl["chosen"] == "hipMalloc":
        assert abs(out["ms_per_step"] - (ov["ms_per_step_off"] if ov["chosen"] == "off" else ov["ms_per_step_on"])) < 1e-3


def test_bench_ranks_under_torch_distributed_run_import_no_torch():
    """The driver's command: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
    torch lives in that launcher process only; the ranks find each other over the socket named after
    MASTER_PORT, exchange the RCCL unique id there and never import torch - the multi-GPU line runs
    on the same system ROCm stack as the N = 1 line."""
    from conftest import free_port, run_ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--size", "4096", "--steps", "3",
           "--warmup", "1", "--no-cpu"]
    out = _one_json_line(run_ranks(cmd, _clean_env(), "bench2_torchrun"))
    assert out["launcher"] == {"plumbing": "stdlib", "torch_imported_in_ranks": False,
                               "started_by": "torch.distributed.run"}
    _check_multi_gpu_fields(out, 2, 4096)


def test_bench_two_ranks_kept_on_the_overlapped_order_sweep_the_reserved_compute_units():
    """`bench.py --gpus 2 --keep-overlap`: ranks that share one card always measure the serial order faster, so the
    line of the other multi-rank tests never reaches what follows the overlapped order on a real node - the sweep of
    `overlap_reserve` (0 / 8 / 16 / 32 / 64 compute units left to RCCL's kernels), its choice and the re-timed
    region.  Here the order is kept and the sweep must run on every rank in step (it is made of collectively
    timed regions), report all five values and leave a consistent line."""
    from conftest import run_ranks
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--size", "4096", "--steps", "3",
           "--warmup", "1", "--no-cpu", "--keep-overlap"]
    out = _one_json_line(run_ranks(cmd, _clean_env(), "bench2_keep_overlap"))
    assert out["overlap"]["chosen"] == "on"
    _check_multi_gpu_fields(out, 2, 4096)


def test_bench_self_launch_three_ranks_reports_passes_overlap_and_placement():
    """`python bench.py --gpus 3` starts its ranks through wavelets_amd.launch (a middle rank with two
    neighbours): per-pass exchange / interior / edge times, overlap on / off, hipMalloc vs scattered."""
    from conftest import run_ranks
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--shared-gpu", "--size", "6144",
           "--steps", "3", "--warmup", "1", "--no-cpu"]
    out = _one_json_line(run_ranks(cmd, _clean_env(), "bench3"))
    assert out["launcher"]["plumbing"] == "stdlib" and out["launcher"]["torch_imported_in_ranks"] is False
    assert out["launcher"]["started_by"].startswith("bench.py")
    _check_multi_gpu_fields(out, 3, 6144)


def test_bench_launcher_torch_is_still_available():
    """--launcher torch: the round-3 plumbing (torch.distributed / gloo in every rank), kept as an option."""
    from conftest import run_ranks
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--size", "2048",
           "--steps", "3", "--warmup", "1", "--no-cpu", "--launcher", "torch", "--no-scatter-ab"]
    out = _one_json_line(run_ranks(cmd, _clean_env(), "bench2_torch"))
    assert out["launcher"]["plumbing"] == "torch" and out["launcher"]["torch_imported_in_ranks"] is True
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["halo_selfcheck"]["ok"]


def test_bench_time_limit_kills_the_ranks_and_reports_json():
    """A launch that cannot finish inside --time-limit is killed (every rank's process group) and the
    parent prints one JSON error line: a hung ncclCommInitRank cannot eat the lease."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--size", "2048",
                        "--steps", "20000000", "--warmup", "1", "--no-cpu", "--no-build", "--time-limit", "40"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert "time limit" in json.loads(lines[0])["error"]


class Even4(__import__("wavelets_amd").wavelets.AbstractScalingFunction):        # (module level: picklable)
    coefficients_1d = np.array([0.1, 0.4, 0.3, 0.2])
    sigma_e_1d = np.array([0.7, 0.3, 0.2, 0.12, 0.08, 0.06])
    sigma_e_2d = np.array([0.9, 0.2, 0.09, 0.04, 0.02, 0.01])
    sigma_e_3d = np.array([0.95, 0.12, 0.04, 0.014, 0.005])

    def __init__(self, *args, **kwargs):
        super().__init__("even4", *args, **kwargs)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_generic_tap_coefficients_survive_copy_and_pickle(L, dtype):
    """Coefficients of a scaling function with an even number of taps run on the
    generic operator; a deep copy, a shallow copy, an unpickled object and Coefficients(ndarray, sf)
    have no plan and must get a storage plan back on their first device operation - denoise,
    significance, get_noise and np.sum(axis=0) then equal the original's results."""
    import wavelets_amd as WA
    img = (rnd((96, 130), 3) * 2 + 7).astype(dtype)
    ref = WA.AtrousTransform(Even4)(img, 3)
    planes = np.array(ref.data, copy=True)
    noise = ref.get_noise()
    sig = np.array(ref.significance(3, 1), copy=True)
    ref.denoise([5, 3])
    want = np.array(ref.data, copy=True)
    want_sum = np.sum(ref, axis=0)
    fresh = WA.AtrousTransform(Even4)(img, 3)
    for make in (copy.deepcopy, copy.copy, lambda c: pickle.loads(pickle.dumps(c)),
                 lambda c: WA.Coefficients(np.array(c.data, copy=True), Even4(2))):
        c = make(fresh)
        assert c._plan is None
        np.testing.assert_array_equal(np.asarray(c.data), planes)
        assert c.get_noise() == noise
        np.testing.assert_array_equal(c.significance(3, 1), sig)
        c.denoise([5, 3])
        np.testing.assert_array_equal(np.asarray(c.data), want)
        np.testing.assert_array_equal(np.sum(c, axis=0), want_sum)
        assert c.data.dtype == dtype


# --------------------------------------------------------------------------- float64 (the reference's default dtype)
def _plan64(L, img, taps, level):
    p = L.Plan64(L.default_context(), img.shape[0], img.shape[1], taps, level)
    p.upload(L.PLANE_INPUT, img)
    return p


B3_TAPS = (1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16)
TRI_TAPS = (1 / 4, 1 / 2, 1 / 4)


def test_float64_erf_of_the_threshold_kernels_vs_scipy(L):
    """wt_erf64 (branch-free: one degree-26 polynomial + expm1) against scipy.special.erf over the whole
    range, tiny arguments and the clamp included: absolute error <= 1e-15 (the float64 parity bound of
    the golden tests is 1e-12)."""
    from scipy.special import erf
    y = np.concatenate([np.linspace(0, 7, 40000), 10.0 ** np.linspace(-300, 0, 3000), [0.0, 5.999, 6.0, 6.001, 30.0, 1e300]])
    y = np.concatenate([y, -y])
    img = np.resize(y, (64, 2048)).astype(np.float64)
    p = _plan64(L, img, B3_TAPS, 1)
    p.significance(L.PLANE_INPUT, L.PLANE_OUT, 1.0, True)            # erf(|v| / 1)
    got = p.download(L.PLANE_OUT)
    assert float(np.abs(got - erf(np.abs(img))).max()) <= 1e-15
    assert got[img == 0].max() == 0.0 and got.max() <= 1.0
    tiny = np.abs(img) < 1e-3
    rel = np.abs(got[tiny & (img != 0)] / erf(np.abs(img[tiny & (img != 0)])) - 1)
    assert float(rel.max()) <= 1e-13                                 # expm1 keeps small arguments relative
    p.significance(L.PLANE_INPUT, L.PLANE_OUT, 0.5, False)           # hard threshold
    np.testing.assert_array_equal(p.download(L.PLANE_OUT), (np.abs(img) > 0.5).astype(np.float64))
    p.close()


@pytest.mark.parametrize("shape,taps,level", [((300, 517), B3_TAPS, 5), ((257, 1024), TRI_TAPS, 8), ((64, 33), B3_TAPS, 2)])
def test_float64_denoise_sum_kernel_equals_the_two_step_form_bitwise(L, shape, taps, level):
    """wt64_denoise_sum (thresholds + plane sum in one kernel) against wt64_significance per plane and
    wt64_plane_sum: planes (written back or not) and the sum, bit for bit; soft and hard thresholds, a
    weight-only plane (tau = 0), a noise map."""
    img = (rnd(shape, 11) * 3 + 100).astype(np.float64)
    nz = (1.0 + 0.3 * np.abs(rnd(shape, 12))).astype(np.float64)
    for soft in (True, False):
        for noise_plane in (L.PLANE_NONE, L.PLANE_SCRATCH(7)):
            a, b = _plan64(L, img, taps, level), _plan64(L, img, taps, level)
            for p in (a, b):
                p.decompose(L.PLANE_INPUT, level)
                if noise_plane != L.PLANE_NONE:
                    p.upload(noise_plane, nz)
            taus, wgts = [1.7, 0.0, 0.4][:min(3, level)], [1.0, 0.5, 2.0][:min(3, level)]
            for s, (t, w) in enumerate(zip(taus, wgts)):
                a.denoise(s, t, w, soft, noise_plane)                  # (tau = 0: weight only)
            a.plane_sum(0, level + 1, L.PLANE_OUT)
            keep = [b.download(s).copy() for s in range(level + 1)]
            b.denoise_sum(level + 1, taus, wgts, soft, noise_plane, write_back=False)
            np.testing.assert_array_equal(b.download(L.PLANE_OUT).view(np.uint64), a.download(L.PLANE_OUT).view(np.uint64))
            for s in range(level + 1):
                np.testing.assert_array_equal(b.download(s), keep[s])  # untouched without write_back
            b.denoise_sum(level + 1, taus, wgts, soft, noise_plane, write_back=True)
            for s in range(level + 1):
                np.testing.assert_array_equal(b.download(s).view(np.uint64), a.download(s).view(np.uint64))
            np.testing.assert_array_equal(b.download(L.PLANE_OUT).view(np.uint64), a.download(L.PLANE_OUT).view(np.uint64))
            a.close()
            b.close()


@pytest.mark.parametrize("case", ["gauss_even", "gauss_odd", "quantised", "constant", "two_values", "tiny", "wide_range",
                                  "upper_in_next_bin"])
def test_float64_median_select_is_exact(L, case):
    """np.median(np.abs(plane)) in float64 must be exact for any data: continuous planes take the
    gathered-list finish (two radix levels, one collect pass, one workgroup), planes with ties whose bin
    does not fit the list fall back to the radix passes; with and without the first level riding on the
    transform's first pass (flag bit4), both list and radix paths forced."""
    rng = np.random.default_rng(7)
    if case == "gauss_even":
        img = rng.standard_normal((1024, 2048))
    elif case == "gauss_odd":
        img = rng.standard_normal((333, 1001)) * 1e-3 + 5.0
    elif case == "quantised":
        img = np.round(rng.standard_normal((2048, 2048)) * 3)          # ~10 distinct magnitudes: 4M ties
    elif case == "constant":
        img = np.full((1500, 1500), -2.5)
    elif case == "two_values":
        img = np.where(rng.random((1200, 1100)) < 0.5, 1.0, 1.0 + 2.0 ** -40)
    elif case == "tiny":
        img = rng.standard_normal((3, 2))
    elif case == "wide_range":
        img = rng.standard_normal((512, 512)) * 10.0 ** rng.integers(-200, 200, (512, 512))
    else:   # the lower median is the largest key of its level-2 bin: the upper one is not in the list
        img = np.abs(rng.standard_normal((600, 1000)))
        s = np.sort(img.ravel())
        lo = s[img.size // 2 - 1]
        hi = np.float64(np.frombuffer((np.float64(lo).view(np.uint64) | np.uint64((1 << 41) - 1)).tobytes(), np.float64)[0])
        img = np.where(img > lo, np.maximum(img, hi * (1 + 2.0 ** -10)), img)     # nothing else left in lo's bin above lo
    img = np.ascontiguousarray(img, dtype=np.float64)
    want = np.median(np.abs(img))
    p = _plan64(L, img, B3_TAPS, 0)
    try:
        for use_list in (1, 0):
            L.set_option("select64_list", use_list)
            assert p.abs_median(L.PLANE_INPUT) == want, use_list
    finally:
        L.set_option("select64_list", 1)
    p.close()
    # through the public API: the transform's first pass histograms |w_0| (fused images), then get_noise
    import wavelets_amd as WA
    if min(img.shape) >= 2:
        c = WA.AtrousTransform(WA.B3spline)(img, 3)
        w0 = np.array(c.data[0], copy=True)
        assert c.get_noise() == np.median(np.abs(w0)) / 0.6745 / c.sigma_e[0]
        c2 = WA.AtrousTransform(WA.Triangle)(img, 4)
        c2.data[0][...] = c2.data[0] * 0.5                              # an edited mirror: the riding histogram is stale
        assert c2.get_noise() == np.median(np.abs(c2.data[0])) / 0.6745 / c2.sigma_e[0]


def test_float64_denoise_interleaved_equals_the_plain_sequence(L):
    """utils.denoise of a float64 image places the threshold step between the fused passes
    (wt64_decompose_pass + histogram, wt64_denoise_sum, wt64_decompose_pass_sum): the same operations in
    the same order as transform -> Coefficients.denoise -> np.sum(axis=0), so identical bits; and within
    1e-12 of the numpy oracle."""
    import wavelets_amd as WA
    from oracle import atrous_numpy as O
    img = (rnd((1024, 1536), 5) * 4 + 1e3).astype(np.float64)
    for cls, name, sig in ((WA.Triangle, "triangle", [5, 3, 2, 0, 0, 0, 0, 0]), (WA.B3spline, "b3spline", [5, 3]),
                           (WA.B3spline, "b3spline", [4, 0, 2, 0, 0, 0])):
        for soft in (True, False):
            got = WA.denoise(img, list(sig), cls, soft_threshold=soft)
            c = WA.AtrousTransform(cls)(img, len(sig))
            c.denoise(list(sig), soft_threshold=soft)
            want = np.sum(c, axis=0)
            assert got.dtype == np.float64
            np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))
            ref = O.denoise(img.copy(), list(sig), name, soft_threshold=soft)
            if soft:          # (a hard threshold can flip on a last-bit difference of a coefficient)
                assert float(np.abs(got - ref).max()) <= 1e-12 * float(np.abs(img).max())
    got = WA.denoise(np.abs(img), [5, 3, 1], WA.B3spline, anscombe=True)
    ref = O.denoise(np.abs(img).copy(), [5, 3, 1], "b3spline", anscombe=True)
    assert float(np.abs(got - ref).max()) <= 1e-11 * float(np.abs(img).max())


def test_standalone_select_places_its_window_from_a_sample_of_the_plane(L):
    """Round 4 (late): a plane no fused pass has histogrammed - bilateral / recursive transforms, an edited
    plane, wt_abs_median on any plane - gets the same 21-bit window, placed from a 4096-sample of the
    plane itself: TWO passes over the plane instead of three.  Exact median on ordinary, quantised,
    constant and heavy-tailed planes, even / odd counts, option on and off; a plane whose sample
    mispredicts (non-zero only at the sample points / zero only there) falls back to the three passes;
    small planes keep them."""
    ctx = L.default_context()
    rng = np.random.default_rng(23)
    H, W = 1031, 1100                                        # odd count: the median is one element
    ys = ((2 * np.arange(64) + 1) * H) >> 7
    xs = ((2 * np.arange(64) + 1) * W) >> 7
    g = rng.standard_normal((H, W)).astype(np.float32)
    spikes = np.zeros((H, W), np.float32)
    spikes[ys[:, None], xs[None, :]] = 7.0
    holes = g.copy()
    holes[ys[:, None], xs[None, :]] = 0.0
    cases = [("gauss", g, 2), ("offset", g * 1e-3 + 40, 2), ("quantised", np.round(g * 3) / 3, 2),
             ("constant", np.full((H, W), 2.5, np.float32), 2), ("cauchy", rng.standard_cauchy((H, W)).astype(np.float32), 2),
             ("spikes", spikes, 5), ("holes", holes, 5)]    # 5: the windowed pass and its refinement in vain, then three
    p = L.Plan(ctx, H, W, L.B3SPLINE, 1)
    try:
        for name, a, n_hist in cases:
            p.upload(0, a)
            want = np.median(np.abs(a))
            for window in (1, 0):
                L.set_option("hist_window", window)
                try:
                    ctx.profile(True)
                    ctx.profile_reset()
                    got = p.abs_median(0)
                    ent = ctx.profile_entries()
                    ctx.profile(False)
                finally:
                    L.set_option("hist_window", 1)
                assert got == want, (name, window, got, want)
                nh = ent.get("wt_hist_kernel", (0, 0))[0]
                assert nh == (n_hist if window else 3), (name, window, ent)
    finally:
        p.close()
    # even count (two middle elements), and a plane below the size threshold
    for shape, nh_want in (((1024, 1024), 2), ((512, 600), 3)):
        a = rng.standard_normal(shape).astype(np.float32)
        p = L.Plan(ctx, shape[0], shape[1], L.TRIANGLE, 1)
        try:
            p.upload(0, a)
            ctx.profile(True)
            ctx.profile_reset()
            got = p.abs_median(0)
            ent = ctx.profile_entries()
            ctx.profile(False)
            assert got == np.median(np.abs(a)) and ent.get("wt_hist_kernel", (0, 0))[0] == nh_want, (shape, ent)
        finally:
            p.close()


def test_float64_cfg3_at_full_size_properties(L):
    """BASELINE configs[2] on the float64 engine at its full size (8192 x 8192, Triangle, 8 scales,
    denoise([5, 3, 2]) soft; bench.py's `float64_cfg3` entry) through size-independent properties: the
    interleaved flow gives the bits of transform -> Coefficients.denoise -> np.sum(axis=0), the noise
    estimate is the exact median of the plane 0 the engine produced, and soft thresholds shrink."""
    import wavelets_amd as WA
    rng = np.random.default_rng(8)
    img = rng.standard_normal((8192, 8192))
    sig = [5, 3, 2, 0, 0, 0, 0, 0]
    got = WA.denoise(img, list(sig), WA.Triangle)
    assert got.dtype == np.float64 and got.shape == img.shape
    c = WA.AtrousTransform(WA.Triangle)(img, len(sig))
    w0 = np.abs(c.data[0])
    noise = c.get_noise()
    assert noise == np.median(w0) / 0.6745 / c.sigma_e[0]
    del w0
    c.denoise(list(sig))
    want = np.sum(c, axis=0)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # soft thresholds shrink: the thresholded image has less energy than the input, and is not the input
    assert float(np.abs(got).mean()) < float(np.abs(img).mean()) and not np.array_equal(got, img)


# --------------------------------------------------------------------------- the last refusals (g22)
def _sf(name, taps, e1, e2, e3):
    import wavelets_amd as WA

    class SF(WA.wavelets.AbstractScalingFunction):
        coefficients_1d = np.asarray(taps)
        sigma_e_1d, sigma_e_2d, sigma_e_3d = np.asarray(e1), np.asarray(e2), np.asarray(e3)

        def __init__(self, *args, **kwargs):
            super().__init__(name, *args, **kwargs)
    return SF


def test_remaining_np_pad_modes_of_atrous_convolution_vs_reference_golden(L):
    """g22 (hard pin: the reference's own numpy loop): atrous_convolution under np.pad's 'linear_ramp',
    'maximum', 'mean', 'median' and 'minimum' - the border is synthesised by np.pad as in the reference
    (wavelets.py:77), the taps run on the generic operator - plain and range-weighted, signals and
    images, float32 and float64; unknown modes raise np.pad's ValueError."""
    from conftest import load_golden
    from wavelets_amd import atrous_convolution
    g = load_golden("g22_refusals")
    a, sig, var = g["img"], g["sig"], g["var"]
    tol = 2e-6 * float(np.abs(a).max())
    for mode in ("linear_ramp", "maximum", "mean", "median", "minimum"):
        for name in ("k3x3", "k4x2"):
            for s in (0, 2):
                got = atrous_convolution(a, g[name], s=s, mode=mode)
                assert got.dtype == np.float32 and got.shape == a.shape
                assert float(np.abs(got - g[f"ac_{name}_{mode}_s{s}"]).max()) <= tol, (mode, name, s)
        assert float(np.abs(atrous_convolution(a, g["k3x3"], var, s=1, mode=mode) - g[f"acb_k3x3_{mode}_s1"]).max()) <= 10 * tol
        assert float(np.abs(atrous_convolution(sig, g["k1d3"], s=1, mode=mode) - g[f"ac1_{mode}_s1"]).max()) <= tol
    a64 = a.astype(np.float64) * 1e3 + 7e5
    got = atrous_convolution(a64, g["k4x2"], s=1, mode="mean")
    assert got.dtype == np.float64 and float(np.abs(got - g["ac_f64_k4x2_mean_s1"]).max()) <= 1e-12 * 7e5
    out = np.empty_like(a)
    assert atrous_convolution(a, g["k3x3"], s=1, mode="median", output=out) is out
    np.testing.assert_allclose(out, g["ac_k3x3_median_s1"] if "ac_k3x3_median_s1" in g.files else
                               atrous_convolution(a, g["k3x3"], s=1, mode="median"), rtol=0, atol=tol)


def test_bilateral_and_recursive_transforms_with_even_or_long_taps_vs_reference_golden(L):
    """g22: AtrousTransform(cls, bilateral=...) and recursive=True for scaling functions with 4 (even)
    and 17 taps - refused until round 3 - on signals, images and cubes, float32 and float64, against the
    unmodified reference's output (wavelets.py:330-406, 433-440)."""
    from conftest import load_golden
    import wavelets_amd as WA
    g = load_golden("g22_refusals")
    a, sig, cube = g["img"], g["sig"], g["cube"]
    amax = float(np.abs(a).max())
    e = {"even4": ([0.7, 0.3, 0.2, 0.12, 0.08, 0.06], [0.9, 0.2, 0.09, 0.04, 0.02, 0.01], [0.95, 0.12, 0.04, 0.014, 0.005]),
         "long17": ([0.5, 0.3, 0.2, 0.12, 0.08, 0.06], [0.6, 0.2, 0.09, 0.04, 0.02, 0.01], [0.7, 0.12, 0.04, 0.014, 0.005])}

    def check(c, ref, tol, what):
        assert c.data.shape == ref.shape and c.data.dtype == ref.dtype, what
        d = float(np.abs(c.data - ref).max())
        assert d <= tol, f"{what}: {d:.3e} > {tol:.3e}"
    plain, bil = 3e-6 * amax, 1e-4 * amax           # (bilateral errors chain through the scales: 2e-5 per scale x 5)
    for name in ("even4", "long17"):
        cls = _sf(name, g[f"{name}_taps"], *e[name])
        check(WA.AtrousTransform(cls, bilateral=1)(a, 3), g[f"{name}_bil2_L3"], bil, f"{name} bilateral 2-D")
        check(WA.AtrousTransform(cls, bilateral=[1.5, 0.7], bilateral_scaling=True)(a, 2), g[f"{name}_bil2_scaled_L2"], bil,
              f"{name} bilateral scaled")
        check(WA.AtrousTransform(cls, bilateral=2)(sig, 2), g[f"{name}_bil1_L2"], bil, f"{name} bilateral 1-D")
        check(WA.AtrousTransform(cls)(a, 3, recursive=True), g[f"{name}_rec2_L3"], plain, f"{name} recursive 2-D")
        check(WA.AtrousTransform(cls)(sig, 3, recursive=True), g[f"{name}_rec1_L3"], plain, f"{name} recursive 1-D")
        check(WA.AtrousTransform(cls, bilateral=1)(a, 2, recursive=True), g[f"{name}_recbil2_L2"], bil, f"{name} recursive bilateral 2-D")
        check(WA.AtrousTransform(cls, bilateral=1)(sig, 2, recursive=True), g[f"{name}_recbil1_L2"], bil, f"{name} recursive bilateral 1-D")
    cls = _sf("even4", g["even4_taps"], *e["even4"])
    check(WA.AtrousTransform(cls, bilateral=1)(cube, 2), g["even4_bil3_L2"], bil, "bilateral 3-D")
    check(WA.AtrousTransform(cls)(cube, 2, recursive=True), g["even4_rec3_L2"], plain, "recursive 3-D")
    check(WA.AtrousTransform(cls, bilateral=1)(cube, 2, recursive=True), g["even4_recbil3_L2"], bil, "recursive bilateral 3-D")
    a64 = a.astype(np.float64) + 1e4
    check(WA.AtrousTransform(cls)(a64, 3, recursive=True), g["even4_rec2_f64_L3"], 1e-11 * 1e4, "recursive float64")
    check(WA.AtrousTransform(cls, bilateral=1)(a64, 2), g["even4_bil2_f64_L2"], 1e-10 * 1e4, "bilateral float64")
    # the Coefficients of such a transform are ordinary ones
    c = WA.AtrousTransform(cls)(a, 3, recursive=True)
    assert np.isfinite(c.get_noise())
    c.denoise([5, 3])
    assert np.sum(c, axis=0).shape == a.shape


# --------------------------------------------------------------------------- circular products through the FFT
@pytest.mark.parametrize("shape,dtype", [((64, 128), np.float32), ((512, 256), np.float32), ((2, 8), np.float32),
                                         ((2048, 2048), np.float32), ((256, 512), np.float64), ((8192, 64), np.float32)])
def test_fft_circular_products_vs_numpy(L, shape, dtype):
    """wt_fft_spectrum / wt_fft_apply (row FFTs in LDS, transposes, the spectrum product fused into the
    first inverse pass) against numpy's irfft2(rfft2(x) * rfft2(k)) and the conjugate product, float32
    and float64, square and very oblong power-of-two shapes."""
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(dtype)
    k = np.zeros(shape, dtype)
    kh, kw = min(shape[0], 9), min(shape[1], 7)
    k[:kh, :kw] = rng.random((kh, kw))
    k /= k.sum()
    k = np.roll(k, (-(kh // 2), -(kw // 2)), axis=(0, 1))
    f = np.fft.rfft2(k.astype(np.float64))
    conv = np.fft.irfft2(np.fft.rfft2(x.astype(np.float64)) * f, s=shape)
    corr = np.fft.irfft2(np.fft.rfft2(x.astype(np.float64)) * f.conj(), s=shape)
    if dtype == np.float32:
        p = L.Plan(L.default_context(), shape[0], shape[1], L.B3SPLINE, 0)
    else:
        p = L.Plan64(L.default_context(), shape[0], shape[1], B3_TAPS, 0)
    S = L.PLANE_SCRATCH(6)
    p.upload(S, k)
    p.upload(L.PLANE_INPUT, x)
    p.fft_spectrum(S)
    tol = (4e-6 if dtype == np.float32 else 1e-13) * float(np.abs(x).max())
    p.fft_apply(L.PLANE_INPUT, L.PLANE_OUT, False)
    assert float(np.abs(p.download(L.PLANE_OUT) - conv).max()) <= tol
    p.fft_apply(L.PLANE_INPUT, L.PLANE_OUT, True)
    assert float(np.abs(p.download(L.PLANE_OUT) - corr).max()) <= tol
    p.close()
    assert L.fft_supported(*shape) and not L.fft_supported(56, 40) and not L.fft_supported(16384, 64)


def test_richardson_lucy_fft_large_psf_vs_reference_golden(L):
    """g23: richardson_lucy(fft=True) with PSFs of 575 and 768 taps on a 64 x 128 image - the FFT path of
    the engine - against the unmodified reference (numpy rfft2 in its loop), soft / hard thresholds, an
    even-sized PSF, float64; and against the direct periodic form of the same products."""
    from conftest import load_golden
    import wavelets_amd as WA
    from wavelets_amd import utils as WU
    g = load_golden("g23_rl_fft_large")
    d = g["data"]
    cases = (("rl_fft_soft", "psf", dict(iterations=4)),
             ("rl_fft_hard", "psf", dict(iterations=3, threshold_type="hard", persistent_mrs=False)),
             ("rl_fft_even", "psf_even", dict(iterations=3, denoise_coefficients=(4, 2))))
    assert g["psf"].size >= WU._FFT_MIN_TAPS and g["psf_even"].size >= WU._FFT_MIN_TAPS
    direct = {}
    for name, psf, kw in cases:
        got = WA.richardson_lucy(d.copy(), g[psf], fft=True, **kw)
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, g[name], atol=2e-4 * np.abs(g[name]).max(), rtol=2e-4)
        keep, WU._FFT_MIN_TAPS = WU._FFT_MIN_TAPS, 1 << 30          # the direct periodic correlations
        try:
            direct[name] = WA.richardson_lucy(d.copy(), g[psf], fft=True, **kw)
        finally:
            WU._FFT_MIN_TAPS = keep
        if "hard" not in name:      # (a hard threshold may flip on a last-bit difference)
            np.testing.assert_allclose(got, direct[name], atol=5e-5 * np.abs(got).max(), rtol=0)
    got = WA.richardson_lucy(d.astype(np.float64) * 10 + 100, g["psf"].astype(np.float64), iterations=3, fft=True)
    assert got.dtype == np.float64
    np.testing.assert_allclose(got, g["rl_fft_f64"], atol=1e-9 * np.abs(g["rl_fft_f64"]).max(), rtol=0)


def test_fft_products_beat_the_banded_direct_form_for_a_129x129_psf(L):
    """a 129 x 129 PSF at 2048^2 - the FFT form of one circular product must be at
    least 5 x faster than the banded direct form (16 641 taps per pixel), with the same result."""
    H = W = 2048
    rng = np.random.default_rng(3)
    x = rng.standard_normal((H, W)).astype(np.float32)
    ky, kx = np.mgrid[0:129, 0:129]
    psf = np.exp(-((ky - 64.) ** 2 + (kx - 64.) ** 2) / 600.).astype(np.float32)
    psf /= psf.sum()
    ctx = L.default_context()
    p = L.Plan(ctx, H, W, L.B3SPLINE, 0)
    p.upload(L.PLANE_INPUT, x)
    pad = np.zeros((H, W), np.float32)
    pad[H // 2 - 64:H // 2 + 65, W // 2 - 64:W // 2 + 65] = psf
    S, A, B = L.PLANE_SCRATCH(6), L.PLANE_SCRATCH(7), L.PLANE_SCRATCH(8)
    p.upload(S, np.roll(pad, (H // 2, W // 2), axis=(0, 1)))
    p.fft_spectrum(S)

    def timed(fn, reps):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps
    t_fft = timed(lambda: p.fft_apply(L.PLANE_INPUT, A, False), 10)
    flipped = np.ascontiguousarray(psf[::-1, ::-1])
    t_dir = timed(lambda: p.filter2d(L.PLANE_INPUT, B, flipped, anchor=(64, 64), periodic=True), 2)
    a, b = p.download(A), p.download(B)
    p.close()
    try:
        with open(os.path.join(ROOT, "gpurun_out", "fft_vs_direct.txt"), "a") as f:
            f.write(f"2048^2, 129x129 PSF: fft {t_fft:.3f} ms, banded direct {t_dir:.3f} ms, ratio {t_dir / t_fft:.1f}\n")
    except OSError:
        pass
    assert float(np.abs(a - b).max()) <= 2e-5 * float(np.abs(x).max())
    assert t_dir >= 5 * t_fft, (t_fft, t_dir)


def test_denoise_with_given_noise_is_pipelined_over_pcie_and_bitwise_the_serial_sequence(L):
    """utils.denoise(img, sigmas, noise=<scalar>) on a large float32 image: every threshold is known up
    front, so upload, passes, thresholds + start of the sum, passes and download run as ONE pipelined
    host-to-host call over blocks of rows (wt_denoise_sum_host).  Same kernels on row sub-ranges: the
    result equals the serial sequence (host_pipeline off) bit for bit; row-strided inputs go down as
    they are; both families; the oracle agrees."""
    import wavelets_amd as WA
    from oracle import cref
    cref.build()
    img = rnd((2304, 2048), 9) * 2 + 1
    wide = rnd((2304, 2100), 10)
    view = wide[:, 11:11 + 2048]                                  # a row-strided view (stride 2100)
    for cls, fam, sig in ((WA.B3spline, "b3spline", [5, 3, 0, 0, 0, 0]), (WA.Triangle, "triangle", [4, 3, 2, 0, 0, 0, 0, 0]),
                          (WA.B3spline, "b3spline", [5, 0, 3, 0, 0])):
        for soft in (True, False):
            for src in (img, view):
                got = WA.denoise(src, list(sig), cls, noise=0.8, soft_threshold=soft)
                try:
                    L.set_option("host_pipeline", 0)
                    want = WA.denoise(np.ascontiguousarray(src), list(sig), cls, noise=0.8, soft_threshold=soft)
                finally:
                    L.set_option("host_pipeline", 1)
                np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        # against the C oracle (soft thresholds)
        planes = cref.decompose(img, len(sig), fam)
        for s, sg in enumerate(sig):
            if sg:
                cref.denoise_plane(planes[s], sg * 0.8 * cref.sigma_e(fam)[s])
        ref = cref.plane_sum(planes)
        got = WA.denoise(img, list(sig), cls, noise=0.8)
        assert float(np.abs(got - ref).max()) <= 4e-6 * float(np.abs(img).max())
    # the pipelined call really ran: its profile shows row-range launches of the denoise_sum kernel
    ctx = L.default_context()
    ctx.profile(True)
    ctx.profile_reset()
    WA.denoise(img, [5, 3, 0, 0, 0, 0], noise=0.8)
    ent = ctx.profile_entries()
    ctx.profile(False)
    assert ent["wt_denoise_sum_kernel"][0] > 4, ent


def test_windowed_riding_histogram_exact_median_and_fallback_when_the_window_misses(L):
    """Round 4: the first fused pass bins 21-bit keys of |w_0| in a window that a 4096-pixel sample of
    the input places around the predicted median (wt_median_window_kernel): get_noise then reads the plane
    ONCE more instead of twice.  The median stays EXACT: (a) ordinary images, both families, windowed on
    and off give np.median; (b) images built so that the sample mispredicts (zeros around every sample
    point; a plane whose magnitudes spread over 80 octaves) land in bin 0 / 2047 and are redone with the
    ordinary passes; (c) the select really took the short path where it should (one wt_hist launch)."""
    import wavelets_amd as WA
    ctx = L.default_context()
    H, W = 1100, 1300
    rng = np.random.default_rng(17)
    base = rng.standard_normal((H, W)).astype(np.float32)
    holes = base.copy()
    ys = ((2 * np.arange(64) + 1) * H) >> 7
    xs = ((2 * np.arange(64) + 1) * W) >> 7
    for y in ys:
        for x in xs:
            holes[max(y - 4, 0):y + 5, max(x - 4, 0):x + 5] = 0.0          # every sample sees |w_0| = 0
    spread = (base * np.float32(10.0) ** rng.integers(-12, 12, (H, W)).astype(np.float32)).astype(np.float32)
    spikes = np.zeros((H, W), np.float32)
    spikes[ys[:, None], xs[None, :]] = 1e6                                   # ... and here only the samples are non-zero
    for name, img in (("gauss", base), ("offset", base * 1e-3 + 40), ("holes", holes), ("spread", spread), ("spikes", spikes)):
        for cls in (WA.B3spline, WA.Triangle):
            for window in (1, 0):
                L.set_option("hist_window", window)
                try:
                    ctx.profile(True)
                    ctx.profile_reset()
                    c = WA.AtrousTransform(cls)(img, 3)
                    got = c.get_noise()
                    ent = ctx.profile_entries()
                    ctx.profile(False)
                finally:
                    L.set_option("hist_window", 1)
                want = np.median(np.abs(c.data[0])) / 0.6745 / c.sigma_e[0]
                assert got == want, (name, cls.__name__, window, got, want)
                nh = ent.get("wt_hist_kernel", (0, 0))[0]
                if window and name in ("gauss", "offset"):
                    assert nh == 1 and "wt_median_window_kernel" in ent, (name, ent)      # the short path
                if window and name in ("holes", "spikes"):
                    assert nh == 4, (name, ent)            # one pass in vain, then the ordinary three
                if not window:
                    assert nh == 2 and "wt_median_window_kernel" not in ent, (name, ent)
    # float64: the windowed histogram carries 22 bits; the select goes straight to its gather pass
    for name, img in (("gauss", base), ("holes", holes), ("spikes", spikes), ("spread", spread)):
        img64 = img.astype(np.float64)
        for window in (1, 0):
            L.set_option("hist_window", window)
            try:
                ctx.profile(True)
                ctx.profile_reset()
                c = WA.AtrousTransform(WA.B3spline)(img64, 3)
                got = c.get_noise()
                ent = ctx.profile_entries()
                ctx.profile(False)
            finally:
                L.set_option("hist_window", 1)
            assert c.data.dtype == np.float64
            assert got == np.median(np.abs(c.data[0])) / 0.6745 / c.sigma_e[0], (name, window)
            nh = ent.get("wt64_hist_kernel", (0, 0))[0]
            if window and name == "gauss":
                assert nh == 0 and ent["wt64_collect_kernel"][0] == 1 and "wt_median_window_kernel" in ent, ent
            if not window and name == "gauss":
                assert nh == 1 and ent["wt64_collect_kernel"][0] == 1, ent


def test_bench_emits_the_stored_measurement_when_a_later_phase_hangs():
    """The main multi-GPU measurement is stored before the optional placement A/B starts; if that later
    phase never finishes (here: a testing aid makes every rank sleep), the watchdog inside rank 0 emits
    the stored line - marked `time_limit_hit` - instead of losing the run, and the ranks leave with 0."""
    import subprocess
    env = dict(_clean_env(), WT_BENCH_TEST_HANG_AFTER_MAIN="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--size", "2048",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--no-build", "--time-limit", "70"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads(lines[0])
    assert "time_limit_hit" in out and out["n_gpus"] == 2 and out["value"] > 0 and out["halo_selfcheck"]["ok"]
    assert out["strip_planes"]["chosen"] == "hipMalloc" and "scattered_ms_per_step" not in out["strip_planes"]
    assert r.returncode == 0, r.returncode


def test_richardson_lucy_fft_on_sides_that_are_not_powers_of_two(L):
    """g24: richardson_lucy(fft=True) on 72 x 100 and 75 x 100 images (odd height: the reference's row
    anchors move by one) against the unmodified reference, float32 and float64, odd and even PSFs, and
    against the direct periodic form of the same products - twice: through the mixed-radix FFT on the
    image's own sides (round 5: 72 = 2^3 3^2, 75 = 3 5^2, 100 = 2^2 5^2) and through the periodically
    EXTENDED frame (utils._ExtendedFFT), which sides with a prime factor above 5 take."""
    from conftest import load_golden
    import wavelets_amd as WA
    from wavelets_amd import utils as WU
    g = load_golden("g24_rl_fft_nonpow2")
    calls = []
    keep_apply, keep_min, keep_own = WU._ExtendedFFT.apply, WU._FFT_MIN_TAPS, L.Plan.fft_apply
    keep_own64 = L.Plan64.fft_apply

    def counting(self, src, dst, conj):
        calls.append((self.Mh, self.Mw, bool(conj)))
        return keep_apply(self, src, dst, conj)

    def counting_own(self, src, dst, conj):
        calls.append((self.H, self.W, bool(conj)))
        return keep_own(self, src, dst, conj)

    def counting_own64(self, src, dst, conj):
        calls.append((self.H, self.W, bool(conj)))
        return keep_own64(self, src, dst, conj)

    try:
        for extended in (False, True):
            WU._FFT_FORCE_EXTENDED = extended
            if extended:
                WU._ExtendedFFT.apply = counting
                L.Plan.fft_apply, L.Plan64.fft_apply = keep_own, keep_own64
            else:
                L.Plan.fft_apply, L.Plan64.fft_apply = counting_own, counting_own64
            # frames: 72 + 2 * 12 = 96, 100 + 2 * 11 = 122 -> 125; 75 + 2 * 13 = 101 -> 108
            for tag, frame, own in (("a", (96, 125), (72, 100)), ("odd", (108, 125), (75, 100))):
                d = g[f"data_{tag}"]
                for name, psf, kw in ((f"rl_{tag}_soft", "psf", dict(iterations=3)),
                                      (f"rl_{tag}_even", "psf_even", dict(iterations=3, denoise_coefficients=(4, 2)))):
                    WU._FFT_MIN_TAPS = 1                                   # small images: force the FFT form
                    del calls[:]
                    got = WA.richardson_lucy(d.copy(), g[psf], fft=True, **kw)
                    if extended:
                        assert len(calls) == 6 and all(c[0] >= own[0] + g[psf].shape[0] - 1 and L.fft_supported(c[0], c[1])
                                                       for c in calls), calls
                        if psf == "psf":
                            assert calls == [(frame[0], frame[1], False), (frame[0], frame[1], True)] * 3, calls
                    else:
                        assert calls == [(own[0], own[1], False), (own[0], own[1], True)] * 3, calls
                    assert got.dtype == np.float32
                    np.testing.assert_allclose(got, g[name], atol=2e-4 * np.abs(g[name]).max(), rtol=2e-4)
                    WU._FFT_MIN_TAPS = 1 << 30                             # the direct periodic correlations
                    direct = WA.richardson_lucy(d.copy(), g[psf], fft=True, **kw)
                    np.testing.assert_allclose(got, direct, atol=5e-5 * np.abs(got).max(), rtol=0)
                WU._FFT_MIN_TAPS = 1
                del calls[:]
                got = WA.richardson_lucy(d.astype(np.float64) * 10 + 100, g["psf"].astype(np.float64), iterations=3, fft=True)
                assert got.dtype == np.float64 and len(calls) == 6
                np.testing.assert_allclose(got, g[f"rl_{tag}_f64"], atol=1e-9 * np.abs(g[f"rl_{tag}_f64"]).max(), rtol=0)
        # the default threshold: a 25 x 23 PSF (575 taps) on 72 x 100 takes the FFT on the image's own sides, and
        # stays on the direct form where only the extended frame (1.7x the image) is available
        WU._FFT_MIN_TAPS = keep_min
        del calls[:]
        WA.richardson_lucy(g["data_a"].copy(), g["psf"], fft=True, iterations=1)
        assert calls == []
        WU._FFT_FORCE_EXTENDED = False
        L.Plan.fft_apply = counting_own
        WA.richardson_lucy(g["data_a"].copy(), g["psf"], fft=True, iterations=1)
        assert calls == [(72, 100, False), (72, 100, True)], calls
    finally:
        WU._ExtendedFFT.apply, WU._FFT_MIN_TAPS, WU._FFT_FORCE_EXTENDED = keep_apply, keep_min, False
        L.Plan.fft_apply, L.Plan64.fft_apply = keep_own, keep_own64


def test_fft_beats_the_direct_form_on_3072_and_3066_square_images(L):
    """A 65 x 65 PSF on 3072 x 3072 (2^10 3: e.g. a full-disc EUV imager frame; the mixed-radix FFT on the image's
    own sides since round 5) and on 3066 x 3066 (2 3 7 73: through the extended 3200 x 3200 frame): one circular
    product against the banded direct periodic form."""
    from wavelets_amd import utils as WU
    k = 65
    rng = np.random.default_rng(31)
    psf = rng.uniform(0.1, 1.0, (k, k)).astype(np.float32)
    psf /= psf.sum()
    ctx = L.default_context()

    def timed(fn, n):
        fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(n):
            fn()
        return ctx.timer_stop() / n

    for side in (3072, 3066):
        x = rng.standard_normal((side, side), dtype=np.float32)
        p = L.Plan(ctx, side, side, L.B3SPLINE, 1)
        A, B, K = L.PLANE_SCRATCH(2), L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4)
        ext = None
        try:
            p.upload(L.PLANE_INPUT, x)
            if L.fft_supported(side, side):
                assert side == 3072
                padded = np.zeros((side, side), np.float32)
                h = side // 2 - k // 2
                padded[h:h + k, h:h + k] = psf
                p.upload(K, np.roll(padded, (side // 2, side // 2), axis=(0, 1)))
                p.fft_spectrum(K)
                fn, what = (lambda: p.fft_apply(L.PLANE_INPUT, A, False)), "mixed-radix fft"
            else:
                ext = WU._ExtendedFFT(p, False, psf)
                assert ext.ok and (ext.Mh, ext.Mw) == (3200, 3200) and ext.worth_it(WU._FFT_MIN_TAPS)
                ext.prepare(WU.B3spline(2))
                fn, what = (lambda: ext.apply(L.PLANE_INPUT, A, False)), "extended-frame fft (3200^2)"
            kern, kw = WU._periodic_operand(np.ascontiguousarray(psf[::-1, ::-1]), k - 1 - k // 2, k - 1 - k // 2)
            t_fft = timed(fn, 5)
            t_dir = timed(lambda: p.filter2d(L.PLANE_INPUT, B, kern, **kw), 2)
            a, b = p.download(A), p.download(B)
            np.testing.assert_allclose(a, b, atol=2e-5 * np.abs(b).max(), rtol=0)
            try:
                with open(os.path.join(ROOT, "gpurun_out", "fft_vs_direct.txt"), "a") as f:
                    f.write(f"{side}^2, 65x65 PSF: {what} {t_fft:.3f} ms, banded direct {t_dir:.3f} ms, ratio {t_dir / t_fft:.1f}\n")
            except OSError:
                pass
            assert t_dir >= 1.5 * t_fft, (side, t_fft, t_dir)
        finally:
            if ext is not None:
                ext.close()
            p.close()


@pytest.mark.parametrize("dtype", ["?", "i1", "u1", "<i2", "<u2", "<i4", "<u4", "<i8", "<u8",
                                   ">i2", ">u2", ">i4", ">u4", ">i8", ">u8", ">f4", ">f8", "<f4"])
def test_integer_and_big_endian_images_are_widened_on_the_device_exactly(L, dtype):
    """Round 4 (late): the reference recasts integer and big-endian images to float64 on the host (ref
    wavelets.py:297, 319-320); here they cross PCIe as they are and wt64_upload_int widens (and byte-swaps)
    them on the device.  The plane must hold exactly numpy's astype(float64) - including 64-bit integers
    beyond 2^53 (round to nearest even) and float specials - for contiguous and row-strided sources."""
    ctx = L.default_context()
    rng = np.random.default_rng(3)
    H, W = 37, 1030
    dt = np.dtype(dtype)
    if dt.kind == "b":
        a = rng.random((H, W)) < 0.5
    elif dt.kind == "f":
        a = (rng.standard_normal((H, W)) * 10.0 ** rng.integers(-20, 20, (H, W))).astype(dt)
        a[0, :5] = [0.0, -0.0, np.inf, -np.inf, np.nan]
    else:
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, (H, W), dtype=dt.newbyteorder("="), endpoint=True).astype(dt)
        a[0, :4] = [info.min, info.max, info.max - 1 if info.max > 1 else 0, 0]
    assert a.dtype == dt and L.Plan64.device_widens(dt) == (dt.str != "<f4")
    wide = np.zeros((H, W + 7), dtype=dt)
    wide[:, :W] = a
    want = a.astype(np.float64)
    p = L.Plan64(ctx, H, W, (0.25, 0.5, 0.25), 1)
    try:
        for src in (a, wide[:, :W]):                         # contiguous rows, and rows 7 elements apart
            p.fill(L.PLANE_INPUT, -1.0)
            p.upload(L.PLANE_INPUT, src)
            got = p.download(L.PLANE_INPUT)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), dtype
    finally:
        p.close()


def test_integer_image_through_the_api_equals_its_float64_promotion(L):
    """AtrousTransform / denoise / wow on an int16 image (a FITS frame) = the same on image.astype(float64),
    bit for bit, float64 results - and the integer really travelled as an integer."""
    import wavelets_amd as WA
    rng = np.random.default_rng(4)
    img = (1000 + 80 * rng.standard_normal((600, 700))).astype(np.int16)
    seen = []
    keep = L.Plan64.upload

    def spy(self, plane, host):
        seen.append(np.asarray(host).dtype)
        return keep(self, plane, host)

    L.Plan64.upload = spy
    try:
        c = WA.AtrousTransform(WA.B3spline)(img, 4)
        d = WA.denoise(img, [5, 3, 2], WA.Triangle)
        w, _ = WA.wow(img)
    finally:
        L.Plan64.upload = keep
    assert seen and all(t == np.int16 for t in seen), seen
    f = img.astype(np.float64)
    c64 = WA.AtrousTransform(WA.B3spline)(f, 4)
    assert c.data.dtype == np.float64 and np.array_equal(np.asarray(c.data), np.asarray(c64.data))
    d64 = WA.denoise(f, [5, 3, 2], WA.Triangle)
    assert d.dtype == np.float64 and np.array_equal(d, d64)
    w64, _ = WA.wow(f)
    assert w.dtype == np.float64 and np.array_equal(w, w64)
    # a big-endian float32 frame (FITS): recast to float64 by the reference ('>f4' in its list, ref:297)
    be = (f * 0.37).astype(">f4")
    seen.clear()
    L.Plan64.upload = spy
    try:
        cb = WA.AtrousTransform(WA.Triangle)(be, 3)
    finally:
        L.Plan64.upload = keep
    assert seen == [np.dtype(">f4")] and cb.data.dtype == np.float64
    assert np.array_equal(np.asarray(cb.data), np.asarray(WA.AtrousTransform(WA.Triangle)(be.astype(np.float64), 3).data))


@pytest.mark.parametrize("dtype", ["?", "i1", "u1", "<i2", "<u2", "<i4", "<u4", "<i8", "<u8", ">i2", ">u2", ">i4", ">u4",
                                   ">i8", ">u8", ">f4", ">f8"])
def test_float32_plans_widen_other_element_types_on_the_device(L, dtype):
    """wt_upload_int: what the reference does not recast to float64 (uint8 pictures, raw big-endian FITS
    integers, ...) is served in float32 - widened and byte-swapped on the device, bit for bit numpy's
    astype(float32) (round to nearest even where float32 cannot hold the integer); contiguous and strided rows."""
    ctx = L.default_context()
    rng = np.random.default_rng(5)
    H, W = 41, 1028
    dt = np.dtype(dtype)
    if dt.kind == "b":
        a = rng.random((H, W)) < 0.5
    elif dt.kind == "f":
        a = (rng.standard_normal((H, W)) * 10.0 ** rng.integers(-10, 10, (H, W))).astype(dt)
        a[0, :5] = [0.0, -0.0, np.inf, -np.inf, np.nan]
    else:
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, (H, W), dtype=dt.newbyteorder("="), endpoint=True).astype(dt)
        a[0, :4] = [info.min, info.max, info.max - 1 if info.max > 1 else 0, 0]
    assert L.device_widens(dt)
    wide = np.zeros((H, W + 5), dtype=dt)
    wide[:, :W] = a
    with np.errstate(over="ignore"):
        want = a.astype(np.float32)
    p = L.Plan(ctx, H, W, L.B3SPLINE, 1)
    try:
        for src in (a, wide[:, :W]):
            p.fill(L.PLANE_INPUT, -1.0)
            p.upload(L.PLANE_INPUT, src)
            got = p.download(L.PLANE_INPUT)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), dtype
    finally:
        p.close()


def test_uint8_and_big_endian_integer_images_through_the_float32_api(L):
    """A uint8 picture and a raw big-endian int16 FITS frame (neither is in the reference's recast list: served
    in float32) through AtrousTransform / denoise / wow: identical to the same calls on image.astype(float32),
    and the integers travelled as integers."""
    import wavelets_amd as WA
    rng = np.random.default_rng(6)
    u8 = np.clip(120 + 40 * rng.standard_normal((520, 640)), 0, 255).astype(np.uint8)
    be = (1000 + 80 * rng.standard_normal((520, 640))).astype(">i2")
    seen = []
    keep = L.Plan.upload

    def spy(self, plane, host):
        seen.append(np.asarray(host).dtype)
        return keep(self, plane, host)

    for img in (u8, be):
        del seen[:]
        L.Plan.upload = spy
        try:
            c = WA.AtrousTransform(WA.B3spline)(img, 4)
            d = WA.denoise(img, [5, 3, 2], WA.Triangle)
            w, _ = WA.wow(img)
            s = np.sum(WA.AtrousTransform(WA.B3spline)(img, 3, with_sum=True), axis=0)
        finally:
            L.Plan.upload = keep
        assert seen and all(t == img.dtype for t in seen), seen
        f = img.astype(np.float32)
        assert c.data.dtype == np.float32
        assert np.array_equal(np.asarray(c.data), np.asarray(WA.AtrousTransform(WA.B3spline)(f, 4).data))
        assert d.dtype == np.float32 and np.array_equal(d, WA.denoise(f, [5, 3, 2], WA.Triangle))
        assert np.array_equal(w, WA.wow(f)[0])
        assert np.array_equal(s, np.sum(WA.AtrousTransform(WA.B3spline)(f, 3, with_sum=True), axis=0))
        sf = WA.B3spline(2)
        assert np.array_equal(WA.sdev_loc(img, sf, 1), WA.sdev_loc(f, sf, 1))
        assert np.array_equal(WA.convolution(img, sf, 2), WA.convolution(f, sf, 2))
        assert np.array_equal(WA.atrous_convolution(img, sf.kernel, s=1), WA.atrous_convolution(f, sf.kernel, s=1))
        psf = np.outer(np.hanning(7), np.hanning(7)).astype(np.float32)
        psf /= psf.sum()
        assert np.array_equal(WA.richardson_lucy(img, psf, iterations=2), WA.richardson_lucy(f, psf, iterations=2))


def test_float64_wow_scale_equals_smooth_plus_update_bitwise(L):
    """wt64_wow_scale (row pass of the squares + column pass with the wow update as its epilogue, in place)
    against wt64_smooth(square) + wt64_wow_update on a second plan: identical bits for the coefficient
    plane and the gamma accumulator - with and without threshold, noise map, gamma, both families - and
    wow() of a float64 image still matches the float64 golden cases (test_float64_engine_vs_golden)."""
    ctx = L.default_context()
    rng = np.random.default_rng(12)
    H, W = 300, 517
    c0 = rng.standard_normal((H, W)) * 3
    nz0 = rng.uniform(0.5, 2.0, (H, W))
    for taps in ((1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16), (0.25, 0.5, 0.25), (0.2, 0.6, 0.2)):   # (the last: user taps, generic kernels)
        for s in (0, 2, 5):
            for tau, soft, use_noise, use_gamma in ((0.0, True, False, False), (1.3, True, False, True),
                                                    (0.8, False, True, False), (2.0, True, True, True)):
                a, b = L.Plan64(ctx, H, W, taps, 1), L.Plan64(ctx, H, W, taps, 1)
                try:
                    NZ, GM, PW = L.PLANE_SCRATCH(2), L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4)
                    for p in (a, b):
                        p.upload(0, c0)
                        p.upload(NZ, nz0)
                        p.fill(GM, 0.25)
                    npl = NZ if use_noise else L.PLANE_NONE
                    gpl = GM if use_gamma else L.PLANE_NONE
                    a.wow_scale(0, s, tau, soft, npl, 0.7, gpl)
                    b.smooth(0, PW, s, True)
                    b.wow_update(0, PW, tau, soft, npl, 0.7, gpl)
                    assert np.array_equal(a.download(0).view(np.uint64), b.download(0).view(np.uint64)), (taps, s, tau)
                    assert np.array_equal(a.download(GM).view(np.uint64), b.download(GM).view(np.uint64)), (taps, s, tau)
                finally:
                    a.close()
                    b.close()


def test_float64_two_pixel_kernels_equal_the_one_pixel_ones_bitwise(L):
    """wt64_rows2 / wt64_cols2 / wt64_wow_axis2 (two pixels per thread, 16-byte accesses; even widths, even
    dilations for the row filter) against the one-pixel kernels (option f64_pairs = 0): smooth with and
    without squared input, the detail plane of a generic-path scale, wt64_wow_scale - several dilations,
    an odd width (falls back), images narrower than the taps' reach (reflected pairs)."""
    ctx = L.default_context()
    rng = np.random.default_rng(21)
    taps5 = (1 / 16, 1 / 4, 3 / 8, 1 / 4, 1 / 16)
    for (H, W) in ((130, 518), (64, 10), (33, 257)):
        a = rng.standard_normal((H, W)) * 2 + 5
        nz = rng.uniform(0.5, 2.0, (H, W))
        res = {}
        for pairs in (1, 0):
            L.set_option("f64_pairs", pairs)
            L.set_option("stencil64", 0)          # (round 5: built-in taps take wt_stencil.h otherwise; these are the generic kernels)
            try:
                p = L.Plan64(ctx, H, W, taps5, 1)
                out = []
                try:
                    A, B, NZ, GM = L.PLANE_SCRATCH(2), L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4), L.PLANE_SCRATCH(5)
                    p.upload(A, a)
                    p.upload(NZ, nz)
                    for s in (0, 1, 3, 6):
                        p.smooth(A, B, s)
                        out.append(p.download(B))
                        p.smooth(A, B, s, True)
                        out.append(p.download(B))
                        p.copy(A, 0)
                        p.fill(GM, 0.5)
                        p.wow_scale(0, s, 1.1, True, NZ, 0.9, GM)
                        out.append(p.download(0))
                        out.append(p.download(GM))
                finally:
                    p.close()
                res[pairs] = out
            finally:
                L.set_option("f64_pairs", 1)
                L.set_option("stencil64", 1)
        assert len(res[0]) == len(res[1]) == 16
        for k, (u, v) in enumerate(zip(res[1], res[0])):
            assert np.array_equal(u.view(np.uint64), v.view(np.uint64)), ((H, W), k)
