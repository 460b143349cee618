"""GPU parity tests added in round 2 (run with -m gpu on an MI355X).

* the literal bench step (wt_decompose_sum, 8192^2 B3spline L=6: the kernel instantiations
  bench.py times) against the C oracle, all 7 planes + reconstruction;
* the fused passes' fast addressing (aligned-group reflection, single-bounce rows) against the
  generic addressing, bit for bit, on shapes at and around its admission thresholds;
* bench.py --gpus 2 through the plain command (self-launching, RCCL ranks reported).
"""
import json
import os
import subprocess
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


@pytest.fixture(scope="module")
def C():
    from oracle import cref
    cref.build()
    return cref


def rnd(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def close(a, b, atol):
    d = float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
    assert d <= atol, f"max abs diff {d:.3e} > {atol:.3e}"


@pytest.mark.parametrize("fam,level", [("b3spline", 6), ("triangle", 8)])
def test_bench_step_8192_vs_c_oracle(L, C, fam, level):
    """wt_decompose_sum at the headline size and on the cfg3 transform: every plane and the
    carried reconstruction vs oracle decompose + plane_sum, tol 1e-5 * max|a| (fp32 sum order:
    the engine filters separably, the oracle with the dense K x K kernel)."""
    a = rnd((8192, 8192), 0)
    amax = float(np.abs(a).max())
    f = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam]
    plan = L.Plan(L.default_context(), 8192, 8192, f, level)
    plan.upload(L.PLANE_INPUT, a)
    plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, L.FLAG_FUSED)
    ref = C.decompose(a, level, fam)
    for s in range(level + 1):
        close(plan.download(s), ref[s], 1e-5 * amax)
    close(plan.download(L.PLANE_OUT), C.plane_sum(ref), 2e-5 * amax)
    plan.close()


FAST_SHAPES = [
    # (H, W, family, level)   W % 4 == 0 everywhere (fast path admissible)
    (8, 32, "b3spline", 3),         # H below D*(LAT+1): generic anyway
    (15, 32, "b3spline", 3),        # H == LAT_IN + 1 (d1x3: 14 + 1), W == HX
    (16, 36, "b3spline", 3),
    (120, 128, "b3spline", 6),      # d8x3 needs H >= 8 * 15 = 120, W >= 128
    (121, 132, "b3spline", 6),
    (119, 128, "b3spline", 6),      # second pass generic, first fast
    (300, 1000, "b3spline", 6),
    (1100, 2100, "b3spline", 6),
    (700, 964, "triangle", 8),      # d64x2 (triangle: 3 * 64 + 64 = 256 rows)
    (513, 260, "triangle", 5),
    (2048, 1024, "b3spline", 8),    # b3 d64x2: H >= 64 * 7 = 448, W >= 384
]


def _all_outputs(L, plan, level, flags):
    for s in range(level + 1):
        plan.fill(s, np.nan)
    plan.fill(L.PLANE_OUT, np.nan)
    plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, flags)
    out = [plan.download(s).view(np.uint32).copy() for s in range(level + 1)]
    out.append(plan.download(L.PLANE_OUT).view(np.uint32).copy())
    plan.decompose(L.PLANE_INPUT, level, flags)
    return out + [plan.download(s).view(np.uint32).copy() for s in range(level + 1)]


@pytest.mark.parametrize("H,W,fam,level", FAST_SHAPES)
def test_fused_fast_addressing_is_bitwise_the_generic_one(L, H, W, fam, level):
    f = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam]
    plan = L.Plan(L.default_context(), H, W, f, level)
    plan.upload(L.PLANE_INPUT, rnd((H, W), H * 7 + W))
    got = {}
    try:
        for mode in (1, 0):
            L.set_option("fused_fast", mode)
            got[mode] = _all_outputs(L, plan, level, L.FLAG_FUSED)
    finally:
        L.set_option("fused_fast", 1)
    for i, (x, y) in enumerate(zip(got[1], got[0])):
        np.testing.assert_array_equal(x, y, err_msg=f"output {i}")
    plan.close()


@pytest.mark.parametrize("fam,level,k,shape", [("b3spline", 6, 3, (2100, 1100)),
                                               ("triangle", 8, 2, (1024, 512))])
def test_fused_fast_addressing_on_strips_is_bitwise_the_generic_whole_image(L, fam, level, k, shape):
    """strips (rows beyond a strip come from its margins, the global border reflects) with the
    fast addressing vs the unsharded plan with the generic addressing"""
    from test_gpu_strips import exchange_all, gather, make_strips
    f = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam]
    ctx = L.default_context()
    img = rnd(shape, 31)
    whole = L.Plan(ctx, shape[0], shape[1], f, level)
    whole.upload(L.PLANE_INPUT, img)
    try:
        L.set_option("fused_fast", 0)
        whole.decompose(L.PLANE_INPUT, level, L.FLAG_FUSED)
        whole.plane_sum(0, level + 1)
    finally:
        L.set_option("fused_fast", 1)
    plans = make_strips(L, ctx, img, f, level, k)
    cur = L.PLANE_INPUT
    sched = L.schedule(f, level, True)
    for i, (s0, ns, halo) in enumerate(sched):
        nxt = level if s0 + ns == level else L.PLANE_SCRATCH(i & 1)
        exchange_all(L, plans, cur, halo)
        for p in plans:
            p.decompose_pass_sum(cur, nxt, s0, ns, L.FLAG_FUSED | L.FLAG_NO_EXCHANGE, L.PLANE_OUT,
                                 first=i == 0, last=i == len(sched) - 1)
        cur = nxt
    for s in range(level + 1):
        np.testing.assert_array_equal(gather(plans, s), whole.download(s), err_msg=f"plane {s}")
    np.testing.assert_array_equal(gather(plans, L.PLANE_OUT), whole.download(L.PLANE_OUT))
    for p in plans + [whole]:
        p.close()


def test_bench_self_launches_two_ranks_on_the_shared_gpu():
    """`python bench.py --gpus 2` (no torchrun around it) must start its own ranks, report
    n_gpus == 2 and the communicator's own size, and exit 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu",
                        "--size", "2048", "--steps", "3", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["config"]["image"] == [2048, 2048] and "2048x2048" in out["metric"]
    assert out["roofline"]["frac"] > 0
