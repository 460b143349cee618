"""GPU: the sharded path must equal the unsharded path BIT FOR BIT.

"Virtual strips": k strip plans (rank i of k) live on the one GPU the test box has; the RCCL
send/recv group of wt_halo_exchange is replaced by wt_halo_exchange_local (device-to-device
copies of exactly the same rows into exactly the same margins) and every kernel runs with
FLAG_NO_EXCHANGE.  Everything else - strip geometry, margins, global-border reflection,
chain/fused kernels reading neighbour rows from the margins - is the production path.
The RCCL transport itself is covered by test_rccl_single_rank_selftest and, on CPU, the
schedule/partition logic by tests/test_strips_gloo_cpu.py.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as entry
    entry.build()
    from wavelets_amd import _lib
    return _lib


def rnd(shape, seed=0):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def make_strips(L, ctx, img, fam, level, k, halo=0):
    from wavelets_amd.parallel import partition_rows
    H, W = img.shape
    plans = []
    for r, (row0, n) in enumerate(partition_rows(H, k)):
        p = L.Plan(ctx, H, W, fam, level, row0=row0, nrows=n, halo_rows=halo, rank=r, nranks=k)
        p.upload(L.PLANE_INPUT, img[row0:row0 + n])
        plans.append(p)
    return plans


def exchange_all(L, plans, plane, rows):
    for up, lo in zip(plans[:-1], plans[1:]):
        L.Plan.halo_exchange_local(up, lo, plane, rows)


def gather(plans, plane):
    return np.concatenate([p.download(plane) for p in plans])


@pytest.mark.parametrize("fam_name,level,fused,k,shape", [
    ("b3spline", 6, True, 2, (512, 300)),
    ("b3spline", 6, True, 4, (1024, 1030)),
    ("b3spline", 5, True, 3, (700, 257)),
    ("triangle", 8, True, 2, (1024, 192)),
    ("b3spline", 4, False, 3, (200, 130)),
    ("triangle", 3, False, 5, (170, 64)),
])
def test_sharded_decompose_equals_unsharded_bitwise(L, fam_name, level, fused, k, shape):
    fam = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam_name]
    ctx = L.default_context()
    img = rnd(shape, 11)
    flags = L.FLAG_FUSED if fused else 0
    whole = L.Plan(ctx, shape[0], shape[1], fam, level)
    whole.upload(L.PLANE_INPUT, img)
    whole.decompose(L.PLANE_INPUT, level, flags)

    plans = make_strips(L, ctx, img, fam, level, k)
    cur = L.PLANE_INPUT
    for i, (s0, ns, halo) in enumerate(L.schedule(fam, level, fused)):
        nxt = level if s0 + ns == level else L.PLANE_SCRATCH(i & 1)
        exchange_all(L, plans, cur, halo)
        for p in plans:
            p.decompose_pass(cur, nxt, s0, ns, flags | L.FLAG_NO_EXCHANGE)
        cur = nxt
    for s in range(level + 1):
        np.testing.assert_array_equal(gather(plans, s), whole.download(s), err_msg=f"plane {s}")
    # plane sum is pointwise: trivially shards
    for p in plans:
        p.plane_sum(0, level + 1)
    whole.plane_sum(0, level + 1)
    np.testing.assert_array_equal(gather(plans, L.PLANE_OUT), whole.download(L.PLANE_OUT))


@pytest.mark.parametrize("fam_name,level,k,shape", [
    ("b3spline", 6, 3, (768, 1100)),
    ("triangle", 8, 2, (1024, 160)),
    ("b3spline", 3, 4, (130, 70)),
    ("triangle", 5, 2, (300, 257)),
])
def test_sharded_decompose_sum_equals_unsharded_bitwise(L, fam_name, level, k, shape):
    """the sum carried through the fused passes (wt_decompose_sum) on strips: planes AND the
    reconstruction equal the unsharded two-call result bit for bit"""
    fam = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam_name]
    ctx = L.default_context()
    img = rnd(shape, 13)
    whole = L.Plan(ctx, shape[0], shape[1], fam, level)
    whole.upload(L.PLANE_INPUT, img)
    whole.decompose(L.PLANE_INPUT, level, L.FLAG_FUSED)
    whole.plane_sum(0, level + 1)
    plans = make_strips(L, ctx, img, fam, level, k)
    cur = L.PLANE_INPUT
    sched = L.schedule(fam, level, True)
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


def test_sharded_operators_equal_unsharded(L):
    """smooth / smooth of squares / local variance / bilateral conv on strips with halos"""
    fam = L.B3SPLINE
    ctx = L.default_context()
    img = rnd((640, 200), 12)
    whole = L.Plan(ctx, 640, 200, fam, 0)
    whole.upload(L.PLANE_INPUT, img)
    plans = make_strips(L, ctx, img, fam, 0, 4, halo=64)
    S0, S1 = L.PLANE_SCRATCH(3), L.PLANE_SCRATCH(4)
    for s in (0, 2, 5):
        exchange_all(L, plans, L.PLANE_INPUT, 2 << s)
        for sq in (False, True):
            whole.smooth(L.PLANE_INPUT, S0, s, sq)
            for p in plans:
                p.smooth(L.PLANE_INPUT, S0, s, sq, flags=L.FLAG_NO_EXCHANGE)
            np.testing.assert_array_equal(gather(plans, S0), whole.download(S0))
        whole.local_variance(L.PLANE_INPUT, S0, s, 1.5, 2.0)
        whole.bilateral_conv(L.PLANE_INPUT, S0, S1, s)
        for p in plans:
            p.local_variance(L.PLANE_INPUT, S0, s, 1.5, 2.0, flags=L.FLAG_NO_EXCHANGE)
            p.bilateral_conv(L.PLANE_INPUT, S0, S1, s, flags=L.FLAG_NO_EXCHANGE)
        np.testing.assert_array_equal(gather(plans, S0), whole.download(S0))
        np.testing.assert_array_equal(gather(plans, S1), whole.download(S1))


def test_strip_errors(L):
    ctx = L.default_context()
    with pytest.raises(L.WatrooHipError, match="single strip must cover"):
        L.Plan(ctx, 64, 64, L.B3SPLINE, 2, row0=0, nrows=32)
    p = L.Plan(ctx, 256, 64, L.B3SPLINE, 6, row0=0, nrows=128, rank=0, nranks=2)
    assert p.halo == 112                           # cumulative halo of the (3,3) fused pass
    with pytest.raises(L.WatrooHipError, match="no RCCL communicator"):
        p.decompose(L.PLANE_INPUT, 6)              # multi-rank plan needs wt_ctx_comm_init
    q = L.Plan(ctx, 256, 64, L.B3SPLINE, 2, row0=0, nrows=128, halo_rows=4, rank=0, nranks=2)
    with pytest.raises(L.WatrooHipError, match="halo"):
        q.smooth(L.PLANE_INPUT, L.PLANE_OUT, 4, flags=L.FLAG_NO_EXCHANGE)   # needs 32 rows


def test_strip_transform_single_rank(L):
    """StripTransform with nranks == 1 is the plain engine (what bench.py runs at N=1)."""
    from oracle import atrous_numpy as O
    from wavelets_amd.parallel import StripTransform
    img = rnd((256, 192), 13)
    st = StripTransform(L.default_context(), 256, 192, 4)
    st.upload(img)
    st.decompose()
    ref = O.Coeffs(O.atrous_standard(img, 4), "b3spline")
    np.testing.assert_allclose(st.get_noise(), ref.get_noise(), rtol=1e-5)
    st.denoise([5, 3])
    ref.denoise([5, 3])
    np.testing.assert_allclose(st.sum(), ref.data.sum(axis=0), atol=1e-5 * np.abs(img).max())
    st6 = StripTransform(L.default_context(), 256, 192, 6)
    st6.upload(img)
    rec = st6.decompose_sum()
    st6.decompose()
    np.testing.assert_array_equal(rec, st6.sum())


def test_paste_and_crop_scatter_gather_strips(L):
    """wt_paste_plane / wt_crop_plane move row strips between resident plans (no host trip)."""
    from wavelets_amd.parallel import partition_rows
    ctx = L.default_context()
    a = np.random.default_rng(3).standard_normal((90, 70)).astype(np.float32)
    whole = L.Plan(ctx, 90, 70, L.B3SPLINE, 1)
    whole.fill(L.PLANE_INPUT, 0.0)
    for r, (row0, n) in enumerate(partition_rows(90, 4)):
        p = L.Plan(ctx, 90, 70, L.B3SPLINE, 1, row0=row0, nrows=n, rank=r, nranks=4)
        p.upload(L.PLANE_INPUT, a[row0:row0 + n])
        p.paste_into(whole, L.PLANE_INPUT, L.PLANE_INPUT, row0, 0)
        p.crop_from(whole, L.PLANE_INPUT, L.PLANE_OUT, row0, 0)
        assert np.array_equal(p.download(L.PLANE_OUT), a[row0:row0 + n])
        with pytest.raises(L.WatrooHipError, match="outside"):
            p.paste_into(whole, L.PLANE_INPUT, L.PLANE_INPUT, 90 - n + 1, 0)
        p.close()
    assert np.array_equal(whole.download(L.PLANE_INPUT), a)
    whole.close()


def test_large_geometry_device_side_checks():
    """tools/check_large.py at 16384^2 (the per-GPU strip geometry of the N=8 bench is
    4096 x 32768; the script at 32768 takes ~15 s and ~90 GB - run it by hand)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "check_large.py"), "16384"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "check_large: OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("nranks,shape", [(2, (1536, 1100)), (3, (1200, 520))])
def test_real_rccl_ranks_share_the_gpu(nranks, shape):
    """The production RCCL sequence with REAL ranks (tools/check_rccl_ranks.py): one process per
    rank started by wavelets_amd.launch (no torch in any rank: the system ROCm stack, as in the
    bench), unique-id broadcast over a local socket, ncclCommInitRank, grouped
    ncclSend/ncclRecv of the halo rows on the compute stream before every pass, all-reduced
    histograms / moments.  The box has one GPU, so every rank gets its own NCCL_HOSTID: RCCL
    treats them as separate hosts and carries the rows over its socket transport.  Each rank
    compares planes, reconstruction, noise and denoised sum bit for bit with the unsharded plan."""
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [_sys.executable, os.path.join(root, "tools", "check_rccl_ranks.py"), "--ranks", str(nranks),
           "--shape", str(shape[0]), str(shape[1])]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from conftest import run_ranks
    r = run_ranks(cmd, env, f"ranks{nranks}", timeout=420)
    assert r.returncode == 0 and "0 mismatches in total" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "torch not imported" in r.stdout


@pytest.mark.parametrize("fam_name,level,nrows,W,rank,nranks", [
    ("b3spline", 6, 700, 1100, 1, 3),      # both neighbours: two edge ranges + interior
    ("b3spline", 6, 512, 300, 0, 2),       # bottom neighbour only
    ("b3spline", 5, 333, 257, 2, 3),       # top neighbour only, d1x3 + d8x2
    ("triangle", 8, 1000, 192, 1, 3),      # two four-scale passes: D = 1, 16 sub-ranges
])
def test_split_launches_equal_whole_pass_bitwise(L, fam_name, level, nrows, W, rank, nranks):
    """The overlapped multi-GPU schedule launches a pass as edge rows + interior rows (row
    sub-ranges of the fused kernel).  Option split_dry does exactly that on a strip plan without
    any exchange: every plane and the carried sum must equal the whole-pass launch bit for bit."""
    fam = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}[fam_name]
    ctx = L.default_context()
    plan = L.Plan(ctx, nranks * nrows, W, fam, level, row0=rank * nrows, nrows=nrows, rank=rank, nranks=nranks)
    plan.upload(L.PLANE_INPUT, rnd((nrows, W), 21))
    flags = L.FLAG_FUSED | L.FLAG_NO_EXCHANGE
    got = {}
    try:
        for mode in (0, 1):
            L.set_option("split_dry", mode)
            for s in range(level + 1):
                plan.fill(s, 0.0)
            plan.fill(L.PLANE_OUT, 0.0)
            plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, flags)
            got[mode] = [plan.download(s).view(np.uint32).copy() for s in list(range(level + 1)) + [L.PLANE_OUT]]
            plan.decompose(L.PLANE_INPUT, level, flags)
            got[mode] += [plan.download(s).view(np.uint32).copy() for s in range(level + 1)]
    finally:
        L.set_option("split_dry", 0)
    for a, b in zip(got[0], got[1]):
        np.testing.assert_array_equal(a, b)
    plan.close()
