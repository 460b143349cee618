"""CPU model of the multi-GPU path: world_size 2, 3 and 8 over gloo.

What is under test is the HOST LOGIC that the GPU path shares - the row partition
(wavelets_amd.parallel.partition_rows), the pass schedule and its halo sizes
(wt_schedule in libwatroo_hip.so) - with gloo send/recv standing in for the RCCL halo
exchange and the numpy oracle standing in for the kernels.  Each rank holds only its strip,
exchanges exactly the rows the schedule prescribes before each pass, and must reproduce its
rows of the unsharded transform BIT FOR BIT.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange(dist, torch, rank, world, own, halo):
    """Strip neighbours swap `halo` boundary rows (the RCCL send/recv group of
    wt_halo_exchange): returns (rows_from_upper_neighbour, rows_from_lower_neighbour)."""
    reqs, up, dn = [], None, None
    W = own.shape[1]
    if rank > 0:
        up = torch.empty((halo, W), dtype=torch.float32)
        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(own[:halo])), rank - 1))
        reqs.append(dist.irecv(up, rank - 1))
    if rank < world - 1:
        dn = torch.empty((halo, W), dtype=torch.float32)
        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(own[-halo:])), rank + 1))
        reqs.append(dist.irecv(dn, rank + 1))
    for r in reqs:
        r.wait()
    return (None if up is None else up.numpy()), (None if dn is None else dn.numpy())


def _worker(rank, world, port, H, W, family, level, fused, result_dir):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    from oracle import atrous_numpy as O
    from wavelets_amd import _lib
    from wavelets_amd.parallel import partition_rows, required_halo

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    entry.build()
    fam_id = {"b3spline": _lib.B3SPLINE, "triangle": _lib.TRIANGLE}[family]
    img = np.random.default_rng(5).standard_normal((H, W)).astype(np.float32)
    row0, nrows = partition_rows(H, world)[rank]
    assert required_halo(fam_id, level, fused) <= min(n for _, n in partition_rows(H, world))
    cur = img[row0:row0 + nrows].copy()            # this rank only ever touches its strip
    planes = np.empty((level + 1, nrows, W), np.float32)
    for s0, ns, halo in _lib.schedule(fam_id, level, fused):
        up, dn = _exchange(dist, torch, rank, world, cur, halo)
        ext = np.concatenate([a for a in (up, cur, dn) if a is not None])
        top = 0 if up is None else halo
        for s in range(s0, s0 + ns):
            nxt = O.convolution(ext, family, s)     # symmetric pad at ext edges: exact at the
            planes[s] = (ext - nxt)[top:top + nrows]  # global border, contaminates < halo rows
            ext = nxt                                 # at interior strip edges
        cur = ext[top:top + nrows].copy()
    planes[level] = cur
    ref = O.atrous_standard(img, level, family)[:, row0:row0 + nrows]
    ok = np.array_equal(planes, ref)
    # global scalars: all-reduced moments equal the unsharded ones
    mom = torch.tensor([planes[0].astype(np.float64).sum(), float(planes[0].size)],
                       dtype=torch.float64)
    dist.all_reduce(mom)
    full0 = O.atrous_standard(img, level, family)[0].astype(np.float64)
    ok_mom = abs(mom[0].item() - full0.sum()) < 1e-6 and mom[1].item() == full0.size
    with open(os.path.join(result_dir, f"r{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} {int(ok_mom)}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,family,level,fused", [
    (2, 96, 40, "b3spline", 4, True),      # passes (0,3,halo 14), (3,1,halo 16)
    (3, 100, 37, "b3spline", 3, True),     # ragged partition 34/33/33, one fused pass
    (2, 128, 24, "triangle", 5, True),     # (0,3,7), (3,2,24)
    (3, 99, 20, "b3spline", 4, False),     # per-scale exchange, halos 2,4,8,16
    (2, 520, 16, "triangle", 8, True),     # four-scale passes of the 3-tap family: (0,4,15), (4,4,240)
    (8, 1024, 12, "b3spline", 6, True),    # BASELINE config 4's world: 8 strips of 128 rows, halos 14 / 112
])
def test_strips_match_unsharded_bitwise(tmp_path, world, H, W, family, level, fused):
    # stdlib multiprocessing (spawn): the pytest process itself never imports torch, so a GPU
    # test session is not exposed to torch's bundled ROCm runtime (see bench.py / INTEGRATION.md)
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, family, level, fused,
                                                str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    for r in range(world):
        assert open(tmp_path / f"r{r}.txt").read() == "1 1", f"rank {r} mismatch"


def test_partition_rows():
    from wavelets_amd.parallel import partition_rows
    assert partition_rows(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert partition_rows(32768, 8)[-1] == (28672, 4096)
    for H, n in ((8192, 8), (1000, 7), (5, 5)):
        parts = partition_rows(H, n)
        assert parts[0][0] == 0 and sum(c for _, c in parts) == H
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(n - 1))
    with pytest.raises(ValueError):
        partition_rows(3, 4)


# ---------------------------------------------------------------------------------------------
# sharded utils.wow: per-scale halos of every dilated operator's own input
# plane + all-reduced global scalars, modelled with gloo and the numpy oracle
# ---------------------------------------------------------------------------------------------
def _extended(dist, torch, rank, world, own, halo):
    """rows [-halo, n + halo) of the global plane around this strip: the neighbours' rows, or the
    symmetric reflection at the global top / bottom border"""
    up, dn = _exchange(dist, torch, rank, world, own, halo)
    if up is None:
        up = own[:halo][::-1]
    if dn is None:
        dn = own[-halo:][::-1]
    return np.concatenate([up, own, dn])


def _wow_worker(rank, world, port, H, W, n_scales, bilateral, result_dir):
    import torch
    import torch.distributed as dist
    from oracle import atrous_numpy as O
    from wavelets_amd.parallel import partition_rows

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    fam, hw = "b3spline", 2
    rng = np.random.default_rng(11)
    img = (rng.standard_normal((H, W)) + 2 * np.sin(np.arange(W) / 9.)[None, :]).astype(np.float32)
    row0, n = partition_rows(H, world)[rank]
    assert hw << (n_scales - 1) <= min(c for _, c in partition_rows(H, world))
    kern = O.kernel_2d(fam, np.float32)
    sig_e = O.sigma_e(fam, bilateral, 2)
    dcoef = [5, 2] + [0] * (n_scales - 2) + [1]

    # transform (watroo/wavelets.py:429-442): one exchange of hw * 2**s rows of c_s per scale
    planes = np.empty((n_scales + 1, n, W), np.float32)
    cur = img[row0:row0 + n].copy()
    for s in range(n_scales):
        halo = hw << s
        ext = _extended(dist, torch, rank, world, cur, halo)
        if bilateral is None:
            nxt = O.convolution(ext, fam, s)
        else:                                       # variance and range weights from the same halo
            var = O.sdev_loc(ext, fam, s, variance=True) * bilateral ** 2
            nxt = O.atrous_convolution(ext, kern, var, s)
        nxt = nxt[halo:halo + n]
        planes[s] = cur - nxt
        cur = nxt
    planes[n_scales] = cur

    # global MAD noise (wavelets.py:126-127): the GPU all-reduces radix-select histograms; here
    # the ranks gather |w_0| (exactness of the select itself is a GPU test)
    parts = [None] * world
    dist.all_gather_object(parts, planes[0])
    noise = np.median(np.abs(np.concatenate(parts))) / 0.6745 / sig_e[0]

    for s in range(n_scales):                       # utils.py:174-203
        halo = hw << s
        ext = _extended(dist, torch, rank, world, planes[s], halo)        # halo of the plane itself
        lp = O.convolution(ext ** 2, fam, s)[halo:halo + n]
        lp[lp <= 0] = 1e-15
        np.sqrt(lp, out=lp)
        if dcoef[s] != 0:
            r = np.abs(planes[s] / (dcoef[s] * noise * sig_e[s]))
            from scipy import special
            planes[s] *= special.erf(r)
        planes[s] *= 1 / lp
    c = planes[n_scales].astype(np.float64)         # np.std -> all-reduced (sum, sum of squares)
    mom = torch.tensor([c.sum(), (c * c).sum()], dtype=torch.float64)
    dist.all_reduce(mom)
    npix = float(H) * W
    std = np.float32(np.sqrt(max(mom[1].item() / npix - (mom[0].item() / npix) ** 2, 0.0)))
    planes[n_scales] *= 1 / std
    recon = np.sum(planes, axis=0)

    ref_r, ref_c = O.wow(img.copy(), fam, n_scales=n_scales, denoise_coefficients=[5, 2],
                         bilateral=bilateral)
    assert len(ref_c) == n_scales + 1              # utils.py:122 does not cap it at this size
    tol = 2e-5 * max(1.0, float(np.abs(ref_r).max()))
    ok = (np.abs(recon - ref_r[row0:row0 + n]).max() <= tol
          and np.abs(planes - ref_c.data[:, row0:row0 + n]).max() <= 2e-5 * np.abs(ref_c.data).max()
          and abs(noise - ref_c.noise) <= 1e-6 * ref_c.noise)
    with open(os.path.join(result_dir, f"w{rank}.txt"), "w") as f:
        f.write(f"{int(ok)}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,n_scales,bilateral", [(2, 96, 64, 4, None), (3, 96, 60, 4, 1),
                                                          (2, 128, 120, 5, 1)])
def test_sharded_wow_model_matches_unsharded_oracle(tmp_path, world, H, W, n_scales, bilateral):
    """What StripTransform.wow exchanges and all-reduces is enough: strips + halos of each
    operator's own input plane + all-reduced moments reproduce the unsharded wow (to the fp32
    rounding of np.std's pairwise sum vs fp64 moments)."""
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_wow_worker, args=(r, world, port, H, W, n_scales, bilateral,
                                                    str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"worker exited with {p.exitcode}"
    for r in range(world):
        assert open(tmp_path / f"w{r}.txt").read() == "1", f"rank {r} mismatch"
