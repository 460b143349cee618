#!/usr/bin/env python3
"""Real multi-rank check of the RCCL strip path on a box with ONE GPU.

    python tools/check_rccl_ranks.py --ranks 2 [--shape 1536 1100]

starts the ranks itself (wavelets_amd/launch.py: child processes, a local socket for the unique id,
a wall-clock limit; no rank imports torch - the same plumbing and the same system ROCm stack as
bench.py --gpus N).  It also runs as a rank of somebody else's launcher (RANK / WORLD_SIZE in the
environment, e.g. `python -m torch.distributed.run --nproc-per-node 2 ... tools/check_rccl_ranks.py`),
and --launcher torch uses torch.distributed (gloo) between the ranks as rounds 1-3 did.

RCCL refuses two ranks of one communicator on the same device ("Duplicate GPU detected"), so
every rank gets its own NCCL_HOSTID: RCCL then believes the ranks live on different hosts and
moves the halo rows through its socket transport over `lo`.  The transport is slower than xGMI;
the CALL SEQUENCE (unique id broadcast, ncclCommInitRank, grouped ncclSend/ncclRecv on the compute
stream before every pass, all-reduced histograms / moments) is exactly the production one, which
is what this script verifies: every rank compares its strip of

  * the 7 planes and the reconstruction of wt_decompose_sum (fused passes, B3spline L = 6),
  * the planes of the per-scale (unfused) schedule and of Triangle L = 8,
  * the global MAD noise (all-reduced radix-select histograms), denoise([5,3,2]) + plane sum,
  * wt_reduce's {sum, sum^2, min, max},
  * StripTransform.wow (plain and bilateral=1, denoise_coefficients=[5,2], 5 scales: halos of every
    scale's own plane, all-reduced median / moments) and StripTransform.denoise_sum

BIT FOR BIT with an unsharded plan computed on the same GPU by the same rank.
Where the devices differ (a real multi-GPU node) pass --own-gpu: rank r uses device LOCAL_RANK.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import faulthandler
    faulthandler.enable(all_threads=True)          # a rank that dies on a signal leaves its Python stack on stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=2, default=[1536, 1100])
    ap.add_argument("--own-gpu", action="store_true")
    ap.add_argument("--ranks", type=int, default=0, help="start this many ranks (not needed under a launcher)")
    ap.add_argument("--launcher", choices=("stdlib", "torch"), default="stdlib")
    ap.add_argument("--time-limit", type=float, default=400.0)
    args = ap.parse_args()
    if "RANK" not in os.environ:
        if args.ranks < 1:
            sys.exit("not running as a rank: pass --ranks N")
        import subprocess
        from wavelets_amd import launch
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        rc = subprocess.run([sys.executable, "-c", "import __graft_entry__ as e; e.build()"], cwd=ROOT, env=env).returncode
        if rc:
            sys.exit(rc)
        if args.launcher == "torch":
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port = s_.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.ranks}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.run(cmd, env=env, timeout=args.time_limit).returncode)
        rc, reason = launch.spawn(args.ranks, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  time_limit=args.time_limit, env=env)
        if rc:
            print(f"check_rccl_ranks: FAILED ({reason})")
        sys.exit(rc)
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not args.own_gpu:
        os.environ["NCCL_HOSTID"] = f"wt-virtual-host-{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")

    import numpy as np
    if args.launcher == "torch":
        import torch  # noqa: F401  (first: one ROCm runtime per process, see bench.py)
        sys.path.insert(0, ROOT)
        from bench import TorchGroup
        group, dog = TorchGroup(rank, world), None
    else:
        from wavelets_amd.launch import SocketGroup, Watchdog
        dog = Watchdog(args.time_limit, lambda: sys.stderr.write(f"[rank {rank}] time limit exceeded\n"))
        group = SocketGroup(rank, world, op_timeout=args.time_limit)
    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    group.barrier()
    from wavelets_amd import _lib as L
    from wavelets_amd.parallel import init_comm, StripTransform
    from wavelets_amd.wavelets import B3spline, Triangle

    dev = local_rank if args.own_gpu else 0
    os.environ["WATROO_HIP_DEVICE"] = str(dev)     # the unsharded reference (default context) on the same GPU
    ctx = L.Context(dev)

    init_comm(ctx, rank, world, group.bcast)
    assert ctx.comm_selftest(1 << 18), "ring send/recv + all-reduce self-test failed"

    H, W = args.shape
    img = np.random.default_rng(5).standard_normal((H, W), dtype=np.float32)
    img[H // 3, W // 2] = 40.0
    checks = []

    def same(name, got, exp):
        ok = np.array_equal(got, exp)
        checks.append((name, bool(ok)))
        if not ok:
            d = np.abs(np.asarray(got, np.float64) - np.asarray(exp, np.float64))
            print(f"[rank {rank}] MISMATCH {name}: max |diff| {d.max():.3e} at "
                  f"{np.unravel_index(d.argmax(), d.shape)}", file=sys.stderr, flush=True)

    for fam_cls, fam, level, fused in ((B3spline, L.B3SPLINE, 6, True),
                                       (Triangle, L.TRIANGLE, 8, True),
                                       (B3spline, L.B3SPLINE, 4, False)):
        flags = L.FLAG_FUSED if fused else 0
        tag = f"{fam_cls.__name__}/L{level}/{'fused' if fused else 'unfused'}"
        whole = L.Plan(ctx, H, W, fam, level)
        whole.upload(L.PLANE_INPUT, img)
        whole.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, flags)

        st = StripTransform(ctx, H, W, level, fam_cls, fused=fused)
        r0, n = st.row0, st.nrows
        st.upload(img[r0:r0 + n])
        recon = st.decompose_sum()
        for s in range(level + 1):
            same(f"{tag} plane {s}", st.plane(s), whole.download(s)[r0:r0 + n])
        same(f"{tag} reconstruction", recon, whole.download(L.PLANE_OUT)[r0:r0 + n])

        # two-call form
        st.decompose()
        for s in range(level + 1):
            same(f"{tag} decompose plane {s}", st.plane(s), whole.download(s)[r0:r0 + n])

        # global scalars
        noise_whole = whole.abs_median(0) / 0.6745 / st.sigma_e[0]
        same(f"{tag} noise", np.float64(st.get_noise()), np.float64(noise_whole))
        ra, rb = np.array(st.plan.reduce(level)), np.array(whole.reduce(level))
        same(f"{tag} reduce min/max", ra[2:], rb[2:])
        # fp64 partial sums are folded per rank and then all-reduced: same value up to fp64 rounding
        checks.append((f"{tag} reduce sums", bool(np.allclose(ra[:2], rb[:2], rtol=1e-12, atol=1e-9))))

        # denoise + sum
        sig = [5, 3, 2][:level]
        st.denoise(sig)
        for scl, sg in enumerate(sig):
            whole.denoise(scl, sg * noise_whole * st.sigma_e[scl], 1, True, L.PLANE_NONE)
        whole.plane_sum(0, level + 1, L.PLANE_OUT)
        same(f"{tag} denoised sum", st.sum(), whole.download(L.PLANE_OUT)[r0:r0 + n])
        st.plan.close()
        whole.close()

    # ---- sharded wow / denoise_sum against the unsharded public API on the same GPU
    import wavelets_amd as WA
    n_scales = 5
    if (2 << (n_scales - 1)) <= H // world:
        simg = (img + 2 * np.sin(np.arange(W, dtype=np.float32) / 9.)[None, :]).astype(np.float32)
        for kw in (dict(denoise_coefficients=[5, 2]), dict(bilateral=1, denoise_coefficients=[5, 2]),
                   dict(h=.5, gamma=2, denoise_coefficients=[5, 2], preserve_variance=True)):
            tag = "wow " + ",".join(f"{k}={v}" for k, v in kw.items())
            ref_img, ref_c = WA.wow(simg.copy(), n_scales=n_scales,
                                    **{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
            st = StripTransform(ctx, H, W, n_scales, B3spline)
            r0, n = st.row0, st.nrows
            st.upload(simg[r0:r0 + n])
            got = st.wow(**{k: (list(v) if isinstance(v, list) else v) for k, v in kw.items()})
            # detail planes: no global moment enters unless preserve_variance -> bit for bit; the
            # last plane and the image carry std / rms / min / max from fp64 partial sums that are
            # folded per rank before the all-reduce: equal to the fp32 rounding of those scalars
            exact = "preserve_variance" not in kw
            for s in range(n_scales):
                if exact:
                    same(f"{tag} plane {s}", st.plane(s), ref_c.data[s][r0:r0 + n])
                else:
                    checks.append((f"{tag} plane {s}", bool(np.allclose(st.plane(s), ref_c.data[s][r0:r0 + n], rtol=2e-6, atol=0))))
            checks.append((f"{tag} smooth plane", bool(np.allclose(st.plane(n_scales), ref_c.data[n_scales][r0:r0 + n], rtol=2e-6, atol=0))))
            checks.append((f"{tag} image", bool(np.allclose(got, ref_img[r0:r0 + n], rtol=2e-6, atol=2e-6 * float(np.abs(ref_img).max())))))
            same(f"{tag} noise", np.float64(st.noise), np.float64(ref_c.noise))
            st.plan.close()
        st = StripTransform(ctx, H, W, 4, Triangle)
        r0, n = st.row0, st.nrows
        st.upload(simg[r0:r0 + n])
        st.decompose()
        c = WA.AtrousTransform(Triangle)(simg, 4)
        c.denoise([4, 2], weights=[.5, 2])
        same("denoise_sum", st.denoise_sum([4, 2], weights=[.5, 2]), np.sum(c, axis=0)[r0:r0 + n])
        for s in range(5):
            same(f"denoise_sum plane {s}", st.plane(s), c.data[s][r0:r0 + n])
        st.plan.close()

    # placement of the strip planes chosen by measurement (StripTransform(planes="auto"), collective): a shape whose
    # planes exceed 8 MiB, both placements timed and compared through the all-reduced moments; whatever wins, the
    # result must be the unsharded one bit for bit, and every rank must have reached the same decision
    from wavelets_amd import parallel as PAR
    Hb, Wb = 2048 * world, 2304
    big = np.random.default_rng(9).standard_normal((Hb, Wb), dtype=np.float32)
    st = StripTransform(ctx, Hb, Wb, 6, B3spline, planes="auto")
    meas = PAR._STRIP_PLANES.get((Hb, Wb, 6, L.B3SPLINE, world, True, "measured"))
    checks.append(("auto placement measured both", bool(meas) and meas.get("hipmalloc") is not None))
    votes = group.gather(st.planes)
    checks.append(("auto placement agreed by every rank", rank != 0 or len(set(votes)) == 1))
    r0, n = st.row0, st.nrows
    st.upload(big[r0:r0 + n])
    recon = st.decompose_sum()
    whole = L.Plan(ctx, Hb, Wb, L.B3SPLINE, 6)
    whole.upload(L.PLANE_INPUT, big)
    whole.decompose_sum(L.PLANE_INPUT, 6, L.PLANE_OUT, L.FLAG_FUSED)
    same(f"auto placement ({st.planes}) reconstruction", recon, whole.download(L.PLANE_OUT)[r0:r0 + n])
    st.plan.close()
    for forced in ("scattered", "hipmalloc"):
        st = StripTransform(ctx, Hb, Wb, 6, B3spline, planes=forced)
        st.upload(big[r0:r0 + n])
        same(f"{forced} strip planes reconstruction", st.decompose_sum(), whole.download(L.PLANE_OUT)[r0:r0 + n])
        mem = st.plan.memory()
        checks.append((f"{forced} strip planes placement", (mem[1] > 0) == (forced == "scattered")))
        st.plan.close()
    whole.close()
    if rank == 0:
        print(f"check_rccl_ranks: strip planes chosen by measurement: {votes[0]} {meas}", flush=True)

    # a self-test between a transform whose first pass histogrammed |w_0| and the median that would
    # start from those bins: the self-test zeroes words of the same buffer, so the marker must go
    # (constant image: every |w_0| is exactly 0 and lives in the bins the self-test wipes)
    st = StripTransform(ctx, H, W, 3, B3spline)
    st.upload(np.zeros((st.nrows, W), np.float32))
    st.plan.decompose(L.PLANE_INPUT, 3, L.FLAG_FUSED | L.FLAG_MEDIAN_HIST)
    assert ctx.comm_selftest(4096)
    same("median after a self-test (constant image)", np.float64(st.get_noise()), np.float64(0.0))
    st.plan.close()

    ctx.sync()
    bad = [n for n, ok in checks if not ok]
    nbad = int(group.allreduce(len(bad), sum))
    if rank == 0:
        print(f"check_rccl_ranks: {world} ranks ({args.launcher} plumbing, torch "
              f"{'imported' if 'torch' in sys.modules else 'not imported'}), image {H}x{W}, {len(checks)} comparisons per "
              f"rank, {nbad} mismatches in total", flush=True)
    group.barrier()
    group.close()
    if dog is not None:
        dog.cancel()
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
