#!/usr/bin/env python3
"""Headline benchmark: Mpix/s of a-trous decompose + plane sum (B3spline, 6 scales, float32).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 : the BASELINE.json headline workload - 8192 x 8192 float32, B3spline, 6 scales,
        decompose (7 planes materialised in HBM) followed by the plane sum; input already
        resident in HBM when the timed region starts.
N > 1 : launched by torch.distributed.run, one rank per GPU.  BASELINE config 4: one
        32768 x 32768 image split into N row strips, halo rows exchanged with the strip
        neighbours over RCCL (ncclSend/ncclRecv on the compute stream) before every pass.
        torch.distributed (gloo) is only the launcher plumbing: rendezvous, broadcast of the
        RCCL unique id, barrier and the MAX over ranks of the timed region.

One step = one pass of the hot path over the image.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LEVEL = 6
FAMILY = "b3spline"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured copy


def algorithmic_bytes_per_pixel(kernel, level=LEVEL):
    """Compulsory HBM bytes per pixel attributed to one launch of `kernel`.

    SURVEY 8(d): decompose = 4*(L+2) (read the input, write L+1 planes), sum = 4*(L+2) (read
    L+1 planes, write one); 64 B/pixel for L = 6.  A fused pass over NS scales is charged its
    share: 4*NS for the detail planes it writes, +4 for the input (first pass) or the smooth
    plane (last pass); the accumulate variants additionally the sum's share of the planes they
    fold in: 4*NS, +8 in the last pass (smooth plane read, reconstruction written).  The
    intermediate smooth plane between two passes is NOT algorithmic (it is the price of the
    two-pass structure).  Shares add up to 32 (decompose) and 64 (decompose + sum) for L = 6."""
    if kernel.startswith("wt_plane_sum"):
        return 4.0 * (level + 2)                 # read level+1 planes, write one
    if kernel.startswith("wt_chain_kernel<decomp>"):
        return 8.0                               # write w_s; read c_0 / write c_L once overall
    if kernel.startswith("wt_fused"):
        tag = kernel[kernel.index("<") + 1:-1]                    # e.g. d8x3
        d, ns = int(tag[1:tag.index("x")]), int(tag.split("x")[1])
        first = d == 1
        last = {1: 0, 8: 3, 64: 6}[d] + ns == level
        b = 4.0 * ns + (4.0 if first else 0.0) + (4.0 if last else 0.0)
        if kernel.startswith("wt_fused_acc") or kernel.startswith("wt_fused_sum"):
            b += 4.0 * ns + (8.0 if last else 0.0)
        return b
    return None


def make_strip(nrows, W, seed):
    """Synthetic N(0,1) float32 strip; tall strips repeat a 2048-row block (generation time)."""
    rng = np.random.default_rng(seed)
    block = rng.standard_normal((min(nrows, 2048), W), dtype=np.float32)
    if nrows <= 2048:
        return block
    reps = (nrows + 2047) // 2048
    return np.ascontiguousarray(np.tile(block, (reps, 1))[:nrows])


def cpu_baseline():
    """The C/OpenMP oracle (a port of the reference algorithm, oracle/atrous_ref.c) timed on
    this host on a bounded sample of the same workload."""
    from oracle import cref
    cref.build()
    cores = cref.usable_cpus()
    threads = min(cores, cref.num_threads())
    cref.set_threads(threads)
    probe = np.random.default_rng(0).standard_normal((1024, 1024), dtype=np.float32)
    cref.plane_sum(cref.decompose(probe, LEVEL, FAMILY))      # spawn the OpenMP team
    t = time.perf_counter()
    cref.plane_sum(cref.decompose(probe, LEVEL, FAMILY))
    per_pix = (time.perf_counter() - t) / probe.size
    side = 8192
    while side > 1024 and per_pix * side * side > 15.0:
        side //= 2
    img = np.random.default_rng(0).standard_normal((side, side), dtype=np.float32)
    reps, t_tot = 0, 0.0
    while reps < 5 and t_tot < 10.0:
        t = time.perf_counter()
        cref.plane_sum(cref.decompose(img, LEVEL, FAMILY))
        t_tot += time.perf_counter() - t
        reps += 1
    return {"value": round(img.size * reps / t_tot / 1e6, 2), "unit": "Mpix/s",
            "cores": threads, "kind": "port",
            "sample": f"{reps} x decompose+sum of {side}x{side} f32 {FAMILY} L={LEVEL} "
                      f"(oracle/atrous_ref.c, gcc -O3 -fopenmp, {threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup", type=float, default=0.25,
                    help="seconds of untimed steps before the warm-up (clock ramp)")
    ap.add_argument("--size", type=int, default=0, help="override image side (testing)")
    ap.add_argument("--rows", type=int, default=0, help="override image height (strip-shaped tests)")
    ap.add_argument("--unfused", action="store_true", help="one kernel per scale")
    ap.add_argument("--two-call", action="store_true",
                    help="decompose and plane sum as two calls (the sum re-reads the planes) "
                         "instead of wt_decompose_sum")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--brief", action="store_true", help="one short line (tuning sweeps)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the launcher plumbing (gloo rendezvous, RCCL communicator) even "
                         "with one rank (plumbing check on a 1-GPU box)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="testing on a 1-GPU box: all ranks use device 0 and each gets its own "
                         "NCCL_HOSTID, so RCCL treats them as separate hosts (socket transport)")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line (rank 0).  Native libraries print banners on fd 1
    # (gloo: "[Gloo] Rank 0 is connected ...", RCCL: ROCm version / hostname / library path), so
    # fd 1 points to stderr for the whole run and the result goes out through the saved fd.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(result_fd, (line + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.shared_gpu:
        local_rank = 0
        os.environ["NCCL_HOSTID"] = f"wt-virtual-host-{rank}"
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world

    # ORDER MATTERS: torch wheels bundle their own ROCm runtime (libamdhip64 / libhsa-runtime64 /
    # librccl).  If libwatroo_hip.so pulls in the system ROCm first and torch is imported
    # afterwards, the process ends up with TWO HSA runtimes and whichever initialises second
    # sees "no ROCm-capable device".  Importing torch first makes the dynamic loader resolve
    # our library's sonames to the already-loaded (torch) copies: one consistent stack.
    dist = None
    if world > 1 or args.force_dist:
        import torch  # noqa: F401  (before anything loads libwatroo_hip.so)
        import torch.distributed as dist
    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if dist is not None:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    from wavelets_amd import _lib
    from wavelets_amd._lib import PLANE_INPUT, PLANE_OUT

    ctx = _lib.Context(local_rank)
    if dist is not None:
        def bcast(obj, src):
            box = [obj]
            dist.broadcast_object_list(box, src=src)
            return box[0]
        from wavelets_amd.parallel import init_comm
        init_comm(ctx, rank, world, bcast)
        if args.force_dist:
            assert ctx.comm_selftest(1 << 20), "RCCL self-test failed"

    side = args.size or (8192 if world == 1 else 32768)
    H = W = side
    if args.rows:
        H = args.rows
    nrows = H // world
    row0 = rank * nrows
    if rank == world - 1:
        nrows = H - row0
    fam = _lib.B3SPLINE
    plan = _lib.Plan(ctx, H, W, fam, LEVEL, row0=row0, nrows=nrows, rank=rank, nranks=world)
    plan.upload(PLANE_INPUT, make_strip(nrows, W, seed=rank))
    flags = 0 if args.unfused else _lib.FLAG_FUSED

    two_call = args.two_call or args.unfused

    def step():
        if two_call:
            plan.decompose(PLANE_INPUT, LEVEL, flags)
            plan.plane_sum(0, LEVEL + 1, PLANE_OUT)
        else:       # same outputs (7 planes + reconstruction, bit-identical), sum carried along
            plan.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, flags)

    def fence():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        ctx.sync()

    # Untimed spin-up: the GPU idles at 94 MHz and its clocks ramp over the first milliseconds of
    # work; W = 3 warm-up steps are only 2.5 ms.  Run the same step for a quarter of a second so
    # that the timed region starts at steady clocks (measured: K = 5 reads 4 % low otherwise).
    if dist is None:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup:
            for _ in range(20):
                step()
            ctx.sync()
    elif args.spinup > 0:
        # every rank must issue the same number of halo exchanges: a fixed count, not a clock
        for _ in range(60):
            step()
        ctx.sync()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    dev_ms = ctx.timer_stop()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    ms_per_step = elapsed / args.steps * 1e3
    value = H * W * args.steps / elapsed / 1e6

    # ---- live per-kernel timing (HIP events on the launch stream) for the roofline figure
    ctx.profile(True)
    ctx.profile_reset()
    nprof = max(3, min(args.steps, 10))
    for _ in range(nprof):
        step()
    prof = ctx.profile_entries()
    ctx.profile(False)
    roofline = None
    kernels = {}
    my_pix = float(nrows) * W
    for name, (calls, ms) in prof.items():
        # a pass may be launched in several parts (multi-GPU: edge rows, then interior rows): its
        # algorithmic bytes are per STEP, whatever the number of launches
        bpp = algorithmic_bytes_per_pixel(name)
        kernels[name] = {"calls_per_step": calls // nprof,
                         "avg_ms": round(ms / calls, 4),
                         "algorithmic_GBs": None if bpp is None else
                         round(bpp * my_pix * nprof / (ms * 1e-3) / 1e9, 1)}
    # Dominant kernel = the SOURCE kernel with the largest total time.  The two fused passes are
    # instantiations of one kernel (wt_fused_kernel<..., D=1> and <..., D=8>; rocprof lists them
    # as two rows) and together take ~2/3 of a step, so the roofline entry describes them: per
    # launch algorithmic bytes / average launch duration over both instantiations.
    groups = {}
    for n, (c, ms) in prof.items():
        bpp = algorithmic_bytes_per_pixel(n)
        if bpp is None:
            continue
        key = "wt_fused_kernel" if n.startswith("wt_fused") else n.split("<")[0]
        f = groups.setdefault(key, {"ms": 0.0, "calls": 0, "bytes": 0.0, "members": []})
        f["ms"] += ms
        f["calls"] += c
        f["bytes"] += bpp * my_pix * nprof
        f["members"].append(n)
    if groups:
        dom = max(groups, key=lambda k: groups[k]["ms"])
        f = groups[dom]
        achieved = f["bytes"] / (f["ms"] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            vals = [tj.get(f"{m}@{side}") for m in f["members"]]
            if vals and all(v is not None for v in vals):
                traffic = round(sum(vals) / len(vals))             # HBM bytes per launch
        roofline = {"bound": "hbm", "kernel": dom, "instantiations": sorted(f["members"]),
                    "launches_per_step": f["calls"] // nprof,
                    "avg_launch_ms": round(f["ms"] / f["calls"], 4),
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic}

    if rank == 0:
        whole_job_GBs = 8.0 * (LEVEL + 2) * H * W * args.steps / elapsed / 1e9
        out = {
            "metric": "Mpix/s decompose+sum, 8192^2 f32 B3spline 6 scales; %HBM roofline",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{H}x{W} float32 N(0,1), {FAMILY} L={LEVEL}, decompose "
                                   f"({LEVEL + 1} planes in HBM) + plane sum; device-resident"
                                   + ("" if world == 1 else f"; {world} row strips, RCCL halo "
                                      "exchange per pass"),
                       "image": [H, W], "levels": LEVEL, "family": FAMILY,
                       "fused": not args.unfused, "sum_in_passes": not two_call,
                       "schedule": _lib.schedule(fam, LEVEL, not args.unfused),
                       "parallelism": f"strips{world}"},
            "device_ms_per_step": round(dev_ms / args.steps, 4),
            "whole_path": {"algorithmic_GBs": round(whole_job_GBs, 1),
                           "frac_of_hbm_peak": round(whole_job_GBs / (HBM_PEAK_GBS * world), 4),
                           "bytes_per_pixel": 8 * (LEVEL + 2)},
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1:
            # PCIe-inclusive rate (host numpy in, reconstruction out as a numpy array) - never
            # `value`.  First call: the result's page-locked block is allocated; later calls
            # reuse it from the host pool (wavelets_amd/_lib.py _HostPool), which is the steady
            # state of a frame loop.
            img = make_strip(nrows, W, seed=0)
            rates = []
            for _ in range(3):
                t = time.perf_counter()
                plan.upload(PLANE_INPUT, img)
                step()
                recon = plan.download(PLANE_OUT)
                rates.append(H * W / (time.perf_counter() - t) / 1e6)
                del recon
            out["pcie_inclusive_first_call_mpix_s"] = round(rates[0], 1)
            out["pcie_inclusive_mpix_s"] = round(max(rates[1:]), 1)
            if not args.no_cpu and not args.brief:
                out["cpu_baseline"] = cpu_baseline()
        if args.brief:
            emit(f"{out['value']:.0f} Mpix/s  {ms_per_step:.4f} ms  " + "  ".join(
                f"{k}={v['avg_ms']}" for k, v in kernels.items()))
        else:
            emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
