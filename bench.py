#!/usr/bin/env python3
"""Benchmark of the a-trous hot path: Mpix/s, roofline fraction, CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|cfg2|cfg3|cfg5]

--config headline (default; BASELINE.json `metric`)
    N = 1 : 8192 x 8192 float32 N(0,1), B3spline, 6 scales: decompose (7 planes materialised in
            HBM) + plane sum; input already resident in HBM when the timed region starts.
    N > 1 : BASELINE config 4: one 32768 x 32768 image split into N row strips, halo rows
            exchanged with the strip neighbours over RCCL before every pass.  One process per GPU.
            `python bench.py --gpus N` starts its N ranks itself (child processes, before anything
            touches the GPU; wavelets_amd/launch.py) unless it already runs as a rank (RANK /
            WORLD_SIZE in the environment: `python -m torch.distributed.run ... bench.py --gpus N`,
            the driver's command).  Either way the ranks rendezvous over a local socket - the 128-
            byte RCCL unique id, barriers and the MAX over ranks - and NO rank imports torch: N = 1
            and N > 1 run on the same (system) ROCm stack.  --launcher torch restores the round-3
            plumbing (torch.distributed / gloo in every rank).  The line carries per-pass
            exchange / interior / edge times, the step time with the overlap switched off, and an
            A/B of the strip planes' placement (plain hipMalloc vs scattered 2-MiB chunks); a
            wall-clock limit (--time-limit) kills a hung launch and reports it as JSON.
--config cfg2 | cfg3 | cfg5  (N = 1; the other BASELINE.json configs, same JSON shape)
    cfg2  4096^2 B3spline L=6 decompose + reconstruct
    cfg3  8192^2 Triangle L=8 + denoise([5,3,2]) soft threshold + reconstruct
    cfg5  8192^2 wow(bilateral=1, denoise_coefficients=[5,2])  (B3spline, 11 scales)

One step = one pass of the hot path over the image.  Rank 0 prints ONE JSON line.

The default invocation (N = 1, --config headline) additionally times cfg2, cfg3 and cfg5 after the
headline and appends them to the same line as `"configs": {"cfg2": {...}, "cfg3": {...}, "cfg5":
{...}}` (value, ms_per_step, SURVEY 8(d) bytes per pixel and fraction, PMC-based achieved-HBM
fraction, dominant kernel), so that every BASELINE.json configuration is timed by whoever runs the
benchmark; --no-configs skips them.

Two HBM fractions are reported, and they are different things:
  `frac` / `frac_of_hbm_peak`   SURVEY 8(d) ALGORITHMIC bytes (a property of the problem statement)
                                / time / 8 TB/s - a throughput-equivalent;
  `hbm_achieved`                bytes that really crossed the HBM interface (rocprofv3 PMC passes,
                                2 x FETCH_SIZE + WRITE_SIZE, stored per kernel in
                                profiles/traffic.json - NOT re-measured in this run) / time / 8 TB/s.
The fused passes carry the plane sum along and therefore move FEWER bytes than the algorithmic
figure: hbm_achieved < frac.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LEVEL = 6
MULTI_GPU_SIDE = 32768     # BASELINE configs[3]: the image the N > 1 lines split into row strips
FAMILY = "b3spline"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured copy
VALU_PEAK_TFLOPS = 157.3     # FP32 vector peak (MI355X_MICROARCH.md chip table)
XGMI_LINK_GBS = 153.0        # one xGMI link, one direction (7 links per GPU; the task's figure)

CONFIGS = {
    # name: (side, family, level, workload text)
    "headline": (8192, "b3spline", 6, "decompose ({n} planes in HBM) + plane sum"),
    "cfg2": (4096, "b3spline", 6, "decompose ({n} planes in HBM) + plane sum (BASELINE configs[1])"),
    "cfg3": (8192, "triangle", 8, "decompose + MAD noise + denoise([5,3,2]) soft + plane sum "
                                  "(BASELINE configs[2])"),
    "cfg5": (8192, "b3spline", 11, "wow(bilateral=1, denoise_coefficients=[5,2]) (BASELINE configs[4])"),
}


def algorithmic_bytes_per_pixel(kernel, level=LEVEL, interleaved=False, n_fold=3):
    """Compulsory HBM bytes per pixel attributed to one launch of `kernel`.

    SURVEY 8(d): decompose = 4*(L+2) (read the input, write L+1 planes), sum = 4*(L+2) (read
    L+1 planes, write one); 64 B/pixel for L = 6.  A fused pass over NS scales is charged its
    share: 4*NS for the detail planes it writes, +4 for the input (first pass) or the smooth
    plane (last pass); the accumulate variants additionally the sum's share of the planes they
    fold in: 4*NS, +8 in the last pass (smooth plane read, reconstruction written).  The
    intermediate smooth plane between two passes is NOT algorithmic (it is the price of the
    two-pass structure).  Shares add up to 32 (decompose) and 64 (decompose + sum) for L = 6.
    cfg3 (108 B/pixel): the MAD select reads plane 0 once (4; the histogram
    launches share it), denoise([..3 sigmas..]) is 8 per thresholded plane; fused into the sum
    kernel those 24 ride on wt_denoise_sum_kernel next to its share of the sum: all 4*(L+2) when
    it sums every plane, 4 per thresholded plane when the threshold step runs between the fused
    passes (`interleaved`: the accumulate passes behind it are charged the rest of the sum).
    cfg5: per-scale operators - bilateral scale 12 (read c_s, write w_s and c_{s+1}), fused wow
    update 8 (read c, write c)."""
    if kernel.startswith("wt_plane_sum"):
        return 4.0 * (level + 2)                 # read level+1 planes, write one
    if kernel.startswith("wt_denoise_sum"):
        # interleaved: the sum's share is 4 per plane the kernel folds (the planes of the first pass)
        return (4.0 * n_fold if interleaved else 4.0 * (level + 2)) + 8.0 * 3   # + RMW of the three thresholded planes
    if kernel.startswith("wt_hist"):
        # the launches of a select share the one compulsory read (at the bench's sizes two passes over
        # the plane: a windowed first one - riding on the transform's first pass or on its own - and
        # the refinement)
        return 4.0 / 2.0
    if kernel.startswith("wt_signif"):
        return 8.0
    if kernel.startswith("wt_bilateral"):
        return 12.0
    if kernel.startswith(("wt_row_kernel<wow>", "wt_lattice_kernel<wow>", "wt_chain_kernel<wow>",
                          "wt_wow_kernel")):
        return 8.0
    if kernel.startswith(("wt_chain_kernel<decomp>", "wt_row_kernel<decomp>",
                          "wt_lattice_kernel<decomp>")):
        return 8.0                               # write w_s; read c_0 / write c_L once overall
    if kernel.startswith("wt_fused"):
        tag = kernel[kernel.index("<") + 1:-1]                    # e.g. d8x3
        d, ns = int(tag[1:tag.index("x")]), int(tag.split("x")[1])
        first = d == 1
        last = {1: 0, 8: 3, 16: 4, 64: 6}[d] + ns == level
        b = 4.0 * ns + (4.0 if first else 0.0) + (4.0 if last else 0.0)
        if kernel.startswith("wt_fused_acc") or kernel.startswith("wt_fused_sum"):
            b += 4.0 * ns + (8.0 if last else 0.0)
        return b
    return None


def whole_path_bytes_per_pixel(config, level):
    """SURVEY 8(d) per-pixel figure of the whole step."""
    if config == "cfg3":
        return 4.0 * (level + 2) + 4.0 + 24.0 + 4.0 * (level + 2)    # 108 for L = 8
    if config == "cfg5":
        # per scale: bilateral 12 + whitening update 8; MAD read 4; last-plane update 8; sum 4*(n+2)
        return 20.0 * level + 4.0 + 8.0 + 4.0 * (level + 2)
    return 8.0 * (level + 2)


def make_strip(nrows, W, seed):
    """Synthetic N(0,1) float32 strip: np.random.default_rng(seed).standard_normal((nrows, W))
    (SURVEY 8d), generated in row blocks so that the temporary stays small."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = np.empty((nrows, W), dtype=np.float32)
    for r in range(0, nrows, 1024):
        rng.standard_normal(dtype=np.float32, out=out[r:r + 1024])
    return out


def step_spread(samples):
    """median / min / max of the per-step HIP-event samples (ms), SURVEY 8(d)"""
    if not samples:
        return {}
    v = sorted(float(x) for x in samples)
    n = len(v)
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"ms_per_step_median": round(med, 4), "ms_per_step_min": round(v[0], 4), "ms_per_step_max": round(v[-1], 4),
            "ms_per_step_samples": n}


def scaling_text(world, H, W):
    """What the line scales (`scaling_detail`, next to the contract's one-word `scaling`).  The first word is the
    contract's ("weak": per-GPU work fixed, "strong": total work fixed); the rest says which image and how many GPUs,
    because the N = 1 line (the BASELINE metric's 8192^2
    configuration) and the N > 1 lines (ONE 32768^2 image cut into N row strips) are different images: among the
    N > 1 lines the total work is fixed (strong scaling), and the same-image anchor for N = 1 is `n1_same_image`."""
    if world == 1:
        return (f"weak (N=1: the whole {H}x{W} image on one GPU - the BASELINE metric's configuration; the N>1 lines split one "
                f"32768x32768 image, whose single-GPU figure is n1_same_image in this line)")
    return (f"strong (N={world}: one {H}x{W} image as {world} row strips of {H // world} rows, total work fixed for every N>1; "
            f"the N=1 line is the 8192x8192 BASELINE configuration - compare with its n1_same_image for a same-image 1->N ratio)")


def cpu_baseline(config, side, family, level, budget=15.0):
    """The C/OpenMP oracle (a port of the reference algorithm, oracle/atrous_ref.c) timed on
    this host on a bounded sample of the same workload (`budget` seconds of CPU work per run).
    cfg5: the sample image is smaller than 8192^2, and wow() derives its scale count from the
    image size (ref utils.py:122) - the sample therefore runs FEWER scales than the GPU's 11; the
    baseline is normalised per scale (value = Mpix/s at the GPU's scale count, assuming equal cost
    per scale; the sample's own n_scales is stated)."""
    import numpy as np
    from oracle import cref
    cref.build()
    cores = cref.usable_cpus()
    threads = min(cores, cref.num_threads())
    cref.set_threads(threads)

    def run(img):
        if config == "cfg5":
            return cref.wow(img, family, bilateral=1, denoise_coefficients=[5, 2])[0]
        planes = cref.decompose(img, level, family)
        if config == "cfg3":
            noise = float(cref.abs_median(planes[0])) / 0.6745 / cref.sigma_e(family)[0]
            for s, sg in enumerate((5, 3, 2)):
                cref.denoise_plane(planes[s], sg * noise * cref.sigma_e(family)[s])
        return cref.plane_sum(planes)

    probe_side = 256 if config == "cfg5" else 1024
    probe = np.random.default_rng(0).standard_normal((probe_side, probe_side), dtype=np.float32)
    run(probe)                                                       # spawn the OpenMP team
    t = time.perf_counter()
    run(probe)
    per_pix = (time.perf_counter() - t) / probe.size
    s = side
    # the full-size image where ONE run of it stays within ~2 budgets (cfg3 / cfg5 at 8192^2: 6-8 s on 16
    # threads - the same workload as the GPU's, not a quarter of it), else halved until a run fits
    while s > probe_side and per_pix * s * s > 2.2 * budget:
        s //= 2
    img = np.random.default_rng(0).standard_normal((s, s), dtype=np.float32)
    reps, t_tot = 0, 0.0
    while reps < 5 and t_tot < budget * 0.67:
        t = time.perf_counter()
        run(img)
        t_tot += time.perf_counter() - t
        reps += 1
    rate = img.size * reps / t_tot / 1e6
    extra = {}
    what = {"cfg3": f"decompose + MAD + denoise([5,3,2]) + sum, {family} L={level}"}.get(
        config, f"decompose+sum, {family} L={level}")
    if config == "cfg5":
        ns_sample = int(np.round(np.log2(s) - np.log2(5)))           # ref utils.py:122
        what = (f"wow(bilateral=1, denoise_coefficients=[5,2]) at the sample's own n_scales = "
                f"{ns_sample}, rate scaled by {ns_sample}/{level} to the GPU run's {level} scales")
        extra = {"n_scales_sample": ns_sample, "n_scales_gpu": level,
                 "value_at_sample_scales": round(rate, 3)}
        rate *= ns_sample / float(level)
    out = {"value": round(rate, 3), "unit": "Mpix/s",
           "cores": threads, "kind": "port",
           "sample": f"{reps} x {what} of {s}x{s} f32 "
                     f"(oracle/atrous_ref.c, gcc -O3 -fopenmp, {threads} threads)"}
    out.update(extra)
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` outside any launcher: start the N ranks as child processes (never
    exec: nothing here has touched the GPU yet, and nothing will), forward rank 0's one JSON line
    and the return code.  A rank that fails, or the wall-clock limit, ends every rank's process
    group; if no result line has come out by then this process prints {"error": ...} instead."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if args.launcher == "torch":
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + argv
        try:
            return subprocess.run(cmd, env=env, timeout=args.time_limit).returncode
        except subprocess.TimeoutExpired:
            print(json.dumps({"error": f"time limit of {args.time_limit:.0f} s exceeded", "n_gpus": args.gpus}))
            return 124
    from wavelets_amd import launch
    if not args.no_build:
        # build once, before the ranks start (they must not race for the library) and in a child of
        # its own: this process never loads the HIP library
        rc = subprocess.run([sys.executable, "-c", "import __graft_entry__ as e; e.build()"], cwd=ROOT, env=env).returncode
        if rc != 0:
            print(json.dumps({"error": f"__graft_entry__.build() failed with code {rc}", "n_gpus": args.gpus}))
            return rc
    seen = []
    rc, reason = launch.spawn(args.gpus, [sys.executable, os.path.abspath(__file__)] + argv + ["--no-build"],
                              time_limit=args.time_limit, env=env, tee_rank0=seen)
    if rc != 0 and not any(ln.startswith("{") for ln in seen):
        print(json.dumps({"error": reason, "n_gpus": args.gpus, "launcher": "stdlib"}))
    return rc


class TorchGroup:
    """--launcher torch: the collectives of wavelets_amd.launch.SocketGroup over torch.distributed
    (gloo).  Imports torch - and with it torch's bundled ROCm runtime - into the rank."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.rank, self.world = dist, rank, world

    def bcast(self, obj, src=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out if self.rank == 0 else None

    def allreduce(self, value, op=max):
        out = [None] * self.world
        self.dist.all_gather_object(out, value)
        return op(out)

    def barrier(self):
        self.dist.barrier()

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def source_digest():
    """sha1 over the kernel sources (wavelets_amd/csrc + include), comments and whitespace removed:
    what profiles/traffic.json is tied to - PMC bytes per launch describe the kernels they were
    measured on (tools/pmc_summary.py stores the digest, tests/test_abi_cpu.py compares)."""
    import hashlib
    import re
    h = hashlib.sha1()
    files = [os.path.join(ROOT, "include", "watroo_hip.h")]
    csrc = os.path.join(ROOT, "wavelets_amd", "csrc")
    files += [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if os.path.isfile(os.path.join(csrc, f))]
    for f in files:
        text = open(f, encoding="utf-8").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(os.path.basename(f).encode())
        h.update("".join(text.split()).encode())
    return h.hexdigest()


def load_traffic(path=None, digest=None, lib_overridden=None):
    """(db, stale_reason): the PMC traffic table, and why it must not be used (None: usable).  The
    table is stale when it was measured on other kernel sources than the ones the loaded library
    was built from, or when a different library was put in place (WATROO_HIP_LIB)."""
    path = path or os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return {}, "profiles/traffic.json is missing"
    db = json.load(open(path))
    meta = db.get("_meta") or {}
    if lib_overridden is None:
        lib_overridden = bool(os.environ.get("WATROO_HIP_LIB"))
    if lib_overridden:
        return db, "WATROO_HIP_LIB overrides the library the table was measured on"
    if not meta.get("source_digest"):
        return db, "profiles/traffic.json carries no source digest"
    if meta["source_digest"] != (digest or source_digest()):
        return db, ("profiles/traffic.json was measured on other kernel sources (digest "
                    f"{meta['source_digest'][:12]}): re-run tools/profile_round.sh")
    return db, None


def main():
    # a rank that dies on a signal (SIGSEGV inside the runtime stack was seen once on the shared-GPU test rig) leaves
    # its Python stack on stderr - children inherit the launcher's stderr - instead of just a signal number
    import faulthandler
    faulthandler.enable(all_threads=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline")
    ap.add_argument("--spinup", type=float, default=0.25,
                    help="seconds of untimed steps before the warm-up (clock ramp)")
    ap.add_argument("--size", type=int, default=0, help="override image side (testing)")
    ap.add_argument("--rows", type=int, default=0, help="override image height (strip-shaped tests)")
    ap.add_argument("--unfused", action="store_true", help="one kernel per scale")
    ap.add_argument("--two-call", action="store_true",
                    help="decompose and plane sum as two calls (the sum re-reads the planes) "
                         "instead of wt_decompose_sum")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-build", action="store_true",
                    help="do not run __graft_entry__.build() (profiled runs: no child process "
                         "under the profiler's preload)")
    ap.add_argument("--brief", action="store_true", help="one short line (tuning sweeps)")
    ap.add_argument("--no-configs", action="store_true",
                    help="headline only: do not append cfg2 / cfg3 / cfg5 to the line")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the launcher plumbing (gloo rendezvous, RCCL communicator) even "
                         "with one rank (plumbing check on a 1-GPU box)")
    ap.add_argument("--no-exchange", action="store_true",
                    help="testing: skip the halo exchange of multi-rank runs (wrong results at strip "
                         "boundaries - the halo self-check must then report ok = false)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="testing on a 1-GPU box: all ranks use device 0 and each gets its own "
                         "NCCL_HOSTID, so RCCL treats them as separate hosts (socket transport)")
    ap.add_argument("--launcher", choices=("stdlib", "torch"), default="stdlib",
                    help="plumbing between the ranks: local socket (default; no rank imports torch) or "
                         "torch.distributed / gloo as in round 3")
    ap.add_argument("--time-limit", type=float, default=900.0,
                    help="wall-clock limit of a multi-rank run in seconds: at the limit every rank is "
                         "killed and a JSON line says so (the last complete measurement, if there is one)")
    ap.add_argument("--keep-overlap", action="store_true",
                    help="multi-GPU: keep the overlapped order whatever the serial one measures (tests: ranks that "
                         "share one card always measure the serial order faster, and the sweep of overlap_reserve "
                         "behind the overlapped order would never run there)")
    ap.add_argument("--no-scatter-ab", action="store_true",
                    help="multi-rank runs: skip the A/B of the strip planes' placement (plain hipMalloc only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.config != "headline":
            sys.exit("--config cfg2/cfg3/cfg5 are single-GPU workloads")
        sys.exit(self_launch(args, sys.argv[1:]))

    import numpy as np

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
        args.gpus = world
    if world > 1 and args.config != "headline":
        sys.exit("--config cfg2/cfg3/cfg5 are single-GPU workloads")

    # The plumbing between the ranks.  Default: a local socket (wavelets_amd/launch.py) - no rank
    # imports torch, so N = 1 and N > 1 run on the same system ROCm stack; under the driver's
    # `python -m torch.distributed.run` torch lives in the launcher process only.
    # --launcher torch: torch.distributed (gloo) in every rank.  ORDER MATTERS there: torch wheels
    # bundle their own ROCm runtime (libamdhip64 / libhsa-runtime64 / librccl).  If libwatroo_hip.so
    # pulls in the system ROCm first and torch is imported afterwards, the process ends up with TWO
    # HSA runtimes and whichever initialises second sees "no ROCm-capable device"; importing torch
    # first makes the dynamic loader resolve our library's sonames to the already-loaded copies.
    state = {"result": None, "main_done": False}       # what the watchdog reports at the time limit
    dog = None
    if world > 1:
        from wavelets_amd.launch import Watchdog

        def on_expire(what=None):
            what = what or f"time limit of {args.time_limit:.0f} s exceeded"
            sys.stderr.write(f"[bench rank {rank}] {what}\n")
            if rank == 0:
                if state["result"] is not None:
                    state["result"]["time_limit_hit"] = (f"{what}: this is the last complete "
                                                         "measurement; a later phase of the run did not finish")
                    state["result"].pop("_brief", None)
                    emit(json.dumps(state["result"]))
                else:
                    emit(json.dumps({"error": f"{what} before the first complete measurement", "n_gpus": world,
                                     "launcher": args.launcher}))
        # (a little ahead of the launcher's own limit, so that the stored result gets out; a rank that
        #  has finished the main measurement leaves with code 0)
        dog = Watchdog(max(5.0, args.time_limit - 20.0), on_expire,
                       code=lambda: 0 if state["main_done"] else 124)
    group = None
    if world > 1 or args.force_dist:
        if args.launcher == "torch":
            import torch  # noqa: F401  (before anything loads libwatroo_hip.so)
            group = TorchGroup(rank, world)
        else:
            from wavelets_amd.launch import SocketGroup
            group = SocketGroup(rank, world, op_timeout=args.time_limit)
    if rank == 0 and not args.no_build:
        import __graft_entry__ as entry
        entry.build()
    if group is not None:
        group.barrier()
    from wavelets_amd import _lib
    from wavelets_amd._lib import PLANE_INPUT, PLANE_OUT

    if os.environ.get("WT_BENCH_REPORT_MODULES"):     # (tests: what this rank has imported before its first GPU call)
        sys.stderr.write(f"[bench rank {rank}] torch_in_sys_modules={'torch' in sys.modules}\n")
    device = local_rank
    ndev = _lib.device_count()
    if world > 1 and ndev and local_rank >= ndev:
        # fewer visible devices than local ranks: a launcher that pins every rank to its own GPU with
        # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES shows each of them exactly one device
        device = local_rank % ndev
        sys.stderr.write(f"[bench rank {rank}] LOCAL_RANK {local_rank} but {ndev} visible device(s): using device {device}\n")
    ctx = _lib.Context(device)
    rccl_ranks = 1
    if group is not None:
        from wavelets_amd.parallel import init_comm
        init_comm(ctx, rank, world, group.bcast)
        rccl_ranks = ctx.comm_info()[1]          # what ncclCommCount says, not what we asked for
        # who ran where: every rank's GPU (HIP ordinal, PCI bus id) and the RCCL the engine loaded - a run on a
        # shared or mis-bound node (two ranks on one card, a stale library) explains itself in the line
        me = dict(ctx.device_info(), rank=rank, local_rank=local_rank, pid=os.getpid(),
                  visible=os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES"))
        rank_table = group.gather(me)
        rccl_version = _lib.comm_version()
        if args.force_dist:
            assert ctx.comm_selftest(1 << 20), "RCCL self-test failed"

    traffic_db, traffic_stale = load_traffic()                    # from rocprofv3 --pmc passes
    traffic_meta = traffic_db.get("_meta") or {}

    def run_workload(config, steps, warmup, full):
        """Time `steps` steps of BASELINE configuration `config`; returns the JSON object (rank 0)
        or None.  full: PCIe-inclusive rate and CPU baseline too (the line's own configuration)."""
        side, family, level, what = CONFIGS[config]
        if config == "headline" and world > 1:
            side = 32768
        side = args.size or side
        H = W = side
        if args.rows:
            H = args.rows
        if config == "cfg5":                          # wow's own scale count (utils.py:122)
            level = int(np.round(np.log2(min(H, W)) - np.log2(5)))
        nrows = H // world
        row0 = rank * nrows
        if rank == world - 1:
            nrows = H - row0
        fam = {"b3spline": _lib.B3SPLINE, "triangle": _lib.TRIANGLE}[family]
        if config == "cfg5":
            # the cfg5 image of tools/bench_configs.py: noise on a smooth structure (pure noise has
            # no edges for the bilateral weights to act on); the workload string says so
            img0 = (make_strip(nrows, W, seed=0)
                    + 3 * np.sin(np.arange(W, dtype=np.float32) / 50.)[None, :]).astype(np.float32)
        else:
            img0 = make_strip(nrows, W, seed=rank)
        flags = 0 if args.unfused else _lib.FLAG_FUSED
        if args.no_exchange:
            flags |= _lib.FLAG_NO_EXCHANGE
        two_call = args.two_call or args.unfused
        sched = _lib.schedule(fam, level, not args.unfused)
        my_pix = float(nrows) * W

        def make_plan():
            plan = _lib.Plan(ctx, H, W, fam, level, row0=row0, nrows=nrows, rank=rank, nranks=world)
            plan.upload(PLANE_INPUT, img0)
            return plan

        def make_step(plan):
            """(step, coefficients): one pass of the hot path over the image on `plan`"""
            if config in ("headline", "cfg2"):
                def step():
                    if two_call:
                        plan.decompose(PLANE_INPUT, level, flags)
                        plan.plane_sum(0, level + 1, PLANE_OUT)
                    else:   # same outputs (planes + reconstruction, bit-identical), sum carried along
                        plan.decompose_sum(PLANE_INPUT, level, PLANE_OUT, flags)
                return step, None
            import wavelets_amd as WA
            if config == "cfg3":
                coefficients = WA.Coefficients(plan, WA.Triangle(2))
                from wavelets_amd.wavelets import _decompose_denoise_sum
                transform = WA.AtrousTransform(WA.Triangle)

                def step():     # transform, Coefficients.denoise([5,3,2]), np.sum(coefficients, axis=0)
                    coefficients.noise = None                      # lazy MAD estimate, every step
                    if two_call:
                        plan.decompose(PLANE_INPUT, level, flags)
                        coefficients._denoise_sum([5, 3, 2], soft_threshold=True, write_back=True)
                    else:       # the same three results with the threshold step between the passes
                        _decompose_denoise_sum(transform, plan, level, coefficients, [5, 3, 2],
                                               soft_threshold=True, write_back=True)
                return step, coefficients
            from wavelets_amd import utils as WU
            transform = WA.AtrousTransform(WA.B3spline, bilateral=[1] * (level + 1))
            coefficients = WA.Coefficients(plan, WA.B3spline(2), [1] * (level + 1))

            def step():         # utils.wow without its PCIe legs (ref utils.py:148-217)
                transform._run(plan, level)
                coefficients.noise = None
                WU._wow_device(coefficients, level, [], True, [5, 2], True, False, 3.2, None, None, 0)
            return step, coefficients

        def fence():
            ctx.sync()
            if group is not None:
                group.barrier()
            ctx.sync()

        def timed(step, nsteps):
            """nsteps steps between two fences; (seconds - MAX over the ranks -, device ms of this rank)"""
            fence()
            t0 = time.perf_counter()
            ctx.timer_start()
            for _ in range(nsteps):
                step()
            dev_ms = ctx.timer_stop()
            fence()
            elapsed = time.perf_counter() - t0
            if group is not None:
                elapsed = float(group.allreduce(elapsed, max))
            return elapsed, dev_ms

        def measure(step, nsteps, spin=True):
            """spin-up, warm-up, the timed region, then `nprof` more steps under the live per-kernel
            profiler (HIP events on the launch streams)"""
            # Untimed spin-up: the GPU idles at 94 MHz and its clocks ramp over the first milliseconds of
            # work; W = 3 warm-up steps are only 2.5 ms.  Run the same step for a quarter of a second so
            # that the timed region starts at steady clocks (measured: K = 5 reads 4 % low otherwise).
            if group is None and spin:
                t_spin = time.perf_counter()
                while time.perf_counter() - t_spin < args.spinup:
                    for _ in range(20 if config in ("headline", "cfg2") else 2):
                        step()
                    ctx.sync()
            elif spin and args.spinup > 0:
                # every rank must issue the same number of halo exchanges: a fixed count, not a clock
                for _ in range(60):
                    step()
                ctx.sync()
            for _ in range(warmup):
                step()
            elapsed, dev_ms = timed(step, nsteps)
            # the spread of the step, from one HIP-event pair per step on the launch stream (>= 20 samples;
            # the batch above stays the line's `value`: its steps run back to back, these are fenced one by one)
            samples = []
            for _ in range(max(20, min(nsteps, 50))):
                ctx.timer_start()
                step()
                samples.append(ctx.timer_stop())
            ctx.profile(True)
            ctx.profile_reset()
            nprof = max(3, min(nsteps, 10))
            for _ in range(nprof):
                step()
            raw = ctx.profile_entries()
            ctx.profile(False)
            return {"elapsed": elapsed, "dev_ms": dev_ms, "steps": nsteps, "prof_raw": raw, "nprof": nprof, "samples": samples}

        def pmc_bytes(name):
            """HBM bytes per launch of `name` at this image size from profiles/traffic.json (the
            rocprofv3 --pmc passes of an earlier run of the same command on the SAME kernel sources;
            None: not recorded, or the table is stale)."""
            if world > 1 or H != W or traffic_stale:
                return None
            sizes = traffic_meta.get("image") or {}
            if config in sizes and list(sizes[config]) != [H, W]:
                return None
            return traffic_db.get(f"{name}@{config}", traffic_db.get(f"{name}@{side}") if config == "headline" else None)

        def report(m):
            """the JSON object of measurement `m` (every rank computes its per-pass times; rank 0 returns
            the object, the others None)"""
            elapsed, nsteps, nprof = m["elapsed"], m["steps"], m["nprof"]
            ms_per_step = elapsed / nsteps * 1e3
            value = H * W * nsteps / elapsed / 1e6
            # The two parts of a split pass (multi-GPU: "<kernel>/interior", "<kernel>/edge") and the
            # per-pass exchanges ("rccl_halo_exchange/pass<i>") are timed under their own names: merged
            # here into one entry per kernel, kept apart for the per-pass table.
            prof, parts, exch = {}, {}, {}
            for name, (calls, ms) in m["prof_raw"].items():
                base, _, part = name.partition("/")
                if base == "rccl_halo_exchange" and part.startswith("pass"):
                    exch[int(part[4:])] = (calls, ms)
                elif part:
                    parts.setdefault(base, {})[part] = (calls, ms)
                c0, m0 = prof.get(base, (0, 0.0))
                # (a split pass counts once per step: its launches are the parts of ONE pass)
                prof[base] = (c0 + (calls if part in ("", "interior") or base == "rccl_halo_exchange" else 0), m0 + ms)
            for base, pp in parts.items():          # (an edge-only pass - strips thinner than 2 halos - cannot occur: run_schedule)
                if "interior" not in pp:
                    prof[base] = (prof[base][0] + pp["edge"][0], prof[base][1])
            interleaved = any(k.startswith("wt_fused_hist") for k in prof)

            def algo_bytes(name, calls):
                """algorithmic bytes `name` moved during the nprof profiled steps (None: not priced).  A
                fused pass may be launched in several parts (multi-GPU: edge rows, then interior rows):
                its bytes are per STEP; every other kernel's are per launch."""
                bpp = algorithmic_bytes_per_pixel(name, level, interleaved=interleaved,
                                                  n_fold=sched[0][1] if sched else 3)
                if bpp is None:
                    return None
                return bpp * my_pix * (nprof if name.startswith("wt_fused") else calls)

            kernels = {}
            step_traffic, traffic_complete = 0.0, True
            for name, (calls, ms) in prof.items():
                ab = algo_bytes(name, calls)
                tb = pmc_bytes(name)
                kernels[name] = {"calls_per_step": calls // nprof,
                                 "avg_ms": round(ms / max(calls, 1), 4),
                                 "algorithmic_GBs": None if ab is None else round(ab / (ms * 1e-3) / 1e9, 1)}
                if name in parts:
                    kernels[name]["parts_ms"] = {k: round(v[1] / max(v[0], 1), 4) for k, v in parts[name].items()}
                if tb is not None:
                    kernels[name]["hbm_GBs"] = round(tb * calls / (ms * 1e-3) / 1e9, 1)
                    step_traffic += tb * calls / nprof
                elif ms / nprof > 0.002 and not name.startswith("rccl"):   # (select steps etc. move nothing)
                    traffic_complete = False
            # Dominant kernel = the SOURCE kernel with the largest total time.  The fused passes are
            # instantiations of one kernel (wt_fused_kernel<..., D=1> and <..., D=8>; rocprof lists them
            # as two rows), so the roofline entry describes them together: per launch algorithmic bytes /
            # average launch duration over the instantiations.
            groups = {}
            for n, (c, ms) in prof.items():
                ab = algo_bytes(n, c)
                if ab is None:
                    continue
                key = "wt_fused_kernel" if n.startswith("wt_fused") else n.split("<")[0]
                f = groups.setdefault(key, {"ms": 0.0, "calls": 0, "bytes": 0.0, "members": [], "pmc": 0.0,
                                            "pmc_ok": True})
                f["ms"] += ms
                f["calls"] += c
                f["bytes"] += ab
                f["members"].append(n)
                tb = pmc_bytes(n)
                if tb is None:
                    f["pmc_ok"] = False
                else:
                    f["pmc"] += tb * c
            roofline = None
            if groups:
                dom = max(groups, key=lambda k: groups[k]["ms"])
                f = groups[dom]
                achieved = f["bytes"] / (f["ms"] * 1e-3) / 1e9
                have_pmc = f["pmc_ok"] and f["pmc"] > 0
                roofline = {"bound": "hbm", "kernel": dom, "instantiations": sorted(f["members"]),
                            "launches_per_step": f["calls"] // nprof,
                            "avg_launch_ms": round(f["ms"] / f["calls"], 4),
                            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4),
                            "achieved_is": "SURVEY 8(d) algorithmic bytes / measured launch time "
                                           "(throughput-equivalent, not bytes on the HBM interface)",
                            "traffic": round(f["pmc"] / f["calls"]) if have_pmc else None,
                            "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + "
                                              "WRITE_SIZE passes of an earlier run of this command on the "
                                              "same kernel sources; not re-measured here)" if have_pmc else None,
                            "hbm_achieved_GBs": round(f["pmc"] / (f["ms"] * 1e-3) / 1e9, 1) if have_pmc else None,
                            "hbm_achieved": round(f["pmc"] / (f["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                            if have_pmc else None}
                if traffic_stale and world == 1:
                    roofline["traffic_stale"] = True
                    roofline["traffic_stale_reason"] = traffic_stale
                if dom.startswith("wt_bilateral"):
                    # the bilateral operator is VALU / transcendental bound - K*K-1 taps of
                    # (sub, mul, fma into the exponent, v_exp, two fmas) + variance + normalisation per
                    # pixel and scale: ~10 flop per tap + 40.  HBM stays the `bound` the schema knows;
                    # the vector-ALU fraction is reported beside it.
                    taps = 24 if family == "b3spline" else 8
                    flops = (10.0 * taps + 40.0) * my_pix * f["calls"]
                    roofline["valu"] = {"algorithmic_flop_per_pixel_scale": 10.0 * taps + 40.0,
                                        "achieved_TFLOPs": round(flops / (f["ms"] * 1e-3) / 1e12, 1),
                                        "peak_TFLOPs": VALU_PEAK_TFLOPS,
                                        "frac": round(flops / (f["ms"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)}

            # ---- multi-GPU: what every pass of the schedule cost on this rank (live events, both streams)
            strip_passes = None
            if world > 1:
                mine = []
                for i, (s0, ns, halo) in enumerate(sched):
                    tag = f"<d{1 << s0}x{ns}>"
                    base = next((n for n in prof if n.startswith("wt_fused") and n.endswith(tag)), None)
                    e = {"scales": [s0, s0 + ns], "halo_rows": halo, "kernel": base,
                         "exchange_bytes_per_neighbour": halo * W * 4}
                    if i in exch:
                        e["exchange_ms"] = round(exch[i][1] / max(exch[i][0], 1), 4)
                    pp = parts.get(base, {})
                    if "interior" in pp:
                        e["interior_ms"] = round(pp["interior"][1] / max(pp["interior"][0], 1), 4)
                    if "edge" in pp:
                        e["edge_ms"] = round(pp["edge"][1] / max(pp["edge"][0], 1), 4)
                    if base and not pp:
                        e["whole_ms"] = round(prof[base][1] / max(prof[base][0], 1), 4)
                    mine.append(e)
                serial = None
                if "rccl_halo_exchange" in m["prof_raw"]:        # serial order (overlap off): not per pass
                    c_, ms_ = m["prof_raw"]["rccl_halo_exchange"]
                    serial = {"exchanges_per_step": c_ // nprof, "exchange_ms_per_step": round(ms_ / nprof, 4)}
                allp = group.gather(mine)
                if rank == 0:
                    strip_passes = {"rank0": mine, "max_over_ranks": []}
                    if serial:
                        strip_passes["serial_exchanges_rank0"] = serial
                    for i in range(len(sched)):
                        mx = {"scales": mine[i]["scales"]}
                        for key in ("exchange_ms", "interior_ms", "edge_ms", "whole_ms"):
                            vals = [p[i][key] for p in allp if i < len(p) and key in p[i]]
                            if vals:
                                mx[key] = max(vals)
                        strip_passes["max_over_ranks"].append(mx)

            if rank != 0:
                return None
            bpp_whole = whole_path_bytes_per_pixel(config, level)
            whole_job_GBs = bpp_whole * H * W * nsteps / elapsed / 1e9
            metric = {"headline": f"Mpix/s decompose+sum, {H}x{W} f32 B3spline 6 scales; %HBM roofline"
                                  if (world > 1 or side != 8192) else
                                  "Mpix/s decompose+sum, 8192^2 f32 B3spline 6 scales; %HBM roofline",
                      "cfg2": "Mpix/s decompose+sum, 4096^2 f32 B3spline 6 scales; %HBM roofline",
                      "cfg3": "Mpix/s decompose+denoise([5,3,2])+sum, 8192^2 f32 Triangle 8 scales; %HBM roofline",
                      "cfg5": "Mpix/s wow(bilateral=1, denoise_coefficients=[5,2]), 8192^2 f32"}[config]
            whole = {"algorithmic_GBs": round(whole_job_GBs, 1),
                     "frac_of_hbm_peak": round(whole_job_GBs / (HBM_PEAK_GBS * world), 4),
                     "bytes_per_pixel": bpp_whole}
            if traffic_complete and step_traffic > 0:
                whole["hbm_bytes_per_pixel"] = round(step_traffic / my_pix, 2)
                whole["hbm_achieved_GBs"] = round(step_traffic / (ms_per_step * 1e-3) / 1e9, 1)
                whole["hbm_achieved"] = round(step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                whole["hbm_traffic_source"] = "profiles/traffic.json (PMC passes, not re-measured here)"
            elif traffic_stale and world == 1:
                whole["hbm_achieved"] = None
                whole["traffic_stale"] = True
            data = ("np.random.default_rng(0).standard_normal + 3*sin(x/50) along the columns (noise on a smooth "
                    "structure: the bilateral weights need edges)" if config == "cfg5"
                    else "np.random.default_rng(seed).standard_normal")
            out = {
                "metric": metric,
                "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": nsteps,
                "warmup": warmup, "ms_per_step": round(ms_per_step, 4),
                **step_spread(m.get("samples")),
                "higher_is_better": True, "scaling": "weak" if world == 1 else "strong",      # (the contract's two words)
                "scaling_detail": scaling_text(world, H, W),
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{H}x{W} float32 {data}, "
                                       f"{family} L={level}, " + what.format(n=level + 1)
                                       + "; device-resident"
                                       + ("" if world == 1 else f"; {world} row strips, RCCL halo "
                                          "exchange per pass"),
                           "name": config, "image": [H, W], "levels": level, "family": family,
                           "fused": not args.unfused, "sum_in_passes": not two_call,
                           "schedule": sched,
                           "parallelism": f"strips{world}"},
                "rccl_ranks": rccl_ranks,
                "device_ms_per_step": round(m["dev_ms"] / nsteps, 4),
                "whole_path": whole,
                "roofline": roofline,
                "kernels": kernels,
            }
            if world > 1:
                out["launcher"] = {"plumbing": args.launcher,
                                   "torch_imported_in_ranks": "torch" in sys.modules,
                                   "started_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                                   else ("bench.py (wavelets_amd.launch.spawn)" if "WT_RDZV" in os.environ else "other")}
                out["strip_passes"] = strip_passes
                out["ranks"] = rank_table
                out["distinct_gpus"] = len({(r_.get("pci"), r_.get("device")) for r_ in rank_table or []})
                out["rccl_version"] = rccl_version
                out["nccl_env"] = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))}
            out["_brief"] = f"{out['value']:.0f} Mpix/s  {ms_per_step:.4f} ms  " + "  ".join(
                f"{k}={v['avg_ms']}" for k, v in kernels.items())
            return out

        def halo_check(plan):
            """Halo self-check over the real transport (outside the timed region, every rank): a linear
            ramp in y is reproduced exactly by the smoothing filters (symmetric taps, unit gain) away
            from the image's own top / bottom border, so every detail plane must vanish there - unless
            a strip boundary was filtered with missing, stale or misplaced neighbour rows.  Rows within
            the transform's reach of a global border (reflection is not linear) are cropped off.
            Returns the JSON entry (the same on every rank)."""
            worst, check_err = 0.0, None
            try:                         # (a failing check must not cost the timing line: every rank
                reach = sum(h for _, _, h in sched)          #  still reaches the all-reduce below)
                ramp = (np.arange(row0, row0 + nrows, dtype=np.float32)[:, None]
                        * np.ones((1, W), dtype=np.float32))
                plan.upload(PLANE_INPUT, ramp)
                del ramp
                plan.decompose_sum(PLANE_INPUT, level, PLANE_OUT, flags)
                top = reach if rank == 0 else 0
                bot = reach if rank == world - 1 else 0
                if nrows - top - bot > 0:
                    sub = _lib.Plan(ctx, nrows - top - bot, W, fam, 0)
                    for s_ in range(level):
                        sub.crop_from(plan, s_, PLANE_INPUT, top, 0)
                        _, _, lo, hi = sub.reduce(PLANE_INPUT)
                        worst = max(worst, abs(lo), abs(hi))
                    sub.close()
            except Exception as e:
                worst, check_err = 1e30, repr(e)
            worst = float(group.allreduce(worst, max))
            # (float32 rounding of values up to H: a few ulp of 3e4; a wrong halo row shows as O(1..H))
            entry = {"input": "f(y, x) = y", "max_abs_detail_off_the_global_border": worst,
                     "bound": 0.05, "ok": bool(worst <= 0.05)}
            if check_err:
                entry[f"error_on_rank_{rank}"] = check_err
            return entry

        plan = make_plan()
        step, coefficients = make_step(plan)
        m = measure(step, steps)
        out = report(m)
        if world > 1 and out is not None:
            state["result"] = dict(out)               # a stall in a later phase reports at least this line

        def release(pl, co):
            if co is not None:
                co._plan = None                       # the plan is ours, not the pool's
            pl.close()

        if world == 1:
            if out is not None and full and config in ("headline", "cfg2") and not args.brief:
                # (not under --brief: the PMC passes of tools/profile_round.sh average per dispatch, and
                #  the pipelined host call launches the same kernels on row blocks)
                # PCIe-inclusive rate (host numpy in, reconstruction out as a numpy array) - never
                # `value`.  First call: the result's page-locked block is allocated; later calls
                # reuse it from the host pool (wavelets_amd/_lib.py _HostPool), which is the steady
                # state of a frame loop.
                img = make_strip(nrows, W, seed=0)
                rates = []
                for _ in range(4):
                    t = time.perf_counter()
                    if two_call:
                        plan.upload(PLANE_INPUT, img)
                        step()
                        recon = plan.download(PLANE_OUT)
                    else:       # the same three legs, pipelined over blocks of rows (wt_decompose_sum_host)
                        recon = plan.decompose_sum_host(img, level, PLANE_OUT)
                    rates.append(H * W / (time.perf_counter() - t) / 1e6)
                    del recon
                out["pcie_inclusive_first_call_mpix_s"] = round(rates[0], 1)
                out["pcie_inclusive_mpix_s"] = round(max(rates[1:]), 1)
                # for comparison: upload, passes, download one after the other
                t = time.perf_counter()
                plan.upload(PLANE_INPUT, img)
                step()
                recon = plan.download(PLANE_OUT)
                out["pcie_inclusive_serial_mpix_s"] = round(H * W / (time.perf_counter() - t) / 1e6, 1)
                del recon
                if config == "headline":
                    # a SEQUENCE of frames, numpy to numpy (the reference's real use: image series, ref utils.py:83-102 per
                    # frame): denoise(frame, [5, 3]) in a loop against sequence.denoise_many on three lanes - upload of one
                    # frame, passes of another, download of a third side by side; bit-identical results
                    try:
                        import wavelets_amd as WA
                        nseq = 16
                        frames = [img + np.float32(i) for i in range(nseq)]
                        loop_ms = many_ms = None
                        for _ in range(2):                 # (first round: lanes, plans, page-locked result blocks)
                            t = time.perf_counter()
                            res = [WA.denoise(f, [5, 3]) for f in frames]
                            loop_ms = (time.perf_counter() - t) / nseq * 1e3
                            ref7 = res[7].copy()
                            del res
                        for _ in range(3):
                            t = time.perf_counter()
                            res = WA.denoise_many(frames, [5, 3])
                            dt = (time.perf_counter() - t) / nseq * 1e3
                            many_ms = dt if many_ms is None or _ > 0 and dt < many_ms else many_ms
                            same = bool(np.array_equal(res[7], ref7))
                            del res
                        del frames
                        out["pcie_inclusive_sequence_mpix_s"] = round(H * W / many_ms / 1e3, 1)
                        out["sequence"] = {"what": f"{nseq} frames of {side}^2 float32, denoise(frame, [5, 3]) with the per-frame MAD noise "
                                                   "estimate, numpy to numpy (pageable inputs, page-locked results)",
                                           "loop_ms_per_frame": round(loop_ms, 3), "denoise_many_ms_per_frame": round(many_ms, 3),
                                           "lanes": 3, "bitwise_equal_to_the_loop": same,
                                           "loop_mpix_s": round(H * W / loop_ms / 1e3, 1)}
                    except Exception as e:
                        out["sequence"] = {"error": repr(e)}
                if config == "headline":
                    # what a one-shot user sees: the FIRST denoise(img, [5, 3]) of a fresh process, numpy to numpy,
                    # next to its steady state (tools/first_call.py in a child process; ~3 s of wall time)
                    try:
                        import subprocess

                        def child(*extra):
                            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "first_call.py"), str(side), *extra],
                                               capture_output=True, text=True, timeout=120)
                            got = {}
                            for ln in r.stdout.splitlines():
                                name, _, ms = ln.rpartition(" ms")[0].rpartition(" ")
                                if ln.rstrip().endswith("ms") and ms:
                                    got[name.strip()] = float(ms)
                            return got, [v for k, v in got.items() if k.startswith("denoise(img")]
                        # (a) context created and synced first (the context's warm-up thread has finished: the
                        #     runtime's copy queues exist), (b) no sync: context creation straight into the call
                        got, calls = child()
                        got_b, calls_b = child("nosync")
                        ctx_key, sync_key = "context (HIP runtime init, stream, scratch)", "sync (joins the context's warm-up thread)"
                        out["first_call"] = {"what": f"denoise(img, [5, 3]) at {side}^2 float32, numpy to numpy, fresh process",
                                             "first_ms": calls[0], "steady_ms": min(calls[1:]),
                                             "context_ms": got.get(ctx_key), "warmup_join_ms": got.get(sync_key),
                                             "library_load_ms": got.get("load libwatroo_hip.so"),
                                             "one_shot": {"what": "the same without a sync between context creation and the call",
                                                          "context_ms": got_b.get(ctx_key), "first_ms": calls_b[0]}}
                    except Exception as e:
                        out["first_call"] = {"error": repr(e)}
            if out is not None and not args.no_cpu and not args.brief:
                out["cpu_baseline"] = cpu_baseline(config, side, family, level,
                                                   budget=15.0 if full else 5.0)
            release(plan, coefficients)
            return out

        # ------------------------------------------------------------------ multi-GPU extras
        # (1) the same steps with the exchanges in serial order on the compute stream (no overlap)
        # (the library's default hides every exchange behind the interior rows of its own pass on a second,
        #  high-priority stream; whether that pays on a given node - RCCL's kernels need compute units of
        #  their own - is measured here, and the faster order becomes the reported value)
        ab_steps = max(3, min(steps, 10))
        overlap = {"default": "on", "chosen": "on", "ms_per_step_on": round(m["elapsed"] / steps * 1e3, 4)}
        _lib.set_option("overlap", 0)
        try:
            for _ in range(3):
                step()
            e_off, _ = timed(step, ab_steps)
            overlap["ms_per_step_off"] = round(e_off / ab_steps * 1e3, 4)
            overlap["steps_off"] = ab_steps
            if e_off / ab_steps < 0.98 * m["elapsed"] / steps and not args.keep_overlap:      # (the same on every rank: MAX-reduced times)
                overlap["chosen"] = "off"
                m = measure(step, steps, spin=False)                # the full timed region in the serial order
                overlap["ms_per_step_off"] = round(m["elapsed"] / steps * 1e3, 4)
                overlap["steps_off"] = steps
                out = report(m)
        finally:
            if overlap["chosen"] == "on":
                _lib.set_option("overlap", 1)
        # (1b) how many compute units the interior launch should leave to RCCL's kernels while an exchange runs
        # beside it (option overlap_reserve, default 16 of 256: a guess until a node has run this).  Measured, like
        # the order itself: a value at least 2 % faster than the default becomes the one the line is timed with.
        if overlap["chosen"] == "on":
            # (its own, short limit like the placement A/B below: if a rank stalls in here the stored line goes out
            #  after two minutes, not at the run's limit)
            from wavelets_amd.launch import Watchdog
            sweep_limit = min(120.0, max(5.0, args.time_limit / 5))
            dog_sweep = Watchdog(sweep_limit, lambda: on_expire(f"the overlap_reserve sweep did not finish within {sweep_limit:.0f} s"),
                                 code=lambda: 0)
            try:
                base_s = m["elapsed"] / steps
                sweep, best_r, best_s = {"16": round(base_s * 1e3, 4)}, 16, base_s
                for r in (0, 8, 32, 64):
                    _lib.set_option("overlap_reserve", r)
                    for _ in range(3):
                        step()
                    e_r, _ = timed(step, ab_steps)                  # (MAX over the ranks: the same choice everywhere)
                    sweep[str(r)] = round(e_r / ab_steps * 1e3, 4)
                    if e_r / ab_steps < 0.98 * best_s:
                        best_r, best_s = r, e_r / ab_steps
                _lib.set_option("overlap_reserve", best_r)
                overlap["reserve_sweep_ms_per_step"] = sweep
                overlap["reserve_chosen"] = best_r
                if best_r != 16:
                    m = measure(step, steps, spin=False)            # the full timed region with the chosen value
                    overlap["ms_per_step_on"] = round(m["elapsed"] / steps * 1e3, 4)
                    out = report(m)
            except Exception as e:                                  # (never lose the line over the sweep)
                _lib.set_option("overlap_reserve", 16)
                overlap["reserve_sweep_error"] = repr(e)
            finally:
                dog_sweep.cancel()
        # (2) the ramp check of the halo exchange on the plan that was timed
        check = halo_check(plan)
        planes_ab = {"chosen": "hipMalloc", "hipMalloc_ms_per_step": round(m["elapsed"] / steps * 1e3, 4),
                     "overlap_during_this_test": overlap["chosen"]}
        # (2b) what this line SHOULD read: every rank times its own strip as a stand-alone image (same rows,
        # same kernels, no exchange, no split launches) on the GPU and in the run it was measured on; the halo
        # rows a step moves to each neighbour are priced at one xGMI link.  predicted = the slower of the two
        # under the overlapped order, their sum under the serial one; measured / predicted says at first
        # contact with a node whether compute, the exchange or the launch split lost the rest.
        model, solo_ms = None, None
        try:                                      # (every rank reaches the gather below, whatever happens here)
            solo = _lib.Plan(ctx, nrows, W, fam, level)
            try:
                solo.upload(PLANE_INPUT, img0)
                for _ in range(3):
                    solo.decompose_sum(PLANE_INPUT, level, PLANE_OUT, flags & ~_lib.FLAG_NO_EXCHANGE)
                ctx.sync()
                ctx.timer_start()
                for _ in range(ab_steps):
                    solo.decompose_sum(PLANE_INPUT, level, PLANE_OUT, flags & ~_lib.FLAG_NO_EXCHANGE)
                solo_ms = round(ctx.timer_stop() / ab_steps, 4)
            finally:
                solo.close()
        except Exception as e:
            solo_ms = {"error": repr(e)}
        solo_all = group.gather(solo_ms)
        try:
            if rank == 0 and any(not isinstance(v, float) for v in solo_all):
                model = {"error": "a rank could not time its strip", "strip_compute_ms_per_rank": solo_all}
            elif rank == 0:
                link_gbs = XGMI_LINK_GBS
                neigh_bytes = sum(h for _, _, h in sched) * W * 4            # to EACH neighbour, per step
                exch_ms = neigh_bytes / (link_gbs * 1e9) * 1e3 + 0.02 * len(sched)     # + ~20 us of launch / handshake per pass
                comp_ms = max(solo_all)
                measured_ms = m["elapsed"] / steps * 1e3
                pred = {"overlapped": max(comp_ms, exch_ms), "serial": comp_ms + exch_ms}
                order = "overlapped" if overlap["chosen"] == "on" else "serial"
                model = {"strip_compute_ms_per_rank": solo_all, "strip_compute_ms": round(comp_ms, 4),
                         "strip_compute_is": f"this rank's {nrows} x {W} strip as a stand-alone image (no exchange, whole launches), "
                                             f"{ab_steps} steps, timed in this run",
                         "exchange_bytes_per_neighbour_per_step": neigh_bytes,
                         "link_GBs": link_gbs, "exchange_ms": round(exch_ms, 4),
                         "exchange_is": "halo bytes to one neighbour / one xGMI link (up and down use different links) + 20 us per pass",
                         "predicted_ms_per_step": {k: round(v, 4) for k, v in pred.items()},
                         "order": order,
                         "predicted_value_mpix_s": round(H * W / (pred[order] * 1e-3) / 1e6, 1),
                         "measured_ms_per_step": round(measured_ms, 4),
                         "measured_over_predicted_time": round(measured_ms / pred[order], 3),
                         "n1_equivalent": "value / n_gpus against the N = 1 line of the same driver run is the scaling efficiency; "
                                          "strip_compute_ms x n_gpus / that line's ms_per_step x (8192^2 x n_gpus / image pixels) "
                                          "says how much of it the strip shape alone costs"}
        except Exception as e:
            model = {"error": repr(e)}
        if out is not None:
            out["overlap"] = overlap
            out["halo_selfcheck"] = check
            out["strip_planes"] = planes_ab
            out["scaling_model"] = model
            state["result"] = out                     # from here on the time limit reports THIS line
        # (2c) replicas: the natural sharding of the reference's real workload (a SEQUENCE of frames, ref utils.py:83-102 per
        # frame): every rank runs the N = 1 headline step on its own 8192^2 frame at the same time - no RCCL, no halo -
        # between the same fences as the line; aggregate = N frames' pixels / the slowest rank's time.  A second,
        # collective-free multi-process datum from the one run a node gives.
        replicas = None
        rep_ms, rep_err = None, None
        rep_steps = max(10, min(steps, 30))
        rp = None
        try:                                       # (set-up: no collective inside, every rank reaches the fence below)
            rp = _lib.Plan(ctx, 8192, 8192, _lib.B3SPLINE, LEVEL)
            rp.upload(PLANE_INPUT, make_strip(8192, 8192, seed=100 + rank))
            for _ in range(5):
                rp.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
        except Exception as e:
            rep_err = repr(e)
        fence()                                    # all ranks start their steps together ...
        if rep_err is None:
            try:
                t_r = time.perf_counter()
                for _ in range(rep_steps):
                    rp.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
                ctx.sync()
                rep_ms = (time.perf_counter() - t_r) / rep_steps * 1e3
            except Exception as e:
                rep_err = repr(e)
        fence()                                    # ... and nobody leaves before the slowest has finished
        if rp is not None:
            try:
                rp.close()
            except Exception:
                pass
        rep_all = group.gather(rep_ms if rep_err is None else {"error": rep_err})
        if rank == 0:
            if all(isinstance(v, float) for v in rep_all):
                replicas = {"what": f"{world} independent 8192x8192 float32 frames, one per GPU, the N=1 headline step (b3spline L=6 decompose + "
                                    "plane sum, device-resident) at the same time; no RCCL",
                            "ms_per_step_per_rank": [round(v, 4) for v in rep_all], "steps": rep_steps,
                            "value": round(world * 8192 * 8192 / (max(rep_all) * 1e-3) / 1e6, 1), "unit": "Mpix/s", "scaling": "weak"}
            else:
                replicas = {"error": rep_all}
        if out is not None:
            out["replicas"] = replicas
        group.barrier()
        state["main_done"] = True
        release(plan, coefficients)
        if os.environ.get("WT_BENCH_TEST_HANG_AFTER_MAIN"):       # testing aid: a later phase that never finishes -
            time.sleep(1e6)                                        # the watchdog must emit the stored line
        # (3) strip planes over scattered 2-MiB chunks (what single-GPU plans use) instead of
        # plain hipMalloc: measured AFTER the line above is safe, guarded by the same ramp check on the
        # mapped planes over the real transport; the faster placement becomes the reported value.
        if not args.no_scatter_ab and not args.no_exchange:
            # (its own, short limit: mapped planes have never met a real xGMI transport - if this phase
            #  stalls, the stored line goes out after two and a half minutes, not at the run's limit)
            from wavelets_amd.launch import Watchdog
            ab_limit = min(150.0, max(5.0, args.time_limit / 4))
            dog_ab = Watchdog(ab_limit, lambda: on_expire(f"the scattered-planes A/B did not finish within {ab_limit:.0f} s"),
                              code=lambda: 0)
            try:
                _lib.set_option("scatter_strips", 1)
                plan2 = make_plan()
                mapped = plan2.memory()[1]
                ok2 = bool(group.allreduce(mapped > 0, all))
                if not ok2:
                    planes_ab["scattered"] = "not available (planes below 8 MiB, or no virtual-memory API)"
                    plan2.close()
                else:
                    check2 = halo_check(plan2)
                    planes_ab["scattered_halo_selfcheck_ok"] = check2["ok"]
                    if check2["ok"]:
                        plan2.upload(PLANE_INPUT, img0)
                        step2, _ = make_step(plan2)
                        m2 = measure(step2, steps)
                        planes_ab["scattered_ms_per_step"] = round(m2["elapsed"] / steps * 1e3, 4)
                        if m2["elapsed"] < 0.99 * m["elapsed"]:
                            out2 = report(m2)
                            if out2 is not None:
                                planes_ab["chosen"] = "scattered"
                                out2["overlap"] = dict(overlap, note="on / off compared on the hipMalloc'ed planes")
                                out2["halo_selfcheck"] = check2
                                out2["strip_planes"] = planes_ab
                                out2["scaling_model"] = dict(model or {}, note="strip timed on hipMalloc'ed planes; the line on scattered ones")
                                out2["replicas"] = replicas
                                out = out2
                                state["result"] = out
                    plan2.close()
            except Exception as e:
                planes_ab["scattered_error"] = repr(e)
            finally:
                dog_ab.cancel()
                _lib.set_option("scatter_strips", 0)
        _lib.set_option("overlap", 1)
        _lib.set_option("overlap_reserve", 16)
        return out

    out = run_workload(args.config, args.steps, args.warmup, full=True)
    if (out is not None and args.config == "headline" and world == 1 and not args.no_configs
            and not args.size and not args.rows and not args.unfused and not args.two_call
            and not args.brief):
        # every other single-GPU BASELINE.json configuration, timed in the same run
        extra = {}
        for cfg in ("cfg2", "cfg3", "cfg5"):
            try:
                # (cfg2's step is 0.18 ms: 20 steps are a 3.6 ms timed region, which read 0.170-0.202 ms
                #  across boxes and runs; 100 steps average over clock / power-state noise)
                nst = {"cfg2": max(100, args.steps), "cfg3": max(5, min(args.steps, 20)), "cfg5": 5}[cfg]
                o = run_workload(cfg, nst, args.warmup, full=False)
            except Exception as e:          # never lose the headline line over an extra configuration
                extra[cfg] = {"error": repr(e)}
                continue
            r = o["roofline"] or {}
            extra[cfg] = {"metric": o["metric"], "value": o["value"], "unit": o["unit"],
                          "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                          **{k: o[k] for k in ("ms_per_step_median", "ms_per_step_min", "ms_per_step_max", "ms_per_step_samples") if k in o},
                          "workload": o["config"]["workload"], "schedule": o["config"]["schedule"],
                          "bytes_per_pixel": o["whole_path"]["bytes_per_pixel"],
                          "frac_of_hbm_peak": o["whole_path"]["frac_of_hbm_peak"],
                          "hbm_achieved": o["whole_path"].get("hbm_achieved"),
                          "hbm_bytes_per_pixel": o["whole_path"].get("hbm_bytes_per_pixel"),
                          "dominant_kernel": r.get("kernel"),
                          "dominant_kernel_frac": r.get("frac"),
                          "dominant_kernel_hbm_achieved": r.get("hbm_achieved"),
                          "dominant_kernel_valu_frac": (r.get("valu") or {}).get("frac"),
                          "kernels": o["kernels"],
                          "cpu_baseline": o.get("cpu_baseline")}
        out["configs"] = extra
        # the N > 1 lines' image on ONE GPU (32768^2: 4 GiB planes, 44 GiB for the flow): the same-image anchor of a 1 -> N
        # curve, from the same run.  Data: a 4096-row N(0,1) block repeated down the image (the kernels have no
        # data-dependent path; generating 2^30 normal samples on the host would take longer than every other entry).
        try:
            big = MULTI_GPU_SIDE
            pl = _lib.Plan(ctx, big, big, _lib.B3SPLINE, LEVEL)
            pl.upload(PLANE_INPUT, np.tile(make_strip(4096, big, seed=0), (big // 4096, 1)))
            for _ in range(2):
                pl.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
            ctx.sync()
            nb = 5
            t_b = time.perf_counter()
            for _ in range(nb):
                pl.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
            ctx.sync()
            ms_b = (time.perf_counter() - t_b) / nb * 1e3
            sm = []
            for _ in range(20):
                ctx.timer_start()
                pl.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
                sm.append(ctx.timer_stop())
            pl.close()
            bppb = 8.0 * (LEVEL + 2)
            out["n1_same_image"] = {"workload": f"{big}x{big} float32 (a 4096-row N(0,1) block repeated), b3spline L={LEVEL}, decompose + plane sum "
                                                "on ONE GPU: the image the N>1 lines split into row strips; device-resident",
                                    "value": round(big * big / ms_b / 1e3, 1), "unit": "Mpix/s", "ms_per_step": round(ms_b, 4), "steps": nb,
                                    **step_spread(sm), "bytes_per_pixel": bppb,
                                    "frac_of_hbm_peak": round(bppb * big * big / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        except Exception as e:
            out["n1_same_image"] = {"error": repr(e)}
        # widths the 16-byte groups do not divide (the reference takes any width, ref wavelets.py:35-45): the headline
        # step on images a little narrower than 8192^2 and on the 3066^2 frame of tools/e2e_check.py, per pixel
        # against the 8192^2 figure of this line
        odd = {}
        for hs, ws in ((8190, 8190), (8191, 8191), (3066, 3066)):
            try:
                pl = _lib.Plan(ctx, hs, ws, _lib.B3SPLINE, LEVEL)
                pl.upload(PLANE_INPUT, make_strip(hs, ws, seed=0))
                for _ in range(5):
                    pl.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
                ctx.sync()
                sm = []
                for _ in range(20):
                    ctx.timer_start()
                    pl.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT, _lib.FLAG_FUSED)
                    sm.append(ctx.timer_stop())
                pl.close()
                sp = step_spread(sm)
                odd[f"{hs}x{ws}"] = {**sp, "value": round(hs * ws / sp["ms_per_step_median"] / 1e3, 1), "unit": "Mpix/s",
                                     "per_pixel_vs_headline": round((sp["ms_per_step_median"] / (hs * ws)) / (out["ms_per_step"] / float(out["config"]["image"][0] * out["config"]["image"][1])), 3)}
            except Exception as e:
                odd[f"{hs}x{ws}"] = {"error": repr(e)}
        out["odd_width"] = {"what": "headline step (b3spline L=6 decompose + plane sum) at widths that are not multiples of 4; "
                                    "per_pixel_vs_headline = time per pixel / the 8192x8192 line's (1.0 = no cliff)", **odd}
        # the reference's DEFAULT dtype (README flows are float64, ref wavelets.py:297,319-320): the
        # headline workload in float64 on the fused double passes (wt64_decompose_sum) - not a
        # BASELINE.json configuration, reported beside them
        try:
            import wavelets_amd as WA
            p64 = _lib.Plan64(ctx, 8192, 8192, tuple(float(t) for t in WA.B3spline.coefficients_1d), LEVEL)
            p64.upload(PLANE_INPUT, make_strip(8192, 8192, seed=0).astype(np.float64))
            for _ in range(5):
                fused64 = p64.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT)
            ctx.sync()
            n64 = max(5, min(args.steps, 20))
            ctx.timer_start()
            for _ in range(n64):
                p64.decompose_sum(PLANE_INPUT, LEVEL, PLANE_OUT)
            ms64 = ctx.timer_stop() / n64
            p64.close()
            bpp64 = 16.0 * (LEVEL + 2)                               # SURVEY 8(d) at 8 bytes per sample
            out["float64"] = {"workload": "8192x8192 float64, b3spline L=6, decompose (7 planes in HBM) + plane sum "
                                          "(wt64_decompose_sum); device-resident",
                              "value": round(8192 * 8192 / ms64 / 1e3, 1), "unit": "Mpix/s",
                              "ms_per_step": round(ms64, 4), "fused_passes": bool(fused64),
                              "bytes_per_pixel": bpp64,
                              "frac_of_hbm_peak": round(bpp64 * 8192 * 8192 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        except Exception as e:                                       # never lose the headline line over the extra
            out["float64"] = {"error": repr(e)}
        # ... and BASELINE configs[2] in float64: 8192^2 Triangle L = 8 + denoise([5,3,2]) + sum on the
        # float64 engine (fused double passes, the first one histogramming |w_0|; exact median on the
        # gathered list; wt64_denoise_sum between the passes).  SURVEY 8(d) at 8 bytes per sample: 216 B/pixel.
        try:
            import wavelets_amd as WA
            from wavelets_amd.wavelets import _decompose_denoise_sum
            p64 = _lib.Plan64(ctx, 8192, 8192, tuple(float(t) for t in WA.Triangle.coefficients_1d), 8)
            p64.upload(PLANE_INPUT, make_strip(8192, 8192, seed=0).astype(np.float64))
            c64 = WA.Coefficients(p64, WA.Triangle(2))
            tr64 = WA.AtrousTransform(WA.Triangle)

            def step64():
                c64.noise = None
                _decompose_denoise_sum(tr64, p64, 8, c64, [5, 3, 2], soft_threshold=True, write_back=True)
            for _ in range(3):
                step64()
            ctx.sync()
            n64 = max(5, min(args.steps, 20))
            t64 = time.perf_counter()
            for _ in range(n64):
                step64()
            ctx.sync()
            ms64 = (time.perf_counter() - t64) / n64 * 1e3
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(3):
                step64()
            k64 = {k: round(ms / 3, 4) for k, (calls, ms) in ctx.profile_entries().items()}
            ctx.profile(False)
            c64._plan = None
            p64.close()
            bpp = 2.0 * whole_path_bytes_per_pixel("cfg3", 8)
            out["float64_cfg3"] = {"workload": "8192x8192 float64, triangle L=8, decompose + MAD noise + denoise([5,3,2]) soft "
                                               "+ plane sum (BASELINE configs[2] on the float64 engine); device-resident",
                                   "value": round(8192 * 8192 / ms64 / 1e3, 1), "unit": "Mpix/s",
                                   "ms_per_step": round(ms64, 4), "steps": n64, "bytes_per_pixel": bpp,
                                   "frac_of_hbm_peak": round(bpp * 8192 * 8192 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "kernels_ms_per_step": k64}
        except Exception as e:
            out["float64_cfg3"] = {"error": repr(e)}
        # ... and BASELINE configs[4] in float64 (round 5): wow(bilateral=1, denoise_coefficients=[5,2]) at 8192^2,
        # 11 scales, on the float64 engine - what an int16 / FITS frame becomes (ref wavelets.py:297,319-320):
        # the float64 bilateral march (wt_bilateral64.h, VALU-bound: 24 polynomial exponentials per pixel per
        # scale), the fused float64 wow updates (wt_stencil.h for double) on the side stream, the plane sum.
        try:
            import wavelets_amd as WA
            from wavelets_amd import utils as WU
            lv5 = int(np.round(np.log2(8192) - np.log2(5)))
            sb5 = [1] * (lv5 + 1)
            p64 = _lib.Plan64(ctx, 8192, 8192, tuple(float(t) for t in WA.B3spline.coefficients_1d), lv5)
            p64.upload(PLANE_INPUT, (make_strip(8192, 8192, seed=0)
                                     + 3 * np.sin(np.arange(8192, dtype=np.float32) / 50.)[None, :]).astype(np.float64))
            c64 = WA.Coefficients(p64, WA.B3spline(2), sb5)
            tr64 = WA.AtrousTransform(WA.B3spline, bilateral=sb5)

            def step64w():
                tr64._run(p64, lv5)
                c64.noise = None
                WU._wow_device(c64, lv5, [], True, [5, 2], True, False, 3.2, None, None, 0)
            for _ in range(2):
                step64w()
            ctx.sync()
            n64 = 5
            t64 = time.perf_counter()
            for _ in range(n64):
                step64w()
            ctx.sync()
            ms64 = (time.perf_counter() - t64) / n64 * 1e3
            ctx.profile(True)
            ctx.profile_reset()
            for _ in range(2):
                step64w()
            k64 = {k: round(ms / 2, 4) for k, (calls, ms) in ctx.profile_entries().items()}
            ctx.profile(False)
            c64._plan = None
            p64.close()
            bpp = 2.0 * whole_path_bytes_per_pixel("cfg5", lv5)
            f32_ms = (extra.get("cfg5") or {}).get("ms_per_step")
            out["float64_cfg5"] = {"workload": "8192x8192 float64 (N(0,1) + 3 sin(x/50)), b3spline, wow(bilateral=1, denoise_coefficients=[5,2]), "
                                               "11 scales (BASELINE configs[4] on the float64 engine); device-resident",
                                   "value": round(8192 * 8192 / ms64 / 1e3, 1), "unit": "Mpix/s",
                                   "ms_per_step": round(ms64, 4), "steps": n64, "bytes_per_pixel": bpp,
                                   "frac_of_hbm_peak": round(bpp * 8192 * 8192 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "vs_float32_cfg5": round(ms64 / f32_ms, 3) if f32_ms else None,
                                   "bound": "VALU (float64 bilateral march: ~330 double-precision operations per pixel per scale, "
                                            "0.93 of the issue rate at the 2.0 GHz the chip holds under it - profiles/r06_c_cfg5_clocks.csv)",
                                   "kernels_ms_per_step (overlapped kernels both count)": k64}
        except Exception as e:
            out["float64_cfg5"] = {"error": repr(e)}
    if out is not None:
        brief = out.pop("_brief")
        emit(brief if args.brief else json.dumps(out))
    if dog is not None:
        dog.cancel()
    if group is not None:
        group.barrier()
        group.close()


if __name__ == "__main__":
    main()
