import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def measured(what, got, ref, bound):
    """Max abs difference of two arrays, asserted against `bound` and appended to
    gpurun_out/parity_errors.log (test id, what, measured, bound) - the full-size parity tests assert
    a small multiple of what they measure, and this log is where those numbers come from."""
    d = float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\t{d:.4e}\t{bound:.4e}\n")
    except OSError:
        pass
    assert d <= bound, f"{what}: max abs diff {d:.3e} > {bound:.3e}"
    return d


def measured_tol(what, got, ref, atol, rtol=0.0, allow=0):
    """|got - ref| <= atol + rtol * |ref| elementwise, at most `allow` samples beyond it (a pixel within rounding
    of a hard threshold may flip).  Appends the WORST ratio |got - ref| / (atol + rtol |ref|) - the error in units
    of its tolerance - to gpurun_out/parity_errors.log: round 5 set every tolerance that goes through here to
    4 x what that log recorded on MI355X (ratio 0.25)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    tol = atol + rtol * np.abs(ref)
    err = np.abs(got - ref)
    # (a zero tolerance - atol = 0 at a reference value of exactly 0 - admits only an exact match: ratio 0 or inf, never 0 / 0)
    ratio = np.divide(err, tol, out=np.where(err == 0, 0.0, np.inf), where=tol > 0)
    order = np.sort(ratio, axis=None)
    worst = float(order[-1 - allow]) if order.size > allow else 0.0
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\tratio {worst:.4e}\t"
                    f"atol {atol:.4e} rtol {rtol:.1e} allow {allow}\n")
    except OSError:
        pass
    assert worst <= 1.0, (f"{what}: {int((ratio > 1.0).sum())} of {ratio.size} samples beyond atol {atol:.3e} + rtol {rtol:.1e} |ref| "
                          f"(allowed {allow}), worst at {float(order[-1]):.3f} x the tolerance")
    return worst


# Full-size parity bounds (round 3): 4 x the error MEASURED on MI355X against the C oracle for
# N(0,1) float32 images (gpurun_out/parity_errors.log of the run that set them; max|input| = 5.98):
# plane s of the transform, either family, fused passes vs the oracle's dense K x K form:
#   measured 7.2e-7, 4.8e-7, 2.4e-7, 1.0e-7, 5.2e-8, 3.0e-8, 1.5e-8, 6.5e-9, 3.7e-9
#   (= 1 ulp of the largest coefficients: the planes shrink by ~2x per scale)
# reconstruction (sum of planes) 9.5e-7, denoised reconstruction of cfg3 1.43e-6.
# The bounds scale with max|input| like the rounding errors they budget for.  (Until round 3 these
# tests asserted 1e-5 * max|input| = 6e-5: 100 x what the engine delivers.)
_PLANE_BOUND = (2.9e-6, 1.9e-6, 9.6e-7, 4.2e-7, 2.1e-7, 1.2e-7, 6.0e-8, 2.7e-8, 1.5e-8)
_REF_AMAX = 5.98


def plane_bound(s, amax):
    """bound for plane s of a full-size transform of an image with max|input| = amax"""
    return _PLANE_BOUND[min(s, len(_PLANE_BOUND) - 1)] * float(amax) / _REF_AMAX


def recon_bound(amax, denoised=False):
    return (5.8e-6 if denoised else 3.9e-6) * float(amax) / _REF_AMAX


# Small-fixture bounds (round 4; the golden groups g0 / g1, ragged shapes, the random shape sweep):
# 4 x the errors measured on MI355X (gpurun_out/parity_errors.log of round 4; until then 1e-5 and 2e-5).
SMALL_PLANES = 8e-7          # planes / single-scale operators, times max|input| (measured 2.0e-7 at worst: ragged shapes)
SMALL_RECON = 5.2e-7          # np.sum(planes, axis=0) against the input image (measured 1.3e-7)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(cmd, env, tag, timeout=600):
    """Run a multi-process command (torch.distributed.run rendezvous + RCCL bootstrap over loopback
    sockets).  A launcher-level failure - non-zero exit WITHOUT a result line, seen once in a dozen
    runs on the shared boxes - is retried once on a fresh port; the output of a failed attempt is
    kept under gpurun_out/ either way.  A run that completes and reports wrong results is never
    retried."""
    import subprocess
    r = None
    for attempt in (0, 1):
        if "--master-port" in cmd:
            cmd = list(cmd)
            cmd[cmd.index("--master-port") + 1] = str(free_port())
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        if r.returncode == 0:
            return r
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"ranks_failure_{tag}_{attempt}.log"), "w") as f:
                f.write(r.stdout[-20000:] + "\n---- stderr ----\n" + r.stderr[-40000:])
        except OSError:
            pass
        # (a result line - a JSON object with a "metric" - or a reported mismatch: it ran to the end, a real failure.  The
        #  launcher's own error object - {"error": "rank 1 was killed by signal 11", ...}: a rank of three or four
        #  processes that share ONE GPU over RCCL's socket transport died inside the runtime stack, seen once in this
        #  round's ~10 runs of that set-up and never with one process per GPU - is a launcher-level failure: one retry)
        if "mismatch" in r.stdout or any(ln.startswith("{") and '"metric"' in ln for ln in r.stdout.splitlines()):
            return r
    return r
