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
    assert out["config"]["image"] == [size, size] and out["config"]["parallelism"] == f"strips{nranks}"
    assert out["halo_selfcheck"]["ok"], out["halo_selfcheck"]
    # per pass: the exchange on the communication stream, the interior launch beside it, the edge rows
    sp = out["strip_passes"]
    assert len(sp["rank0"]) == len(out["config"]["schedule"]) == len(sp["max_over_ranks"])
    for e, (s0, ns, halo) in zip(sp["rank0"], out["config"]["schedule"]):
        assert e["scales"] == [s0, s0 + ns] and e["halo_rows"] == halo and e["kernel"].startswith("wt_fused")
        assert e["exchange_ms"] > 0 and e["interior_ms"] > 0 and e["edge_ms"] > 0, e
    for k, v in out["kernels"].items():
        if k.startswith("wt_fused"):
            assert set(v["parts_ms"]) == {"interior", "edge"} and v["calls_per_step"] == 1, (k, v)
    # the same steps with the exchanges in serial order, and the placement A/B of the strip planes
    ov = out["overlap"]
    assert ov["default"] == "on" and ov["ms_per_step_on"] > 0 and ov["ms_per_step_off"] > 0
    pl = out["strip_planes"]
    assert pl["chosen"] in ("hipMalloc", "scattered") and pl["hipMalloc_ms_per_step"] > 0
    if size * (size // nranks) * 4 >= (8 << 20):          # planes of 8 MiB and more can be scattered
        assert pl.get("scattered_halo_selfcheck_ok") is True, pl
        assert pl["scattered_ms_per_step"] > 0
    chosen_ms = pl["scattered_ms_per_step"] if pl["chosen"] == "scattered" else pl["hipMalloc_ms_per_step"]
    assert abs(out["ms_per_step"] - chosen_ms) < 1e-3


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
