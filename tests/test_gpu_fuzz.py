"""The randomised differential tests of tools/ under the driver's GPU run (round 5): until now "300 cases clean"
was builder-only evidence.  Fixed seeds, case counts sized for about a minute in all; each fuzzer prints one
line per failing case (replayable: same count and seed) and exits non-zero on any failure."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (script, cases, seed): tools/fuzz.py - fused / per-scale / chain schedules vs the C oracle, virtual strips vs the
# unsharded result, bilateral transform and wow() with random keywords, float64 passes, the pipelined host call,
# the generic tap-list operator; fuzz_lattice.py - lattice kernel vs row / chain kernels bitwise at dilations up to
# 4096; fuzz_round4.py - denoise (float32 / float64), richardson_lucy, pad modes, generic taps, element types;
# fuzz_round5.py - float64 stencils / bilateral / wow, side stream + early plane sum, mixed-radix FFT.
# fuzz_round6.py - bilateral march's paired loads vs the generic ones (bitwise), fused passes' fast addressing at any
# width in both precisions (bitwise), sequences on lanes vs the per-call loop (bitwise), the exact median behind the
# riding histogram's register counters on hostile distributions.
FUZZERS = [("fuzz.py", 24, 5), ("fuzz_lattice.py", 40, 5), ("fuzz_round4.py", 48, 5), ("fuzz_round5.py", 50, 5), ("fuzz_round6.py", 60, 5)]


@pytest.mark.parametrize("script,cases,seed", FUZZERS)
def test_fuzzer_runs_clean(script, cases, seed):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    t = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(cases), str(seed)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    dt = time.perf_counter() - t
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "fuzz_times.log"), "a") as f:
            f.write(f"{script} {cases} cases seed {seed}: {dt:.1f} s, rc {r.returncode}\n")
    except OSError:
        pass
    fails = [ln for ln in r.stdout.splitlines() if ln.startswith("FAIL")]
    assert r.returncode == 0 and not fails, "\n".join(fails[:10]) + "\n" + r.stdout[-1500:] + r.stderr[-3000:]
    assert f"{cases} cases" in r.stdout and "0 failures" in r.stdout, r.stdout[-500:]
