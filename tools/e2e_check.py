import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import wavelets_amd as W
from wavelets_amd import utils as WU
rng = np.random.default_rng(0)
def t(f, n=2):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return r, (time.perf_counter() - t0) / n * 1e3
# RL on 3072^2 (mixed radix) vs the direct periodic form
img = (rng.random((3072, 3072)) * 100 + 10).astype(np.float32)
y, x = np.mgrid[-32:33, -32:33]
psf = np.exp(-(x * x + y * y) / 50.0).astype(np.float32); psf /= psf.sum()
a, ta = t(lambda: W.richardson_lucy(img, psf, iterations=3, fft=True))
keep = WU._FFT_MIN_TAPS; WU._FFT_MIN_TAPS = 1 << 30
b, tb = t(lambda: W.richardson_lucy(img, psf, iterations=3, fft=True), 1)
WU._FFT_MIN_TAPS = keep
print(f"RL fft 3072^2 65x65: fft {ta:.1f} ms, direct {tb:.1f} ms, max rel diff {np.abs(a-b).max()/np.abs(b).max():.2e}", flush=True)
# odd-height, 5-smooth: 3645 x 3000 float64
img64 = rng.random((3645, 3000)) * 100 + 10
a, ta = t(lambda: W.richardson_lucy(img64, psf.astype(np.float64), iterations=2, fft=True), 1)
WU._FFT_MIN_TAPS = 1 << 30
b, tb = t(lambda: W.richardson_lucy(img64, psf.astype(np.float64), iterations=2, fft=True), 1)
WU._FFT_MIN_TAPS = keep
print(f"RL fft 3645x3000 f64: fft {ta:.1f} ms, direct {tb:.1f} ms, max rel diff {np.abs(a-b).max()/np.abs(b).max():.2e}", flush=True)
# wow on odd sizes, both dtypes, bilateral
for dt in (np.float32, np.float64):
    im = (rng.standard_normal((3001, 4999)) + 3 * np.sin(np.arange(4999) / 50.0)[None, :]).astype(dt)
    (r, c), tw = t(lambda: W.wow(im, bilateral=1, denoise_coefficients=[5, 2]), 1)
    print(f"wow bilateral 3001x4999 {np.dtype(dt).name}: {tw:.1f} ms, out {r.dtype} finite {np.isfinite(r).all()} planes {len(c)}", flush=True)
# int16 denoise
im16 = (rng.standard_normal((4096, 6000)) * 50 + 1000).astype(np.int16)
r, td = t(lambda: W.denoise(im16, [5, 3]))
print(f"denoise int16 4096x6000: {td:.1f} ms -> {r.dtype}", flush=True)
# 17-tap user scaling function
class Long(W.B3spline):
    pass
try:
    from wavelets_amd.wavelets import AbstractScalingFunction
    class Han17(AbstractScalingFunction):
        coefficients_1d = np.hanning(19)[1:-1] / np.hanning(19)[1:-1].sum()
        sigma_e_1d = sigma_e_2d = sigma_e_3d = np.ones(16)
        def __init__(self, *a, **k):
            super().__init__("han17", *a, **k)
    im = rng.standard_normal((2048, 2048)).astype(np.float32)
    c, tt = t(lambda: W.AtrousTransform(Han17)(im, 5))
    print(f"17-tap transform 2048^2 L=5: {tt:.2f} ms, telescopes {np.abs(np.sum(c.data, axis=0) - im).max():.2e}", flush=True)
except Exception as e:
    print("17-tap:", repr(e))
