import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import wavelets_amd as WA
from wavelets_amd import _lib as L
class Long17(WA.wavelets.AbstractScalingFunction):
    coefficients_1d = np.hanning(19)[1:-1] / np.hanning(19)[1:-1].sum()
    sigma_e_1d = sigma_e_2d = sigma_e_3d = np.array([0.6, 0.2, 0.09, 0.04, 0.02, 0.01])
    def __init__(self, *a, **k): super().__init__('long17', *a, **k)
img = np.random.default_rng(0).standard_normal((2048, 2048)).astype(np.float32)
ctx = L.default_context()
for rep in range(3):
    t = time.perf_counter(); c = WA.AtrousTransform(Long17)(img, 3); ctx.sync(); print("transform L=3, 17 taps, 2048^2: %.2f ms" % ((time.perf_counter() - t) * 1e3))
# the K**2-tap single launch the axis-by-axis form replaced (round 3), for comparison
from wavelets_amd.wavelets import _filter_taps, _generic_plan, _PAD_MODES, PLANE_INPUT, PLANE_OUT, PLANE_NONE, release_plan
sf = Long17(2)
plan = _generic_plan(img.shape, False, 0)
plan.upload(PLANE_INPUT, img)
for s in (0, 2):
    offs, wts = _filter_taps(sf.kernel, s)
    plan.taps_conv(PLANE_INPUT, PLANE_NONE, PLANE_OUT, offs, wts, None, pad_mode=_PAD_MODES["symmetric"]); ctx.sync()
    ctx.timer_start()
    for _ in range(3):
        plan.taps_conv(PLANE_INPUT, PLANE_NONE, PLANE_OUT, offs, wts, None, pad_mode=_PAD_MODES["symmetric"])
    t_full = ctx.timer_stop() / 3
    from wavelets_amd.wavelets import _generic_smooth
    _generic_smooth(plan, sf, 2, img.shape, PLANE_INPUT, PLANE_OUT, s); ctx.sync()
    ctx.timer_start()
    for _ in range(3):
        _generic_smooth(plan, sf, 2, img.shape, PLANE_INPUT, PLANE_OUT, s)
    t_sep = ctx.timer_stop() / 3
    print(f"conv_s, s = {s}, 17 taps, 2048^2: 289-tap launch {t_full:.3f} ms, axis by axis {t_sep:.3f} ms ({t_full / t_sep:.1f} x)")
release_plan(plan)
