#!/usr/bin/env python3
"""Randomised differential test on the GPU box (not part of the pytest suite: minutes, not seconds).

Random (H, W, level, family) images: fused / per-scale (row + chain kernels) / chain-only
schedules against the C oracle; virtual row strips against the unsharded result (bitwise);
recursive=True against the numpy oracle; the fused passes' fast addressing against the generic
one (bitwise); and - round 2 - the bilateral transform and wow() (plain / bilateral, random keyword
combinations, up to 9 scales) against the C-backed oracle; round 3: the float64 fused passes with the
carried sum, the pipelined host-to-host call against the serial legs (bitwise, random block sizes) and
the generic tap-list operator (random kernels, pad modes, dimensionalities).  One line per failure + a summary.

    python tools/fuzz.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import atrous_numpy as O        # noqa: E402
from oracle import cref                     # noqa: E402
import wavelets_amd as W                    # noqa: E402
from wavelets_amd import _lib as L          # noqa: E402
from wavelets_amd.parallel import partition_rows, required_halo   # noqa: E402

FAM = {"b3spline": L.B3SPLINE, "triangle": L.TRIANGLE}


def planes(plan, n):
    return np.stack([plan.download(s) for s in range(n)])


TRACE = bool(os.environ.get("WT_FUZZ_TRACE"))


def note(msg):
    """WT_FUZZ_TRACE=1: one flushed line before every block of a case (to locate a GPU fault)"""
    if TRACE:
        print("    . " + msg, flush=True)


def burn(rng, case):
    """Consume exactly the random draws case `case` would make, without touching the GPU
    (WT_FUZZ_FROM=n replays cases 0..n-1 this way, so that case n sees the same parameters)."""
    kind = case % 4
    H = int(rng.integers(1, 1600)) if kind else int(rng.integers(1, 64))
    Wd = int(rng.integers(1, 3000)) if kind != 1 else int(rng.integers(1, 64))
    level = int(rng.integers(1, 10))
    fam = ("b3spline", "triangle")[int(rng.integers(0, 2))]
    rng.standard_normal((H, Wd))
    rng.integers(2, 6), rng.integers(0, 2)
    cls = W.B3spline if fam == "b3spline" else W.Triangle
    if case % 3 == 0 and H >= 24 and Wd >= 24 and H * Wd <= 1500000:
        rng.uniform(0.5, 2.0), rng.integers(0, 3), rng.integers(0, 2), rng.integers(0, 3), rng.integers(0, 3)
    if H * Wd >= 2:
        lev = min(level, len(cls(2).sigma_e()) - 1)
        rng.choice([0., 1., 2., 3., 5.], int(rng.integers(1, lev + 1)))
        rng.integers(0, 2)
    if case % 7 == 3:
        Hp, Wp = int(rng.integers(1100, 2600)), int(rng.integers(500, 1030)) * 4
        rng.integers(2, 7)
        rng.standard_normal((Hp, Wp))
        rng.choice([0, 128, 192, 256, 320, 512, 1024])
    if case % 6 == 2:
        nd = int(rng.integers(1, 4))
        shp = {1: (int(rng.integers(5, 400)),), 2: (int(rng.integers(3, 90)), int(rng.integers(3, 120))),
               3: (int(rng.integers(2, 9)), int(rng.integers(2, 20)), int(rng.integers(2, 30)))}[nd]
        rng.random(tuple(int(rng.integers(1, 6)) for _ in range(nd)))
        rng.standard_normal(shp)
        rng.choice(["symmetric", "reflect", "edge", "wrap", "constant"])
        rng.integers(0, 3)
        if rng.integers(0, 2):
            rng.standard_normal(shp)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ctx = L.default_context()
    cref.build()
    fails = 0
    first = int(os.environ.get("WT_FUZZ_FROM", "0"))
    for case in range(n_cases):
        if case < first:
            burn(rng, case)
            continue
        kind = case % 4
        if case % 10 == 0 or TRACE:
            print(f"... case {case} of {n_cases}, {fails} failures so far", flush=True)
        H = int(rng.integers(1, 1600)) if kind else int(rng.integers(1, 64))
        Wd = int(rng.integers(1, 3000)) if kind != 1 else int(rng.integers(1, 64))
        level = int(rng.integers(1, 10))
        fam = ("b3spline", "triangle")[int(rng.integers(0, 2))]
        a = rng.standard_normal((H, Wd)).astype(np.float32)
        ref = cref.decompose(a, level, fam)
        tol = 1e-5 * max(1.0, float(np.abs(a).max()))
        tag = f"case {case}: {H}x{Wd} L={level} {fam}"
        note(tag)
        got = {}
        for name, flags, row in (("fused", L.FLAG_FUSED, 1), ("perscale", 0, 1), ("chain", 0, 0)):
            note("schedule " + name)
            L.set_option("row_kernel", row)
            plan = L.Plan(ctx, H, Wd, FAM[fam], level)
            plan.upload(L.PLANE_INPUT, a)
            plan.decompose(L.PLANE_INPUT, level, flags)
            got[name] = planes(plan, level + 1)
            plan.plane_sum(0, level + 1)
            rec = plan.download(L.PLANE_OUT)
            if name == "fused":
                rec_fused = rec          # the sum carried through the passes: identical bits
                plan.fill(L.PLANE_OUT, np.nan)
                plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, flags)
                if not (np.array_equal(plan.download(L.PLANE_OUT), rec)
                        and np.array_equal(planes(plan, level + 1), got[name])):
                    fails += 1
                    print(f"FAIL {tag}: decompose_sum != decompose + plane_sum")
            plan.close()
            err = float(np.abs(got[name] - ref).max())
            if not err <= tol or not np.array_equal(rec, got[name].sum(axis=0)):
                fails += 1
                print(f"FAIL {tag} [{name}] max err {err:.3e} tol {tol:.1e}")
        L.set_option("row_kernel", 1)
        if not np.array_equal(got["perscale"], got["chain"]):
            fails += 1
            print(f"FAIL {tag}: row kernel != chain kernel")
        # virtual strips == unsharded (bitwise), when the strips are tall enough for the halo
        k = int(rng.integers(2, 6))
        fused = bool(rng.integers(0, 2))
        if H >= k and required_halo(FAM[fam], level, fused) <= H // k:
            note(f"{k} virtual strips, fused={fused}")
            flags = L.FLAG_FUSED if fused else 0
            plans = []
            for r, (row0, n) in enumerate(partition_rows(H, k)):
                p = L.Plan(ctx, H, Wd, FAM[fam], level, row0=row0, nrows=n, rank=r, nranks=k)
                p.upload(L.PLANE_INPUT, a[row0:row0 + n])
                plans.append(p)
            cur = L.PLANE_INPUT
            for i, (s0, ns, halo) in enumerate(L.schedule(FAM[fam], level, fused)):
                nxt = level if s0 + ns == level else L.PLANE_SCRATCH(i & 1)
                for up, lo in zip(plans[:-1], plans[1:]):
                    L.Plan.halo_exchange_local(up, lo, cur, halo)
                sched = L.schedule(FAM[fam], level, fused)
                use_sum = fused and all((s_, n_) in ((0, 2), (0, 3), (3, 2), (3, 3), (6, 2)) for s_, n_, _ in sched)
                for p in plans:
                    if use_sum:
                        p.decompose_pass_sum(cur, nxt, s0, ns, flags | L.FLAG_NO_EXCHANGE, L.PLANE_OUT,
                                             first=i == 0, last=i == len(sched) - 1)
                    else:
                        p.decompose_pass(cur, nxt, s0, ns, flags | L.FLAG_NO_EXCHANGE)
                cur = nxt
            sh = np.concatenate([planes(p, level + 1) for p in plans], axis=1)
            if not np.array_equal(sh, got["fused" if fused else "perscale"]):
                fails += 1
                print(f"FAIL {tag}: {k} strips (fused={fused}) != unsharded")
            if use_sum and not np.array_equal(np.concatenate([p.download(L.PLANE_OUT) for p in plans]),
                                              got["fused"].sum(axis=0) if False else rec_fused):
                fails += 1
                print(f"FAIL {tag}: {k} strips decompose_pass_sum reconstruction != unsharded")
            for p in plans:
                p.close()
        # fast vs generic addressing of the fused passes (bitwise; any width since round 6)
        if True:
            note("fast vs generic addressing")
            outs = {}
            for mode in (1, 0):
                L.set_option("fused_fast", mode)
                plan = L.Plan(ctx, H, Wd, FAM[fam], level)
                plan.upload(L.PLANE_INPUT, a)
                plan.decompose_sum(L.PLANE_INPUT, level, L.PLANE_OUT, L.FLAG_FUSED)
                outs[mode] = np.concatenate([planes(plan, level + 1), plan.download(L.PLANE_OUT)[None]])
                plan.close()
            L.set_option("fused_fast", 1)
            if not np.array_equal(outs[0], outs[1]):
                fails += 1
                print(f"FAIL {tag}: fast addressing != generic addressing")
        # bilateral transform and wow against the C-backed oracle (expf vs v_exp_f32: stated tolerance)
        if case % 3 == 0 and H >= 24 and Wd >= 24 and H * Wd <= 1500000:
            cls = W.B3spline if fam == "b3spline" else W.Triangle
            b = (a + 3 * np.sin(np.arange(Wd, dtype=np.float32) / 11.)[None, :]).astype(np.float32)
            lev = max(1, min(level, int(np.log2(min(H, Wd))) - 2, len(cls(2).sigma_e(bilateral=1)) - 1))
            sb = float(rng.uniform(0.5, 2.0))
            note(f"bilateral transform lev={lev}")
            got_b = W.AtrousTransform(cls, bilateral=sb)(b, lev).data
            ref_b = cref.decompose_bilateral(b, lev, fam, sb)
            e = float(np.abs(got_b - ref_b).max())
            if not e <= 1e-4 * float(np.abs(b).max()):
                fails += 1
                print(f"FAIL {tag}: bilateral transform L={lev} sigma_b={sb:.2f} max err {e:.3e}")
            kw = dict(denoise_coefficients=[5, 2][:int(rng.integers(0, 3))])
            if rng.integers(0, 2):
                kw["bilateral"] = 1
            if rng.integers(0, 3) == 0:
                kw["h"], kw["gamma"] = 0.5, 2.0
            if rng.integers(0, 3) == 0:
                kw["preserve_variance"] = True
            cp = lambda d: {k: (list(v) if isinstance(v, list) else v) for k, v in d.items()}
            note(f"wow {kw}")
            rec_w, coef_w = W.wow(b.copy(), cls, **cp(kw))
            ref_r, ref_p = cref.wow(b.copy(), fam, **cp(kw))
            bad = np.abs(coef_w.data - ref_p) > 2e-4 * max(1.0, float(np.abs(ref_p).max())) + 2e-4 * np.abs(ref_p)
            badr = np.abs(rec_w - ref_r) > 2e-4 * max(1.0, float(np.abs(ref_r).max())) + 2e-4 * np.abs(ref_r)
            # bilateral scales chain (expf vs v_exp_f32 in every range weight, then a division by the
            # local power): a handful of pixels of the deepest planes may land just outside the band
            allow = max(2, coef_w.data.size // 200000) if "bilateral" in kw else 0
            if coef_w.data.shape != ref_p.shape or bad.sum() > allow or badr.any():
                fails += 1
                print(f"FAIL {tag}: wow({kw}) {int(bad.sum())} plane / {int(badr.sum())} image pixels beyond tolerance")
                if coef_w.data.shape == ref_p.shape and bad.any():
                    for idx in np.argwhere(bad)[:4]:
                        idx = tuple(idx)
                        print(f"     plane {idx[0]} pixel {idx[1:]}: got {coef_w.data[idx]!r} ref {ref_p[idx]!r}  "
                              f"(plane max {float(np.abs(ref_p[idx[0]]).max()):.3e}, noise got {coef_w.noise!r})")
        # MAD noise estimate + denoise (the first fused pass histograms |w_0|; threshold step between
        # the passes) against np.median / the C oracle's planes
        if H * Wd >= 2:
            cls = W.B3spline if fam == "b3spline" else W.Triangle
            lev = min(level, len(cls(2).sigma_e()) - 1)
            note(f"get_noise / denoise lev={lev}")
            c = W.AtrousTransform(cls)(a, lev)
            n_got = c.get_noise()
            n_ref = np.median(np.abs(ref[0])) / 0.6745 / cls(2).sigma_e()[0]
            planes0 = c.data[0]
            n_own = np.median(np.abs(planes0)) / 0.6745 / cls(2).sigma_e()[0]
            if n_got != n_own or not abs(n_got - n_ref) <= 2e-5 * max(abs(n_ref), 1e-30) + 1e-6:
                fails += 1
                print(f"FAIL {tag}: get_noise {n_got!r} vs own plane {n_own!r} vs oracle {n_ref!r}")
            nsig = int(rng.integers(1, lev + 1))
            sig = [float(x) for x in rng.choice([0., 1., 2., 3., 5.], nsig)]
            soft = bool(rng.integers(0, 2))
            note(f"denoise sigma={sig} soft={soft}")
            got_d = W.denoise(a, sig, cls, soft_threshold=soft)
            cd = O.Coeffs(cref.decompose(a, len(sig), fam), fam)
            cd.denoise(sig, soft_threshold=soft)
            ref_d = np.sum(cd.data, axis=0)
            bad = np.abs(got_d - ref_d) > 3e-5 * max(1.0, float(np.abs(a).max()))
            # a hard threshold may flip for coefficients within rounding of tau
            if bad.sum() > (0 if soft else max(2, a.size // 20000)):
                fails += 1
                print(f"FAIL {tag}: denoise(sigma={sig}, soft={soft}) {int(bad.sum())} pixels beyond tolerance")
        # float64 engine: float64 input is computed in float64 (ref wavelets.py:319-320)
        if case % 2 == 0 and H * Wd <= 600000:
            cls = W.B3spline if fam == "b3spline" else W.Triangle
            lev = min(level, 6)
            a64 = (a.astype(np.float64) * 37.0 + 1e4) if case % 4 else a.astype(np.float64)
            note(f"float64 transform lev={lev}")
            c64 = W.AtrousTransform(cls)(a64, lev)
            r64 = O.atrous_standard(a64, lev, fam)
            e = float(np.abs(c64.data - r64).max())
            n64 = c64.get_noise()
            n_own = np.median(np.abs(c64.data[0])) / 0.6745 / cls(2).sigma_e()[0]
            if c64.data.dtype != np.float64 or not e <= 1e-12 * max(1.0, float(np.abs(a64).max())) or n64 != n_own:
                fails += 1
                print(f"FAIL {tag}: float64 transform max err {e:.3e}, noise {n64!r} vs {n_own!r}")
            sig64 = [5.0, 3.0][:min(2, lev)]
            d64 = W.denoise(a64, sig64, cls)
            e = float(np.abs(d64 - O.denoise(a64.copy(), sig64, fam)).max())
            if not e <= 1e-11 * max(1.0, float(np.abs(a64).max())):
                fails += 1
                print(f"FAIL {tag}: float64 denoise max err {e:.3e}")
        # round 3: float64 fused passes - wt64_decompose_sum (sum carried through the passes) against
        # wt64_plane_sum over the same planes (bitwise) and against the generic kernels / the oracle
        if case % 2 == 1 and H * Wd <= 1500000:
            cls = W.B3spline if fam == "b3spline" else W.Triangle
            lev = min(level, 8)
            a64 = a.astype(np.float64) * 11.0 + 3e3
            note(f"float64 decompose_sum lev={lev}")
            p64 = L.Plan64(ctx, H, Wd, tuple(float(t) for t in cls.coefficients_1d), lev)
            p64.upload(L.PLANE_INPUT, a64)
            fused64 = p64.decompose_sum(L.PLANE_INPUT, lev, L.PLANE_OUT)
            pl = np.stack([p64.download(s_) for s_ in range(lev + 1)])
            car = p64.download(L.PLANE_OUT).copy()
            p64.plane_sum(0, lev + 1, L.PLANE_SCRATCH(7))
            r64 = O.atrous_standard(a64, lev, fam)
            e = float(np.abs(pl - r64).max())
            if not np.array_equal(car, p64.download(L.PLANE_SCRATCH(7))) or not e <= 1e-12 * float(np.abs(a64).max()):
                fails += 1
                print(f"FAIL {tag}: float64 decompose_sum (fused={fused64}) max err {e:.3e} or carried sum != plane sum")
            p64.close()
        # round 3: the pipelined host-to-host call against the serial legs (bitwise), random block sizes
        if case % 7 == 3:
            Hp, Wp = int(rng.integers(1100, 2600)), int(rng.integers(500, 1030)) * 4
            lev = int(rng.integers(2, 7))
            big = rng.standard_normal((Hp, Wp)).astype(np.float32)
            plan = L.Plan(ctx, Hp, Wp, FAM[fam], lev)
            blk = int(rng.choice([0, 128, 192, 256, 320, 512, 1024]))
            note(f"pipelined host call {Hp}x{Wp} L={lev} blk={blk}")
            got_p = plan.decompose_sum_host(big, lev, L.PLANE_OUT, block_rows=blk).copy()
            pl_p = planes(plan, lev + 1)
            L.set_option("host_pipeline", 0)
            ref_p = plan.decompose_sum_host(big, lev, L.PLANE_OUT).copy()
            L.set_option("host_pipeline", 1)
            if not (np.array_equal(got_p, ref_p) and np.array_equal(pl_p, planes(plan, lev + 1))):
                fails += 1
                print(f"FAIL case {case}: pipelined host call {Hp}x{Wp} L={lev} {fam} block_rows={blk} != serial legs")
            plan.close()
        # round 3: the generic tap-list operator - random kernels / pad modes / dimensionalities
        if case % 6 == 2:
            nd = int(rng.integers(1, 4))
            shp = {1: (int(rng.integers(5, 400)),), 2: (int(rng.integers(3, 90)), int(rng.integers(3, 120))),
                   3: (int(rng.integers(2, 9)), int(rng.integers(2, 20)), int(rng.integers(2, 30)))}[nd]
            ker = rng.random(tuple(int(rng.integers(1, 6)) for _ in range(nd)))
            ker /= ker.sum()
            img_g = rng.standard_normal(shp).astype(np.float32)
            mode = str(rng.choice(["symmetric", "reflect", "edge", "wrap", "constant"]))
            s_g = int(rng.integers(0, 3))
            if mode in ("symmetric", "reflect", "wrap") and any((k // 2) * 2 ** s_g > 3 * n for k, n in zip(ker.shape, shp)):
                s_g = 0
            var_g = (np.abs(rng.standard_normal(shp)) + 0.2).astype(np.float32) if rng.integers(0, 2) else None
            from wavelets_amd.wavelets import atrous_convolution
            note(f"generic operator {shp} kernel {ker.shape} {mode} s={s_g}")
            got_g = atrous_convolution(img_g, ker, var_g, s=s_g, mode=mode)
            ref_g = O.atrous_convolution_nd(img_g, ker.astype(np.float32), var_g, s_g, mode)
            e = float(np.abs(got_g - ref_g).max())
            if not e <= (3e-5 if var_g is not None else 3e-6) * max(1.0, float(np.abs(img_g).max())):
                fails += 1
                print(f"FAIL case {case}: atrous_convolution {shp} kernel {ker.shape} mode={mode} s={s_g} "
                      f"var={'yes' if var_g is not None else 'no'} max err {e:.3e}")
        if case % 5 == 0 and H * Wd < 400000 and level <= 6:
            note("recursive")
            r = W.AtrousTransform(W.B3spline if fam == "b3spline" else W.Triangle)(a, level, recursive=True)
            e = float(np.abs(r.data - O.atrous_recursive(a, level, fam)).max())
            if not e <= tol:
                fails += 1
                print(f"FAIL {tag}: recursive max err {e:.3e}")
    print(f"fuzz: {n_cases} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
