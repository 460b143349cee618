"""Multi-GPU a-trous transform: one process per GPU, 1-D row strips, RCCL halo exchange.

The reference has no parallelism of any kind (SURVEY.md sections 2, 5); this module adds the one
strategy the path admits: spatial domain decomposition.  Rank r owns rows
``[row0, row0 + nrows)`` of the global image and all W columns, so halos are contiguous row
blocks: before every pass of the schedule (``_lib.schedule``) a rank exchanges the pass's
cumulative halo rows of the pass INPUT plane with its <= 2 strip neighbours
(ncclGroupStart; ncclSend/ncclRecv x <= 4; ncclGroupEnd on the compute stream, inside
``wt_decompose_pass``).  Borders of the global image are reflected locally.  Global scalars
(MAD median, moments, min/max) are all-reduced inside ``wt_abs_median`` / ``wt_reduce``.

Launcher plumbing (rendezvous, broadcasting the 128-byte RCCL unique id) is NOT done here:
``init_comm`` takes a ``bcast(obj, src)`` callable - bench.py passes a torch.distributed (gloo)
broadcast, a test can pass anything else.
"""
import numpy as np

from . import _lib
from ._lib import PLANE_INPUT, PLANE_OUT, PLANE_NONE, FLAG_FUSED, Plan, Context
import copy

from .wavelets import AtrousTransform, B3spline, Coefficients, _family_of

__all__ = ["partition_rows", "init_comm", "StripTransform"]


def partition_rows(H, nranks):
    """Balanced contiguous row strips: [(row0, nrows)] for ranks 0..nranks-1."""
    if nranks < 1 or H < nranks:
        raise ValueError(f"cannot split {H} rows over {nranks} ranks")
    base, extra = divmod(H, nranks)
    out, row0 = [], 0
    for r in range(nranks):
        n = base + (1 if r < extra else 0)
        out.append((row0, n))
        row0 += n
    return out


def required_halo(family, level, fused=True):
    """Margin rows a strip plan needs for `level` scales (largest pass halo of the schedule)."""
    sched = _lib.schedule(family, level, fused)
    return max([h for _, _, h in sched], default=0)


def init_comm(ctx, rank, nranks, bcast):
    """Create the RCCL communicator of `ctx`.  ``bcast(obj, src)`` must return rank `src`'s
    object on every rank (e.g. a torch.distributed / MPI / file broadcast)."""
    uid = Context.unique_id() if rank == 0 else None
    uid = bcast(uid, 0)
    ctx.comm_init(rank, nranks, uid)
    return ctx


_STRIP_PLANES = {}       # (H, W, level, family, nranks, fused) -> "hipmalloc" | "scattered": what "auto" measured in this process


def _make_strip_plan(ctx, H, W, family, level, row0, nrows, halo, rank, nranks, scattered):
    """a strip plan whose planes >= 8 MiB are mapped over shuffled physical chunks (scattered) or plain hipMalloc"""
    with _lib._pool_lock:                      # (the option is read when the plan is created and stays with the plan)
        keep = _lib._option_values.get("scatter_strips", 0)
        _lib.set_option("scatter_strips", int(bool(scattered)))
        try:
            return Plan(ctx, H, W, family, level, row0=row0, nrows=nrows, halo_rows=halo, rank=rank, nranks=nranks)
        finally:
            _lib.set_option("scatter_strips", keep)


def choose_strip_planes(ctx, H, W, family, level, row0, nrows, halo, rank, nranks, fused=True, steps=3):
    """Placement of a strip plan's planes, MEASURED over the real transport (collective: every rank calls it with the
    same arguments).  Single-GPU plans are ~20 % faster with their planes mapped over shuffled chunks (DESIGN.md
    section 2); whether RCCL's transport moves halo rows in and out of such planes as fast - and correctly - is a
    property of the node, so: `steps` steps of decompose_sum on a synthetic strip with each placement (after one
    warm-up step), the slowest rank's time all-reduced (wt_reduce of a filled plane), and the all-reduced moments of
    the reconstruction and of w_0 compared between the two runs.  "scattered" only if it is correct and at least
    3 % faster.  Cached per (shape, ranks) in the process; WATROO_HIP_STRIP_PLANES=hipmalloc|scattered skips the
    measurement."""
    import os
    env = os.environ.get("WATROO_HIP_STRIP_PLANES", "auto").lower()
    if env in ("hipmalloc", "scattered"):
        return env
    key = (H, W, level, family, nranks, bool(fused))
    if key in _STRIP_PLANES:
        return _STRIP_PLANES[key]
    plane_bytes = (nrows + 2 * halo) * ((W + 3) // 4 * 4) * 4
    if nranks == 1 or plane_bytes < (8 << 20) or isinstance(family, tuple):
        _STRIP_PLANES[key] = "hipmalloc"       # (planes below 8 MiB are never mapped; nothing to choose)
        return "hipmalloc"
    strip = (np.arange(row0, row0 + nrows, dtype=np.float32)[:, None] * np.float32(0.25)
             + (np.arange(W, dtype=np.float32) % 7)[None, :]).astype(np.float32)
    flags = FLAG_FUSED if fused else 0
    seen = {}
    for name in ("hipmalloc", "scattered"):
        try:
            plan = _make_strip_plan(ctx, H, W, family, level, row0, nrows, halo, rank, nranks, name == "scattered")
            plan.upload(PLANE_INPUT, strip)
            plan.decompose_sum(PLANE_INPUT, level, PLANE_OUT, flags)
            ctx.sync()
            ctx.timer_start()
            for _ in range(steps):
                plan.decompose_sum(PLANE_INPUT, level, PLANE_OUT, flags)
            ms = ctx.timer_stop() / steps
            moments = plan.reduce(PLANE_OUT) + plan.reduce(0)
            plan.fill(_lib.PLANE_SCRATCH(0), ms)
            slowest = plan.reduce(_lib.PLANE_SCRATCH(0))[3]
            plan.close()
            seen[name] = (slowest, moments)
        except _lib.WatrooHipError:
            if name == "hipmalloc":
                raise
            seen[name] = None                  # mapped planes unavailable here: hipMalloc it is
    best = "hipmalloc"
    if seen.get("scattered") is not None and seen["scattered"][1] == seen["hipmalloc"][1] \
            and seen["scattered"][0] < 0.97 * seen["hipmalloc"][0]:
        best = "scattered"
    _STRIP_PLANES[key] = best
    _STRIP_PLANES[key + ("measured",)] = {k: (v[0] if v else None) for k, v in seen.items()}
    return best


class StripTransform:
    """The a-trous transform of one row strip of a global H x W image on this rank's GPU.

    Mirrors ``AtrousTransform`` + ``Coefficients`` + ``utils.wow`` for the sharded case:
    ``decompose`` -> planes in HBM, ``get_noise``/``denoise`` (global MAD via all-reduced
    histograms), ``sum`` -> this rank's rows of the reconstruction, ``denoise_sum`` and ``wow``
    (every dilated operator - the bilateral transform's variance and
    range-weighted convolution, the local power conv_s(c^2) - exchanges hw * 2**s rows of ITS OWN
    input plane with the strip neighbours first; np.std / np.mean / min / max become all-reduced
    fp64 moments inside wt_reduce)."""

    def __init__(self, ctx, H, W, level, scaling_function_class=B3spline, rank=None,
                 nranks=None, fused=True, planes="auto"):
        self.ctx = ctx
        self.rank = ctx.rank if rank is None else rank
        self.nranks = ctx.nranks if nranks is None else nranks
        self.level = level
        self.scaling_function = scaling_function_class(2)
        self.family = _family_of(self.scaling_function)
        self.fused = fused
        self.row0, self.nrows = partition_rows(H, self.nranks)[self.rank]
        halo = required_halo(self.family, level, fused) if self.nranks > 1 else 0
        if self.nranks > 1 and halo > min(n for _, n in partition_rows(H, self.nranks)):
            raise ValueError(f"strips of {H // self.nranks} rows are thinner than the "
                             f"{halo}-row halo of {level} scales: use fewer ranks")
        # planes: "hipmalloc", "scattered" (mapped over shuffled physical chunks, as single-GPU plans are) or "auto":
        # measured once per (shape, ranks) over the real transport - choose_strip_planes; collective
        if planes == "auto":
            planes = choose_strip_planes(ctx, H, W, self.family, level, self.row0, self.nrows, halo, self.rank,
                                         self.nranks, fused)
        if planes not in ("hipmalloc", "scattered"):
            raise ValueError("planes must be 'auto', 'hipmalloc' or 'scattered'")
        self.planes = planes
        self.plan = _make_strip_plan(ctx, H, W, self.family, level, self.row0, self.nrows, halo, self.rank,
                                     self.nranks, planes == "scattered")
        self.noise = None

    def upload(self, strip):
        self.plan.upload(PLANE_INPUT, strip)

    def decompose(self):
        self.plan.decompose(PLANE_INPUT, self.level, FLAG_FUSED if self.fused else 0)

    def decompose_sum(self, out=None):
        """Transform and reconstruction of this strip in the same passes (wt_decompose_sum)."""
        self.plan.decompose_sum(PLANE_INPUT, self.level, PLANE_OUT,
                                FLAG_FUSED if self.fused else 0)
        return self.plan.download(PLANE_OUT, out)

    @property
    def sigma_e(self):
        return self.scaling_function.sigma_e()

    def get_noise(self):
        return self.plan.abs_median(0) / 0.6745 / self.sigma_e[0]

    def denoise(self, sigma, weights=None, soft_threshold=True):
        if weights is None:
            weights = (1,) * len(sigma)
        for scl, (sig, wgt) in enumerate(zip(sigma, weights)):
            if scl > self.level:
                break
            if sig != 0:
                if self.noise is None:
                    self.noise = self.get_noise()
                if self.noise != 0:
                    self.plan.denoise(scl, sig * self.noise * self.sigma_e[scl], wgt,
                                      soft_threshold, PLANE_NONE)
                    continue
            if wgt != 1:
                self.plan.wow_update(scl, PLANE_NONE, 0.0, True, PLANE_NONE, wgt, PLANE_NONE)

    def sum(self, out=None):
        self.plan.plane_sum(0, self.level + 1, PLANE_OUT)
        return self.plan.download(PLANE_OUT, out)

    def plane(self, s):
        return self.plan.download(s)

    # -- sharded applications (watroo/utils.py:83-102, 105-219) -------------------------------
    def _coefficients(self, bilateral=None):
        """A Coefficients view of this strip's plan (global scalars are all-reduced by the
        library); the view never owns the plan."""
        c = Coefficients(self.plan, self.scaling_function, bilateral)
        c.noise = self.noise
        return c

    def denoise_sum(self, sigma, weights=None, soft_threshold=True, write_back=True, out=None):
        """Coefficients.denoise(sigma, weights) fused with the plane sum (wt_denoise_sum); returns
        this rank's rows of the sum."""
        c = self._coefficients()
        try:
            c._denoise_sum(sigma, weights, soft_threshold, write_back)
            self.noise = c.noise
        finally:
            c._plan = None
        return self.plan.download(PLANE_OUT, out)

    def wow(self, weights=[], whitening=True, denoise_coefficients=[], noise=None, bilateral=None,
            bilateral_scaling=False, soft_threshold=True, preserve_variance=False, gamma=3.2,
            gamma_min=None, gamma_max=None, h=0, out=None):
        """utils.wow (watroo/utils.py:105-219) of the strip uploaded with ``upload``: the number
        of scales is this transform's ``level`` (the reference derives it from the image size,
        ref:122; sharded, every scale's halo hw * 2**s must fit a strip).  Returns this rank's
        rows of the enhanced image; the whitened planes stay on the plan (``plane(s)``)."""
        from .utils import _wow_device
        n_scales = self.level
        if bilateral is None:                                             # ref:140-146
            sigma_bilateral = None
        else:
            sigma_bilateral = copy.copy(bilateral) if type(bilateral) is list \
                else [bilateral, ] * (n_scales + 1)
            if len(sigma_bilateral) <= n_scales:
                sigma_bilateral.extend([1, ] * (n_scales - len(sigma_bilateral) + 1))
        transform = AtrousTransform(type(self.scaling_function), bilateral=sigma_bilateral,
                                    bilateral_scaling=bilateral_scaling)
        transform._run(self.plan, n_scales, flags=FLAG_FUSED if self.fused else 0)   # ref:148-151
        c = self._coefficients(sigma_bilateral)
        c.noise = noise
        try:
            _wow_device(c, n_scales, weights, whitening, denoise_coefficients, soft_threshold,
                        preserve_variance, gamma, gamma_min, gamma_max, h)
            self.noise = c.noise
        finally:
            c._plan = None
        return self.plan.download(PLANE_OUT, out)
