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
                 nranks=None, fused=True):
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
        self.plan = Plan(ctx, H, W, self.family, level, row0=self.row0, nrows=self.nrows,
                         halo_rows=halo, rank=self.rank, nranks=self.nranks)
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
