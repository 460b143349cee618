"""Sequences of frames and channels: the package's numpy-to-numpy calls over an iterable of images, double-buffered.

The reference processes image sequences and the channels of a colour image one after the other
(/root/reference/watroo/utils.py:60-78 is the per-channel loop of ``enhance``; ``denoise`` / ``wow`` are called
per frame by its users, ref utils.py:83-102, 105-219).  On this engine a numpy-to-numpy call at 8192^2 is two PCIe
legs (4.7 ms up, 4.8 ms down) around ~1.3 ms of GPU work, so a loop of such calls leaves the GPU and one direction of
the full-duplex link idle most of the time.  ``map_frames`` runs the SAME per-frame call on a few worker lanes - a
host thread with a context (HIP stream, scratch, pooled plan) of its own each - so that the upload of frame i+1, the
passes of frame i and the download of frame i-1 overlap.  Nothing about a frame's computation changes: results are
bit-identical to the per-call API, in the order of the input.

    from wavelets_amd import denoise_many, wow_many, transform_many
    clean = denoise_many(frames, [5, 3])                  # list of arrays, frames[i] -> clean[i]
"""
import queue
import threading

import numpy as np

from . import _lib

__all__ = ['map_frames', 'denoise_many', 'wow_many', 'transform_many']

DEFAULT_LANES = 3        # one frame uploading, one in the passes, one downloading


def map_frames(fn, frames, lanes=None, out=None, device=None, devices=None):
    """``[fn(frame) for frame in frames]`` with the calls spread over ``lanes`` worker lanes (threads; each runs
    its calls on a context of its own, `_lib.use_context`), results in input order.  ``frames`` may be any iterable
    (a generator is consumed as lanes become free: at most ``lanes`` frames are in flight).  ``out``: an optional
    sequence / array that receives ``out[i] = fn(frames[i])`` instead of the returned list.  The first exception
    of any lane is re-raised after the lanes have stopped.
    ``devices``: a list of HIP ordinals, or "all" - ``lanes`` lanes on EACH of these GPUs, frames dealt to whichever
    lane is free (independent frames need no exchange between GPUs: the replica form of multi-GPU work, one PCIe
    link per GPU); default: the one device of ``device`` / WATROO_HIP_DEVICE."""
    lanes = DEFAULT_LANES if lanes is None else int(lanes)
    if lanes < 1:
        raise ValueError("lanes must be >= 1")
    if devices is not None:
        devs = list(range(_lib.device_count())) if isinstance(devices, str) and devices == "all" else sorted({int(d) for d in devices})
        if not devs:
            raise ValueError("devices: no GPU given")
    else:
        devs = None
    if lanes == 1 and devs is None:                    # the plain loop (on the caller's thread and context)
        res = []
        for i, f in enumerate(frames):
            r = fn(f)
            if out is not None:
                out[i] = r
            else:
                res.append(r)
        return out if out is not None else res
    ctxs = _lib.lane_contexts(lanes, device) if devs is None else [c for d in devs for c in _lib.lane_contexts(lanes, d)]
    todo = queue.Queue(maxsize=len(ctxs))              # (index, frame): bounded, so a generator is not run ahead
    results, errors = {}, []
    lock = threading.Lock()

    def lane(ctx):
        with _lib.use_context(ctx):
            while True:
                item = todo.get()
                if item is None:
                    return
                if errors:
                    continue                           # drain: another lane failed
                i, f = item
                try:
                    r = fn(f)
                    if out is not None:
                        out[i] = r
                    else:
                        with lock:
                            results[i] = r
                except BaseException as e:             # noqa: BLE001 - re-raised on the caller's thread
                    with lock:
                        errors.append(e)

    threads = [threading.Thread(target=lane, args=(c,), name=f"wavelets_amd-lane{k}", daemon=True) for k, c in enumerate(ctxs)]
    for t in threads:
        t.start()
    n = 0
    try:
        for i, f in enumerate(frames):
            if errors:
                break
            todo.put((i, f))
            n = i + 1
    finally:
        for _ in threads:
            todo.put(None)
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return out if out is not None else [results[i] for i in range(n)]


def denoise_many(frames, weights, scaling_function=None, noise=None, bilateral=None, soft_threshold=True,
                 anscombe=False, lanes=None, out=None, devices=None):
    """``[denoise(f, weights, ...) for f in frames]`` (ref utils.py:83-102 per frame), double-buffered over PCIe.
    ``noise``: None (each frame's own MAD estimate), a scalar / map shared by all frames, or a list with one entry
    per frame."""
    from .utils import denoise
    from .wavelets import B3spline
    sf = B3spline if scaling_function is None else scaling_function
    per_frame = isinstance(noise, (list, tuple))

    def one(item):
        i, f = item
        # (a preallocated array target receives the download directly where its rows allow it - no host copy)
        tgt = out[i] if isinstance(out, np.ndarray) and out.ndim == 3 else None
        res = denoise(f, list(weights), sf, noise[i] if per_frame else noise, bilateral, soft_threshold, anscombe, _out=tgt)
        return None if (tgt is not None and res is tgt) else res

    if isinstance(out, np.ndarray) and out.ndim == 3:
        class _Skip:                               # map_frames' `out[i] = r` for frames that are already in place
            def __setitem__(self, i, r):
                if r is not None:
                    out[i] = r
        map_frames(one, enumerate(frames), lanes, _Skip(), devices=devices)
        return out
    return map_frames(one, enumerate(frames), lanes, out, devices=devices)


def wow_many(frames, lanes=None, out=None, devices=None, **kwargs):
    """``[wow(f, **kwargs) for f in frames]`` (ref utils.py:105-219 per frame), double-buffered over PCIe; every
    element is what ``wow`` returns for that frame (the image, or ``(image, coefficients)``)."""
    from .utils import wow
    return map_frames(lambda f: wow(f, **kwargs), frames, lanes, out, devices=devices)


def transform_many(frames, level, scaling_function=None, lanes=None, devices=None, **kwargs):
    """``[AtrousTransform(scaling_function, **kwargs)(f, level) for f in frames]`` (ref wavelets.py:290-328 per
    frame): a list of ``Coefficients`` whose planes stay on the device (each on the lane context that made it)."""
    from .wavelets import AtrousTransform, B3spline
    tr = AtrousTransform(B3spline if scaling_function is None else scaling_function, **kwargs)
    return map_frames(lambda f: tr(f, level), frames, lanes, devices=devices)
