"""Host logic of wavelets_amd.sequence.map_frames without a GPU: the lanes are threads over a bounded queue; contexts are
stand-ins (the real ones need a device - tests/test_gpu_round6.py runs the same calls on lanes with real contexts)."""
import threading
import time

import numpy as np
import pytest

from wavelets_amd import _lib, sequence


class FakeCtx:
    _h = 1


@pytest.fixture()
def fake_lanes(monkeypatch):
    made, by_dev = [], {}

    def lane_contexts(n, device=None):
        lst = by_dev.setdefault(device, [])
        while len(lst) < n:
            c = FakeCtx()
            c.device = device
            lst.append(c)
            made.append(c)
        return lst[:n]
    monkeypatch.setattr(_lib, "lane_contexts", lane_contexts)
    monkeypatch.setattr(_lib, "device_count", lambda: 4)
    return made


def test_results_come_back_in_input_order_and_each_lane_sees_its_own_context(fake_lanes):
    seen = {}

    def fn(x):
        time.sleep(0.002 * (7 - x % 7))                      # later frames finish earlier
        seen.setdefault(id(_lib.default_context()), set()).add(threading.current_thread().name)
        return x * x
    got = sequence.map_frames(fn, range(40), lanes=4)
    assert got == [x * x for x in range(40)]
    assert len(seen) == 4 and all(len(t) == 1 for t in seen.values())        # one context per lane, one thread per context
    assert set(seen) == {id(c) for c in fake_lanes[:4]}
    assert getattr(_lib._tls, "ctx", None) is None                           # the caller's thread is untouched


def test_a_generator_is_not_run_ahead_of_the_lanes(fake_lanes):
    produced, consumed = [], []

    def frames():
        for i in range(30):
            produced.append(i)
            yield i

    def fn(x):
        time.sleep(0.003)
        consumed.append(x)
        assert len(produced) - len(consumed) <= 2 * 3 + 1   # at most `lanes` queued + `lanes` in flight (+ the one being put)
        return x
    assert sequence.map_frames(fn, frames(), lanes=3) == list(range(30))


def test_out_target_one_lane_and_exceptions(fake_lanes):
    out = np.zeros((10, 4))
    assert sequence.map_frames(lambda i: np.full(4, i, float), range(10), lanes=3, out=out) is out
    np.testing.assert_array_equal(out[:, 0], np.arange(10))
    assert sequence.map_frames(lambda x: x + 1, [1, 2, 3], lanes=1) == [2, 3, 4]       # the plain loop, caller's thread
    with pytest.raises(ValueError, match="lanes"):
        sequence.map_frames(lambda x: x, [1], lanes=0)

    def bad(x):
        if x == 5:
            raise KeyError("frame 5")
        return x
    with pytest.raises(KeyError, match="frame 5"):
        sequence.map_frames(bad, range(100), lanes=3)
    assert threading.active_count() < 10                     # the lanes have stopped


def test_use_context_nests_and_restores():
    a, b = FakeCtx(), FakeCtx()
    with _lib.use_context(a):
        assert _lib.default_context() is a
        with _lib.use_context(b):
            assert _lib.default_context() is b
        assert _lib.default_context() is a
    assert getattr(_lib._tls, "ctx", None) is None


def test_lanes_on_several_devices(fake_lanes):
    """devices=[...] / "all": `lanes` lanes on each GPU, frames dealt to whichever lane is free, order kept"""
    used = set()

    def fn(x):
        time.sleep(0.002)
        used.add(_lib.default_context().device)
        return -x
    assert sequence.map_frames(fn, range(60), lanes=2, devices=[2, 0, 2]) == [-x for x in range(60)]
    assert used == {0, 2} and len(fake_lanes) == 4
    used.clear()
    assert sequence.map_frames(fn, range(60), lanes=1, devices="all") == [-x for x in range(60)]
    assert used == {0, 1, 2, 3}
    with pytest.raises(ValueError, match="no GPU"):
        sequence.map_frames(fn, [1], devices=[])
