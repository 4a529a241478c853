"""Several handles in one process: on different devices (vsg_options.device; skipped on a one-GPU
box) and from concurrent host threads on one device.  Kernel attributes that belong to a (function,
device) pair -- the bilateral filter's dynamic LDS size -- and the per-handle mailbox / counters must
not leak from one handle to another."""
import threading

import pytest

from test_gpu_parity import run_streams, vsg  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_handles_on_two_devices_in_one_process(vsg):
    from video_segment_amd import _lib
    if _lib.lib().vsg_device_count() < 2:
        pytest.skip("one HIP device")
    # device 1 first: the process has not launched the 69 KiB-LDS bilateral kernel anywhere yet
    for dev in (1, 0, 1):
        run_streams(vsg, 96, 64, 26, "smooth", True, 10, device=dev)
    errors = []

    def work(dev):
        try:
            run_streams(vsg, 128, 96, 30, "bench", True, 10, device=dev)
        except BaseException as e:   # noqa: BLE001
            errors.append((dev, e))
    ts = [threading.Thread(target=work, args=(d,)) for d in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_concurrent_handles_on_one_device(vsg):
    """Four streams from four host threads on device 0 (what bench.py --streams does), each compared
    with its own oracle stream."""
    errors = []

    def work(i):
        try:
            run_streams(vsg, 96 + 16 * i, 64 + 8 * i, 24 + i, ("bench", "smooth", "noise", "probe")[i], True, 8 + i,
                        seed=20 + i)
        except BaseException as e:   # noqa: BLE001
            errors.append((i, e))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("mode", [None, "1", "2"])
def test_more_streams_than_cores(vsg, monkeypatch, mode):
    """A stream's host thread waits for the device's scalars about 150 times per chunk (the mailbox,
    DESIGN 4.8).  Spinning is the fastest way to wait while every stream has a core of its own; six
    streams on two cores (sched_setaffinity) would starve each other -- and the oracle threads
    beside them -- if they only ever spun.  Unset, VSG_MAIL_YIELD yields by itself once the process
    holds more graphs than half its cores; 1 and 2 force the yielding and the sleeping wait.  Same
    bytes, and the test finishes (the box has a hard timeout on hangs)."""
    import os
    import time
    if mode is None:
        monkeypatch.delenv("VSG_MAIL_YIELD", raising=False)
    else:
        monkeypatch.setenv("VSG_MAIL_YIELD", mode)
    old = os.sched_getaffinity(0)
    cores = sorted(old)[:2]
    errors = []

    def work(i):
        try:
            run_streams(vsg, 96 + 16 * (i % 3), 64 + 8 * (i % 3), 22 + i, ("bench", "smooth", "noise")[i % 3], True,
                        8 + (i % 3), seed=40 + i)
        except BaseException as e:   # noqa: BLE001
            errors.append((i, e))
    t0 = time.time()
    try:
        os.sched_setaffinity(0, cores)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        os.sched_setaffinity(0, old)
    assert not errors, errors
    assert time.time() - t0 < 240
