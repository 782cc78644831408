"""Handles are independent: two extractors / matchers used from two threads at once (the reference runs the left
and right extractor on two std::threads, src/Frame.cc:78-81) give the single-threaded results."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_extractors_two_threads(orbx, synth):
    imgs = [synth.texture(40 + i, 640, 480) for i in range(2)]
    ref = []
    for im in imgs:
        ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
        ref.append(ex(im))
        ex.close()
    exs = [orbx.ORBextractor(1000, max_width=640, max_height=480) for _ in range(2)]
    errors = []

    def work(t):
        try:
            for _ in range(25):
                k, d = exs[t](imgs[t])
                if k.tobytes() != ref[t][0].tobytes() or not np.array_equal(d, ref[t][1]):
                    errors.append("thread %d: result differs" % t)
                    return
        except Exception as e:          # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_fresh_handles_in_two_threads(orbx, synth):
    """Creation, the table upload of the first call and the graph capture of the second call of one handle while ANOTHER thread is in the
    same phases of its own handle: every synchronous runtime call of one thread meets a stream capture of the other sooner or later
    (on this runtime such a meeting invalidated the capture and failed the copy; the library serialises those phases and runs a call
    plainly when its capture is lost).  Twelve rounds of create / four calls / close per thread."""
    imgs = [synth.texture(60 + i, 640, 480) for i in range(2)]
    ref = []
    for im in imgs:
        ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
        ref.append(ex(im))
        ex.close()
    errors = []
    gate = threading.Barrier(2)

    def work(t):
        try:
            for _ in range(12):
                gate.wait(timeout=60)
                ex = orbx.ORBextractor(1000, max_width=640, max_height=480)
                for _ in range(4):      # plain, capture, two replays
                    k, d = ex(imgs[t])
                    if k.tobytes() != ref[t][0].tobytes() or not np.array_equal(d, ref[t][1]):
                        errors.append("thread %d: result differs" % t)
                ex.close()
        except Exception as e:          # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))
            gate.abort()

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_two_matchers_two_threads(orbx):
    rng = np.random.default_rng(0)
    sets = [(rng.integers(0, 256, (1500, 32), dtype=np.uint8), rng.integers(0, 256, (1400, 32), dtype=np.uint8)) for _ in range(2)]
    ms = [orbx.ORBmatcher() for _ in range(2)]
    ref = [ms[i].best2(*sets[i]) for i in range(2)]
    errors = []

    def work(t):
        for _ in range(25):
            r = ms[t].best2(*sets[t])
            if not all(np.array_equal(a, b) for a, b in zip(r, ref[t])):
                errors.append(t)
                return

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors
