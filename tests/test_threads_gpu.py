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
