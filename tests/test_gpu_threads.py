"""SURVEY 8(b) threading clause: "one in-flight search per handle (internal mutex), independent handles are independent".
Generation pipelines call retrieve() from several event loops / worker threads (embedding models hop to threads through
asyncio.to_thread, colpali.py:144), so one handle must tolerate concurrent callers and two handles must not disturb each
other.  Every answer is compared with the single-threaded answer bit for bit."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_threads(fns):
    errs = []

    def wrap(f):
        try:
            f()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_one_handle_many_threads_and_two_handles(native_built):
    import autorag_research_amd as pkg

    rng = np.random.default_rng(8)
    d = 256
    C1 = rng.standard_normal((30000, d)).astype(np.float32)
    C2 = rng.standard_normal((20000, d)).astype(np.float32)
    Qs = [rng.standard_normal((b, d)).astype(np.float32) for b in (1, 7, 130, 300, 64, 9)]
    a, b = pkg.Mi355Index(d), pkg.Mi355Index(d, "ip")
    a.add(C1)
    b.add(C2)
    want_a = [a.search(q, 10) for q in Qs]
    want_b = [b.search(q, 5) for q in Qs]

    def hammer(ix, want, k, order):
        def f():
            for rep in range(3):
                for i in order:
                    dist, rows = ix.search(Qs[i], k)
                    assert np.array_equal(rows, want[i][1]) and np.array_equal(dist.view(np.uint64), want[i][0].view(np.uint64))
        return f

    # four threads on handle a (different block shapes interleave: the per-handle mutex serialises them), two on handle b
    _run_threads([hammer(a, want_a, 10, [0, 1, 2, 3, 4, 5]), hammer(a, want_a, 10, [5, 4, 3, 2, 1, 0]),
                  hammer(a, want_a, 10, [2, 0, 3, 1, 5, 4]), hammer(a, want_a, 10, [3, 3, 0, 0, 2, 2]),
                  hammer(b, want_b, 5, [0, 1, 2, 3, 4, 5]), hammer(b, want_b, 5, [4, 2, 0, 5, 3, 1])])
    # searches racing an append on the same handle: each answer is the top-k of the corpus before OR after the append
    extra = rng.standard_normal((5000, d)).astype(np.float32)
    before = a.search(Qs[2], 10)
    results = []

    def searcher():
        for _ in range(6):
            results.append(a.search(Qs[2], 10))

    _run_threads([searcher, lambda: a.add(extra), searcher])
    after = a.search(Qs[2], 10)
    for dist, rows in results:
        ok_before = np.array_equal(rows, before[1]) and np.array_equal(dist.view(np.uint64), before[0].view(np.uint64))
        ok_after = np.array_equal(rows, after[1]) and np.array_equal(dist.view(np.uint64), after[0].view(np.uint64))
        assert ok_before or ok_after
    a.close()
    b.close()


def test_maxsim_and_single_vector_calls_share_a_handle(native_built):
    import autorag_research_amd as pkg

    rng = np.random.default_rng(12)
    d = 64
    ix = pkg.Mi355Index(d)
    ix.add(rng.standard_normal((8000, d)).astype(np.float32))
    lens = rng.integers(1, 40, size=500)
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ix.add_multivec(tok, off)
    q = rng.standard_normal((40, d)).astype(np.float32)
    qoff = np.array([0, 10, 25, 40], dtype=np.int32)
    want_s, want_m = ix.search(q, 10), ix.search_maxsim(q, qoff, 5)

    def single():
        for _ in range(10):
            dd, rr = ix.search(q, 10)
            assert np.array_equal(rr, want_s[1]) and np.array_equal(dd.view(np.uint64), want_s[0].view(np.uint64))

    def multi():
        for _ in range(10):
            dd, rr = ix.search_maxsim(q, qoff, 5)
            assert np.array_equal(rr, want_m[1]) and np.array_equal(dd.view(np.uint32), want_m[0].view(np.uint32))

    _run_threads([single, multi, single, multi])
    ix.close()
