"""GPU parity tests proper: search through the C ABI vs the CPU oracle -- ids and float8 distances bit-exact."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(native_built):
    import autorag_research_amd as p

    return p


def _check(idx, oracle, C, Q, k, metric="cosine"):
    dist, rows = idx.search(Q, k)
    rd, rr = oracle.topk_search(C, Q, k, metric=metric)
    assert np.array_equal(rows, rr)
    assert np.array_equal(np.isnan(dist), np.isnan(rd))  # NaN payload/sign is not part of the contract
    ok = ~np.isnan(dist)
    assert np.array_equal(dist[ok].view(np.uint64), rd[ok].view(np.uint64))
    return dist, rows


@pytest.mark.parametrize("path", ["scan", "screen", "auto"])
@pytest.mark.parametrize("n,d,B,k", [(5183, 384, 33, 10), (3000, 768, 1, 10), (2500, 768, 130, 100), (999, 100, 7, 5)])
def test_search_matches_oracle(pkg, oracle, path, n, d, B, k):
    rng = np.random.default_rng(n + d + B)
    C = rng.standard_normal((n, d)).astype(np.float32)
    C *= rng.uniform(0.05, 20.0, size=(n, 1)).astype(np.float32)  # un-normalised corpus
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.set_option("path", path)
        idx.add(C[: n // 2])
        idx.add(C[n // 2:])  # appended in two batches
        assert len(idx) == n
        _check(idx, oracle, C, Q, k)


def test_k_larger_than_n_and_empty(pkg, oracle):
    rng = np.random.default_rng(5)
    C = rng.standard_normal((6, 32)).astype(np.float32)
    Q = rng.standard_normal((3, 32)).astype(np.float32)
    with pkg.Mi355Index(32) as idx:
        d0, r0 = idx.search(Q, 4)  # empty index
        assert (r0 == -1).all() and np.isnan(d0).all()
        idx.add(C)
        dist, rows = _check(idx, oracle, C, Q, 10)
        assert (rows[:, 6:] == -1).all() and np.isnan(dist[:, 6:]).all()


def test_ties_duplicates_and_zero_rows(pkg, oracle):
    """duplicate rows (exact ties -> lower row first), zero-norm rows (NaN, last), zero query (all NaN)."""
    rng = np.random.default_rng(11)
    base = rng.standard_normal((40, 64)).astype(np.float32)
    C = np.concatenate([base, base, base[:10] * 2.0, np.zeros((3, 64), np.float32), base[::-1]])
    Q = np.concatenate([base[:5] + 0.01 * rng.standard_normal((5, 64)).astype(np.float32),
                        np.zeros((1, 64), np.float32)])
    for path in ("screen", "scan"):
        with pkg.Mi355Index(64) as idx:
            idx.set_option("path", path)
            idx.add(C)
            for k in (1, 4, 50, len(C)):
                _check(idx, oracle, C, Q, k)


def test_candidate_overflow_falls_back_exactly(pkg, oracle):
    """a tiny candidate buffer forces the overflow -> exact-scan fallback; results must not change."""
    rng = np.random.default_rng(3)
    n, d = 6000, 128
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((20, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        idx.set_option("cand_cap", 16)
        _check(idx, oracle, C, Q, 10)
        assert idx.stat("fallback_queries") > 0


def test_adversarial_order_ascending_similarity(pkg, oracle):
    """rows sorted by ascending similarity to the query: every row beats the running threshold."""
    rng = np.random.default_rng(8)
    n, d = 20000, 64
    q = rng.standard_normal(d).astype(np.float32)
    C = rng.standard_normal((n, d)).astype(np.float32)
    sims = (C @ q) / np.linalg.norm(C, axis=1)
    C = C[np.argsort(sims)]
    Q = np.stack([q, -q, rng.standard_normal(d).astype(np.float32)])
    for path in ("screen", "scan"):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("path", path)
            idx.add(C)
            _check(idx, oracle, C, Q, 10)


def test_inner_product_metric(pkg, oracle):
    rng = np.random.default_rng(21)
    C = rng.standard_normal((3000, 96)).astype(np.float32)
    Q = rng.standard_normal((9, 96)).astype(np.float32)
    with pkg.Mi355Index(96, "ip") as idx:
        idx.add(C)
        _check(idx, oracle, C, Q, 10, metric="ip")


def test_near_ties_stress(pkg, oracle):
    """many rows within a few ulp of each other: ranking must still equal the oracle's bit for bit."""
    rng = np.random.default_rng(99)
    d = 256
    c0 = rng.standard_normal(d).astype(np.float32)
    C = np.tile(c0, (4000, 1))
    C += (rng.standard_normal(C.shape) * 1e-6).astype(np.float32)
    Q = (c0[None, :] + 1e-3 * rng.standard_normal((4, d))).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        _check(idx, oracle, C, Q, 25)
