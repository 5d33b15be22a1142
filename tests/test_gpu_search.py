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


SCREENS = ["bf16", "i8"]


@pytest.mark.parametrize("path", ["scan", "screen:bf16", "screen:i8", "auto"])
@pytest.mark.parametrize("n,d,B,k", [(5183, 384, 33, 10), (3000, 768, 1, 10), (2500, 768, 130, 100), (999, 100, 7, 5)])
def test_search_matches_oracle(pkg, oracle, path, n, d, B, k):
    rng = np.random.default_rng(n + d + B)
    C = rng.standard_normal((n, d)).astype(np.float32)
    C *= rng.uniform(0.05, 20.0, size=(n, 1)).astype(np.float32)  # un-normalised corpus
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        if ":" in path:
            path, screen = path.split(":")
            idx.set_option("screen_dtype", screen)
        idx.set_option("path", path)
        idx.add(C[: n // 2])
        idx.add(C[n // 2:])  # appended in two batches
        assert len(idx) == n
        _check(idx, oracle, C, Q, k)


@pytest.mark.parametrize("path", ["scan", "screen:bf16", "screen:i8", "auto"])
@pytest.mark.parametrize("metric", ["cosine", "ip"])
@pytest.mark.parametrize("n,d,B,k", [(9000, 1024, 40, 10), (6000, 1536, 300, 10), (5000, 2048, 1, 10), (4000, 3072, 130, 20),
                                     (3000, 4096, 64, 10), (3000, 1500, 17, 5)])
def test_wide_embeddings(pkg, oracle, path, metric, n, d, B, k):
    """embedding sizes above the headline's 768 (e5-large 1024, OpenAI 1536 / 3072, 2048 / 4096-wide encoders, and one that is
    not a multiple of 32): the reference's `VECTOR(dim)` column takes any `embedding_dim` (orm/schema_factory.py:31)"""
    rng = np.random.default_rng(n + d + B)
    C = rng.standard_normal((n, d)).astype(np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    if metric == "ip":
        C *= rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d, metric) as idx:
        if ":" in path:
            path, screen = path.split(":")
            idx.set_option("screen_dtype", screen)
        idx.set_option("path", path)
        idx.add(C)
        rd, rr = oracle.topk_search(C, Q, k, metric=metric)
        dist, rows = idx.search(Q, k)
        assert np.array_equal(rows, rr)
        assert np.array_equal(dist.view(np.uint64), rd.view(np.uint64))


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
    for path, screen in (("screen", "bf16"), ("screen", "i8"), ("scan", "auto")):
        with pkg.Mi355Index(64) as idx:
            idx.set_option("path", path)
            idx.set_option("screen_dtype", screen)
            idx.add(C)
            for k in (1, 4, 50, len(C)):
                _check(idx, oracle, C, Q, k)


def test_candidate_overflow_falls_back_exactly(pkg, oracle):
    """a tiny candidate buffer forces the overflow -> exact-scan fallback; results must not change."""
    rng = np.random.default_rng(3)
    n, d = 6000, 128
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((20, d)).astype(np.float32)
    for screen in SCREENS:
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_dtype", screen)
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
    for path, screen in (("screen", "bf16"), ("screen", "i8"), ("scan", "auto")):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("path", path)
            idx.set_option("screen_dtype", screen)
            idx.add(C)
            _check(idx, oracle, C, Q, 10)


def test_inner_product_metric(pkg, oracle):
    rng = np.random.default_rng(21)
    C = rng.standard_normal((3000, 96)).astype(np.float32)
    Q = rng.standard_normal((9, 96)).astype(np.float32)
    with pkg.Mi355Index(96, "ip") as idx:
        idx.add(C)
        _check(idx, oracle, C, Q, 10, metric="ip")


@pytest.mark.parametrize("screen", SCREENS + ["auto"])
def test_inner_product_screens_without_row_norms_in_the_threshold(pkg, oracle, screen):
    """round 3: the inner-product shadows hold the rows themselves, so the threshold is dot_k / |q| - E whatever the norms and
    whatever the SIGN of the k-th best dot (round 2 needed a positive one and the largest row norm): norms spread over a
    decade, a block of rows whose dots are all negative, k up to the general prune's range, 300 queries -- the oracle's
    ids and distances bit for bit, nobody on the exact-scan path."""
    rng = np.random.default_rng(5)
    n, d = 50_000, 128
    C = rng.standard_normal((n, d)).astype(np.float32)
    C *= np.exp(rng.uniform(-1.2, 1.2, size=(n, 1))).astype(np.float32)
    Q = rng.standard_normal((300, d)).astype(np.float32)
    Cneg = (-np.abs(C[:6000]) - 0.1).astype(np.float32)   # every dot with a non-negative query is negative
    Qpos = np.abs(Q[:40])
    for k in (10, 100, 600):
        with pkg.Mi355Index(d, "ip") as idx:
            idx.set_option("screen_dtype", screen)
            idx.add(C)
            idx.reset_stats()
            _check(idx, oracle, C, Q, k, metric="ip")
            assert idx.stat("fallback_queries") == 0
    with pkg.Mi355Index(d, "ip") as idx:
        idx.set_option("screen_dtype", screen)
        idx.add(Cneg)
        idx.reset_stats()
        dist, _ = _check(idx, oracle, Cneg, Qpos, 10, metric="ip")
        assert (dist > 0).all() and idx.stat("fallback_queries") == 0   # distance = -dot


def test_near_ties_stress(pkg, oracle):
    """many rows within a few ulp of each other: ranking must still equal the oracle's bit for bit."""
    rng = np.random.default_rng(99)
    d = 256
    c0 = rng.standard_normal(d).astype(np.float32)
    C = np.tile(c0, (4000, 1))
    C += (rng.standard_normal(C.shape) * 1e-6).astype(np.float32)
    Q = (c0[None, :] + 1e-3 * rng.standard_normal((4, d))).astype(np.float32)
    for screen in SCREENS:
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_dtype", screen)
            idx.add(C)
            _check(idx, oracle, C, Q, 25)


@pytest.mark.parametrize("screen", SCREENS)
@pytest.mark.parametrize("n,d,B,k", [(7000, 64, 600, 10), (40000, 128, 257, 7), (2600, 200, 513, 3), (66000, 96, 1024, 10)])
def test_persistent_screen_shapes(pkg, oracle, screen, n, d, B, k):
    """k_screen256's persistent walk at its corner shapes: 1-4 query tiles (3 -> 30 workgroups per XCD), one K-step
    per tile (d <= 64 bf16 / 128 int8), chunks smaller than the grid, ragged last tiles."""
    rng = np.random.default_rng(n + B)
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.set_option("screen_dtype", screen)
        idx.set_option("path", "screen")
        idx.add(C)
        _check(idx, oracle, C, Q, k)
        assert idx.stat("fallback_queries") == 0


@pytest.mark.parametrize("n,d,B,k", [(70_000, 768, 1024, 10), (40_000, 384, 300, 10), (33_000, 128, 257, 7), (26_000, 200, 513, 3),
                                      (30_000, 500, 700, 10), (90_000, 640, 129, 20), (120_000, 768, 600, 100), (30_000, 896, 400, 10)])
def test_register_resident_query_screen_equals_tile_screen(pkg, oracle, n, d, B, k):
    """Query blocks above 128 over an int8 shadow of at most 768 B per row go through k_screen_rq (query operand resident in
    registers, 128-row tiles); option screen_rq = 0 keeps k_screen256c: same results either way, equal to the oracle's.
    Shapes: every K-step count 1..6 (d = 128, 200, 384, 500, 640, 768), 1-4 query tiles, ragged last row tile and last query
    tile, rows of very different norms, k = 100; d = 896 (7 K-steps) has no register-resident form and must not take it."""
    rng = np.random.default_rng(n * 3 + d + B)
    C = rng.standard_normal((n, d)).astype(np.float32)
    C *= rng.uniform(0.1, 5.0, size=(n, 1)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    res = []
    for rq in (1, 0):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_dtype", "i8")
            idx.set_option("path", "screen")
            idx.set_option("screen_rq", rq)
            idx.add(C)
            idx.reset_stats()
            res.append(_check(idx, oracle, C, Q, k))
            assert idx.stat("fallback_queries") == 0
            assert (idx.stat("screen_rq_launches") > 0) == (rq == 1 and d <= 768)
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("drift", [0, 1, 3, 1024])
def test_sibling_drift_limiter_changes_no_result(pkg, oracle, drift):
    """k_screen_rq's drift limiter (option screen_drift: tiles a workgroup may lead the slowest workgroup on the same row tiles
    by; 0 = off) is progress control only: every setting -- off, the tightest (a leader waits at every flush of a sibling's
    hit-lane queue), the default, one that never engages -- returns the oracle's answer, over several launches of one handle
    (the launch stamp in the progress words advances; words of older launches must be ignored, not waited for)."""
    n, d, B, k = 600_000, 768, 1024, 10
    rng = np.random.default_rng(4242)
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.set_option("screen_dtype", "i8")
        idx.set_option("path", "screen")
        idx.set_option("screen_drift", drift)
        idx.add(C)
        idx.reset_stats()
        first = _check(idx, oracle, C, Q[:512], k)
        for _ in range(3):   # the same block again and a full block: launch stamps 2, 3, ...
            d2, r2 = idx.search(Q[:512], k)
            assert np.array_equal(r2, first[1])
        _check(idx, oracle, C, Q, k)
        assert idx.stat("fallback_queries") == 0 and idx.stat("screen_rq_launches") > 0
    with pytest.raises(Exception):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_drift", -1)


@pytest.mark.parametrize("screen", SCREENS)
@pytest.mark.parametrize("n,d,B,k", [(30000, 768, 1, 10), (20000, 768, 33, 10), (9000, 100, 64, 5), (50000, 384, 17, 20),
                                      (3000, 1536, 32, 10), (2100, 2048, 5, 10)])
def test_small_query_blocks_streaming_screen(pkg, oracle, screen, n, d, B, k):
    """Query blocks of at most 64 go through k_screen_stream (resident query block, ring of row stages, persistent
    workgroups) when the query image fits its 48 KiB, else through k_screen: same results either way, equal to the oracle's.
    Shapes: one and two query blocks of 32, a ragged last corpus tile, d that needs padding, d whose query image does not fit."""
    rng = np.random.default_rng(n * 7 + d + B)
    C = rng.standard_normal((n, d)).astype(np.float32)
    C *= rng.uniform(0.1, 5.0, size=(n, 1)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    res = []
    for stream in (1, 0):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_dtype", screen)
            idx.set_option("path", "screen")
            idx.set_option("screen_stream", stream)
            idx.add(C)
            res.append(_check(idx, oracle, C, Q, k))
            assert idx.stat("fallback_queries") == 0
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("screen", SCREENS)
@pytest.mark.parametrize("n,d,B,k", [(70_000, 768, 300, 10), (40_000, 128, 1024, 32), (9_000, 64, 20, 1), (300_000, 96, 64, 24)])
def test_starter_pass_equals_ladder_pass(pkg, oracle, screen, n, d, B, k):
    """round 3 pass schedule: a sampled threshold estimator (best value per 64-row slab of the first rows, exact re-score of
    the best-looking ones, threshold only -- nothing kept) replaces the smallest chunks and the chunk ends are planned; the
    result is the oracle's bit for bit, the same as the emit-all ladder's, whatever the sample holds -- here with the best
    rows of some queries INSIDE the sample (re-screened, no duplicates), duplicated rows across the sample's edge, and
    loose / zero rows in it."""
    rng = np.random.default_rng(n + k)
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    C[5] = Q[0] * 3.0                       # the best row of query 0 sits in the sample
    C[n - 1] = Q[0] * 3.0                   # ... and its exact duplicate at the far end (tie -> lower row first)
    C[100:104] = Q[1][None, :] + 0.05 * rng.standard_normal((4, d)).astype(np.float32)   # four near-ties in ONE slab
    C[7] = 0.0                              # zero row (irregular) in the sample
    C[9, 3] = 40.0 * np.abs(C[9]).max()     # outlier component: loose row for the int8 shadow
    res = []
    for starter, defer, companion in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0)):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_dtype", screen)
            idx.set_option("starter", starter)
            idx.set_option("defer_round_b", defer)  # survivors of a prune's cut carried to the next prune / re-scored at once
            idx.set_option("prune_companion", companion)  # 0: what the one-wave prune cannot hold is re-screened
            idx.add(C)
            idx.reset_stats()
            res.append(_check(idx, oracle, C, Q, k))
            assert idx.stat("starters") == starter * idx.stat("passes")
            assert idx.stat("fallback_queries") == 0
    assert all(np.array_equal(res[0][1], r[1]) for r in res[1:])


@pytest.mark.parametrize("k,expect_dtype", [(10, 2), (24, 2), (26, 2), (100, 2), (133, 2), (140, 1), (400, 1)])
def test_schedule_adapts_to_k(pkg, oracle, k, expect_dtype):
    """the wider int8 bound keeps ~16x k candidates per chunk, the bf16 bound ~3x: AUTO keeps int8 while its chunks can still
    grow (k <= 133: it wins up to k ~ 160 at the headline corpus) and the chunk growth shrinks with k, so no query overflows
    its candidate list (which would cost an exact re-scan)"""
    rng = np.random.default_rng(1000 + k)
    n, d, B = 150_000, 128, 300
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        idx.reset_stats()
        _check(idx, oracle, C, Q, k)
        assert idx.stat("fallback_queries") == 0
        assert idx.stat("screen_dtype_active") == expect_dtype


def test_overflow_is_rescreened_before_the_exact_scan(pkg, oracle):
    """a dense neighbourhood (4600 rows within ~0.02 cosine of each other around the query) overflows the candidate
    list under the int8 bound (E ~ 0.017 keeps ~90 % of them, ~3000 in the last chunk against 2048 slots) but not under
    the bf16 bound (E ~ 0.008, smaller chunks): those queries are re-screened with bf16, nobody pays the exact scan,
    results unchanged; AUTO then stays with bf16 for the index"""
    rng = np.random.default_rng(4)
    n, d, B, k = 60_000, 256, 64, 10
    C = rng.standard_normal((n, d)).astype(np.float32)
    v = rng.standard_normal(d).astype(np.float32)
    v /= np.linalg.norm(v)
    # cluster: v + noise of growing size -> cosines to v spread evenly over about [0.975, 0.995]
    m = 4600
    amp = np.sqrt(1.0 / np.linspace(0.995, 0.975, m) ** 2 - 1.0).astype(np.float32)
    noise = rng.standard_normal((m, d)).astype(np.float32)
    noise -= (noise @ v)[:, None] * v[None, :]
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    pos = rng.choice(n, size=m, replace=False)
    C[pos] = v[None, :] + amp[:, None] * noise
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[:8] = v[None, :] + 0.01 * rng.standard_normal((8, d)).astype(np.float32)  # 8 queries look into the cluster
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        assert idx.stat("screen_dtype_active") == 2
        idx.reset_stats()
        _check(idx, oracle, C, Q, k)
        assert idx.stat("retry_queries") >= 8 and idx.stat("fallback_queries") == 0
        assert idx.stat("i8_demoted") == 1 and idx.stat("screen_dtype_active") == 1   # 8 of 64 > 1 %
        idx.reset_stats()
        _check(idx, oracle, C, Q, k)                                                   # now bf16 from the start
        assert idx.stat("retry_queries") == 0 and idx.stat("fallback_queries") == 0
        # demotion is a probation, not a verdict: 16 more blocks at such a k, then int8 gets another try (and, overflowing
        # again, waits 32)
        q_easy = rng.standard_normal((8, d)).astype(np.float32)
        for _ in range(15):
            idx.search(q_easy, k)
        assert idx.stat("screen_dtype_active") == 2 and idx.stat("i8_demoted") == 0
        idx.reset_stats()
        _check(idx, oracle, C, Q, k)
        assert idx.stat("retry_queries") >= 8 and idx.stat("i8_demoted") == 1
        idx.set_option("screen_dtype", "auto")                                         # re-arms AUTO at once
        assert idx.stat("screen_dtype_active") == 2
    # a denser neighbourhood (6000 rows within 0.005): re-screening (bf16 at half the growth, then -- if that list
    # overflows too -- in chunks of <= 20 % of the rows, which splits the neighbourhood up) still avoids the exact scan
    m2 = 6000
    amp2 = np.sqrt(1.0 / np.linspace(0.995, 0.990, m2) ** 2 - 1.0).astype(np.float32)
    noise2 = rng.standard_normal((m2, d)).astype(np.float32)
    noise2 -= (noise2 @ v)[:, None] * v[None, :]
    noise2 /= np.linalg.norm(noise2, axis=1, keepdims=True)
    C2 = rng.standard_normal((n, d)).astype(np.float32)
    C2[rng.choice(n, size=m2, replace=False)] = v[None, :] + amp2[:, None] * noise2
    with pkg.Mi355Index(d) as idx:
        idx.add(C2)
        idx.reset_stats()
        _check(idx, oracle, C2, Q, k)
        assert idx.stat("retry_queries") >= 8 and idx.stat("fallback_queries") == 0


def test_int8_loose_rows_and_auto_fallback(pkg, oracle):
    """rows with outlier components do not quantise within the residual limit: they stay out of the int8 shadow and
    are re-scored for every query (results unchanged); with too many of them AUTO keeps the bf16 screen."""
    rng = np.random.default_rng(123)
    n, d = 6000, 256
    C = rng.standard_normal((n, d)).astype(np.float32)
    spikes = rng.choice(n, size=40, replace=False)
    C[spikes, rng.integers(0, d, size=40)] += 40.0  # one dominant component (|c_hat_k| ~ 0.9): clipped by the int8 grid
    Q = rng.standard_normal((12, d)).astype(np.float32)
    Q[:4] = C[spikes[:4]] + 0.1 * rng.standard_normal((4, d)).astype(np.float32)  # the spiky rows ARE the answers
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        assert idx.stat("loose_rows") >= 40 and idx.stat("irregular_rows") == 0
        assert idx.stat("screen_dtype_active") == 2
        _, rows = _check(idx, oracle, C, Q, 10)
        assert all(rows[i, 0] == spikes[i] for i in range(4))
        assert idx.stat("fallback_queries") == 0
    C2 = C.copy()
    C2[:1500, 0] += 40.0  # > 1024 loose rows
    with pkg.Mi355Index(d) as idx:
        idx.add(C2)
        assert idx.stat("loose_rows") > 1024
        assert idx.stat("screen_dtype_active") == 1  # AUTO resolved to bf16
        _check(idx, oracle, C2, Q, 10)
        idx.set_option("screen_dtype", "i8")
        with pytest.raises(pkg.NativeError):
            idx.search(Q, 10)


def test_merge_topk_device_equals_host_merge(pkg):
    """the shard-merge kernel == the host merge (total order incl. cross-shard ties, NaN, -1 padding)."""
    from autorag_research_amd.sharded import merge_topk_host

    rng = np.random.default_rng(5)
    world, B, k = 4, 37, 10
    d = np.sort(rng.random((world, B, k)), axis=2)
    r = rng.integers(0, 10_000_000_000, size=(world, B, k))  # global rows beyond int32
    d[1, :, 3] = d[0, :, 2]            # cross-shard distance ties -> lower row wins
    d[2, 5, 7:] = np.nan               # NaN tail on one shard
    r[3, :, 8:] = -1                   # short shard list
    d[3, :, 8:] = np.nan
    with pkg.Mi355Index(8) as idx:
        pd, pr = idx.dev_alloc(d.nbytes), idx.dev_alloc(r.nbytes)
        od, orr = idx.dev_alloc(B * k * 8), idx.dev_alloc(B * k * 8)
        idx.dev_upload(pd, d)
        idx.dev_upload(pr, r.astype(np.int64))
        idx.merge_topk_device(pd, pr, world, B, k, od, orr)
        idx.synchronize()
        gd, gr = np.empty((B, k)), np.empty((B, k), dtype=np.int64)
        idx.dev_download(od, gd)
        idx.dev_download(orr, gr)
        # packed layout [world][2][B][k] (what one all-gather delivers)
        packed = np.stack([d.view(np.int64), r.astype(np.int64)], axis=1)
        pp = idx.dev_alloc(packed.nbytes)
        idx.dev_upload(pp, packed)
        idx.merge_topk_packed_device(pp, world, B, k, od, orr)
        idx.synchronize()
        gd2, gr2 = np.empty((B, k)), np.empty((B, k), dtype=np.int64)
        idx.dev_download(od, gd2)
        idx.dev_download(orr, gr2)
        # and the packer itself
        one = idx.dev_alloc(2 * B * k * 8)
        idx.pack_topk_device(pd, pr, B, k, one)
        got = np.empty((2, B, k), dtype=np.int64)
        idx.synchronize()
        idx.dev_download(one, got)
        for p in (pd, pr, od, orr, pp, one):
            idx.dev_free(p)
    hd, hr = merge_topk_host(d, r.astype(np.int64), k)
    assert np.array_equal(gr, hr) and np.array_equal(gr2, hr)
    assert np.array_equal(gd, hd, equal_nan=True) and np.array_equal(gd2, hd, equal_nan=True)
    assert np.array_equal(got[0], d[0].view(np.int64)) and np.array_equal(got[1], r[0])


def test_search_device_buffers_and_row_offset(pkg, oracle):
    """device-resident queries/outputs (the bench path) + shard row offset."""
    rng = np.random.default_rng(6)
    n, d, B, k = 4000, 256, 50, 10
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        pc = idx.dev_alloc(C.nbytes)
        idx.dev_upload(pc, C)
        idx.add_device(pc, n)
        idx.dev_free(pc)
        idx.set_option("row_offset", 1_000_000)
        pq, od, orr = idx.dev_alloc(Q.nbytes), idx.dev_alloc(B * k * 8), idx.dev_alloc(B * k * 8)
        idx.dev_upload(pq, Q)
        idx.search_device(pq, B, k, od, orr)
        gd, gr = np.empty((B, k)), np.empty((B, k), dtype=np.int64)
        idx.dev_download(od, gd)
        idx.dev_download(orr, gr)
        assert np.array_equal(idx.get_rows(10, 3), C[10:13])
    rd, rr = oracle.topk_search(C, Q, k)
    assert np.array_equal(gr, rr + 1_000_000) and np.array_equal(gd, rd)


@pytest.mark.parametrize("cand_cap", [2048, 24])
def test_async_blocks_in_flight_equal_blocking_calls(pkg, oracle, cand_cap):
    """mi355dr_search_device_async / mi355dr_search_wait: seven blocks issued back to back (the ring holds four: the fifth
    completes the oldest), every one the oracle's answer -- also the blocks whose queries need the host's fix-ups AFTER later
    blocks were enqueued behind them: a zero query (the screen cannot rank it: exact scan), and with a 24-slot candidate
    buffer every query (overflow: re-screen, then exact scan).  Waits out of order and a blocking call in between."""
    rng = np.random.default_rng(77)
    n, d, k = 30_000, 128, 10
    C = rng.standard_normal((n, d)).astype(np.float32)
    sizes = [200, 1, 130, 64, 300, 7, 1024]
    Qs = [rng.standard_normal((b, d)).astype(np.float32) for b in sizes]
    Qs[2][5] = 0.0        # irregular query in block 2
    Qs[4][0] = C[17]      # an exact hit
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        idx.set_option("cand_cap", cand_cap)
        bufs = []
        for Q in Qs:
            pq, od, orr = idx.dev_alloc(Q.nbytes), idx.dev_alloc(len(Q) * k * 8), idx.dev_alloc(len(Q) * k * 8)
            idx.dev_upload(pq, Q)
            bufs.append((pq, od, orr))
        tickets = [idx.search_device_async(pq, len(Q), k, od, orr) for Q, (pq, od, orr) in zip(Qs, bufs)]
        assert tickets == sorted(tickets)
        idx.search_wait(tickets[2])                      # completes blocks 0..2
        _check(idx, oracle, C, Qs[1], k)                 # a blocking call in between completes the rest first
        idx.search_wait(tickets[-1])
        idx.search_wait(tickets[0])                      # waiting again is a no-op
        for Q, (pq, od, orr) in zip(Qs, bufs):
            gd, gr = np.empty((len(Q), k)), np.empty((len(Q), k), dtype=np.int64)
            idx.dev_download(od, gd)
            idx.dev_download(orr, gr)
            rd, rr = oracle.topk_search(C, Q, k)
            assert np.array_equal(gr, rr)
            ok = ~np.isnan(rd)
            assert np.array_equal(np.isnan(gd), np.isnan(rd)) and np.array_equal(gd[ok].view(np.uint64), rd[ok].view(np.uint64))
        if cand_cap == 24:
            assert idx.stat("fallback_queries") + idx.stat("retry_queries") > 0
        with pytest.raises(pkg.NativeError):
            idx.search_wait(tickets[-1] + 5)
        for b in bufs:
            for ptr_ in b:
                idx.dev_free(ptr_)


def test_large_corpus_properties(pkg, oracle):
    """N = 2M x d=768 (6 GB fp32 + 3 GB shadow): size-independent properties + planted answers + oracle spot checks.

    * every query's planted near-duplicates are found, at their exact (oracle) distances;
    * lists are sorted by (distance, row), rows unique and in range;
    * block search == one-by-one search (B=1 path) == scan path, bit for bit, on a subset;
    * idempotence: searching twice gives the same bits.
    """
    rng = np.random.default_rng(2026)
    n, d, B, k = 2_000_000, 768, 300, 10
    chunk = 250_000
    planted = {}
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.reserve(n)
        for c in range(n // chunk):
            X = np.random.default_rng(1234 + c).standard_normal((chunk, d), dtype=np.float32)
            X /= np.linalg.norm(X, axis=1, keepdims=True)
            if c in (0, 3, 7):  # plant 3 neighbours per query at known global rows
                sig = {0: 0.3, 3: 0.6, 7: 1.0}[c]
                for b in range(B):
                    pos = 1000 + 37 * b
                    v = Q[b] / np.linalg.norm(Q[b]) + sig / np.sqrt(d) * rng.standard_normal(d).astype(np.float32)
                    X[pos] = v.astype(np.float32)
                    planted.setdefault(b, []).append((c * chunk + pos, X[pos].copy()))
            idx.add(X)
        dist, rows = idx.search(Q, k)
        dist2, rows2 = idx.search(Q, k)
        assert np.array_equal(rows, rows2) and np.array_equal(dist, dist2)
        assert rows.min() >= 0 and rows.max() < n
        assert all(len(set(r)) == k for r in rows)
        assert (np.diff(dist, axis=1) >= 0).all()
        for b in range(B):
            for grow, vec in planted[b]:
                j = np.nonzero(rows[b] == grow)[0]
                assert j.size == 1, (b, grow)
                assert dist[b, j[0]] == oracle.cosine_distance(Q[b], vec)
            assert set(rows[b][:3]) == {g for g, _ in planted[b]}  # sigma = 0.3/0.6/1.0 neighbours outrank random rows
        sub = [0, 17, 299]
        for b in sub:
            d1, r1 = idx.search(Q[b], k)
            assert np.array_equal(r1[0], rows[b]) and np.array_equal(d1[0], dist[b])
        idx.set_option("path", "scan")
        ds, rs = idx.search(Q[sub], k)
        assert np.array_equal(rs, rows[sub]) and np.array_equal(ds, dist[sub])
        # oracle on a 300k-row slice containing the first planted chunk: same ids/distances as a GPU index of that slice
        sl = idx.get_rows(0, 300_000)
    rd, rr = oracle.topk_search(sl, Q[:16], k)
    with pkg.Mi355Index(d) as small:
        small.add(sl)
        gd, gr = small.search(Q[:16], k)
    assert np.array_equal(gr, rr) and np.array_equal(gd, rd)


def test_multi_block_queries_and_large_k(pkg, oracle):
    """B > 1024 is processed in internal blocks; k up to 1024 is served (kept lists, sorts, merges at their limits)."""
    rng = np.random.default_rng(44)
    n, d = 3000, 64
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((1100, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        _check(idx, oracle, C, Q, 5)            # two internal blocks (1024 + 76)
        _check(idx, oracle, C, Q[:9], 1024)     # k = 1024
        _check(idx, oracle, C, Q[:3], 700)
        with pytest.raises(pkg.NativeError):
            idx.search(Q[:1], 1025)             # beyond kKMax: refused loudly, not truncated


def test_encode_and_index_without_leaving_the_gpu(pkg, oracle):
    """embed -> index on the device: a (random-init) torch encoder's output tensors go to the index by pointer
    (mi355dr_add_rows_device); the stored rows are the encoder's rows bit for bit and the search equals the oracle's"""
    import torch

    from test_embeddings_ingest import _TinyEnc, _TinyTok
    from autorag_research_amd.embeddings import TorchEncoderEmbeddings
    from autorag_research_amd.ingest import index_texts_on_device

    rng = np.random.default_rng(12)
    words = [f"w{i}" for i in range(200)]
    texts = [" ".join(rng.choice(words, size=int(rng.integers(3, 12)))) for _ in range(1500)]
    enc = TorchEncoderEmbeddings(_TinyEnc(64), _TinyTok(), pooling="mean", device="cuda:0", batch_size=256)
    with pkg.Mi355Index(64) as idx:
        assert index_texts_on_device(enc, texts, idx, batch_size=400) == len(texts)
        assert len(idx) == len(texts)
        ref = enc.encode_to_device(texts).cpu().numpy()
        # batches of 400 vs one pass of 256-row model batches: same rows (row-wise model, deterministic kernels)
        stored = idx.get_rows(0, len(texts))
        assert np.allclose(stored, ref, atol=1e-6)
        q = enc.encode_to_device(texts[:7]).cpu().numpy()
        dist, rows = idx.search(q, 5)
        rd, rr = oracle.topk_search(stored, q, 5)
        assert np.array_equal(rows, rr) and np.array_equal(dist.view(np.uint64), rd.view(np.uint64))
        assert (rows[:, 0] == np.arange(7)).all() or (dist[:, 0] < 1e-6).all()  # every text finds itself (or a duplicate)


@pytest.mark.parametrize("mode", ["gauss", "clustered"])
def test_bf16_second_screen_inside_the_prune_keeps_results_exact(pkg, oracle, mode):
    """option "prefilter16" (off by default: no measurable gain): round-B candidates of the int8 screen are screened once
    more on their bf16 shadow rows before the exact re-score.  Same ids, same float8 distances, fewer exact re-scores."""
    rng = np.random.default_rng(31)
    n, d, B, k = 60_000, 256, 200, 10
    C = rng.standard_normal((n, d)).astype(np.float32)
    if mode == "clustered":
        cen = rng.standard_normal((300, d)).astype(np.float32)
        C = cen[rng.integers(0, 300, size=n)] + (0.05 * rng.standard_normal((n, d))).astype(np.float32)
    C[17] = 0.0
    C[18] = np.nan
    C[19] *= 1e20
    Q = C[rng.integers(0, n, size=B)] + (0.05 * rng.standard_normal((B, d))).astype(np.float32)
    rd, rr = oracle.topk_search(C, Q, k)
    rescored = {}
    for pf in (0, 1):
        with pkg.Mi355Index(d) as idx:
            idx.set_option("screen_dtype", "i8")
            idx.set_option("prefilter16", pf)
            idx.add(C)
            dist, rows = idx.search(Q, k)
            rescored[pf] = idx.stat("rescored")
        assert np.array_equal(rows, rr)
        m = ~np.isnan(rd)
        assert np.array_equal(np.isnan(dist), ~m) and np.array_equal(dist[m].view(np.uint64), rd[m].view(np.uint64))
    assert rescored[1] <= rescored[0]


@pytest.mark.parametrize("k", [40, 100, 128])
@pytest.mark.parametrize("order", ["ascending", "ties", "descending"])
def test_two_wave_prune_on_adversarial_row_orders(pkg, oracle, k, order):
    """k_prune_wide (33 <= k <= 128) where its bookkeeping is stressed instead of its common case:
    * `ascending`: the rows' similarity to the queries GROWS with the row index, so every chunk's candidates beat everything kept
      (round A cannot hold the entrants, round B passes its filter by the hundred: the 512-entry exact buffer is re-selected
      again and again, thresholds lag the data);
    * `ties`: a third of the corpus are exact copies of 50 rows (float-image ties far beyond k: selections by image return more
      than k, the (key, row) sort decides);
    * `descending`: the best rows first (the starter's sample already holds the final top-k; later chunks append almost nothing).
    Also with every prune re-scoring its survivors at once (defer_round_b = 0: round B and the shrink path in EVERY prune) and with the
    round-5 schedule (prune_wide = 0).  All against the oracle, bit for bit."""
    rng = np.random.default_rng(900 + k)
    n, d, B = 220_000, 128, 96
    Q = rng.standard_normal((B, d)).astype(np.float32)
    C = rng.standard_normal((n, d)).astype(np.float32)
    axis = Q[:48].mean(axis=0)
    axis /= np.linalg.norm(axis)
    if order in ("ascending", "descending"):
        w = np.linspace(0.0, 3.0, n, dtype=np.float32)   # the component along the queries' common direction grows with the row
        if order == "descending":
            w = w[::-1].copy()
        C += w[:, None] * axis[None, :]
    else:
        base = rng.standard_normal((50, d)).astype(np.float32) + 2.0 * axis[None, :]
        pos = rng.choice(n, size=n // 3, replace=False)
        C[pos] = base[rng.integers(0, 50, size=pos.size)]
    rd, rr = oracle.topk_search(C, Q, k)
    small_steps = {"chunk_growth": 1, "starter_rows_wide": 4096, "defer_round_b": 0}   # nine chunks, round B in every prune
    for opts in ({}, {"defer_round_b": 0}, small_steps, {"prune_wide": 0}):
        with pkg.Mi355Index(d) as idx:
            for key, val in opts.items():
                idx.set_option(key, val)
            idx.add(C)
            idx.reset_stats()
            dist, rows = idx.search(Q, k)
            assert np.array_equal(rows, rr), (order, k, opts)
            assert np.array_equal(dist.view(np.uint64), rd.view(np.uint64)), (order, k, opts)
            assert idx.stat("fallback_queries") == 0
            if order == "ties" and k >= 100 and opts is small_steps:
                # ~1 460 copies of the best row tie at the k-th place: round B re-scores them by the hundred and every one passes the
                # buffer's filter (equal images), so the 512-entry exact buffer is re-selected several times per prune
                assert idx.stat("rescored") / B > 1000, idx.stat("rescored") / B
