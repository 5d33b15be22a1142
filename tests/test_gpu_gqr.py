"""GPU parity of the Guided Query Refinement kernels (csrc/k_gqr.h, through the C ABI) against the reference's golden
outputs (tests/golden/gqr_golden.*) and the CPU oracle (oracle/gqr_ref.py)."""

import time

import numpy as np
import pytest
from helpers import build_golden_stores, check_gqr_flow, load_gqr_golden

pytestmark = pytest.mark.gpu

# float64 on both sides; the only differences are summation order and exp's last bit, compounded over <= 40 steps
ATOL = 1e-10


@pytest.fixture(scope="module")
def pkg(native_built):
    import autorag_research_amd as p

    return p


def test_single_vector_refinement_matches_reference_golden(pkg):
    g, _ = load_gqr_golden()
    with pkg.Mi355Index(g["single_C"].shape[1]) as idx:
        idx.add(g["single_C"])
        for s, (n, lr, T, a) in enumerate(g["params"]):
            got = idx.gqr_refine(g["single_Q"], g["single_pools"], g["single_comp"], int(n), lr, T, a)
            exp = g["single_expected"][s]
            assert np.array_equal(np.isnan(got), np.isnan(exp))  # padding comes back as NaN
            live = ~np.isnan(exp)
            assert np.abs(got[live] - exp[live]).max() <= ATOL
        # zero query: all scores exactly 0; zero candidate row: norm floored, score exactly 0
        assert (got[4, :33] == 0).all() and got[2, 2] == 0.0


def test_multi_vector_refinement_matches_reference_golden(pkg):
    g, _ = load_gqr_golden()
    with pkg.Mi355Index(g["multi_tok"].shape[1]) as idx:
        idx.add_multivec(g["multi_tok"], g["multi_off"])
        for s, (n, lr, T, a) in enumerate(g["params"]):
            got = idx.gqr_refine_maxsim(g["multi_qtok"], g["multi_qoff"], g["multi_pools"], g["multi_comp"], int(n), lr, T, a)
            exp = g["multi_expected"][s]
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            live = ~np.isnan(exp)
            assert np.abs(got[live] - exp[live]).max() <= ATOL


def test_score_space_refinement_matches_reference_golden(pkg):
    g, _ = load_gqr_golden()
    with pkg.Mi355Index(8) as idx:  # no rows needed: the scores themselves are refined
        for s, (n, lr, T, a) in enumerate(g["params"]):
            got = idx.gqr_refine_scores(g["score_primary"], g["score_counts"], g["score_comp"], int(n), lr, T, a)
            exp = g["score_expected"][s]
            live = ~np.isnan(exp)
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            assert np.abs(got[live] - exp[live]).max() <= ATOL


def test_gqr_pipeline_on_gpu_matches_reference_dicts(pkg):
    """the whole caller (children -> pool -> refinement kernels -> ranking) against the reference's _retrieve_by_id."""
    from autorag_research_amd.pipelines import Mi355VectorSearchRetrievalPipeline

    store, _ = build_golden_stores()
    check_gqr_flow(store, lambda mode: Mi355VectorSearchRetrievalPipeline(lambda: store, f"vs_{mode}", search_mode=mode),
                   atol=1e-9)


def test_bench_shaped_pools_match_oracle_and_row_offset(pkg, oracle):
    """d = 768, 256 queries x 40 candidates (top_k 10 x fetch multiplier 2, two retrievers), reference defaults; also
    with global row ids (a shard with row_offset)."""
    from oracle import gqr_ref

    rng = np.random.default_rng(5)
    n, d, B, P = 5000, 768, 256, 40
    C = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32).astype(np.float64)
    pools = np.stack([rng.choice(n, size=P, replace=False) for _ in range(B)]).astype(np.int64)
    pools[7, 31:] = -1  # ragged pool
    comp = rng.dirichlet(np.ones(P), size=B)
    comp[7, 31:] = 0
    comp[7] /= comp[7].sum()
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        t0 = time.perf_counter()
        got = idx.gqr_refine(Q, pools, comp, 25, 0.1, 1.0, 0.5)
        dt = time.perf_counter() - t0
        print(f"gqr_refine: {B} queries x {P} candidates x d={d}, 25 steps: {dt * 1e3:.2f} ms on the GPU (host API)")
        idx.set_option("row_offset", 1_000_000)
        shifted = np.where(pools >= 0, pools + 1_000_000, pools)
        got2 = idx.gqr_refine(Q, shifted, comp, 25, 0.1, 1.0, 0.5)
        assert np.array_equal(got, got2, equal_nan=True)
    Cd = C.astype(np.float64)
    t0 = time.perf_counter()
    for b in (0, 7, 100, 255):
        m = int((pools[b] >= 0).sum())
        exp = gqr_ref.refine_single(Q[b], Cd[pools[b, :m]], comp[b, :m], 25, 0.1, 1.0, 0.5)
        assert np.abs(got[b, :m] - exp).max() <= ATOL
        assert np.isnan(got[b, m:]).all()
    print(f"numpy oracle: {(time.perf_counter() - t0) / 4 * 1e3:.2f} ms per query")


def test_late_interaction_pools_at_colbert_shape_match_oracle(pkg, oracle):
    """d = 128, 32 query vectors, docs of 20..180 vectors (ColBERT / ColPali shapes), 16 queries x 40 candidates."""
    from oracle import gqr_ref

    rng = np.random.default_rng(6)
    d, n_docs, B, P = 128, 400, 16, 40
    lens = rng.integers(20, 181, size=n_docs)
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q_lens = rng.integers(5, 33, size=B)
    q_lens[0] = 32
    qtok = rng.standard_normal((int(q_lens.sum()), d)).astype(np.float32)
    qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
    qtok = qtok.astype(np.float64)
    qoff = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    pools = np.stack([rng.choice(n_docs, size=P, replace=False) for _ in range(B)]).astype(np.int64)
    comp = rng.dirichlet(np.ones(P), size=B)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        t0 = time.perf_counter()
        got = idx.gqr_refine_maxsim(qtok, qoff, pools, comp, 25, 0.1, 1.0, 0.5)
        print(f"gqr_refine_maxsim: {B} queries x {P} docs, 25 steps: {(time.perf_counter() - t0) * 1e3:.2f} ms")
        # a page-sized block (the 16 queries repeated): what a batched run() launches at once
        rep = 32
        big_q = np.concatenate([qtok] * rep, axis=0)
        big_off = np.concatenate([[0], np.cumsum(np.tile(q_lens, rep))]).astype(np.int32)
        t0 = time.perf_counter()
        big = idx.gqr_refine_maxsim(big_q, big_off, np.tile(pools, (rep, 1)), np.tile(comp, (rep, 1)), 25, 0.1, 1.0, 0.5)
        print(f"gqr_refine_maxsim: {B * rep} queries x {P} docs, 25 steps: {(time.perf_counter() - t0) * 1e3:.2f} ms")
        assert np.array_equal(big[:B], got) and np.array_equal(big[-B:], got)
    tokd = tok.astype(np.float64)
    for b in (0, 5, 15):
        docs = [tokd[off[i]:off[i + 1]] for i in pools[b]]
        exp = gqr_ref.refine_multi(qtok[qoff[b]:qoff[b + 1]], docs, comp[b], 25, 0.1, 1.0, 0.5)
        assert np.abs(got[b] - exp).max() <= ATOL


def test_argument_errors(pkg):
    C = np.eye(8, dtype=np.float32)
    comp = np.full((1, 3), 1 / 3)
    q = np.ones((1, 8))
    with pkg.Mi355Index(8) as idx:
        idx.add(C)
        ok = idx.gqr_refine(q, [[0, 1, 2]], comp, 2, 0.1, 1.0, 0.5)
        assert ok.shape == (1, 3)
        with pytest.raises(pkg.NativeError, match="not a row"):
            idx.gqr_refine(q, [[0, 1, 99]], comp, 2, 0.1, 1.0, 0.5)
        with pytest.raises(pkg.NativeError, match="padding"):
            idx.gqr_refine(q, [[0, -1, 2]], comp, 2, 0.1, 1.0, 0.5)
        for bad in ((0, 0.1, 1.0, 0.5), (2, 0.0, 1.0, 0.5), (2, 0.1, 0.0, 0.5), (2, 0.1, 1.0, 1.5)):
            with pytest.raises(pkg.NativeError):
                idx.gqr_refine(q, [[0, 1, 2]], comp, *bad)
        with pytest.raises(pkg.NativeError, match="no multi-vector"):
            idx.gqr_refine_maxsim(q, [0, 1], [[0, 1, 2]], comp, 2, 0.1, 1.0, 0.5)
        assert idx.gqr_refine(np.zeros((0, 8)), np.zeros((0, 3), np.int64), np.zeros((0, 3)), 2, 0.1, 1.0, 0.5).shape == (0, 3)
