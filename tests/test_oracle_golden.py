"""CPU: pin the oracle against the golden vectors generated from the imported reference (tests/golden/)."""

import numpy as np
import pytest

from helpers import GOLDEN


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN / "scores_golden.npz")


def test_cosine_matches_reference_inprocess_math(oracle, gold):
    """oracle cosine distance vs gqr_hybrid._cosine_scores / calculate_cosine_similarity (float64), <= 1e-6."""
    C, Q = gold["cos_C"], gold["cos_Q"]
    ref = gold["cos_scores"]  # [5, 257] cosine similarity (float64)
    got = np.array([[1.0 - oracle.cosine_distance(q, c) for c in C] for q in Q])
    assert np.abs(got - ref).max() <= 1e-6
    pair = gold["cos_pair"]
    assert np.abs(got[:, :8] - pair).max() <= 1e-6
    # ranking agrees wherever the reference's own scores are separated by more than the fp32 noise
    dist, rows = oracle.topk_search(C, Q, 20)
    for b in range(Q.shape[0]):
        order = np.argsort(-ref[b], kind="stable")[:20]
        gaps = np.abs(np.diff(ref[b][order]))
        if gaps.min() > 1e-6:
            assert np.array_equal(rows[b], order)
        assert set(rows[b][:10]) <= set(np.argsort(-ref[b])[:12])


def test_maxsim_matches_reference_inprocess_math(oracle, gold):
    tok, off = gold["ms_tok"], gold["ms_offsets"]
    qtok, qoff = gold["ms_qtok"], gold["ms_qoff"]
    ref = gold["ms_scores"]  # [4 queries, 40 docs] = (1/n_q) sum max   (gqr_hybrid._maxsim_scores)
    assert np.abs(ref - gold["ms_scores_heaven"]).max() < 1e-12  # the reference's two restatements agree
    n_docs = off.shape[0] - 1
    for b in range(qoff.shape[0] - 1):
        q = qtok[qoff[b]:qoff[b + 1]]
        got = np.array([-oracle.maxsim_distance(tok[off[i]:off[i + 1]], q) / q.shape[0] for i in range(n_docs)])
        assert np.abs(got - ref[b]).max() <= 1e-6
    dist, rows = oracle.maxsim_topk(tok, off, qtok, qoff, 5)
    for b in range(qoff.shape[0] - 1):
        nq = qoff[b + 1] - qoff[b]
        order = np.argsort(-ref[b], kind="stable")[:5]
        if np.abs(np.diff(ref[b][order])).min() > 1e-6:
            assert np.array_equal(rows[b], order)
        assert np.abs(-dist[b].astype(np.float64) / nq - ref[b][rows[b]]).max() <= 1e-6


def test_reference_hand_computed_maxsim_known_answer(oracle, gold):
    """tests/autorag_research/pipelines/retrieval/test_gqr_hybrid_pipeline.py:65-74 -> [0.75, 0.6]."""
    q = gold["ms_known_q"].astype(np.float32)
    c0, c1 = gold["ms_known_c0"].astype(np.float32), gold["ms_known_c1"].astype(np.float32)
    got = [-oracle.maxsim_distance(c0, q) / 2, -oracle.maxsim_distance(c1, q) / 2]
    assert np.allclose(got, [0.75, 0.6], atol=1e-7)
    assert np.allclose(gold["ms_known_scores"], [0.75, 0.6])


def test_canonical_chain_vs_literal_pgvector_loop(oracle):
    """fused chain (canonical) vs separate mul+add (literal loop): same value within fp32 accumulation noise."""
    rng = np.random.default_rng(1)
    for d in (3, 100, 768, 1536):
        a, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
        d1, d2 = oracle.cosine_distance(a, b), oracle.cosine_distance(a, b, seq=True)
        assert abs(d1 - d2) <= 1e-6
        assert abs(oracle.dot(a, b) - float(a.astype(np.float64) @ b.astype(np.float64))) <= 1e-4


def test_c_oracle_vs_numpy_float64_topk(oracle):
    rng = np.random.default_rng(2)
    C = rng.standard_normal((4000, 96)).astype(np.float32)
    Q = rng.standard_normal((17, 96)).astype(np.float32)
    d1, r1 = oracle.topk_search(C, Q, 10)
    d2, r2 = oracle.np_topk_from_distance(oracle.np_cosine_distance_matrix(C, Q), 10)
    assert np.array_equal(r1, r2)
    assert np.abs(d1 - d2).max() < 1e-6
    for th in (1, 3):  # thread count must not change anything
        d3, r3 = oracle.topk_search(C, Q, 10, threads=th)
        assert np.array_equal(r3, r1) and np.array_equal(d3, d1)


def test_oracle_edge_cases(oracle):
    rng = np.random.default_rng(3)
    C = rng.standard_normal((5, 8)).astype(np.float32)
    C[2] = 0  # zero norm -> NaN, sorts last
    C[4] = C[1]  # exact duplicate -> lower row first
    Q = np.stack([C[1] * 3.0, np.zeros(8, np.float32)])
    dist, rows = oracle.topk_search(C, Q, 7)
    assert rows[0, 0] == 1 and rows[0, 1] == 4 and dist[0, 0] == dist[0, 1]
    assert rows[0, 4] == 2 and np.isnan(dist[0, 4])
    assert (rows[0, 5:] == -1).all() and np.isnan(dist[0, 5:]).all()
    assert np.array_equal(rows[1, :5], np.arange(5)) and np.isnan(dist[1]).all()  # zero query: all NaN, row order
    d0, r0 = oracle.topk_search(np.zeros((0, 8), np.float32), Q, 3)
    assert (r0 == -1).all()
    # self-distance ~ 0, ascending distances (the reference's own property tests, test_base_vector_repository.py)
    d, r = oracle.topk_search(C, C[[0, 1, 3]], 5)
    assert (r[:, 0] == [0, 1, 3]).all() and (np.abs(d[:, 0]) < 1e-6).all()
    assert (np.diff(np.where(np.isnan(d), np.inf, d), axis=1) >= 0).all()
    # inner product metric: distance = -dot
    di, ri = oracle.topk_search(C, Q[:1], 2, metric="ip")
    assert ri[0, 0] in (1, 4) and di[0, 0] == -oracle.dot(C[1], Q[0])


def test_ragged_maxsim_empty_docs(oracle):
    rng = np.random.default_rng(4)
    tok = rng.standard_normal((10, 4)).astype(np.float32)
    off = np.array([0, 3, 3, 7, 10], dtype=np.int64)  # doc 1 is empty (NULL embeddings) -> skipped
    q = rng.standard_normal((2, 4)).astype(np.float32)
    dist, rows = oracle.maxsim_topk(tok, off, q, np.array([0, 2], np.int32), 5)
    assert set(rows[0][:3]) == {0, 2, 3} and (rows[0][3:] == -1).all()
    assert (np.diff(dist[0][:3]) >= 0).all()


def test_gqr_oracle_matches_reference_loops():
    """oracle/gqr_ref.py vs the reference's _optimize_query_embedding / _optimize_query_multi_embedding /
    _optimize_in_score_space outputs (tests/golden/gqr_golden.npz, generated by importing gqr_hybrid.py)."""
    import numpy as np

    from helpers import load_gqr_golden
    from oracle import gqr_ref as G

    g, _ = load_gqr_golden()
    C = g["single_C"].astype(np.float64)
    tok, off, qoff = g["multi_tok"].astype(np.float64), g["multi_off"], g["multi_qoff"]
    for s, (n, lr, T, a) in enumerate(g["params"]):
        n = int(n)
        for b, pool in enumerate(g["single_pools"]):
            m = int((pool >= 0).sum())
            got = G.refine_single(g["single_Q"][b], C[pool[:m]], g["single_comp"][b, :m], n, lr, T, a)
            assert np.allclose(got, g["single_expected"][s, b, :m], rtol=0, atol=1e-12)
            assert np.isnan(g["single_expected"][s, b, m:]).all()
        for b, pool in enumerate(g["multi_pools"]):
            m = int((pool >= 0).sum())
            docs = [tok[off[i]:off[i + 1]] for i in pool[:m]]
            got = G.refine_multi(g["multi_qtok"][qoff[b]:qoff[b + 1]], docs, g["multi_comp"][b, :m], n, lr, T, a)
            assert np.allclose(got, g["multi_expected"][s, b, :m], rtol=0, atol=1e-12)
        for b, m in enumerate(g["score_counts"]):
            got = G.refine_scores(g["score_primary"][b, :m], g["score_comp"][b, :m], n, lr, T, a)
            assert np.allclose(got, g["score_expected"][s, b, :m], rtol=0, atol=1e-12)
    # the zero query and the zero candidate row of the fixture (norm floors, gqr_hybrid.py:68-73)
    assert (g["single_expected"][:, 4, :33] == 0).all() and g["single_expected"][0, 2, 2] == 0.0
