"""GPU parity: MaxSim (VectorChord `@#`) through the C ABI vs the CPU oracle -- fp32 distances and doc ids bit-exact."""

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(native_built):
    import autorag_research_amd as p

    return p


def _ragged(rng, n_docs, d, tmin, tmax, unit=True):
    lens = rng.integers(tmin, tmax + 1, size=n_docs)
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    if unit:
        tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    return tok, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def _queries(rng, lens, d, unit=True):
    qs = []
    for t in lens:
        m = rng.standard_normal((t, d)).astype(np.float32)
        if unit and t:
            m /= np.linalg.norm(m, axis=1, keepdims=True)
        qs.append(m)
    qtok = np.concatenate(qs, axis=0) if qs else np.zeros((0, d), np.float32)
    return qtok, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def _check(idx, oracle, tok, off, qtok, qoff, k):
    """both forms of the search -- bf16 screen + exact re-score of the candidates (default) and the exact kernel over
    every doc -- must equal the oracle bit for bit"""
    rd, rr = oracle.maxsim_topk(tok, off, qtok, qoff, k)
    for screen in (1, 0):
        idx.set_option("maxsim_screen", screen)
        dist, rows = idx.search_maxsim(qtok, qoff, k)
        assert np.array_equal(rows, rr), screen
        assert np.array_equal(np.isnan(dist), np.isnan(rd))
        ok = ~np.isnan(dist)
        assert np.array_equal(dist[ok].view(np.uint32), rd[ok].view(np.uint32)), screen
    idx.set_option("maxsim_screen", 1)


@pytest.mark.parametrize("d,n_docs,tmax,qlens,k", [
    (128, 800, 60, [300, 32, 129, 257], 10),   # queries with more than 128 vectors next to ordinary ones
    (768, 120, 40, [20, 70, 33], 5),           # d = 768: one launch stages 32 query vectors, everything longer runs in tiles
    (1000, 40, 12, [40], 3),                   # dim 1000 (one launch: 32 vectors of 1004 floats)
])
def test_long_queries_and_wide_vectors_run_in_tiles(pkg, oracle, d, n_docs, tmax, qlens, k):
    """round 3: VectorChord's `@#` takes any number of query vectors (reference base.py:518-524) and any dimension; the kernel
    stages min(128, what 160 KiB of LDS hold) query vectors per launch and scores a longer query in TILES, each launch
    continuing the per-document fp32 sums of the one before it -- the oracle's one chain over the query's vectors, bit for
    bit, in the full search and in the candidate-subset form (HEAVEN stage 2 / reranker)."""
    rng = np.random.default_rng(d + n_docs)
    tok, off = _ragged(rng, n_docs, d, 1, tmax)
    qtok, qoff = _queries(rng, qlens, d)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        _check(idx, oracle, tok, off, qtok, qoff, k)
        # subset form: every query against a list of docs (some ids not in the store) == the oracle's full distances
        ids = rng.integers(-2, n_docs + 2, size=(len(qlens), 17)).astype(np.int64)
        got = idx.maxsim_subset(qtok, qoff, ids)
        fd, fr = oracle.maxsim_topk(tok, off, qtok, qoff, n_docs)   # every doc's distance
        for b in range(len(qlens)):
            want = {int(r): x for r, x in zip(fr[b], fd[b])}
            for j, i in enumerate(ids[b]):
                if 0 <= i < n_docs:
                    assert got[b, j].view(np.uint32) == np.float32(want[int(i)]).view(np.uint32)
                else:
                    assert np.isnan(got[b, j])


def test_golden_inputs(pkg, oracle):
    g = np.load(GOLDEN / "scores_golden.npz")
    tok, off, qtok, qoff = g["ms_tok"], g["ms_offsets"], g["ms_qtok"], g["ms_qoff"]
    with pkg.Mi355Index(tok.shape[1]) as idx:
        idx.add_multivec(tok, off)
        assert idx.n_docs() == off.shape[0] - 1
        _check(idx, oracle, tok, off, qtok, qoff, 5)
        # and the documented score = -distance / n_q agrees with the reference's float64 helper
        dist, rows = idx.search_maxsim(qtok, qoff, 5)
        for b in range(qoff.shape[0] - 1):
            nq = qoff[b + 1] - qoff[b]
            assert np.abs(-dist[b].astype(np.float64) / nq - g["ms_scores"][b][rows[b]]).max() <= 1e-6


@pytest.mark.parametrize("d,n_docs,tmin,tmax,qlens,k", [
    (128, 3000, 1, 180, [32, 24, 1, 32, 32, 7], 10),      # ColBERT-like text docs, several queries per launch
    (128, 60, 1030, 1030, [24, 20], 10),                  # ColPali-like pages: 1030 patches per doc
    (128, 500, 1, 40, [128, 100], 100),                   # 4 column blocks, k=100
    (96, 700, 1, 70, [5, 33], 10),                        # dim not a power of two, 2 column blocks
    (20, 300, 1, 9, [3], 400),                            # dim padded to 24, k > n_docs
])
def test_maxsim_matches_oracle(pkg, oracle, d, n_docs, tmin, tmax, qlens, k):
    rng = np.random.default_rng(d * 1000 + n_docs)
    tok, off = _ragged(rng, n_docs, d, tmin, tmax)
    qtok, qoff = _queries(rng, qlens, d)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        _check(idx, oracle, tok, off, qtok, qoff, k)


@pytest.mark.gpu
@pytest.mark.parametrize("tmin,tmax,qlens", [
    (200, 420, [24, 20, 33, 7]),      # 7..14 blocks per doc: the four waves get uneven shares, two column blocks
    (1, 70, [24, 128, 5]),            # short docs forced through the cooperative form: waves without a block, empty docs
    (1030, 1030, [24, 24, 24, 24, 24, 24, 24, 24]),   # pages, two groups of four queries
])
def test_candidate_lists_scored_by_one_workgroup_per_candidate(pkg, oracle, tmin, tmax, qlens):
    """the exact kernel on the screen's candidate lists, cooperative form (option maxsim_coop: the four waves of a workgroup
    share one candidate's blocks, column maxima meet in LDS): the same bits as one wave per candidate and as the oracle"""
    rng = np.random.default_rng(tmax * 7 + len(qlens))
    d, n_docs, k = 128, 90 if tmax > 500 else 400, 10
    tok, off = _ragged(rng, n_docs, d, tmin if tmin > 1 else 0, tmax)
    qtok, qoff = _queries(rng, qlens, d)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        got = {}
        for coop in (1, 0, -1):
            idx.set_option("maxsim_coop", coop)
            idx.reset_stats()
            _check(idx, oracle, tok, off, qtok, qoff, k)
            assert idx.stat("maxsim_screened") > 0
            got[coop] = idx.search_maxsim(qtok, qoff, k)
        assert np.array_equal(got[1][1], got[0][1]) and np.array_equal(got[1][0].view(np.uint32), got[0][0].view(np.uint32))


def test_empty_docs_appends_and_unnormalised(pkg, oracle):
    rng = np.random.default_rng(77)
    d = 64
    tok, off = _ragged(rng, 200, d, 0, 12, unit=False)  # some docs have zero vectors (NULL embeddings)
    assert (np.diff(off) == 0).any()
    qtok, qoff = _queries(rng, [4, 0, 9], d, unit=False)  # an empty query returns nothing
    half = 100
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok[: off[half]], off[: half + 1])
        idx.add_multivec(tok[off[half]:], off[half:] - off[half])  # appended in two calls
        assert idx.n_docs() == 200
        rd, rr = oracle.maxsim_topk(tok, off, qtok, qoff, 20)
        for screen in (1, 0):
            idx.set_option("maxsim_screen", screen)
            dist, rows = idx.search_maxsim(qtok, qoff, 20)
            assert np.array_equal(rows[[0, 2]], rr[[0, 2]])
            assert np.array_equal(dist[[0, 2]].view(np.uint32), rd[[0, 2]].view(np.uint32))
            assert (rows[1] == -1).all()


def test_maxsim_screen_is_used_and_survives_hard_cases(pkg, oracle):
    """the screen path really runs (stats), its candidate lists stay small on ordinary data, and the cases it cannot
    bound or hold fall back to the exact full scan with unchanged results: near-duplicate docs (hundreds of docs within
    the bound of each other), a candidate overflow, non-finite stored values."""
    rng = np.random.default_rng(5)
    d = 128
    tok, off = _ragged(rng, 20000, d, 4, 40)
    qtok, qoff = _queries(rng, [32, 16, 32, 8, 32], d)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        idx.reset_stats()
        _check(idx, oracle, tok, off, qtok, qoff, 10)
        assert idx.stat("maxsim_screened") == 5 and idx.stat("maxsim_fallbacks") == 0
        assert idx.stat("maxsim_candidates") < 5 * 2000  # a small fraction of the 20000 docs is re-scored
    # near-duplicates: 9000 noisy copies of one doc -> every copy is within 2E of the k-th best
    base_tok, base_off = _ragged(rng, 1, d, 20, 20)
    reps = 9000
    tok2 = np.concatenate([base_tok + (1e-4 * rng.standard_normal(base_tok.shape)).astype(np.float32) for _ in range(reps)])
    off2 = np.arange(0, reps + 1, dtype=np.int64) * 20
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok2, off2)
        idx.reset_stats()
        _check(idx, oracle, tok2, off2, qtok[:32], qoff[:2], 10)
        assert idx.stat("maxsim_fallbacks") >= 1  # 9000 candidates > the 8192-entry list: exact full scan took over
    tok3 = tok.copy()
    tok3[7, 3] = np.inf
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok3, off)
        idx.reset_stats()
        dist, rows = idx.search_maxsim(qtok, qoff, 10)
        assert idx.stat("maxsim_screened") == 0  # a store with non-finite values is never screened
        rd, rr = oracle.maxsim_topk(tok3, off, qtok, qoff, 10)
        assert np.array_equal(rows, rr)


def test_service_and_pipelines_on_gpu_match_reference_dicts(pkg):
    """the host mirror over the REAL GPU index reproduces the reference's output dicts (service_golden.json)."""
    import asyncio

    from helpers import build_golden_stores, load_service_golden
    from autorag_research_amd.pipelines import (Mi355ImageVectorSearchRetrievalPipeline,
                                                 Mi355VectorSearchRetrievalPipeline)
    from autorag_research_amd.service import Mi355RetrievalService

    store, g = build_golden_stores()
    gold = load_service_golden()
    k = gold["top_k"]

    def same(got, exp):
        assert [r["doc_id"] for r in got] == [r["doc_id"] for r in exp]
        assert [r["content"] for r in got] == [r["content"] for r in exp]
        assert np.allclose([r["score"] for r in got], [r["score"] for r in exp], rtol=0, atol=1e-12)

    s = Mi355RetrievalService(lambda: store)
    qids = [f"q{i}" for i in range(6)]
    for got, exp in zip(s.vector_search(qids, k, "single"), gold["service_single"], strict=True):
        same(got, exp)
    for got, exp in zip(s.vector_search(qids, k, "multi"), gold["service_multi"], strict=True):
        same(got, exp)
    same(s.vector_search_by_embedding([float(x) for x in g["Q"][2]], k), gold["service_by_embedding"])
    s.close()
    p = Mi355ImageVectorSearchRetrievalPipeline(lambda: store, "img", search_mode="multi")
    same(asyncio.run(p._retrieve_by_id("q2", k)), gold["image_pipeline_multi_q2"])
    stats = p.run(top_k=4)
    assert stats["total_queries"] == 6 and stats["failed_queries"] == ["q_noemb"]
    p.close()
    p1 = Mi355VectorSearchRetrievalPipeline(lambda: store, "txt", search_mode="single")
    same(asyncio.run(p1._retrieve_by_id("q1", k)), gold["pipeline_single_q1"])
    p1.close()
    # HEAVEN: stage 1 (single-vector search) and stage 2 (candidate MaxSim) both on the GPU
    from autorag_research_amd.heaven import Mi355HEAVENRetrievalPipeline

    ph = Mi355HEAVENRetrievalPipeline(lambda: store, "heaven", **gold["heaven_config"],
                                      pos_tagger=lambda toks: [(t, "NN" if len(t) % 2 == 0 else "VB") for t in toks])
    for qid, exp in gold["heaven"].items():
        got = asyncio.run(ph._retrieve_by_id(qid, k))
        assert [r["doc_id"] for r in got] == [r["doc_id"] for r in exp]
        assert np.allclose([r["score"] for r in got], [r["score"] for r in exp], rtol=0, atol=1e-6)
    ph.close()


def test_maxsim_subset_matches_oracle_and_heaven_golden(pkg, oracle):
    """candidate re-scoring (HEAVEN stage 2 / GQR pools): exact MaxSim distance for explicit (query, doc) lists ==
    oracle bit for bit; on the golden inputs -distance/n_q == the reference's HEAVEN `_score_candidates` (float64)."""
    g = np.load(GOLDEN / "scores_golden.npz")
    tok, off, qtok, qoff = g["ms_tok"], g["ms_offsets"], g["ms_qtok"], g["ms_qoff"]
    n_docs, B = off.shape[0] - 1, qoff.shape[0] - 1
    with pkg.Mi355Index(tok.shape[1]) as idx:
        idx.add_multivec(tok, off)
        ids = np.tile(np.arange(n_docs, dtype=np.int64), (B, 1))
        dist = idx.maxsim_subset(qtok, qoff, ids)
        for b in range(B):
            nq = qoff[b + 1] - qoff[b]
            assert np.abs(-dist[b].astype(np.float64) / nq - g["ms_scores_heaven"][b]).max() <= 1e-6
    rng = np.random.default_rng(31)
    d = 96
    tok, off = _ragged(rng, 700, d, 0, 70, unit=False)  # includes docs without vectors
    qtok, qoff = _queries(rng, [5, 0, 33, 128, 1], d, unit=False)
    B, m = 5, 57
    ids = rng.integers(0, 700, size=(B, m)).astype(np.int64)
    ids[0, :3] = [-1, 700, 10**12]  # not rows of the store: skipped
    ids[2, 5] = ids[2, 6]           # duplicates are fine
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok[: off[300]], off[:301])
        idx.add_multivec(tok[off[300]:], off[300:] - off[300])
        dist = idx.maxsim_subset(qtok, qoff, ids)
        idx.set_option("row_offset", 5000)  # shard: ids are global rows
        dist_off = idx.maxsim_subset(qtok, qoff, ids + 5000)
    assert np.array_equal(dist.view(np.uint32), dist_off.view(np.uint32))
    for b in range(B):
        q = qtok[qoff[b]:qoff[b + 1]]
        for j in range(m):
            i = ids[b, j]
            empty = not (0 <= i < 700) or off[i + 1] == off[i] or q.shape[0] == 0
            if empty:
                assert np.isnan(dist[b, j]), (b, j)
            else:
                exp = np.float32(oracle.maxsim_distance(tok[off[i]:off[i + 1]], q))
                assert dist[b, j].view(np.uint32) == exp.view(np.uint32), (b, j)


@pytest.mark.parametrize("tmin,tmax,n_docs", [(20, 150, 3000), (1030, 1030, 120)])
def test_sixteen_queries_ride_one_screen_pass(pkg, oracle, tmin, tmax, n_docs):
    """round 4: one pass of the bf16 screen over the token store serves up to FOUR groups of <= 4 queries (16 column blocks =
    128 KiB of query fragments in LDS, 8 waves per workgroup beyond 8 blocks).  Every number of groups per pass must give the
    oracle's lists bit for bit -- with ragged query lengths (a query may start anywhere in a column block and span blocks), a
    zero-length query, a query longer than a launch stages (it takes the tile path and cuts the pass), a non-finite query (it
    cuts the pass and takes the exact scan), duplicates of one query in different groups, and a pass that ends mid-group."""
    rng = np.random.default_rng(41)
    d = 128
    tok, off = _ragged(rng, n_docs, d, tmin, tmax)
    lens = [32, 24, 7, 32, 31, 1, 33, 32, 32, 32, 32, 32, 24, 24, 24, 24, 24, 24, 0, 32, 200, 32, 5, 32, 32, 17, 32, 32, 32, 9,
            32, 32, 32, 32, 32, 32, 32]
    qtok, qoff = _queries(rng, lens, d)
    qtok[qoff[10]:qoff[11]] = qtok[qoff[3]:qoff[4]]            # the same 32-vector query in two groups of a pass
    bad = qoff[26]
    qtok[bad + 3, 5] = np.inf                                   # a non-finite query: the pass must not take its neighbours down
    k = 10
    rd, rr = oracle.maxsim_topk(tok, off, qtok, qoff, k)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        for groups, wg, bps in ((4, -1, 4), (1, -1, 4), (3, 1, 2), (2, 0, 4), (4, 2, 4), (4, 0, 4), (4, 1, 4), (4, 2, 2), (4, 1, 2)):
            idx.set_option("maxsim_tighten", int(bps == 4))   # the candidate band narrowed by the starter's exact distances / the 2E band
            idx.set_option("maxsim_wg_pipe", int(wg != 2))   # the software-pipelined form / the plain one
            idx.set_option("maxsim_pass_groups", groups)
            idx.set_option("maxsim_wg", wg)   # -1 by document length / 1 parked / 2 immediate epilogue / 0 one wave per document
            idx.set_option("maxsim_wg_bps", bps)   # 32-token blocks per ring stage of the workgroup form
            idx.reset_stats()
            dist, rows = idx.search_maxsim(qtok, qoff, k)
            live = np.asarray(lens) > 0   # (a query without vectors: the reference returns [] before any SQL, base.py:506-507)
            assert (rows[~live] == -1).all() and np.isnan(dist[~live]).all()
            assert np.array_equal(rows[live], rr[live]), groups
            ok = ~np.isnan(rd) & live[:, None]
            assert np.array_equal(np.isnan(dist[live]), np.isnan(rd[live])), groups
            assert np.array_equal(dist[ok].view(np.uint32), rd[ok].view(np.uint32)), groups
            if groups == 4 and wg == -1:
                # all but the empty one, the long one and the GROUP of the non-finite one (its three neighbours take the exact scan with it)
                assert idx.stat("maxsim_screened") >= len(lens) - 6
        with pytest.raises(pkg.NativeError):
            idx.set_option("maxsim_pass_groups", 5)
        with pytest.raises(pkg.NativeError):
            idx.set_option("maxsim_wg_bps", 3)
        # exactly 8 column blocks (8 queries of 32 vectors): the workgroup form for short documents, either epilogue
        q8, o8 = _queries(rng, [32] * 8, d)
        r8d, r8r = oracle.maxsim_topk(tok, off, q8, o8, k)
        for wg, wmin in ((1, 8), (2, 8), (-1, 8), (-1, 9)):
            idx.set_option("maxsim_wg", wg)
            idx.set_option("maxsim_wg_min", wmin)
            dist, rows = idx.search_maxsim(q8, o8, k)
            assert np.array_equal(rows, r8r) and np.array_equal(dist.view(np.uint32), r8d.view(np.uint32)), (wg, wmin)
        idx.set_option("maxsim_wg", -1)
        # every query exactly one column block (ColBERT's 32-vector queries; the last one shorter): a wave sums its own two queries
        qa_, oa_ = _queries(rng, [32] * 15 + [17], d)
        rad, rar = oracle.maxsim_topk(tok, off, qa_, oa_, k)
        for aligned, wg in ((1, 1), (0, 1), (1, 2), (1, -1)):
            idx.set_option("maxsim_aligned", aligned)
            idx.set_option("maxsim_wg", wg)
            dist, rows = idx.search_maxsim(qa_, oa_, k)
            assert np.array_equal(rows, rar) and np.array_equal(dist.view(np.uint32), rad.view(np.uint32)), (aligned, wg)
        idx.set_option("maxsim_wg", -1)
        idx.set_option("maxsim_aligned", 1)
        # k above the fast path's 64: one group per pass, same answers
        idx.set_option("maxsim_pass_groups", 4)
        rd2, rr2 = oracle.maxsim_topk(tok, off, qtok[: qoff[9]], qoff[:10], 70)
        d2, r2 = idx.search_maxsim(qtok[: qoff[9]], qoff[:10], 70)
        assert np.array_equal(r2, rr2) and np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))


@pytest.mark.parametrize("shape", ["tiny_docs", "ragged", "long"])
def test_sixteen_query_screen_over_hard_document_shapes(pkg, oracle, shape):
    """The 16-query workgroup screen over document shapes that stress its per-document bookkeeping: thousands of documents of
    0..5 tokens (a document ends at almost every block: more documents between two ring barriers than the usual store has),
    ragged documents with a tenth of them empty, 1000-token pages -- and a store grown by a second add between two searches.
    Same answers as the oracle, bit for bit.  (These shapes were written for round 4's packed bf16 copy, k_maxsim_wgp.h: measured
    2-3 % slower than the padded copy then, removed in round 6; the shapes stay.)"""
    rng = np.random.default_rng({"tiny_docs": 11, "ragged": 12, "long": 13}[shape])
    d = 128
    if shape == "tiny_docs":
        lens = rng.integers(0, 6, size=9000)
    elif shape == "ragged":
        lens = rng.integers(0, 200, size=3000)
        lens[rng.random(3000) < 0.1] = 0
    else:
        lens = rng.integers(900, 1100, size=120)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tok = rng.standard_normal((int(off[-1]), d)).astype(np.float32)
    tok /= np.maximum(np.linalg.norm(tok, axis=1, keepdims=True), 1e-9)
    qlens = [32, 24, 32, 17, 32, 32, 5, 32, 32, 32, 1, 32, 32, 31, 32, 32, 32, 32, 24]
    qtok, qoff = _queries(rng, qlens, d)
    k = 10
    rd, rr = oracle.maxsim_topk(tok, off, qtok, qoff, k)
    half = len(lens) // 2
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok[:off[half]], off[:half + 1])
        d0, r0 = idx.search_maxsim(qtok, qoff, k)
        rd0, rr0 = oracle.maxsim_topk(tok[:off[half]], off[:half + 1], qtok, qoff, k)
        assert np.array_equal(r0, rr0)
        idx.add_multivec(tok[off[half]:], off[half:] - off[half])
        idx.reset_stats()
        dist, rows = idx.search_maxsim(qtok, qoff, k)
        assert idx.stat("maxsim_screened") >= len(qlens) - 1 and idx.stat("maxsim_fallbacks") == 0
        assert np.array_equal(rows, rr)
        ok = ~np.isnan(rd)
        assert np.array_equal(np.isnan(dist), np.isnan(rd))
        assert np.array_equal(dist[ok].view(np.uint32), rd[ok].view(np.uint32))
        idx.set_option("maxsim_wg", 0)                        # one wave per document: the same lists
        dist2, rows2 = idx.search_maxsim(qtok, qoff, k)
        assert np.array_equal(rows2, rows) and np.array_equal(dist2.view(np.uint32), dist.view(np.uint32))
