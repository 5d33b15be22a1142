"""CPU: the plugin behind a SQLAlchemy-style `session_factory`, the way the reference Executor and its wrapper pipelines
construct it (executor.py:326-333, 408-416; hybrid.py `_load_pipeline`): Mi355RetrievalService detects that
`session_factory()` is not a store and goes through the reference's RetrievalPipelineService -- here a duck-typed stand-in
with the same repositories (tests/helpers.py: FakeRefService) -- via store.UowStore."""

import sys
import types

import numpy as np
import pytest

from helpers import FakeRefService, FakeSessionmaker, OracleIndex, build_golden_stores, load_service_golden, ref_tables_from_store


@pytest.fixture()
def ref_env(monkeypatch, oracle):
    import autorag_research_amd.service as svc

    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    store, g = build_golden_stores()
    tables = ref_tables_from_store(store)
    made = []

    def make(session_factory, schema=None):
        s = FakeRefService(session_factory, schema)
        made.append(s)
        return s

    mod = types.ModuleType("autorag_research.orm.service.retrieval_pipeline")
    mod.RetrievalPipelineService = make
    for name in ("autorag_research", "autorag_research.orm", "autorag_research.orm.service"):
        monkeypatch.setitem(sys.modules, name, sys.modules.get(name) or types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "autorag_research.orm.service.retrieval_pipeline", mod)
    return store, tables, FakeSessionmaker(tables), made


def test_pipeline_runs_over_a_sessionmaker_like_the_executor_passes(ref_env):
    from autorag_research_amd.pipelines import Mi355VectorSearchPipelineConfig

    store, tables, sessionmaker, made = ref_env
    gold = load_service_golden()
    cfg = Mi355VectorSearchPipelineConfig(name="mi355_vs", search_mode="single", top_k=gold["top_k"], batch_size=4)
    # exactly the Executor's construction (executor.py:408-416)
    p = cfg.get_pipeline_class()(session_factory=sessionmaker, name=cfg.name, schema=None, **cfg.get_pipeline_kwargs())
    assert len(made) == 1 and made[0].session_factory is sessionmaker and sessionmaker.sessions == 1
    stats = p.run(**{**cfg.get_run_kwargs(), "retry_delay": 0.0})
    assert stats["pipeline_id"] == p.pipeline_id and stats["total_queries"] == 6 and stats["failed_queries"] == ["q_noemb"]
    assert stats["total_results"] == 6 * gold["top_k"] == len(tables["chunk_results"])
    # rows went through the reference's result repository in the reference's row format, ranked like its own service ranks
    by_q = {}
    for r in tables["chunk_results"]:
        assert r["pipeline_id"] == p.pipeline_id
        by_q.setdefault(r["query_id"], []).append((r["chunk_id"], r["rel_score"]))
    for i, exp in enumerate(gold["service_single"]):
        got = by_q[f"q{i}"]
        assert [d for d, _ in got] == [e["doc_id"] for e in exp]
        assert np.allclose([s for _, s in got], [e["score"] for e in exp], rtol=0, atol=1e-12)
    # resume: a second run finds every query completed (failed one retried, still failing) and inserts nothing new
    again = p.run(**{**cfg.get_run_kwargs(), "retry_delay": 0.0})
    assert again["total_results"] == 0 and again["failed_queries"] == ["q_noemb"]
    # the Executor's health-check cleanup path
    assert p._service.delete_pipeline_results(p.pipeline_id) == 6 * gold["top_k"] and tables["chunk_results"] == []
    p.close()


def test_per_query_contract_and_multi_vector_over_the_uow(ref_env):
    import asyncio

    from autorag_research_amd.pipelines import Mi355ImageVectorSearchRetrievalPipeline, Mi355VectorSearchRetrievalPipeline

    store, tables, sessionmaker, _ = ref_env
    gold = load_service_golden()
    k = gold["top_k"]

    def same(got, exp):
        assert [r["doc_id"] for r in got] == [r["doc_id"] for r in exp]
        assert [r["content"] for r in got] == [r["content"] for r in exp]
        assert np.allclose([r["score"] for r in got], [r["score"] for r in exp], rtol=0, atol=1e-12)

    p = Mi355VectorSearchRetrievalPipeline(sessionmaker, "p_multi", search_mode="multi")
    same(asyncio.run(p._retrieve_by_id("q1", k)), gold["pipeline_multi_q1"])
    same(asyncio.run(p.retrieve("query text 1", k)), gold["pipeline_multi_q1"])  # find_by_contents -> by id
    pi = Mi355ImageVectorSearchRetrievalPipeline(sessionmaker, "p_img", search_mode="single")
    same(asyncio.run(pi._retrieve_by_id("q2", k)), gold["image_pipeline_single_q2"])
    with pytest.raises(ValueError, match="Query nope not found"):
        asyncio.run(p._retrieve_by_id("nope", k))


def test_a_store_factory_is_still_taken_as_is(oracle, monkeypatch):
    import autorag_research_amd.service as svc

    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    store, _ = build_golden_stores()
    s = svc.Mi355RetrievalService(lambda: store)
    assert s._uow_store is None and s._store() is store


def test_block_failure_falls_back_to_the_per_query_path(oracle, monkeypatch):
    """an exception inside a block must not abort run(); the page is retried query by query, so only the bad
    query lands in failed_queries (reference retry / isolation semantics, retrieval_pipeline.py:222-236)."""
    import autorag_research_amd.service as svc
    from autorag_research_amd.pipelines import Mi355VectorSearchRetrievalPipeline

    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    store, _ = build_golden_stores()
    p = Mi355VectorSearchRetrievalPipeline(lambda: store, "p_iso", search_mode="single")
    calls = {"block": 0}
    real = p._service.vector_search

    def flaky(query_ids, top_k=10, search_mode="single", unit="chunk"):
        if len(query_ids) > 1:
            calls["block"] += 1
            raise RuntimeError("one malformed embedding poisons the whole block")
        if query_ids[0] == "q3":
            raise ValueError("Query q3 is malformed")
        return real(query_ids, top_k, search_mode, unit)

    monkeypatch.setattr(p._service, "vector_search", flaky)
    stats = p.run(top_k=3, batch_size=4, max_retries=2, retry_delay=0.0)
    assert calls["block"] >= 1
    assert sorted(stats["failed_queries"]) == ["q3", "q_noemb"] and stats["total_queries"] == 5 and stats["total_results"] == 15


def test_export_pages_by_ordered_primary_key():
    """A repository with the reference's `get_all_ids` (ORDER BY id) + `get_by_ids` (an IN query: any order) is exported by
    key -- every row once, in id order -- even when its unordered `get_all` pages would repeat and skip rows; duplicate keys
    from an unordered fallback are refused."""
    from autorag_research_amd.store import UowStore

    rng = np.random.default_rng(0)
    rows = [types.SimpleNamespace(id=i, contents=f"c{i}", embedding=rng.standard_normal(4).astype(np.float32), embeddings=None)
            for i in range(23)]

    class Repo:
        def __init__(self, keyed):
            self.calls = 0
            if keyed:
                self.get_all_ids = lambda limit=None, offset=0: sorted(r.id for r in rows)[offset:None if limit is None else offset + limit]
                self.get_by_ids = lambda ids: [r for r in reversed(rows) if r.id in set(ids)]  # IN (...): arbitrary order

        def get_all(self, limit=None, offset=None):  # unordered paging: every page starts over from a shuffled table
            self.calls += 1
            order = np.random.default_rng(self.calls).permutation(len(rows))
            return [rows[i] for i in order][offset or 0:(offset or 0) + limit]

    class Uow:
        def __init__(self, repo):
            self.chunks = repo

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    for keyed in (True, False):
        repo = Repo(keyed)
        svc = types.SimpleNamespace(_create_uow=lambda repo=repo: Uow(repo))
        st = UowStore(svc)
        st.EXPORT_PAGE = 5
        if keyed:
            t = st.chunks
            assert t.ids == list(range(23)) and repo.calls == 0
            assert np.array_equal(t.embedding, np.stack([r.embedding for r in rows]))
        else:
            with pytest.raises(RuntimeError, match="duplicate primary keys"):
                st.chunks
