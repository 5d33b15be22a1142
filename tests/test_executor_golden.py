"""The reference's own plugin registry and Executor drove this plugin in the build container (tests/golden/make_golden.py:
make_executor -> executor_golden.json).  Here the same flow -- construct the pipeline the way executor.py:326-333 / 408-416
does, health-check with query_limit (then clean up), full run, completion check -- is replayed without the reference and
must leave the same stats and the same persisted rows.  CPU: oracle-backed index; GPU: the real one."""

import json

import numpy as np
import pytest

from helpers import GOLDEN, FakeRefService, FakeSessionmaker, OracleIndex, build_golden_stores, ref_tables_from_store

GOLD = json.loads((GOLDEN / "executor_golden.json").read_text())


def _store():
    store, _ = build_golden_stores()
    del store.queries["q_noemb"]
    store.query_order.remove("q_noemb")
    return store


def _executor_flow(session_factory, rows_of, verify):
    from autorag_research_amd.pipelines import Mi355VectorSearchPipelineConfig

    cfg = Mi355VectorSearchPipelineConfig(name="mi355_vector_search", search_mode="single", top_k=4, batch_size=4, retry_delay=0.0)
    # health check (executor.py:308-354): temporary pipeline, query_limit, cleanup
    hc = cfg.get_pipeline_class()(session_factory=session_factory, name=f"{cfg.name}_health_check", schema=None,
                                  **cfg.get_pipeline_kwargs())
    r = hc.run(**{**cfg.get_run_kwargs(), "query_limit": 2})
    assert r["total_queries"] == 2 and r["failed_queries"] == [] and r["pipeline_id"] == hc.pipeline_id
    assert hc._service.delete_pipeline_results(hc.pipeline_id) == 8
    hc.close()
    left = len(rows_of())
    p = cfg.get_pipeline_class()(session_factory=session_factory, name=cfg.name, schema=None, **cfg.get_pipeline_kwargs())
    run = p.run(**cfg.get_run_kwargs())
    assert verify(p.pipeline_id)
    p.close()
    return left, run, sorted(rows_of(), key=lambda r: (str(r[0]), -r[2], str(r[1])))


def _check(kind, left, run, rows):
    g = GOLD[kind]
    assert left == g["rows_left_by_health_check"] == 0
    pr = g["pipeline_result"]
    assert pr["success"] and pr["retries_used"] == 0 and pr["pipeline_type"] == "retrieval"
    assert run["total_queries"] == pr["total_queries"] == 6 and run["failed_queries"] == []
    assert run["total_results"] == len(g["persisted"]) == len(rows)
    assert [[q, c] for q, c, _ in rows] == [[q, c] for q, c, _ in g["persisted"]]
    assert np.allclose([s for *_, s in rows], [s for *_, s in g["persisted"]], rtol=0, atol=1e-12)


def test_registry_scan_recorded_by_the_reference():
    names = {"mi355_vector_search", "mi355_image_vector_search", "mi355_heaven", "mi355_gqr_hybrid", "mi355_hybrid_rrf",
             "mi355_hybrid_cc", "mi355_hyde"}
    assert {r[0] for r in GOLD["registry_scan"]} == names
    assert all(r[1:] == ["retrieval", "pipelines", "mi355_vector_search"] for r in GOLD["registry_scan"])


def _flows(monkeypatch):
    import sys
    import types

    store = _store()
    yield "store_factory", (lambda: store), (lambda: [(q, c, s) for (pid, q), lst in store.chunk_results.items() for c, s in lst]), \
        (lambda pid: all(store.chunk_results.get((pid, q)) for q in store.query_order))
    tables = ref_tables_from_store(_store())
    mod = types.ModuleType("autorag_research.orm.service.retrieval_pipeline")
    mod.RetrievalPipelineService = lambda sf, schema=None: FakeRefService(sf, schema)
    for name in ("autorag_research", "autorag_research.orm", "autorag_research.orm.service"):
        monkeypatch.setitem(sys.modules, name, sys.modules.get(name) or types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "autorag_research.orm.service.retrieval_pipeline", mod)
    yield "sessionmaker", FakeSessionmaker(tables), \
        (lambda: [(r["query_id"], r["chunk_id"], r["rel_score"]) for r in tables["chunk_results"]]), \
        FakeRefService(tables=tables).verify_pipeline_completion


def test_executor_flow_on_cpu(monkeypatch, oracle):
    import autorag_research_amd.service as svc

    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    for kind, sf, rows_of, verify in _flows(monkeypatch):
        _check(kind, *_executor_flow(sf, rows_of, verify))


@pytest.mark.gpu
def test_executor_flow_on_gpu(monkeypatch, native_built):
    for kind, sf, rows_of, verify in _flows(monkeypatch):
        _check(kind, *_executor_flow(sf, rows_of, verify))
