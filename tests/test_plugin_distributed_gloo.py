"""CPU, world_size 2 over gloo: the PLUGIN boundary in its multi-GPU form.  Both ranks construct the same pipelines the way the
reference Executor does (executor.py:326-333, 408-416) and call `run()`; the service detects the process group, row-shards the
corpus over the ranks (the image table by cumulative token count), answers every page with all ranks together (local top-k ->
all-gather -> merge) and lets rank 0 alone read the page's query ids and persist.  What rank 0 stores must be what the
reference's own Executor run left behind (tests/golden/executor_golden.json) and what its service returns
(tests/golden/service_golden.json); the other rank stores nothing."""

import json
import os
import socket
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import autorag_research_amd.service as svc
    from helpers import OracleIndex, build_golden_stores, load_service_golden
    from autorag_research_amd.pipelines import Mi355ImageVectorSearchPipelineConfig, Mi355VectorSearchPipelineConfig

    svc.Mi355Index = OracleIndex  # CPU stand-in for the native handle (tests/helpers.py); the real one on the GPU box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    # ---- the Executor's flow (health check with query_limit -> cleanup -> full run) on the golden store
    store, _ = build_golden_stores()
    del store.queries["q_noemb"]
    store.query_order.remove("q_noemb")
    cfg = Mi355VectorSearchPipelineConfig(name="mi355_vector_search", search_mode="single", top_k=4, batch_size=4, retry_delay=0.0)
    hc = cfg.get_pipeline_class()(session_factory=lambda: store, name=f"{cfg.name}_health_check", schema=None,
                                  **cfg.get_pipeline_kwargs())
    assert hc._service._world is not None and hc._service._world.size == world and hc._service._device == rank
    r = hc.run(**{**cfg.get_run_kwargs(), "query_limit": 2})
    assert r["total_queries"] == 2 and r["failed_queries"] == [] and r["total_results"] == 8
    assert hc._service.delete_pipeline_results(hc.pipeline_id) == 8  # (rank 0's count, on every rank)
    hc.close()
    p = cfg.get_pipeline_class()(session_factory=lambda: store, name=cfg.name, schema=None, **cfg.get_pipeline_kwargs())
    run = p.run(**cfg.get_run_kwargs())
    u = p._service._unit("chunk")
    out["shard_rows"] = len(u.single_sharded.index)   # this rank holds only its share of the corpus
    out["run"] = run
    out["rows"] = sorted(((q, c, s) for (pid, q), lst in store.chunk_results.items() for c, s in lst),
                         key=lambda t: (str(t[0]), -t[2], str(t[1])))
    # resume: nothing left to do, decided by rank 0 for everybody
    again = p.run(**cfg.get_run_kwargs())
    assert again["total_queries"] == 0 and again["total_results"] == 0
    # ---- the service calls behind the pipelines, against the reference's own output dicts
    store2, g = build_golden_stores()
    gold = load_service_golden()
    p2 = cfg.get_pipeline_class()(session_factory=lambda: store2, name="svc", schema=None, **cfg.get_pipeline_kwargs())
    out["single"] = p2._service.vector_search([f"q{i}" for i in range(len(gold["service_single"]))], gold["top_k"], "single")
    icfg = Mi355ImageVectorSearchPipelineConfig(name="img", search_mode="multi", top_k=gold["top_k"], batch_size=4, retry_delay=0.0)
    ip = icfg.get_pipeline_class()(session_factory=lambda: store2, name=icfg.name, schema=None, **icfg.get_pipeline_kwargs())
    out["image_run"] = ip.run(**icfg.get_run_kwargs())
    out["image_rows"] = {str(q): lst for (pid, q), lst in store2.image_chunk_results.items()}
    iu = ip._service._unit("image_chunk")
    out["image_shard_docs"] = iu.multi_sharded.index.n_docs()
    p.close(), p2.close(), ip.close()
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_pipeline_run_row_sharded_over_two_ranks(tmp_path, oracle):
    import torch.multiprocessing as mp

    from helpers import GOLDEN, build_golden_stores, load_service_golden

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (json.loads((tmp_path / f"r{r}.json").read_text()) for r in range(world))
    gold_exec = json.loads((GOLDEN / "executor_golden.json").read_text())["store_factory"]
    gold = load_service_golden()
    store, _ = build_golden_stores()
    n_rows = int((~np.isnan(store.chunks.embedding).all(axis=1)).sum())
    # the corpus is split, not replicated
    assert r0["shard_rows"] + r1["shard_rows"] == n_rows and 0 < r0["shard_rows"] < n_rows
    # both ranks report the reference Executor's stats; rank 0 alone persisted, and persisted the reference's rows
    pr = gold_exec["pipeline_result"]
    for r in (r0, r1):
        assert r["run"]["total_queries"] == pr["total_queries"] == 6 and r["run"]["failed_queries"] == []
        assert r["run"]["total_results"] == len(gold_exec["persisted"])
    assert r1["rows"] == [] and r1["image_rows"] == {}
    assert [[q, c] for q, c, _ in r0["rows"]] == [[q, c] for q, c, _ in gold_exec["persisted"]]
    assert np.allclose([s for *_, s in r0["rows"]], [s for *_, s in gold_exec["persisted"]], rtol=0, atol=1e-12)
    # service results on every rank == the reference service's dicts
    for r in (r0, r1):
        for got, exp in zip(r["single"], gold["service_single"], strict=True):
            assert [x["doc_id"] for x in got] == [e["doc_id"] for e in exp]
            assert np.allclose([x["score"] for x in got], [e["score"] for e in exp], rtol=0, atol=1e-12)
            assert [x["content"] for x in got] == [e["content"] for e in exp]
    # the image pipeline (multi-vector, token-balanced shards): what rank 0 stored == the reference pipeline's lists
    assert r0["image_shard_docs"] + r1["image_shard_docs"] == len(store.image_chunks.ids)
    exp = gold["image_pipeline_multi_q2"]
    got = r0["image_rows"]["q2"]
    assert [c for c, _ in got] == [e["doc_id"] for e in exp]
    assert np.allclose([s for _, s in got], [e["score"] for e in exp], rtol=0, atol=1e-6)
