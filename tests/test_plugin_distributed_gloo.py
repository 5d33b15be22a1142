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


def _fault_worker(rank: int, world: int, port: int, out_dir: str):
    """Rank-local faults under a _World: every rank must leave through the same door."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist

    import autorag_research_amd.service as svc
    from helpers import OracleIndex, build_golden_stores
    from autorag_research_amd.pipelines import Mi355VectorSearchPipelineConfig

    svc.Mi355Index = OracleIndex
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    cfg = Mi355VectorSearchPipelineConfig(name="faults", search_mode="single", top_k=4, batch_size=4, retry_delay=0.0)

    def make(store, name="faults"):
        return cfg.get_pipeline_class()(session_factory=lambda: store, name=name, schema=None, **cfg.get_pipeline_kwargs())

    def golden_store():
        store, _ = build_golden_stores()
        del store.queries["q_noemb"]
        store.query_order.remove("q_noemb")
        return store

    # (1) rank 0's database call fails inside from_root: the SAME exception on every rank, nobody left in the broadcast
    store = golden_store()
    if rank == 0:
        def boom(name, config):
            raise KeyError("pipeline table is gone")
        store.get_or_create_pipeline = boom
    try:
        make(store)
        out["from_root"] = "no error"
    except KeyError as e:
        out["from_root"] = f"KeyError:{e.args[0]}"
    # (2) the ranks exported the table in different orders: refused on every rank before anything is sharded
    store = golden_store()
    if rank == 1:
        t = store.chunks
        t.ids[0], t.ids[1] = t.ids[1], t.ids[0]
    p = make(store, "order")
    try:
        p._service._unit("chunk")
        out["digest"] = "no error"
    except RuntimeError as e:
        out["digest"] = "refused" if "different tables" in str(e) else str(e)
    p.close()
    # (3) + (4) rank 1 alone fails AFTER its block / its first per-query attempt: both ranks fall back / retry together
    store = golden_store()
    p = make(store, "flaky")
    calls = {"block": 0, "one": 0}
    real_block, real_one = p._retrieve_block, p._retrieve_by_id

    def flaky_block(qids, k):
        res = real_block(qids, k)
        calls["block"] += 1
        if rank == 1 and calls["block"] == 1:
            raise MemoryError("host conversion of the page failed on this rank")
        return res

    async def flaky_one(qid, k):
        res = await real_one(qid, k)
        calls["one"] += 1
        if rank == 1 and calls["one"] == 1:
            raise MemoryError("first attempt failed on this rank")
        return res

    p._retrieve_block, p._retrieve_by_id = flaky_block, flaky_one
    out["run"] = p.run(**cfg.get_run_kwargs())
    out["calls"] = calls
    out["rows"] = sorted(((q, c, s) for (pid, q), lst in store.chunk_results.items() for c, s in lst),
                         key=lambda t: (str(t[0]), -t[2], str(t[1])))
    p.close()
    with open(os.path.join(out_dir, f"f{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_local_faults_do_not_split_the_ranks(tmp_path, oracle):
    import torch.multiprocessing as mp

    from helpers import GOLDEN

    world, port = 2, _free_port()
    mp.spawn(_fault_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (json.loads((tmp_path / f"f{r}.json").read_text()) for r in range(world))
    gold_exec = json.loads((GOLDEN / "executor_golden.json").read_text())["store_factory"]
    for r in (r0, r1):
        assert r["from_root"] == "KeyError:pipeline table is gone"
        assert r["digest"] == "refused"
        assert r["run"]["total_queries"] == 6 and r["run"]["failed_queries"] == []
        # page 1: the block failed on rank 1 -> both ranks answered its 4 queries one by one, the first of them twice
        # (rank 1's first attempt failed); page 2 went through as a block on both
        assert r["calls"] == {"block": 2, "one": 5}
    assert r1["rows"] == []
    assert [[q, c] for q, c, _ in r0["rows"]] == [[q, c] for q, c, _ in gold_exec["persisted"]]


def _fake_tagger(tokens):
    """(the deterministic POS stand-in tests/golden/make_golden.py patched into the reference: nltk is not installed)"""
    return [(t, "NN" if len(t) % 2 == 0 else "VB") for t in tokens]


def _rescore_worker(rank: int, world: int, port: int, out_dir: str):
    """The candidate re-scorers behind a _World: HEAVEN stage 2 = `maxsim_subset` over the TOKEN-SHARDED
    store (every rank scores the candidates it owns, one all-gather of [1, m] fp32, the owner's value wins); GQR = the pools'
    vectors staged into a scratch store per page.  No rank builds a whole-table index."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import asyncio

    import torch.distributed as dist

    import autorag_research_amd.service as svc
    from helpers import OracleIndex, build_golden_stores, check_gqr_flow, load_service_golden
    from autorag_research_amd.heaven import Mi355HEAVENPipelineConfig, Mi355HEAVENRetrievalPipeline
    from autorag_research_amd.pipelines import Mi355VectorSearchRetrievalPipeline

    svc.Mi355Index = OracleIndex
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store, _ = build_golden_stores()
    gold = load_service_golden()
    out = {}
    cfg = Mi355HEAVENPipelineConfig(name="heaven", **gold["heaven_config"])
    p = Mi355HEAVENRetrievalPipeline(lambda: store, "heaven", pos_tagger=_fake_tagger,
                                     **{k: v for k, v in cfg.get_pipeline_kwargs().items() if "model" not in k})
    out["heaven"] = {qid: asyncio.run(p._retrieve_by_id(qid, gold["top_k"])) for qid in gold["heaven"]}
    u = p._service._unit("image_chunk")
    out["heaven_whole_table_indices"] = int(u.single is not None) + int(u.multi is not None)
    out["heaven_shard_docs"] = u.multi_sharded.index.n_docs()
    out["heaven_shard_rows"] = len(u.single_sharded.index)
    p.close()
    # GQR: every golden case (embedding, multi-vector, forced-single and score-space branches), per query and as one block
    services = []

    def make_primary(mode):
        c = Mi355VectorSearchRetrievalPipeline(lambda: store, f"vs_{mode}", search_mode=mode)
        services.append(c._service)
        return c

    import autorag_research_amd.gqr as gqr_mod

    made = []
    real_init = gqr_mod.Mi355GQRHybridRetrievalPipeline.__init__

    def spy_init(self, *a, **kw):
        real_init(self, *a, **kw)
        made.append(self._service)

    gqr_mod.Mi355GQRHybridRetrievalPipeline.__init__ = spy_init
    # (units are closed with their pipeline: count whole-table indices at close time)
    whole = {"n": 0}
    real_close = svc._UnitIndex.close

    def spy_close(self):
        whole["n"] += int(self.single is not None) + int(self.multi is not None)
        real_close(self)

    svc._UnitIndex.close = spy_close
    check_gqr_flow(store, make_primary, atol=1e-9)
    for s in services:
        s.close()
    out["gqr_services"] = len(made)
    out["gqr_whole_table_indices"] = whole["n"]
    with open(os.path.join(out_dir, f"s{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_candidate_rescorers_run_on_the_sharded_stores(tmp_path, oracle):
    import torch.multiprocessing as mp

    from helpers import build_golden_stores, load_service_golden

    world, port = 2, _free_port()
    mp.spawn(_rescore_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (json.loads((tmp_path / f"s{r}.json").read_text()) for r in range(world))
    gold = load_service_golden()
    store, _ = build_golden_stores()
    n_docs = len(store.image_chunks.ids)
    # each rank holds about half of the tokens / rows and no rank holds a whole-table index
    assert r0["heaven_shard_docs"] + r1["heaven_shard_docs"] == n_docs and 0 < r0["heaven_shard_docs"] < n_docs
    assert 0 < r0["heaven_shard_rows"] < n_docs
    for r in (r0, r1):
        assert r["heaven_whole_table_indices"] == 0 and r["gqr_whole_table_indices"] == 0 and r["gqr_services"] > 0
        for qid, exp in gold["heaven"].items():     # the reference's _retrieve_by_id dicts, on every rank
            got = r["heaven"][qid]
            assert [x["doc_id"] for x in got] == [e["doc_id"] for e in exp]
            assert np.allclose([x["score"] for x in got], [e["score"] for e in exp], rtol=0, atol=1e-6)


def _hyde_worker(rank: int, world: int, port: int, out_dir: str):
    """HyDE under a _World: the LLM samples and may fail per rank -- rank 0 alone generates and embeds, every
    rank searches the same vectors, and a rank-local LLM fault on rank 1 (which never calls its LLM) or a flaky LLM on rank 0
    (retried inside the page, no collective per retry) leaves both ranks in step."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import asyncio

    import torch.distributed as dist

    import autorag_research_amd.service as svc
    from helpers import OracleIndex, build_golden_stores
    from autorag_research_amd.hyde import Mi355HyDEPipelineConfig

    svc.Mi355Index = OracleIndex
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store, _ = build_golden_stores()
    del store.queries["q_noemb"]
    store.query_order.remove("q_noemb")
    calls = {"llm": 0, "embed": 0}

    class LLM:
        """rank 1's model is broken outright; rank 0's fails on its first call for q2 and then answers -- with a rank-dependent
        text, so that ranks that each asked their own model would search different vectors"""

        def __init__(self):
            self.seen = set()

        async def ainvoke(self, prompt):
            calls["llm"] += 1
            if rank == 1:
                raise RuntimeError("this rank's LLM endpoint is down")
            q = prompt.split("Question: ")[1].split("\n")[0]
            if q == "query text 2" and q not in self.seen:
                self.seen.add(q)
                raise TimeoutError("transient")
            return f"passage written on rank {rank} about: {q}"

    class Emb:
        async def aembed_query(self, text):
            calls["embed"] += 1
            seed = sum((i + 1) * b for i, b in enumerate(text.encode())) % (2**32)
            return [float(x) for x in np.random.default_rng(seed).standard_normal(32).astype(np.float32)]

    cfg = Mi355HyDEPipelineConfig(name="hyde_world", llm=LLM(), embedding=Emb(), top_k=3, batch_size=4, max_retries=3, retry_delay=0.0)
    p = cfg.get_pipeline_class()(session_factory=lambda: store, name=cfg.name, schema=None, **cfg.get_pipeline_kwargs())
    out = {"run": p.run(**cfg.get_run_kwargs())}
    out["rows"] = sorted(((str(q), c, s) for (pid, q), lst in store.chunk_results.items() for c, s in lst))
    out["calls_after_run"] = dict(calls)
    out["by_text"] = asyncio.run(p.retrieve("a question nobody stored", top_k=3))
    out["by_id"] = asyncio.run(p._retrieve_by_id("q1", 3))
    out["calls"] = dict(calls)
    p.close()
    with open(os.path.join(out_dir, f"h{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_hyde_generates_on_rank_zero_only_and_survives_rank_local_llm_faults(tmp_path, oracle):
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_hyde_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (json.loads((tmp_path / f"h{r}.json").read_text()) for r in range(world))
    for r in (r0, r1):
        assert r["run"]["total_queries"] == 6 and r["run"]["failed_queries"] == [] and r["run"]["total_results"] == 18
    assert r1["rows"] == [] and len(r0["rows"]) == 18        # rank 0 alone persists
    # rank 1's broken LLM was never asked; rank 0 asked 6 + 1 retry in the run, then once per ad-hoc call
    assert r1["calls"] == {"llm": 0, "embed": 0}
    assert r0["calls_after_run"] == {"llm": 7, "embed": 6} and r0["calls"] == {"llm": 9, "embed": 8}
    # every rank returned the SAME ad-hoc answers (rank 0's passage, rank 0's vector, all shards searched)
    assert r0["by_text"] == r1["by_text"] and len(r0["by_text"]) == 3
    assert r0["by_id"] == r1["by_id"] and len(r0["by_id"]) == 3
    # and they are the exact top-3 of rank 0's vectors over the WHOLE table
    from helpers import build_golden_stores

    store, g = build_golden_stores()
    text = "passage written on rank 0 about: query text 1"
    seed = sum((i + 1) * b for i, b in enumerate(text.encode())) % (2**32)
    v = np.random.default_rng(seed).standard_normal(32).astype(np.float32)
    od, orow = oracle.topk_search(g["C"], v[None, :], 3)
    assert [d["doc_id"] for d in r0["by_id"]] == [g["ids"][int(i)] for i in orow[0]]
    assert [d["score"] for d in r0["by_id"]] == [1.0 - float(x) for x in od[0]]


def _deadline_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), MI355DR_COLLECTIVE_TIMEOUT_S="3")
    import time

    import torch.distributed as dist

    import autorag_research_amd.service as svc
    from helpers import OracleIndex, build_golden_stores

    svc.Mi355Index = OracleIndex
    dist.init_process_group("gloo", rank=rank, world_size=world)
    store, _ = build_golden_stores()
    s = svc.Mi355RetrievalService(lambda: store)            # (every rank: creates the deadline group together)
    w = s._world
    assert w is not None and w.group is not None
    assert w.agree(True) and not w.agree(rank == 0)          # the plugin's collectives work over the group
    res = s.vector_search(["q1", "q2"], 3)                   # ... and so does the sharded search behind it
    assert len(res) == 2 and len(res[0]) == 3
    out = {"rank": rank}
    if rank == 0:
        t0 = time.time()
        try:
            w.agree(True)                                     # rank 1 never comes: a deadline, not a hang
            out["outcome"] = "returned"
        except Exception as e:  # noqa: BLE001
            out["outcome"] = type(e).__name__
        out["waited_s"] = time.time() - t0
    else:
        time.sleep(8.0)                                       # "died" outside the collective its peer is in
    (Path(out_dir) / f"deadline{rank}.json").write_text(json.dumps(out))
    os._exit(0)   # (the group is poisoned by the timeout: no orderly destroy)


def test_collective_deadline_turns_a_lost_rank_into_an_error(tmp_path, oracle):
    """`_World.agree` cannot reconcile a rank that is lost while its peers are inside a collective.  With
    MI355DR_COLLECTIVE_TIMEOUT_S the plugin's collectives (and its sharded searchers') run over a group of their own with that
    deadline: the waiting rank gets an exception after ~3 s instead of the launcher's default of 30 minutes."""
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_deadline_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = json.loads((tmp_path / "deadline0.json").read_text())
    assert r0["outcome"] != "returned" and 2.0 < r0["waited_s"] < 7.5, r0
    assert json.loads((tmp_path / "deadline1.json").read_text())["rank"] == 1
