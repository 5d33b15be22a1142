"""SURVEY 8(a) row a12 against the reference itself: tests/golden/ingest_golden.json is `BaseIngestionService._embed_entities`
(orm/service/base_ingestion.py:326-495) run over a fake Unit of Work (make_golden.py:make_ingest).  `autorag_research_amd.ingest`
must reproduce, case by case: the return value, every row's final embedding, which payloads reached the embedding function,
and the sequence of fetches -- over the same kind of Unit of Work (`UowTarget`), with the reference's call shape (a coroutine
per item) and with the batched model path (`BatchEmbedder`), and over the in-memory tables (`StoreTarget`)."""

import json
from pathlib import Path

import numpy as np
import pytest

GOLDEN = json.loads((Path(__file__).parent / "golden" / "ingest_golden.json").read_text())


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _payload(row):
    c = row["contents"]
    return c.encode() if (row["bytes"] and c is not None) else c


def _fake_service(case):
    """Repositories with the reference's method names over plain Python rows (base_ingestion.py's view of a Unit of Work)."""
    col = "embedding" if case["embedding_type"] == "single" else "embeddings"
    table = [_Obj(id=r["id"], contents=_payload(r), embedding=r.get("embedding"), embeddings=r.get("embeddings")) for r in case["rows"]]
    fetches = []
    bm25_calls = []

    class Repo:
        def _missing(self):
            return [e for e in table if getattr(e, col) is None]

        def count_without_embeddings(self):
            return len(self._missing())

        count_without_multi_embeddings = count_without_embeddings

        def get_without_embeddings(self, limit=None, offset=None, excluded_ids=None):
            rows = [e for e in self._missing() if not excluded_ids or e.id not in excluded_ids]
            fetches.append([e.id for e in rows[:limit]])
            return rows[:limit]

        get_without_multi_embeddings = get_without_embeddings

        def get_by_id(self, pk):
            return next((e for e in table if e.id == pk), None)

        def set_multi_vector_embeddings_batch(self, entity_ids, embeddings_list, vector_column="embeddings", id_column="id"):
            n = 0
            for pk, emb in zip(entity_ids, embeddings_list, strict=True):
                e = self.get_by_id(pk)
                if e is not None:
                    setattr(e, vector_column, emb)
                    n += 1
            return n

        def batch_update_bm25_tokens(self, tokenizer="bert", batch_size=1000):
            bm25_calls.append({"tokenizer": tokenizer, "batch_size": batch_size,
                               "rows_embedded_at_call": sum(1 for e in table if getattr(e, col) is not None)})
            if case.get("bm25_raises"):
                raise RuntimeError("function tokenize(text, unknown) does not exist")
            return len(table)

    repo = Repo()
    commits = []

    class Uow:
        session = object()
        queries = chunks = image_chunks = repo

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def commit(self):
            commits.append(1)

    return _Obj(_create_uow=lambda: Uow(), bm25_calls=bm25_calls), table, fetches, col, commits


def _vectors(case, col):
    """payload key -> the vector the fixture's embedding function produced for it (read off the final table)."""
    by_id = {f["id"]: f[col] for f in case["final"]}
    pre = {r["id"] for r in case["rows"] if r.get(col) is not None}
    return {r["contents"]: by_id[r["id"]] for r in case["rows"] if r["id"] not in pre and by_id[r["id"]] is not None}


def _key(data):
    return data.decode() if isinstance(data, bytes) else data


def _check_final(case, col, final):
    assert [f["id"] for f in final] == [f["id"] for f in case["final"]]
    for got, exp in zip(final, case["final"]):
        if exp[col] is None:
            assert got[col] is None, got["id"]
        else:
            assert got[col] is not None and np.array_equal(np.asarray(got[col], dtype=np.float64), np.asarray(exp[col])), got["id"]


@pytest.mark.parametrize("ci", range(len(GOLDEN["cases"])))
def test_uow_target_with_the_reference_call_shape(ci):
    from autorag_research_amd.ingest import UowTarget, embed_entities_report

    case = GOLDEN["cases"][ci]
    svc, table, fetches, col, commits = _fake_service(case)
    vec, calls = _vectors(case, col), []

    async def embed(data):
        k = _key(data)
        calls.append(k)
        if k in case["bad_raise"]:
            raise RuntimeError(f"cannot embed {k}")
        if k in case["bad_none"]:
            return None
        return vec[k]

    rep = embed_entities_report(UowTarget(svc), case["entity_type"], case["embedding_type"], embed, batch_size=case["batch_size"],
                                max_concurrency=3, bm25_tokenizer=case.get("bm25_tokenizer"))
    assert rep.total_embedded == case["returned"]
    # bm25_tokens: the repository call the reference makes behind the loop (tokenizer, batch size, and only after every row is stored)
    assert svc.bm25_calls == case.get("bm25_calls", [])
    assert rep.bm25_updated == (0 if case.get("bm25_raises") or not case.get("bm25_calls") else len(case["rows"]))
    assert sorted(calls) == case["embed_calls"] and len(calls) == case["n_embed_calls"]
    assert fetches == case["fetches"]
    _check_final(case, col, [{"id": e.id, col: getattr(e, col)} for e in table])
    assert sorted(map(str, rep.failed_ids)) == sorted(str(r["id"]) for r in case["rows"] if r["contents"] in case["bad_raise"] + case["bad_none"])
    assert rep.skipped_none_content == sum(1 for r in case["rows"] if r["contents"] is None and case["entity_type"] == "image_chunk"
                                           and r.get(col) is None)
    assert len(commits) == sum(1 for f in case["fetches"] if f) - sum(
        1 for f in case["fetches"] if f and all(next(r for r in case["rows"] if r["id"] == pk)["contents"] is None
                                                or next(r for r in case["rows"] if r["id"] == pk)["contents"] in case["bad_raise"] + case["bad_none"]
                                                for pk in f))   # one Unit of Work committed per batch that stored something


class _Model:
    """A model object whose BATCH methods fail as a whole when a bad item is in the batch (what a real forward does)."""

    def __init__(self, case, vec, multi):
        self.case, self.vec, self.multi, self.batches, self.singles = case, vec, multi, 0, 0

    def _many(self, items):
        self.batches += 1
        keys = [_key(i) for i in items]
        if any(k in self.case["bad_raise"] for k in keys):
            raise RuntimeError("batch forward failed")
        return [None if k in self.case["bad_none"] else self.vec[k] for k in keys]

    def _one(self, item):
        self.singles += 1
        k = _key(item)
        if k in self.case["bad_raise"]:
            raise RuntimeError(f"cannot embed {k}")
        return None if k in self.case["bad_none"] else self.vec[k]

    embed_queries = embed_images = embed_documents = _many
    embed_query = embed_image = _one


@pytest.mark.parametrize("target_kind", ["uow", "store"])
@pytest.mark.parametrize("ci", range(len(GOLDEN["cases"])))
def test_batched_model_path_has_the_reference_outcome(ci, target_kind):
    from autorag_research_amd.ingest import BatchEmbedder, StoreTarget, UowTarget, embed_entities_report
    from autorag_research_amd.store import InMemoryStore

    case = GOLDEN["cases"][ci]
    entity, emb_type = case["entity_type"], case["embedding_type"]
    svc, table, fetches, col, _ = _fake_service(case)
    model = _Model(case, _vectors(case, col), emb_type == "multi_vector")
    embedder = BatchEmbedder(model, "image" if entity == "image_chunk" else "query")
    if target_kind == "uow":
        rep = embed_entities_report(UowTarget(svc), entity, emb_type, embedder, batch_size=case["batch_size"],
                                    bm25_tokenizer=case.get("bm25_tokenizer"))
        final = [{"id": e.id, col: getattr(e, col)} for e in table]
        assert fetches == case["fetches"]
        assert svc.bm25_calls == case.get("bm25_calls", [])
    else:
        store = InMemoryStore()
        ids = [r["id"] for r in case["rows"]]
        if entity == "query":
            store.add_queries(ids, contents=[r["contents"] for r in case["rows"]], embedding=[r.get("embedding") for r in case["rows"]],
                              embeddings=[r.get("embeddings") for r in case["rows"]])
        else:
            single = None
            if any(r.get("embedding") is not None for r in case["rows"]):
                d = len(next(r["embedding"] for r in case["rows"] if r.get("embedding") is not None))
                single = np.full((len(ids), d), np.nan, np.float32)
                for i, r in enumerate(case["rows"]):
                    if r.get("embedding") is not None:
                        single[i] = r["embedding"]
            multivec = [r.get("embeddings") for r in case["rows"]] if any(r.get("embeddings") is not None for r in case["rows"]) else None
            if entity == "chunk":
                store.set_chunks(ids, [r["contents"] for r in case["rows"]], embedding=single, multivec=multivec)
            else:
                store.set_image_chunks(ids, embedding=single, multivec=multivec, contents=[_payload(r) for r in case["rows"]])
        rep = embed_entities_report(StoreTarget(store), entity, emb_type, embedder, batch_size=case["batch_size"])
        final = []
        for i, pk in enumerate(ids):
            if entity == "query":
                v = getattr(store.queries[pk], col)
            else:
                t = store.image_chunks if entity == "image_chunk" else store.chunks
                if emb_type == "single":
                    v = None if t.embedding is None or np.isnan(t.embedding[i]).all() else t.embedding[i]
                else:
                    v = None if t.mv_offsets is None or t.mv_offsets[i + 1] == t.mv_offsets[i] else t.mv_tokens[t.mv_offsets[i]:t.mv_offsets[i + 1]]
            final.append({"id": pk, col: None if v is None else np.asarray(v, dtype=np.float32)})
        for f, e in zip(final, case["final"]):   # the store keeps fp32 (VECTOR(d) is float4): compare at that precision
            assert (f[col] is None) == (e[col] is None)
            if e[col] is not None:
                assert np.array_equal(f[col], np.asarray(e[col], dtype=np.float32))
        final = None
    assert rep.total_embedded == case["returned"]
    if final is not None:
        _check_final(case, col, final)
    n_bad = sum(1 for r in case["rows"] if r["contents"] in case["bad_raise"] + case["bad_none"])
    assert len(rep.failed_ids) == n_bad
    # one forward per batch; item-by-item only inside the batches whose forward raised
    n_batches = sum(1 for f in case["fetches"] if any(next(r for r in case["rows"] if r["id"] == pk)["contents"] is not None for pk in f))
    assert model.batches == n_batches
    assert (model.singles > 0) == any(any(next(r for r in case["rows"] if r["id"] == pk)["contents"] in case["bad_raise"] for pk in f)
                                      for f in case["fetches"])


def test_default_tokenizer_is_the_references_and_a_repository_without_the_method_is_a_warning():
    """`bm25_tokenizer` defaults to "bert" (base_ingestion.py:336); a repository whose `batch_update_bm25_tokens` is missing or
    raises costs a warning, not the run (:529-537)."""
    from autorag_research_amd.ingest import UowTarget, embed_entities_report

    case = next(c for c in GOLDEN["cases"] if c.get("bm25_tokenizer") == "bert" and c["bm25_calls"])
    svc, table, _, col, _ = _fake_service(case)
    vec = _vectors(case, col)

    async def embed(data):
        return None if data in case["bad_none"] else vec[data]

    rep = embed_entities_report(UowTarget(svc), case["entity_type"], case["embedding_type"], embed, batch_size=case["batch_size"])
    assert rep.total_embedded == case["returned"] and [c["tokenizer"] for c in svc.bm25_calls] == ["bert"]


def test_store_target_fetches_in_linear_time():
    """`fetch_without` rescanned the table from row 0 for every batch (O(n^2 / batch): 20 k rows took 3.6 s).  The
    cursor serves 60 k rows in well under a second, skips excluded ids, and a second run over the same target sees what the
    first one left."""
    import time

    from autorag_research_amd.ingest import StoreTarget
    from autorag_research_amd.store import InMemoryStore

    n, d = 60_000, 8
    store = InMemoryStore()
    emb = np.full((n, d), np.nan, np.float32)
    emb[::7] = 1.0                                    # every 7th row already embedded
    store.set_chunks(list(range(n)), [f"t{i}" for i in range(n)], embedding=emb)
    tgt = StoreTarget(store)
    assert tgt.count_without("chunk", "single") == n - len(range(0, n, 7))
    t0 = time.perf_counter()
    seen, failed = [], {5, 6, 8}
    while True:
        items = tgt.fetch_without("chunk", "single", 128, failed)
        if not items:
            break
        seen.extend(pk for pk, _ in items)
        keep = [pk for pk, _ in items if pk % 1000 != 1]          # some rows "fail": excluded from then on
        failed.update(pk for pk, _ in items if pk % 1000 == 1)
        tgt.set_embeddings("chunk", "single", keep, [np.zeros(d, np.float32)] * len(keep))
    dt = time.perf_counter() - t0
    expect = [i for i in range(n) if i % 7 != 0 and i not in (5, 6, 8)]
    assert seen == expect and dt < 2.0, dt
    again = tgt.fetch_without("chunk", "single", 10_000, set())        # a new run: the rows left NULL come back, in table order
    assert [pk for pk, _ in again] == sorted({5, 6, 8} | {i for i in expect if i % 1000 == 1})


def test_rejects_what_the_reference_rejects():
    from autorag_research_amd.ingest import UowTarget, embed_entities
    from autorag_research_amd.store import InMemoryStore

    with pytest.raises(KeyError):
        embed_entities(InMemoryStore(), "page", "single", lambda x: x)
    with pytest.raises(ValueError, match="embedding_type"):
        embed_entities(InMemoryStore(), "chunk", "dense", lambda x: x)

    class NoImages:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    with pytest.raises(RuntimeError, match="image_chunks"):   # RepositoryNotSupportedError in the reference (:381-383)
        embed_entities(UowTarget(_Obj(_create_uow=lambda: NoImages())), "image_chunk", "single", lambda x: x)
