"""Embedding of un-embedded rows (the encode half of the path), batched.

Mirrors, behaviour for behaviour (pinned by tests/golden/ingest_golden.json, which is the reference's own `_embed_entities`
run over a fake Unit of Work):
  BaseIngestionService._embed_entities / _fetch_unembedded_batch / _embed_batch / _set_embeddings
      autorag_research/orm/service/base_ingestion.py:199-247, 326-495
  embed_all_queries / embed_all_chunks / ..._multi_vector          base_ingestion.py:542-624
  TextEmbeddingDataIngestor.embed_all, MultiModalEmbeddingDataIngestor.embed_all / embed_all_late_interaction
      autorag_research/data/base.py:57-72, 95-125
What the reference does and this does too: chunk and query rows get their `bm25_tokens` from the repository's
`batch_update_bm25_tokens` right behind the loop (base_ingestion.py:429-430, 497-540; default tokenizer "bert", None skips it;
`UowTarget` only -- the in-memory tables have no such column); rows lacking the column are fetched `batch_size` at a time, EXCLUDING the ids that
already failed in this run; image chunks whose content is NULL are skipped and counted (never handed to the model); an item
whose embedding raised or came back as None is remembered for the run and the rest of its batch is still stored; the loop ends
when a fetch comes back empty or a batch made no progress; the return value is the number of rows stored.  Queries AND chunks
go through the model's QUERY side (`aembed_query`, data/base.py:63-71), image chunks through `aembed_image`.

What is different: the model runs ONE forward per batch (`embed_queries` / `embed_documents` / `embed_images`) instead of one
per item under a semaphore; when a batch forward raises, its items are retried one by one so that the failing item alone is
recorded -- the outcome (which rows are stored, which ids failed) is the reference's.  A coroutine function `data -> embedding`
(the reference's call shape) is accepted as well and is then gathered under `max_concurrency` exactly like
`util.run_with_concurrency_limit` (util.py:184-244).

Two targets: `StoreTarget` (the in-memory tables of store.py) and `UowTarget` (the REFERENCE's repositories behind any service
with `_create_uow()`: `count_without_*`, `get_without_*(limit, excluded_ids)`, `get_by_id` + attribute assignment + commit for
`embedding VECTOR(d)`, `set_multi_vector_embeddings_batch` for `embeddings VECTOR(d)[]` -- the calls base_ingestion.py makes).
"""

from __future__ import annotations

import asyncio
import inspect
import logging
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from .embeddings import MultiVectorBaseEmbedding
from .store import ChunkTable, InMemoryStore

logger = logging.getLogger("AutoRAG-Research")

# entity_type -> (repository attribute, data attribute, display name, skip rows whose data is None); base_ingestion.py:40-46
ENTITY_CONFIG: dict[str, tuple[str, str, str, bool]] = {
    "query": ("queries", "contents", "queries", False),
    "chunk": ("chunks", "contents", "chunks", False),
    "image_chunk": ("image_chunks", "contents", "image chunks", True),
}


@dataclass
class IngestReport:
    """What one `_embed_entities` run did (the reference logs these three numbers and returns the first)."""

    total_embedded: int = 0
    failed_ids: list = field(default_factory=list)      # embedding raised / returned None; not retried in this run
    skipped_none_content: int = 0                         # image chunks with NULL content
    skipped_ids: list = field(default_factory=list)
    bm25_updated: int = 0                                 # rows whose bm25_tokens the repository filled behind the loop


# ---- targets ---------------------------------------------------------------------------------------------------------
class StoreTarget:
    """`InMemoryStore` tables: NULL = a NaN row of `embedding`, an empty span of the ragged store, None on a query row."""

    def __init__(self, store: InMemoryStore):
        self.store = store
        self._pending_mv: dict[str, dict[int, np.ndarray]] = {"chunk": {}, "image_chunk": {}}
        self._cursor: dict[tuple[str, str], list] = {}   # (entity, embedding type) -> [positions lacking the column, next to serve]

    def _table(self, entity: str) -> ChunkTable:
        return self.store.image_chunks if entity == "image_chunk" else self.store.chunks

    def _is_null(self, entity: str, emb_type: str, i: int) -> bool:
        """row i of the table (or query i of `query_order`) still lacks the column"""
        if entity == "query":
            q = self.store.queries[self.store.query_order[i]]
            return getattr(q, "embedding" if emb_type == "single" else "embeddings") is None
        t = self._table(entity)
        if emb_type == "single":
            return t.embedding is None or bool(np.isnan(t.embedding[i]).all())
        return i not in self._pending_mv[entity] and (t.mv_offsets is None or t.mv_offsets[i + 1] == t.mv_offsets[i])

    def _missing_positions(self, entity: str, emb_type: str) -> list[int]:
        """positions (table order) of every row lacking the column: ONE vectorised pass over the table"""
        if entity == "query":
            attr = "embedding" if emb_type == "single" else "embeddings"
            qs = self.store.queries
            return [i for i, qid in enumerate(self.store.query_order) if getattr(qs[qid], attr) is None]
        t = self._table(entity)
        n = len(t.ids)
        if emb_type == "single":
            if t.embedding is None:
                return list(range(n))
            return np.flatnonzero(np.isnan(t.embedding).all(axis=1)).tolist()
        null = np.ones(n, dtype=bool) if t.mv_offsets is None else (np.diff(t.mv_offsets) == 0)
        pend = self._pending_mv[entity]
        return [int(i) for i in np.flatnonzero(null) if int(i) not in pend]

    def _row(self, entity: str, i: int) -> tuple[Any, Any]:
        if entity == "query":
            qid = self.store.query_order[i]
            return qid, self.store.queries[qid].contents
        t = self._table(entity)
        return t.ids[i], t.contents[i]

    def count_without(self, entity: str, emb_type: str) -> int:
        return len(self._missing_positions(entity, emb_type))

    def fetch_without(self, entity: str, emb_type: str, limit: int, excluded: set) -> list[tuple[Any, Any]]:
        """The next `limit` rows lacking the column, table order, `excluded` left out.  The list of candidates is computed once
        and served from a cursor (a row handed out is stored or excluded by the caller, so nothing behind the cursor can come
        back during a run); every candidate is re-checked when served, and an exhausted list is rebuilt ONCE per call -- a
        second run over the same target sees the rows the first one failed on.  (Round 5 rescanned from row 0 per batch with a
        Python-level test per row: O(n^2 / batch) -- hours at 1 M rows.)"""
        key = (entity, emb_type)
        cur = self._cursor.get(key)
        if cur is None:
            cur = self._cursor[key] = [self._missing_positions(entity, emb_type), 0]
        out: list[tuple[Any, Any]] = []
        rebuilt = False
        while len(out) < limit:
            if cur[1] >= len(cur[0]):
                if rebuilt:
                    break
                taken = {pk for pk, _ in out}   # (handed out by this very call and therefore still NULL)
                cur[0] = [i for i in self._missing_positions(entity, emb_type)
                          if self._row(entity, i)[0] not in excluded and self._row(entity, i)[0] not in taken]
                cur[1] = 0
                rebuilt = True
                if not cur[0]:
                    break
                continue
            i = cur[0][cur[1]]
            cur[1] += 1
            pk, data = self._row(entity, i)
            if pk in excluded or not self._is_null(entity, emb_type, i):
                continue
            out.append((pk, data))
        return out

    def populate_bm25_tokens(self, entity: str, tokenizer: str, batch_size: int) -> int:
        """The in-memory tables have no `bm25_tokens` column (BM25 is the reference's own SQL: out of scope); nothing to do."""
        return 0

    def set_embeddings(self, entity: str, emb_type: str, ids: list, embeddings: list) -> int:
        if len(ids) != len(embeddings):
            raise ValueError("Length mismatch: entity_ids and embeddings")   # LengthMismatchError in the reference
        if entity == "query":
            n = 0
            for qid, e in zip(ids, embeddings):
                q = self.store.queries.get(qid)
                if q is not None:
                    setattr(q, "embedding" if emb_type == "single" else "embeddings", np.ascontiguousarray(e, dtype=np.float32))
                    n += 1
            return n
        t = self._table(entity)
        pos = getattr(t, "_pos", None)
        if pos is None or len(pos) != len(t.ids):
            pos = {pk: i for i, pk in enumerate(t.ids)}
            t._pos = pos  # type: ignore[attr-defined]
        n = 0
        for pk, e in zip(ids, embeddings):
            i = pos.get(pk)
            if i is None:
                continue
            v = np.ascontiguousarray(e, dtype=np.float32)
            if emb_type == "single":
                if t.embedding is None:
                    t.embedding = np.full((len(t.ids), v.shape[0]), np.nan, dtype=np.float32)
                t.embedding[i] = v
            else:
                self._pending_mv[entity][i] = v.reshape(-1, v.shape[-1])
            n += 1
        return n

    def finish(self) -> None:
        """Merge the multi-vector rows stored during the run into the ragged arrays (once, not per batch)."""
        for entity, pend in self._pending_mv.items():
            if not pend:
                continue
            t = self._table(entity)
            n = len(t.ids)
            docs: list = [None] * n
            if t.mv_offsets is not None:
                for i in range(n):
                    if t.mv_offsets[i + 1] > t.mv_offsets[i]:
                        docs[i] = t.mv_tokens[t.mv_offsets[i]: t.mv_offsets[i + 1]]
            for i, m in pend.items():
                docs[i] = m
            d = next(x.shape[1] for x in docs if x is not None)
            lens = [0 if x is None else x.shape[0] for x in docs]
            t.mv_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            t.mv_tokens = np.concatenate([x for x in docs if x is not None], axis=0) if any(lens) else np.zeros((0, d), np.float32)
            pend.clear()


class UowTarget:
    """The reference's own repositories behind a service object with `_create_uow()` (its ingestion services, or the
    RetrievalPipelineService the Executor builds): the statements base_ingestion.py issues, one Unit of Work per call."""

    def __init__(self, ref_service: Any):
        self._svc = ref_service

    def _repo(self, uow, entity: str):
        repo_attr = ENTITY_CONFIG[entity][0]
        repo = getattr(uow, repo_attr, None)
        if repo is None:
            raise RuntimeError(f"Repository '{repo_attr}' is not supported by {type(uow).__name__}")  # RepositoryNotSupportedError
        return repo

    def count_without(self, entity: str, emb_type: str) -> int:
        with self._svc._create_uow() as uow:
            repo = self._repo(uow, entity)
            return (repo.count_without_embeddings if emb_type == "single" else repo.count_without_multi_embeddings)()

    def fetch_without(self, entity: str, emb_type: str, limit: int, excluded: set) -> list[tuple[Any, Any]]:
        data_attr = ENTITY_CONFIG[entity][1]
        with self._svc._create_uow() as uow:
            repo = self._repo(uow, entity)
            fetch = repo.get_without_embeddings if emb_type == "single" else repo.get_without_multi_embeddings
            return [(e.id, getattr(e, data_attr)) for e in fetch(limit=limit, excluded_ids=excluded)]

    def set_embeddings(self, entity: str, emb_type: str, ids: list, embeddings: list) -> int:
        if len(ids) != len(embeddings):
            raise ValueError("Length mismatch: entity_ids and embeddings")
        n = 0
        with self._svc._create_uow() as uow:
            if getattr(uow, "session", True) is None:
                raise RuntimeError("Session is not set")   # SessionNotSetError
            repo = self._repo(uow, entity)
            if emb_type == "multi_vector":
                n = repo.set_multi_vector_embeddings_batch(ids, embeddings, vector_column="embeddings", id_column="id")
            else:
                for pk, e in zip(ids, embeddings, strict=True):
                    row = repo.get_by_id(pk)
                    if row:
                        row.embedding = e
                        n += 1
            uow.commit()
        return n

    def populate_bm25_tokens(self, entity: str, tokenizer: str, batch_size: int) -> int:
        """`_populate_bm25_tokens` (base_ingestion.py:497-540): the repository's own batch update; a failure (VectorChord-BM25
        not installed) is a warning and 0, like the reference."""
        repo_attr = "chunks" if entity == "chunk" else "queries"
        with self._svc._create_uow() as uow:
            repo = getattr(uow, repo_attr, None)
            if repo is None:
                raise RuntimeError(f"Repository '{repo_attr}' is not supported by {type(uow).__name__}")
            try:
                updated = repo.batch_update_bm25_tokens(tokenizer=tokenizer, batch_size=batch_size)
            except Exception as e:  # noqa: BLE001 - the reference swallows exactly this
                logger.warning(f"Failed to generate BM25 tokens for {entity}s (extension may not be installed): {e}")
                return 0
            logger.info(f"Generated BM25 tokens for {updated} {entity}s using tokenizer '{tokenizer}'")
            return updated

    def finish(self) -> None:
        pass


def as_target(where: Any):
    if hasattr(where, "count_without") and hasattr(where, "fetch_without"):
        return where
    if isinstance(where, InMemoryStore):
        return StoreTarget(where)
    if hasattr(where, "_create_uow"):
        return UowTarget(where)
    raise TypeError("expected an InMemoryStore, a service with _create_uow(), or an ingest target")


# ---- embedders ---------------------------------------------------------------------------------------------------------
def _to_lists(x: Any) -> Any:
    return x.tolist() if hasattr(x, "tolist") else x


class BatchEmbedder:
    """`list[data] -> list[embedding | None]` over a model object: ONE forward per batch; a batch that raises is retried item
    by item so that only the failing items come back as None (what the reference gets from one call per item)."""

    def __init__(self, model: Any, kind: str = "query"):
        if kind not in ("query", "document", "image"):
            raise ValueError("kind must be 'query' (the reference's text side), 'document' or 'image'")
        self.model, self.kind = model, kind

    def _batch(self, items: list) -> list:
        m = self.model
        # image / document forwards are ONE padded batch in the wrappers (like the reference's): cut an ingest batch by the
        # model's own `embed_batch_size` so that 128 page images do not go through a VLM in one forward
        step = int(getattr(m, "embed_batch_size", 0) or 0)
        if self.kind in ("image", "document") and step > 0 and len(items) > step:
            fn = m.embed_images if self.kind == "image" else m.embed_documents
            out: list = []
            for i in range(0, len(items), step):
                out.extend(fn(items[i: i + step]))
            return out
        if self.kind == "image":
            return list(m.embed_images(items))
        if self.kind == "document":
            return list(m.embed_documents(items))
        if hasattr(m, "embed_queries"):
            return list(m.embed_queries(items))
        return [m.embed_query(t) for t in items]

    def _one(self, item: Any):
        m = self.model
        if self.kind == "image":
            return m.embed_image(item)
        if self.kind == "document":
            return m.embed_documents([item])[0]
        return m.embed_query(item)

    def __call__(self, items: list, error_msg: str = "Failed to embed") -> list:
        try:
            out = self._batch(items)
            if len(out) == len(items):
                return [None if e is None else _to_lists(e) for e in out]
        except Exception as e:  # noqa: BLE001 - isolate the failing item(s) below
            logger.warning(f"{error_msg}: a batch of {len(items)} raised {type(e).__name__}: {e}; retrying its items one by one")
        res = []
        for it in items:
            try:
                e = self._one(it)
                res.append(None if e is None else _to_lists(e))
            except Exception:  # noqa: BLE001
                logger.exception(error_msg)
                res.append(None)
        return res


async def _gather_limited(items: list, func, max_concurrency: int, error_msg: str) -> list:
    """util.run_with_concurrency_limit (util.py:184-244): exceptions are logged and become None, order is kept."""
    sem = asyncio.Semaphore(max_concurrency)

    async def one(item):
        async with sem:
            try:
                return await func(item)
            except Exception:  # noqa: BLE001
                logger.exception(error_msg)
                return None

    return list(await asyncio.gather(*[one(i) for i in items]))


def _run_embed(embed: Any, items: list, max_concurrency: int, error_msg: str) -> list:
    if isinstance(embed, BatchEmbedder):
        return embed(items, error_msg)
    if inspect.iscoroutinefunction(embed) or inspect.iscoroutinefunction(getattr(embed, "__call__", None)):
        return asyncio.run(_gather_limited(items, embed, max_concurrency, error_msg))
    raise TypeError("embed must be a BatchEmbedder or a coroutine function `data -> embedding`")


# ---- the loop (base_ingestion.py:326-437, 461-495) ---------------------------------------------------------------------
def embed_entities_report(where: Any, entity_type: str, embedding_type: str, embed: Any, batch_size: int = 128,
                          max_concurrency: int = 16, bm25_tokenizer: str | None = "bert") -> IngestReport:
    if entity_type not in ENTITY_CONFIG:
        raise KeyError(entity_type)
    if embedding_type not in ("single", "multi_vector"):
        raise ValueError("embedding_type must be 'single' or 'multi_vector'")
    target = as_target(where)
    _, _, display, filter_none = ENTITY_CONFIG[entity_type]
    suffix = " with multi-vector" if embedding_type == "multi_vector" else ""
    error_msg = f"Failed to embed {'image' if entity_type == 'image_chunk' else 'text'}{suffix}"
    rep = IngestReport()
    if target.count_without(entity_type, embedding_type) == 0:
        logger.info(f"No {display} to embed{suffix}")
        return rep
    failed: set = set()
    try:
        while True:
            items = target.fetch_without(entity_type, embedding_type, batch_size, failed)
            if not items:
                break
            if filter_none:
                none_rows = [(pk, d) for pk, d in items if d is None]
                items = [(pk, d) for pk, d in items if d is not None]
                if none_rows:
                    failed.update(pk for pk, _ in none_rows)
                    rep.skipped_none_content += len(none_rows)
                    rep.skipped_ids.extend(pk for pk, _ in none_rows)
                if not items:
                    continue
            items = items[:batch_size]
            embs = _run_embed(embed, [d for _, d in items], max_concurrency, error_msg)
            good = [(pk, e) for (pk, _), e in zip(items, embs, strict=True) if e is not None]
            bad = [pk for (pk, _), e in zip(items, embs, strict=True) if e is None]
            if bad:
                failed.update(bad)
                rep.failed_ids.extend(bad)
                logger.warning(f"Skipping {len(bad)} {display} that failed to embed in this run "
                               f"(IDs: {bad[:5]}{'...' if len(bad) > 5 else ''})")
            if good:
                rep.total_embedded += target.set_embeddings(entity_type, embedding_type, [pk for pk, _ in good], [e for _, e in good])
            elif not bad:
                break
    finally:
        target.finish()
    # base_ingestion.py:429-430: chunk and query rows get their `bm25_tokens` right behind the embeddings (the reference's BM25
    # and hybrid pipelines read nothing else); a target without the column does nothing
    if entity_type in ("chunk", "query") and bm25_tokenizer is not None and hasattr(target, "populate_bm25_tokens"):
        rep.bm25_updated = target.populate_bm25_tokens(entity_type, bm25_tokenizer, batch_size)
    logger.info(f"Total {display} embedded{suffix}: {rep.total_embedded} (skipped_failed={len(rep.failed_ids)}, "
                f"skipped_empty_content={rep.skipped_none_content})")
    return rep


def embed_entities(where: Any, entity_type: str, embedding_type: str, embed: Any, batch_size: int = 128,
                   max_concurrency: int = 16, bm25_tokenizer: str | None = "bert") -> int:
    """`BaseIngestionService._embed_entities`: the number of rows embedded and stored."""
    return embed_entities_report(where, entity_type, embedding_type, embed, batch_size, max_concurrency, bm25_tokenizer).total_embedded


def _embedder(model_or_func: Any, kind: str):
    if isinstance(model_or_func, BatchEmbedder) or inspect.iscoroutinefunction(model_or_func):
        return model_or_func
    return BatchEmbedder(model_or_func, kind)


def _emb_type(model_or_func: Any, embedding_type: str | None) -> str:
    if embedding_type is not None:
        return embedding_type
    model = model_or_func.model if isinstance(model_or_func, BatchEmbedder) else model_or_func
    return "multi_vector" if isinstance(model, MultiVectorBaseEmbedding) else "single"


def embed_all_queries(where: Any, model: Any, batch_size: int = 128, max_concurrency: int = 16,
                      embedding_type: str | None = None, bm25_tokenizer: str | None = "bert") -> int:
    """embed_all_queries / embed_all_queries_multi_vector (base_ingestion.py:542-582): by the model's kind unless told."""
    return embed_entities(where, "query", _emb_type(model, embedding_type), _embedder(model, "query"), batch_size, max_concurrency,
                          bm25_tokenizer)


def embed_all_chunks(where: Any, model: Any, batch_size: int = 128, max_concurrency: int = 16, unit: str = "chunk",
                     side: str = "query", embedding_type: str | None = None, bm25_tokenizer: str | None = "bert") -> int:
    """embed_all_chunks[_multi_vector] (base_ingestion.py:584-624) and, with `unit="image_chunk"`, embed_all_image_chunks
    [_multi_vector]: text chunks through the model's QUERY side like the reference (data/base.py:68-72; `side="document"` is an
    explicit deviation for passage-side vectors), image chunks' bytes through the image embedder (data/base.py:110-124)."""
    if side not in ("query", "document"):
        raise ValueError("side must be 'query' (the reference's behaviour) or 'document'")
    if unit == "image_chunk":
        return embed_entities(where, "image_chunk", _emb_type(model, embedding_type), _embedder(model, "image"), batch_size,
                              max_concurrency)
    return embed_entities(where, "chunk", _emb_type(model, embedding_type), _embedder(model, side), batch_size, max_concurrency,
                          bm25_tokenizer)


def embed_all(where: Any, model: Any, max_concurrency: int = 16, batch_size: int = 128, unit: str = "chunk") -> None:
    """TextEmbeddingDataIngestor.embed_all (unit "chunk") / MultiModalEmbeddingDataIngestor.embed_all (unit "image_chunk") and,
    for a multi-vector model, embed_all_late_interaction: queries first, then the corpus table."""
    if model is None:
        raise RuntimeError("Embedding model is not set")   # EmbeddingError
    embed_all_queries(where, model, batch_size, max_concurrency)
    embed_all_chunks(where, model, batch_size, max_concurrency, unit=unit)


def index_texts_on_device(model: Any, texts: list[str], index: Any, batch_size: int = 1024) -> int:
    """Encode `texts` and append the vectors to `index` without leaving the GPU.

    `model` is a single-vector encoder with `encode_to_device(texts) -> fp32 [n, d] tensor on the index's device`
    (TorchEncoderEmbeddings); each batch's tensor is handed to `Mi355Index.add_device` by pointer
    (mi355dr_add_rows_device: device-to-device copy + norms + shadows on the library's stream).  The reference's path for
    the same work is model -> `.tolist()` -> per-row SQL UPDATE with a text literal (base_ingestion.py:199-247, 326-495).
    Returns the number of rows added.
    """
    import torch

    done = 0
    for i in range(0, len(texts), batch_size):
        v = model.encode_to_device(texts[i: i + batch_size])
        if v.dtype != torch.float32 or not v.is_contiguous():
            v = v.float().contiguous()
        if v.shape[0] == 0:
            continue
        if v.shape[1] != index.dim:
            raise ValueError(f"encoder output dim {v.shape[1]} != index dim {index.dim}")
        torch.cuda.current_stream(v.device).synchronize()  # the library copies on its own stream
        index.add_device(v.data_ptr(), v.shape[0])
        done += v.shape[0]
    return done
