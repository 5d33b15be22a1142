"""InMemoryStore -- the handful of tables the Vector Search path touches, without PostgreSQL.

The reference keeps these in PostgreSQL (postgresql/db/init/001-schema.sql:98-138 chunk / image_chunk /
query with `embedding VECTOR(d)` and `embeddings VECTOR(d)[]`; :186-201 the retrieved-result tables
with PK (query_id, pipeline_id, chunk_id) and `rel_score FLOAT`).  This store holds the same columns as
numpy arrays so the service/pipeline mirror (service.py, pipelines.py) can run standalone -- on the GPU
box, in the bench harness and in tests -- and so a DB exporter has a concrete target format
(INTEGRATION.md).  Primary keys may be int (BIGINT) or str (VARCHAR) as in schema_factory.py:63-76.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import numpy as np


@dataclass
class QueryRow:
    id: int | str
    contents: str | None = None
    embedding: np.ndarray | None = None     # [d] fp32 (VECTOR(d)) or None (NULL)
    embeddings: np.ndarray | None = None    # [n_q, d] fp32 (VECTOR(d)[]) or None


@dataclass
class RetrievalRelation:
    """Ground truth row (reference 001-schema.sql retrieval_relation: group_index = AND, group_order = OR)."""

    query_id: int | str
    group_index: int
    group_order: int
    chunk_id: int | str | None = None
    image_chunk_id: int | str | None = None
    score: int | None = None


@dataclass
class ChunkTable:
    """chunk or image_chunk: ids, optional contents, single- and/or multi-vector embeddings."""

    ids: list[int | str] = field(default_factory=list)
    contents: list[str | None] = field(default_factory=list)
    embedding: np.ndarray | None = None          # [N, d] fp32; rows of NaN = NULL embedding (skipped by SQL)
    mv_tokens: np.ndarray | None = None          # [sum_T, d] fp32 ragged multi-vector store
    mv_offsets: np.ndarray | None = None         # [N+1] int64; empty span = NULL embeddings

    def __len__(self) -> int:
        return len(self.ids)


class InMemoryStore:
    def __init__(self) -> None:
        self.queries: dict[int | str, QueryRow] = {}
        self.query_order: list[int | str] = []
        self.chunks = ChunkTable()
        self.image_chunks = ChunkTable()
        self.relations: dict[int | str, list[RetrievalRelation]] = {}
        self.pipelines: dict[int, dict[str, Any]] = {}
        self._pipeline_by_name: dict[str, int] = {}
        # result tables: {(pipeline_id, query_id): [(chunk_id, rel_score), ...]} in insertion order
        self.chunk_results: dict[tuple[int, int | str], list[tuple[int | str, float]]] = {}
        self.image_chunk_results: dict[tuple[int, int | str], list[tuple[int | str, float]]] = {}

    # ---- filling ----
    def add_queries(self, ids, contents=None, embedding=None, embeddings=None) -> None:
        for i, qid in enumerate(ids):
            row = QueryRow(
                id=qid,
                contents=None if contents is None else contents[i],
                embedding=None if embedding is None or embedding[i] is None
                else np.ascontiguousarray(embedding[i], dtype=np.float32),
                embeddings=None if embeddings is None or embeddings[i] is None
                else np.ascontiguousarray(embeddings[i], dtype=np.float32),
            )
            if qid not in self.queries:
                self.query_order.append(qid)
            self.queries[qid] = row

    def _fill(self, table: ChunkTable, ids, contents, embedding, multivec) -> None:
        table.ids = list(ids)
        table.contents = list(contents) if contents is not None else [None] * len(table.ids)
        if embedding is not None:
            table.embedding = np.ascontiguousarray(embedding, dtype=np.float32)
            if table.embedding.shape[0] != len(table.ids):
                raise ValueError("embedding rows != ids")
        if multivec is not None:
            toks = [np.ascontiguousarray(m, dtype=np.float32) if m is not None and len(m) else None for m in multivec]
            if len(toks) != len(table.ids):
                raise ValueError("multivec docs != ids")
            d = next((t.shape[1] for t in toks if t is not None), 0)
            lens = [0 if t is None else t.shape[0] for t in toks]
            table.mv_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            table.mv_tokens = (np.concatenate([t for t in toks if t is not None], axis=0)
                               if any(t is not None for t in toks) else np.zeros((0, d), np.float32))

    def set_chunks(self, ids, contents=None, embedding=None, multivec=None) -> None:
        self._fill(self.chunks, ids, contents, embedding, multivec)

    def set_image_chunks(self, ids, embedding=None, multivec=None, contents=None) -> None:
        """`contents` = the image bytes of each row (image_chunk.contents BYTEA in the reference; None = NULL)."""
        self._fill(self.image_chunks, ids, contents, embedding, multivec)

    def add_relations(self, rels: list[RetrievalRelation]) -> None:
        for r in rels:
            self.relations.setdefault(r.query_id, []).append(r)

    # ---- what the service reads (names follow the reference repositories) ----
    def get_query(self, qid):
        return self.queries.get(qid)

    def get_all_queries(self, limit: int, offset: int) -> list[QueryRow]:
        return [self.queries[q] for q in self.query_order[offset: offset + limit]]

    def find_query_by_text(self, text: str):
        for qid in self.query_order:
            if self.queries[qid].contents == text:
                return self.queries[qid]
        return None

    # ---- pipelines / results ----
    def get_or_create_pipeline(self, name: str, config: dict[str, Any]) -> tuple[int, bool]:
        if name in self._pipeline_by_name:
            return self._pipeline_by_name[name], False
        pid = len(self.pipelines) + 1
        self.pipelines[pid] = {"name": name, "config": dict(config)}
        self._pipeline_by_name[name] = pid
        return pid, True

    def results_table(self, unit: str):
        return self.image_chunk_results if unit == "image_chunk" else self.chunk_results

    def completed_query_ids(self, unit: str, pipeline_id: int, query_ids) -> set:
        tab = self.results_table(unit)
        return {q for q in query_ids if tab.get((pipeline_id, q))}

    def insert_page(self, unit: str, pipeline_id: int, query_ids: list, results: list) -> int:
        """A page of ranked lists (None = failed query) at once; same table contents as bulk_insert of the reference's
        row dicts {query_id, pipeline_id, chunk_id|image_chunk_id, rel_score}.  Returns the number of rows stored."""
        tab = self.results_table(unit)
        n = 0
        for qid, res in zip(query_ids, results, strict=True):
            if res:
                tab.setdefault((pipeline_id, qid), []).extend([(r["doc_id"], float(r["score"])) for r in res])
                n += len(res)
        return n

    def bulk_insert(self, unit: str, rows: list[dict[str, Any]]) -> None:
        tab = self.results_table(unit)
        key = "image_chunk_id" if unit == "image_chunk" else "chunk_id"
        for r in rows:
            tab.setdefault((r["pipeline_id"], r["query_id"]), []).append((r[key], float(r["rel_score"])))

    def pipeline_config(self, pipeline_id) -> dict[str, Any]:
        return dict(self.pipelines.get(pipeline_id, {}).get("config", {}))

    def delete_pipeline_results(self, pipeline_id) -> int:
        """Rows of both result tables for one pipeline (reference delete_pipeline_results, retrieval_pipeline.py:359-372)."""
        n = 0
        for tab in (self.chunk_results, self.image_chunk_results):
            for key in [k for k in tab if k[0] == pipeline_id]:
                n += len(tab.pop(key))
        return n

    def delete_pipeline(self, pipeline_id) -> None:
        """What the reference Executor's health-check cleanup does after the results are gone (executor.py:372-383)."""
        name = self.pipelines.pop(pipeline_id, {}).get("name")
        self._pipeline_by_name.pop(name, None)


def _as_vec(x) -> np.ndarray | None:
    return None if x is None else np.ascontiguousarray(x, dtype=np.float32)


class UowStore:
    """The same store interface over the REFERENCE's own service / Unit-of-Work objects -- what makes the plugin run inside
    the reference's Executor, which hands every pipeline a SQLAlchemy sessionmaker (executor.py:326-333, 408-416).

    `ref_service` is an `autorag_research.orm.service.retrieval_pipeline.RetrievalPipelineService` (or anything with the
    same `_create_uow()` / `get_or_create_pipeline()`): queries are paged through `uow.queries`, the corpus is exported once
    per table through `uow.chunks` / `uow.image_chunks` (`embedding VECTOR(d)`, `embeddings VECTOR(d)[]`, NULLs kept as
    NULLs), result rows go back through the result repositories' `bulk_insert` in the reference's own row-dict format
    (retrieval_pipeline.py:151-182, 283-288), so evaluation and reporting read them like any other pipeline's.
    Mi355RetrievalService builds one of these by itself when `session_factory()` does not return a store."""

    EXPORT_PAGE = 50_000

    def __init__(self, ref_service: Any, require_total_order: bool = False):
        self._svc = ref_service
        self._tables: dict[str, ChunkTable] = {}
        # one process per GPU: every rank must export the SAME order (global row ids are positions in it) -> primary-key order
        # is enforced and keys that cannot be ordered are an error.  One process: the order the repository returned is kept
        # when it is already the key order (the keyed path: `ORDER BY id`) -- no O(n log n) Python sort over millions of rows
        # -- and unorderable keys only cost a warning.
        self.require_total_order = require_total_order

    # ---- pipelines ----
    def get_or_create_pipeline(self, name: str, config: dict[str, Any]):
        return self._svc.get_or_create_pipeline(name, config)

    def pipeline_config(self, pipeline_id) -> dict[str, Any]:
        with self._svc._create_uow() as uow:
            p = uow.pipelines.get_by_id(pipeline_id)
            return dict(getattr(p, "config", None) or {}) if p is not None else {}

    def delete_pipeline_results(self, pipeline_id) -> int:
        return self._svc.delete_pipeline_results(pipeline_id)

    # ---- queries ----
    @staticmethod
    def _query_row(q) -> QueryRow:
        return QueryRow(id=q.id, contents=getattr(q, "contents", None), embedding=_as_vec(getattr(q, "embedding", None)),
                        embeddings=_as_vec(getattr(q, "embeddings", None)))

    def get_query(self, qid):
        with self._svc._create_uow() as uow:
            q = uow.queries.get_by_id(qid)
            return None if q is None else self._query_row(q)

    def get_all_queries(self, limit: int, offset: int) -> list[QueryRow]:
        with self._svc._create_uow() as uow:
            return [self._query_row(q) for q in uow.queries.get_all(limit=limit, offset=offset)]

    def find_query_by_text(self, text: str):
        with self._svc._create_uow() as uow:
            q = uow.queries.find_by_contents(text)
            return None if q is None else self._query_row(q)

    # ---- corpus export (once per table) ----
    def _export(self, repo_name: str) -> ChunkTable:
        if repo_name in self._tables:
            return self._tables[repo_name]
        ids, contents, single, multi = [], [], [], []

        def take(c) -> None:
            ids.append(c.id)
            contents.append(getattr(c, "contents", None))
            single.append(_as_vec(getattr(c, "embedding", None)))
            multi.append(_as_vec(getattr(c, "embeddings", None)))

        with self._svc._create_uow() as uow:
            keyed = hasattr(getattr(uow, repo_name), "get_all_ids") and hasattr(getattr(uow, repo_name), "get_by_ids")
        if keyed:
            # page by ORDERED primary key (ChunkRepository.get_all_ids is `ORDER BY id`, orm/repository/chunk.py:123-136), then
            # fetch each page by key: `get_all(limit, offset)` is a SELECT without ORDER BY, one transaction per page -- PostgreSQL
            # promises no stable order across such statements, and a row seen twice or never would shift every row -> id mapping
            with self._svc._create_uow() as uow:
                all_ids = list(getattr(uow, repo_name).get_all_ids(limit=None, offset=0))
            for p0 in range(0, len(all_ids), self.EXPORT_PAGE):
                page_ids = all_ids[p0:p0 + self.EXPORT_PAGE]
                with self._svc._create_uow() as uow:
                    by_id = {c.id: c for c in getattr(uow, repo_name).get_by_ids(page_ids)}
                    for i in page_ids:
                        if i in by_id:  # (a row deleted between the two statements is simply absent)
                            take(by_id[i])
        else:
            offset = 0
            while True:
                with self._svc._create_uow() as uow:
                    page = getattr(uow, repo_name).get_all(limit=self.EXPORT_PAGE, offset=offset)
                    for c in page:
                        take(c)
                if len(page) < self.EXPORT_PAGE:
                    break
                offset += len(page)
        if len(set(ids)) != len(ids):
            raise RuntimeError(f"{repo_name}: the export returned {len(ids) - len(set(ids))} duplicate primary keys "
                               "(unordered paging over a table that changed?)")
        # Rows in PRIMARY-KEY order whatever the statements returned: index row ids (and, one process per GPU, every rank's
        # global row ids) are positions in this order.  The reference's ImageChunkRepository has no `get_all_ids`, so that
        # table still pages with an unordered `get_all(limit, offset)`; N ranks scanning it at once is exactly what
        # PostgreSQL's synchronize_seqscans reorders.  (Ties between equal distances fall to the smaller key -- the
        # reference leaves that order to the database.)  service._unit() compares a digest of the result across ranks.
        order = None
        try:
            if not all(ids[i] <= ids[i + 1] for i in range(len(ids) - 1)):   # one linear pass; usually already ordered
                order = sorted(range(len(ids)), key=ids.__getitem__)
        except TypeError as e:  # mixed key types cannot be ordered: no deterministic export exists
            if self.require_total_order:
                raise RuntimeError(f"{repo_name}: primary keys of mixed types cannot be ordered") from e
            import logging  # noqa: PLC0415

            logging.getLogger("AutoRAG-Research").warning(
                f"{repo_name}: primary keys of mixed types cannot be ordered; keeping the order the repository returned "
                "(ties between equal distances then follow that order)")
        if order is not None:
            ids, contents = [ids[i] for i in order], [contents[i] for i in order]
            single, multi = [single[i] for i in order], [multi[i] for i in order]
        t = ChunkTable(ids=ids, contents=contents)
        d1 = next((v.shape[0] for v in single if v is not None), 0)
        if d1:
            t.embedding = np.full((len(ids), d1), np.nan, dtype=np.float32)  # NaN row = NULL (skipped like `IS NOT NULL`)
            for i, v in enumerate(single):
                if v is not None:
                    t.embedding[i] = v
        if any(m is not None and len(m) for m in multi):
            dm = next(m.shape[1] for m in multi if m is not None and len(m))
            lens = [0 if m is None else m.shape[0] for m in multi]
            t.mv_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            t.mv_tokens = np.concatenate([m.reshape(-1, dm) for m in multi if m is not None and len(m)], axis=0)
        self._tables[repo_name] = t
        return t

    @property
    def chunks(self) -> ChunkTable:
        return self._export("chunks")

    @property
    def image_chunks(self) -> ChunkTable:
        return self._export("image_chunks")

    # ---- results ----
    @staticmethod
    def _result_repo(uow, unit: str):
        return uow.image_chunk_results if unit == "image_chunk" else uow.chunk_results

    def completed_query_ids(self, unit: str, pipeline_id, query_ids) -> set:
        with self._svc._create_uow() as uow:
            return {r.query_id for r in self._result_repo(uow, unit).get_by_query_and_pipeline(list(query_ids), pipeline_id)}

    def bulk_insert(self, unit: str, rows: list[dict[str, Any]]) -> None:
        with self._svc._create_uow() as uow:
            self._result_repo(uow, unit).bulk_insert(rows)
            uow.commit()
