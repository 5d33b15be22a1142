"""Mi355RetrievalService -- host-side mirror of RetrievalPipelineService for the Vector Search path.

Same method names, argument meaning, return shapes and error behaviour as the reference
(autorag_research/orm/service/retrieval_pipeline.py):
  vector_search(query_ids, top_k, search_mode)          :467-525  score = 1 - distance | -distance / n_q
  vector_search_by_embedding(embedding, top_k)          :527-550
  run_pipeline / run_image_pipeline / _run_pipeline     :184-357  paging, resume-skip, retry, persistence, stats
  _collect_retrieval_results                            :151-182
  _make_retrieval_result                                :374-384  {"doc_id","score","content"}
  find_query_by_text                                    :386-400
The difference is WHERE the arithmetic runs: the two SQL operators the reference sends to PostgreSQL
(orm/repository/base.py:409-415 `<=>`, :518-524 `@#`) are answered by libmi355dr on the GPU, and a whole
list of query ids is scored as ONE block per corpus pass instead of one SQL statement per query.
"""

from __future__ import annotations

import asyncio
import logging
import os
from collections.abc import Awaitable, Callable
from typing import Any, Literal

import numpy as np

from .index import Mi355Index
from .store import ChunkTable, InMemoryStore, UowStore

logger = logging.getLogger("AutoRAG-Research")

RetrievalFunc = Callable[[int | str, int], Awaitable[list[dict[str, Any]]]]


class _World:
    """The process group a pipeline runs in when it is launched one process per GPU (`torchrun --nproc-per-node N`, or any
    launcher that initialises torch.distributed before the Executor constructs its pipelines): every rank constructs the same
    pipeline over the same database; the corpus is ROW-SHARDED over the ranks (multi-vector tables by cumulative token
    count), every page of `run()` is answered by all ranks together -- local top-k, one all-gather over RCCL / xGMI, merge --
    and rank 0 alone reads the page's query ids and writes the results (reference caller: executor.py:383-463 ->
    pipelines/retrieval/base.py:156-199 -> retrieval_pipeline.py:184-307)."""

    def __init__(self, dist: Any):
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        # The plugin's own collectives can be given a deadline: MI355DR_COLLECTIVE_TIMEOUT_S = seconds makes every rank create
        # (together: `new_group` is itself a collective, and `detect()` runs on every rank when the pipeline is constructed) a
        # group over the whole world with that timeout, and every collective below -- and the sharded searchers' -- uses it.
        # A rank that died or left through a rank-local error while its peers are inside a collective then costs them that many
        # seconds and a RuntimeError instead of the launcher's default (30 min under gloo, 10 under nccl) -- `agree` reconciles
        # failures OUTSIDE collectives only.  Unset / 0: the default group, as before.
        self.group = None
        try:
            t = float(os.environ.get("MI355DR_COLLECTIVE_TIMEOUT_S", "0") or 0)
        except ValueError:
            t = 0.0
        if t > 0:
            import datetime  # noqa: PLC0415

            self.group = dist.new_group(ranks=list(range(self.size)), timeout=datetime.timedelta(seconds=t))

    @classmethod
    def detect(cls) -> "_World | None":
        if os.environ.get("MI355DR_DISTRIBUTED", "1") == "0":
            return None
        try:
            import torch.distributed as dist  # noqa: PLC0415
        except ImportError:  # pragma: no cover
            return None
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
            return None
        return cls(dist)

    def from_root(self, make: Callable[[], Any]) -> Any:
        """`make()` on rank 0, its (picklable) value on every rank.  An exception raised by `make()` on rank 0 travels the
        same way and is re-raised on EVERY rank: rank 0 always reaches the broadcast, so the other ranks never block on a
        collective that rank 0 left through an exception (a database error in `get_or_create_pipeline` / `next_page`)."""
        box: list[Any] = [None]
        if self.rank == 0:
            try:
                box[0] = (True, make())
            except Exception as e:  # noqa: BLE001 - re-raised below, on every rank
                box[0] = (False, self._portable(e))
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        ok, value = box[0]
        if not ok:
            raise value
        return value

    @staticmethod
    def _portable(e: Exception) -> Exception:
        """The exception itself when it pickles (ValueError, KeyError ... the ones the reference's callers test for), else a
        RuntimeError carrying its text."""
        import pickle  # noqa: PLC0415

        try:
            pickle.loads(pickle.dumps(e))
        except Exception:  # noqa: BLE001
            return RuntimeError(f"{type(e).__name__}: {e}")
        return e

    def agree(self, ok: bool) -> bool:
        """True iff `ok` on every rank: ranks settle the outcome of a rank-local step BEFORE any of them branches into a
        different sequence of collectives (block answer vs per-query fallback vs retry)."""
        flags: list[Any] = [None] * self.size
        self.dist.all_gather_object(flags, bool(ok), group=self.group)
        return all(flags)

    def same_everywhere(self, what: str, digest: str) -> None:
        """Fail loudly, on every rank, when the ranks did not compute the same `digest` (the exported table's key order:
        global row ids are positions in that order, so two ranks that saw different orders would silently map each other's
        rows to the wrong primary keys)."""
        seen: list[Any] = [None] * self.size
        self.dist.all_gather_object(seen, digest, group=self.group)
        if any(d != seen[0] for d in seen):
            raise RuntimeError(f"{what}: the ranks exported different tables ({len(set(seen))} distinct digests over "
                               f"{self.size} ranks) -- the table changed during the export or the keys are not totally "
                               "ordered; refusing to search with inconsistent global row ids")

    def bind_device(self, device: int) -> None:
        """nccl collectives and broadcast_object_list run on torch's CURRENT device: make it this rank's GPU (a launcher
        that did not call torch.cuda.set_device would run every rank's collectives on cuda:0 -- a duplicate-GPU error or a
        hang).  No-op on the gloo backend."""
        if self.dist.get_backend() != "nccl":
            return
        import torch  # noqa: PLC0415

        torch.cuda.set_device(device)


def _table_digest(table: ChunkTable) -> str:
    """Key order + NULL pattern of an exported table (what global row ids depend on)."""
    import hashlib  # noqa: PLC0415

    h = hashlib.sha256()
    h.update(repr(len(table.ids)).encode())
    for pk in table.ids:
        h.update(repr(pk).encode())
        h.update(b"\x00")
    if table.embedding is not None:
        h.update(np.ascontiguousarray(np.isnan(table.embedding).all(axis=1)).tobytes())
    if table.mv_offsets is not None:
        h.update(np.ascontiguousarray(table.mv_offsets, dtype=np.int64).tobytes())
    return h.hexdigest()


class _UnitIndex:
    """GPU index of one table (chunk / image_chunk): row <-> primary-key mapping + the native handle."""

    def __init__(self, table: ChunkTable, device: int):
        self.table = table
        self.single: Mi355Index | None = None
        self.multi: Mi355Index | None = None
        self.single_rows: np.ndarray | None = None  # index row -> table position (NULL embeddings skipped)
        self.multi_rows: np.ndarray | None = None
        self.device = device
        self.single_sharded: Any | None = None      # ShardedSearcher over this rank's rows (a _World is active)
        self.multi_sharded: Any | None = None

    def ensure_single_sharded(self, world: "_World") -> Any:
        """This rank's contiguous share of the NOT NULL rows behind a ShardedSearcher (global row ids = positions in the
        table's not-null order, the same ids the unsharded index uses)."""
        if self.single_sharded is None:
            from .sharded import ShardedSearcher, shard_bounds  # noqa: PLC0415

            emb = self.table.embedding
            if emb is None:
                raise ValueError("table has no single-vector embeddings")
            not_null = ~np.isnan(emb).all(axis=1)
            self.single_rows = np.nonzero(not_null)[0]
            lo, hi = shard_bounds(int(self.single_rows.shape[0]), world.size, world.rank)
            s = ShardedSearcher(emb.shape[1], "cosine", self.device, index_factory=Mi355Index, group=world.group)
            s.add_local(emb[self.single_rows[lo:hi]], lo)
            self.single_sharded = s
        return self.single_sharded

    def ensure_multi_sharded(self, world: "_World") -> Any:
        """Multi-vector table: docs cut by cumulative TOKEN count (the MaxSim pass streams token rows: SURVEY 8(e))."""
        if self.multi_sharded is None:
            from .sharded import ShardedSearcher, shard_bounds_by_tokens  # noqa: PLC0415

            tok, off = self.table.mv_tokens, self.table.mv_offsets
            if tok is None or off is None:
                raise ValueError("table has no multi-vector embeddings")
            lo, hi = shard_bounds_by_tokens(off, world.size, world.rank)
            s = ShardedSearcher(tok.shape[1], "cosine", self.device, index_factory=Mi355Index, group=world.group)
            s.add_local_multivec(tok[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], lo)
            self.multi_sharded = s
            self.multi_rows = np.arange(off.shape[0] - 1)
        return self.multi_sharded

    def single_positions(self) -> np.ndarray:
        """index row -> table position for the single-vector column (`WHERE embedding IS NOT NULL` order), without building
        an index (one process per GPU: no rank holds the whole table on its GPU)."""
        if self.single_rows is None:
            emb = self.table.embedding
            if emb is None:
                raise ValueError("table has no single-vector embeddings")
            self.single_rows = np.nonzero(~np.isnan(emb).all(axis=1))[0]
        return self.single_rows

    def ensure_single(self) -> Mi355Index:
        if self.single is None:
            emb = self.table.embedding
            if emb is None:
                raise ValueError("table has no single-vector embeddings")
            not_null = ~np.isnan(emb).all(axis=1)  # WHERE embedding IS NOT NULL
            self.single_rows = np.nonzero(not_null)[0]
            self.single = Mi355Index(emb.shape[1], "cosine", self.device)
            self.single.add(emb[not_null] if not not_null.all() else emb)
        return self.single

    def ensure_multi(self) -> Mi355Index:
        if self.multi is None:
            tok, off = self.table.mv_tokens, self.table.mv_offsets
            if tok is None or off is None:
                raise ValueError("table has no multi-vector embeddings")
            self.multi = Mi355Index(tok.shape[1], "cosine", self.device)
            self.multi.add_multivec(tok, off)
            self.multi_rows = np.arange(off.shape[0] - 1)
        return self.multi

    def close(self) -> None:
        for ix in (self.single, self.multi, self.single_sharded, self.multi_sharded):
            if ix is not None:
                ix.close()
        self.single = self.multi = self.single_sharded = self.multi_sharded = None


def _is_store(obj: Any) -> bool:
    return all(hasattr(obj, a) for a in ("get_or_create_pipeline", "get_all_queries", "completed_query_ids", "chunks"))


class Mi355RetrievalService:
    """`session_factory` is what the caller of the pipeline passes:

    * a callable returning a store (InMemoryStore, or any object with that interface) -- standalone use, bench, tests;
    * a SQLAlchemy `sessionmaker`, which is what the reference's Executor and its wrapper pipelines pass
      (executor.py:326-333, hybrid.py `_load_pipeline`).  The service then builds the reference's own
      `RetrievalPipelineService(session_factory, schema)` and reads / writes the database through it (store.UowStore);
      that needs the `autorag_research` package importable, which it is wherever the Executor runs.
    """

    def __init__(self, session_factory: Callable[[], Any], schema: Any | None = None, device: int = 0):
        self.session_factory = session_factory
        self._schema = schema
        # one process per GPU under torch.distributed: this rank's GPU, row-sharded units, rank 0 reads ids / writes results
        self._world = _World.detect()
        if self._world is not None and "LOCAL_RANK" in os.environ:
            device = int(os.environ["LOCAL_RANK"])
        self._device = device
        if self._world is not None:
            self._world.bind_device(device)
        self._units: dict[str, _UnitIndex] = {}
        self._uow_store: UowStore | None = None
        probe = session_factory()
        if not _is_store(probe):
            close = getattr(probe, "close", None)
            if callable(close):
                close()
            try:
                from autorag_research.orm.service.retrieval_pipeline import RetrievalPipelineService  # noqa: PLC0415
            except ImportError as e:  # pragma: no cover - only outside a reference installation
                raise TypeError(
                    f"session_factory() returned a {type(probe).__name__}, not a store; a SQLAlchemy sessionmaker is served "
                    "through the reference's RetrievalPipelineService, which needs `autorag_research` importable") from e
            self._uow_store = UowStore(RetrievalPipelineService(session_factory, schema), require_total_order=self._world is not None)

    # ---- plumbing ----
    def _store(self) -> Any:
        return self._uow_store if self._uow_store is not None else self.session_factory()

    def delete_pipeline_results(self, pipeline_id) -> int:
        """Reference RetrievalPipelineService.delete_pipeline_results (:359-372): the Executor's health-check cleanup."""
        if self._world is not None:
            return self._world.from_root(lambda: self._store().delete_pipeline_results(pipeline_id))
        return self._store().delete_pipeline_results(pipeline_id)

    def _unit(self, unit: str) -> _UnitIndex:
        if unit not in self._units:
            store = self._store()
            table = store.image_chunks if unit == "image_chunk" else store.chunks
            if self._world is not None:
                # every rank exported the table by itself: global row ids are positions in the export order, so the ranks
                # must have seen the SAME keys in the SAME order with the same NULL pattern before any of them shards it
                self._world.same_everywhere(f"table {unit!r}", _table_digest(table))
            self._units[unit] = _UnitIndex(table, self._device)
        return self._units[unit]

    def get_queries(self, query_ids: list) -> list:
        """The stored query rows (None = no such query).  One process per GPU: rank 0 reads them and every rank gets the same
        rows -- a transient database error then happens once, on rank 0, and reaches every rank as the same exception instead
        of sending one rank down the retry path while the others wait in the block's collective."""
        read = lambda: [self._store().get_query(q) for q in query_ids]  # noqa: E731
        return self._world.from_root(read) if self._world is not None else read()

    async def on_root(self, make: Callable[[], Awaitable[Any]]) -> Any:
        """`await make()` -- under a _World on rank 0 only, its value (or its exception) on every rank: an embedding-model
        call in `_retrieve_by_text` is a rank-local step that may fail or differ between ranks."""
        if self._world is None:
            return await make()
        box: list[Any] = [None]
        if self._world.rank == 0:
            try:
                box[0] = (True, await make())
            except Exception as e:  # noqa: BLE001
                box[0] = (False, _World._portable(e))
        self._world.dist.broadcast_object_list(box, src=0, group=self._world.group)
        ok, value = box[0]
        if not ok:
            raise value
        return value

    def close(self) -> None:
        for u in self._units.values():
            u.close()
        self._units.clear()
        if getattr(self, "_scratch", None) is not None:
            self._scratch.close()
            self._scratch = None

    def get_or_create_pipeline(self, name: str, config: dict[str, Any]) -> tuple[int, bool]:
        if self._world is not None:  # one row in the pipeline table, created by rank 0
            return tuple(self._world.from_root(lambda: tuple(self._store().get_or_create_pipeline(name, config))))
        return self._store().get_or_create_pipeline(name, config)

    def find_query_by_text(self, query_text: str):
        if self._world is not None:
            return self._world.from_root(lambda: self._store().find_query_by_text(query_text))
        return self._store().find_query_by_text(query_text)

    def _make_retrieval_result(self, table: ChunkTable, pos: int, score: float, with_content: bool) -> dict[str, Any]:
        return {"doc_id": table.ids[pos], "score": score, "content": table.contents[pos] if with_content else None}

    # ---- the hot path ----
    def vector_search(self, query_ids: list[int | str], top_k: int = 10,
                      search_mode: Literal["single", "multi"] = "single", unit: str = "chunk") -> list[list[dict]]:
        """Top-k for every query id, scored as one block.  Raises ValueError exactly like the reference."""
        queries = []
        for qid, q in zip(query_ids, self.get_queries(list(query_ids)), strict=True):
            if q is None:
                raise ValueError(f"Query {qid} not found")  # noqa: TRY003
            if search_mode == "multi":
                if q.embeddings is None:
                    raise ValueError(f"Query {qid} has no multi-vector embeddings")  # noqa: TRY003
            elif q.embedding is None:
                msg = f"Query {qid} has no embedding" if unit == "chunk" else f"Query {qid} has no single-vector embedding"
                raise ValueError(msg)
            queries.append(q)
        if not queries:
            return []
        if search_mode == "multi":
            return self.maxsim_search_by_embeddings([q.embeddings for q in queries], top_k, unit)
        Q = np.stack([q.embedding for q in queries]).astype(np.float32, copy=False)
        return self._single_block(Q, top_k, unit)

    @staticmethod
    def _results_from_block(table: ChunkTable, pos_of_row: np.ndarray, rows: np.ndarray, scores: np.ndarray,
                            with_content: bool) -> list[list[dict]]:
        """[B,k] index rows (-1 padded at the tail) + float64 scores -> the reference's list of result dicts per query.
        Built from whole-array conversions: a page is B*k results, and one Python-level numpy access per result costs
        more than the GPU pass that produced them."""
        k = rows.shape[1]
        neg = rows < 0
        n_valid = np.where(neg.any(axis=1), neg.argmax(axis=1), k).tolist()
        pos = pos_of_row[np.where(neg, 0, rows)].tolist()
        sc = scores.tolist()
        ids, contents = table.ids, table.contents
        if with_content:
            return [[{"doc_id": ids[p], "score": s, "content": contents[p]} for p, s in zip(pr[:n], sr[:n])]
                    for pr, sr, n in zip(pos, sc, n_valid)]
        return [[{"doc_id": ids[p], "score": s, "content": None} for p, s in zip(pr[:n], sr[:n])]
                for pr, sr, n in zip(pos, sc, n_valid)]

    def _single_block(self, Q: np.ndarray, top_k: int, unit: str) -> list[list[dict]]:
        u = self._unit(unit)
        if self._world is not None:  # every rank: local top-k of its rows, all-gather, merge -> the same global lists
            dist, rows = u.ensure_single_sharded(self._world).search(Q, top_k)
        else:
            dist, rows = u.ensure_single().search(Q, top_k)
        # reference: score = 1 - distance (retrieval_pipeline.py:522-524) in Python float arithmetic = IEEE double
        return self._results_from_block(u.table, u.single_rows, rows, 1.0 - dist, unit == "chunk")

    def vector_search_by_embedding(self, embedding: list[float], top_k: int = 10, unit: str = "chunk") -> list[dict]:
        if len(embedding) == 0:  # reference: `if not query_vector: return []` (base.py:403-404)
            return []
        return self._single_block(np.asarray(embedding, dtype=np.float32)[None, :], top_k, unit)[0]

    def maxsim_search_by_embeddings(self, query_vectors: list, top_k: int, unit: str = "chunk") -> list[list[dict]]:
        u = self._unit(unit)
        ix = u.ensure_multi_sharded(self._world) if self._world is not None else u.ensure_multi()
        dim = u.table.mv_tokens.shape[1]
        mats = [np.asarray(qv, dtype=np.float32).reshape(-1, dim) for qv in query_vectors]
        lens = [m.shape[0] for m in mats]
        live = [i for i, n in enumerate(lens) if n > 0]
        out: list[list[dict]] = [[] for _ in mats]  # reference: `if not query_vectors: return []`
        if not live:
            return out
        qtok = np.concatenate([mats[i] for i in live], axis=0)
        qoff = np.concatenate([[0], np.cumsum([lens[i] for i in live])]).astype(np.int32)
        dist, rows = ix.search_maxsim(qtok, qoff, top_k)
        # reference: score = -distance / n_query_vectors (retrieval_pipeline.py:511-514): float(f32) negated, divided
        n_q = np.maximum(1, np.asarray([lens[i] for i in live], dtype=np.int64))[:, None]
        scores = -dist.astype(np.float64) / n_q
        for i, res in zip(live, self._results_from_block(u.table, u.multi_rows, rows, scores, unit == "chunk")):
            out[i] = res
        return out

    def maxsim_score_candidates(self, query_vectors, doc_ids: list, unit: str = "chunk") -> dict:
        """Late-interaction score of explicit candidates: {doc_id: mean_i max_j <q_i, d_j>} (reference HEAVEN
        `_score_candidates`, heaven.py:244-266).  Ids unknown to the table or without multi-vector embeddings are left
        out (as `_fetch_candidate_multi_embeddings` does, heaven.py:224-241); no query vectors -> every score 0.0."""
        u = self._unit(unit)
        pos = getattr(u, "_pos_of_id", None)
        if pos is None:
            pos = u._pos_of_id = {pk: i for i, pk in enumerate(u.table.ids)}
        off = u.table.mv_offsets
        known = [(pk, pos[pk]) for pk in doc_ids if pk in pos and off is not None and off[pos[pk] + 1] > off[pos[pk]]]
        if not known:
            return {}
        q = np.asarray(query_vectors, dtype=np.float32)
        if q.size == 0:
            return {pk: 0.0 for pk, _ in known}
        # one process per GPU: the token-sharded store -- every rank scores the candidates it owns, one all-gather of [1, m] fp32
        ix = u.ensure_multi_sharded(self._world) if self._world is not None else u.ensure_multi()
        q = q.reshape(-1, u.table.mv_tokens.shape[1])
        rows = np.array([[p for _, p in known]], dtype=np.int64)
        dist = ix.maxsim_subset(q, np.array([0, q.shape[0]], dtype=np.int32), rows)[0]
        return {pk: -float(dv) / q.shape[0] for (pk, _), dv in zip(known, dist) if dv == dv}

    # ---- Guided Query Refinement support (reference retrieval_pipeline.py:573-641 + gqr_hybrid.py:306-362) ----
    def get_query_embedding(self, query_id) -> np.ndarray | None:
        """Stored single-vector query embedding as float64, None when the query or its embedding is missing (:573-587)."""
        q = self.get_queries([query_id])[0]
        return None if q is None or q.embedding is None else np.asarray(q.embedding, dtype=np.float64)

    def get_query_multi_embedding(self, query_id) -> np.ndarray | None:
        """Stored multi-vector query embedding [n_q, d] float64, or None (:589-603)."""
        q = self.get_queries([query_id])[0]
        return None if q is None or q.embeddings is None else np.asarray(q.embeddings, dtype=np.float64)

    def _gqr_handle(self) -> Mi355Index:
        """Any native handle (the score-space refinement needs a device, not an index)."""
        u = self._unit("chunk")
        if u.single is not None or u.multi is not None:
            return u.single if u.single is not None else u.multi
        if getattr(self, "_scratch", None) is None:
            self._scratch = Mi355Index(8, "cosine", self._device)
        return self._scratch

    def chunk_rows_single(self, doc_ids: list) -> np.ndarray | None:
        """Index rows of chunks, or None when one of them has no stored single-vector embedding (the reference's
        `len(embedding_ids) == len(candidate_ids)` test, gqr_hybrid.py:454-456)."""
        u = self._unit("chunk")
        if u.table.embedding is None:
            return None
        if self._world is not None:
            u.single_positions()   # (the mapping only: the rows themselves are staged per page, see gqr_refine_single)
        else:
            u.ensure_single()
        inv = getattr(u, "_row_of_pos", None)
        if inv is None:
            inv = u._row_of_pos = {int(p): r for r, p in enumerate(u.single_rows)}
            u._pos_of_id = getattr(u, "_pos_of_id", None) or {pk: i for i, pk in enumerate(u.table.ids)}
        rows = [inv.get(u._pos_of_id.get(pk, -1), -1) for pk in doc_ids]
        return None if any(r < 0 for r in rows) else np.asarray(rows, dtype=np.int64)

    def chunk_rows_multi(self, doc_ids: list) -> np.ndarray | None:
        """Same for multi-vector embeddings (gqr_hybrid.py:439-441)."""
        u = self._unit("chunk")
        off = u.table.mv_offsets
        if off is None:
            return None
        if self._world is None:
            u.ensure_multi()
        if getattr(u, "_pos_of_id", None) is None:
            u._pos_of_id = {pk: i for i, pk in enumerate(u.table.ids)}
        pos = [u._pos_of_id.get(pk, -1) for pk in doc_ids]
        if any(p < 0 or off[p + 1] <= off[p] for p in pos):
            return None
        return np.asarray(pos, dtype=np.int64)

    # The refinement is a loop of n_steps (25) dependent steps over a pool of a few dozen candidates per query: under a
    # _World it is NOT spread over the ranks (a collective per step for a few KB of work).  The pools' vectors -- a page's
    # worth: kilobytes to a few MB -- are staged from the exported table into a scratch store on every rank and refined there,
    # identically on every rank; no rank ever holds the whole table on its GPU for it (reference gqr_hybrid.py:366-406
    # fetches the same vectors out of PostgreSQL per query).
    @staticmethod
    def _compact_pools(pools: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """[B,P] rows (-1 padded) -> (the distinct rows, ascending; the same pools in positions of that list)."""
        pools = np.asarray(pools, dtype=np.int64)
        live = pools >= 0
        uniq, inv = np.unique(pools[live], return_inverse=True)
        local = np.full(pools.shape, -1, dtype=np.int64)
        local[live] = inv
        return uniq, local

    def gqr_refine_single(self, queries: np.ndarray, pools: np.ndarray, comp: np.ndarray, **prm) -> np.ndarray:
        u = self._unit("chunk")
        if self._world is None:
            return u.ensure_single().gqr_refine(queries, pools, comp, **prm)
        uniq, local = self._compact_pools(pools)
        emb = u.table.embedding[u.single_positions()[uniq]]
        with Mi355Index(emb.shape[1], "cosine", self._device) as scratch:
            scratch.add(emb)
            return scratch.gqr_refine(queries, local, comp, **prm)

    def gqr_refine_multi(self, qtok: np.ndarray, q_offsets: np.ndarray, pools: np.ndarray, comp: np.ndarray,
                         **prm) -> np.ndarray:
        u = self._unit("chunk")
        if self._world is None:
            return u.ensure_multi().gqr_refine_maxsim(qtok, q_offsets, pools, comp, **prm)
        uniq, local = self._compact_pools(pools)   # (multi-vector rows are table positions)
        tok, off = u.table.mv_tokens, u.table.mv_offsets
        lens = off[uniq + 1] - off[uniq]
        sub = np.concatenate([tok[off[p]:off[p + 1]] for p in uniq], axis=0) if len(uniq) else tok[:0]
        with Mi355Index(tok.shape[1], "cosine", self._device) as scratch:
            scratch.add_multivec(sub, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
            return scratch.gqr_refine_maxsim(qtok, q_offsets, local, comp, **prm)

    def gqr_refine_scores(self, primary: np.ndarray, counts: np.ndarray, comp: np.ndarray, **prm) -> np.ndarray:
        return self._gqr_handle().gqr_refine_scores(primary, counts, comp, **prm)

    # ---- batch driver (reference _run_pipeline) ----
    @staticmethod
    def _collect_retrieval_results(query_ids, results, pipeline_id, failed_queries, result_id_key) -> list[dict]:
        rows = []
        for qid, res in zip(query_ids, results, strict=True):
            if res is None:
                failed_queries.append(qid)
                continue
            for r in res:
                rows.append({"query_id": qid, "pipeline_id": pipeline_id, result_id_key: r["doc_id"], "rel_score": r["score"]})
        return rows

    def _run_pipeline(self, retrieval_func: RetrievalFunc | None, pipeline_id: int, unit: str, top_k: int = 10,
                      batch_size: int = 128, max_concurrency: int = 16, max_retries: int = 3, retry_delay: float = 1.0,
                      query_limit: int | None = None,
                      block_func: Callable[[list, int], list[list[dict] | None]] | None = None) -> dict[str, Any]:
        """Page through queries, skip completed ones, retrieve, persist; returns the reference's stats dict.

        `retrieval_func` is the reference's per-query coroutine contract (retry with exponential backoff,
        at most `max_concurrency` in flight).  `block_func`, when given, scores a whole page of ids in one
        GPU block instead (results aligned with the ids, None = failed query).
        """
        store = self._store()
        result_id_key = "image_chunk_id" if unit == "image_chunk" else "chunk_id"
        read_unit = lambda: store.pipeline_config(pipeline_id).get("retrieval_unit", "chunk")  # noqa: E731
        configured = self._world.from_root(read_unit) if self._world is not None else read_unit()
        if configured == "mixed":
            raise ValueError(f"Pipeline {pipeline_id!r} is configured for mixed results, which cannot be persisted directly.")
        if configured != unit:
            raise ValueError(f"Pipeline {pipeline_id!r} is configured for {configured} results; "
                             f"refusing to persist {unit} results into the same pipeline identity.")

        world = self._world

        async def one(qid) -> list[dict] | None:
            assert retrieval_func is not None
            delay = retry_delay
            for attempt in range(max(1, max_retries)):
                res, err = None, None
                try:
                    res = await retrieval_func(qid, top_k)
                except Exception as e:  # noqa: BLE001
                    err = e
                # one process per GPU: an attempt counts only if it succeeded on EVERY rank -- all ranks then retry (or give
                # the query up) together, and their sequences of collective searches stay aligned
                if (err is None) if world is None else world.agree(err is None):
                    return res
                if attempt + 1 >= max(1, max_retries):
                    logger.error(f"Retrieval failed for query {qid} after {max_retries} attempts", exc_info=err)
                    return None
                await asyncio.sleep(min(max(delay, retry_delay), 60))
                delay *= 2
            return None

        async def page(qids) -> list[list[dict] | None]:
            sem = asyncio.Semaphore(max(1, max_concurrency))

            async def guarded(q):
                async with sem:
                    return await one(q)

            return list(await asyncio.gather(*[guarded(q) for q in qids]))

        writer = world is None or world.rank == 0  # one process per GPU: rank 0 reads the page's ids and persists
        if world is not None:
            max_concurrency = 1  # the per-query fallback is a sequence of collective searches: the same order on every rank

        def next_page(eff: int, offset: int):
            """(number of queries on the page, ids still to answer) -- (0, []) at the end."""
            queries = store.get_all_queries(limit=eff, offset=offset)
            if not queries:
                return 0, []
            ids = [q.id for q in queries]
            done = store.completed_query_ids(unit, pipeline_id, ids)
            return len(queries), [q for q in ids if q not in done]

        total_queries = total_results = offset = 0
        failed: list = []
        while True:
            if query_limit is not None and total_queries >= query_limit:
                break
            eff = min(batch_size, query_limit - total_queries) if query_limit is not None else batch_size
            n_page, qids = world.from_root(lambda: next_page(eff, offset)) if world is not None else next_page(eff, offset)
            if n_page == 0:
                break
            if not qids:
                offset += batch_size
                continue
            if block_func is not None:
                results, block_err = None, None
                try:
                    results = block_func(qids, top_k)
                except Exception as e:  # noqa: BLE001
                    block_err = e
                # (one process per GPU: the page is answered as a block only if every rank's block succeeded.  LIMIT of this
                # agreement: it reconciles failures that happen OUTSIDE a collective.  A rank that dies, or raises from a
                # rank-local HIP / out-of-memory error, while its peers are already inside the block's all-gather leaves them
                # blocked there -- what ends that is the process group's own timeout (`init_process_group(timeout=...)`,
                # torchrun's failure detection), not this code; INTEGRATION.md "one process per GPU" says so.)
                if not ((block_err is None) if world is None else world.agree(block_err is None)):
                    # one bad query (missing / malformed embedding, too many query vectors for a block, an embedding-batch
                    # error) must not abort the run: the page falls back to the reference's per-query path, where retries,
                    # backoff and `failed_queries` apply to that query alone (retrieval_pipeline.py:222-236)
                    logger.error(f"block retrieval failed for a page of {len(qids)} queries; retrying it query by query",
                                 exc_info=block_err)
                    if retrieval_func is None:
                        raise block_err if block_err is not None else RuntimeError("block retrieval failed on another rank")
                    results = asyncio.run(page(qids))
            else:
                results = asyncio.run(page(qids))
            insert_page = getattr(store, "insert_page", None)
            if not writer:  # (the same lists on every rank: count what rank 0 stores)
                failed.extend(q for q, r in zip(qids, results, strict=True) if r is None)
                total_results += sum(len(r) for r in results if r is not None)
            elif callable(insert_page):
                # a store that takes a page as it is (ranked lists per query): skips flattening it into one dict per
                # result row only to regroup them by query again
                failed.extend(q for q, r in zip(qids, results, strict=True) if r is None)
                total_results += insert_page(unit, pipeline_id, qids, results)
            else:
                rows = self._collect_retrieval_results(qids, results, pipeline_id, failed, result_id_key)
                if rows:
                    store.bulk_insert(unit, rows)
                    total_results += len(rows)
            total_queries += len([r for r in results if r is not None])
            offset += n_page
            logger.info(f"Processed {total_queries} queries, stored {total_results} results")
        if failed:
            logger.warning(f"Failed to process {len(failed)} queries after retries: {failed}")
        return {"pipeline_id": pipeline_id, "total_queries": total_queries, "total_results": total_results,
                "failed_queries": failed}

    def run_pipeline(self, retrieval_func: RetrievalFunc, pipeline_id: int, **kw) -> dict[str, Any]:
        return self._run_pipeline(retrieval_func, pipeline_id, "chunk", **kw)

    def run_image_pipeline(self, retrieval_func: RetrievalFunc, pipeline_id: int, **kw) -> dict[str, Any]:
        return self._run_pipeline(retrieval_func, pipeline_id, "image_chunk", **kw)
