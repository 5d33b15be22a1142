"""Retrieval-pipeline plugin classes for the MI355X path (the drop-in boundary on the Python side).

Mirrors, name for name and signature for signature:
  VectorSearchPipelineConfig / VectorSearchRetrievalPipeline          pipelines/retrieval/vector_search.py:19-191
  ImageVectorSearchPipelineConfig / ImageVectorSearchRetrievalPipeline pipelines/retrieval/image_vector_search.py:22-139
  BaseRetrievalPipeline (retrieve / run / abstract hooks)              pipelines/retrieval/base.py:49-199
Discovery: entry-point group "autorag_research.pipelines" -> this PACKAGE (the registry scans its directory), YAML
`retrieval/mi355_vector_search.yaml` with `_target_: autorag_research_amd.pipelines.Mi355VectorSearchPipelineConfig`
(reference plugin_registry.py:199-255, docs/plugins/retrieval-pipeline.md).

What is new relative to the reference: `run()` scores a whole page of query ids as ONE block on the GPU
(the reference issues one SQL top-k per query, serially: vector_search.py:167-169) while `_retrieve_by_id`
/ `_retrieve_by_text` keep the per-query contract that wrapper pipelines (hybrid, HyDE, HEAVEN) call.
"""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from inspect import getattr_static
from typing import Any, Literal

from .compat import BasePipeline, BaseRetrievalPipelineConfig, EmbeddingError, require_retrieval_unit
from .service import Mi355RetrievalService

logger = logging.getLogger("AutoRAG-Research")
_MISSING = object()


def get_retrieval_pipeline_unit(pipeline: object):
    """Typed `retrieval_unit` attribute first, persisted config second (reference base.py:34-46)."""
    if getattr_static(pipeline, "retrieval_unit", _MISSING) is not _MISSING:
        unit = require_retrieval_unit(pipeline.retrieval_unit)  # type: ignore[attr-defined]
        if unit is not None:
            return unit
    get_config = getattr(pipeline, "_get_pipeline_config", None)
    config = get_config() if callable(get_config) else {}
    return require_retrieval_unit(config.get("retrieval_unit") if isinstance(config, dict) else None)


class Mi355BaseRetrievalPipeline(BasePipeline, ABC):
    """BaseRetrievalPipeline contract over Mi355RetrievalService (reference pipelines/retrieval/base.py:49-199)."""

    retrieval_unit: str | None = None

    def __init__(self, session_factory: Any, name: str, schema: Any | None = None, device: int = 0):
        super().__init__(session_factory, name, schema)
        self._service = Mi355RetrievalService(session_factory, schema, device=device)
        self.pipeline_id, self._is_new_pipeline = self._service.get_or_create_pipeline(
            name=name, config=self._get_pipeline_config())
        if not self._is_new_pipeline:
            logger.info(f"Resuming existing retrieval pipeline '{name}' (pipeline_id={self.pipeline_id})")

    @abstractmethod
    async def _retrieve_by_id(self, query_id: int | str, top_k: int) -> list[dict[str, Any]]: ...

    @abstractmethod
    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]: ...

    def _retrieve_block(self, query_ids: list, top_k: int) -> list[list[dict] | None]:  # pragma: no cover
        raise NotImplementedError

    async def retrieve(self, query_text: str, top_k: int = 10) -> list[dict[str, Any]]:
        query = self._service.find_query_by_text(query_text)
        if query is not None:
            return await self._retrieve_by_id(query.id, top_k)
        return await self._retrieve_by_text(query_text, top_k)

    def run(self, top_k: int = 10, batch_size: int = 128, max_concurrency: int = 16, max_retries: int = 3,
            retry_delay: float = 1.0, query_limit: int | None = None, block: bool = True) -> dict[str, Any]:
        """Same kwargs / stats dict as the reference's run(); `block=True` scores each page as one GPU block."""
        unit = get_retrieval_pipeline_unit(self) or "chunk"
        if unit == "mixed":
            raise ValueError("Mixed retrieval_unit persistence is not supported; override run() with an explicit persistence path.")
        self._run_retry = (max(1, max_retries), retry_delay)  # block forms that call fallible services (HyDE's LLM) honour it
        return self._service._run_pipeline(
            retrieval_func=self._retrieve_by_id, pipeline_id=self.pipeline_id, unit=unit, top_k=top_k,
            batch_size=batch_size, max_concurrency=max_concurrency, max_retries=max_retries, retry_delay=retry_delay,
            query_limit=query_limit, block_func=self._retrieve_block if block else None)

    def close(self) -> None:
        """Called by the reference Executor when present (executor.py:353-354, 449-450)."""
        self._service.close()


def child_page(child: Any, query_ids: list, top_k: int) -> list[list[dict] | None]:
    """What a child retrieval pipeline answers for a page of query ids (wrapper pipelines: hybrid fusion, GQR).  One GPU
    block when the child can (`_retrieve_block`), else query by query through its `_retrieve_by_id` contract; a query
    the child fails on is None (reported in `failed_queries` by the caller's run)."""
    import asyncio  # noqa: PLC0415

    block = getattr(child, "_retrieve_block", None)
    if callable(block):
        try:
            return block(query_ids, top_k)
        except NotImplementedError:
            pass

    async def one_by_one():
        out: list[list[dict] | None] = []
        for qid in query_ids:
            try:
                out.append(await child._retrieve_by_id(qid, top_k))
            except Exception:  # noqa: BLE001 - a failed child fails that query only
                logger.exception(f"child pipeline {getattr(child, 'name', child)!r} failed for query {qid}")
                out.append(None)
        return out

    return asyncio.run(one_by_one())


class _VectorSearchMixin:
    search_mode: str
    retrieval_unit: str | None
    _service: Mi355RetrievalService
    _embedding_model: Any

    async def _retrieve_by_id(self, query_id, top_k: int) -> list[dict[str, Any]]:
        results = self._service.vector_search([query_id], top_k, search_mode=self.search_mode,
                                              unit=self.retrieval_unit or "chunk")
        return results[0] if results else []

    def _retrieve_block(self, query_ids: list, top_k: int) -> list[list[dict] | None]:
        """One GPU block for the whole page; a query that cannot be scored (missing row / embedding) fails alone."""
        unit = self.retrieval_unit or "chunk"
        ok, bad = [], set()
        for qid, q in zip(query_ids, self._service.get_queries(list(query_ids)), strict=True):
            has = q is not None and ((q.embeddings is not None) if self.search_mode == "multi" else (q.embedding is not None))
            if has:
                ok.append(qid)
            else:
                bad.add(qid)
        res = self._service.vector_search(ok, top_k, search_mode=self.search_mode, unit=unit) if ok else []
        by_id = dict(zip(ok, res))
        for qid in bad:
            logger.error(f"Retrieval failed for query {qid}")
        return [None if qid in bad else by_id[qid] for qid in query_ids]


@dataclass(kw_only=True)
class Mi355VectorSearchPipelineConfig(BaseRetrievalPipelineConfig):
    """Config of the MI355X vector search pipeline (fields as VectorSearchPipelineConfig, vector_search.py:19-71)."""

    search_mode: Literal["single", "multi"] = field(default="single")
    embedding_model: Any | str | None = field(default=None)
    device: int = 0

    def get_pipeline_class(self) -> type["Mi355VectorSearchRetrievalPipeline"]:
        return Mi355VectorSearchRetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"search_mode": self.search_mode, "embedding_model": self.embedding_model, "device": self.device}

    def __setattr__(self, name: str, value: Any) -> None:
        # a config-name string is resolved to an instance and KEPT (the image variant's behaviour,
        # image_vector_search.py:40-45; the text variant of the reference loads and discards it)
        if name == "embedding_model" and isinstance(value, str):
            from .embeddings import load_embedding_model

            value = load_embedding_model(value)
        super().__setattr__(name, value)


class Mi355VectorSearchRetrievalPipeline(_VectorSearchMixin, Mi355BaseRetrievalPipeline):
    """Vector search over `chunk` rows on the GPU (reference VectorSearchRetrievalPipeline, vector_search.py:74-191)."""

    retrieval_unit = "chunk"

    def __init__(self, session_factory: Any, name: str, search_mode: Literal["single", "multi"] = "single",
                 embedding_model: Any | None = None, schema: Any | None = None, device: int = 0):
        # set BEFORE super().__init__: _get_pipeline_config() is called there (vector_search.py:138-143)
        self.search_mode = search_mode
        self._embedding_model = embedding_model
        super().__init__(session_factory, name, schema, device=device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        return {"type": "mi355_vector_search", "retrieval_unit": self.retrieval_unit, "search_mode": self.search_mode}

    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]:
        if self._embedding_model is None:
            raise EmbeddingError
        # (one process per GPU: rank 0 embeds, every rank searches the same vector)
        query_embedding = await self._service.on_root(lambda: self._embedding_model.aembed_query(query_text))
        return self._service.vector_search_by_embedding(query_embedding, top_k)


@dataclass(kw_only=True)
class Mi355ImageVectorSearchPipelineConfig(BaseRetrievalPipelineConfig):
    search_mode: Literal["single", "multi"] = field(default="multi")
    embedding_model: Any | str | None = field(default=None)
    device: int = 0

    def get_pipeline_class(self) -> type["Mi355ImageVectorSearchRetrievalPipeline"]:
        return Mi355ImageVectorSearchRetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"search_mode": self.search_mode, "embedding_model": self.embedding_model, "device": self.device}

    def __setattr__(self, name: str, value: Any) -> None:
        if name == "embedding_model" and isinstance(value, str):
            from .embeddings import load_embedding_model

            value = load_embedding_model(value)
        super().__setattr__(name, value)


class Mi355ImageVectorSearchRetrievalPipeline(_VectorSearchMixin, Mi355BaseRetrievalPipeline):
    """Vector search over `image_chunk` rows (reference image_vector_search.py:48-139); content is None."""

    retrieval_unit = "image_chunk"

    def __init__(self, session_factory: Any, name: str, search_mode: Literal["single", "multi"] = "multi",
                 embedding_model: Any | None = None, schema: Any | None = None, device: int = 0):
        self.search_mode = search_mode
        self._embedding_model = embedding_model
        super().__init__(session_factory, name, schema, device=device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        return {"type": "mi355_image_vector_search", "retrieval_unit": self.retrieval_unit, "search_mode": self.search_mode}

    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]:
        if self._embedding_model is None:
            raise EmbeddingError
        if self.search_mode == "multi":
            query_vectors = await self._service.on_root(lambda: self._embedding_model.aembed_query(query_text))
            return self._service.maxsim_search_by_embeddings([query_vectors], top_k, unit="image_chunk")[0]
        query_vector = await self._service.on_root(lambda: self._embedding_model.aembed_query(query_text))
        return self._service.vector_search_by_embedding(query_vector, top_k, unit="image_chunk")


__all__ = ["Mi355VectorSearchPipelineConfig", "Mi355VectorSearchRetrievalPipeline",
           "Mi355ImageVectorSearchPipelineConfig", "Mi355ImageVectorSearchRetrievalPipeline",
           "Mi355BaseRetrievalPipeline", "get_retrieval_pipeline_unit"]
