"""HyDE retrieval (hypothetical-document embeddings) on the MI355X path.

Mirrors the reference's HyDEPipelineConfig / HyDERetrievalPipeline (pipelines/retrieval/hyde.py:31-241): an LLM writes a
passage that would answer the query, the passage -- not the query -- is embedded, and the embedding is searched
(`vector_search_by_embedding`, :232-238).  The LLM and the embedding model are the caller's (anything with `ainvoke` /
`aembed_query`, i.e. LangChain objects, or plain callables); the search is `mi355dr_search`.  What this module adds over
the reference's per-query flow is the page form used by `run()`: all passages of a page are generated concurrently, embedded (through the
model's query side, batched when it offers `embed_queries`), and searched as ONE block on the GPU.
"""

from __future__ import annotations

import asyncio
import logging
import inspect
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from .compat import BaseRetrievalPipelineConfig
from .pipelines import Mi355BaseRetrievalPipeline

logger = logging.getLogger("AutoRAG-Research")

DEFAULT_HYDE_PROMPT_TEMPLATE = """Please write a passage to answer the question.
Question: {query}
Passage:"""  # the paper's general template (Gao et al. 2022), as the reference uses (hyde.py:25-27)


@dataclass(kw_only=True)
class Mi355HyDEPipelineConfig(BaseRetrievalPipelineConfig):
    """Fields as HyDEPipelineConfig (hyde.py:31-91) + `device`.  String values name configs of the host framework: the
    embedding is resolved like everywhere in this package, an LLM name needs the reference's `load_llm`."""

    llm: Any
    embedding: Any
    prompt_template: str = field(default=DEFAULT_HYDE_PROMPT_TEMPLATE)
    device: int = 0

    def __setattr__(self, name: str, value: Any) -> None:
        if name == "embedding" and isinstance(value, str):
            from .embeddings import load_embedding_model

            value = load_embedding_model(value)
        elif name == "llm" and isinstance(value, str):
            try:
                from autorag_research.injection import load_llm  # type: ignore
            except Exception as e:  # noqa: BLE001
                raise ValueError(f"llm {value!r} was given by name, which needs autorag_research.injection.load_llm; "  # noqa: TRY003
                                 "pass a model instance instead") from e
            value = load_llm(value)
        super().__setattr__(name, value)

    def get_pipeline_class(self) -> type["Mi355HyDERetrievalPipeline"]:
        return Mi355HyDERetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"llm": self.llm, "embedding": self.embedding, "prompt_template": self.prompt_template, "device": self.device}


class Mi355HyDERetrievalPipeline(Mi355BaseRetrievalPipeline):
    """LLM passage -> embedding -> exact cosine top-k over `chunk` rows on the GPU."""

    retrieval_unit = "chunk"

    def __init__(self, session_factory: Any, name: str, llm: Any, embedding: Any,
                 prompt_template: str = DEFAULT_HYDE_PROMPT_TEMPLATE, schema: Any | None = None, device: int = 0):
        self.llm = llm
        self.embedding = embedding
        if "{query}" not in prompt_template:
            raise ValueError("prompt_template must contain '{query}' placeholder")
        self.prompt_template = prompt_template
        super().__init__(session_factory, name, schema, device=device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        return {"type": "mi355_hyde", "retrieval_unit": self.retrieval_unit, "prompt_template": self.prompt_template}

    @staticmethod
    def _extract_response_content(response: Any) -> str:
        """Chat models answer with a message object, completion models with a string (hyde.py:190-203)."""
        return str(response.content) if hasattr(response, "content") else str(response)

    async def _generate_hypothetical_document(self, query_text: str) -> str:
        prompt = self.prompt_template.format(query=query_text)
        if hasattr(self.llm, "ainvoke"):
            response = await self.llm.ainvoke(prompt)
        else:
            response = self.llm(prompt)
            if inspect.isawaitable(response):
                response = await response
        return self._extract_response_content(response)

    def _query_text(self, query_id) -> str:
        q = self._service.get_queries([query_id])[0]
        if q is None:
            raise ValueError(f"Query {query_id} not found")  # noqa: TRY003  (fetch_query_texts, retrieval_pipeline.py:552-571)
        return q.contents

    async def _retrieve_by_id(self, query_id, top_k: int) -> list[dict[str, Any]]:
        return await self._retrieve_by_text(self._query_text(query_id), top_k)

    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]:
        async def passage_embedding():
            passage = await self._generate_hypothetical_document(query_text)
            return await self.embedding.aembed_query(passage)

        # One process per GPU: the LLM and the embedding model answer on rank 0 alone and the vector (or the exception)
        # reaches every rank -- an LLM samples, so two ranks would otherwise search DIFFERENT vectors, and a rank-local
        # failure would send one rank down the retry path while the others sit in the search's all-gather.
        embedding = await self._service.on_root(passage_embedding)
        return self._service.vector_search_by_embedding(embedding=embedding, top_k=top_k)

    def _retrieve_block(self, query_ids: list, top_k: int) -> list[list[dict] | None]:
        """A page: passages generated concurrently, embedded, searched as one GPU block."""
        # ONE read of the page's query rows, before anything asynchronous: under a process group `get_queries` is a
        # collective (rank 0 reads, everybody receives), and collectives issued from concurrently scheduled coroutines --
        # let alone from inside a per-passage retry loop -- would differ in number and order between ranks.
        rows = self._service.get_queries(list(query_ids))

        def generate_and_embed():
            async def passages():
                attempts, delay0 = getattr(self, "_run_retry", (1, 0.0))

                async def one(qid, row):
                    # the reference retries a failing query with exponential backoff before it gives it up
                    # (retrieval_pipeline.py:222-236); so does the block form, per passage
                    delay = delay0
                    for attempt in range(attempts):
                        try:
                            if row is None:
                                raise ValueError(f"Query {qid} not found")  # noqa: TRY003, TRY301
                            return await self._generate_hypothetical_document(row.contents)
                        except Exception:  # noqa: BLE001
                            if attempt + 1 >= attempts:
                                logger.exception(f"HyDE passage generation failed for query {qid} after {attempts} attempts")
                                return None  # that query is reported as failed
                            await asyncio.sleep(min(max(delay, delay0), 60))
                            delay *= 2
                    return None

                return await asyncio.gather(*[one(q, r) for q, r in zip(query_ids, rows)])

            docs = asyncio.run(passages())
            live = [i for i, p in enumerate(docs) if p is not None]
            if not live:
                return live, None
            texts = [docs[i] for i in live]
            # the passages go through `aembed_query` like in the reference (hyde.py:229-230) -- NOT `embed_documents`:
            # asymmetric models encode queries and documents differently, and the block form must return what the
            # per-query form returns.  A model may offer `embed_queries` (a batch of query-side embeddings) to batch this.
            if hasattr(self.embedding, "embed_queries"):
                vecs = self.embedding.embed_queries(texts)
            else:
                async def embed_all():
                    return await asyncio.gather(*[self.embedding.aembed_query(t) for t in texts])

                vecs = asyncio.run(embed_all())
            return live, np.asarray(vecs, dtype=np.float32)

        # generation + embedding: rank-local work that samples and may fail -> rank 0 alone, one broadcast of (live, vectors);
        # the retries above issue no collective
        world = getattr(self._service, "_world", None)
        live, vecs = world.from_root(generate_and_embed) if world is not None else generate_and_embed()
        out: list[list[dict] | None] = [None] * len(query_ids)
        if not live:
            return out
        block = self._service._single_block(vecs, top_k, "chunk")
        for i, res in zip(live, block):
            out[i] = res
        return out


__all__ = ["DEFAULT_HYDE_PROMPT_TEMPLATE", "Mi355HyDEPipelineConfig", "Mi355HyDERetrievalPipeline"]
