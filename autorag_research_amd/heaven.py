"""HEAVEN two-stage retrieval on the MI355X path.

Mirrors the reference's HEAVENPipelineConfig / HEAVENRetrievalPipeline (pipelines/retrieval/heaven.py:103-340):
stage 1 = single-vector cosine top-`stage1_candidate_count` over `image_chunk` rows, stage 2 = late-interaction
re-scoring of those candidates with a budget of "key" query vectors, then of the best quarter with the rest, and a
convex combination of the three scores.  What changes is where the arithmetic runs:

  reference                                                          here
  uow.image_chunks.vector_search_with_scores (SQL `<=>`, :208-222)   Mi355RetrievalService.vector_search_by_embedding
  _fetch_candidate_multi_embeddings + _score_candidates              Mi355RetrievalService.maxsim_score_candidates
  (rows pulled into Python lists, triple Python loop, :224-266)      -> mi355dr_maxsim_subset (exact fp32 MFMA kernel)

Scores equal the reference's to fp32 rounding (its loops run in float64 on the same fp32 values): <= 1e-6 on the
golden inputs (tests/golden/service_golden.json["heaven"]).

The key-vector budget comes from the noun share of the query text (heaven.py:31-56).  The reference tags with nltk; here
the tagger is injectable (`pos_tagger=`, a callable tokens -> [(token, tag)]), defaults to nltk.pos_tag when nltk is
importable and to the configured `default_key_token_ratio` otherwise.
"""

from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Any, Callable

from .compat import BaseRetrievalPipelineConfig, EmbeddingError
from .pipelines import Mi355BaseRetrievalPipeline

_WORD = re.compile(r"[A-Za-z0-9']+")
PosTagger = Callable[[list[str]], list[tuple[str, str]]]


def default_pos_tagger() -> PosTagger | None:
    try:
        import nltk  # noqa: PLC0415

        return nltk.pos_tag
    except Exception:  # noqa: BLE001 - nltk missing or its model not downloaded: fall back to the configured ratio
        return None


def key_vector_budget(query_text: str, n_query_vectors: int, default_keep_ratio: float, pos_tagger: PosTagger | None) -> int:
    """How many leading query vectors count as "key" vectors: ceil(n * noun share), at least 1 (heaven.py:31-56)."""
    if n_query_vectors <= 0:
        return 0
    words = _WORD.findall(query_text.lower())
    share = default_keep_ratio
    if words and pos_tagger is not None:
        try:
            nouns = sum(1 for _, tag in pos_tagger(words) if tag.startswith("NN"))
        except LookupError:
            nouns = 0
        if nouns > 0:
            share = nouns / len(words)
    return max(1, min(n_query_vectors, math.ceil(n_query_vectors * share)))


def rank_by_key_score(stage1: list[dict[str, Any]], key_scores: dict, how_many: int) -> list:
    """Candidate ids by descending key score, stage-1 order breaking ties; the first `how_many` (heaven.py:69-83)."""
    if how_many <= 0:
        return []
    first_seen = {r["doc_id"]: i for i, r in enumerate(stage1)}
    order = sorted(key_scores, key=lambda pk: (-key_scores[pk], first_seen.get(pk, math.inf)))
    return order[:how_many]


def blend_scores(stage1: list[dict[str, Any]], key_scores: dict, rest_scores: dict, refined: set, stage1_weight: float,
                 top_k: int) -> list[dict[str, Any]]:
    """final = w * stage1 + (1 - w) * (key + rest-if-refined), sorted descending (stable), top_k (heaven.py:86-108)."""
    blended = []
    for r in stage1:
        pk = r["doc_id"]
        late = key_scores.get(pk, 0.0) + (rest_scores.get(pk, 0.0) if pk in refined else 0.0)
        blended.append({"doc_id": pk, "score": stage1_weight * float(r["score"]) + (1 - stage1_weight) * late})
    blended.sort(key=lambda item: item["score"], reverse=True)
    return blended[:top_k]


@dataclass(kw_only=True)
class Mi355HEAVENPipelineConfig(BaseRetrievalPipelineConfig):
    """Fields as HEAVENPipelineConfig (heaven.py:103-139) + `device`."""

    stage1_candidate_count: int = 200
    stage2_refine_ratio: float = 0.25
    stage1_weight: float = 0.3
    default_key_token_ratio: float = 0.5
    single_vector_embedding_model: Any | str | None = field(default=None)
    multi_vector_embedding_model: Any | str | None = field(default=None)
    device: int = 0

    def get_pipeline_class(self) -> type["Mi355HEAVENRetrievalPipeline"]:
        return Mi355HEAVENRetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"stage1_candidate_count": self.stage1_candidate_count, "stage2_refine_ratio": self.stage2_refine_ratio,
                "stage1_weight": self.stage1_weight, "default_key_token_ratio": self.default_key_token_ratio,
                "single_vector_embedding_model": self.single_vector_embedding_model,
                "multi_vector_embedding_model": self.multi_vector_embedding_model, "device": self.device}

    def __setattr__(self, name: str, value: Any) -> None:
        if name in {"single_vector_embedding_model", "multi_vector_embedding_model"} and isinstance(value, str):
            from .embeddings import load_embedding_model

            value = load_embedding_model(value)
        super().__setattr__(name, value)


class Mi355HEAVENRetrievalPipeline(Mi355BaseRetrievalPipeline):
    """Two-stage HEAVEN retrieval over `image_chunk` rows, both stages on the GPU."""

    retrieval_unit = "image_chunk"

    def __init__(self, session_factory: Any, name: str, stage1_candidate_count: int = 200, stage2_refine_ratio: float = 0.25,
                 stage1_weight: float = 0.3, default_key_token_ratio: float = 0.5,
                 single_vector_embedding_model: Any | None = None, multi_vector_embedding_model: Any | None = None,
                 schema: Any | None = None, device: int = 0, pos_tagger: PosTagger | None | str = "auto"):
        if stage1_candidate_count <= 0:
            raise ValueError("stage1_candidate_count must be positive")
        if not 0 < stage2_refine_ratio <= 1:
            raise ValueError("stage2_refine_ratio must be in (0, 1]")
        if not 0 <= stage1_weight <= 1:
            raise ValueError("stage1_weight must be in [0, 1]")
        if not 0 < default_key_token_ratio <= 1:
            raise ValueError("default_key_token_ratio must be in (0, 1]")
        self.stage1_candidate_count = stage1_candidate_count
        self.stage2_refine_ratio = stage2_refine_ratio
        self.stage1_weight = stage1_weight
        self.default_key_token_ratio = default_key_token_ratio
        self._single_vector_embedding_model = single_vector_embedding_model
        self._multi_vector_embedding_model = multi_vector_embedding_model
        self._pos_tagger = default_pos_tagger() if pos_tagger == "auto" else pos_tagger
        super().__init__(session_factory, name, schema, device=device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        return {"type": "mi355_heaven", "retrieval_unit": self.retrieval_unit,
                "stage1_candidate_count": self.stage1_candidate_count, "stage2_refine_ratio": self.stage2_refine_ratio,
                "stage1_weight": self.stage1_weight, "default_key_token_ratio": self.default_key_token_ratio}

    def _stored_query(self, query_id) -> tuple[str, Any, Any]:
        q = self._service.get_queries([query_id])[0]
        if q is None:
            raise ValueError(f"Query {query_id} not found")  # noqa: TRY003
        if q.embedding is None:
            raise ValueError(f"Query {query_id} has no single-vector embedding")  # noqa: TRY003
        if q.embeddings is None:
            raise ValueError(f"Query {query_id} has no multi-vector embeddings")  # noqa: TRY003
        return q.contents or "", q.embedding, q.embeddings

    def _search(self, query_text: str, single_vec, multi_vecs, top_k: int) -> list[dict[str, Any]]:
        limit = max(top_k, self.stage1_candidate_count)
        stage1 = self._service.vector_search_by_embedding(list(single_vec), limit, unit="image_chunk")
        if not stage1:
            return []
        n_vec = len(multi_vecs)
        if n_vec == 0:
            return stage1[:top_k]
        ids = [r["doc_id"] for r in stage1]
        n_key = key_vector_budget(query_text, n_vec, self.default_key_token_ratio, self._pos_tagger)
        n_key = max(0, min(n_vec, n_key))
        key_scores = self._service.maxsim_score_candidates(multi_vecs[:n_key], ids, unit="image_chunk")
        if not key_scores:  # no candidate has multi-vector embeddings (heaven.py:282-283)
            return stage1[:top_k]
        n_refine = min(len(ids), max(top_k, math.ceil(len(ids) * self.stage2_refine_ratio)))
        refined = rank_by_key_score(stage1, key_scores, n_refine)
        rest = multi_vecs[n_key:]
        rest_scores = self._service.maxsim_score_candidates(rest, refined, unit="image_chunk") if len(rest) else {}
        return blend_scores(stage1, key_scores, rest_scores, set(refined), self.stage1_weight, top_k)

    async def _retrieve_by_id(self, query_id, top_k: int) -> list[dict[str, Any]]:
        text, single_vec, multi_vecs = self._stored_query(query_id)
        return self._search(text, single_vec, multi_vecs, top_k)

    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]:
        if self._single_vector_embedding_model is None or self._multi_vector_embedding_model is None:
            raise EmbeddingError
        async def embed_both():
            return (await self._single_vector_embedding_model.aembed_query(query_text),
                    await self._multi_vector_embedding_model.aembed_query(query_text))

        single_vec, multi_vecs = await self._service.on_root(embed_both)   # one process per GPU: rank 0 embeds, all ranks search
        return self._search(query_text, single_vec, multi_vecs, top_k)

    def run(self, top_k: int = 10, batch_size: int = 128, max_concurrency: int = 16, max_retries: int = 3,
            retry_delay: float = 1.0, query_limit: int | None = None, block: bool = False) -> dict[str, Any]:
        """Per-query driver (the two stages depend on each other per query): the reference's run_image_pipeline."""
        return super().run(top_k=top_k, batch_size=batch_size, max_concurrency=max_concurrency, max_retries=max_retries,
                           retry_delay=retry_delay, query_limit=query_limit, block=False)


__all__ = ["Mi355HEAVENPipelineConfig", "Mi355HEAVENRetrievalPipeline", "key_vector_budget", "rank_by_key_score",
           "blend_scores"]
