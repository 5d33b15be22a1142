"""Guided Query Refinement (GQR) hybrid retrieval on the MI355X path.

Mirrors the reference's GQRHybridRetrievalPipelineConfig / GQRHybridRetrievalPipeline
(pipelines/retrieval/gqr_hybrid.py:130-523): a primary and a complementary child pipeline each return
`top_k * fetch_k_multiplier` results, their union (or the primary list) is the candidate pool, the primary query
representation is refined for `n_steps` against a consensus of both score distributions, and the pool is re-ranked
by the refined scores.  What changes is where the arithmetic runs:

  reference                                                            here
  get_chunk_embeddings / get_chunk_multi_embeddings: candidate         the candidates are named by row id; their
  vectors pulled out of PostgreSQL into float lists (:284-304)         vectors stay in HBM
  _optimize_query_embedding: numpy loop, one query at a time           mi355dr_gqr_refine: one workgroup per query,
  (:321-340)                                                           a page of queries per launch (float64)
  _optimize_query_multi_embedding (:342-362)                           mi355dr_gqr_refine_maxsim
  _optimize_in_score_space (:306-319)                                  mi355dr_gqr_refine_scores

The host keeps the bookkeeping: pool construction (:258-276), the missing-score floor (:54-62), the complementary
distribution (one softmax of P numbers per query, :436) and the final stable sort (:278-282).  Scores agree with the
reference's float64 numpy results to ~1e-12 (tests/golden/gqr_golden.*).

One deviation, forced by the store: a chunk whose `embeddings` column holds an EMPTY array is indistinguishable from
NULL here, so a pool containing one takes the score-space branch where the reference would score it 0.
"""

from __future__ import annotations

import asyncio
import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Literal

import numpy as np

from .compat import BaseRetrievalPipelineConfig, EmbeddingError
from .pipelines import Mi355BaseRetrievalPipeline, child_page

CandidatePoolMode = Literal["primary", "union"]
ScorerMode = Literal["auto", "single", "multi"]
_EPSILON = 1e-8
logger = logging.getLogger("AutoRAG-Research")


def _check_params(fetch_k_multiplier, n_steps, learning_rate, temperature, mixture_alpha, candidate_pool_mode,
                  scorer_mode) -> None:
    """The reference's argument checks, same messages (gqr_hybrid.py:143-163, :202-222)."""
    if fetch_k_multiplier <= 0:
        raise ValueError("fetch_k_multiplier must be positive")
    if n_steps <= 0:
        raise ValueError("n_steps must be positive")
    if learning_rate <= 0:
        raise ValueError("learning_rate must be positive")
    if temperature <= 0:
        raise ValueError("temperature must be positive")
    if not 0 <= mixture_alpha <= 1:
        raise ValueError("mixture_alpha must be between 0 and 1")
    if candidate_pool_mode not in {"primary", "union"}:
        raise ValueError("candidate_pool_mode must be either 'primary' or 'union'")
    if scorer_mode not in {"auto", "single", "multi"}:
        raise ValueError("scorer_mode must be one of 'auto', 'single', or 'multi'")


def score_distribution(scores: np.ndarray, temperature: float) -> np.ndarray:
    """Softmax of scores / T with the reference's guards: uniform if the normaliser is not finite or ~0 (:39-51)."""
    if scores.size == 0:
        return scores
    z = scores / max(temperature, _EPSILON)
    e = np.exp(z - np.max(z))
    total = float(np.sum(e))
    if not np.isfinite(total) or total <= _EPSILON:
        return np.full(scores.shape, 1.0 / scores.size, dtype=np.float64)
    return e / total


def missing_score_floor(score_map: dict) -> float:
    """What a retriever "would have scored" a pool member it did not return: min - max(1, max - min)  (:54-62)."""
    if not score_map:
        return -1.0
    lo, hi = min(score_map.values()), max(score_map.values())
    return lo - max(1.0, hi - lo)


def candidate_pool(primary: list[dict], complementary: list[dict], mode: str) -> list:
    """Primary ids, or primary then complementary ids in first-seen order (:258-276)."""
    if mode == "primary":
        return [r["doc_id"] for r in primary]
    seen: set = set()
    out = []
    for r in [*primary, *complementary]:
        if r["doc_id"] not in seen:
            seen.add(r["doc_id"])
            out.append(r["doc_id"])
    return out


def score_vector(ids: list, results: list[dict]) -> np.ndarray:
    """Raw retriever scores aligned with the pool; absent members get the floor (:252-256, :268-272)."""
    score_map = {r["doc_id"]: float(r["score"]) for r in results}
    floor = missing_score_floor(score_map)
    return np.asarray([score_map.get(pk, floor) for pk in ids], dtype=np.float64)


@dataclass(kw_only=True)
class Mi355GQRHybridPipelineConfig(BaseRetrievalPipelineConfig):
    """Fields as GQRHybridRetrievalPipelineConfig (gqr_hybrid.py:130-183) + `device`."""

    primary_retrieval_pipeline_name: str
    complementary_retrieval_pipeline_name: str
    fetch_k_multiplier: int = 2
    n_steps: int = 25
    learning_rate: float = 0.1
    temperature: float = 1.0
    mixture_alpha: float = 0.5
    candidate_pool_mode: CandidatePoolMode = "union"
    scorer_mode: ScorerMode = "auto"
    device: int = 0

    def __post_init__(self) -> None:
        _check_params(self.fetch_k_multiplier, self.n_steps, self.learning_rate, self.temperature, self.mixture_alpha,
                      self.candidate_pool_mode, self.scorer_mode)

    def get_pipeline_class(self) -> type["Mi355GQRHybridRetrievalPipeline"]:
        return Mi355GQRHybridRetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"primary_retrieval_pipeline": self.primary_retrieval_pipeline_name,
                "complementary_retrieval_pipeline": self.complementary_retrieval_pipeline_name,
                "fetch_k_multiplier": self.fetch_k_multiplier, "n_steps": self.n_steps,
                "learning_rate": self.learning_rate, "temperature": self.temperature,
                "mixture_alpha": self.mixture_alpha, "candidate_pool_mode": self.candidate_pool_mode,
                "scorer_mode": self.scorer_mode, "device": self.device}


def _load_child(name: str, session_factory: Any, schema: Any, config_dir: Path | None):
    """A child named by its YAML: resolved by the reference's own loader (hybrid.py:335-364) where it is installed."""
    try:
        from autorag_research.pipelines.retrieval.hybrid import HybridRetrievalPipeline  # type: ignore
    except Exception as e:  # noqa: BLE001
        raise ValueError(  # noqa: TRY003
            f"child pipeline {name!r} was given by name, which needs the autorag_research config loader; "
            "pass a pipeline instance instead") from e
    return HybridRetrievalPipeline._load_pipeline(name, session_factory, schema, config_dir)


class Mi355GQRHybridRetrievalPipeline(Mi355BaseRetrievalPipeline):
    """GQR over two child retrieval pipelines; the refinement loops run on the GPU, a page of queries per launch."""

    def __init__(self, session_factory: Any, name: str, primary_retrieval_pipeline: Any,
                 complementary_retrieval_pipeline: Any, fetch_k_multiplier: int = 2, n_steps: int = 25,
                 learning_rate: float = 0.1, temperature: float = 1.0, mixture_alpha: float = 0.5,
                 candidate_pool_mode: CandidatePoolMode = "union", scorer_mode: ScorerMode = "auto",
                 schema: Any | None = None, config_dir: Path | None = None, device: int = 0):
        _check_params(fetch_k_multiplier, n_steps, learning_rate, temperature, mixture_alpha, candidate_pool_mode,
                      scorer_mode)
        if isinstance(primary_retrieval_pipeline, str):
            primary_retrieval_pipeline = _load_child(primary_retrieval_pipeline, session_factory, schema, config_dir)
        if isinstance(complementary_retrieval_pipeline, str):
            complementary_retrieval_pipeline = _load_child(complementary_retrieval_pipeline, session_factory, schema,
                                                           config_dir)
        self._primary_retrieval_pipeline = primary_retrieval_pipeline
        self._complementary_retrieval_pipeline = complementary_retrieval_pipeline
        self.fetch_k_multiplier = fetch_k_multiplier
        self.n_steps = n_steps
        self.learning_rate = learning_rate
        self.temperature = temperature
        self.mixture_alpha = mixture_alpha
        self.candidate_pool_mode = candidate_pool_mode
        self.scorer_mode = scorer_mode
        super().__init__(session_factory, name, schema, device=device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        return {"type": "mi355_gqr_hybrid",
                "primary_retrieval_pipeline": self._primary_retrieval_pipeline.name,
                "complementary_retrieval_pipeline": self._complementary_retrieval_pipeline.name,
                "fetch_k_multiplier": self.fetch_k_multiplier, "n_steps": self.n_steps,
                "learning_rate": self.learning_rate, "temperature": self.temperature,
                "mixture_alpha": self.mixture_alpha, "candidate_pool_mode": self.candidate_pool_mode,
                "scorer_mode": self.scorer_mode}

    def _params(self) -> dict[str, Any]:
        return {"n_steps": self.n_steps, "learning_rate": self.learning_rate, "temperature": self.temperature,
                "mixture_alpha": self.mixture_alpha}

    def _resolve_scorer_mode(self) -> str:
        if self.scorer_mode != "auto":
            return self.scorer_mode
        return "multi" if getattr(self._primary_retrieval_pipeline, "search_mode", "single") == "multi" else "single"

    # ---- the refinement of a page of queries (reference _run_gqr, :415-470, one query at a time) ----
    def _run_gqr_block(self, items: list[dict[str, Any]]) -> list[list[dict[str, Any]]]:
        """items: {"top_k", "query_embedding", "query_multi_embedding", "primary", "complementary"} per query."""
        mode = self._resolve_scorer_mode()
        svc = self._service
        out: list[list[dict[str, Any]]] = [[] for _ in items]
        groups: dict[str, list[dict[str, Any]]] = {"single": [], "multi": [], "scores": []}
        for slot, it in enumerate(items):
            ids = candidate_pool(it["primary"], it["complementary"], self.candidate_pool_mode)
            if not ids:
                continue
            comp = score_distribution(score_vector(ids, it["complementary"]), self.temperature)
            job = {"slot": slot, "ids": ids, "comp": comp, "top_k": it["top_k"]}
            rows = None
            if mode == "multi":
                qm = it["query_multi_embedding"]
                rows = svc.chunk_rows_multi(ids) if qm is not None else None
                if rows is not None:
                    if qm.size == 0:  # no query vectors: every late-interaction score is 0 (:97-98)
                        out[slot] = self._rank(ids, np.zeros(len(ids)), it["top_k"])
                        continue
                    job.update(rows=rows, q=np.asarray(qm, dtype=np.float64).reshape(qm.shape[0], -1))
                    groups["multi"].append(job)
                    continue
            else:
                qv = it["query_embedding"]
                rows = svc.chunk_rows_single(ids) if qv is not None else None
                if rows is not None:
                    job.update(rows=rows, q=np.asarray(qv, dtype=np.float64).reshape(-1))
                    groups["single"].append(job)
                    continue
            job["primary"] = score_vector(ids, it["primary"])  # vectors missing: refine the scores themselves
            groups["scores"].append(job)
        for kind, jobs in groups.items():
            if not jobs:
                continue
            P = max(len(j["ids"]) for j in jobs)
            comp = np.zeros((len(jobs), P))
            pools = np.full((len(jobs), P), -1, dtype=np.int64)
            for b, j in enumerate(jobs):
                comp[b, :len(j["ids"])] = j["comp"]
                if kind != "scores":
                    pools[b, :len(j["ids"])] = j["rows"]
            if kind == "single":
                scores = svc.gqr_refine_single(np.stack([j["q"] for j in jobs]), pools, comp, **self._params())
            elif kind == "multi":
                q_off = np.concatenate([[0], np.cumsum([j["q"].shape[0] for j in jobs])]).astype(np.int32)
                scores = svc.gqr_refine_multi(np.concatenate([j["q"] for j in jobs], axis=0), q_off, pools, comp,
                                              **self._params())
            else:
                prim = np.zeros((len(jobs), P))
                for b, j in enumerate(jobs):
                    prim[b, :len(j["ids"])] = j["primary"]
                counts = np.asarray([len(j["ids"]) for j in jobs], dtype=np.int32)
                scores = svc.gqr_refine_scores(prim, counts, comp, **self._params())
            for b, j in enumerate(jobs):
                out[j["slot"]] = self._rank(j["ids"], scores[b, :len(j["ids"])], j["top_k"])
        return out

    @staticmethod
    def _rank(ids: list, scores: np.ndarray, top_k: int) -> list[dict[str, Any]]:
        """Descending by score, pool order between equals (a stable sort of the score map, :278-282)."""
        order = sorted(range(len(ids)), key=lambda i: float(scores[i]), reverse=True)
        return [{"doc_id": ids[i], "score": float(scores[i])} for i in order[:top_k]]

    async def _children_by_id(self, query_id, fetch_k: int) -> tuple[list[dict], list[dict]]:
        primary = await self._primary_retrieval_pipeline._retrieve_by_id(query_id, fetch_k)
        complementary = await self._complementary_retrieval_pipeline._retrieve_by_id(query_id, fetch_k)
        return primary, complementary

    def _item_by_id(self, query_id, top_k: int, primary: list[dict], complementary: list[dict]) -> dict[str, Any]:
        return {"top_k": top_k, "primary": primary, "complementary": complementary,
                "query_embedding": self._service.get_query_embedding(query_id),
                "query_multi_embedding": (self._service.get_query_multi_embedding(query_id)
                                          if self._resolve_scorer_mode() == "multi" else None)}

    async def _retrieve_by_id(self, query_id, top_k: int) -> list[dict[str, Any]]:
        primary, complementary = await self._children_by_id(query_id, top_k * self.fetch_k_multiplier)
        return self._run_gqr_block([self._item_by_id(query_id, top_k, primary, complementary)])[0]

    def _retrieve_block(self, query_ids: list, top_k: int) -> list[list[dict] | None]:
        """A page of queries: each child answers the page (one GPU block when it can), the refinement is ONE launch per
        branch for the whole page."""
        fetch_k = top_k * self.fetch_k_multiplier

        page_p = child_page(self._primary_retrieval_pipeline, query_ids, fetch_k)
        page_c = child_page(self._complementary_retrieval_pipeline, query_ids, fetch_k)
        pairs = [None if a is None or b is None else (a, b) for a, b in zip(page_p, page_c)]
        live = [(i, qid, pc) for i, (qid, pc) in enumerate(zip(query_ids, pairs)) if pc is not None]
        ranked = self._run_gqr_block([self._item_by_id(qid, top_k, pc[0], pc[1]) for _, qid, pc in live])
        out: list[list[dict] | None] = [None] * len(query_ids)
        for (i, _, _), r in zip(live, ranked):
            out[i] = r
        return out

    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]:
        """Ad-hoc text: a child without an embedding model is replaced by the other child's list (:491-505); the
        multi-vector scorer has no by-text embedding interface and takes the score-space branch (:509-513)."""
        fetch_k = top_k * self.fetch_k_multiplier
        try:
            primary = await self._primary_retrieval_pipeline._retrieve_by_text(query_text, fetch_k)
        except EmbeddingError:
            complementary = await self._complementary_retrieval_pipeline._retrieve_by_text(query_text, fetch_k)
            primary = complementary
        else:
            try:
                complementary = await self._complementary_retrieval_pipeline._retrieve_by_text(query_text, fetch_k)
            except EmbeddingError:
                complementary = primary
        model = getattr(self._primary_retrieval_pipeline, "_embedding_model", None)
        qv = None if model is None else np.asarray(await model.aembed_query(query_text), dtype=np.float64)
        if self._resolve_scorer_mode() == "multi":
            logger.info("GQR multi-vector by-text retrieval has no standard embedding interface; using score-space fallback")
        return self._run_gqr_block([{"top_k": top_k, "primary": primary, "complementary": complementary,
                                     "query_embedding": qv, "query_multi_embedding": None}])[0]


__all__ = ["Mi355GQRHybridPipelineConfig", "Mi355GQRHybridRetrievalPipeline", "candidate_pool", "missing_score_floor",
           "score_distribution", "score_vector"]
