"""Hybrid retrieval (score fusion of two child pipelines) on the MI355X path.

Mirrors the reference's HybridRRFRetrievalPipeline / HybridCCRetrievalPipeline and their configs
(pipelines/retrieval/hybrid.py:181-641): both children are asked for `top_k * fetch_k_multiplier` results per query and
the two ranked lists are fused -- Reciprocal Rank Fusion (:46-97) or a convex combination of normalised scores (:100-178,
normalisers util.py:371-530).  The fusion itself is a few dozen numbers per query and stays on the host; what this
module adds is that it runs without the reference installed, and that `run()` asks GPU children for a whole page of
queries at once (their `_retrieve_block`) instead of one SQL statement per query and child.

Children are any objects with the retrieval-pipeline surface (`name`, async `_retrieve_by_id` / `_retrieve_by_text`):
the pipelines of this package, the reference's own (BM25, ...), or a mix.  Differences from the reference, both in
places where it is itself order-dependent: candidates are visited in first-seen order (primary list, then the second
list's new ids) where `_cc_fuse` iterates a Python set, so exact score ties resolve deterministically here.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Literal

from .compat import BaseRetrievalPipelineConfig
from .gqr import _load_child
from .pipelines import Mi355BaseRetrievalPipeline, child_page, get_retrieval_pipeline_unit

NormalizationMethod = Literal["mm", "tmm", "z", "dbsf"]
logger = logging.getLogger("AutoRAG-Research")

# what a list "would have scored" a document it did not return, after normalisation (hybrid.py:33-43)
MISSING_SCORE_FLOORS: dict[str, float] = {"mm": 0.0, "tmm": 0.0, "z": -3.0, "dbsf": 0.0}


def _present(scores: list) -> list[float]:
    return [s for s in scores if s is not None]


def normalize_minmax(scores: list) -> list:
    """(s - min) / (max - min) over the present entries, 0.5 each when they are all equal; None stays None (util.py:371-404)."""
    live = _present(scores)
    if not live:
        return list(scores)
    lo, span = min(live), max(live) - min(live)
    if span == 0:
        return [None if s is None else 0.5 for s in scores]
    return [None if s is None else (s - lo) / span for s in scores]


def normalize_tmm(scores: list, theoretical_min: float) -> list:
    """(s - theoretical_min) / (max - theoretical_min), 0.5 each on a zero span (util.py:407-443)."""
    live = _present(scores)
    if not live:
        return list(scores)
    span = max(live) - theoretical_min
    if span == 0:
        return [None if s is None else 0.5 for s in scores]
    return [None if s is None else (s - theoretical_min) / span for s in scores]


def _mean_std(live: list[float]) -> tuple[float, float]:
    n = len(live)
    mean = sum(live) / n
    return mean, (sum((s - mean) ** 2 for s in live) / n) ** 0.5  # population standard deviation


def normalize_zscore(scores: list) -> list:
    """(s - mean) / std, all 0.0 when std is 0 (util.py:446-480)."""
    live = _present(scores)
    if not live:
        return list(scores)
    mean, std = _mean_std(live)
    if std == 0:
        return [None if s is None else 0.0 for s in scores]
    return [None if s is None else (s - mean) / std for s in scores]


def normalize_dbsf(scores: list) -> list:
    """Position inside [mean - 3 std, mean + 3 std], clipped to [0, 1]; 0.5 each when std is 0 (util.py:483-530)."""
    live = _present(scores)
    if not live:
        return list(scores)
    mean, std = _mean_std(live)
    if std == 0:
        return [None if s is None else 0.5 for s in scores]
    lo, span = mean - 3 * std, (mean + 3 * std) - (mean - 3 * std)
    return [None if s is None else max(0.0, min(1.0, (s - lo) / span)) for s in scores]


def rrf_fuse(results_1: list[dict], results_2: list[dict], k: int, top_k: int, fetch_k: int) -> list[dict[str, Any]]:
    """sum_i 1 / (k + rank_i(d)); a list that did not return d counts as rank fetch_k + 1 (hybrid.py:46-97)."""
    absent = 1.0 / (k + fetch_k + 1)
    in_1 = {r["doc_id"] for r in results_1}
    in_2 = {r["doc_id"] for r in results_2}
    fused: dict[Any, float] = {}
    for results in (results_1, results_2):
        for rank, r in enumerate(results, start=1):
            fused[r["doc_id"]] = fused.get(r["doc_id"], 0.0) + 1.0 / (k + rank)
    for pk in fused:
        if (pk in in_1) != (pk in in_2):
            fused[pk] += absent
    ranked = sorted(fused.items(), key=lambda item: item[1], reverse=True)  # stable: first-seen order between equals
    return [{"doc_id": pk, "score": score} for pk, score in ranked[:top_k]]


def cc_fuse(results_1: list[dict], results_2: list[dict], weight: float, top_k: int, normalize_method: str,
            pipeline_1_min: float | None = None, pipeline_2_min: float | None = None) -> list[dict[str, Any]]:
    """weight * norm(scores_1) + (1 - weight) * norm(scores_2); absent scores take the method's floor after
    normalisation and do not enter its statistics (hybrid.py:100-178)."""
    raw_1 = {r["doc_id"]: r["score"] for r in results_1}
    raw_2 = {r["doc_id"]: r["score"] for r in results_2}
    ids = list(raw_1) + [pk for pk in raw_2 if pk not in raw_1]
    col_1 = [raw_1.get(pk) for pk in ids]
    col_2 = [raw_2.get(pk) for pk in ids]
    if normalize_method == "mm":
        n1, n2 = normalize_minmax(col_1), normalize_minmax(col_2)
    elif normalize_method == "tmm":
        if pipeline_1_min is None:
            raise ValueError("TMM normalization requires pipeline_1_min")
        if pipeline_2_min is None:
            raise ValueError("TMM normalization requires pipeline_2_min")
        n1, n2 = normalize_tmm(col_1, pipeline_1_min), normalize_tmm(col_2, pipeline_2_min)
    elif normalize_method == "z":
        n1, n2 = normalize_zscore(col_1), normalize_zscore(col_2)
    elif normalize_method == "dbsf":
        n1, n2 = normalize_dbsf(col_1), normalize_dbsf(col_2)
    else:
        raise ValueError(f"Unknown normalization method: {normalize_method}")
    floor = MISSING_SCORE_FLOORS[normalize_method]
    fused = {pk: weight * (floor if a is None else a) + (1 - weight) * (floor if b is None else b)
             for pk, a, b in zip(ids, n1, n2)}
    ranked = sorted(fused.items(), key=lambda item: item[1], reverse=True)
    return [{"doc_id": pk, "score": score} for pk, score in ranked[:top_k]]


@dataclass(kw_only=True)
class _HybridConfig(BaseRetrievalPipelineConfig):
    retrieval_pipeline_1_name: str
    retrieval_pipeline_2_name: str
    fetch_k_multiplier: int = 2
    device: int = 0


@dataclass(kw_only=True)
class Mi355HybridRRFPipelineConfig(_HybridConfig):
    """Fields as HybridRRFRetrievalPipelineConfig (hybrid.py:199-237) + `device`."""

    rrf_k: int = 60

    def get_pipeline_class(self) -> type["Mi355HybridRRFRetrievalPipeline"]:
        return Mi355HybridRRFRetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"retrieval_pipeline_1": self.retrieval_pipeline_1_name, "retrieval_pipeline_2": self.retrieval_pipeline_2_name,
                "rrf_k": self.rrf_k, "fetch_k_multiplier": self.fetch_k_multiplier, "device": self.device}


@dataclass(kw_only=True)
class Mi355HybridCCPipelineConfig(_HybridConfig):
    """Fields as HybridCCRetrievalPipelineConfig (hybrid.py:239-286) + `device`."""

    weight: float = 0.5
    normalize_method: NormalizationMethod = "mm"
    pipeline_1_min: float | None = None
    pipeline_2_min: float | None = None

    def get_pipeline_class(self) -> type["Mi355HybridCCRetrievalPipeline"]:
        return Mi355HybridCCRetrievalPipeline

    def get_pipeline_kwargs(self) -> dict[str, Any]:
        return {"retrieval_pipeline_1": self.retrieval_pipeline_1_name, "retrieval_pipeline_2": self.retrieval_pipeline_2_name,
                "weight": self.weight, "normalize_method": self.normalize_method, "pipeline_1_min": self.pipeline_1_min,
                "pipeline_2_min": self.pipeline_2_min, "fetch_k_multiplier": self.fetch_k_multiplier, "device": self.device}


class _Mi355HybridBase(Mi355BaseRetrievalPipeline):
    """Two children, one fusion rule (reference HybridRetrievalPipeline, hybrid.py:289-437)."""

    def __init__(self, session_factory: Any, name: str, retrieval_pipeline_1: Any, retrieval_pipeline_2: Any,
                 fetch_k_multiplier: int = 2, schema: Any | None = None, config_dir: Path | None = None, device: int = 0):
        if isinstance(retrieval_pipeline_1, str):
            retrieval_pipeline_1 = _load_child(retrieval_pipeline_1, session_factory, schema, config_dir)
        if isinstance(retrieval_pipeline_2, str):
            retrieval_pipeline_2 = _load_child(retrieval_pipeline_2, session_factory, schema, config_dir)
        self._retrieval_pipeline_1 = retrieval_pipeline_1
        self._retrieval_pipeline_2 = retrieval_pipeline_2
        self.fetch_k_multiplier = fetch_k_multiplier
        if self.retrieval_unit == "mixed":  # raw doc ids of two tables would collide (hybrid.py:375-380)
            raise ValueError("Mixed retrieval_unit hybrid pipelines are not supported until fused results carry entity namespaces.")
        super().__init__(session_factory, name, schema, device=device)

    @property
    def retrieval_unit(self):  # type: ignore[override]
        """The children's shared unit, "mixed" when they differ (hybrid.py:366-373)."""
        a = get_retrieval_pipeline_unit(self._retrieval_pipeline_1)
        b = get_retrieval_pipeline_unit(self._retrieval_pipeline_2)
        return a if a == b else "mixed"

    def _fuse_results(self, results_1: list[dict], results_2: list[dict], top_k: int, fetch_k: int) -> list[dict]:
        raise NotImplementedError

    async def _retrieve_by_id(self, query_id, top_k: int) -> list[dict[str, Any]]:
        fetch_k = top_k * self.fetch_k_multiplier
        results_1 = await self._retrieval_pipeline_1._retrieve_by_id(query_id, fetch_k)
        results_2 = await self._retrieval_pipeline_2._retrieve_by_id(query_id, fetch_k)
        return self._fuse_results(results_1, results_2, top_k, fetch_k)

    async def _retrieve_by_text(self, query_text: str, top_k: int) -> list[dict[str, Any]]:
        fetch_k = top_k * self.fetch_k_multiplier
        results_1 = await self._retrieval_pipeline_1._retrieve_by_text(query_text, fetch_k)
        results_2 = await self._retrieval_pipeline_2._retrieve_by_text(query_text, fetch_k)
        return self._fuse_results(results_1, results_2, top_k, fetch_k)

    def _retrieve_block(self, query_ids: list, top_k: int) -> list[list[dict] | None]:
        fetch_k = top_k * self.fetch_k_multiplier
        page_1 = child_page(self._retrieval_pipeline_1, query_ids, fetch_k)
        page_2 = child_page(self._retrieval_pipeline_2, query_ids, fetch_k)
        return [None if a is None or b is None else self._fuse_results(a, b, top_k, fetch_k) for a, b in zip(page_1, page_2)]


class Mi355HybridRRFRetrievalPipeline(_Mi355HybridBase):
    """Reciprocal Rank Fusion of two children (reference HybridRRFRetrievalPipeline, hybrid.py:440-534)."""

    def __init__(self, session_factory: Any, name: str, retrieval_pipeline_1: Any, retrieval_pipeline_2: Any, rrf_k: int = 60,
                 fetch_k_multiplier: int = 2, schema: Any | None = None, config_dir: Path | None = None, device: int = 0):
        self.rrf_k = rrf_k
        super().__init__(session_factory, name, retrieval_pipeline_1, retrieval_pipeline_2, fetch_k_multiplier, schema,
                         config_dir, device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        return {"type": "mi355_hybrid_rrf", "retrieval_unit": self.retrieval_unit,
                "retrieval_pipeline_1": self._retrieval_pipeline_1.name, "retrieval_pipeline_2": self._retrieval_pipeline_2.name,
                "rrf_k": self.rrf_k, "fetch_k_multiplier": self.fetch_k_multiplier}

    def _fuse_results(self, results_1, results_2, top_k: int, fetch_k: int) -> list[dict[str, Any]]:
        return rrf_fuse(results_1, results_2, self.rrf_k, top_k, fetch_k)


class Mi355HybridCCRetrievalPipeline(_Mi355HybridBase):
    """Convex combination of normalised scores (reference HybridCCRetrievalPipeline, hybrid.py:537-641)."""

    def __init__(self, session_factory: Any, name: str, retrieval_pipeline_1: Any, retrieval_pipeline_2: Any,
                 weight: float = 0.5, normalize_method: NormalizationMethod = "mm", pipeline_1_min: float | None = None,
                 pipeline_2_min: float | None = None, fetch_k_multiplier: int = 2, schema: Any | None = None,
                 config_dir: Path | None = None, device: int = 0):
        self.weight = weight
        self.normalize_method = normalize_method
        self.pipeline_1_min = pipeline_1_min
        self.pipeline_2_min = pipeline_2_min
        super().__init__(session_factory, name, retrieval_pipeline_1, retrieval_pipeline_2, fetch_k_multiplier, schema,
                         config_dir, device)

    def _get_pipeline_config(self) -> dict[str, Any]:
        config = {"type": "mi355_hybrid_cc", "retrieval_unit": self.retrieval_unit,
                  "retrieval_pipeline_1": self._retrieval_pipeline_1.name,
                  "retrieval_pipeline_2": self._retrieval_pipeline_2.name, "weight": self.weight,
                  "normalize_method": self.normalize_method, "fetch_k_multiplier": self.fetch_k_multiplier}
        if self.normalize_method == "tmm":
            config["pipeline_1_min"] = self.pipeline_1_min
            config["pipeline_2_min"] = self.pipeline_2_min
        return config

    def _fuse_results(self, results_1, results_2, top_k: int, fetch_k: int) -> list[dict[str, Any]]:
        return cc_fuse(results_1, results_2, self.weight, top_k, self.normalize_method, self.pipeline_1_min,
                       self.pipeline_2_min)


__all__ = ["MISSING_SCORE_FLOORS", "Mi355HybridCCPipelineConfig", "Mi355HybridCCRetrievalPipeline",
           "Mi355HybridRRFPipelineConfig", "Mi355HybridRRFRetrievalPipeline", "cc_fuse", "normalize_dbsf",
           "normalize_minmax", "normalize_tmm", "normalize_zscore", "rrf_fuse"]
