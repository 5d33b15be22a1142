"""Retrieval metrics with the reference's AND/OR-group semantics (host side, float64).

Mirrors autorag_research/evaluation/metrics/retrieval.py:11-236 and the `metric` wrapper of
autorag_research/evaluation/metrics/util.py:53-89 (per-input evaluation, `None` for inputs whose
`retrieval_gt` is missing/empty/blank).  Ground truth is a list of OR-groups that are AND-ed:
``[[a, b], [c]]`` means (a or b) and c.  BEIR corpora are ingested as ONE OR-group, hotpotqa as an
AND-chain (reference data/beir.py:191-194), so this nDCG is *not* TREC nDCG.
Pinned by tests/test_metrics_golden.py against fixtures generated from the imported reference.
"""

from __future__ import annotations

import math
from collections.abc import Callable, Sequence
from dataclasses import dataclass
from functools import wraps
from typing import Any

import numpy as np

__all__ = ["MetricInput", "retrieval_ndcg", "retrieval_recall", "retrieval_precision", "retrieval_f1",
           "retrieval_mrr", "retrieval_map", "retrieval_full_recall", "METRICS"]


@dataclass
class MetricInput:
    """The fields of the reference's MetricInput (schema.py:30-44) that retrieval metrics read."""

    retrieval_gt: list[list[str]] | None = None
    retrieved_ids: list[str] | None = None
    relevance_scores: dict[str, int] | None = None
    query: str | None = None


def _leaf_ok(x: Any) -> bool:
    if isinstance(x, str):
        return len(x.strip()) > 0
    if isinstance(x, (list, np.ndarray)):
        return _seq_ok(x)
    return isinstance(x, (int, float)) and not isinstance(x, bool) or type(x) in (int, float)


def _seq_ok(seq: Any) -> bool:
    items = seq.flatten().tolist() if isinstance(seq, np.ndarray) else list(seq)
    if len(items) == 0:
        return False
    return all(it is not None and _leaf_ok(it) for it in items)


def _field_ok(value: Any) -> bool:
    """Same acceptance rule as MetricInput.is_fields_notnone (schema.py:46-58,91-120)."""
    if value is None:
        return False
    try:
        if isinstance(value, str):
            return len(value.strip()) > 0
        if isinstance(value, (list, np.ndarray)):
            return _seq_ok(value)
        return type(value) in (int, float)
    except Exception:  # noqa: BLE001
        return False


def _to_list(x: Any) -> Any:
    if isinstance(x, np.ndarray):
        return [_to_list(v) for v in x.tolist()]
    if isinstance(x, (list, tuple)):
        return [_to_list(v) for v in x]
    return x


def _per_input(fields: Sequence[str]) -> Callable:
    def deco(fn: Callable[[MetricInput], float]) -> Callable:
        @wraps(fn)
        def run(metric_inputs: Sequence[MetricInput] | MetricInput | None = None, **kw) -> list[float | None]:
            inputs = metric_inputs if metric_inputs is not None else kw.pop("metric_inputs")
            if isinstance(inputs, MetricInput):
                inputs = [inputs]
            out: list[float | None] = []
            for mi in inputs:
                if all(_field_ok(getattr(mi, f)) for f in fields):
                    norm = MetricInput(retrieval_gt=_to_list(mi.retrieval_gt), retrieved_ids=_to_list(mi.retrieved_ids),
                                       relevance_scores=mi.relevance_scores, query=mi.query)
                    out.append(fn(norm))
                else:
                    out.append(None)
            return out

        run.single = fn  # type: ignore[attr-defined]
        return run

    return deco


def _groups(gt: list[list[str]]) -> list[list[str]]:
    return [g for g in gt if g and g != [""]]


@_per_input(["retrieval_gt"])
def retrieval_ndcg(mi: MetricInput) -> float:
    """Group nDCG (reference retrieval.py:71-144): a hit counts only when it is the first to satisfy a group."""
    if mi.retrieved_ids is None or mi.retrieval_gt is None:
        return 0.0
    groups = _groups(mi.retrieval_gt)
    if not groups:
        return 0.0
    member: dict[str, list[int]] = {}
    for gi, g in enumerate(groups):
        for item in g:
            if item:
                member.setdefault(item, []).append(gi)
    rel = mi.relevance_scores or {item: 1 for g in groups for item in g}
    done: set[int] = set()
    dcg = 0.0
    for rank, doc in enumerate(mi.retrieved_ids):
        fresh = [gi for gi in member.get(doc, ()) if gi not in done]
        if fresh:
            done.update(fresh)
            dcg += (2 ** rel.get(doc, 0) - 1) / math.log2(rank + 2)
    best = sorted((max((rel.get(item, 0) for item in g if item), default=0) for g in groups), reverse=True)
    idcg = sum((2 ** s - 1) / math.log2(i + 2) for i, s in enumerate(best))
    return dcg / idcg if idcg > 0 else 0.0


@_per_input(["retrieval_gt"])
def retrieval_recall(mi: MetricInput) -> float:
    """Fraction of groups with at least one retrieved member (retrieval.py:30-48)."""
    if mi.retrieved_ids is None or mi.retrieval_gt is None:
        return 0.0
    got = set(mi.retrieved_ids)
    gt = mi.retrieval_gt
    return sum(1 for g in gt if got & set(g)) / len(gt) if gt else 0.0


@_per_input(["retrieval_gt"])
def retrieval_precision(mi: MetricInput) -> float:
    """Distinct retrieved ids that belong to any group / number of retrieved ids (retrieval.py:51-68)."""
    if mi.retrieved_ids is None or mi.retrieval_gt is None:
        return 0.0
    union = set().union(*[set(g) for g in mi.retrieval_gt]) if mi.retrieval_gt else set()
    got = set(mi.retrieved_ids)
    return len(got & union) / len(mi.retrieved_ids) if mi.retrieved_ids else 0.0


@_per_input(["retrieval_gt"])
def retrieval_f1(mi: MetricInput) -> float:
    r = retrieval_recall.single(mi)  # type: ignore[attr-defined]
    p = retrieval_precision.single(mi)  # type: ignore[attr-defined]
    return 0 if r + p == 0 else 2 * r * p / (r + p)


@_per_input(["retrieval_gt"])
def retrieval_full_recall(mi: MetricInput) -> float:
    """1.0 iff every group has a retrieved member (retrieval.py:147-171)."""
    if mi.retrieved_ids is None or mi.retrieval_gt is None:
        return 0.0
    got = set(mi.retrieved_ids)
    return 1.0 if all(got & set(g) for g in mi.retrieval_gt) else 0.0


@_per_input(["retrieval_gt"])
def retrieval_mrr(mi: MetricInput) -> float:
    """Mean over groups of 1/rank of the group's first hit; groups never hit add 0 (retrieval.py:174-202)."""
    if mi.retrieved_ids is None or mi.retrieval_gt is None:
        return 0.0
    rr = []
    for g in mi.retrieval_gt:
        gs = set(g)
        for i, doc in enumerate(mi.retrieved_ids):
            if doc in gs:
                rr.append(1.0 / (i + 1))
                break
    return sum(rr) / len(mi.retrieval_gt) if rr else 0.0


@_per_input(["retrieval_gt"])
def retrieval_map(mi: MetricInput) -> float:
    """Mean over groups of average precision against that group (retrieval.py:205-236)."""
    if mi.retrieved_ids is None or mi.retrieval_gt is None:
        return 0.0
    aps = []
    for g in mi.retrieval_gt:
        gs = set(g)
        hits = 0
        precs = []
        for i, doc in enumerate(mi.retrieved_ids):
            if doc in gs:
                hits += 1
                precs.append(hits / (i + 1))
        aps.append(sum(precs) / len(precs) if precs else 0.0)
    return sum(aps) / len(mi.retrieval_gt) if aps else 0.0


METRICS: dict[str, Callable] = {
    "ndcg": retrieval_ndcg,
    "recall": retrieval_recall,
    "precision": retrieval_precision,
    "f1": retrieval_f1,
    "mrr": retrieval_mrr,
    "map": retrieval_map,
    "full_recall": retrieval_full_recall,
}
