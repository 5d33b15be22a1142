"""Retrieval evaluation over persisted results (host side).

Mirrors the part of RetrievalEvaluationService the nDCG@10 number depends on
(autorag_research/orm/service/retrieval_evaluation.py):
  build_retrieval_gt_from_relations   :23-78    group_index = AND, group_order = OR order, `chunk_` / `image_chunk_` prefixes
  _get_execution_results              :161-217  ranked list = persisted rows as the repositories hand them out (`ORDER BY
                                                rel_score DESC`, orm/repository/chunk_retrieved_result.py:34-38: PostgreSQL
                                                puts NULL scores FIRST on DESC), chunk rows then image_chunk rows, re-sorted
                                                by `(rel_score or 0.0)` DESC with Python's stable sort; ids prefixed
Pinned by tests/golden/evaluation_golden.json (the reference's two functions over a fake Unit of Work).
  evaluate -> mean of per-query scores, None results skipped   orm/service/base_evaluation.py:290-375
"""

from __future__ import annotations

from collections import defaultdict
from collections.abc import Callable
from typing import Any

from .metrics import MetricInput
from .store import InMemoryStore, RetrievalRelation


def build_retrieval_gt_from_relations(relations: list[RetrievalRelation]) -> tuple[list[list[str]], dict[str, int]]:
    grouped: dict[int, list[tuple[int, str]]] = defaultdict(list)
    scores: dict[str, int] = {}
    for rel in relations:
        if rel.chunk_id is not None:
            pid = f"chunk_{rel.chunk_id}"
        elif rel.image_chunk_id is not None:
            pid = f"image_chunk_{rel.image_chunk_id}"
        else:
            continue
        scores[pid] = rel.score if rel.score is not None else 1
        grouped[rel.group_index].append((rel.group_order, pid))
    gt = [[pid for _, pid in sorted(grouped[g], key=lambda x: x[0])] for g in sorted(grouped)]
    return gt, scores


def get_execution_results(store: InMemoryStore, pipeline_id: int, query_ids: list) -> dict[Any, dict[str, Any]]:
    out: dict[Any, dict[str, Any]] = {}

    def sql_order(rows):  # ORDER BY rel_score DESC over rows in insertion order: NULLs first, ties as inserted
        return sorted(rows, key=lambda r: (r[1] is None, r[1] if r[1] is not None else 0.0), reverse=True)

    for qid in query_ids:
        rows = [(s or 0.0, f"chunk_{cid}") for cid, s in sql_order(store.chunk_results.get((pipeline_id, qid), []))]
        rows += [(s or 0.0, f"image_chunk_{cid}")
                 for cid, s in sql_order(store.image_chunk_results.get((pipeline_id, qid), []))]
        rows.sort(key=lambda x: x[0], reverse=True)  # stable, like the reference's python re-sort
        gt, rel = build_retrieval_gt_from_relations(store.relations.get(qid, []))
        out[qid] = {"retrieved_ids": [pid for _, pid in rows], "retrieval_gt": gt, "relevance_scores": rel}
    return out


def evaluate(store: InMemoryStore, pipeline_id: int, metric_func: Callable, query_ids: list | None = None
             ) -> tuple[int, float | None, dict[Any, float | None]]:
    """Returns (queries evaluated, mean score, per-query scores); queries whose metric is None are skipped."""
    qids = list(store.query_order) if query_ids is None else list(query_ids)
    res = get_execution_results(store, pipeline_id, qids)
    inputs = [MetricInput(retrieved_ids=res[q]["retrieved_ids"], retrieval_gt=res[q]["retrieval_gt"],
                          relevance_scores=res[q]["relevance_scores"]) for q in qids]
    vals = metric_func(metric_inputs=inputs)
    per = dict(zip(qids, vals))
    good = [v for v in vals if v is not None]
    return len(good), (sum(good) / len(good) if good else None), per
