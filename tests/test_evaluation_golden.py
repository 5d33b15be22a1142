"""SURVEY 8(a) row a14 against the reference itself: tests/golden/evaluation_golden.json holds the outputs of the reference's
`build_retrieval_gt_from_relations` and `RetrievalEvaluationService._get_execution_results`
(orm/service/retrieval_evaluation.py:23-78, 161-217; make_golden.py:make_evaluation) -- ground-truth structure, ranked-list
order incl. chunk / image-chunk ties and NULL scores.  `autorag_research_amd.evaluation` must return the same objects."""

import json
from pathlib import Path

import pytest

GOLDEN = json.loads((Path(__file__).parent / "golden" / "evaluation_golden.json").read_text())


@pytest.mark.parametrize("ci", range(len(GOLDEN["relations"])))
def test_ground_truth_from_relations(ci):
    from autorag_research_amd.evaluation import build_retrieval_gt_from_relations
    from autorag_research_amd.store import RetrievalRelation

    case = GOLDEN["relations"][ci]
    rels = [RetrievalRelation(query_id="q", **r) for r in case["rows"]]
    gt, scores = build_retrieval_gt_from_relations(rels)
    assert gt == case["retrieval_gt"]
    assert scores == case["relevance_scores"] and list(scores) == list(case["relevance_scores"])   # insertion order too


def _store():
    from autorag_research_amd.store import InMemoryStore, RetrievalRelation

    ex = GOLDEN["execution"]
    s = InMemoryStore()
    s.add_queries(ex["query_ids"])
    pid = ex["pipeline_id"]
    for q, c, sc in ex["chunk_rows"]:
        s.chunk_results.setdefault((pid, q), []).append((c, sc))
    for q, c, sc in ex["image_chunk_rows"]:
        s.image_chunk_results.setdefault((pid, q), []).append((c, sc))
    for q, c, sc in ex["other_pipeline_rows"]:
        s.chunk_results.setdefault((pid + 1, q), []).append((c, sc))
    for q, rows in ex["relations"].items():
        s.add_relations([RetrievalRelation(query_id=q, **r) for r in rows])
    return s, ex


def test_execution_results_equal_the_reference():
    from autorag_research_amd.evaluation import get_execution_results

    s, ex = _store()
    got = get_execution_results(s, ex["pipeline_id"], ex["query_ids"])
    assert list(got) == ex["query_ids"]
    for q in ex["query_ids"]:
        assert got[q] == ex["results"][q], q


def test_ndcg_on_the_reference_ranked_lists():
    """The metric sees exactly the reference's inputs: nDCG of the fixture's ranked lists == nDCG of ours."""
    from autorag_research_amd.evaluation import evaluate
    from autorag_research_amd.metrics import MetricInput, retrieval_ndcg

    s, ex = _store()
    n, mean, per = evaluate(s, ex["pipeline_id"], retrieval_ndcg, ex["query_ids"])
    exp = retrieval_ndcg(metric_inputs=[MetricInput(retrieved_ids=ex["results"][q]["retrieved_ids"],
                                                    retrieval_gt=ex["results"][q]["retrieval_gt"],
                                                    relevance_scores=ex["results"][q]["relevance_scores"]) for q in ex["query_ids"]])
    assert [per[q] for q in ex["query_ids"]] == exp
    good = [v for v in exp if v is not None]
    assert n == len(good) and mean == sum(good) / len(good)
