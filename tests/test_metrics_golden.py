"""CPU: retrieval metrics == the reference's evaluation/metrics/retrieval.py on the committed golden table."""

import json

import pytest

from helpers import GOLDEN


@pytest.fixture(scope="module")
def cases():
    return json.loads((GOLDEN / "metrics_golden.json").read_text())["cases"]


def test_all_metrics_match_reference(cases):
    from autorag_research_amd.metrics import METRICS, MetricInput

    inputs = [MetricInput(retrieval_gt=c["retrieval_gt"], retrieved_ids=c["retrieved_ids"],
                          relevance_scores=c["relevance_scores"]) for c in cases]
    assert len(inputs) > 70
    for name, fn in METRICS.items():
        got = fn(metric_inputs=inputs)
        for c, g in zip(cases, got):
            exp = c["expected"][name]
            if exp is None:
                assert g is None, (name, c)
            else:
                assert g == pytest.approx(exp, rel=1e-12, abs=1e-15), (name, c)


def test_reference_known_answers_verbatim():
    """the numbers of tests/autorag_research/evaluation/metrics/test_retrieval.py:135-148."""
    from autorag_research_amd.metrics import MetricInput, retrieval_ndcg

    gt = [[["test-1", "test-2"], ["test-3"]], [["test-4", "test-5"], ["test-6", "test-7"], ["test-8"]],
          [["test-9", "test-10"]], [["test-11"], ["test-12"], ["test-13"]], [["test-14"]], [[]], [[""]], [["test-15"]]]
    pred = [["test-1", "pred-1", "test-2", "pred-3"], ["test-6", "pred-5", "pred-6", "pred-7"],
            ["test-9", "pred-0", "pred-8", "pred-9"], ["test-13", "test-12", "pred-10", "pred-11"],
            ["test-14", "pred-12"], ["pred-13"], ["pred-14"], ["pred-15", "pred-16", "test-15"]]
    sol = [0.6131471927654584, 0.4693015838914927, 1.0, 0.7653606369886217, 1, None, None, 0.5]
    got = retrieval_ndcg(metric_inputs=[MetricInput(retrieval_gt=g, retrieved_ids=p) for g, p in zip(gt, pred)])
    for s, g in zip(sol, got):
        assert (g is None) if s is None else g == pytest.approx(s, rel=1e-4)  # the reference test's own tolerance
