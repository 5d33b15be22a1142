"""Generate the committed golden fixtures by IMPORTING the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

/root/reference never travels to the GPU box, so its outputs are frozen here as data:
inputs (or the seeds that regenerate them) + expected outputs.  No reference source is
copied; the import recipe (stub modules for uninstalled third-party deps) follows
SURVEY.md Appendix A.

Fixtures written next to this file:
  metrics_golden.json      reference evaluation/metrics/retrieval.py outputs (ndcg/recall/precision/
                           f1/mrr/map/full_recall) on the reference's own known-answer table
                           (tests/autorag_research/evaluation/metrics/test_retrieval.py:15-41) and on
                           seeded synthetic OR-group / AND-chain ground truth.
  scores_golden.npz        seeded small inputs + outputs of the reference's in-process math:
                           gqr_hybrid._cosine_scores / _maxsim_scores (pipelines/retrieval/gqr_hybrid.py:65-106),
                           HEAVENRetrievalPipeline._score_candidates (pipelines/retrieval/heaven.py:246-267),
                           evaluation.metrics.util.calculate_cosine_similarity (evaluation/metrics/util.py:10-24).
  service_golden.json      output dicts of the reference's RetrievalPipelineService.vector_search /
                           vector_search_by_embedding (orm/service/retrieval_pipeline.py:467-550),
                           VectorSearchRetrievalPipeline._retrieve_by_id (pipelines/retrieval/vector_search.py:157-169)
                           and ImageVectorSearchRetrievalPipeline._retrieve_by_id
                           (pipelines/retrieval/image_vector_search.py:65-94) driven over a fake
                           Unit-of-Work whose SQL operators are answered by the CPU oracle.
"""

from __future__ import annotations

import asyncio
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True


def _stub(name: str) -> None:
    m = types.ModuleType(name)
    m.__path__ = []  # type: ignore[attr-defined]
    m.__getattr__ = lambda n: type(n, (object,), {  # type: ignore[assignment]
        "__init__": lambda s, *a, **k: None,
        "__class_getitem__": classmethod(lambda c, i: c),
    })
    sys.modules[name] = m


for _n in ["tiktoken", "evaluate", "hydra", "hydra.utils", "langchain_core", "langchain_core.embeddings",
           "langchain_core.language_models", "nltk", "omegaconf", "pgvector", "pgvector.sqlalchemy", "rouge_score",
           "rouge_score.rouge_scorer", "sacrebleu", "sacrebleu.metrics", "sacrebleu.metrics.bleu", "tenacity",
           "psycopg", "dotenv"]:
    _stub(_n)

from sqlalchemy.types import UserDefinedType  # noqa: E402


class _Vector(UserDefinedType):
    cache_ok = True

    def __init__(self, dim=None):
        self.dim = dim

    def get_col_spec(self, **kw):
        return f"VECTOR({self.dim})"


sys.modules["pgvector.sqlalchemy"].Vector = _Vector  # type: ignore[attr-defined]

from autorag_research.evaluation.metrics import retrieval as ref_metrics  # noqa: E402
from autorag_research.evaluation.metrics.util import calculate_cosine_similarity  # noqa: E402
from autorag_research.orm.service.retrieval_pipeline import RetrievalPipelineService  # noqa: E402
from autorag_research.pipelines.retrieval.gqr_hybrid import _cosine_scores, _maxsim_scores  # noqa: E402
from autorag_research.pipelines.retrieval.heaven import HEAVENRetrievalPipeline  # noqa: E402
from autorag_research.pipelines.retrieval.image_vector_search import ImageVectorSearchRetrievalPipeline  # noqa: E402
from autorag_research.pipelines.retrieval.vector_search import VectorSearchRetrievalPipeline  # noqa: E402
from autorag_research.schema import MetricInput  # noqa: E402

from oracle import cpu_ref  # noqa: E402

# --------------------------------------------------------------------------------------
# 1. metrics
# --------------------------------------------------------------------------------------

METRIC_FUNCS = {
    "ndcg": ref_metrics.retrieval_ndcg,
    "recall": ref_metrics.retrieval_recall,
    "precision": ref_metrics.retrieval_precision,
    "f1": ref_metrics.retrieval_f1,
    "mrr": ref_metrics.retrieval_mrr,
    "map": ref_metrics.retrieval_map,
    "full_recall": ref_metrics.retrieval_full_recall,
}


def make_metrics() -> dict:
    cases = []
    # the reference's own known-answer table (data, test_retrieval.py:15-41)
    known_gt = [
        [["test-1", "test-2"], ["test-3"]],
        [["test-4", "test-5"], ["test-6", "test-7"], ["test-8"]],
        [["test-9", "test-10"]],
        [["test-11"], ["test-12"], ["test-13"]],
        [["test-14"]],
        [[]],
        [[""]],
        [["test-15"]],
    ]
    known_pred = [
        ["test-1", "pred-1", "test-2", "pred-3"],
        ["test-6", "pred-5", "pred-6", "pred-7"],
        ["test-9", "pred-0", "pred-8", "pred-9"],
        ["test-13", "test-12", "pred-10", "pred-11"],
        ["test-14", "pred-12"],
        ["pred-13"],
        ["pred-14"],
        ["pred-15", "pred-16", "test-15"],
    ]
    for gt, pr in zip(known_gt, known_pred, strict=True):
        cases.append({"retrieval_gt": gt, "retrieved_ids": pr, "relevance_scores": None, "origin": "known"})
    # graded-relevance cases
    cases.append({"retrieval_gt": [["doc_a", "doc_b"]], "retrieved_ids": ["doc_a", "doc_b"],
                  "relevance_scores": {"doc_a": 2, "doc_b": 1}, "origin": "graded"})
    cases.append({"retrieval_gt": [["doc_a", "doc_b"]], "retrieved_ids": ["doc_b", "doc_a"],
                  "relevance_scores": {"doc_a": 2, "doc_b": 1}, "origin": "graded"})
    cases.append({"retrieval_gt": [["a"], ["b"], ["c"]], "retrieved_ids": ["c", "x", "a", "b"],
                  "relevance_scores": {"a": 3, "b": 1, "c": 2}, "origin": "graded"})
    # seeded synthetic: BEIR-style single OR group (data/beir.py:191-194 or_all) and hotpotqa AND chain (and_all)
    rng = np.random.default_rng(20260927)
    for i in range(60):
        n_rel = int(rng.integers(1, 5))
        rel = [f"chunk_{int(x)}" for x in rng.choice(50, size=n_rel, replace=False)]
        retrieved = [f"chunk_{int(x)}" for x in rng.choice(60, size=10, replace=False)]
        if i % 2 == 0:
            gt = [rel]  # one OR group
            origin = "or_all"
        else:
            gt = [[r] for r in rel]  # AND chain
            origin = "and_all"
        scores = None
        if i % 5 == 0:
            scores = {r: int(rng.integers(0, 4)) for r in rel}
        cases.append({"retrieval_gt": gt, "retrieved_ids": retrieved, "relevance_scores": scores, "origin": origin})
    # a few with overlapping groups / duplicates in the ranked list
    cases.append({"retrieval_gt": [["a", "b"], ["b", "c"]], "retrieved_ids": ["b", "a", "c"], "relevance_scores": None,
                  "origin": "overlap"})
    cases.append({"retrieval_gt": [["a"], ["b"]], "retrieved_ids": ["a", "a", "b", "b"], "relevance_scores": None,
                  "origin": "dups"})
    cases.append({"retrieval_gt": [["a"], ["b"]], "retrieved_ids": [], "relevance_scores": None, "origin": "empty_pred"})

    inputs = [MetricInput(retrieval_gt=c["retrieval_gt"], retrieved_ids=c["retrieved_ids"],
                          relevance_scores=c["relevance_scores"]) for c in cases]
    for name, fn in METRIC_FUNCS.items():
        outs = fn(metric_inputs=inputs)
        for c, o in zip(cases, outs, strict=True):
            c.setdefault("expected", {})[name] = None if o is None else float(o)
    return {"cases": cases}


# --------------------------------------------------------------------------------------
# 2. in-process score math
# --------------------------------------------------------------------------------------


def make_scores() -> dict[str, np.ndarray]:
    rng = np.random.default_rng(777)
    out: dict[str, np.ndarray] = {}
    # single-vector cosine
    C = rng.standard_normal((257, 48)).astype(np.float32)
    Q = rng.standard_normal((5, 48)).astype(np.float32)
    C[13] *= 37.5  # un-normalised rows: cosine must not depend on length
    C[100] *= 1e-3
    out["cos_C"] = C
    out["cos_Q"] = Q
    out["cos_scores"] = np.stack([_cosine_scores(q.astype(np.float64), C.astype(np.float64)) for q in Q])
    out["cos_pair"] = np.array([[calculate_cosine_similarity(q, C[i]) for i in range(8)] for q in Q], dtype=np.float64)
    # multi-vector maxsim: ragged docs, unit-norm tokens (ColBERT/ColPali emit L2-normalised token vectors)
    d = 16
    lens = rng.integers(1, 12, size=40)
    docs = []
    for t in lens:
        m = rng.standard_normal((int(t), d)).astype(np.float32)
        m /= np.linalg.norm(m, axis=1, keepdims=True)
        docs.append(m)
    out["ms_tok"] = np.concatenate(docs, axis=0)
    out["ms_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    qs = []
    q_lens = [1, 3, 7, 32]
    for t in q_lens:
        m = rng.standard_normal((t, d)).astype(np.float32)
        m /= np.linalg.norm(m, axis=1, keepdims=True)
        qs.append(m)
    out["ms_qtok"] = np.concatenate(qs, axis=0)
    out["ms_qoff"] = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    ms = []
    heaven = []
    for q in qs:
        ms.append(_maxsim_scores(q.astype(np.float64), [dm.astype(np.float64) for dm in docs]))
        sc = HEAVENRetrievalPipeline._score_candidates(
            [[float(x) for x in v] for v in q], {i: [[float(x) for x in v] for v in dm] for i, dm in enumerate(docs)})
        heaven.append(np.array([sc[i] for i in range(len(docs))], dtype=np.float64))
    out["ms_scores"] = np.stack(ms)
    out["ms_scores_heaven"] = np.stack(heaven)
    # the reference's own hand-computed known answer (tests/.../test_gqr_hybrid_pipeline.py:65-74 -> [0.75, 0.6])
    qm = np.array([[1.0, 0.0], [0.0, 1.0]])
    cands = [np.array([[1.0, 0.0], [0.0, 0.5]]), np.array([[0.5, 0.5], [0.2, 0.7]])]
    out["ms_known_q"] = qm
    out["ms_known_c0"] = cands[0]
    out["ms_known_c1"] = cands[1]
    out["ms_known_scores"] = _maxsim_scores(qm, cands)
    return out


# --------------------------------------------------------------------------------------
# 3. service / pipeline dicts over a fake UoW
# --------------------------------------------------------------------------------------


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _FakeChunkRepo:
    """Stands in for BaseVectorRepository (orm/repository/base.py:378-426, 487-535): SQL answered by the oracle."""

    def __init__(self, ids, contents, single=None, tok=None, offsets=None):
        self.ids, self.contents, self.single, self.tok, self.offsets = ids, contents, single, tok, offsets

    def vector_search_with_scores(self, query_vector, vector_column="embedding", limit=10):
        if not query_vector:
            return []
        dist, rows = cpu_ref.topk_search(self.single, np.asarray(query_vector, dtype=np.float32)[None, :], limit)
        return [(_Obj(id=self.ids[r], contents=self.contents[r]), float(dv)) for dv, r in zip(dist[0], rows[0]) if r >= 0]

    def get_by_ids(self, ids):
        """BaseRepository.get_by_ids: rows with their multi-vector embeddings (HEAVEN stage 2, heaven.py:224-241)."""
        pos = {pk: i for i, pk in enumerate(self.ids)}
        out = []
        for pk in ids:
            if pk in pos:
                i = pos[pk]
                emb = [[float(x) for x in v] for v in self.tok[self.offsets[i]:self.offsets[i + 1]]]
                out.append(_Obj(id=pk, contents=self.contents[i], embeddings=emb))
        return out

    def maxsim_search(self, query_vectors, vector_column="embeddings", limit=10):
        if not query_vectors:
            return []
        q = np.asarray(query_vectors, dtype=np.float32)
        dist, rows = cpu_ref.maxsim_topk(self.tok, self.offsets, q, np.array([0, q.shape[0]], dtype=np.int32), limit)
        return [(_Obj(id=self.ids[r], contents=self.contents[r]), float(dv)) for dv, r in zip(dist[0], rows[0]) if r >= 0]


class _FakeQueryRepo:
    def __init__(self, queries):
        self.queries = queries

    def get_by_id(self, qid):
        return self.queries.get(qid)


class _FakeUow:
    def __init__(self, queries, chunks, image_chunks):
        self.queries, self.chunks, self.image_chunks = _FakeQueryRepo(queries), chunks, image_chunks

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _FakeService(RetrievalPipelineService):
    def __init__(self, uow):  # noqa: D107 - bypass DB wiring
        self._uow = uow

    def _create_uow(self):
        return self._uow


def make_service() -> dict:
    rng = np.random.default_rng(4242)
    n, d = 300, 32
    C = rng.standard_normal((n, d)).astype(np.float32)
    ids = [int(1000 + 3 * i) for i in range(n)]  # non-dense BIGINT pks
    contents = [f"chunk text {i}" for i in range(n)]
    dm = 8
    lens = rng.integers(2, 9, size=n)
    tok = rng.standard_normal((int(lens.sum()), dm)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    Q = rng.standard_normal((6, d)).astype(np.float32)
    # multi-vector queries live in a DB of dim dm (one embedding_dim per database, schema_factory.py:31)
    Qm = []
    for t in (1, 4, 5, 2, 3, 6):
        m = rng.standard_normal((t, dm)).astype(np.float32)
        m /= np.linalg.norm(m, axis=1, keepdims=True)
        Qm.append(m)
    queries = {}
    for i in range(6):
        queries[f"q{i}"] = _Obj(id=f"q{i}", contents=f"query text {i}", embedding=[float(x) for x in Q[i]],
                                embeddings=[[float(x) for x in v] for v in Qm[i]])
    queries["q_noemb"] = _Obj(id="q_noemb", embedding=None, embeddings=None)
    chunk_repo = _FakeChunkRepo(ids, contents, single=C, tok=tok, offsets=offsets)
    img_ids = [f"img-{i}" for i in range(n)]  # VARCHAR pks
    img_repo = _FakeChunkRepo(img_ids, [None] * n, single=C, tok=tok, offsets=offsets)
    svc = _FakeService(_FakeUow(queries, chunk_repo, img_repo))

    out = {"seed": 4242, "n": n, "d": d, "dm": dm, "chunk_ids": ids, "image_chunk_ids": img_ids, "top_k": 7}
    k = 7
    out["service_single"] = svc.vector_search([f"q{i}" for i in range(6)], top_k=k, search_mode="single")
    out["service_multi"] = svc.vector_search([f"q{i}" for i in range(6)], top_k=k, search_mode="multi")
    out["service_by_embedding"] = svc.vector_search_by_embedding([float(x) for x in Q[2]], top_k=k)

    def _pipe(cls, mode):
        p = cls.__new__(cls)
        p.search_mode = mode
        p._service = svc
        p._embedding_model = None
        return p

    loop = asyncio.new_event_loop()
    out["pipeline_single_q1"] = loop.run_until_complete(
        _pipe(VectorSearchRetrievalPipeline, "single")._retrieve_by_id("q1", k))
    out["pipeline_multi_q1"] = loop.run_until_complete(
        _pipe(VectorSearchRetrievalPipeline, "multi")._retrieve_by_id("q1", k))
    out["image_pipeline_multi_q2"] = loop.run_until_complete(
        _pipe(ImageVectorSearchRetrievalPipeline, "multi")._retrieve_by_id("q2", k))
    out["image_pipeline_single_q2"] = loop.run_until_complete(
        _pipe(ImageVectorSearchRetrievalPipeline, "single")._retrieve_by_id("q2", k))
    # HEAVEN two-stage flow (pipelines/retrieval/heaven.py:268-311) over the same fake UoW.  nltk is not installed
    # here: its POS tagger is replaced by a deterministic stand-in (even-length token -> noun), which the test
    # injects into the MI355X pipeline as well.
    import autorag_research.pipelines.retrieval.heaven as ref_heaven

    ref_heaven.nltk.pos_tag = lambda toks: [(t, "NN" if len(t) % 2 == 0 else "VB") for t in toks]
    hp = HEAVENRetrievalPipeline.__new__(HEAVENRetrievalPipeline)
    hp.stage1_candidate_count, hp.stage2_refine_ratio, hp.stage1_weight, hp.default_key_token_ratio = 40, 0.25, 0.3, 0.5
    hp._service = svc
    out["heaven_config"] = {"stage1_candidate_count": 40, "stage2_refine_ratio": 0.25, "stage1_weight": 0.3,
                            "default_key_token_ratio": 0.5}
    out["heaven"] = {qid: loop.run_until_complete(hp._retrieve_by_id(qid, k)) for qid in ("q1", "q2", "q5")}
    errs = {}
    for name, fn in {
        "missing": lambda: svc.vector_search(["nope"], 3),
        "noemb_single": lambda: svc.vector_search(["q_noemb"], 3, "single"),
        "noemb_multi": lambda: svc.vector_search(["q_noemb"], 3, "multi"),
    }.items():
        try:
            fn()
            errs[name] = None
        except Exception as e:  # noqa: BLE001
            errs[name] = [type(e).__name__, str(e)]
    out["errors"] = errs
    loop.close()
    return out


def main() -> None:
    (HERE / "metrics_golden.json").write_text(json.dumps(make_metrics(), indent=1))
    np.savez_compressed(HERE / "scores_golden.npz", **make_scores())
    (HERE / "service_golden.json").write_text(json.dumps(make_service(), indent=1))
    print("wrote", sorted(p.name for p in HERE.iterdir()))


if __name__ == "__main__":
    main()
