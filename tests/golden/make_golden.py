"""Generate the committed golden fixtures by IMPORTING the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
(no hash seed needed: the one case that depends on set iteration order is stored in an order-independent form)

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
  hybrid_golden.json       _rrf_fuse / _cc_fuse (pipelines/retrieval/hybrid.py:46-178, all four normalisers of util.py:371-530)
                           on seeded and edge-case result lists, and HybridRRF/CC _retrieve_by_id (:403-419) over the fake
                           Unit-of-Work with the recorded lexical child.
  hyde_golden.json         HyDERetrievalPipeline._retrieve_by_id / _retrieve_by_text (pipelines/retrieval/hyde.py:205-238) with
                           deterministic stand-in LLM / embedding objects (hyde_fake_models, shared with the test).
  executor_golden.json     the reference's plugin_registry scan of this package and its Executor's health-check -> run -> verify
                           flow (executor.py:308-463) over Mi355VectorSearchPipelineConfig: PipelineResult + persisted rows.
  pgtext_golden.json       VectorArray.process_bind_param / process_result_value (orm/types.py:210-277), _vec_to_pg_literal /
                           _vecs_to_pg_array (orm/repository/base.py:54-76): the text forms of VECTOR(d) / VECTOR(d)[].
  rerank_golden.npz        ColBERTReranker._maxsim_score (rerankers/colbert.py:63-84) on seeded padded token tensors.
  embeddings_golden.npz    every public method of the reference's ColPaliEmbeddings (embeddings/colpali.py:88-245) and
                           BiPaliEmbeddings (embeddings/bipali.py:95-255) over the stand-in `colpali_engine` module of
                           tests/helpers_hf.py (a seeded tiny ColPaliForRetrieval), float32 on the CPU.
  evaluation_golden.json   build_retrieval_gt_from_relations and RetrievalEvaluationService._get_execution_results
                           (orm/service/retrieval_evaluation.py:23-78, 161-217) over a fake Unit of Work: ranked-list order,
                           chunk / image-chunk ties, NULL scores, AND / OR ground truth.
  ingest_golden.json       BaseIngestionService._embed_entities (orm/service/base_ingestion.py:326-495) over a fake Unit of
                           Work: NULL-content image chunks skipped, failed items remembered and the rest embedded, single and
                           multi-vector columns.
  gqr_golden.npz / .json   Guided Query Refinement: outputs of GQRHybridRetrievalPipeline._optimize_query_embedding /
                           _optimize_query_multi_embedding / _optimize_in_score_space (pipelines/retrieval/
                           gqr_hybrid.py:306-362) on seeded pools, and of _retrieve_by_id (:472-489) over the fake
                           Unit-of-Work with recorded child-pipeline results (embedding, multi-vector and score-space
                           branches of _run_gqr, :415-470).
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


sys.path.insert(0, str(HERE))
from ref_import import import_reference  # noqa: E402

import_reference()

from autorag_research.evaluation.metrics import retrieval as ref_metrics  # noqa: E402
from autorag_research.evaluation.metrics.util import calculate_cosine_similarity  # noqa: E402
from autorag_research.orm.service.retrieval_pipeline import RetrievalPipelineService  # noqa: E402
from autorag_research.pipelines.retrieval.gqr_hybrid import (  # noqa: E402
    GQRHybridRetrievalPipeline,
    _cosine_scores,
    _maxsim_scores,
    _softmax,
)
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
                single = None if self.single is None else [float(x) for x in self.single[i]]
                out.append(_Obj(id=pk, contents=self.contents[i], embeddings=emb, embedding=single))
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


_LAST: dict = {}  # the fake service of make_service(), reused by make_gqr_flow


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
    _LAST.update(svc=svc, ids=ids)

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


# --------------------------------------------------------------------------------------
# 4. Guided Query Refinement
# --------------------------------------------------------------------------------------

GQR_PARAM_SETS = [  # (n_steps, learning_rate, temperature, mixture_alpha)
    (25, 0.1, 1.0, 0.5),   # the reference defaults (gqr_hybrid.py:137-140)
    (3, 0.5, 0.05, 1.0),   # sharp softmax, pure complementary target
    (40, 1.5, 0.3, 0.25),  # long run, large steps
]


def _gqr(n_steps, lr, temperature, alpha, **kw):
    p = GQRHybridRetrievalPipeline.__new__(GQRHybridRetrievalPipeline)
    p.n_steps, p.learning_rate, p.temperature, p.mixture_alpha = n_steps, lr, temperature, alpha
    p.fetch_k_multiplier, p.candidate_pool_mode, p.scorer_mode = 2, "union", "auto"
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def make_gqr_arrays() -> dict[str, np.ndarray]:
    rng = np.random.default_rng(31337)
    out: dict[str, np.ndarray] = {"params": np.array(GQR_PARAM_SETS, dtype=np.float64)}
    # ---- single-vector pools over a small corpus
    n, d, P = 80, 48, 40
    C = rng.standard_normal((n, d)).astype(np.float32)
    C[5] *= 25.0
    C[6] *= 1e-3
    C[7] = 0.0  # zero row: its norm is floored at eps
    sizes = [1, 7, 20, 40, 33, 12]
    pools = np.full((len(sizes), P), -1, dtype=np.int64)
    comp = np.zeros((len(sizes), P))
    Q = rng.standard_normal((len(sizes), d)).astype(np.float32).astype(np.float64)
    Q[4] = 0.0  # zero query: scores and gradients vanish
    for b, m in enumerate(sizes):
        pools[b, :m] = rng.choice(n, size=m, replace=False)
        comp[b, :m] = _softmax(rng.standard_normal(m) * 2.0, 1.0)
    pools[2, :3] = [5, 6, 7]
    out.update(single_C=C, single_pools=pools, single_comp=comp, single_Q=Q)
    exp = np.full((len(GQR_PARAM_SETS), len(sizes), P), np.nan)
    for s_i, prm in enumerate(GQR_PARAM_SETS):
        p = _gqr(*prm)
        for b, m in enumerate(sizes):
            exp[s_i, b, :m] = p._optimize_query_embedding(Q[b], C[pools[b, :m]].astype(np.float64), comp[b, :m])
    out["single_expected"] = exp
    # ---- multi-vector pools
    dm = 16
    lens = rng.integers(1, 70, size=50)
    tok = rng.standard_normal((int(lens.sum()), dm)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    docs = [tok[off[i]:off[i + 1]] for i in range(len(lens))]
    q_lens = [1, 3, 17, 32]
    qtok = rng.standard_normal((sum(q_lens), dm)).astype(np.float32)
    qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
    qoff = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    Pm = 24
    msizes = [24, 5, 13, 1]
    mpools = np.full((len(q_lens), Pm), -1, dtype=np.int64)
    mcomp = np.zeros((len(q_lens), Pm))
    for b, m in enumerate(msizes):
        mpools[b, :m] = rng.choice(len(lens), size=m, replace=False)
        mcomp[b, :m] = _softmax(rng.standard_normal(m), 0.5)
    out.update(multi_tok=tok, multi_off=off, multi_qtok=qtok.astype(np.float64), multi_qoff=qoff, multi_pools=mpools,
               multi_comp=mcomp)
    mexp = np.full((len(GQR_PARAM_SETS), len(q_lens), Pm), np.nan)
    for s_i, prm in enumerate(GQR_PARAM_SETS):
        p = _gqr(*prm)
        for b, m in enumerate(msizes):
            mexp[s_i, b, :m] = p._optimize_query_multi_embedding(
                qtok[qoff[b]:qoff[b + 1]].astype(np.float64), [docs[i].astype(np.float64) for i in mpools[b, :m]],
                mcomp[b, :m])
    out["multi_expected"] = mexp
    # ---- score-space form
    Ps = 30
    ssizes = [30, 1, 9, 18]
    prim = np.zeros((len(ssizes), Ps))
    scomp = np.zeros((len(ssizes), Ps))
    for b, m in enumerate(ssizes):
        prim[b, :m] = rng.standard_normal(m) * (10.0 if b == 3 else 1.0)  # BM25-sized raw scores in one row
        scomp[b, :m] = _softmax(rng.standard_normal(m), 1.0)
    out.update(score_primary=prim, score_comp=scomp, score_counts=np.array(ssizes, dtype=np.int32))
    sexp = np.full((len(GQR_PARAM_SETS), len(ssizes), Ps), np.nan)
    for s_i, prm in enumerate(GQR_PARAM_SETS):
        p = _gqr(*prm)
        for b, m in enumerate(ssizes):
            sexp[s_i, b, :m] = p._optimize_in_score_space(prim[b, :m], scomp[b, :m])
    out["score_expected"] = sexp
    return out


class _RecordedChild:
    """A child retrieval pipeline that answers from a recorded table (what the test replays on the MI355X side)."""

    def __init__(self, name, table, search_mode="single"):
        self.name, self._table, self.search_mode = name, table, search_mode
        self._embedding_model = None

    async def _retrieve_by_id(self, query_id, top_k):
        return [dict(r) for r in self._table[query_id][:top_k]]


def make_gqr_flow(svc: "_FakeService", chunk_ids: list) -> dict:
    """_retrieve_by_id of the reference pipeline over the fake UoW of make_service (same seeds), with a recorded
    lexical child as the complementary retriever (a BM25 stand-in: arbitrary positive scores, partly disjoint ids)."""
    rng = np.random.default_rng(99)
    k = 5
    loop = asyncio.new_event_loop()

    def primary(mode):
        p = VectorSearchRetrievalPipeline.__new__(VectorSearchRetrievalPipeline)
        p.search_mode, p._service, p._embedding_model, p.name = mode, svc, None, f"vs_{mode}"
        return p

    lexical = {}
    for qi in range(6):
        picks = [int(x) for x in rng.choice(len(chunk_ids), size=10, replace=False)]
        scores = np.sort(rng.gamma(2.0, 4.0, size=10))[::-1]
        lexical[f"q{qi}"] = [{"doc_id": chunk_ids[i], "score": float(s), "content": f"chunk text {i}"}
                             for i, s in zip(picks, scores)]
    # q5's lexical list names a chunk that does not exist: its embedding is missing -> score-space branch
    lexical["q5"][2]["doc_id"] = 999_999
    out = {"top_k": k, "lexical": lexical, "cases": []}
    cases = [
        ("single_union", "single", "union", "auto", (25, 0.1, 1.0, 0.5), ["q0", "q1", "q2"]),
        ("single_primary_pool", "single", "primary", "auto", (25, 0.1, 1.0, 0.5), ["q3"]),
        ("multi_union", "multi", "union", "auto", (10, 0.2, 0.5, 0.7), ["q1", "q4"]),
        ("multi_primary_forced_single", "multi", "union", "single", (25, 0.1, 1.0, 0.5), ["q2"]),
        ("score_space_fallback", "single", "union", "auto", (25, 0.1, 1.0, 0.5), ["q5"]),
    ]
    for name, mode, pool_mode, scorer, prm, qids in cases:
        p = _gqr(*prm, candidate_pool_mode=pool_mode, scorer_mode=scorer, _service=svc,
                 _primary_retrieval_pipeline=primary(mode),
                 _complementary_retrieval_pipeline=_RecordedChild("lexical", lexical))
        res = {qid: loop.run_until_complete(p._retrieve_by_id(qid, k)) for qid in qids}
        out["cases"].append({"name": name, "primary_search_mode": mode, "candidate_pool_mode": pool_mode,
                             "scorer_mode": scorer, "params": list(prm), "results": res})
    loop.close()
    return out


# --------------------------------------------------------------------------------------
# 5. hybrid fusion
# --------------------------------------------------------------------------------------


def make_hybrid(svc: "_FakeService", chunk_ids: list) -> dict:
    import autorag_research.pipelines.retrieval.hybrid as ref_hybrid

    rng = np.random.default_rng(2718)

    def ranked(ids, scores):
        order = np.argsort(-np.asarray(scores), kind="stable")
        return [{"doc_id": ids[i], "score": float(scores[i])} for i in order]

    lists = []
    for case in range(6):
        n1, n2 = int(rng.integers(1, 21)), int(rng.integers(1, 21))
        pool = [int(x) for x in rng.choice(500, size=40, replace=False)]
        a = ranked(pool[:n1], rng.standard_normal(n1))
        b = ranked([pool[i] for i in rng.choice(40, size=n2, replace=False)], rng.gamma(2.0, 3.0, size=n2))
        lists.append((a, b))
    # edge cases: one empty list, identical scores, fully disjoint, string ids
    lists.append(([], ranked([1, 2, 3], [3.0, 2.0, 1.0])))
    lists.append((ranked([1, 2, 3], [0.5, 0.5, 0.5]), ranked([2, 3, 4], [7.0, 7.0, 7.0])))
    lists.append((ranked([1, 2], [0.9, 0.1]), ranked([3, 4], [5.0, 4.0])))
    lists.append((ranked(["a", "b", "c"], [0.9, 0.8, 0.1]), ranked(["c", "d"], [12.0, 3.0])))
    out = {"lists": [{"results_1": a, "results_2": b} for a, b in lists], "rrf": [], "cc": []}
    for a, b in lists:
        out["rrf"].append({"k": 60, "top_k": 10, "fetch_k": 20, "expected": ref_hybrid._rrf_fuse(a, b, 60, 10, 20)})
        per = {}
        for method in ("mm", "tmm", "z", "dbsf"):
            for w in (0.5, 0.2):
                fused = ref_hybrid._cc_fuse(a, b, w, 10, method, -1.0 if method == "tmm" else None,
                                            0.0 if method == "tmm" else None)
                if any(isinstance(r["doc_id"], str) for r in a + b):
                    # The reference iterates a SET of the ids (hybrid.py:133): with string ids its iteration order -- hence
                    # the order of exact score ties and the last bits of the z / dbsf statistics -- changes from one
                    # interpreter to the next.  Freeze an order-independent image: scores on a 1e-12 grid, ties by id
                    # (tests compare with atol 1e-12 and accept any order inside a tie).
                    fused = sorted(({"doc_id": r["doc_id"], "score": round(r["score"], 12)} for r in fused),
                                   key=lambda r: (-r["score"], str(r["doc_id"])))
                per[f"{method}:{w}"] = fused
        out["cc"].append(per)
    # pipeline flows over the fake UoW: dense child = the reference's VectorSearchRetrievalPipeline, second child = recorded
    gq = json.loads((HERE / "gqr_golden.json").read_text())
    lexical = gq["lexical"]
    loop = asyncio.new_event_loop()
    dense = VectorSearchRetrievalPipeline.__new__(VectorSearchRetrievalPipeline)
    dense.search_mode, dense._service, dense._embedding_model, dense.name = "single", svc, None, "vs_single"
    flows = {}
    for name, cls, attrs in (
        ("rrf", ref_hybrid.HybridRRFRetrievalPipeline, {"rrf_k": 60}),
        ("cc_mm", ref_hybrid.HybridCCRetrievalPipeline, {"weight": 0.6, "normalize_method": "mm", "pipeline_1_min": None,
                                                         "pipeline_2_min": None}),
        ("cc_tmm", ref_hybrid.HybridCCRetrievalPipeline, {"weight": 0.5, "normalize_method": "tmm", "pipeline_1_min": -1.0,
                                                          "pipeline_2_min": 0.0}),
        ("cc_z", ref_hybrid.HybridCCRetrievalPipeline, {"weight": 0.3, "normalize_method": "z", "pipeline_1_min": None,
                                                        "pipeline_2_min": None}),
    ):
        p = cls.__new__(cls)
        p._retrieval_pipeline_1, p._retrieval_pipeline_2, p.fetch_k_multiplier = dense, _RecordedChild("lexical", lexical), 2
        for k_, v_ in attrs.items():
            setattr(p, k_, v_)
        flows[name] = {"attrs": attrs, "results": {q: loop.run_until_complete(p._retrieve_by_id(q, 5)) for q in ("q0", "q3")}}
    loop.close()
    out["flows"] = flows
    return out


# --------------------------------------------------------------------------------------
# 6. HyDE
# --------------------------------------------------------------------------------------


def hyde_fake_models(dim: int):
    """Deterministic stand-ins shared with the test: the "LLM" echoes the question inside a passage (as a chat message
    object), the "embedding" derives a vector from the text's bytes."""

    class FakeLLM:
        async def ainvoke(self, prompt):
            return _Obj(content="A passage about: " + prompt.split("Question: ")[1].split("\n")[0])

    class FakeEmbedding:
        @staticmethod
        def vec(text):
            seed = sum((i + 1) * b for i, b in enumerate(text.encode())) % (2**32)
            return [float(x) for x in np.random.default_rng(seed).standard_normal(dim).astype(np.float32)]

        async def aembed_query(self, text):
            return self.vec(text)

        def embed_documents(self, texts):
            return [self.vec(t) for t in texts]

    return FakeLLM(), FakeEmbedding()


def make_hyde(svc: "_FakeService") -> dict:
    from autorag_research.pipelines.retrieval.hyde import DEFAULT_HYDE_PROMPT_TEMPLATE, HyDERetrievalPipeline

    llm, emb = hyde_fake_models(32)
    p = HyDERetrievalPipeline.__new__(HyDERetrievalPipeline)
    p.llm, p.embedding, p.prompt_template, p._service = llm, emb, DEFAULT_HYDE_PROMPT_TEMPLATE, svc
    loop = asyncio.new_event_loop()
    out = {"prompt_template": DEFAULT_HYDE_PROMPT_TEMPLATE, "top_k": 6,
           "results": {q: loop.run_until_complete(p._retrieve_by_id(q, 6)) for q in ("q0", "q4")},
           "by_text": loop.run_until_complete(p._retrieve_by_text("what is late interaction?", 6))}
    loop.close()
    return out


# --------------------------------------------------------------------------------------
# 8. the reference's own plugin registry + Executor driving THIS plugin
# --------------------------------------------------------------------------------------


def make_executor() -> dict:
    """`plugin_registry._scan_module_yamls` (plugin_registry.py:90-120) over the installed package, then the reference
    Executor's health-check -> run -> verify flow (`_health_check_pipeline` executor.py:308-354, `_run_pipeline_with_retry`
    :383-463) with `Mi355VectorSearchPipelineConfig`, twice: the pipeline handed (a) a store factory and (b) a sessionmaker-like
    factory as the Executor gets from DBConnection, in which case the plugin goes through a RetrievalPipelineService (here the
    duck-typed one of tests/helpers.py -- no PostgreSQL in the build container).  No GPU here either: the index is the
    oracle-backed stand-in, so what is frozen is the HOST flow (stats dicts, PipelineResult, persisted rows)."""
    import dataclasses

    import autorag_research.executor as ref_executor
    import autorag_research.plugin_registry as ref_registry
    from autorag_research.config import ExecutorConfig

    sys.path.insert(0, str(HERE.parent))
    import helpers

    import autorag_research_amd as pkg
    import autorag_research_amd.compat as compat
    import autorag_research_amd.service as amd_service
    from autorag_research_amd.pipelines import Mi355VectorSearchPipelineConfig

    assert compat.HAVE_REFERENCE, "the plugin must subclass the reference's own config / pipeline bases here"
    amd_service.Mi355Index = helpers.OracleIndex
    out: dict = {}
    infos = ref_registry._scan_module_yamls(pkg, "mi355_vector_search", "pipelines")
    out["registry_scan"] = sorted([i.config_name, i.subcategory, i.category, i.plugin_name] for i in infos)

    class EvalService:  # what the Executor needs of RetrievalEvaluationService when no metric is configured
        def __init__(self, verify, delete_pipeline):
            self.verify_pipeline_completion, self._delete_pipeline = verify, delete_pipeline

        def _create_uow(self):
            outer = self

            class U:
                evaluation_results = type("R", (), {"delete_by_pipeline": staticmethod(lambda pid: 0)})()
                pipelines = type("P", (), {"delete_by_id": staticmethod(lambda pid: outer._delete_pipeline(pid))})()

                def __enter__(self):
                    return self

                def __exit__(self, *a):
                    return False

                def commit(self):
                    pass

            return U()

    def drive(session_factory, eval_service, rows_of):
        cfg = Mi355VectorSearchPipelineConfig(name="mi355_vector_search", search_mode="single", top_k=4, batch_size=4,
                                              retry_delay=0.0)
        e = ref_executor.Executor.__new__(ref_executor.Executor)
        e.session_factory, e._schema, e._config_dir = session_factory, None, None
        e.config = ExecutorConfig(pipelines=[cfg], metrics=[], max_retries=1, health_check_queries=2)
        e._retrieval_eval_service = eval_service
        e._health_check_pipeline(cfg)            # raises HealthCheckError unless 2 queries ran clean and were cleaned up
        after_health = len(rows_of())
        res = e._run_pipeline_with_retry(cfg)
        rows = sorted(rows_of(), key=lambda r: (str(r[0]), -r[2], str(r[1])))
        return {"rows_left_by_health_check": after_health, "pipeline_result": dataclasses.asdict(res) | {
            "pipeline_type": res.pipeline_type.value}, "persisted": [[q, c, s] for q, c, s in rows]}

    # (a) store factory
    store, _ = helpers.build_golden_stores()
    del store.queries["q_noemb"]                 # (the health check demands failed_queries == [])
    store.query_order.remove("q_noemb")

    def store_rows():
        return [(q, c, s) for (pid, q), lst in store.chunk_results.items() for c, s in lst]

    def store_verify(pid):
        return all(store.chunk_results.get((pid, q)) for q in store.query_order)

    out["store_factory"] = drive(lambda: store, EvalService(store_verify, store.delete_pipeline), store_rows)
    # (b) sessionmaker-like factory -> RetrievalPipelineService -> UowStore
    store2, _ = helpers.build_golden_stores()
    del store2.queries["q_noemb"]
    store2.query_order.remove("q_noemb")
    tables = helpers.ref_tables_from_store(store2)
    import autorag_research.orm.service.retrieval_pipeline as ref_rps

    real_service = ref_rps.RetrievalPipelineService
    ref_rps.RetrievalPipelineService = lambda sf, schema=None: helpers.FakeRefService(sf, schema)
    try:
        fake = helpers.FakeRefService(tables=tables)
        out["sessionmaker"] = drive(helpers.FakeSessionmaker(tables),
                                    EvalService(fake.verify_pipeline_completion, lambda pid: tables["pipelines"].pop(pid, None)),
                                    lambda: [(r["query_id"], r["chunk_id"], r["rel_score"]) for r in tables["chunk_results"]])
    finally:
        ref_rps.RetrievalPipelineService = real_service
    assert out["sessionmaker"]["persisted"] == out["store_factory"]["persisted"]
    return out


# --------------------------------------------------------------------------------------
# 9. ColBERT reranker MaxSim (rerankers/colbert.py:63-84)
# --------------------------------------------------------------------------------------


def make_rerank() -> dict:
    """`ColBERTReranker._maxsim_score` (the method, unbound -- no checkpoint needed) on seeded L2-normalised token tensors:
    ragged documents behind padding masks, padded query tokens, a document with no valid token, negative maxima (clamp)."""
    import torch

    from autorag_research.rerankers.colbert import ColBERTReranker

    g = torch.Generator().manual_seed(20260928)
    out = {}
    for case, (d, lq, n_valid_q, doc_lens, ld) in enumerate([(16, 9, 7, [5, 12, 1, 0, 9, 12], 12), (128, 32, 32, [40, 3, 64, 17], 64),
                                                             (24, 4, 2, [6, 6], 6)]):
        q = torch.nn.functional.normalize(torch.randn((1, lq, d), generator=g), dim=-1)
        qm = torch.zeros((1, lq), dtype=torch.long)
        qm[0, :n_valid_q] = 1
        docs = torch.nn.functional.normalize(torch.randn((len(doc_lens), ld, d), generator=g), dim=-1)
        if case == 0:
            docs[4] = -q[0, :1].expand(ld, d) * 0.5 + 0.01 * docs[4]   # every similarity negative -> clamp(min=0) matters
            docs[4] = torch.nn.functional.normalize(docs[4], dim=-1)
        dm = torch.zeros((len(doc_lens), ld), dtype=torch.long)
        for i, n in enumerate(doc_lens):
            dm[i, :n] = 1
        sc = [ColBERTReranker._maxsim_score(None, q, qm, docs[i:i + 1], dm[i:i + 1]) for i in range(len(doc_lens))]
        out[f"q{case}"], out[f"qm{case}"] = q.numpy(), qm.numpy()
        out[f"d{case}"], out[f"dm{case}"] = docs.numpy(), dm.numpy()
        out[f"score{case}"] = np.asarray(sc, dtype=np.float64)
    return out


# --------------------------------------------------------------------------------------
# 10. PostgreSQL text formats of the vector columns (orm/types.py, orm/repository/base.py)
# --------------------------------------------------------------------------------------


def make_pgtext() -> dict:
    from autorag_research.orm.repository.base import _vec_to_pg_literal, _vecs_to_pg_array
    from autorag_research.orm.types import VectorArray

    rng = np.random.default_rng(1618)
    va = VectorArray(4)
    cases = []
    mats = [rng.standard_normal((3, 4)).astype(np.float32), (rng.standard_normal((1, 4)) * 1e-6).astype(np.float32),
            np.array([[1.0, -2.5, 0.0, 1e20], [3.0, 4.0, -0.0, 5e-324]], dtype=np.float64), np.zeros((0, 4), np.float32)]
    for m in mats:
        as_list = [[float(x) for x in row] for row in m]
        bind = va.process_bind_param(as_list, None)
        cases.append({"values": as_list, "bind": bind, "parsed": va.process_result_value(bind, None),
                      "sql_array": _vecs_to_pg_array(as_list) if as_list else None,
                      "literals": [_vec_to_pg_literal(r) for r in as_list]})
    hand = ['{"[1,2,3,4]","[5, 6 ,7,8]"}', '{"[0.5,-1e-05,3.25E+2,7]"}', "{}"]
    return {"cases": cases, "null_bind": va.process_bind_param(None, None), "null_result": va.process_result_value(None, None),
            "hand_strings": [{"text": h, "parsed": va.process_result_value(h, None)} for h in hand],
            "preparsed": va.process_result_value([[1, 2], np.array([3.5, 4.5])], None)}


# --------------------------------------------------------------------------------------
# 11. ColPali / BiPali embedding wrappers (embeddings/colpali.py:88-245, embeddings/bipali.py:53-255)
# --------------------------------------------------------------------------------------


def make_embeddings() -> dict[str, np.ndarray]:
    """The REFERENCE's ColPaliEmbeddings / BiPaliEmbeddings run over the stand-in `colpali_engine` module of
    tests/helpers_hf.py (transformers' own ColPaliForRetrieval from a small seeded config behind colpali_engine's call shape;
    the dependency itself is not installed) in float32 on the CPU: every public embedding method, frozen as arrays.
    Ragged outputs are stored flat with offsets.  BiPaliEmbeddings' constructor forwards its keyword arguments to
    `langchain_core.embeddings.Embeddings.__init__` (bipali.py:91-93) -- a plain ABC in langchain-core, stubbed here --, so the
    object is allocated with __new__ and `_load_model()` (bipali.py:95-111) is called on it: the loading and embedding code under
    test is the reference's."""
    sys.path.insert(0, str(REPO / "tests"))
    import helpers_hf as hf
    import torch

    from autorag_research.embeddings.bipali import BiPaliEmbeddings
    from autorag_research.embeddings.colpali import ColPaliEmbeddings

    torch.set_num_threads(1)
    seen: dict = {}
    hf.install_colpali_engine(seen=seen)
    out: dict[str, np.ndarray] = {}

    def ragged(name: str, docs: list) -> None:
        lens = [len(d) for d in docs]
        out[name] = np.asarray([v for d in docs for v in d], dtype=np.float32).reshape(sum(lens), -1)
        out[name + "_off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    texts = hf.EMBED_TEXTS
    imgs = hf.images(3)
    pngs = [hf.png_bytes(a) for a in imgs]
    for i, a in enumerate(imgs):
        out[f"image{i}"] = a
    col = ColPaliEmbeddings(model_name="tiny/colpali", model_type="pali", device="cpu", torch_dtype="float32")
    assert seen["name"] == "tiny/colpali" and seen["dtype"] == torch.float32 and seen["trust_remote_code"] is True
    out["col_embed_batch_size"] = np.asarray(col.embed_batch_size)
    ragged("col_embed_text", [col.embed_text(t) for t in texts[:4]])
    ragged("col_embed_query", [col.embed_query(t) for t in texts[:4]])
    ragged("col_aembed_query", [asyncio.run(col.aembed_query(texts[0]))])
    ragged("col_embed_documents", col.embed_documents(texts))            # ONE padded batch of 12 (colpali.py:189-216)
    ragged("col_embed_documents_batch", col.embed_documents_batch(texts))  # batches of embed_batch_size = 10 (base.py:77-83)
    ragged("col_embed_image", [col.embed_image(pngs[0])])
    ragged("col_embed_images", col.embed_images(pngs))
    assert col.embed_documents([]) == [] and col.embed_images([]) == []
    bi = BiPaliEmbeddings.__new__(BiPaliEmbeddings)
    bi.model_name, bi.model_type, bi.device, bi.torch_dtype, bi.embed_batch_size = "tiny/bipali", "pali", "cpu", "float32", 5
    bi._load_model()
    out["bi_embed_query"] = np.asarray([bi.embed_query(t) for t in texts[:4]], dtype=np.float32)
    out["bi_embed_documents"] = np.asarray(bi.embed_documents(texts), dtype=np.float32)   # batches of 5 (bipali.py:236-255)
    out["bi_embed_image"] = np.asarray(bi.embed_image(pngs[1]), dtype=np.float32)
    out["bi_embed_images"] = np.asarray(bi.embed_images(pngs), dtype=np.float32)
    out["bi_aembed_image"] = np.asarray(asyncio.run(bi.aembed_image(pngs[1])), dtype=np.float32)
    return out


# --------------------------------------------------------------------------------------
# 12. evaluation ordering + ground truth (orm/service/retrieval_evaluation.py:23-78, 161-217)
# --------------------------------------------------------------------------------------


def make_evaluation() -> dict:
    """`build_retrieval_gt_from_relations` on relation rows (group_index = AND, group_order = OR order, both id kinds, NULL
    scores, rows with neither id, an id listed in two groups), and `RetrievalEvaluationService._get_execution_results` over a
    fake Unit of Work whose result repositories answer like the reference's (`ORDER BY rel_score DESC`,
    orm/repository/chunk_retrieved_result.py:34-38): chunk and image-chunk rows of one query merged by the stable Python
    re-sort (ties: chunk rows first), NULL rel_score as 0.0, a query with no rows, a query with no ground truth."""
    from autorag_research.orm.service.retrieval_evaluation import (
        RetrievalEvaluationService,
        build_retrieval_gt_from_relations,
    )

    rel_cases = [
        [dict(group_index=0, group_order=0, chunk_id=1, image_chunk_id=None, score=2),
         dict(group_index=0, group_order=1, chunk_id=None, image_chunk_id=2, score=1),
         dict(group_index=1, group_order=0, chunk_id=3, image_chunk_id=None, score=None)],
        [dict(group_index=2, group_order=1, chunk_id="b", image_chunk_id=None, score=0),
         dict(group_index=2, group_order=0, chunk_id="a", image_chunk_id=None, score=3),
         dict(group_index=0, group_order=5, chunk_id=None, image_chunk_id="p9", score=None),
         dict(group_index=0, group_order=2, chunk_id="a", image_chunk_id=None, score=1),   # same id again: later score wins
         dict(group_index=1, group_order=0, chunk_id=None, image_chunk_id=None, score=2)],  # neither id: dropped
        [],
        [dict(group_index=0, group_order=0, chunk_id=7, image_chunk_id=8, score=None)],     # both ids: the chunk id is taken
        [dict(group_index=5, group_order=1, chunk_id=10, image_chunk_id=None, score=1),
         dict(group_index=5, group_order=1, chunk_id=11, image_chunk_id=None, score=1),     # equal group_order: input order kept
         dict(group_index=5, group_order=0, chunk_id=12, image_chunk_id=None, score=2)],
    ]
    out: dict = {"relations": []}
    for rows in rel_cases:
        gt, rel = build_retrieval_gt_from_relations([_Obj(**r) for r in rows])
        out["relations"].append({"rows": rows, "retrieval_gt": gt, "relevance_scores": rel})

    pid = 3
    chunk_rows = [  # (query_id, chunk_id, rel_score) in INSERTION order
        ("q1", 11, 0.9), ("q1", 12, 0.5), ("q1", 13, 0.7), ("q1", 14, 0.5), ("q1", 15, None),
        ("q2", 21, 0.25), ("q2", 22, 0.75),
        ("q4", 41, -0.5), ("q4", 42, 0.0), ("q4", 43, None),
        ("q9", 91, 1.0),                        # a query nobody asks for
    ]
    image_rows = [("q1", "img-a", 0.7), ("q1", "img-b", 0.95), ("q1", "img-c", 0.5),
                  ("q3", "img-d", 0.1), ("q3", "img-e", 0.1), ("q4", "img-f", 0.0)]
    other_pipeline_rows = [("q1", 99, 5.0)]
    relations = {"q1": rel_cases[0], "q2": rel_cases[4], "q3": rel_cases[1], "q4": [], "q5": rel_cases[3]}

    class _Results:
        def __init__(self, rows, key):
            self.rows, self.key = rows, key

        def get_by_query_and_pipeline(self, query_ids, pipeline_id):
            qs = set(query_ids)
            sel = [r for r in self.rows if r[0] == pipeline_id and r[1] in qs]
            # ORDER BY rel_score DESC (PostgreSQL: NULLs first on DESC); ties in insertion order
            sel.sort(key=lambda r: (r[3] is None, r[3] if r[3] is not None else 0.0), reverse=True)
            return [_Obj(query_id=r[1], pipeline_id=r[0], rel_score=r[3], **{self.key: r[2]}) for r in sel]

    class _Relations:
        def get_by_query_id(self, qid):
            return [_Obj(**r) for r in relations.get(qid, [])]

    class _Uow:
        chunk_results = _Results([(pid, q, c, s) for q, c, s in chunk_rows] + [(pid + 1, q, c, s) for q, c, s in other_pipeline_rows],
                                 "chunk_id")
        image_chunk_results = _Results([(pid, q, c, s) for q, c, s in image_rows], "image_chunk_id")
        retrieval_relations = _Relations()

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    svc = RetrievalEvaluationService.__new__(RetrievalEvaluationService)
    svc._create_uow = lambda: _Uow()
    qids = ["q1", "q2", "q3", "q4", "q5"]
    res = svc._get_execution_results(pid, qids)
    out["execution"] = {"pipeline_id": pid, "query_ids": qids, "chunk_rows": [list(r) for r in chunk_rows],
                        "image_chunk_rows": [list(r) for r in image_rows],
                        "other_pipeline_rows": [list(r) for r in other_pipeline_rows], "relations": relations,
                        "results": {q: res[q] for q in qids}}
    return out


# --------------------------------------------------------------------------------------
# 13. ingestion: BaseIngestionService._embed_entities (orm/service/base_ingestion.py:326-495)
# --------------------------------------------------------------------------------------


def ingest_vector(data, dim: int = 6) -> list[float]:
    """Deterministic stand-in embedding of a text / image payload (shared with the test through the fixture: the expected
    vectors are stored, this function only has to be a function)."""
    b = data if isinstance(data, bytes) else str(data).encode()
    h = np.frombuffer(__import__("hashlib").sha256(b).digest()[: 4 * dim], dtype=np.uint32).astype(np.float64)
    return [float(x) for x in (h / 2.0**32 - 0.5)]


def make_ingest() -> dict:
    """The reference's `_embed_entities` driven over a fake Unit of Work (repositories answering `count_without_*`,
    `get_without_*(limit, excluded_ids)`, `get_by_id`, `set_multi_vector_embeddings_batch` like orm/repository/base.py:459-485,
    587-689): rows that already have embeddings, an image chunk with NULL content (skipped and counted, :44,400-406), items
    whose embedding function raises or returns None (remembered for the run, not retried, the others still embedded,
    :59-71,461-495), single- and multi-vector columns, batches smaller than the table.  Frozen: the return value, every row's
    final embedding, which payloads the embedding function was called with and in which order per batch."""
    import logging

    from autorag_research.orm.service.base_ingestion import BaseIngestionService

    logging.getLogger("AutoRAG-Research").setLevel(logging.CRITICAL)
    out: dict = {"cases": []}

    def run(entity, emb_type, rows, batch_size, bad_raise=(), bad_none=(), bm25=None, bm25_raises=False):
        table = [_Obj(id=r["id"], contents=r["contents"], embedding=r.get("embedding"), embeddings=r.get("embeddings")) for r in rows]
        col = "embedding" if emb_type == "single" else "embeddings"
        calls: list = []
        fetches: list = []
        bm25_calls: list = []

        class _Repo:
            session = object()

            def _missing(self):
                return [e for e in table if getattr(e, col) is None]

            def count_without_embeddings(self):
                return len(self._missing())

            count_without_multi_embeddings = count_without_embeddings

            def get_without_embeddings(self, limit=None, offset=None, excluded_ids=None):
                rows_ = [e for e in self._missing() if not excluded_ids or e.id not in excluded_ids]
                fetches.append([e.id for e in rows_[: limit]])
                return rows_[: limit]

            get_without_multi_embeddings = get_without_embeddings

            def get_by_id(self, pk):
                return next((e for e in table if e.id == pk), None)

            def set_multi_vector_embeddings_batch(self, entity_ids, embeddings_list, vector_column="embeddings", id_column="id"):
                n = 0
                for pk, emb in zip(entity_ids, embeddings_list, strict=True):
                    e = self.get_by_id(pk)
                    if e is not None:
                        setattr(e, vector_column, emb)
                        n += 1
                return n

            def batch_update_bm25_tokens(self, tokenizer="bert", batch_size=1000):   # orm/repository/chunk.py, query.py
                bm25_calls.append({"tokenizer": tokenizer, "batch_size": batch_size,
                                   "rows_embedded_at_call": sum(1 for e in table if getattr(e, col) is not None)})
                if bm25_raises:
                    raise RuntimeError("function tokenize(text, unknown) does not exist")
                return len(table)

        repo = _Repo()

        class _Uow:
            session = object()
            queries = chunks = image_chunks = repo

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

            def commit(self):
                pass

        async def embed(data):
            key = data.decode() if isinstance(data, bytes) else data
            calls.append(key)
            if key in bad_raise:
                raise RuntimeError(f"cannot embed {key}")
            if key in bad_none:
                return None
            v = ingest_vector(data)
            return v if emb_type == "single" else [v, [x * 0.5 for x in v]]

        class _Svc(BaseIngestionService):   # the two abstract hooks of BaseService; nothing of the code under test
            def _create_uow(self):
                return _Uow()

            def _get_schema_classes(self):
                return {}

        svc = _Svc.__new__(_Svc)
        n = svc._embed_entities(entity, emb_type, embed, batch_size=batch_size, max_concurrency=3, bm25_tokenizer=bm25)
        out["cases"].append({
            "entity_type": entity, "embedding_type": emb_type, "batch_size": batch_size,
            "bm25_tokenizer": bm25, "bm25_raises": bm25_raises, "bm25_calls": bm25_calls,
            "rows": [{k: (v.decode() if isinstance(v, bytes) else v) for k, v in r.items()} | {"bytes": isinstance(r["contents"], bytes)}
                     for r in rows],
            "bad_raise": list(bad_raise), "bad_none": list(bad_none), "returned": n, "embed_calls": sorted(calls),
            "n_embed_calls": len(calls), "fetches": fetches,
            "final": [{"id": e.id, col: getattr(e, col)} for e in table]})

    pre = ingest_vector("already there")
    run("chunk", "single",
        [{"id": 1, "contents": "alpha"}, {"id": 2, "contents": "beta", "embedding": pre}, {"id": 3, "contents": "RAISE"},
         {"id": 4, "contents": "gamma"}, {"id": 5, "contents": "NONE"}, {"id": 6, "contents": "delta"}, {"id": 7, "contents": "epsilon"}],
        batch_size=3, bad_raise=("RAISE",), bad_none=("NONE",))
    run("query", "multi_vector",
        [{"id": "q1", "contents": "what is late interaction"}, {"id": "q2", "contents": "RAISE"},
         {"id": "q3", "contents": "second query", "embeddings": [pre]}, {"id": "q4", "contents": "third"}],
        batch_size=2, bad_raise=("RAISE",))
    run("image_chunk", "single",
        [{"id": "i1", "contents": b"png-bytes-1"}, {"id": "i2", "contents": None}, {"id": "i3", "contents": b"png-bytes-3"},
         {"id": "i4", "contents": None}, {"id": "i5", "contents": b"NONE"}, {"id": "i6", "contents": b"png-bytes-6"}],
        batch_size=2, bad_none=("NONE",))
    run("image_chunk", "multi_vector",
        [{"id": 10, "contents": None}, {"id": 11, "contents": None}, {"id": 12, "contents": b"page-12"}],
        batch_size=2)
    run("chunk", "single", [{"id": 1, "contents": "RAISE"}, {"id": 2, "contents": "NONE"}], batch_size=8,
        bad_raise=("RAISE",), bad_none=("NONE",))                      # nothing can be embedded: the loop still ends
    run("chunk", "multi_vector", [{"id": 1, "contents": "x", "embeddings": [pre]}], batch_size=4)   # nothing to do: returns 0
    # bm25_tokens (base_ingestion.py:429-430, 497-540): one repository call behind the loop for chunk / query rows, none for
    # image chunks, a failing call (extension not installed) swallowed; and -- what the early return at :383-386 means -- NOT
    # made when nothing lacked an embedding
    run("chunk", "single", [{"id": 1, "contents": "alpha"}, {"id": 2, "contents": "NONE"}, {"id": 3, "contents": "gamma"}],
        batch_size=2, bad_none=("NONE",), bm25="bert")
    run("query", "single", [{"id": "q1", "contents": "one"}, {"id": "q2", "contents": "two"}], batch_size=8, bm25="wiki_tocken",
        bm25_raises=True)
    run("image_chunk", "single", [{"id": "i1", "contents": b"png-bytes-1"}], batch_size=2, bm25="bert")
    run("chunk", "single", [{"id": 1, "contents": "x", "embedding": pre}], batch_size=4, bm25="bert")
    return out


def main() -> None:
    (HERE / "metrics_golden.json").write_text(json.dumps(make_metrics(), indent=1))
    np.savez_compressed(HERE / "scores_golden.npz", **make_scores())
    (HERE / "service_golden.json").write_text(json.dumps(make_service(), indent=1))
    np.savez_compressed(HERE / "gqr_golden.npz", **make_gqr_arrays())
    (HERE / "gqr_golden.json").write_text(json.dumps(make_gqr_flow(_LAST["svc"], _LAST["ids"]), indent=1))
    (HERE / "hybrid_golden.json").write_text(json.dumps(make_hybrid(_LAST["svc"], _LAST["ids"]), indent=1))
    (HERE / "hyde_golden.json").write_text(json.dumps(make_hyde(_LAST["svc"]), indent=1))
    (HERE / "executor_golden.json").write_text(json.dumps(make_executor(), indent=1))
    np.savez_compressed(HERE / "rerank_golden.npz", **make_rerank())
    (HERE / "pgtext_golden.json").write_text(json.dumps(make_pgtext(), indent=1))
    np.savez_compressed(HERE / "embeddings_golden.npz", **make_embeddings())
    (HERE / "evaluation_golden.json").write_text(json.dumps(make_evaluation(), indent=1))
    (HERE / "ingest_golden.json").write_text(json.dumps(make_ingest(), indent=1))
    print("wrote", sorted(p.name for p in HERE.iterdir()))


if __name__ == "__main__":
    main()
