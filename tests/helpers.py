"""Shared test helpers: oracle-backed stand-in for the GPU index (CPU tests), golden-input regeneration."""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


class OracleIndex:
    """Same surface as autorag_research_amd.index.Mi355Index for what service.py uses, answered by the CPU oracle.

    Test infrastructure: lets the host logic (service / pipelines / sharding) run under `-m "not gpu"`.
    """

    def __init__(self, dim: int, metric: str = "cosine", device: int = 0):
        from oracle import cpu_ref

        self._o = cpu_ref
        self.dim, self.metric, self.device = dim, metric, device
        self._rows = np.zeros((0, dim), np.float32)
        self._tok = None
        self._off = None
        self.row_offset = 0

    def add(self, rows):
        self._rows = np.concatenate([self._rows, np.ascontiguousarray(rows, dtype=np.float32)], axis=0)

    def __len__(self):
        return self._rows.shape[0]

    def search(self, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        d, r = self._o.topk_search(self._rows, q, k, metric=self.metric)
        r = np.where(r >= 0, r + self.row_offset, r)
        return d, r

    def add_multivec(self, vecs, offsets):
        self._tok = np.ascontiguousarray(vecs, dtype=np.float32)
        self._off = np.ascontiguousarray(offsets, dtype=np.int64)

    def n_docs(self):
        return 0 if self._off is None else self._off.shape[0] - 1

    def search_maxsim(self, qtok, q_offsets, k):
        d, r = self._o.maxsim_topk(self._tok, self._off, qtok, q_offsets, k)
        return d, np.where(r >= 0, r + self.row_offset, r)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass

    def maxsim_subset(self, qtok, q_offsets, doc_ids, clamp0=False):
        q = np.ascontiguousarray(qtok, dtype=np.float32).reshape(-1, self.dim)
        ids = np.asarray(doc_ids, dtype=np.int64)
        out = np.full(ids.shape, np.nan, dtype=np.float32)
        n_docs = self._off.shape[0] - 1
        for b in range(ids.shape[0]):
            qb = q[q_offsets[b]:q_offsets[b + 1]]
            for j, i in enumerate(ids[b] - self.row_offset):   # ids are GLOBAL rows, like the search results (mi355dr.h)
                if 0 <= i < n_docs and self._off[i + 1] > self._off[i] and qb.shape[0]:
                    doc = self._tok[self._off[i]:self._off[i + 1]]
                    if clamp0:  # ColBERT reranker form: every query token contributes max(0, max_j <q_i, d_j>)
                        out[b, j] = -np.maximum((qb @ doc.T).max(axis=1), np.float32(0)).sum(dtype=np.float32)
                    else:
                        out[b, j] = self._o.maxsim_distance(doc, qb)
        return out

    # ---- GQR refinement: same surface as Mi355Index.gqr_refine*, answered by oracle/gqr_ref.py ----
    def gqr_refine(self, queries, cand_rows, comp_dist, n_steps, learning_rate, temperature, mixture_alpha):
        from oracle import gqr_ref

        pools = np.asarray(cand_rows, dtype=np.int64)
        out = np.full(pools.shape, np.nan)
        C = self._rows.astype(np.float64)
        for b in range(pools.shape[0]):
            m = int((pools[b] >= 0).sum())
            out[b, :m] = gqr_ref.refine_single(np.asarray(queries, dtype=np.float64)[b], C[pools[b, :m] - self.row_offset],
                                               np.asarray(comp_dist)[b, :m], n_steps, learning_rate, temperature,
                                               mixture_alpha)
        return out

    def gqr_refine_maxsim(self, qtok, q_offsets, doc_ids, comp_dist, n_steps, learning_rate, temperature, mixture_alpha):
        from oracle import gqr_ref

        pools = np.asarray(doc_ids, dtype=np.int64)
        out = np.full(pools.shape, np.nan)
        q = np.asarray(qtok, dtype=np.float64).reshape(-1, self.dim)
        for b in range(pools.shape[0]):
            m = int((pools[b] >= 0).sum())
            docs = [self._tok[self._off[i]:self._off[i + 1]] for i in pools[b, :m] - self.row_offset]
            out[b, :m] = gqr_ref.refine_multi(q[q_offsets[b]:q_offsets[b + 1]], docs, np.asarray(comp_dist)[b, :m],
                                              n_steps, learning_rate, temperature, mixture_alpha)
        return out

    def gqr_refine_scores(self, primary_scores, counts, comp_dist, n_steps, learning_rate, temperature, mixture_alpha):
        from oracle import gqr_ref

        z = np.asarray(primary_scores, dtype=np.float64)
        out = np.full(z.shape, np.nan)
        for b, m in enumerate(counts):
            out[b, :m] = gqr_ref.refine_scores(z[b, :m], np.asarray(comp_dist)[b, :m], n_steps, learning_rate,
                                               temperature, mixture_alpha)
        return out

    def set_option(self, key, value):
        if key == "row_offset":
            self.row_offset = int(value)

    def close(self):
        pass


def service_golden_inputs():
    """Regenerate the arrays tests/golden/make_golden.py:make_service() used (same rng call order)."""
    rng = np.random.default_rng(4242)
    n, d = 300, 32
    C = rng.standard_normal((n, d)).astype(np.float32)
    ids = [int(1000 + 3 * i) for i in range(n)]
    contents = [f"chunk text {i}" for i in range(n)]
    dm = 8
    lens = rng.integers(2, 9, size=n)
    tok = rng.standard_normal((int(lens.sum()), dm)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    Q = rng.standard_normal((6, d)).astype(np.float32)
    Qm = []
    for t in (1, 4, 5, 2, 3, 6):
        m = rng.standard_normal((t, dm)).astype(np.float32)
        m /= np.linalg.norm(m, axis=1, keepdims=True)
        Qm.append(m)
    multivec = [tok[offsets[i]:offsets[i + 1]] for i in range(n)]
    return dict(C=C, ids=ids, contents=contents, tok=tok, offsets=offsets, multivec=multivec, Q=Q, Qm=Qm,
                img_ids=[f"img-{i}" for i in range(n)])


def load_gqr_golden():
    return np.load(GOLDEN / "gqr_golden.npz"), json.loads((GOLDEN / "gqr_golden.json").read_text())


class RecordedChild:
    """A child retrieval pipeline answering from a recorded table (the lexical retriever of gqr_golden.json)."""

    retrieval_unit = "chunk"

    def __init__(self, name, table, search_mode="single"):
        self.name, self._table, self.search_mode, self._embedding_model = name, table, search_mode, None

    async def _retrieve_by_id(self, query_id, top_k):
        return [dict(r) for r in self._table[query_id][:top_k]]


def check_gqr_flow(store, make_primary, atol=1e-9):
    """Replay every case of gqr_golden.json through Mi355GQRHybridRetrievalPipeline (per query AND as one block) and
    compare with the reference's _retrieve_by_id output: same ids in the same order, scores within atol."""
    import asyncio

    from autorag_research_amd.gqr import Mi355GQRHybridRetrievalPipeline

    _, flow = load_gqr_golden()
    k = flow["top_k"]
    for case in flow["cases"]:
        n_steps, lr, temp, alpha = case["params"]
        p = Mi355GQRHybridRetrievalPipeline(
            lambda: store, f"gqr_{case['name']}", make_primary(case["primary_search_mode"]),
            RecordedChild("lexical", flow["lexical"]), n_steps=int(n_steps), learning_rate=lr, temperature=temp,
            mixture_alpha=alpha, candidate_pool_mode=case["candidate_pool_mode"], scorer_mode=case["scorer_mode"])
        qids = list(case["results"])
        per_query = [asyncio.run(p._retrieve_by_id(q, k)) for q in qids]
        block = p._retrieve_block(qids, k)
        for qid, got, got_b in zip(qids, per_query, block):
            exp = case["results"][qid]
            for res in (got, got_b):
                assert [r["doc_id"] for r in res] == [r["doc_id"] for r in exp], (case["name"], qid)
                assert np.allclose([r["score"] for r in res], [r["score"] for r in exp], rtol=0, atol=atol), (case["name"], qid)
        p.close()


def same_ranking(got: list, exp: list, atol: float = 1e-12) -> None:
    """Same scores position by position; same ids wherever the score is unique (the reference breaks exact ties by the
    iteration order of a Python set)."""
    assert len(got) == len(exp)
    gs, es = [r["score"] for r in got], [r["score"] for r in exp]
    assert np.allclose(gs, es, rtol=0, atol=atol)
    for i, (g, e) in enumerate(zip(got, exp)):
        tied = sum(abs(x - e["score"]) <= atol for x in es) > 1
        if not tied:
            assert g["doc_id"] == e["doc_id"], (i, g, e)
    tie_ids = lambda rs: sorted(str(r["doc_id"]) for r in rs if sum(abs(x - r["score"]) <= atol for x in es) > 1)  # noqa: E731
    if len(got) < 10:  # a tie group cut by top_k may legitimately keep different members
        assert tie_ids(got) == tie_ids(exp)


def load_service_golden():
    return json.loads((GOLDEN / "service_golden.json").read_text())


def build_golden_stores():
    """Two stores like the reference's two databases: single-vector (dim 32) and multi-vector (dim 8)."""
    from autorag_research_amd.store import InMemoryStore

    g = service_golden_inputs()
    s = InMemoryStore()
    s.set_chunks(g["ids"], g["contents"], embedding=g["C"], multivec=g["multivec"])
    s.set_image_chunks(g["img_ids"], embedding=g["C"], multivec=g["multivec"])
    qids = [f"q{i}" for i in range(6)]
    s.add_queries(qids, contents=[f"query text {i}" for i in range(6)], embedding=list(g["Q"]), embeddings=g["Qm"])
    s.add_queries(["q_noemb"], contents=["no embedding"], embedding=[None], embeddings=[None])
    return s, g


# ---- a stand-in for the REFERENCE's RetrievalPipelineService / Unit of Work (only its public shape) --------------------
class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FakeRefService:
    """Duck-typed `autorag_research.orm.service.retrieval_pipeline.RetrievalPipelineService` over plain Python tables:
    `_create_uow()` -> context manager with the repositories the Vector Search path touches (names and call shapes as in
    retrieval_pipeline.py:255-288, 359-372 and orm/repository/*.py).  Vectors come back as Python lists / None (NULL),
    like SQLAlchemy hands them out.  Test infrastructure for store.UowStore."""

    def __init__(self, session_factory=None, schema=None, *, tables=None):
        self.session_factory, self.schema = session_factory, schema
        self.t = tables if tables is not None else getattr(session_factory, "tables")
        self.uow_opened = 0

    class _Uow:
        def __init__(self, svc):
            t = svc.t

            class Queries:
                def get_by_id(self, qid):
                    return next((q for q in t["queries"] if q.id == qid), None)

                def get_all(self, limit=None, offset=None):
                    return t["queries"][offset or 0:(offset or 0) + (limit if limit is not None else len(t["queries"]))]

                def find_by_contents(self, text):
                    return next((q for q in t["queries"] if q.contents == text), None)

            class Chunks:
                def __init__(self, rows):
                    self.rows = rows

                def get_all(self, limit=None, offset=None):
                    return self.rows[offset or 0:(offset or 0) + (limit if limit is not None else len(self.rows))]

            class Results:
                def __init__(self, rows, key):
                    self.rows, self.key = rows, key

                def get_by_query_and_pipeline(self, query_ids, pipeline_id):
                    qs = set(query_ids)
                    return [_Obj(**r) for r in self.rows if r["pipeline_id"] == pipeline_id and r["query_id"] in qs]

                def bulk_insert(self, rows):
                    for r in rows:
                        assert set(r) == {"query_id", "pipeline_id", self.key, "rel_score"}, r
                    self.rows.extend(dict(r) for r in rows)

                def delete_by_pipeline(self, pipeline_id):
                    n = len(self.rows)
                    self.rows[:] = [r for r in self.rows if r["pipeline_id"] != pipeline_id]
                    return n - len(self.rows)

            class Pipelines:
                def get_by_id(self, pid):
                    p = t["pipelines"].get(pid)
                    return None if p is None else _Obj(id=pid, name=p["name"], config=p["config"])

                def delete_by_id(self, pid):
                    t["pipelines"].pop(pid, None)

            self.queries, self.pipelines = Queries(), Pipelines()
            self.chunks, self.image_chunks = Chunks(t["chunks"]), Chunks(t["image_chunks"])
            self.chunk_results = Results(t["chunk_results"], "chunk_id")
            self.image_chunk_results = Results(t["image_chunk_results"], "image_chunk_id")
            self.evaluation_results = _Obj(delete_by_pipeline=lambda pid: 0)
            self.committed = 0

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def commit(self):
            self.committed += 1

    def _create_uow(self):
        self.uow_opened += 1
        return FakeRefService._Uow(self)

    def get_or_create_pipeline(self, name, config, *, strict=False):
        for pid, p in self.t["pipelines"].items():
            if p["name"] == name:
                return pid, False
        pid = max(self.t["pipelines"], default=0) + 1
        self.t["pipelines"][pid] = {"name": name, "config": dict(config)}
        return pid, True

    def delete_pipeline_results(self, pipeline_id):
        with self._create_uow() as uow:
            return uow.chunk_results.delete_by_pipeline(pipeline_id) + uow.image_chunk_results.delete_by_pipeline(pipeline_id)

    def verify_pipeline_completion(self, pipeline_id):
        done = {r["query_id"] for r in self.t["chunk_results"] + self.t["image_chunk_results"] if r["pipeline_id"] == pipeline_id}
        return all(q.id in done for q in self.t["queries"])


def ref_tables_from_store(store) -> dict:
    """The tables of an InMemoryStore the way the reference's ORM would hand them out (lists / None)."""
    def rows(tab):
        out = []
        for i, pk in enumerate(tab.ids):
            emb = None
            if tab.embedding is not None and not np.isnan(tab.embedding[i]).all():
                emb = [float(x) for x in tab.embedding[i]]
            mv = None
            if tab.mv_offsets is not None and tab.mv_offsets[i + 1] > tab.mv_offsets[i]:
                mv = [[float(x) for x in r] for r in tab.mv_tokens[tab.mv_offsets[i]:tab.mv_offsets[i + 1]]]
            out.append(_Obj(id=pk, contents=tab.contents[i], embedding=emb, embeddings=mv))
        return out

    qs = [_Obj(id=q.id, contents=q.contents, embedding=None if q.embedding is None else [float(x) for x in q.embedding],
               embeddings=None if q.embeddings is None else [[float(x) for x in r] for r in q.embeddings])
          for q in (store.queries[k] for k in store.query_order)]
    return {"queries": qs, "chunks": rows(store.chunks), "image_chunks": rows(store.image_chunks), "pipelines": {},
            "chunk_results": [], "image_chunk_results": []}


class FakeSessionmaker:
    """What the reference Executor passes as `session_factory`: calling it yields a session object, NOT a store."""

    def __init__(self, tables):
        self.tables = tables
        self.sessions = 0

    def __call__(self):
        self.sessions += 1
        return _Obj(close=lambda: None, execute=lambda *a, **k: None)
