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

    def search_maxsim(self, qtok, q_offsets, k):
        d, r = self._o.maxsim_topk(self._tok, self._off, qtok, q_offsets, k)
        return d, np.where(r >= 0, r + self.row_offset, r)

    def maxsim_subset(self, qtok, q_offsets, doc_ids):
        q = np.ascontiguousarray(qtok, dtype=np.float32).reshape(-1, self.dim)
        ids = np.asarray(doc_ids, dtype=np.int64)
        out = np.full(ids.shape, np.nan, dtype=np.float32)
        n_docs = self._off.shape[0] - 1
        for b in range(ids.shape[0]):
            qb = q[q_offsets[b]:q_offsets[b + 1]]
            for j, i in enumerate(ids[b]):
                if 0 <= i < n_docs and self._off[i + 1] > self._off[i] and qb.shape[0]:
                    out[b, j] = self._o.maxsim_distance(self._tok[self._off[i]:self._off[i + 1]], qb)
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
