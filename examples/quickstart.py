"""End-to-end tour of the MI355X path on synthetic data (needs an MI355X and a built libmi355dr.so).

    python -c "import __graft_entry__ as g; g.build()"
    python examples/quickstart.py

1. single-vector index: add fp32 rows, exact cosine top-k for a block of queries (ids + float8 distances)
2. multi-vector store: ragged docs, exact MaxSim top-k, candidate re-scoring
3. the reference-shaped pipelines over an in-memory store: vector search, image (MaxSim) search, HEAVEN two-stage,
   Guided Query Refinement, RRF / convex-combination fusion and HyDE over child retrievers
4. group-nDCG of the persisted results
"""
import asyncio
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import autorag_research_amd as amd  # noqa: E402
from autorag_research_amd.evaluation import evaluate  # noqa: E402
from autorag_research_amd.gqr import Mi355GQRHybridRetrievalPipeline  # noqa: E402
from autorag_research_amd.heaven import Mi355HEAVENRetrievalPipeline  # noqa: E402
from autorag_research_amd.hybrid import Mi355HybridCCRetrievalPipeline, Mi355HybridRRFRetrievalPipeline  # noqa: E402
from autorag_research_amd.hyde import Mi355HyDERetrievalPipeline  # noqa: E402
from autorag_research_amd.metrics import retrieval_ndcg  # noqa: E402
from autorag_research_amd.pipelines import Mi355ImageVectorSearchRetrievalPipeline, Mi355VectorSearchRetrievalPipeline  # noqa: E402
from autorag_research_amd.store import InMemoryStore, RetrievalRelation  # noqa: E402

rng = np.random.default_rng(0)

# ---- 1. single vectors -------------------------------------------------------------------------------------------------
n, d = 200_000, 384
corpus = rng.standard_normal((n, d)).astype(np.float32)
queries = corpus[:64] + 0.3 * rng.standard_normal((64, d)).astype(np.float32)  # every query has a planted neighbour
with amd.Mi355Index(d, "cosine") as idx:
    idx.add(corpus)
    dist, rows = idx.search(queries, k=10)          # float8 cosine distances (pgvector <=>), row indices
    print("single-vector: planted neighbour found first for", int((rows[:, 0] == np.arange(64)).sum()), "of 64 queries;",
          "screen =", {1: "bf16", 2: "int8"}[idx.stat("screen_dtype_active")])

# ---- 2. multi vectors (late interaction) ---------------------------------------------------------------------------------
n_docs, dm = 5_000, 128
lens = rng.integers(20, 120, size=n_docs)
tok = rng.standard_normal((int(lens.sum()), dm)).astype(np.float32)
tok /= np.linalg.norm(tok, axis=1, keepdims=True)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
qtok = tok[off[42]:off[42] + 16] + 0.1 * rng.standard_normal((16, dm)).astype(np.float32)  # 16 query vectors near doc 42
qoff = np.array([0, 16], dtype=np.int32)
with amd.Mi355Index(dm) as idx:
    idx.add_multivec(tok, off)
    mdist, mrows = idx.search_maxsim(qtok, qoff, k=5)   # VectorChord @#: -sum_i max_j <q_i, d_j>
    print("MaxSim: best doc", int(mrows[0, 0]), "score", float(-mdist[0, 0] / 16))
    print("candidate re-scoring of docs [42, 7, 9]:", (-idx.maxsim_subset(qtok, qoff, np.array([[42, 7, 9]])) / 16).round(4).tolist())

# ---- 3. the reference-shaped pipelines ------------------------------------------------------------------------------------
store = InMemoryStore()
ids = [f"chunk-{i}" for i in range(2000)]
emb = corpus[:2000]
multivec = [tok[off[i]:off[i + 1]] for i in range(2000)]
store.set_chunks(ids, [f"text of {i}" for i in ids], embedding=emb)
store.set_image_chunks([f"page-{i}" for i in range(2000)], embedding=emb, multivec=multivec)
qids = [f"q{i}" for i in range(8)]
store.add_queries(qids, contents=[f"what is in chunk {i}" for i in range(8)], embedding=list(queries[:8]),
                  embeddings=[multivec[i][:8] for i in range(8)])
store.add_relations([RetrievalRelation(query_id=q, chunk_id=ids[i], group_index=0, group_order=0) for i, q in enumerate(qids)])

p = Mi355VectorSearchRetrievalPipeline(lambda: store, "mi355_vector_search", search_mode="single")
print("pipeline.run:", p.run(top_k=5, batch_size=1024))
print("pipeline.retrieve (one query):", [r["doc_id"] for r in asyncio.run(p._retrieve_by_id("q3", 3))])
n_eval, mean_ndcg, _ = evaluate(store, p.pipeline_id, retrieval_ndcg, qids)
print(f"nDCG@5 over {n_eval} queries: {mean_ndcg:.3f}")
p.close()

img = Mi355ImageVectorSearchRetrievalPipeline(lambda: store, "mi355_image_vector_search", search_mode="multi")
print("image pipeline (MaxSim):", [r["doc_id"] for r in asyncio.run(img._retrieve_by_id("q2", 3))])
img.close()

heaven = Mi355HEAVENRetrievalPipeline(lambda: store, "mi355_heaven", stage1_candidate_count=100, pos_tagger=None)
print("HEAVEN (cosine top-100 -> candidate MaxSim):", [r["doc_id"] for r in asyncio.run(heaven._retrieve_by_id("q2", 3))])
heaven.close()


class ToyLexicalRetriever:
    """Stands in for a BM25 child pipeline: any object with `name` and an async `_retrieve_by_id(query_id, k)`."""

    name, search_mode, retrieval_unit, _embedding_model = "toy_lexical", "single", "chunk", None

    async def _retrieve_by_id(self, query_id, top_k):
        i = int(str(query_id)[1:])
        picks = [ids[i]] + [ids[(37 * i + 11 * j) % 2000] for j in range(1, top_k)]
        return [{"doc_id": pk, "score": 9.0 - 0.5 * j, "content": None} for j, pk in enumerate(picks)]


dense = Mi355VectorSearchRetrievalPipeline(lambda: store, "mi355_vector_search_child", search_mode="single")
gqr = Mi355GQRHybridRetrievalPipeline(lambda: store, "mi355_gqr_hybrid", dense, ToyLexicalRetriever(), n_steps=25)
print("GQR (dense + lexical pool, 25 refinement steps on the GPU):",
      [(r["doc_id"], round(r["score"], 3)) for r in asyncio.run(gqr._retrieve_by_id("q5", 3))])
print("GQR run (whole page refined in one launch):", gqr.run(top_k=5))
gqr.close()

rrf = Mi355HybridRRFRetrievalPipeline(lambda: store, "mi355_hybrid_rrf", dense, ToyLexicalRetriever())
cc = Mi355HybridCCRetrievalPipeline(lambda: store, "mi355_hybrid_cc", dense, ToyLexicalRetriever(), weight=0.7, normalize_method="z")
print("hybrid RRF:", [r["doc_id"] for r in asyncio.run(rrf._retrieve_by_id("q5", 3))],
      "| hybrid CC (z-score, w=0.7):", [r["doc_id"] for r in asyncio.run(cc._retrieve_by_id("q5", 3))])
print("hybrid RRF run:", rrf.run(top_k=5))
rrf.close()
cc.close()
dense.close()


class EchoLLM:
    """Stands in for the LLM of HyDE: anything with an async `ainvoke(prompt)` (a LangChain chat model, ...)."""

    async def ainvoke(self, prompt):
        return "a passage that restates: " + prompt


class NearestRowEmbedding:
    """Stands in for the passage encoder: here it simply returns the stored vector of the chunk the text mentions."""

    async def aembed_query(self, text):
        digits = "".join(ch for ch in text.split("chunk")[-1] if ch.isdigit())
        return emb[int(digits or 0) % 2000].tolist()


hyde = Mi355HyDERetrievalPipeline(lambda: store, "mi355_hyde", EchoLLM(), NearestRowEmbedding())
print("HyDE (LLM passage -> embedding -> exact top-k):", [r["doc_id"] for r in asyncio.run(hyde._retrieve_by_id("q6", 3))])
hyde.close()
