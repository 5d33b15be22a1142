"""CPU: embedding interfaces, loader, batched ingest (tiny random-init torch encoders; no weights offline)."""

import asyncio

import numpy as np
import pytest
import torch


class _TinyTok:
    """whitespace tokenizer -> ids by hash, padded; the subset of the HF tokenizer call the encoders use"""

    def __call__(self, texts, padding=True, truncation=True, max_length=32, return_tensors="pt"):
        ids = [[(hash(w) % 97) + 1 for w in t.split()][:max_length] or [1] for t in texts]
        T = max(len(x) for x in ids)
        inp = torch.zeros((len(ids), T), dtype=torch.long)
        mask = torch.zeros((len(ids), T), dtype=torch.long)
        for i, x in enumerate(ids):
            inp[i, : len(x)] = torch.tensor(x)
            mask[i, : len(x)] = 1
        return {"input_ids": inp, "attention_mask": mask}


class _TinyEnc(torch.nn.Module):
    def __init__(self, h=24):
        super().__init__()
        torch.manual_seed(0)
        self.emb = torch.nn.Embedding(128, h)
        self.lin = torch.nn.Linear(h, h)

    def forward(self, input_ids, attention_mask):
        return torch.tanh(self.lin(self.emb(input_ids)))


def test_single_vector_encoder_batches_and_normalises():
    from autorag_research_amd.embeddings import Embeddings, TorchEncoderEmbeddings, health_check_embedding

    m = TorchEncoderEmbeddings(_TinyEnc(), _TinyTok(), pooling="mean", device="cpu", batch_size=3)
    assert isinstance(m, Embeddings) and health_check_embedding(m) == 24
    texts = [f"doc number {i} about topic {i % 3}" for i in range(8)]
    docs = np.asarray(m.embed_documents(texts))
    assert docs.shape == (8, 24) and np.allclose(np.linalg.norm(docs, axis=1), 1.0, atol=1e-5)
    one = np.asarray(m.embed_query(texts[5]))
    assert np.allclose(one, docs[5], atol=1e-6)  # batching does not change the vectors
    assert np.allclose(asyncio.run(m.aembed_query(texts[5])), one, atol=1e-6)
    dev = m.encode_to_device(texts)
    assert dev.dtype == torch.float32 and dev.is_contiguous() and tuple(dev.shape) == (8, 24)
    cls = TorchEncoderEmbeddings(_TinyEnc(), _TinyTok(), pooling="cls", device="cpu")
    assert len(cls.embed_query("a b c")) == 24


def test_late_interaction_encoder_is_ragged_and_unit_norm():
    from autorag_research_amd.embeddings import (MultiVectorBaseEmbedding, TorchLateInteractionEmbeddings,
                                                  health_check_embedding)

    m = TorchLateInteractionEmbeddings(_TinyEnc(), _TinyTok(), proj=torch.nn.Linear(24, 8), device="cpu", batch_size=2)
    assert isinstance(m, MultiVectorBaseEmbedding) and health_check_embedding(m) == 8
    out = m.embed_documents(["one two three", "four", "five six"])
    assert [len(x) for x in out] == [3, 1, 2] and all(len(v) == 8 for x in out for v in x)
    assert np.allclose([np.linalg.norm(v) for x in out for v in x], 1.0, atol=1e-5)
    assert m.embed_documents_batch(["one two three", "four", "five six"]) == out


def test_loader_and_type_check(tmp_path):
    from autorag_research_amd import embeddings as E

    m = E.load_embedding_model("mock")
    assert isinstance(m, E.Embeddings) and len(m.embed_query("x")) == 384
    assert E.load_embedding_model("mock") is m  # cached
    (tmp_path / "bad.yaml").write_text("_target_: builtins.dict\n")
    with pytest.raises(TypeError):
        E.load_embedding_model("bad", config_dir=tmp_path)
    with pytest.raises(FileNotFoundError):
        E.load_embedding_model("does_not_exist")


def test_embed_all_fills_only_missing_rows_and_search_runs(monkeypatch, oracle):
    import autorag_research_amd.service as svc
    from autorag_research_amd.embeddings import TorchEncoderEmbeddings, TorchLateInteractionEmbeddings
    from autorag_research_amd.ingest import embed_all_chunks, embed_all_queries
    from autorag_research_amd.pipelines import Mi355VectorSearchRetrievalPipeline
    from autorag_research_amd.store import InMemoryStore
    from helpers import OracleIndex

    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    store = InMemoryStore()
    texts = [f"chunk {i} talks about subject {i % 7} and item {i}" for i in range(40)]
    store.set_chunks(list(range(100, 140)), texts)
    store.add_queries(["qa", "qb"], contents=["subject 3 item 10", "chunk 5"])
    single = TorchEncoderEmbeddings(_TinyEnc(), _TinyTok(), pooling="mean", device="cpu", batch_size=16)
    assert embed_all_chunks(store, single, batch_size=16) == 40
    assert embed_all_chunks(store, single) == 0  # nothing left to embed (resume semantics)
    store.chunks.embedding[7] = np.nan
    assert embed_all_chunks(store, single) == 1
    assert embed_all_queries(store, single) == 2 and embed_all_queries(store, single) == 0
    p = Mi355VectorSearchRetrievalPipeline(lambda: store, "ing", search_mode="single", embedding_model=single)
    stats = p.run(top_k=5)
    assert stats["total_queries"] == 2 and stats["total_results"] == 10
    res = asyncio.run(p.retrieve("a brand new question about subject 3", top_k=3))  # text path -> aembed_query
    assert len(res) == 3 and all(r["doc_id"] in range(100, 140) for r in res)
    multi = TorchLateInteractionEmbeddings(_TinyEnc(), _TinyTok(), proj=torch.nn.Linear(24, 8), device="cpu")
    assert embed_all_chunks(store, multi) == 40 and store.chunks.mv_offsets[-1] == store.chunks.mv_tokens.shape[0]
    assert embed_all_queries(store, multi) == 2
    pm = Mi355VectorSearchRetrievalPipeline(lambda: store, "ing_multi", search_mode="multi")
    out = asyncio.run(pm._retrieve_by_id("qa", 4))
    assert len(out) == 4 and out[0]["score"] >= out[-1]["score"] and out[0]["score"] <= 1.0 + 1e-6


@pytest.mark.gpu
def test_ingest_on_the_gpu_then_search_equals_the_oracle(native_built, oracle):
    """Row a12 end to end on the MI355X: `embed_all` (queries, then chunks, through the model's query side like
    data/base.py:57-72) with the encoder on cuda:0 and one row the model cannot embed -- remembered, the rest stored, a second
    run retries it --, then the pipeline's block search over the stored vectors equals the oracle on the same vectors."""
    from autorag_research_amd.embeddings import TorchEncoderEmbeddings
    from autorag_research_amd.ingest import StoreTarget, embed_all, embed_entities_report, BatchEmbedder
    from autorag_research_amd.pipelines import Mi355VectorSearchRetrievalPipeline
    from autorag_research_amd.store import InMemoryStore

    store = InMemoryStore()
    texts = [f"chunk {i} talks about subject {i % 7} and item {i}" for i in range(300)]
    store.set_chunks(list(range(1000, 1300)), texts)
    store.add_queries([f"q{i}" for i in range(5)], contents=[f"subject {i} item {i * 11}" for i in range(5)])
    enc = TorchEncoderEmbeddings(_TinyEnc(), _TinyTok(), pooling="mean", device="cuda:0", batch_size=64)

    class Flaky:
        """fails on one text the first time it sees it (a transient model / service error)"""

        def __init__(self):
            self.failed_once = False

        def embed_queries(self, items):
            if not self.failed_once and texts[123] in items:
                raise RuntimeError("transient")
            return enc.embed_queries(items)

        def embed_query(self, t):
            if not self.failed_once and t == texts[123]:
                self.failed_once = True
                raise RuntimeError("transient")
            return enc.embed_query(t)

    rep = embed_entities_report(StoreTarget(store), "chunk", "single", BatchEmbedder(Flaky(), "query"), batch_size=64)
    assert rep.total_embedded == 299 and rep.failed_ids == [1123] and np.isnan(store.chunks.embedding[123]).all()
    embed_all(store, enc, batch_size=64)     # the next ingest call retries the row and embeds the queries
    assert not np.isnan(store.chunks.embedding).any() and all(store.queries[q].embedding is not None for q in store.query_order)
    p = Mi355VectorSearchRetrievalPipeline(lambda: store, "ing_gpu", search_mode="single")
    stats = p.run(top_k=7)
    assert stats["total_queries"] == 5 and stats["failed_queries"] == []
    Q = np.stack([store.queries[q].embedding for q in store.query_order])
    od, orow = oracle.topk_search(store.chunks.embedding, Q, 7)
    for qi, q in enumerate(store.query_order):
        got = store.chunk_results[(p.pipeline_id, q)]
        assert [c for c, _ in got] == [1000 + int(r) for r in orow[qi]]
        assert [s for _, s in got] == [1.0 - float(d) for d in od[qi]]
    p.close()
