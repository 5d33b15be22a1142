"""GPU, SURVEY 8(a) row a11 and 8(f) row f3 through the real index:
  * `load_embedding_model("mi355_bge_base" / "mi355_colbertv2")` (reference injection.py:111-139, 226-270; configs/embedding/*.yaml)
    resolves the YAMLs against tiny LOCAL checkpoints the test writes (no real checkpoint is reachable offline), the model runs on
    the MI355X, its output reaches the index BY DEVICE POINTER (mi355dr_add_rows_device / mi355dr_add_multivec_device) and the
    search equals the oracle on the very vectors the encoder produced;
  * a COPY-text dump (the text form of VECTOR(d) / VECTOR(d)[] pinned to the reference's converters, tests/golden/
    pgtext_golden.json) -> shard directory -> `Mi355RetrievalService` over the REAL index -> the write-back COPY rows: ids and
    scores are the oracle's on the parsed vectors."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
pytest.importorskip("transformers")


def test_yaml_named_encoders_run_on_the_gpu_and_land_in_the_index_by_device_pointer(tmp_path, monkeypatch, native_built, oracle):
    import autorag_research_amd as pkg
    import autorag_research_amd.embeddings as E
    from autorag_research_amd.ingest import index_texts_on_device
    from test_encoder_configs import _write_checkpoint

    monkeypatch.setattr(E, "_cache", {})
    bert, colbert = tmp_path / "tiny-bert", tmp_path / "tiny-colbert"
    _write_checkpoint(bert, colbert=False)
    _write_checkpoint(colbert, colbert=True)
    monkeypatch.setenv("MI355_ENCODER_DEVICE", "cuda:0")
    monkeypatch.setenv("MI355_BGE_PATH", str(bert))
    monkeypatch.setenv("MI355_COLBERT_PATH", str(colbert))
    words = "dense retrieval on one gpu late interaction row sharded top k merge over xgmi links".split()
    rng = np.random.default_rng(5)
    corpus = [" ".join(rng.choice(words, size=int(rng.integers(2, 9)))) for _ in range(400)]
    queries = ["what is late interaction", "dense retrieval on one gpu", "top k merge"]

    # ---- single vector: mi355_bge_base.yaml -> TorchEncoderEmbeddings.from_pretrained on cuda:0
    bge = E.load_embedding_model("mi355_bge_base")
    assert isinstance(bge, E.TorchEncoderEmbeddings) and str(bge.device).startswith("cuda") and bge.pooling == "cls"
    assert E.health_check_embedding(bge) == 48
    with pkg.Mi355Index(48) as idx:
        assert index_texts_on_device(bge, corpus, idx, batch_size=128) == len(corpus)      # mi355dr_add_rows_device
        C = idx.get_rows(0, len(corpus))
        qd = bge.encode_to_device(queries)
        assert qd.is_cuda and qd.dtype == torch.float32
        od = torch.empty((3, 5), dtype=torch.float64, device="cuda")
        orr = torch.empty((3, 5), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        idx.search_device(qd.data_ptr(), 3, 5, od.data_ptr(), orr.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    rd, rr = oracle.topk_search(C, qd.cpu().numpy(), 5)
    assert np.array_equal(orr.cpu().numpy(), rr) and np.array_equal(od.cpu().numpy().view(np.uint64), rd.view(np.uint64))
    # the GPU forward is the CPU forward of the same checkpoint up to GEMM rounding
    monkeypatch.setattr(E, "_cache", {})
    monkeypatch.setenv("MI355_ENCODER_DEVICE", "cpu")
    cpu = E.load_embedding_model("mi355_bge_base")
    assert np.abs(np.asarray(cpu.embed_documents(corpus[:16])) - C[:16]).max() < 5e-4

    # ---- multi vector: mi355_colbertv2.yaml -> TorchLateInteractionEmbeddings.from_pretrained (projection read from the checkpoint)
    monkeypatch.setattr(E, "_cache", {})
    monkeypatch.setenv("MI355_ENCODER_DEVICE", "cuda:0")
    col = E.load_embedding_model("mi355_colbertv2")
    assert isinstance(col, E.TorchLateInteractionEmbeddings) and E.health_check_embedding(col) == 128
    docs = col.embed_documents(corpus[:120])
    lens = [len(v) for v in docs]
    flat = torch.tensor(np.concatenate([np.asarray(v, np.float32) for v in docs]), device="cuda")
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    qv = col.embed_queries(queries)
    assert all(len(v) == 32 for v in qv)                          # [MASK]-augmented to 32 vectors each
    qtok = np.concatenate([np.asarray(v, np.float32) for v in qv])
    qoff = np.array([0, 32, 64, 96], dtype=np.int32)
    with pkg.Mi355Index(128) as mv:
        torch.cuda.synchronize()
        mv.add_multivec_device(flat.data_ptr(), off)              # mi355dr_add_multivec_device: no host round trip
        d, r = mv.search_maxsim(qtok, qoff, 6)
    md, mr = oracle.maxsim_topk(flat.cpu().numpy(), off, qtok, qoff, 6)
    assert np.array_equal(r, mr) and np.array_equal(d.view(np.uint32), md.view(np.uint32))


def test_copy_dump_to_shard_to_real_index_and_write_back(tmp_path, native_built, oracle):
    from autorag_research_amd import pgtext as pt
    from autorag_research_amd.service import Mi355RetrievalService
    from autorag_research_amd.shards import read_shard
    from autorag_research_amd.store import ChunkTable, InMemoryStore

    rng = np.random.default_rng(4)
    n, d = 3000, 64
    emb = rng.standard_normal((n, d)).astype(np.float32)
    emb[7] = np.nan                                     # NULL embedding
    toks = [rng.standard_normal((int(t), d)).astype(np.float32) for t in rng.integers(0, 6, size=n)]
    lens = [t.shape[0] for t in toks]
    table = ChunkTable(ids=list(range(100, 100 + n)), contents=[f"text\twith tab {i}\nand newline" if i % 9 == 0 else f"t{i}"
                                                                for i in range(n)],
                       embedding=emb, mv_tokens=np.concatenate(toks), mv_offsets=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
    dump = pt.table_to_copy_text(table)                 # `COPY (SELECT id, contents, embedding, embeddings FROM chunk ORDER BY id) TO STDOUT`
    shard = pt.copy_text_to_shard(iter(dump + [r"\."]), tmp_path / "chunk", id_type="int")
    back = read_shard(shard)
    assert back.ids == table.ids and np.array_equal(back.mv_tokens, table.mv_tokens)
    store = InMemoryStore()
    store.chunks = back
    q = rng.standard_normal((5, d)).astype(np.float32)
    qids = [f"q{i}" for i in range(5)]
    store.add_queries(qids, embedding=list(q))
    s = Mi355RetrievalService(lambda: store)           # the real Mi355Index behind the service
    res = s.vector_search(qids, 7)
    live = np.array([i for i in range(n) if i != 7])
    rd, rr = oracle.topk_search(back.embedding[live], q, 7)
    for b in range(5):
        assert [r["doc_id"] for r in res[b]] == [100 + int(live[j]) for j in rr[b]]          # (the NULL row is never returned)
        assert [r["score"] for r in res[b]] == [1.0 - x for x in rd[b].tolist()]            # retrieval_pipeline.py:504-524
    rows = pt.results_to_copy_text(5, qids + ["qz"], res + [None])
    assert len(rows) == 35
    f = rows[0].split("\t")
    assert f[0] == "q0" and f[1] == "5" and int(f[2]) == res[0][0]["doc_id"] and float(f[3]) == res[0][0]["score"]
