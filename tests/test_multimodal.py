"""ColPali / BiPali-shaped wrappers (reference embeddings/colpali.py:56-245, bipali.py:53-250): interface
conformance, return nesting and the image corpus -> MaxSim store -> image pipeline flow with random-init stand-ins (no
checkpoint / colpali_engine offline).  GPU: patch embeddings go from the vision stub to the index by device pointer."""

import asyncio
import io

import numpy as np
import pytest

from helpers import OracleIndex


def _images(n, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 255, size=(64 + 8 * i, 96, 3), dtype=np.uint8) for i in range(n)]


@pytest.fixture(scope="module")
def col():
    from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings, RandomVisualProcessor, make_random_col_model

    return Mi355ColPaliEmbeddings(model_name="stub/colpali", model_type="pali", device="cpu", torch_dtype="float32",
                                  model=make_random_col_model(), processor=RandomVisualProcessor(), batch_size=2)


def test_colpali_interface_and_shapes(col):
    from autorag_research_amd.embeddings import MultiVectorBaseEmbedding, MultiVectorMultiModalEmbedding, health_check_embedding

    assert isinstance(col, MultiVectorMultiModalEmbedding) and isinstance(col, MultiVectorBaseEmbedding)
    assert health_check_embedding(col) == 128                      # injection.py:24-45 probe
    imgs = _images(3)
    one = col.embed_image(imgs[0])
    assert len(one) == 1030 and len(one[0]) == 128                  # (448/14)^2 patches + 6 prefix tokens, like ColPali
    assert abs(np.linalg.norm(one[17]) - 1.0) < 1e-5
    many = col.embed_images(imgs)
    assert [len(m) for m in many] == [1030] * 3 and np.allclose(many[0], one, atol=1e-6)
    assert np.allclose(asyncio.run(col.aembed_image(imgs[1])), many[1], atol=1e-6)
    q = col.embed_query("what is on the page")
    assert len(q) == 5 and len(q[0]) == 128 and col.embed_text("what is on the page") == q
    # like the reference (`[emb.cpu().tolist() for emb in embeddings]`, colpali.py:189-216), every row the model returns for an
    # item is kept -- the padded positions of the shorter items as zero vectors --, and `embed_documents` is ONE padded batch;
    # `embed_documents_batch` cuts the list into `embed_batch_size` (2 here), each padded to its own longest (base.py:77-83)
    docs = col.embed_documents(["a b c", "d", ""])
    assert [len(x) for x in docs] == [3, 3, 3] and col.embed_documents([]) == [] and col.embed_images([]) == []
    assert [len(x) for x in col.embed_documents_batch(["a b c", "d", ""])] == [3, 3, 1]
    assert np.allclose(docs[1][1:], 0.0) and abs(np.linalg.norm(docs[1][0]) - 1.0) < 1e-5
    col.drop_padding = True      # the ragged form: attended positions only
    assert [len(x) for x in col.embed_documents(["a b c", "d", ""])] == [3, 1, 1]
    col.drop_padding = False
    assert col.embed_images_batch(imgs) == many or np.allclose(col.embed_images_batch(imgs)[2], many[2], atol=1e-6)
    # PNG bytes and a file path go through load_image like the reference's ImageType
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(imgs[0]).save(buf, format="PNG")
    assert np.allclose(col.embed_image(buf.getvalue()), one, atol=1e-6)


def test_unknown_model_type_and_missing_engine():
    from autorag_research_amd.multimodal import Mi355BiPaliEmbeddings, Mi355ColPaliEmbeddings

    with pytest.raises(ValueError, match="Unknown model_type"):
        Mi355ColPaliEmbeddings(model_type="nope", model=object(), processor=object())
    with pytest.raises(ImportError, match="colpali_engine is required"):
        Mi355ColPaliEmbeddings(model_type="pali")
    with pytest.raises(ImportError, match="colpali_engine is required"):
        Mi355BiPaliEmbeddings(model_type="pali")


def test_bipali_interface():
    from autorag_research_amd.embeddings import Embeddings, SingleVectorMultiModalEmbedding, health_check_embedding
    from autorag_research_amd.multimodal import Mi355BiPaliEmbeddings, RandomVisualProcessor, make_random_col_model

    bi = Mi355BiPaliEmbeddings(model_name="stub/bipali", model_type="pali", device="cpu", torch_dtype="float32",
                               model=make_random_col_model(pooled=True), processor=RandomVisualProcessor())
    assert isinstance(bi, SingleVectorMultiModalEmbedding) and isinstance(bi, Embeddings)
    assert health_check_embedding(bi) == 128
    imgs = _images(2, seed=3)
    v = bi.embed_image(imgs[0])
    assert len(v) == 128 and abs(np.linalg.norm(v) - 1) < 1e-5
    assert np.allclose(bi.embed_images(imgs)[0], v, atol=1e-6)
    assert np.allclose(asyncio.run(bi.aembed_query("x y")), bi.embed_query("x y"))
    assert len(bi.embed_documents(["a", "b c"])) == 2 and bi.embed_queries(["x y"])[0] == bi.embed_query("x y")


def test_image_corpus_to_image_pipeline(col, monkeypatch, oracle):
    """C5 plumbing: page images -> multi-vector embeddings -> image_chunk table -> ImageVectorSearch (multi) pipeline; the
    page a query was cut from ranks first."""
    import autorag_research_amd.service as svc
    from autorag_research_amd.pipelines import Mi355ImageVectorSearchRetrievalPipeline
    from autorag_research_amd.store import InMemoryStore

    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    imgs = _images(5, seed=9)
    pages = col.embed_images(imgs)
    store = InMemoryStore()
    store.set_image_chunks([f"page{i}" for i in range(5)], multivec=[np.asarray(p, np.float32) for p in pages])
    # a "query" made of 12 patch vectors of page 3: late interaction must pick that page
    store.add_queries(["q"], contents=["from page 3"], embeddings=[np.asarray(pages[3][100:112], np.float32)])
    p = Mi355ImageVectorSearchRetrievalPipeline(lambda: store, "img_multi", search_mode="multi")
    res = asyncio.run(p._retrieve_by_id("q", 3))
    assert res[0]["doc_id"] == "page3" and abs(res[0]["score"] - 1.0) < 1e-5 and res[0]["score"] > res[1]["score"]


@pytest.mark.gpu
def test_patch_embeddings_reach_the_index_by_device_pointer(native_built):
    import torch

    import autorag_research_amd as pkg
    from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings, RandomVisualProcessor, make_random_col_model

    col_gpu = Mi355ColPaliEmbeddings(model_name="stub/colpali", model_type="pali", device="cuda:0", torch_dtype="float32",
                                     model=make_random_col_model(), processor=RandomVisualProcessor(), batch_size=4)
    imgs = _images(9, seed=2)
    flat, off = col_gpu.encode_images_to_device(imgs)
    assert flat.is_cuda and tuple(flat.shape) == (9 * 1030, 128) and off.tolist() == [1030 * i for i in range(10)]
    with pkg.Mi355Index(128) as dev, pkg.Mi355Index(128) as host:
        assert col_gpu.index_images_on_device(dev, imgs) == 9
        host.add_multivec(flat.cpu().numpy(), off)
        q, qoff = col_gpu.encode_texts_to_device(["find the chart", "second query text here"], query=True)
        qh = q.cpu().numpy()
        a = dev.search_maxsim(qh, qoff.astype(np.int32), 5)
        b = host.search_maxsim(qh, qoff.astype(np.int32), 5)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        # late interaction finds the page a patch-query was cut from
        sub = flat[4 * 1030 + 50: 4 * 1030 + 74].cpu().numpy()
        d, r = dev.search_maxsim(sub, np.array([0, 24], np.int32), 2)
        assert r[0, 0] == 4 and abs(-d[0, 0] / 24 - 1.0) < 1e-5
    torch.cuda.synchronize()
