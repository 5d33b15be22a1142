"""SURVEY 8(a) row a10 / a9 with transformers' OWN model classes (no hand-written stand-in): `ColPaliForRetrieval` (PaliGemma =
SigLIP vision tower + Gemma decoder + the 128-d retrieval head) and `BertModel` (the bge family's architecture), instantiated
from small random configs -- there is no network for checkpoints, the ARCHITECTURE and its forward pass are the library's --
and fed through `Mi355ColPaliEmbeddings` / `TorchEncoderEmbeddings`.

Compared with: a direct forward of the same module followed by the reference's own post-processing
(embeddings/colpali.py:109-187: `self._model(**inputs)` under no_grad, `embeddings[0].cpu().tolist()` -- bf16 weights by
default, :81-83 -- ; bge: CLS token + L2 norm), and with an fp32 copy of the module inside bf16 tolerance.  The processor side
(tokenizer / image-processor files) is not reachable offline: `_PaliInputs` builds the tensors `ColPaliProcessor` would
(image-token placeholders + pixel_values; right-padded token ids + attention_mask).

CPU tests run the classes on the host; the `-m gpu` tests run them on the MI355X in bf16 and hand the patch / text vectors to
the index by DEVICE POINTER (mi355dr_add_multivec_device / mi355dr_add_rows_device), then search them."""

import asyncio
import copy
import sys
import types
import zlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from helpers_hf import IMAGE_TOKEN, IMG, N_IMG_TOK, PATCH, VOCAB  # noqa: E402,F401
from helpers_hf import PaliInputs as _PaliInputs  # noqa: E402
from helpers_hf import colpali_config as _colpali_config  # noqa: E402
from helpers_hf import images as _images  # noqa: E402


def _colpali(dtype):
    from transformers import ColPaliForRetrieval

    torch.manual_seed(0)
    return ColPaliForRetrieval(_colpali_config()).to(dtype).eval()


def _reference_post(model, inputs, device):
    """embeddings/colpali.py:120-133 / :176-187, literally: inputs to the device, forward under no_grad, `[i].cpu().tolist()`
    (colpali_engine modules return the tensor; transformers' class returns it as `.embeddings`)."""
    inputs = {k: v.to(device) for k, v in inputs.items()}
    with torch.no_grad():
        out = model(**inputs)
    emb = getattr(out, "embeddings", out)
    return [e.cpu().tolist() for e in emb]


def _check_colpali(device, dtype, dtype_name):
    from autorag_research_amd.embeddings import MultiVectorMultiModalEmbedding, health_check_embedding
    from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings

    model, proc = _colpali(dtype), _PaliInputs()
    col = Mi355ColPaliEmbeddings(model_name="random/colpali-tiny", model_type="pali", device=device, torch_dtype=dtype_name,
                                 model=model, processor=proc, batch_size=2)
    assert isinstance(col, MultiVectorMultiModalEmbedding) and health_check_embedding(col) == 128
    imgs = _images(3)
    # one image: exactly the reference's list for the same module and inputs
    one = col.embed_image(imgs[0])
    exp = _reference_post(col._model, proc.process_images([imgs[0]]), device)[0]
    assert len(one) == N_IMG_TOK + 3 and len(one[0]) == 128 and one == exp
    # a batch (the reference's embed_images, :218-245) -- wrapper batches of 2 + 1 against one batch of 3: same rows
    many = col.embed_images(imgs)
    exp3 = _reference_post(col._model, proc.process_images(imgs), device)
    tol = 0.0 if dtype == torch.float32 and device == "cpu" else 2e-2   # (batch shape changes bf16 / GPU GEMM tiling)
    assert [len(m) for m in many] == [N_IMG_TOK + 3] * 3 and np.allclose(many, exp3, rtol=0, atol=max(tol, 1e-6))
    # queries: process_queries, padded positions zeroed by the head itself
    q = col.embed_query("which page shows the revenue chart")
    assert q == _reference_post(col._model, proc.process_queries(["which page shows the revenue chart"]), device)[0]
    assert asyncio.run(col.aembed_query("which page shows the revenue chart")) == q
    docs = col.embed_documents(["a b c d", "e"])
    assert [len(d) for d in docs] == [5, 5] and np.allclose(docs[1][2:], 0.0) and abs(np.linalg.norm(docs[1][0]) - 1) < 2e-2
    # the same architecture in fp32: the bf16 module's vectors are its bf16-rounded image
    if dtype != torch.float32:
        m32 = copy.deepcopy(model).float()
        ref32 = np.asarray(_reference_post(m32, proc.process_images([imgs[0]]), device)[0])
        assert np.abs(np.asarray(one) - ref32).max() < 6e-2 and abs(np.linalg.norm(ref32[5]) - 1.0) < 1e-5
    return col, imgs


def test_transformers_colpali_through_the_wrapper_cpu():
    _check_colpali("cpu", torch.float32, "float32")
    _check_colpali("cpu", torch.bfloat16, "bfloat16")


def _bert():
    from transformers import BertConfig, BertModel

    torch.manual_seed(1)
    return BertModel(BertConfig(vocab_size=VOCAB, hidden_size=96, num_hidden_layers=2, num_attention_heads=4, intermediate_size=192,
                                max_position_embeddings=128), add_pooling_layer=False).eval()


def _bert_tokenizer(texts, padding=True, truncation=True, max_length=512, return_tensors="pt"):
    t = _PaliInputs()._tok(texts)
    return {k: v[:, :max_length] for k, v in t.items()}


def _check_bert(device):
    from autorag_research_amd.embeddings import Embeddings, TorchEncoderEmbeddings, health_check_embedding

    model = _bert()
    enc = TorchEncoderEmbeddings(model, _bert_tokenizer, pooling="cls", normalize=True, device=device, batch_size=2)
    assert isinstance(enc, Embeddings) and health_check_embedding(enc) == 96
    texts = ["dense retrieval on one gpu", "late interaction", "x", "a longer passage about row sharded top k merge"]
    got = np.asarray(enc.embed_documents(texts), dtype=np.float32)
    # bge: CLS token + L2 norm (sentence-transformers pooling of BAAI/bge-*: pooling_mode_cls_token + Normalize), fp32, per text
    for i, t in enumerate(texts):
        tk = {k: v.to(device) for k, v in _bert_tokenizer([t]).items()}
        with torch.no_grad():
            h = model(**tk).last_hidden_state[:, 0].float()
        ref = torch.nn.functional.normalize(h, dim=1)[0].cpu().numpy()
        assert np.abs(got[i] - ref).max() < 2e-5, i      # (batched with padding vs alone: attention masks out the padding)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    assert np.allclose(enc.embed_query(texts[1]), got[1], atol=2e-5)
    return enc, texts, got


def test_transformers_bert_cls_l2_through_the_encoder_wrapper_cpu():
    _check_bert("cpu")


def test_colpali_engine_loading_branch(monkeypatch):
    """`model=None`: the reference's own loading path (colpali.py:88-108) -- `colpali_engine.models.<Class>.from_pretrained(name,
    dtype=<torch dtype>, trust_remote_code=True)` + `<Processor>.from_pretrained(name)`.  colpali_engine is not installed here; a
    module of that name serves transformers' ColPaliForRetrieval behind colpali_engine's call shape (forward returns the tensor)."""
    import helpers_hf

    from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings

    seen = {}
    helpers_hf.install_colpali_engine(monkeypatch, seen)
    col = Mi355ColPaliEmbeddings(model_name="vidore/colpali-v1.3", model_type="pali", device="cpu")   # torch_dtype: "bfloat16"
    assert seen == {"name": "vidore/colpali-v1.3", "dtype": torch.bfloat16, "trust_remote_code": True,
                    "processor": "vidore/colpali-v1.3"}
    assert next(col._model.parameters()).dtype == torch.bfloat16 and not col._model.training
    v = col.embed_image(_images(1)[0])
    assert len(v) == N_IMG_TOK + 3 and v == _reference_post(col._model, _PaliInputs().process_images(_images(1)), "cpu")[0]
    with pytest.raises(AttributeError, match="Could not find"):
        Mi355ColPaliEmbeddings(model_type="qwen2")   # the module has no ColQwen2


# ---- against the REFERENCE's classes: tests/golden/embeddings_golden.npz (make_golden.py:make_embeddings) --------------------
def _golden():
    from pathlib import Path

    return np.load(Path(__file__).parent / "golden" / "embeddings_golden.npz")


def _ragged_eq(g, name, docs, atol):
    off = g[name + "_off"]
    assert [len(d) for d in docs] == list(np.diff(off)), name          # the padded rows the reference keeps, row for row
    flat = np.asarray([v for d in docs for v in d], dtype=np.float32).reshape(int(off[-1]), -1)
    assert flat.shape == g[name].shape and np.abs(flat - g[name]).max() <= atol, (name, np.abs(flat - g[name]).max())


def _check_against_reference_fixture(monkeypatch, device, atol):
    """Every public method of Mi355ColPaliEmbeddings / Mi355BiPaliEmbeddings, loaded through the same `colpali_engine` stand-in,
    against what the reference's ColPaliEmbeddings / BiPaliEmbeddings returned for the same inputs (float32)."""
    import helpers_hf

    from autorag_research_amd.multimodal import Mi355BiPaliEmbeddings, Mi355ColPaliEmbeddings

    g = _golden()
    helpers_hf.install_colpali_engine(monkeypatch)
    texts = helpers_hf.EMBED_TEXTS
    pngs = [helpers_hf.png_bytes(g[f"image{i}"]) for i in range(3)]
    col = Mi355ColPaliEmbeddings(model_name="tiny/colpali", model_type="pali", device=device, torch_dtype="float32")
    assert col.embed_batch_size == int(g["col_embed_batch_size"])       # the reference's default (base.py:50)
    _ragged_eq(g, "col_embed_text", [col.embed_text(t) for t in texts[:4]], atol)
    _ragged_eq(g, "col_embed_query", [col.embed_query(t) for t in texts[:4]], atol)
    _ragged_eq(g, "col_aembed_query", [asyncio.run(col.aembed_query(texts[0]))], atol)
    _ragged_eq(g, "col_embed_documents", col.embed_documents(texts), atol)              # one padded batch of 12
    _ragged_eq(g, "col_embed_documents_batch", col.embed_documents_batch(texts), atol)  # 10 + 2: other paddings
    _ragged_eq(g, "col_embed_image", [col.embed_image(pngs[0])], atol)
    _ragged_eq(g, "col_embed_images", col.embed_images(pngs), atol)
    _ragged_eq(g, "col_embed_images", col.embed_images_batch(pngs), atol)
    assert col.embed_documents([]) == [] and col.embed_images([]) == []
    bi = Mi355BiPaliEmbeddings(model_name="tiny/bipali", model_type="pali", device=device, torch_dtype="float32", embed_batch_size=5)
    close = lambda a, name: np.abs(np.asarray(a, dtype=np.float32) - g[name]).max() <= atol  # noqa: E731
    assert close([bi.embed_query(t) for t in texts[:4]], "bi_embed_query")
    assert close(bi.embed_documents(texts), "bi_embed_documents")
    assert close(asyncio.run(bi.aembed_documents(texts)), "bi_embed_documents")
    assert close(bi.embed_image(pngs[1]), "bi_embed_image") and close(asyncio.run(bi.aembed_image(pngs[1])), "bi_aembed_image")
    assert close(bi.embed_images(pngs), "bi_embed_images")
    return col, bi, texts, pngs


def test_wrappers_equal_the_reference_classes_cpu(monkeypatch):
    _check_against_reference_fixture(monkeypatch, "cpu", 2e-6)
    with pytest.raises(AttributeError, match="Could not find"):
        from autorag_research_amd.multimodal import Mi355BiPaliEmbeddings

        Mi355BiPaliEmbeddings(model_type="smolvlm")   # in the reference's registry (bipali.py:21), not in this stand-in module


@pytest.mark.gpu
def test_wrappers_equal_the_reference_classes_on_the_gpu(monkeypatch, native_built, oracle):
    """The same fixture with the model on the MI355X (fp32; GPU GEMMs round differently: 5e-4 on unit vectors), then the BiPali
    page vectors go to the index by device pointer and are searched."""
    import autorag_research_amd as pkg

    col, bi, texts, pngs = _check_against_reference_fixture(monkeypatch, "cuda:0", 5e-4)
    with pkg.Mi355Index(128) as idx:
        assert bi.index_images_on_device(idx, pngs) == 3
        C = bi.encode_images_to_device(pngs).cpu().numpy()
        Q = np.asarray(bi.embed_documents(texts[:3]), dtype=np.float32)
        d, r = idx.search(Q, 3)
    od, orow = oracle.topk_search(C, Q, 3)
    assert np.array_equal(r, orow) and np.array_equal(d, od)


@pytest.mark.gpu
def test_transformers_colpali_bf16_on_the_gpu_reaches_the_index_by_device_pointer(native_built, oracle):
    import autorag_research_amd as pkg

    col, imgs = _check_colpali("cuda:0", torch.bfloat16, "bfloat16")
    pages = _images(7, seed=5)
    flat, off = col.encode_images_to_device(pages)
    assert flat.is_cuda and flat.dtype == torch.float32 and tuple(flat.shape) == (7 * (N_IMG_TOK + 3), 128)
    qtok, qoff = col.encode_texts_to_device(["where is the table of contents", "second query"], query=True)
    with pkg.Mi355Index(128) as dev:
        assert col.index_images_on_device(dev, pages) == 7        # mi355dr_add_multivec_device: no host round trip
        d, r = dev.search_maxsim(qtok.cpu().numpy(), qoff.astype(np.int32), 4)
    od, orow = oracle.maxsim_topk(flat.cpu().numpy(), off, qtok.cpu().numpy(), qoff.astype(np.int32), 4)
    assert np.array_equal(r, orow) and np.array_equal(d.view(np.uint32), od.view(np.uint32))


@pytest.mark.gpu
def test_transformers_bert_on_the_gpu_reaches_the_index_by_device_pointer(native_built, oracle):
    import autorag_research_amd as pkg
    from autorag_research_amd.ingest import index_texts_on_device

    enc, texts, got = _check_bert("cuda:0")
    corpus = [f"passage {i} about topic {i % 7} and {'retrieval ' * (i % 5)}" for i in range(300)]
    with pkg.Mi355Index(96) as idx:
        index_texts_on_device(enc, corpus, idx, batch_size=64)     # encoder output -> mi355dr_add_rows_device
        assert len(idx) == 300
        C = enc.encode_to_device(corpus).cpu().numpy()
        Q = enc.encode_to_device(texts).cpu().numpy()
        d, r = idx.search(Q, 5)
    od, orow = oracle.topk_search(C, Q, 5)
    assert np.array_equal(r, orow) and np.array_equal(d, od)
