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

IMAGE_TOKEN, VOCAB, IMG, PATCH = 500, 512, 56, 14
N_IMG_TOK = (IMG // PATCH) ** 2


def _colpali_config():
    from transformers import ColPaliConfig, GemmaConfig, PaliGemmaConfig, SiglipVisionConfig

    vis = SiglipVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=IMG,
                             patch_size=PATCH, projection_dim=96)
    txt = GemmaConfig(vocab_size=VOCAB, hidden_size=96, intermediate_size=192, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=1, head_dim=24, max_position_embeddings=256)
    vlm = PaliGemmaConfig(vision_config=vis, text_config=txt, image_token_index=IMAGE_TOKEN, vocab_size=VOCAB, projection_dim=96,
                          hidden_size=96)
    return ColPaliConfig(vlm_config=vlm, embedding_dim=128)


def _colpali(dtype):
    from transformers import ColPaliForRetrieval

    torch.manual_seed(0)
    return ColPaliForRetrieval(_colpali_config()).to(dtype).eval()


class _PaliInputs:
    """The tensors ColPaliProcessor hands the model: images -> N_IMG_TOK image-token placeholders + a short text suffix and
    `pixel_values`; texts -> right-padded ids + attention_mask (ids hashed from the words: no tokenizer files offline)."""

    def process_images(self, images):
        px = []
        for im in images:
            a = torch.as_tensor(np.asarray(im)).permute(2, 0, 1).float() / 255.0
            px.append(torch.nn.functional.interpolate(a[None], size=(IMG, IMG), mode="bilinear", align_corners=False)[0])
        ids = torch.full((len(images), N_IMG_TOK + 3), IMAGE_TOKEN, dtype=torch.long)
        ids[:, N_IMG_TOK:] = torch.tensor([2, 7, 9])        # <bos> "Describe the image." stand-in
        return {"input_ids": ids, "pixel_values": torch.stack(px), "attention_mask": torch.ones_like(ids)}

    def _tok(self, texts):
        rows = [[2] + [10 + zlib.crc32(w.encode()) % 400 for w in t.split()] for t in texts]
        L = max(len(r) for r in rows)
        ids, mask = torch.zeros((len(rows), L), dtype=torch.long), torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)], mask[i, : len(r)] = torch.tensor(r), 1
        return {"input_ids": ids, "attention_mask": mask}

    def process_queries(self, texts):
        return self._tok(texts)

    def process_texts(self, texts):
        return self._tok(texts)


def _images(n, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 255, size=(60 + 6 * i, 80, 3), dtype=np.uint8) for i in range(n)]


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
    from transformers import ColPaliForRetrieval

    from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings

    seen = {}

    class ColPali(ColPaliForRetrieval):
        @classmethod
        def from_pretrained(cls, name, dtype=None, trust_remote_code=False, **kw):
            seen.update(name=name, dtype=dtype, trust_remote_code=trust_remote_code)
            torch.manual_seed(0)
            return cls(_colpali_config()).to(dtype)

        def forward(self, *a, **kw):
            return super().forward(*a, **kw).embeddings

    class ColPaliProcessor(_PaliInputs):
        @classmethod
        def from_pretrained(cls, name):
            seen["processor"] = name
            return cls()

    eng, models = types.ModuleType("colpali_engine"), types.ModuleType("colpali_engine.models")
    models.ColPali, models.ColPaliProcessor = ColPali, ColPaliProcessor
    eng.models = models
    monkeypatch.setitem(sys.modules, "colpali_engine", eng)
    monkeypatch.setitem(sys.modules, "colpali_engine.models", models)
    col = Mi355ColPaliEmbeddings(model_name="vidore/colpali-v1.3", model_type="pali", device="cpu")   # torch_dtype: "bfloat16"
    assert seen == {"name": "vidore/colpali-v1.3", "dtype": torch.bfloat16, "trust_remote_code": True,
                    "processor": "vidore/colpali-v1.3"}
    assert next(col._model.parameters()).dtype == torch.bfloat16 and not col._model.training
    v = col.embed_image(_images(1)[0])
    assert len(v) == N_IMG_TOK + 3 and v == _reference_post(col._model, _PaliInputs().process_images(_images(1)), "cpu")[0]
    with pytest.raises(AttributeError, match="Could not find"):
        Mi355ColPaliEmbeddings(model_type="qwen2")   # the module has no ColQwen2


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
