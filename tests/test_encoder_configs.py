"""the four BASELINE encoders can be NAMED -- `configs/embedding/mi355_{minilm,bge_base,colbertv2,colpali,
bipali}.yaml` resolve through `load_embedding_model` (reference: injection.py:111-139, 226-240; configs/embedding/colpali.yaml,
huggingface.yaml) to models built by `from_pretrained` from a LOCAL checkpoint directory.  No real checkpoint is reachable
offline: the tests write tiny ones (transformers' own BertModel / BertTokenizerFast, random weights) and point the YAMLs'
`${oc.env:...}` variables at them."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

ROOT = Path(__file__).resolve().parent.parent
VOCAB = ["[PAD]", "[unused0]", "[unused1]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", ".", ","] + \
        [w for w in "dense retrieval on one gpu late interaction row sharded top k merge over xgmi links what is the page about "
                    "health check query passage a b c d".split()]


def _write_checkpoint(d: Path, colbert: bool, hidden: int = 48) -> None:
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import BertConfig, BertModel, PreTrainedTokenizerFast

    d.mkdir(parents=True, exist_ok=True)
    vocab = list(dict.fromkeys(VOCAB))
    v = {w: i for i, w in enumerate(vocab)}
    tk = Tokenizer(models.WordPiece(vocab=v, unk_token="[UNK]"))          # a BERT tokenizer: WordPiece, [CLS] x [SEP]
    tk.normalizer = normalizers.BertNormalizer(lowercase=True)
    tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tk.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                      special_tokens=[("[CLS]", v["[CLS]"]), ("[SEP]", v["[SEP]"])])
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]",
                            mask_token="[MASK]").save_pretrained(str(d))
    torch.manual_seed(3)
    m = BertModel(BertConfig(vocab_size=len(vocab), hidden_size=hidden, num_hidden_layers=2, num_attention_heads=4,
                             intermediate_size=96, max_position_embeddings=64), add_pooling_layer=False)
    m.save_pretrained(str(d))
    if colbert:   # the checkpoint's extra tensor: HF_ColBERT's bias-free `linear` [128, hidden]
        from safetensors.torch import load_file, save_file

        sd = load_file(str(d / "model.safetensors"))
        sd["linear.weight"] = torch.randn((128, hidden), generator=torch.Generator().manual_seed(4)) * 0.1
        save_file(sd, str(d / "model.safetensors"), metadata={"format": "pt"})


@pytest.fixture()
def fresh_loader(monkeypatch):
    import autorag_research_amd.embeddings as E

    monkeypatch.setattr(E, "_cache", {})
    return E


def test_single_vector_yaml_configs_load_local_checkpoints(tmp_path, monkeypatch, fresh_loader):
    E = fresh_loader
    ck = tmp_path / "tiny-bert"
    _write_checkpoint(ck, colbert=False)
    monkeypatch.setenv("MI355_ENCODER_DEVICE", "cpu")
    monkeypatch.setenv("MI355_MINILM_PATH", str(ck))
    monkeypatch.setenv("MI355_BGE_PATH", str(ck))
    mini = E.load_embedding_model("mi355_minilm")
    bge = E.load_embedding_model("mi355_bge_base")
    assert isinstance(mini, E.TorchEncoderEmbeddings) and isinstance(mini, E.Embeddings) and mini.pooling == "mean"
    assert isinstance(bge, E.TorchEncoderEmbeddings) and bge.pooling == "cls" and bge.query_prefix == "" and bge.max_length == 512
    assert E.load_embedding_model("mi355_bge_base") is bge                      # cached like the reference's manager
    texts = ["dense retrieval on one gpu", "late interaction", "what is the page about"]
    for enc in (mini, bge):
        v = np.asarray(enc.embed_documents(texts), dtype=np.float32)
        assert v.shape == (3, 48) and np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-5)
        assert np.allclose(enc.embed_query(texts[1]), v[1], atol=1e-5)
    # the same module by hand: CLS + L2 (bge), masked mean + L2 (MiniLM) -- what the two YAMLs stand for
    tok = transformers.AutoTokenizer.from_pretrained(str(ck))
    model = transformers.AutoModel.from_pretrained(str(ck)).eval()
    enc = tok(texts, padding=True, return_tensors="pt")
    with torch.no_grad():
        h = model(**enc).last_hidden_state
    cls = torch.nn.functional.normalize(h[:, 0], dim=1).numpy()
    m = enc["attention_mask"].unsqueeze(-1).float()
    mean = torch.nn.functional.normalize((h * m).sum(1) / m.sum(1), dim=1).numpy()
    assert np.abs(np.asarray(bge.embed_documents(texts)) - cls).max() < 1e-5
    assert np.abs(np.asarray(mini.embed_documents(texts)) - mean).max() < 1e-5
    # the query instruction is an environment override of the same YAML
    monkeypatch.setattr(E, "_cache", {})
    monkeypatch.setenv("MI355_BGE_QUERY_PREFIX", "query passage ")
    bge2 = E.load_embedding_model("mi355_bge_base")
    assert bge2.query_prefix == "query passage "
    assert np.allclose(bge2.embed_query("late interaction"), bge.embed_documents(["query passage late interaction"])[0], atol=1e-5)
    assert np.allclose(bge2.embed_documents(["late interaction"])[0], bge.embed_documents(["late interaction"])[0], atol=1e-6)


def test_colbert_yaml_reads_the_projection_and_applies_the_input_conventions(tmp_path, monkeypatch, fresh_loader):
    E = fresh_loader
    ck = tmp_path / "tiny-colbert"
    _write_checkpoint(ck, colbert=True)
    monkeypatch.setenv("MI355_ENCODER_DEVICE", "cpu")
    monkeypatch.setenv("MI355_COLBERT_PATH", str(ck))
    col = E.load_embedding_model("mi355_colbertv2")
    assert isinstance(col, E.TorchLateInteractionEmbeddings) and isinstance(col, E.MultiVectorBaseEmbedding)
    assert E.health_check_embedding(col) == 128
    tok = col.tokenizer
    assert (col.query_marker_id, col.doc_marker_id) == (tok.convert_tokens_to_ids("[unused0]"), tok.convert_tokens_to_ids("[unused1]"))
    q = col.embed_query("what is late interaction")
    assert len(q) == 32 and all(len(v) == 128 for v in q)                     # [CLS] [Q] 4 words [SEP] + 25 [MASK]: all kept
    assert np.allclose(np.linalg.norm(np.asarray(q), axis=1), 1.0, atol=1e-5)
    docs = col.embed_documents(["dense retrieval on one gpu", "merge"])
    assert [len(d) for d in docs] == [8, 4]                                      # [CLS] [D] words [SEP]; padding dropped
    # by hand: ids as ColBERT builds them -> BERT -> linear -> L2
    from safetensors.torch import load_file

    W = load_file(str(ck / "model.safetensors"))["linear.weight"]
    model = transformers.AutoModel.from_pretrained(str(ck)).eval()
    ids = [tok.cls_token_id, col.doc_marker_id] + tok.convert_tokens_to_ids("dense retrieval on one gpu".split()) + [tok.sep_token_id]
    with torch.no_grad():
        h = model(input_ids=torch.tensor([ids]), attention_mask=torch.ones((1, len(ids)), dtype=torch.long)).last_hidden_state[0]
    ref = torch.nn.functional.normalize(h @ W.T, dim=-1).numpy()
    assert np.abs(np.asarray(docs[0]) - ref).max() < 1e-5
    qids = [tok.cls_token_id, col.query_marker_id] + tok.convert_tokens_to_ids("what is late interaction".split()) + [tok.sep_token_id]
    n_real = len(qids)
    qids = qids + [tok.mask_token_id] * (32 - len(qids))
    # upstream ColBERTv2 (attend_to_mask_tokens = False): the [MASK] positions are NOT attended, their output vectors are kept
    att = torch.tensor([[1] * n_real + [0] * (32 - n_real)], dtype=torch.long)
    with torch.no_grad():
        hq = model(input_ids=torch.tensor([qids]), attention_mask=att).last_hidden_state[0]
    assert np.abs(np.asarray(q) - torch.nn.functional.normalize(hq @ W.T, dim=-1).numpy()).max() < 1e-5
    assert col.embed_queries(["what is late interaction", "a b"])[0] == q or np.allclose(col.embed_queries(["what is late interaction"])[0], q, atol=1e-6)
    # ... and attended on request (the round-5 behaviour)
    col_att = E.TorchLateInteractionEmbeddings.from_pretrained(str(ck), device="cpu", query_marker="[unused0]", doc_marker="[unused1]",
                                                               query_pad_to=32, attend_to_mask_tokens=True)
    with torch.no_grad():
        hq1 = model(input_ids=torch.tensor([qids]), attention_mask=torch.ones((1, 32), dtype=torch.long)).last_hidden_state[0]
    assert np.abs(np.asarray(col_att.embed_query("what is late interaction")) -
                  torch.nn.functional.normalize(hq1 @ W.T, dim=-1).numpy()).max() < 1e-5
    assert np.abs(np.asarray(col_att.embed_query("what is late interaction")) - np.asarray(q)).max() > 1e-4
    # a query longer than 32 tokens is cut by the TOKENIZER: [CLS] [Q] 29 words [SEP], the [SEP] survives
    words = " ".join(["late", "interaction", "dense", "retrieval"] * 10)
    enc = col._encode([words], query=True)
    assert enc["input_ids"].shape == (1, 32) and int(enc["input_ids"][0, -1]) == tok.sep_token_id and int(enc["attention_mask"].sum()) == 32
    # the punctuation skiplist (upstream mask_punctuation): "." and "," are attended but yield no document vector
    assert set(col.doc_skip_token_ids) >= {tok.convert_tokens_to_ids("."), tok.convert_tokens_to_ids(",")}
    dp = col.embed_documents(["dense retrieval , on one gpu ."])[0]
    pids = [tok.cls_token_id, col.doc_marker_id] + tok.convert_tokens_to_ids("dense retrieval , on one gpu .".split()) + [tok.sep_token_id]
    with torch.no_grad():
        hp = model(input_ids=torch.tensor([pids]), attention_mask=torch.ones((1, len(pids)), dtype=torch.long)).last_hidden_state[0]
    keep = [i for i, t in enumerate(pids) if t not in col.doc_skip_token_ids]
    assert len(dp) == len(pids) - 2 == len(keep)
    assert np.abs(np.asarray(dp) - torch.nn.functional.normalize(hp @ W.T, dim=-1).numpy()[keep]).max() < 1e-5
    # a plain encoder directory has no projection: a clear error, or dim=None
    plain = tmp_path / "plain"
    _write_checkpoint(plain, colbert=False)
    with pytest.raises(FileNotFoundError, match="linear.weight"):
        E.TorchLateInteractionEmbeddings.from_pretrained(str(plain), device="cpu")
    assert len(E.TorchLateInteractionEmbeddings.from_pretrained(str(plain), dim=None, device="cpu").embed_query("a b")[0]) == 48


def test_pali_yaml_configs_take_the_reference_keys(monkeypatch, fresh_loader):
    """mi355_colpali.yaml / mi355_bipali.yaml carry the reference YAMLs' keys and environment variables
    (configs/embedding/colpali.yaml, bipali.yaml) and instantiate through the colpali_engine loading branch."""
    import helpers_hf

    E = fresh_loader
    seen = {}
    helpers_hf.install_colpali_engine(monkeypatch, seen)
    for var in ("COLPALI_DEVICE", "BIPALI_DEVICE"):
        monkeypatch.setenv(var, "cpu")
    monkeypatch.setenv("COLPALI_TORCH_DTYPE", "float32")
    monkeypatch.setenv("BIPALI_TORCH_DTYPE", "float32")
    monkeypatch.setenv("BIPALI_MODEL_TYPE", "pali")
    monkeypatch.setenv("BIPALI_MODEL_NAME", "tiny/bipali")
    col = E.load_embedding_model("mi355_colpali")
    assert seen["name"] == "vidore/colpali-v1.3" and seen["dtype"] == torch.float32 and col.embed_batch_size == 10
    assert isinstance(col, E.MultiVectorMultiModalEmbedding) and E.health_check_embedding(col) == 128
    bi = E.load_embedding_model("mi355_bipali")
    assert seen["name"] == "tiny/bipali" and isinstance(bi, E.SingleVectorMultiModalEmbedding) and len(bi.embed_query("x")) == 128
    import yaml

    ours = yaml.safe_load((ROOT / "autorag_research_amd/configs/embedding/mi355_colpali.yaml").read_text())
    assert set(ours) == {"_target_", "model_name", "model_type", "device", "torch_dtype", "embed_batch_size"}
    ref = Path("/root/reference/configs/embedding/colpali.yaml")
    if ref.exists():   # build container: key for key the reference's file
        theirs = yaml.safe_load(ref.read_text())
        assert set(theirs) == set(ours) and all(ours[k] == theirs[k] for k in ("model_name", "model_type", "torch_dtype", "embed_batch_size"))


def test_interpolation_and_target_resolution():
    from autorag_research_amd import embeddings as E

    os.environ.pop("MI355_TEST_UNSET", None)
    assert E._resolve("${oc.env:MI355_TEST_UNSET,cuda:0}") == "cuda:0" and E._resolve("${oc.env:MI355_TEST_UNSET,}") == "" and E._resolve("${oc.env:MI355_TEST_UNSET,null}") is None
    assert E._resolve({"a": ["${oc.env:MI355_TEST_UNSET,7}", "x${oc.env:MI355_TEST_UNSET,y}z"]}) == {"a": [7, "xyz"]}
    with pytest.raises(KeyError):
        E._resolve("${oc.env:MI355_TEST_UNSET}")
    assert E._locate("autorag_research_amd.embeddings.TorchEncoderEmbeddings.from_pretrained").__self__ is E.TorchEncoderEmbeddings
    with pytest.raises(ImportError):
        E._locate("no_such_package_xyz.Thing")


def test_the_reference_loader_is_preferred_when_it_has_the_config(tmp_path, monkeypatch, fresh_loader):
    """Next to an installed reference whose configs directory holds `<name>.yaml`, the name resolves through the reference's OWN
    `injection.load_embedding_model` (OmegaConf + Hydra, its type and health checks).  Hydra is not installed here: a module of
    that name records the call."""
    import types

    E = fresh_loader
    (tmp_path / "embedding").mkdir()
    (tmp_path / "embedding" / "mi355_bge_base.yaml").write_text("_target_: whatever\n")
    calls = []
    inj, cli_utils = types.ModuleType("autorag_research.injection"), types.ModuleType("autorag_research.cli.utils")
    inj.load_embedding_model = lambda name: calls.append(name) or E.HashingEmbeddings(8)
    cli_utils.get_config_dir = lambda: tmp_path
    for name, mod in {"autorag_research": types.ModuleType("autorag_research"), "autorag_research.cli": types.ModuleType("autorag_research.cli"),
                      "autorag_research.injection": inj, "autorag_research.cli.utils": cli_utils}.items():
        monkeypatch.setitem(sys.modules, name, mod)
    m = E.load_embedding_model("mi355_bge_base")
    assert calls == ["mi355_bge_base"] and len(m.embed_query("x")) == 8
    assert len(E.load_embedding_model("mock").embed_query("x")) == 384 and calls == ["mi355_bge_base"]   # not there: the package's own


@pytest.mark.skipif(not Path("/root/reference/autorag_research/embeddings/base.py").exists(),
                    reason="reference tree only exists in the build container")
def test_multi_vector_wrappers_pass_the_reference_type_check():
    """`injection.load_embedding_model` accepts what a YAML instantiates only if `isinstance(model, (Embeddings,
    MultiVectorBaseEmbedding))` (injection.py:134, 202-206): with the reference importable, the wrappers ARE subclasses of its
    pydantic bases (a fresh process: the bases are chosen at import time)."""
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from ref_import import import_reference\n"
        "import_reference()\n"
        "import torch\n"
        "import autorag_research.embeddings.base as ref\n"
        "import autorag_research_amd.embeddings as E\n"
        "from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings, RandomVisualProcessor, make_random_col_model\n"
        "assert E.HAVE_REFERENCE_EMBEDDINGS\n"
        "col = Mi355ColPaliEmbeddings(model=make_random_col_model(image_size=56, prefix_tokens=2), processor=RandomVisualProcessor(image_size=56), device='cpu')\n"
        "class Tok:\n"
        "    def __call__(self, texts, **kw):\n"
        "        ids = torch.ones((len(texts), 3), dtype=torch.long)\n"
        "        return {'input_ids': ids, 'attention_mask': torch.ones_like(ids)}\n"
        "class Enc(torch.nn.Module):\n"
        "    def __init__(self):\n"
        "        super().__init__(); self.e = torch.nn.Embedding(4, 8)\n"
        "    def forward(self, input_ids, attention_mask):\n"
        "        return self.e(input_ids)\n"
        "late = E.TorchLateInteractionEmbeddings(Enc(), Tok(), device='cpu', batch_size=7, model_name='tiny')\n"
        "out = {'col': [isinstance(col, ref.MultiVectorBaseEmbedding), isinstance(col, ref.MultiVectorMultiModalEmbedding)],\n"
        "       'late': isinstance(late, ref.MultiVectorBaseEmbedding), 'fields': [late.model_name, late.embed_batch_size, col.embed_batch_size],\n"
        "       'q': len(late.embed_query('a b c')), 'docs_batch': len(late.embed_documents_batch(['a'] * 9)),\n"
        "       'img': len(col.embed_image(torch.zeros((3, 56, 56))))}\n"
        "print(json.dumps(out))\n"
    ) % (str(ROOT), str(ROOT / "tests" / "golden"), str(ROOT / "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env={"PYTHONDONTWRITEBYTECODE": "1", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {"col": [True, True], "late": True, "fields": ["tiny", 7, 10], "q": 3, "docs_batch": 9, "img": 18}
