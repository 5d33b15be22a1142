"""Embedding interfaces of the encode half of the path + a batched PyTorch-ROCm encoder.

Interface mirror (same method names and return nesting):
  LangChain `Embeddings` (external, langchain-core==1.2.7): embed_documents / embed_query / aembed_*
  SingleVectorMultiModalEmbedding, MultiVectorBaseEmbedding, MultiVectorMultiModalEmbedding
      autorag_research/embeddings/base.py:12-137
  load_embedding_model / health_check_embedding        autorag_research/injection.py:24-45, 111-139, 226-270
single-vector models return list[float], multi-vector models list[list[float]] (one vector per token).

What is new: `TorchEncoderEmbeddings.embed_documents` really batches (the reference embeds chunks one
text per forward through `aembed_query`, data/base.py:68-72) and `encode_to_device` hands the fp32
matrix to `Mi355Index.add_device` without the `.cpu().tolist()` round trip (colpali.py:133,187).
PyTorch-ROCm is used for the model forward only.
"""

from __future__ import annotations

import asyncio
import hashlib
import importlib
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any

import numpy as np

try:  # pragma: no cover - only where langchain-core is installed
    from langchain_core.embeddings import Embeddings  # type: ignore
except Exception:  # noqa: BLE001

    class Embeddings(ABC):  # type: ignore[no-redef]
        """Stand-in with LangChain's method names (used when langchain-core is absent)."""

        @abstractmethod
        def embed_documents(self, texts: list[str]) -> list[list[float]]: ...

        @abstractmethod
        def embed_query(self, text: str) -> list[float]: ...

        async def aembed_documents(self, texts: list[str]) -> list[list[float]]:
            return await asyncio.to_thread(self.embed_documents, texts)

        async def aembed_query(self, text: str) -> list[float]:
            return await asyncio.to_thread(self.embed_query, text)


MultiVectorEmbedding = list[list[float]]

# Next to an installed reference the multi-vector bases ARE subclasses of the reference's (pydantic models,
# embeddings/base.py:37-137): `injection.load_embedding_model` type-checks what a YAML instantiates with
# `isinstance(model, (Embeddings, MultiVectorBaseEmbedding))` (injection.py:134, 202-206) and pydantic models do not honour
# ABC.register().  `extra="allow"` lets the wrappers keep ordinary attributes.  Without the reference: plain ABCs below.
try:  # pragma: no cover - exercised where the reference (and its dependencies) are importable
    from autorag_research.embeddings.base import MultiVectorBaseEmbedding as _RefMultiVectorBase  # type: ignore
    from autorag_research.embeddings.base import MultiVectorMultiModalEmbedding as _RefMultiVectorMultiModal  # type: ignore
    from pydantic import ConfigDict as _ConfigDict

    HAVE_REFERENCE_EMBEDDINGS = True
except Exception:  # noqa: BLE001
    HAVE_REFERENCE_EMBEDDINGS = False


class SingleVectorMultiModalEmbedding(Embeddings):
    """One vector per text/image (BiPali-style); reference embeddings/base.py:12-30."""

    @abstractmethod
    def embed_image(self, img_file_path: Any) -> list[float]: ...

    @abstractmethod
    async def aembed_image(self, img_file_path: Any) -> list[float]: ...

    def embed_images(self, img_file_paths: list[Any]) -> list[list[float]]:
        return [self.embed_image(p) for p in img_file_paths]

    async def aembed_images(self, img_file_paths: list[Any]) -> list[list[float]]:
        return list(await asyncio.gather(*[self.aembed_image(p) for p in img_file_paths]))


class _MultiVectorBaseABC(ABC):
    """One vector per token (ColBERT-style); reference embeddings/base.py:37-92."""

    model_name: str = "unknown"
    embed_batch_size: int = 10

    @abstractmethod
    def embed_query(self, query: str) -> MultiVectorEmbedding: ...

    @abstractmethod
    async def aembed_query(self, query: str) -> MultiVectorEmbedding: ...

    @abstractmethod
    def embed_text(self, text: str) -> MultiVectorEmbedding: ...

    @abstractmethod
    async def aembed_text(self, text: str) -> MultiVectorEmbedding: ...

    def embed_documents(self, texts: list[str]) -> list[MultiVectorEmbedding]:
        return [self.embed_text(t) for t in texts]

    async def aembed_documents(self, texts: list[str]) -> list[MultiVectorEmbedding]:
        return list(await asyncio.gather(*[self.aembed_text(t) for t in texts]))

    def embed_documents_batch(self, texts: list[str], show_progress: bool = False) -> list[MultiVectorEmbedding]:
        out: list[MultiVectorEmbedding] = []
        for i in range(0, len(texts), self.embed_batch_size):
            out.extend(self.embed_documents(texts[i: i + self.embed_batch_size]))
        return out

    async def aembed_documents_batch(self, texts: list[str], show_progress: bool = False) -> list[MultiVectorEmbedding]:
        out: list[MultiVectorEmbedding] = []
        for i in range(0, len(texts), self.embed_batch_size):
            out.extend(await self.aembed_documents(texts[i: i + self.embed_batch_size]))
        return out


if HAVE_REFERENCE_EMBEDDINGS:  # pragma: no cover
    class MultiVectorBaseEmbedding(_RefMultiVectorBase):  # type: ignore[misc,valid-type]
        """The reference's base itself (its embed_documents / *_batch defaults are inherited), open to extra attributes."""

        model_config = _ConfigDict(arbitrary_types_allowed=True, extra="allow")

else:
    MultiVectorBaseEmbedding = _MultiVectorBaseABC  # type: ignore[misc,assignment]


class _MultiVectorMultiModalABC(_MultiVectorBaseABC):
    """One vector per token/patch, text and image (ColPali-style); reference embeddings/base.py:95-137."""

    @abstractmethod
    def embed_image(self, img_file_path: Any) -> MultiVectorEmbedding: ...

    @abstractmethod
    async def aembed_image(self, img_file_path: Any) -> MultiVectorEmbedding: ...

    def embed_images(self, img_file_paths: list[Any]) -> list[MultiVectorEmbedding]:
        return [self.embed_image(p) for p in img_file_paths]

    async def aembed_images(self, img_file_paths: list[Any]) -> list[MultiVectorEmbedding]:
        return list(await asyncio.gather(*[self.aembed_image(p) for p in img_file_paths]))

    def embed_images_batch(self, img_file_paths: list[Any], show_progress: bool = False) -> list[MultiVectorEmbedding]:
        out: list[MultiVectorEmbedding] = []
        for i in range(0, len(img_file_paths), self.embed_batch_size):
            out.extend(self.embed_images(img_file_paths[i: i + self.embed_batch_size]))
        return out


if HAVE_REFERENCE_EMBEDDINGS:  # pragma: no cover
    class MultiVectorMultiModalEmbedding(_RefMultiVectorMultiModal, MultiVectorBaseEmbedding):  # type: ignore[misc,valid-type]
        model_config = _ConfigDict(arbitrary_types_allowed=True, extra="allow")
else:
    MultiVectorMultiModalEmbedding = _MultiVectorMultiModalABC  # type: ignore[misc,assignment]


def init_multivector_base(obj: Any, model_name: str, embed_batch_size: int) -> None:
    """First statement of a multi-vector wrapper's __init__: next to the reference the base is a pydantic model whose own
    __init__ must have run before any attribute is set (fields `model_name`, `embed_batch_size`, embeddings/base.py:49-50)."""
    if HAVE_REFERENCE_EMBEDDINGS:  # pragma: no cover
        _RefMultiVectorBase.__init__(obj, model_name=model_name, embed_batch_size=embed_batch_size)


def _torch_dtype(torch: Any, dtype: Any) -> Any:
    return getattr(torch, dtype) if isinstance(dtype, str) else dtype


def _checkpoint_tensor(path: str, key: str):
    """One tensor of a LOCAL checkpoint directory by state-dict key (model.safetensors, else pytorch_model.bin), or None."""
    d = Path(path)
    if not d.is_dir():
        return None
    st = d / "model.safetensors"
    if st.exists():
        from safetensors import safe_open  # noqa: PLC0415

        with safe_open(str(st), framework="pt") as f:
            return f.get_tensor(key) if key in f.keys() else None  # noqa: SIM118
    pt = d / "pytorch_model.bin"
    if pt.exists():
        import torch  # noqa: PLC0415

        sd = torch.load(str(pt), map_location="cpu", weights_only=True)
        return sd.get(key)
    return None


class HashingEmbeddings(Embeddings):
    """Deterministic fake (counterpart of langchain's FakeEmbeddings used by configs/embedding/mock.yaml)."""

    def __init__(self, size: int = 384):
        self.size = size

    def _vec(self, text: str) -> list[float]:
        seed = int.from_bytes(hashlib.sha256(text.encode()).digest()[:8], "little")
        return np.random.default_rng(seed).standard_normal(self.size).astype(np.float32).tolist()

    def embed_documents(self, texts: list[str]) -> list[list[float]]:
        return [self._vec(t) for t in texts]

    def embed_query(self, text: str) -> list[float]:
        return self._vec(text)


class TorchEncoderEmbeddings(Embeddings):
    """Single-vector text encoder on PyTorch-ROCm (bge-base: CLS + L2 norm; MiniLM: mean pool + L2 norm).

    `model` is any torch module mapping (input_ids, attention_mask) -> last_hidden_state [B,T,H] (an HF
    AutoModel output with `.last_hidden_state` is accepted); `tokenizer(texts) -> dict of tensors`.
    Weights must already be local: there is no network on the build or GPU boxes.
    """

    def __init__(self, model: Any, tokenizer: Any, pooling: str = "cls", normalize: bool = True,
                 device: str = "cuda:0", batch_size: int = 256, max_length: int = 512, query_prefix: str = "",
                 document_prefix: str = ""):
        import torch

        if pooling not in ("cls", "mean"):
            raise ValueError("pooling must be 'cls' (bge) or 'mean' (MiniLM / sentence-transformers mean pooling)")
        self._torch = torch
        self.model = model.to(device).eval()
        self.tokenizer = tokenizer
        self.pooling = pooling
        self.normalize = normalize
        self.device = device
        self.batch_size = batch_size
        self.max_length = max_length
        # asymmetric models: an instruction in front of queries (bge-*-v1.5's "Represent this sentence for searching relevant
        # passages: ") and / or documents (e5's "passage: "); empty = what langchain's HuggingFaceEmbeddings sends
        self.query_prefix, self.document_prefix = query_prefix or "", document_prefix or ""

    @classmethod
    def from_pretrained(cls, model_name_or_path: str, pooling: str = "cls", normalize: bool = True, device: str = "cuda:0",
                        dtype: Any = "float32", batch_size: int = 256, max_length: int = 512, query_prefix: str = "",
                        document_prefix: str = "", local_files_only: bool = True, trust_remote_code: bool = False):
        """`transformers.AutoModel` + `AutoTokenizer` from a local directory (or the local HF cache: there is no network on the
        build or GPU boxes, hence `local_files_only=True`), in `dtype` on `device`.  The YAML form
        (configs/embedding/mi355_minilm.yaml, mi355_bge_base.yaml) names this classmethod as its `_target_`, the way the
        reference's configs name `langchain_huggingface.HuggingFaceEmbeddings` (configs/embedding/huggingface.yaml)."""
        import torch  # noqa: PLC0415
        from transformers import AutoModel, AutoTokenizer  # noqa: PLC0415

        tok = AutoTokenizer.from_pretrained(model_name_or_path, local_files_only=local_files_only,
                                            trust_remote_code=trust_remote_code)
        model = AutoModel.from_pretrained(model_name_or_path, local_files_only=local_files_only,
                                          trust_remote_code=trust_remote_code, dtype=_torch_dtype(torch, dtype))
        return cls(model, tok, pooling=pooling, normalize=normalize, device=device, batch_size=batch_size, max_length=max_length,
                   query_prefix=query_prefix, document_prefix=document_prefix)

    def _forward(self, texts: list[str]):
        torch = self._torch
        enc = self.tokenizer(texts, padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
        enc = {k: v.to(self.device) for k, v in enc.items()}
        with torch.no_grad():
            out = self.model(**enc)
        h = out.last_hidden_state if hasattr(out, "last_hidden_state") else out
        if self.pooling == "cls":
            v = h[:, 0]
        else:
            m = enc["attention_mask"].unsqueeze(-1).to(h.dtype)
            v = (h * m).sum(1) / m.sum(1).clamp_min(1)
        v = v.float()
        if self.normalize:
            v = torch.nn.functional.normalize(v, dim=1)
        return v.contiguous()

    def encode_to_device(self, texts: list[str]):
        """fp32 [n, H] tensor on the device, batched; feed `.data_ptr()` to Mi355Index.add_device."""
        parts = [self._forward(texts[i: i + self.batch_size]) for i in range(0, len(texts), self.batch_size)]
        return self._torch.cat(parts, dim=0) if parts else self._torch.empty((0, 0), device=self.device)

    def _with(self, prefix: str, texts: list[str]) -> list[str]:
        return [prefix + t for t in texts] if prefix else texts

    def embed_documents(self, texts: list[str]) -> list[list[float]]:
        return self.encode_to_device(self._with(self.document_prefix, texts)).cpu().tolist()

    def embed_queries(self, texts: list[str]) -> list[list[float]]:
        """Batched QUERY-side embeddings (ingest.py, the HyDE block): `query_prefix` + text, one forward per batch."""
        return self.encode_to_device(self._with(self.query_prefix, texts)).cpu().tolist()

    def embed_query(self, text: str) -> list[float]:
        return self._forward(self._with(self.query_prefix, [text]))[0].cpu().tolist()


class TorchLateInteractionEmbeddings(MultiVectorBaseEmbedding):
    """Multi-vector text encoder on PyTorch-ROCm (ColBERT-style: per-token projection + L2 norm).

    `model(input_ids, attention_mask)` -> [B,T,H] (or an object with `.last_hidden_state`); `proj` is an optional
    torch module H -> dim (ColBERT/ColPali use 128).  Padding tokens are dropped, so each text yields its own
    number of vectors -- the ragged `VECTOR(d)[]` shape of the reference (embeddings/colpali.py:120-133).
    """

    def __init__(self, model: Any, tokenizer: Any, proj: Any | None = None, device: str = "cuda:0",
                 batch_size: int = 64, max_length: int = 180, model_name: str = "late-interaction",
                 query_marker_id: int | None = None, doc_marker_id: int | None = None, query_pad_to: int = 0,
                 query_pad_token_id: int | None = None, attend_to_mask_tokens: bool = False,
                 doc_skip_token_ids: Any | None = None):
        import torch

        init_multivector_base(self, model_name, batch_size)
        self._torch = torch
        self.model = model.to(device).eval()
        self.proj = proj.to(device).eval() if proj is not None else None
        self.tokenizer = tokenizer
        self.device = device
        self.embed_batch_size = batch_size
        self.max_length = max_length
        self.model_name = model_name
        # ColBERT's input conventions (Khattab & Zaharia 2020; Santhanam et al. 2022), all optional:
        #   a marker token right behind [CLS] -- [Q] = [unused0] for queries, [D] = [unused1] for documents --; queries padded
        #   to a fixed length with [MASK] tokens whose OUTPUT vectors are kept ("query augmentation", 32 in ColBERTv2) while the
        #   attention mask stays 0 on them (upstream `attend_to_mask_tokens=False`, what colbertv2.0 was trained with; True =
        #   they are attended as well); document vectors of punctuation tokens dropped (upstream `mask_punctuation`: the
        #   skiplist, `doc_skip_token_ids`)
        self.query_marker_id, self.doc_marker_id = query_marker_id, doc_marker_id
        self.query_pad_to, self.query_pad_token_id = int(query_pad_to), query_pad_token_id
        self.attend_to_mask_tokens = bool(attend_to_mask_tokens)
        self.doc_skip_token_ids = sorted({int(t) for t in doc_skip_token_ids}) if doc_skip_token_ids else []

    @classmethod
    def from_pretrained(cls, model_name_or_path: str, dim: int | None = 128, proj_key: str = "linear.weight", device: str = "cuda:0",
                        dtype: Any = "float32", batch_size: int = 64, max_length: int = 180, query_marker: str | None = None,
                        doc_marker: str | None = None, query_pad_to: int = 0, query_pad_token: str = "[MASK]",
                        attend_to_mask_tokens: bool = False, mask_punctuation: bool = False,
                        local_files_only: bool = True, trust_remote_code: bool = False):
        """`AutoModel` + `AutoTokenizer` + the per-token projection of a ColBERT checkpoint: the state-dict tensor `proj_key`
        ([dim, hidden], bias-free `linear` in colbert-ir/colbertv2.0), read from the local checkpoint directory because
        `AutoModel` drops keys it does not know.  `dim=None` = no projection (the encoder's hidden states are the vectors).
        Markers and the pad token are given as tokens of the checkpoint's vocabulary (configs/embedding/mi355_colbertv2.yaml)."""
        import torch  # noqa: PLC0415
        from transformers import AutoModel, AutoTokenizer  # noqa: PLC0415

        tok = AutoTokenizer.from_pretrained(model_name_or_path, local_files_only=local_files_only,
                                            trust_remote_code=trust_remote_code)
        td = _torch_dtype(torch, dtype)
        model = AutoModel.from_pretrained(model_name_or_path, local_files_only=local_files_only,
                                          trust_remote_code=trust_remote_code, dtype=td)
        proj = None
        if dim is not None:
            w = _checkpoint_tensor(model_name_or_path, proj_key)
            if w is None:
                raise FileNotFoundError(f"{model_name_or_path}: no tensor {proj_key!r} in model.safetensors / pytorch_model.bin "
                                        "(a ColBERT checkpoint directory is expected; pass dim=None for a plain encoder)")
            if w.dim() != 2 or w.shape[0] != dim:
                raise ValueError(f"{proj_key} has shape {tuple(w.shape)}, expected [{dim}, hidden]")
            proj = torch.nn.Linear(w.shape[1], w.shape[0], bias=False)
            with torch.no_grad():
                proj.weight.copy_(w)
            proj = proj.to(td)

        def tid(t):
            if t is None:
                return None
            i = tok.convert_tokens_to_ids(t)
            if i is None or i == tok.unk_token_id:
                raise ValueError(f"token {t!r} is not in the checkpoint's vocabulary")
            return int(i)

        skip = None
        if mask_punctuation:   # upstream: every symbol of string.punctuation and its tokenisation (colbert/modeling/colbert.py skiplist)
            import string  # noqa: PLC0415

            skip = set()
            for sym in string.punctuation:
                i = tok.convert_tokens_to_ids(sym)
                if i is not None and i != tok.unk_token_id:
                    skip.add(int(i))
        return cls(model, tok, proj=proj, device=device, batch_size=batch_size, max_length=max_length,
                   model_name=str(model_name_or_path), query_marker_id=tid(query_marker), doc_marker_id=tid(doc_marker),
                   query_pad_to=query_pad_to, query_pad_token_id=tid(query_pad_token) if query_pad_to else None,
                   attend_to_mask_tokens=attend_to_mask_tokens, doc_skip_token_ids=skip)

    def _encode(self, texts: list[str], query: bool) -> dict:
        """-> model inputs + "keep": which output rows become vectors (not passed to the model)."""
        torch = self._torch
        marker = self.query_marker_id if query else self.doc_marker_id
        padded_query = bool(query and self.query_pad_to and self.query_pad_token_id is not None)
        limit = self.query_pad_to if padded_query else self.max_length
        # truncation happens INSIDE the tokenizer (one slot left for the marker), so a long text keeps its [SEP]
        room = limit - (1 if marker is not None else 0)
        enc = self.tokenizer(texts, padding=True, truncation=True, max_length=room, return_tensors="pt")
        ids, mask = enc["input_ids"], enc["attention_mask"]
        if marker is not None:  # [CLS] marker tokens...
            ids = torch.cat([ids[:, :1], torch.full_like(ids[:, :1], marker), ids[:, 1:]], dim=1)
            mask = torch.cat([mask[:, :1], torch.ones_like(mask[:, :1]), mask[:, 1:]], dim=1)
        keep = mask.clone()
        if padded_query:
            L = self.query_pad_to
            if ids.shape[1] < L:
                pad = L - ids.shape[1]
                ids = torch.cat([ids, ids.new_zeros((ids.shape[0], pad))], dim=1)
                mask = torch.cat([mask, mask.new_zeros((mask.shape[0], pad))], dim=1)
            ids, mask = ids.clone(), mask.clone()
            ids[mask == 0] = self.query_pad_token_id   # [MASK] fill: every output row is kept ...
            keep = torch.ones_like(mask)
            if self.attend_to_mask_tokens:             # ... and attended only on request (upstream default: not)
                mask = torch.ones_like(mask)
        elif not query and self.doc_skip_token_ids:
            skip = torch.isin(ids, torch.tensor(self.doc_skip_token_ids, dtype=ids.dtype))
            keep = keep * (~skip).to(keep.dtype)
        out = {"input_ids": ids, "attention_mask": mask, "keep": keep}
        if "token_type_ids" in enc:
            out["token_type_ids"] = torch.zeros_like(ids)
        return out

    def _forward(self, texts: list[str], query: bool = False) -> list[MultiVectorEmbedding]:
        torch = self._torch
        enc = {k: v.to(self.device) for k, v in self._encode(texts, query).items()}
        keep = enc.pop("keep").bool()
        with torch.no_grad():
            out = self.model(**enc)
            h = out.last_hidden_state if hasattr(out, "last_hidden_state") else out
            if self.proj is not None:
                h = self.proj(h)
            h = torch.nn.functional.normalize(h.float(), dim=-1)
        return [h[i][keep[i]].cpu().tolist() for i in range(h.shape[0])]

    def embed_query(self, query: str) -> MultiVectorEmbedding:
        return self._forward([query], query=True)[0]

    def embed_queries(self, texts: list[str]) -> list[MultiVectorEmbedding]:
        """Batched QUERY-side embeddings (ingest.py): one forward per `embed_batch_size` queries."""
        out: list[MultiVectorEmbedding] = []
        for i in range(0, len(texts), self.embed_batch_size):
            out.extend(self._forward(texts[i: i + self.embed_batch_size], query=True))
        return out

    async def aembed_query(self, query: str) -> MultiVectorEmbedding:
        return await asyncio.to_thread(self.embed_query, query)

    def embed_text(self, text: str) -> MultiVectorEmbedding:
        return self._forward([text])[0]

    async def aembed_text(self, text: str) -> MultiVectorEmbedding:
        return await asyncio.to_thread(self.embed_text, text)

    def embed_documents(self, texts: list[str]) -> list[MultiVectorEmbedding]:
        out: list[MultiVectorEmbedding] = []
        for i in range(0, len(texts), self.embed_batch_size):
            out.extend(self._forward(texts[i: i + self.embed_batch_size]))
        return out


# ---- loader (reference injection.py) -------------------------------------------------------------------

_CONFIG_DIRS = [Path(__file__).resolve().parent / "configs" / "embedding"]
_cache: dict[str, Any] = {}


def _locate(path: str) -> Any:
    """`pkg.mod.Class` or `pkg.mod.Class.classmethod` -> the object (Hydra's `_target_` rule: import the longest module prefix,
    then walk attributes)."""
    parts = path.split(".")
    for n in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:n]))
        except ModuleNotFoundError:
            continue
        for a in parts[n:]:
            obj = getattr(obj, a)
        return obj
    raise ImportError(f"cannot locate {path!r}")


_ENV_RE = None


def _resolve(value: Any) -> Any:
    """OmegaConf's `${oc.env:VAR}` / `${oc.env:VAR,default}` in string values (the only interpolation the reference's embedding
    YAMLs use, configs/embedding/colpali.yaml:5-8), for boxes without omegaconf.  A value that is ONE interpolation whose result
    reads as an int / float / bool / null becomes that (OmegaConf parses env values the same way)."""
    global _ENV_RE
    import os  # noqa: PLC0415
    import re  # noqa: PLC0415

    if isinstance(value, dict):
        return {k: _resolve(v) for k, v in value.items()}
    if isinstance(value, list):
        return [_resolve(v) for v in value]
    if not isinstance(value, str) or "${" not in value:
        return value
    if _ENV_RE is None:
        _ENV_RE = re.compile(r"\$\{oc\.env:([A-Za-z_][A-Za-z0-9_]*)(?:,([^}]*))?\}")

    def sub(m):
        if m.group(1) in os.environ:
            return os.environ[m.group(1)]
        if m.group(2) is None:
            raise KeyError(f"environment variable {m.group(1)} is not set and the config gives no default")
        return m.group(2).strip()

    whole = _ENV_RE.fullmatch(value.strip())
    out = _ENV_RE.sub(sub, value)
    if whole:
        import yaml  # noqa: PLC0415

        try:
            parsed = yaml.safe_load(out) if out.strip() else ""
            if parsed is None or isinstance(parsed, (bool, int, float)):
                return parsed
        except yaml.YAMLError:
            pass
    return out


def _instantiate(cfg: dict[str, Any]) -> Any:
    target = cfg.get("_target_")
    if not target:
        raise ValueError("embedding config needs a `_target_`")
    return _locate(target)(**{k: v for k, v in _resolve(cfg).items() if k != "_target_"})


def health_check_embedding(model: Any) -> int:
    """Embed a probe text and return the embedding dim (inner dim for multi-vector), injection.py:24-45."""
    vec = model.embed_query("health check")
    if not vec:
        raise ValueError("embedding health check returned an empty embedding")
    if isinstance(vec[0], (list, tuple)):
        return len(vec[0])
    return len(vec)


def _reference_loader(config_name: str):
    """The reference's own `injection.load_embedding_model` when its configs directory holds `<config_name>.yaml` (a deployment
    next to the reference: OmegaConf + Hydra instantiate, its type check, its health check), else None."""
    try:
        from autorag_research.cli.utils import get_config_dir  # type: ignore  # noqa: PLC0415
        from autorag_research.injection import load_embedding_model as ref_load  # type: ignore  # noqa: PLC0415
    except Exception:  # noqa: BLE001 - no usable reference here
        return None
    try:
        d = Path(get_config_dir()) / "embedding"
    except Exception:  # noqa: BLE001
        return None
    if (d / f"{config_name}.yaml").exists() or (d / f"{config_name}.yml").exists():
        return ref_load
    return None


def load_embedding_model(config_name: str, config_dir: str | Path | None = None) -> Any:
    """YAML `<config_name>.yaml` with `_target_` -> instance (type-checked, health-checked, cached).  Order: an explicit
    `config_dir`; the reference's configs directory THROUGH the reference's loader (injection.py:111-139, 226-240) when the
    reference is installed and has the file (`autorag-research` users copy this package's `configs/embedding/mi355_*.yaml`
    there, or name their own); this package's `configs/embedding/`."""
    if config_name in _cache:
        return _cache[config_name]
    import yaml

    if config_dir is None:
        ref_load = _reference_loader(config_name)
        if ref_load is not None:
            model = ref_load(config_name)
            _cache[config_name] = model
            return model

    dirs = ([Path(config_dir)] if config_dir else []) + _CONFIG_DIRS
    for d in dirs:
        f = d / f"{config_name}.yaml"
        if f.exists():
            model = _instantiate(yaml.safe_load(f.read_text()))
            if not isinstance(model, (Embeddings, MultiVectorBaseEmbedding)):
                raise TypeError(f"{config_name}: expected Embeddings or MultiVectorBaseEmbedding, got {type(model)}")
            health_check_embedding(model)
            _cache[config_name] = model
            return model
    raise FileNotFoundError(f"embedding config '{config_name}.yaml' not found in {[str(d) for d in dirs]}")
