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


class MultiVectorBaseEmbedding(ABC):
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


class MultiVectorMultiModalEmbedding(MultiVectorBaseEmbedding):
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
                 device: str = "cuda:0", batch_size: int = 256, max_length: int = 512):
        import torch

        self._torch = torch
        self.model = model.to(device).eval()
        self.tokenizer = tokenizer
        self.pooling = pooling
        self.normalize = normalize
        self.device = device
        self.batch_size = batch_size
        self.max_length = max_length

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

    def embed_documents(self, texts: list[str]) -> list[list[float]]:
        return self.encode_to_device(texts).cpu().tolist()

    def embed_queries(self, texts: list[str]) -> list[list[float]]:
        """Batched QUERY-side embeddings (ingest.py, the HyDE block).  This encoder has one tower, so it equals
        embed_documents; an asymmetric model puts its query prefix / instruction here."""
        return self.embed_documents(texts)

    def embed_query(self, text: str) -> list[float]:
        return self._forward([text])[0].cpu().tolist()


class TorchLateInteractionEmbeddings(MultiVectorBaseEmbedding):
    """Multi-vector text encoder on PyTorch-ROCm (ColBERT-style: per-token projection + L2 norm).

    `model(input_ids, attention_mask)` -> [B,T,H] (or an object with `.last_hidden_state`); `proj` is an optional
    torch module H -> dim (ColBERT/ColPali use 128).  Padding tokens are dropped, so each text yields its own
    number of vectors -- the ragged `VECTOR(d)[]` shape of the reference (embeddings/colpali.py:120-133).
    """

    def __init__(self, model: Any, tokenizer: Any, proj: Any | None = None, device: str = "cuda:0",
                 batch_size: int = 64, max_length: int = 180, model_name: str = "late-interaction"):
        import torch

        self._torch = torch
        self.model = model.to(device).eval()
        self.proj = proj.to(device).eval() if proj is not None else None
        self.tokenizer = tokenizer
        self.device = device
        self.embed_batch_size = batch_size
        self.max_length = max_length
        self.model_name = model_name

    def _forward(self, texts: list[str]) -> list[MultiVectorEmbedding]:
        torch = self._torch
        enc = self.tokenizer(texts, padding=True, truncation=True, max_length=self.max_length, return_tensors="pt")
        enc = {k: v.to(self.device) for k, v in enc.items()}
        with torch.no_grad():
            out = self.model(**enc)
            h = out.last_hidden_state if hasattr(out, "last_hidden_state") else out
            if self.proj is not None:
                h = self.proj(h)
            h = torch.nn.functional.normalize(h.float(), dim=-1)
        mask = enc["attention_mask"].bool()
        return [h[i][mask[i]].cpu().tolist() for i in range(h.shape[0])]

    def embed_query(self, query: str) -> MultiVectorEmbedding:
        return self._forward([query])[0]

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


def _instantiate(cfg: dict[str, Any]) -> Any:
    target = cfg.get("_target_")
    if not target:
        raise ValueError("embedding config needs a `_target_`")
    mod, _, attr = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), attr)
    return cls(**{k: v for k, v in cfg.items() if k != "_target_"})


def health_check_embedding(model: Any) -> int:
    """Embed a probe text and return the embedding dim (inner dim for multi-vector), injection.py:24-45."""
    vec = model.embed_query("health check")
    if not vec:
        raise ValueError("embedding health check returned an empty embedding")
    if isinstance(vec[0], (list, tuple)):
        return len(vec[0])
    return len(vec)


def load_embedding_model(config_name: str, config_dir: str | Path | None = None) -> Any:
    """YAML `<config_name>.yaml` with `_target_` -> instance (type-checked, health-checked, cached)."""
    if config_name in _cache:
        return _cache[config_name]
    import yaml

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
