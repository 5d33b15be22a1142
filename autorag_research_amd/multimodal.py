"""ColPali / BiPali-shaped multi-modal embedding wrappers on PyTorch-ROCm (config C5: ViDoRe, image corpus).

Mirrors, name for name:
  ColPaliEmbeddings   autorag_research/embeddings/colpali.py:56-245   MultiVectorMultiModalEmbedding: one vector per
                      token / image patch -- embed_text / embed_query / embed_image / embed_documents / embed_images (+ async)
  BiPaliEmbeddings    autorag_research/embeddings/bipali.py:53-250    SingleVectorMultiModalEmbedding: one vector per input
The reference loads `colpali_engine` Col* / Bi* classes by `model_type` and moves every embedding through Python lists
(`embeddings[0].cpu().tolist()`); these wrappers do the same when `colpali_engine` is importable, and take a ready
`model` + `processor` pair otherwise -- a colpali_engine module, or transformers' own `ColPaliForRetrieval` /
`ColQwen2ForRetrieval` (their output object's `.embeddings` is unwrapped).  No checkpoint is reachable offline:
tests/test_hf_models.py runs transformers' ColPaliForRetrieval from a small random config through these wrappers (and the
`colpali_engine` loading branch through a module of that name that serves the same class); the random-init stand-ins below
reproduce the full-size SHAPES -- 1030 patch vectors of 128 dims per page for the `pali` family.

What is new: `encode_images_to_device` / `encode_texts_to_device` return the model's output as ONE ragged device tensor
([sum_T, d] fp32 + host offsets) and `index_images_on_device` hands it to the MaxSim store by pointer
(`mi355dr_add_multivec_device`): patch embeddings never leave HBM between the vision tower and the index.
"""

from __future__ import annotations

import asyncio
import io
from pathlib import Path
from typing import Any, ClassVar

import numpy as np

from .embeddings import (MultiVectorEmbedding, MultiVectorMultiModalEmbedding, SingleVectorMultiModalEmbedding,
                         init_multivector_base)

COL_MODEL_REGISTRY: dict[str, tuple[str, str]] = {  # reference colpali.py:22-29
    "flor": ("ColFlor", "ColFlorProcessor"),
    "modernvbert": ("ColModernVBert", "ColModernVBertProcessor"),
    "smolvlm": ("ColIdefics3", "ColIdefics3Processor"),
    "pali": ("ColPali", "ColPaliProcessor"),
    "qwen2": ("ColQwen2", "ColQwen2Processor"),
    "qwen2_5": ("ColQwen2_5", "ColQwen2_5_Processor"),
}
BI_MODEL_REGISTRY: dict[str, tuple[str, str]] = {  # reference bipali.py:19-25
    "modernvbert": ("BiModernVBert", "BiModernVBertProcessor"),
    "smolvlm": ("BiIdefics3", "BiIdefics3Processor"),
    "pali": ("BiPali", "BiPaliProcessor"),
    "qwen2": ("BiQwen2", "BiQwen2Processor"),
    "qwen2_5": ("BiQwen2_5", "BiQwen2_5_Processor"),
}


def load_image(img: Any):
    """Path / bytes -> PIL image (reference util.load_image); arrays, tensors and PIL images pass through."""
    if isinstance(img, (str, Path, bytes, bytearray)):
        from PIL import Image  # noqa: PLC0415

        return Image.open(io.BytesIO(img) if isinstance(img, (bytes, bytearray)) else img).convert("RGB")
    return img


def _load_engine_classes(registry: dict[str, tuple[str, str]], model_type: str, what: str):
    if model_type not in registry:
        raise ValueError(f"Unknown model_type '{model_type}'. Supported: {list(registry.keys())}")
    m, p = registry[model_type]
    try:
        import colpali_engine.models as models_module  # noqa: PLC0415

        return getattr(models_module, m), getattr(models_module, p)
    except ImportError as e:
        raise ImportError(f"colpali_engine is required for {what} when no `model`/`processor` is passed. "
                          "Install it with: pip install colpali-engine") from e
    except AttributeError as e:
        raise AttributeError(f"Could not find {m} or {p} in colpali_engine.models") from e


class _EngineBacked:
    """Shared plumbing: model + processor, either loaded like the reference or handed in."""

    def _setup(self, registry, what, model_name, model_type, device, torch_dtype, model, processor, batch_size):
        import torch  # noqa: PLC0415

        self._torch = torch
        self.model_name, self.model_type, self.device, self.torch_dtype = model_name, model_type, device, torch_dtype
        self.embed_batch_size = batch_size
        if model is None or processor is None:
            model_class, processor_class = _load_engine_classes(registry, model_type, what)
            dtype = getattr(torch, torch_dtype) if isinstance(torch_dtype, str) else torch_dtype
            processor = processor_class.from_pretrained(model_name)
            model = model_class.from_pretrained(model_name, dtype=dtype, trust_remote_code=True)
        elif model_type not in registry:
            raise ValueError(f"Unknown model_type '{model_type}'. Supported: {list(registry.keys())}")
        self._processor = processor
        self._model = model.to(device).eval()

    def _run(self, inputs: dict):
        torch = self._torch
        inputs = {k: (v.to(self.device) if hasattr(v, "to") else v) for k, v in inputs.items()}
        with torch.no_grad():
            out = self._model(**inputs)
        # colpali_engine's Col* / Bi* modules return the embedding tensor itself (what the reference indexes with `[0]`,
        # colpali.py:130-133); transformers' own ColPaliForRetrieval / ColQwen2ForRetrieval return an output object that
        # carries it as `.embeddings`
        out = getattr(out, "embeddings", out)
        return out, inputs

    def _text_inputs(self, texts: list[str], query: bool):
        if query and hasattr(self._processor, "process_queries"):
            return self._processor.process_queries(texts)
        return self._processor.process_texts(texts)


class Mi355ColPaliEmbeddings(_EngineBacked, MultiVectorMultiModalEmbedding):
    """ColPali-style late-interaction embeddings (text and page images), same interface as `ColPaliEmbeddings`."""

    SUPPORTED_MODEL_TYPES: ClassVar[list[str]] = list(COL_MODEL_REGISTRY.keys())

    def __init__(self, model_name: str = "vidore/colpali-v1.3", model_type: str = "pali", device: str = "cpu",
                 torch_dtype: Any = "bfloat16", model: Any | None = None, processor: Any | None = None, batch_size: int = 10,
                 drop_padding: bool = False, embed_batch_size: int | None = None):
        batch_size = batch_size if embed_batch_size is None else embed_batch_size   # (the reference's YAML key, colpali.yaml)
        init_multivector_base(self, model_name, batch_size)
        self._setup(COL_MODEL_REGISTRY, "ColPaliEmbeddings", model_name, model_type, device, torch_dtype, model, processor,
                    batch_size)
        # The reference keeps EVERY row the model returns for an item of a batch, the padded positions included
        # (`[emb.cpu().tolist() for emb in embeddings]`, colpali.py:218-245; colpali_engine zeroes them, so a stored doc then
        # carries zero vectors and every query vector's max over it is at least 0).  False (default) = exactly that;
        # True = only the attended positions (ragged, fewer rows to stream; scores differ where all real dots are negative).
        self.drop_padding = drop_padding

    # ---- the model's output as ragged device tensors -------------------------------------------------------------
    def _ragged(self, out, inputs):
        """[n, T, d] (+ optional attention_mask) -> (flat fp32 [sum_T, d], host offsets [n+1]); with `drop_padding` the rows
        of masked positions are left out."""
        torch = self._torch
        h = out.float()
        mask = inputs.get("attention_mask") if getattr(self, "drop_padding", False) else None
        if mask is not None and tuple(mask.shape) == tuple(h.shape[:2]):
            keep = mask.bool()
            lens = keep.sum(dim=1).cpu().numpy().astype(np.int64)
            flat = h[keep]
        else:
            lens = np.full((h.shape[0],), h.shape[1], dtype=np.int64)
            flat = h.reshape(-1, h.shape[-1])
        return flat.contiguous(), np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)

    def encode_texts_to_device(self, texts: list[str], query: bool = False, batch_size: int | None = None):
        """`batch_size` texts per forward (default `embed_batch_size`).  Texts of one forward are padded to its longest, and the
        reference keeps the padded rows: how a list is cut into batches is therefore part of the result."""
        torch = self._torch
        bs = max(1, batch_size or self.embed_batch_size)
        parts, offs = [], [np.zeros((1,), np.int64)]
        for i in range(0, len(texts), bs):
            flat, off = self._ragged(*self._run(self._text_inputs(texts[i: i + bs], query)))
            parts.append(flat)
            offs.append(off[1:] + offs[-1][-1])
        d = parts[0].shape[1] if parts else 0
        return (torch.cat(parts, 0) if parts else torch.empty((0, d), device=self.device)), np.concatenate(offs)

    def encode_images_to_device(self, images: list[Any], batch_size: int | None = None):
        torch = self._torch
        bs = max(1, batch_size or self.embed_batch_size)
        parts, offs = [], [np.zeros((1,), np.int64)]
        for i in range(0, len(images), bs):
            batch = [load_image(p) for p in images[i: i + bs]]
            flat, off = self._ragged(*self._run(self._processor.process_images(batch)))
            parts.append(flat)
            offs.append(off[1:] + offs[-1][-1])
        d = parts[0].shape[1] if parts else 0
        return (torch.cat(parts, 0) if parts else torch.empty((0, d), device=self.device)), np.concatenate(offs)

    def index_images_on_device(self, index: Any, images: list[Any]) -> int:
        """Encode page images and append them to a Mi355Index MaxSim store without a host round trip."""
        flat, off = self.encode_images_to_device(images)
        if flat.is_cuda:
            self._torch.cuda.current_stream(flat.device).synchronize()  # the library works on its own stream
            index.add_multivec_device(flat.data_ptr(), off)
        else:
            index.add_multivec(flat.numpy(), off)
        return len(images)

    @staticmethod
    def _lists(flat, off) -> list[MultiVectorEmbedding]:
        rows = flat.cpu().tolist()
        return [rows[off[i]: off[i + 1]] for i in range(len(off) - 1)]

    # ---- the reference's interface (colpali.py:109-245) -------------------------------------------------------------
    def embed_text(self, text: str) -> MultiVectorEmbedding:
        # reference embed_text: process_queries when the processor has it, else process_texts (colpali.py:120-133)
        return self._lists(*self.encode_texts_to_device([text], query=True))[0]

    async def aembed_text(self, text: str) -> MultiVectorEmbedding:
        return await asyncio.to_thread(self.embed_text, text)

    def embed_query(self, query: str) -> MultiVectorEmbedding:
        return self.embed_text(query)

    async def aembed_query(self, query: str) -> MultiVectorEmbedding:
        return await asyncio.to_thread(self.embed_query, query)

    def embed_image(self, img_file_path: Any) -> MultiVectorEmbedding:
        return self._lists(*self.encode_images_to_device([img_file_path]))[0]

    async def aembed_image(self, img_file_path: Any) -> MultiVectorEmbedding:
        return await asyncio.to_thread(self.embed_image, img_file_path)

    def embed_documents(self, texts: list[str]) -> list[MultiVectorEmbedding]:
        # ONE padded batch, like the reference (colpali.py:189-216); `embed_documents_batch` cuts a list into `embed_batch_size`
        return self._lists(*self.encode_texts_to_device(texts, query=False, batch_size=len(texts))) if texts else []

    def embed_images(self, img_file_paths: list[Any]) -> list[MultiVectorEmbedding]:
        # ONE batch (colpali.py:218-245); `embed_images_batch` cuts
        return self._lists(*self.encode_images_to_device(img_file_paths, batch_size=len(img_file_paths))) if img_file_paths else []


class Mi355BiPaliEmbeddings(_EngineBacked, SingleVectorMultiModalEmbedding):
    """BiPali-style single-vector embeddings (text and page images), same interface as `BiPaliEmbeddings`."""

    SUPPORTED_MODEL_TYPES: ClassVar[list[str]] = list(BI_MODEL_REGISTRY.keys())

    def __init__(self, model_name: str = "vidore/bipali", model_type: str = "pali", device: str = "cpu",
                 torch_dtype: Any = "bfloat16", model: Any | None = None, processor: Any | None = None, batch_size: int = 10,
                 embed_batch_size: int | None = None):
        batch_size = batch_size if embed_batch_size is None else embed_batch_size   # (the reference's field, bipali.py:84)
        self._setup(BI_MODEL_REGISTRY, "BiPaliEmbeddings", model_name, model_type, device, torch_dtype, model, processor,
                    batch_size)

    def encode_texts_to_device(self, texts: list[str], query: bool = False):
        parts = [self._run(self._text_inputs(texts[i: i + self.embed_batch_size], query))[0].float()
                 for i in range(0, len(texts), self.embed_batch_size)]
        return self._torch.cat(parts, 0) if parts else self._torch.empty((0, 0), device=self.device)

    def encode_images_to_device(self, images: list[Any]):
        parts = []
        for i in range(0, len(images), self.embed_batch_size):
            batch = [load_image(p) for p in images[i: i + self.embed_batch_size]]
            parts.append(self._run(self._processor.process_images(batch))[0].float())
        return self._torch.cat(parts, 0) if parts else self._torch.empty((0, 0), device=self.device)

    def index_images_on_device(self, index: Any, images: list[Any]) -> int:
        v = self.encode_images_to_device(images).contiguous()
        if v.is_cuda:
            self._torch.cuda.current_stream(v.device).synchronize()
            index.add_device(v.data_ptr(), v.shape[0])
        else:
            index.add(v.numpy())
        return len(images)

    def embed_query(self, text: str) -> list[float]:
        # the reference's BiPali sends queries through `process_texts` too (bipali.py:113-122, 218-234 `_embed_text`)
        return self.encode_texts_to_device([text], query=False)[0].cpu().tolist()

    def embed_queries(self, texts: list[str]) -> list[list[float]]:
        return self.encode_texts_to_device(texts, query=False).cpu().tolist() if texts else []

    async def aembed_query(self, text: str) -> list[float]:
        return await asyncio.to_thread(self.embed_query, text)

    def embed_documents(self, texts: list[str]) -> list[list[float]]:
        return self.encode_texts_to_device(texts, query=False).cpu().tolist() if texts else []

    async def aembed_documents(self, texts: list[str]) -> list[list[float]]:
        return await asyncio.to_thread(self.embed_documents, texts)

    def embed_image(self, img_file_path: Any) -> list[float]:
        return self.encode_images_to_device([img_file_path])[0].cpu().tolist()

    async def aembed_image(self, img_file_path: Any) -> list[float]:
        return await asyncio.to_thread(self.embed_image, img_file_path)

    def embed_images(self, img_file_paths: list[Any]) -> list[list[float]]:
        return self.encode_images_to_device(img_file_paths).cpu().tolist() if img_file_paths else []


# ---- offline stand-ins for a colpali_engine model + processor (random init; shapes of the `pali` family) --------------
class RandomVisualProcessor:
    """process_images -> {"pixel_values": [n, 3, S, S]}; process_queries / process_texts -> {"input_ids", "attention_mask"}.
    Images: PIL images, [H, W, 3] uint8 arrays or [3, H, W] float tensors; resized to S x S (bilinear)."""

    def __init__(self, image_size: int = 448, vocab: int = 4096, max_length: int = 64, query_prefix_tokens: int = 0):
        self.image_size, self.vocab, self.max_length, self.query_prefix_tokens = image_size, vocab, max_length, query_prefix_tokens

    def process_images(self, images: list[Any]) -> dict:
        import torch  # noqa: PLC0415

        out = []
        for im in images:
            a = im if hasattr(im, "dim") else torch.from_numpy(np.array(im))
            if a.dim() == 3 and a.shape[-1] == 3:
                a = a.permute(2, 0, 1)
            a = a.float() / (255.0 if a.max() > 1.5 else 1.0)
            out.append(torch.nn.functional.interpolate(a[None], size=(self.image_size, self.image_size), mode="bilinear",
                                                       align_corners=False)[0])
        return {"pixel_values": torch.stack(out)}

    def _tok(self, texts: list[str], prefix: int) -> dict:
        import zlib  # noqa: PLC0415

        import torch  # noqa: PLC0415

        ids = [([1] * prefix + [2 + zlib.crc32(w.encode()) % (self.vocab - 2) for w in t.split()])[: self.max_length] or [1]
               for t in texts]
        L = max(len(x) for x in ids)
        tok = torch.zeros((len(ids), L), dtype=torch.long)
        mask = torch.zeros((len(ids), L), dtype=torch.long)
        for i, x in enumerate(ids):
            tok[i, : len(x)], mask[i, : len(x)] = torch.tensor(x), 1
        return {"input_ids": tok, "attention_mask": mask}

    def process_queries(self, texts: list[str]) -> dict:
        return self._tok(texts, self.query_prefix_tokens)

    def process_texts(self, texts: list[str]) -> dict:
        return self._tok(texts, 0)


def make_random_col_model(dim: int = 128, patch: int = 14, image_size: int = 448, prefix_tokens: int = 6, vocab: int = 4096,
                          seed: int = 0, pooled: bool = False):
    """A torch module with the call shape of a colpali_engine Col* (pooled=False: [n, T, dim] per-token / per-patch, L2
    normalised, T = (image_size / patch)^2 + prefix_tokens = 1030 for the defaults) or Bi* (pooled=True: [n, dim]) model."""
    import torch  # noqa: PLC0415

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.patchify = torch.nn.Conv2d(3, dim, kernel_size=patch, stride=patch, bias=False)
            self.prefix = torch.nn.Parameter(torch.randn((prefix_tokens, dim), generator=g))
            self.emb = torch.nn.Embedding(vocab, dim)
            with torch.no_grad():
                self.patchify.weight.copy_(torch.randn(self.patchify.weight.shape, generator=g) * 0.05)
                self.emb.weight.copy_(torch.randn(self.emb.weight.shape, generator=g))

        def forward(self, pixel_values=None, input_ids=None, attention_mask=None):
            if pixel_values is not None:
                h = self.patchify(pixel_values.to(self.patchify.weight.dtype)).flatten(2).transpose(1, 2)   # [n, P, dim]
                h = torch.cat([self.prefix[None].expand(h.shape[0], -1, -1).to(h.dtype), h], dim=1)
                m = None
            else:
                h, m = self.emb(input_ids), attention_mask
            if pooled:
                if m is not None:
                    h = (h * m[..., None].to(h.dtype)).sum(1) / m.sum(1, keepdim=True).clamp(min=1).to(h.dtype)
                else:
                    h = h.mean(1)
                return torch.nn.functional.normalize(h.float(), dim=-1)
            h = torch.nn.functional.normalize(h.float(), dim=-1)
            if m is not None:  # colpali_engine's Col* heads zero the padded positions (`proj * attention_mask.unsqueeze(-1)`)
                h = h * m[..., None].to(h.dtype)
            return h

    return _M()


__all__ = ["Mi355ColPaliEmbeddings", "Mi355BiPaliEmbeddings", "RandomVisualProcessor", "make_random_col_model", "load_image",
           "COL_MODEL_REGISTRY", "BI_MODEL_REGISTRY"]
