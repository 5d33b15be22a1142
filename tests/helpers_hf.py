"""Shared by tests/test_hf_models.py and tests/golden/make_golden.py: transformers' own ColPaliForRetrieval from a small random
config, the tensors ColPaliProcessor would hand it, and a module named `colpali_engine` that serves both behind
colpali_engine's call shape (forward returns the embedding tensor; `from_pretrained(name, dtype=, trust_remote_code=)`).

colpali_engine is not installed and no checkpoint is reachable offline: the stand-in is for an ABSENT DEPENDENCY -- the classes
under test (the reference's ColPaliEmbeddings / BiPaliEmbeddings when the fixture is generated, Mi355ColPaliEmbeddings /
Mi355BiPaliEmbeddings in the tests) are the real ones."""

from __future__ import annotations

import io
import sys
import types
import zlib

import numpy as np

IMAGE_TOKEN, VOCAB, IMG, PATCH = 500, 512, 56, 14
N_IMG_TOK = (IMG // PATCH) ** 2


def colpali_config():
    from transformers import ColPaliConfig, GemmaConfig, PaliGemmaConfig, SiglipVisionConfig

    vis = SiglipVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=IMG,
                             patch_size=PATCH, projection_dim=96)
    txt = GemmaConfig(vocab_size=VOCAB, hidden_size=96, intermediate_size=192, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=1, head_dim=24, max_position_embeddings=256)
    vlm = PaliGemmaConfig(vision_config=vis, text_config=txt, image_token_index=IMAGE_TOKEN, vocab_size=VOCAB, projection_dim=96,
                          hidden_size=96)
    return ColPaliConfig(vlm_config=vlm, embedding_dim=128)


class PaliInputs:
    """The tensors ColPaliProcessor hands the model: images -> N_IMG_TOK image-token placeholders + a short text suffix and
    `pixel_values`; texts -> right-padded ids + attention_mask (ids hashed from the words: no tokenizer files offline)."""

    def process_images(self, images):
        import torch

        px = []
        for im in images:
            a = torch.as_tensor(np.array(im)).permute(2, 0, 1).float() / 255.0
            px.append(torch.nn.functional.interpolate(a[None], size=(IMG, IMG), mode="bilinear", align_corners=False)[0])
        ids = torch.full((len(images), N_IMG_TOK + 3), IMAGE_TOKEN, dtype=torch.long)
        ids[:, N_IMG_TOK:] = torch.tensor([2, 7, 9])        # <bos> "Describe the image." stand-in
        return {"input_ids": ids, "pixel_values": torch.stack(px), "attention_mask": torch.ones_like(ids)}

    def _tok(self, texts, first=2):
        import torch

        rows = [[first] + [10 + zlib.crc32(w.encode()) % 400 for w in t.split()] for t in texts]
        L = max(len(r) for r in rows)
        ids, mask = torch.zeros((len(rows), L), dtype=torch.long), torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)], mask[i, : len(r)] = torch.tensor(r), 1
        return {"input_ids": ids, "attention_mask": mask}

    def process_queries(self, texts):
        return self._tok(texts)

    def process_texts(self, texts):
        return self._tok(texts)


def images(n, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 255, size=(60 + 6 * i, 80, 3), dtype=np.uint8) for i in range(n)]


def png_bytes(arr) -> bytes:
    """Lossless PNG of an [H, W, 3] uint8 array: what the reference's `load_image` (util.py:318-342) accepts as `bytes`."""
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(np.asarray(arr)).save(buf, format="PNG")
    return buf.getvalue()


def install_colpali_engine(monkeypatch=None, seen: dict | None = None):
    """Register modules `colpali_engine` / `colpali_engine.models` with ColPali / ColPaliProcessor (multi-vector) and BiPali /
    BiPaliProcessor (single-vector: the L2-normalised masked mean of the same model's token vectors -- [n, 128]).
    The query side of the processors starts with a different first token than the document side, so that a wrapper that
    confuses `process_queries` with `process_texts` is caught.  Returns the models module."""
    import torch
    from transformers import ColPaliForRetrieval

    seen = {} if seen is None else seen

    class ColPali(ColPaliForRetrieval):
        @classmethod
        def from_pretrained(cls, name, dtype=None, trust_remote_code=False, **kw):
            seen.update(name=name, dtype=dtype, trust_remote_code=trust_remote_code)
            torch.manual_seed(0)
            return cls(colpali_config()).to(dtype)

        def forward(self, *a, **kw):
            return super().forward(*a, **kw).embeddings

    class BiPali(ColPali):
        def forward(self, *a, **kw):
            h = super().forward(*a, **kw).float()
            m = kw.get("attention_mask")
            if m is not None:
                h = (h * m[..., None].to(h.dtype)).sum(1) / m.sum(1, keepdim=True).clamp(min=1).to(h.dtype)
            else:
                h = h.mean(1)
            return torch.nn.functional.normalize(h, dim=-1)

    class ColPaliProcessor(PaliInputs):
        @classmethod
        def from_pretrained(cls, name):
            seen["processor"] = name
            return cls()

        def process_queries(self, texts):
            return self._tok(texts, first=3)

    class BiPaliProcessor(ColPaliProcessor):
        pass

    eng, models = types.ModuleType("colpali_engine"), types.ModuleType("colpali_engine.models")
    models.ColPali, models.ColPaliProcessor = ColPali, ColPaliProcessor
    models.BiPali, models.BiPaliProcessor = BiPali, BiPaliProcessor
    eng.models = models
    if monkeypatch is not None:
        monkeypatch.setitem(sys.modules, "colpali_engine", eng)
        monkeypatch.setitem(sys.modules, "colpali_engine.models", models)
    else:
        sys.modules["colpali_engine"], sys.modules["colpali_engine.models"] = eng, models
    return models


EMBED_TEXTS = ["which page shows the revenue chart", "x", "a longer passage about row sharded top k merge over xgmi links",
               "late interaction", "one two three", "alpha beta", "gamma", "delta epsilon zeta eta", "theta", "iota kappa",
               "lambda mu nu xi omicron pi rho", "sigma"]
