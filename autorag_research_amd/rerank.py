"""ColBERT reranker on the MI355X MaxSim kernel (reference autorag_research/rerankers/colbert.py).

The reference scores each (query, document) pair with `_maxsim_score` (colbert.py:63-84): L2-normalised token embeddings,
`sim = Q @ D^T`, padding tokens of the document masked to -inf, max over document tokens, `clamp(min=0)`, padding tokens of
the query multiplied out, mean over the VALID query tokens -- a different normalisation from the VectorChord `@#` operator
(no clamp there, sum instead of mean).  Here the valid tokens of all candidate documents go into a scratch multi-vector
store by device pointer (`mi355dr_add_multivec_device`), and ONE `mi355dr_maxsim_subset_ex(..., MI355DR_MAXSIM_CLAMP0)`
call scores the query against every candidate with the exact fp32 MFMA kernel; score = -distance / n_valid_query_tokens.

Same surface as the reference's reranker (`rerank`, `arerank`, `rerank_documents`, RerankResult(index, text, score), results
sorted by score descending, stable); the token encoder is the caller's (`encode(texts) -> (emb [n, L, d], mask [n, L])`,
what `ColBERTReranker._encode` returns).  No checkpoint is reachable offline, so tests use a random-init encoder and pin
the scoring against the reference's `_maxsim_score` on seeded tensors (tests/golden/rerank_golden.npz).
"""

from __future__ import annotations

import asyncio
from dataclasses import dataclass
from typing import Any

import numpy as np

from .index import Mi355Index


@dataclass
class RerankResult:
    """Single reranked document result (reference rerankers/base.py:11-18)."""

    index: int
    text: str
    score: float


def _to_numpy(x) -> np.ndarray:
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def colbert_maxsim_scores(query_emb, query_mask, doc_embs, doc_masks, device: int = 0, index_factory=None) -> np.ndarray:
    """`_maxsim_score` of one query against n documents, on padded inputs like the reference's (`query_emb` [1, Lq, d] or
    [Lq, d], `query_mask` [1, Lq] or [Lq]; `doc_embs` [n, Ld, d], `doc_masks` [n, Ld]; numpy arrays or torch tensors, on
    any device).  Returns float64 scores [n].  A document without valid tokens scores 0 (every max is -inf -> clamp),
    a query without valid tokens NaN (0 / 0), as in the reference."""
    is_dev = hasattr(doc_embs, "is_cuda") and doc_embs.is_cuda
    qm = _to_numpy(query_mask).reshape(-1).astype(bool)
    q = _to_numpy(query_emb).reshape(qm.shape[0], -1).astype(np.float32)[qm]
    dm = _to_numpy(doc_masks).astype(bool)
    n, d = dm.shape[0], q.shape[1] if q.size else int(doc_embs.shape[-1])
    n_valid = int(qm.sum())
    if n == 0:
        return np.zeros((0,), dtype=np.float64)
    if n_valid == 0:
        return np.full((n,), np.nan)
    lens = dm.sum(axis=1).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    factory = index_factory or Mi355Index
    with factory(d, "cosine", device) as ix:
        if is_dev and hasattr(ix, "add_multivec_device"):
            import torch  # noqa: PLC0415

            flat = doc_embs.reshape(-1, d)[torch.as_tensor(dm.reshape(-1), device=doc_embs.device)].to(torch.float32).contiguous()
            torch.cuda.current_stream(flat.device).synchronize()  # the library works on its own stream
            ix.add_multivec_device(flat.data_ptr(), off)
        else:
            ix.add_multivec(_to_numpy(doc_embs).reshape(-1, d).astype(np.float32)[dm.reshape(-1)], off)
        dist = ix.maxsim_subset(q, np.array([0, n_valid], dtype=np.int32), np.arange(n, dtype=np.int64)[None, :], clamp0=True)[0]
    score = -dist.astype(np.float64) / n_valid
    return np.where(np.isnan(dist), 0.0, score)


class Mi355ColBERTReranker:
    """Drop-in for `ColBERTReranker` (rerankers/colbert.py:16-120) with the scoring on the GPU."""

    def __init__(self, encoder: Any, model_name: str = "colbert-ir/colbertv2.0", device: int = 0, max_length: int = 512,
                 batch_size: int = 64, index_factory=None):
        self.encoder, self.model_name, self.device = encoder, model_name, device
        self.max_length, self.batch_size, self._index_factory = max_length, batch_size, index_factory

    def _encode(self, texts: list[str]):
        """(token embeddings [n, L, d] L2-normalised, attention mask [n, L]) -- reference `_encode` (colbert.py:45-61)."""
        return self.encoder.encode(texts)

    def rerank(self, query: str, documents: list[str], top_k: int | None = None) -> list[RerankResult]:
        if not documents:
            return []
        top_k = min(top_k or len(documents), len(documents))
        q_emb, q_mask = self._encode([query])
        d_emb, d_mask = self._encode(documents)
        scores = colbert_maxsim_scores(q_emb, q_mask, d_emb, d_mask, self.device, self._index_factory)
        results = [RerankResult(index=i, text=documents[i], score=float(s)) for i, s in enumerate(scores)]
        results.sort(key=lambda r: r.score, reverse=True)  # stable, like the reference's
        return results[:top_k]

    async def arerank(self, query: str, documents: list[str], top_k: int | None = None) -> list[RerankResult]:
        return await asyncio.get_running_loop().run_in_executor(None, self.rerank, query, documents, top_k)

    def rerank_documents(self, queries: list[str], documents_list: list[list[str]], top_k: int | None = None):
        return [self.rerank(q, docs, top_k) for q, docs in zip(queries, documents_list, strict=True)]


class RandomTokenEncoder:
    """Offline stand-in for a ColBERT checkpoint: hashed word-piece ids -> a fixed random embedding table, L2-normalised,
    padded to the longest text of the batch (what `AutoModel(...).last_hidden_state` + normalise yields, shape-wise)."""

    def __init__(self, dim: int = 128, vocab: int = 4096, seed: int = 0, device: str | None = None, max_length: int = 512):
        import torch  # noqa: PLC0415

        self._torch = torch
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        g = torch.Generator().manual_seed(seed)
        self.table = torch.nn.functional.normalize(torch.randn((vocab, dim), generator=g), dim=-1).to(self.device)
        self.vocab, self.max_length = vocab, max_length

    def encode(self, texts: list[str]):
        import zlib  # noqa: PLC0415

        torch = self._torch
        ids = [[zlib.crc32(w.encode()) % self.vocab for w in t.split()][: self.max_length] for t in texts]
        L = max(1, max(len(x) for x in ids))
        tok = torch.zeros((len(texts), L), dtype=torch.long)
        mask = torch.zeros((len(texts), L), dtype=torch.long)
        for i, x in enumerate(ids):
            tok[i, : len(x)] = torch.tensor(x, dtype=torch.long)
            mask[i, : len(x)] = 1
        return self.table[tok.to(self.device)], mask.to(self.device)


__all__ = ["RerankResult", "Mi355ColBERTReranker", "RandomTokenEncoder", "colbert_maxsim_scores"]
