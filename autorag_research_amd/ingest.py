"""Batched embedding of un-embedded rows (the encode half of the path).

Mirrors BaseIngestionService._embed_entities / embed_all_queries / embed_all_chunks
(autorag_research/orm/service/base_ingestion.py:326-495, 542-624) and TextEmbeddingDataIngestor.embed_all /
embed_all_late_interaction (autorag_research/data/base.py:57-72, 110-125): find rows whose embedding is NULL,
embed them, store the vectors.  The reference embeds ONE text per model forward (`aembed_query` under a semaphore) -- and it
feeds the model's QUERY side to queries AND chunks (data/base.py:63-71): for asymmetric encoders (query / passage
prefixes, instructions) the stored vectors, hence rankings and nDCG, depend on that.  The default `side="query"` here
does the same, batched through `embed_queries` when the model offers it; `side="document"` (the model's
`embed_documents`) is an explicit deviation for callers that want passage-side chunk vectors.  Multi-vector models
fill the ragged store.  Returns the number of rows embedded, like the reference.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .embeddings import MultiVectorBaseEmbedding
from .store import ChunkTable, InMemoryStore


def _embed_texts(model: Any, texts: list[str], batch_size: int, side: str = "query") -> list:
    if side not in ("query", "document"):
        raise ValueError("side must be 'query' (the reference's behaviour) or 'document'")
    out: list = []
    for i in range(0, len(texts), batch_size):
        part = texts[i: i + batch_size]
        if side == "document":
            out.extend(model.embed_documents(part))
        elif hasattr(model, "embed_queries"):
            out.extend(model.embed_queries(part))
        else:
            out.extend(model.embed_query(t) for t in part)
    return out


def embed_all_chunks(store: InMemoryStore, model: Any, batch_size: int = 128, unit: str = "chunk",
                     side: str = "query") -> int:
    """Fill `embedding` (single-vector model) or `embeddings` (multi-vector model) of every row that lacks it."""
    table: ChunkTable = store.image_chunks if unit == "image_chunk" else store.chunks
    n = len(table)
    if n == 0:
        return 0
    texts = [c if c is not None else "" for c in table.contents]
    if isinstance(model, MultiVectorBaseEmbedding):
        have = table.mv_offsets is not None
        todo = [i for i in range(n) if not have or table.mv_offsets[i + 1] == table.mv_offsets[i]]
        if not todo:
            return 0
        new = _embed_texts(model, [texts[i] for i in todo], batch_size, side)
        docs = [None] * n
        if have:
            for i in range(n):
                if table.mv_offsets[i + 1] > table.mv_offsets[i]:
                    docs[i] = table.mv_tokens[table.mv_offsets[i]: table.mv_offsets[i + 1]]
        for i, m in zip(todo, new):
            docs[i] = np.asarray(m, dtype=np.float32)
        d = next(x.shape[1] for x in docs if x is not None)
        lens = [0 if x is None else x.shape[0] for x in docs]
        table.mv_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        table.mv_tokens = (np.concatenate([x for x in docs if x is not None], axis=0)
                           if any(x is not None for x in docs) else np.zeros((0, d), np.float32))
        return len(todo)
    if table.embedding is None:
        todo = list(range(n))
    else:
        todo = [i for i in range(n) if np.isnan(table.embedding[i]).all()]
    if not todo:
        return 0
    vecs = np.asarray(_embed_texts(model, [texts[i] for i in todo], batch_size, side), dtype=np.float32)
    if table.embedding is None:
        table.embedding = np.full((n, vecs.shape[1]), np.nan, dtype=np.float32)
    table.embedding[todo] = vecs
    return len(todo)


def index_texts_on_device(model: Any, texts: list[str], index: Any, batch_size: int = 1024) -> int:
    """Encode `texts` and append the vectors to `index` without leaving the GPU.

    `model` is a single-vector encoder with `encode_to_device(texts) -> fp32 [n, d] tensor on the index's device`
    (TorchEncoderEmbeddings); each batch's tensor is handed to `Mi355Index.add_device` by pointer
    (mi355dr_add_rows_device: device-to-device copy + norms + shadows on the library's stream).  The reference's path for
    the same work is model -> `.tolist()` -> per-row SQL UPDATE with a text literal (base_ingestion.py:199-247, 326-495).
    Returns the number of rows added.
    """
    import torch

    done = 0
    for i in range(0, len(texts), batch_size):
        v = model.encode_to_device(texts[i: i + batch_size])
        if v.dtype != torch.float32 or not v.is_contiguous():
            v = v.float().contiguous()
        if v.shape[0] == 0:
            continue
        if v.shape[1] != index.dim:
            raise ValueError(f"encoder output dim {v.shape[1]} != index dim {index.dim}")
        torch.cuda.current_stream(v.device).synchronize()  # the library copies on its own stream
        index.add_device(v.data_ptr(), v.shape[0])
        done += v.shape[0]
    return done


def embed_all_queries(store: InMemoryStore, model: Any, batch_size: int = 128) -> int:
    multi = isinstance(model, MultiVectorBaseEmbedding)
    todo = [q for q in store.query_order
            if (store.queries[q].embeddings if multi else store.queries[q].embedding) is None]
    if not todo:
        return 0
    texts = [store.queries[q].contents or "" for q in todo]
    out = _embed_texts(model, texts, batch_size, "query")
    for q, v in zip(todo, out):
        if multi:
            store.queries[q].embeddings = np.asarray(v, dtype=np.float32)
        else:
            store.queries[q].embedding = np.asarray(v, dtype=np.float32)
    return len(todo)
