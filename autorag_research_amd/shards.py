"""fp32 shard files: the on-disk form of a table's vectors that the GPU index loads without a database.

SURVEY section 8(f) row 3: the reference keeps vectors in PostgreSQL (`VECTOR(d)` / `VECTOR(d)[]` columns, read back as
text and parsed per row: orm/repository/base.py:608-619, orm/types.py:234-277) and ships pre-embedded datasets as SQL
dumps.  A shard directory holds the same content in the layout the library consumes:

    meta.json            {"format": "mi355dr-shard-1", "n": rows, "dim": d, "id_type": "int"|"str",
                          "has_embedding": bool, "has_multivec": bool}
    ids.json             primary keys in row order (BIGINT or VARCHAR, schema_factory.py:63-76)
    contents.json        optional text per row (null for image chunks)
    embedding.npy        [n, d] fp32 row-major; a row of NaN = NULL embedding
    mv_tokens.npy        [sum_T, d] fp32, mv_offsets.npy [n+1] int64; an empty span = NULL embeddings

`.npy` files are memory-mapped and uploaded in row chunks, so a 30 GB corpus never needs 30 GB of host memory.
"""

from __future__ import annotations

import json
from pathlib import Path
from typing import Any

import numpy as np

from .store import ChunkTable

FORMAT = "mi355dr-shard-1"


def write_shard(directory: str | Path, table: ChunkTable) -> Path:
    """Write one table (chunk or image_chunk) as a shard directory; returns the directory."""
    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    n = len(table.ids)
    dim = 0
    if table.embedding is not None:
        dim = int(table.embedding.shape[1])
        np.save(d / "embedding.npy", np.ascontiguousarray(table.embedding, dtype=np.float32))
    if table.mv_offsets is not None and table.mv_tokens is not None:
        dim = dim or int(table.mv_tokens.shape[1])
        np.save(d / "mv_tokens.npy", np.ascontiguousarray(table.mv_tokens, dtype=np.float32))
        np.save(d / "mv_offsets.npy", np.ascontiguousarray(table.mv_offsets, dtype=np.int64))
    (d / "ids.json").write_text(json.dumps(list(table.ids)))
    if any(c is not None for c in table.contents):
        (d / "contents.json").write_text(json.dumps(list(table.contents)))
    meta = {"format": FORMAT, "n": n, "dim": dim, "id_type": "str" if any(isinstance(i, str) for i in table.ids) else "int",
            "has_embedding": table.embedding is not None, "has_multivec": table.mv_offsets is not None}
    (d / "meta.json").write_text(json.dumps(meta))
    return d


def read_meta(directory: str | Path) -> dict[str, Any]:
    f = Path(directory) / "meta.json"
    if not f.exists():
        raise ValueError(f"{directory}: not a {FORMAT} shard (no meta.json)")
    meta = json.loads(f.read_text())
    if meta.get("format") != FORMAT:
        raise ValueError(f"{directory}: not a {FORMAT} shard")
    return meta


def read_shard(directory: str | Path, mmap: bool = True) -> ChunkTable:
    """Load a shard as a ChunkTable; vector arrays stay memory-mapped (`mmap=True`) until something copies them."""
    d = Path(directory)
    meta = read_meta(d)
    t = ChunkTable()
    t.ids = json.loads((d / "ids.json").read_text())
    t.contents = json.loads((d / "contents.json").read_text()) if (d / "contents.json").exists() else [None] * len(t.ids)
    mode = "r" if mmap else None
    if meta["has_embedding"]:
        t.embedding = np.load(d / "embedding.npy", mmap_mode=mode)
    if meta["has_multivec"]:
        t.mv_tokens = np.load(d / "mv_tokens.npy", mmap_mode=mode)
        t.mv_offsets = np.load(d / "mv_offsets.npy")
    if len(t.ids) != meta["n"]:
        raise ValueError(f"{directory}: ids.json has {len(t.ids)} rows, meta says {meta['n']}")
    return t


def load_embedding_into(index: Any, directory: str | Path, chunk_rows: int = 1 << 18, row_range: tuple[int, int] | None = None):
    """Stream `embedding.npy` rows [lo, hi) into `index` (Mi355Index.add) in chunks; NULL (all-NaN) rows are skipped like
    `WHERE embedding IS NOT NULL`.  Returns the table positions of the rows added (index row i = table row map[i])."""
    d = Path(directory)
    meta = read_meta(d)
    if not meta["has_embedding"]:
        raise ValueError(f"{directory}: no single-vector embeddings")
    emb = np.load(d / "embedding.npy", mmap_mode="r")
    lo, hi = row_range if row_range is not None else (0, emb.shape[0])
    kept: list[np.ndarray] = []
    for a in range(lo, hi, chunk_rows):
        b = min(hi, a + chunk_rows)
        block = np.ascontiguousarray(emb[a:b], dtype=np.float32)  # the only host copy: one chunk
        live = ~np.isnan(block).all(axis=1)
        if not live.all():
            block = block[live]
        if block.shape[0]:
            index.add(block)
        kept.append(np.nonzero(live)[0] + a)
    return np.concatenate(kept) if kept else np.zeros(0, dtype=np.int64)


def load_multivec_into(index: Any, directory: str | Path, chunk_docs: int = 1 << 14, doc_range: tuple[int, int] | None = None):
    """Stream the ragged store of docs [lo, hi) into `index` (Mi355Index.add_multivec) in chunks of docs."""
    d = Path(directory)
    meta = read_meta(d)
    if not meta["has_multivec"]:
        raise ValueError(f"{directory}: no multi-vector embeddings")
    tok = np.load(d / "mv_tokens.npy", mmap_mode="r")
    off = np.load(d / "mv_offsets.npy")
    lo, hi = doc_range if doc_range is not None else (0, off.shape[0] - 1)
    for a in range(lo, hi, chunk_docs):
        b = min(hi, a + chunk_docs)
        index.add_multivec(np.ascontiguousarray(tok[off[a]:off[b]], dtype=np.float32), off[a:b + 1] - off[a])
    return hi - lo


__all__ = ["write_shard", "read_shard", "read_meta", "load_embedding_into", "load_multivec_into", "FORMAT"]
