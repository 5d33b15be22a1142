"""PostgreSQL text formats of the vector columns <-> fp32 arrays / shard files (the DB side of the bridge).

No PostgreSQL runs in this image, so what is built -- and pinned on the reference's own converters -- is the TEXT layer a DB
export / import goes through:

  VECTOR(d)     '[x1,x2,...]'                          pgvector's output format = orm/repository/base.py:54-63 `_vec_to_pg_literal`
  VECTOR(d)[]   '{"[x1,...]","[y1,...]"}', '{}' empty   orm/types.py:210-231 `process_bind_param`, :234-277 `process_result_value`
  SQL literal   ARRAY['[..]'::vector, ...]              orm/repository/base.py:66-76 `_vecs_to_pg_array` (the `@#` operand)

and the two bulk paths around them:

  * `copy_text_to_shard`: the rows of `COPY (SELECT id, contents, embedding, embeddings FROM chunk) TO STDOUT` (text format:
    tab-separated, `\\N` = NULL, backslash escapes) -> a shard directory (shards.py) the GPU index loads memory-mapped --
    the bulk form of `BaseEmbeddingRepository.get_with_embeddings` (base.py:608-619), which parses one Python list per row;
  * `results_to_copy_text`: a page of ranked results -> `COPY chunk_retrieved_result (query_id, pipeline_id, chunk_id,
    rel_score) FROM STDIN` rows, the bulk form of `bulk_insert` of the reference's row dicts
    (orm/service/retrieval_pipeline.py:171-181, orm/repository/chunk_retrieved_result.py:116-127).
  * `restore_script_rows`: a published pre-embedded dataset is a pg_dump CUSTOM-format archive (`DBConnection.restore_database`,
    orm/connection.py:298; data/hf_storage.py downloads it).  `pg_restore -f -` turns such an archive into a plain SQL script
    WITHOUT a server; the table's data is its `COPY <schema>.<table> (<every column>) FROM stdin;` block.  This reads that block
    out of the script and projects it onto (id, contents, embedding, embeddings), whatever the column order:
        pg_restore -f - --data-only -t chunk dataset.dump | python -m autorag_research_amd.pgtext --table chunk --out shards/chunk
Fixtures: tests/golden/pgtext_golden.json (strings produced / parsed by the imported reference converters).
"""

from __future__ import annotations

import re
from collections.abc import Iterable
from pathlib import Path
from typing import Any

import numpy as np

from .store import ChunkTable

_VEC_RE = re.compile(r"\[([^\]]*)\]")


def format_vector(vec) -> str:
    """'[x1,x2,...]' with Python float repr per component (reference `_vec_to_pg_literal`)."""
    return "[" + ",".join(str(float(x)) for x in vec) + "]"


def format_vector_array(vecs) -> str | None:
    """VECTOR(d)[] bind text (reference VectorArray.process_bind_param): None -> None, [] -> '{}'."""
    if vecs is None:
        return None
    if len(vecs) == 0:
        return "{}"
    return "{" + ",".join('"' + format_vector(v) + '"' for v in vecs) + "}"


def format_vector_array_sql(vecs) -> str:
    """ARRAY['[..]'::vector,...] (reference `_vecs_to_pg_array`, the query-side operand of `@#`)."""
    return "ARRAY[" + ",".join(f"'{format_vector(v)}'::vector" for v in vecs) + "]"


def parse_vector(text: str | None) -> np.ndarray | None:
    """'[x1,x2,...]' -> fp32 [d]; None / '\\N' -> None (NULL)."""
    if text is None:
        return None
    t = text.strip()
    if t == r"\N" or t == "":
        return None
    if not (t.startswith("[") and t.endswith("]")):
        raise ValueError(f"not a vector literal: {text[:40]!r}")
    body = t[1:-1].strip()
    if not body:
        return np.zeros((0,), dtype=np.float32)
    return np.asarray(body.split(","), dtype=np.float64).astype(np.float32)


def parse_vector_array(value: Any) -> np.ndarray | None:
    """What VectorArray.process_result_value accepts -> fp32 [n, d] (None for NULL, shape (0, 0) for '{}'):
    the string form '{"[..]","[..]"}' or an already-parsed list of lists / arrays (psycopg)."""
    if value is None:
        return None
    if isinstance(value, (list, tuple)):
        rows = [np.asarray(v.tolist() if hasattr(v, "tolist") else list(v), dtype=np.float64) for v in value]
        return np.stack(rows).astype(np.float32) if rows else np.zeros((0, 0), dtype=np.float32)
    if isinstance(value, str):
        t = value.strip()
        if t == r"\N":
            return None
        if t == "{}":
            return np.zeros((0, 0), dtype=np.float32)
        rows = [np.asarray(m.split(","), dtype=np.float64) for m in _VEC_RE.findall(t)]
        return np.stack(rows).astype(np.float32) if rows else np.zeros((0, 0), dtype=np.float32)
    return None


# ---- COPY text format -----------------------------------------------------------------------------------------------
_UNESC = {"b": "\b", "f": "\f", "n": "\n", "r": "\r", "t": "\t", "v": "\v", "\\": "\\"}


def _copy_unescape(field: str) -> str | None:
    if field == r"\N":
        return None
    if "\\" not in field:
        return field
    out, i = [], 0
    while i < len(field):
        c = field[i]
        if c == "\\" and i + 1 < len(field):
            out.append(_UNESC.get(field[i + 1], field[i + 1]))
            i += 2
        else:
            out.append(c)
            i += 1
    return "".join(out)


def _copy_escape(v: Any) -> str:
    if v is None:
        return r"\N"
    s = str(v)
    return s.replace("\\", "\\\\").replace("\t", "\\t").replace("\n", "\\n").replace("\r", "\\r")


def copy_text_to_table(lines: Iterable[str], id_type: str = "int", has_contents: bool = True) -> ChunkTable:
    """Rows of `COPY (SELECT id[, contents], embedding, embeddings FROM chunk|image_chunk) TO STDOUT` -> ChunkTable
    (NULL embedding -> a NaN row, NULL / empty embeddings -> an empty span, exactly what the SQL `IS NOT NULL` filters skip)."""
    ids, contents, single, multi = [], [], [], []
    for line in lines:
        line = line.rstrip("\n")
        if not line or line == r"\.":
            continue
        f = line.split("\t")
        want = 4 if has_contents else 3
        if len(f) != want:
            raise ValueError(f"expected {want} tab-separated fields, got {len(f)}: {line[:60]!r}")
        pk = _copy_unescape(f[0])
        ids.append(int(pk) if id_type == "int" else pk)
        contents.append(_copy_unescape(f[1]) if has_contents else None)
        single.append(parse_vector(_copy_unescape(f[-2])))
        multi.append(parse_vector_array(_copy_unescape(f[-1])))
    t = ChunkTable(ids=ids, contents=contents)
    d1 = next((v.shape[0] for v in single if v is not None), 0)
    if d1:
        t.embedding = np.full((len(ids), d1), np.nan, dtype=np.float32)
        for i, v in enumerate(single):
            if v is not None:
                if v.shape[0] != d1:
                    raise ValueError(f"row {ids[i]!r}: embedding has {v.shape[0]} dims, expected {d1}")
                t.embedding[i] = v
    live = [m for m in multi if m is not None and m.shape[0]]
    if live:
        dm = live[0].shape[1]
        lens = [0 if m is None else m.shape[0] for m in multi]
        t.mv_offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        t.mv_tokens = np.concatenate([m.reshape(-1, dm) for m in live], axis=0)
    return t


def copy_text_to_shard(lines: Iterable[str], directory: str | Path, id_type: str = "int", has_contents: bool = True) -> Path:
    from .shards import write_shard  # noqa: PLC0415

    return write_shard(directory, copy_text_to_table(lines, id_type, has_contents))


_COPY_HEAD = re.compile(r'^COPY\s+(?:(?:"[^"]+"|\w+)\.)?(?:"([^"]+)"|(\w+))\s*\((.*)\)\s+FROM\s+stdin;\s*$', re.IGNORECASE)


def restore_script_rows(lines: Iterable[str], table: str = "chunk",
                        columns: tuple[str, ...] = ("id", "contents", "embedding", "embeddings")) -> Iterable[str]:
    """The data rows of `table` in a `pg_restore -f -` / `pg_dump --format=plain` script, re-projected onto `columns`
    (tab-separated COPY text, in that order): what `copy_text_to_table` / `copy_text_to_shard` read.  The block runs from
    `COPY [schema.]table (col, ...) FROM stdin;` to the `\\.` line; a script without the table, or a table without one of the
    columns, is an error (for `image_chunk` pass columns=("id", "embedding", "embeddings") and has_contents=False downstream)."""
    it = iter(lines)
    for line in it:
        m = _COPY_HEAD.match(line.rstrip("\n"))
        if not m or (m.group(1) or m.group(2)) != table:
            continue
        cols = [c.strip().strip('"') for c in m.group(3).split(",")]
        missing = [c for c in columns if c not in cols]
        if missing:
            raise ValueError(f"table {table!r} has no column(s) {missing} (its COPY block lists {cols})")
        take = [cols.index(c) for c in columns]
        for row in it:
            row = row.rstrip("\n")
            if row == r"\.":
                return
            f = row.split("\t")   # (COPY text escapes a tab inside a value as \t: a raw split is the field split)
            if len(f) != len(cols):
                raise ValueError(f"expected {len(cols)} fields in the COPY block of {table!r}, got {len(f)}: {row[:60]!r}")
            yield "\t".join(f[i] for i in take)
        raise ValueError(f"the COPY block of {table!r} does not end with a \\. line (truncated script?)")
    raise ValueError(f"no `COPY ... {table} (...) FROM stdin;` block in the script")


def table_to_copy_text(table: ChunkTable) -> list[str]:
    """The inverse (id, contents, embedding, embeddings) rows: what a pre-embedded dataset dump carries."""
    out = []
    for i, pk in enumerate(table.ids):
        emb = None
        if table.embedding is not None and not np.isnan(table.embedding[i]).all():
            emb = format_vector(table.embedding[i])
        mv = None
        if table.mv_offsets is not None and table.mv_offsets[i + 1] > table.mv_offsets[i]:
            mv = format_vector_array(table.mv_tokens[table.mv_offsets[i]: table.mv_offsets[i + 1]])
        out.append("\t".join(_copy_escape(x) for x in (pk, table.contents[i], emb, mv)))
    return out


def results_to_copy_text(pipeline_id: int | str, query_ids: list, results: list, unit: str = "chunk") -> list[str]:
    """A page of ranked lists (None = failed query) -> `COPY chunk_retrieved_result | image_chunk_retrieved_result
    (query_id, pipeline_id, <unit>_id, rel_score) FROM STDIN` rows; same rows as `_collect_retrieval_results` builds."""
    del unit  # (the column order is the same for both tables; the caller names the table)
    rows = []
    for qid, res in zip(query_ids, results, strict=True):
        for r in res or []:
            rows.append("\t".join(_copy_escape(x) for x in (qid, pipeline_id, r["doc_id"], repr(float(r["score"])))))
    return rows


def _main(argv: list[str] | None = None) -> int:
    """`pg_restore -f - --data-only -t chunk dataset.dump | python -m autorag_research_amd.pgtext --table chunk --out DIR`"""
    import argparse  # noqa: PLC0415
    import sys  # noqa: PLC0415

    ap = argparse.ArgumentParser(description="pg_restore / pg_dump plain script (stdin) -> shard directory of one table")
    ap.add_argument("--table", default="chunk", help="chunk | image_chunk")
    ap.add_argument("--out", required=True, help="shard directory to write (shards.py format)")
    ap.add_argument("--id-type", choices=["int", "str"], default="int", help="primary-key type of the schema (bigint | string)")
    a = ap.parse_args(argv)
    has_contents = a.table != "image_chunk"   # (image_chunk.contents is the image itself: not exported to the index side)
    cols = ("id", "contents", "embedding", "embeddings") if has_contents else ("id", "embedding", "embeddings")
    shard = copy_text_to_shard(restore_script_rows(sys.stdin, a.table, cols), a.out, a.id_type, has_contents)
    print(shard)
    return 0


__all__ = ["format_vector", "format_vector_array", "format_vector_array_sql", "parse_vector", "parse_vector_array",
           "copy_text_to_table", "copy_text_to_shard", "restore_script_rows", "table_to_copy_text", "results_to_copy_text"]

if __name__ == "__main__":
    raise SystemExit(_main())
