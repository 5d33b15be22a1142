"""The PostgreSQL text layer of the DB <-> index bridge against strings the REFERENCE's own converters
produced / parsed (tests/golden/pgtext_golden.json: VectorArray.process_bind_param / process_result_value, _vec_to_pg_literal,
_vecs_to_pg_array), then a COPY-text dump -> shard directory -> (oracle-backed) search round trip and the result write-back rows."""

import json

import numpy as np
import pytest

from helpers import GOLDEN, OracleIndex

G = json.loads((GOLDEN / "pgtext_golden.json").read_text())


def test_text_forms_match_the_reference_converters():
    from autorag_research_amd import pgtext as pt

    assert pt.format_vector_array(None) is G["null_bind"] is None and pt.parse_vector_array(None) is G["null_result"] is None
    for c in G["cases"]:
        vals = c["values"]
        assert pt.format_vector_array(vals) == c["bind"]
        assert [pt.format_vector(v) for v in vals] == c["literals"]
        if vals:
            assert pt.format_vector_array_sql(vals) == c["sql_array"]
        parsed = pt.parse_vector_array(c["bind"])
        assert parsed.shape[0] == len(c["parsed"])
        if vals:
            assert np.array_equal(parsed, np.asarray(c["parsed"], dtype=np.float64).astype(np.float32))
            assert np.array_equal(np.stack([pt.parse_vector(l) for l in c["literals"]]), parsed)
    for h in G["hand_strings"]:
        got = pt.parse_vector_array(h["text"])
        assert got.shape[0] == len(h["parsed"])
        if h["parsed"]:
            assert np.allclose(got, np.asarray(h["parsed"]), rtol=1e-7, atol=0)
    assert np.array_equal(pt.parse_vector_array([[1, 2], np.array([3.5, 4.5])]), np.asarray(G["preparsed"], np.float32))
    assert pt.parse_vector(r"\N") is None and pt.parse_vector_array(r"\N") is None
    with pytest.raises(ValueError, match="not a vector literal"):
        pt.parse_vector("1,2,3")


def test_copy_dump_to_shard_to_search_and_write_back(tmp_path, monkeypatch, oracle):
    import autorag_research_amd.service as svc
    from autorag_research_amd import pgtext as pt
    from autorag_research_amd.shards import read_shard as load_table
    from autorag_research_amd.store import ChunkTable

    rng = np.random.default_rng(4)
    n, d = 40, 16
    emb = rng.standard_normal((n, d)).astype(np.float32)
    emb[7] = np.nan                                     # NULL embedding
    toks = [rng.standard_normal((int(t), d)).astype(np.float32) for t in rng.integers(0, 5, size=n)]
    lens = [t.shape[0] for t in toks]
    table = ChunkTable(ids=list(range(100, 100 + n)), contents=[f"text\twith tab {i}\nand newline" if i % 9 == 0 else f"t{i}"
                                                                for i in range(n)],
                       embedding=emb, mv_tokens=np.concatenate(toks), mv_offsets=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
    table.contents[3] = None
    dump = pt.table_to_copy_text(table)                 # what `COPY (SELECT id, contents, embedding, embeddings ...) TO STDOUT` prints
    assert dump[7].split("\t")[2] == r"\N" and all(len(line.split("\t")) == 4 for line in dump)
    shard = pt.copy_text_to_shard(iter(dump + [r"\."]), tmp_path / "chunk", id_type="int")
    back = load_table(shard)
    assert back.ids == table.ids and back.contents == table.contents
    assert np.array_equal(np.isnan(back.embedding), np.isnan(emb)) and np.array_equal(back.embedding[~np.isnan(emb)], emb[~np.isnan(emb)])
    assert np.array_equal(back.mv_offsets, table.mv_offsets) and np.array_equal(back.mv_tokens, table.mv_tokens)
    # search over the re-loaded table, then the rows of the write-back COPY
    monkeypatch.setattr(svc, "Mi355Index", OracleIndex)
    from autorag_research_amd.store import InMemoryStore

    store = InMemoryStore()
    store.chunks = back
    q = rng.standard_normal((2, d)).astype(np.float32)
    store.add_queries(["qa", "qb"], embedding=list(q))
    s = svc.Mi355RetrievalService(lambda: store)
    res = s.vector_search(["qa", "qb"], 3)
    assert all(r["doc_id"] != 107 for lst in res for r in lst)        # the NULL-embedding row is never returned
    rows = pt.results_to_copy_text(5, ["qa", "qb", "qc"], [res[0], res[1], None])
    assert len(rows) == 6
    f = rows[0].split("\t")
    assert f[0] == "qa" and f[1] == "5" and int(f[2]) == res[0][0]["doc_id"] and float(f[3]) == res[0][0]["score"]


def test_pg_restore_script_to_shard(tmp_path):
    """A published dataset is a pg_dump custom-format archive (orm/connection.py:298); `pg_restore -f -` prints it as a script
    whose `COPY public.chunk (<every column>) FROM stdin;` block carries the rows.  The block is found among the other tables',
    projected onto (id, contents, embedding, embeddings) whatever the column order, and lands in a shard directory -- by the
    function and by the `python -m autorag_research_amd.pgtext` one-liner of INTEGRATION.md."""
    import subprocess
    import sys

    from autorag_research_amd import pgtext as pt
    from autorag_research_amd.shards import read_shard

    rng = np.random.default_rng(9)
    n, d = 12, 8
    emb = rng.standard_normal((n, d)).astype(np.float32)
    toks = [rng.standard_normal((int(t), d)).astype(np.float32) for t in rng.integers(0, 4, size=n)]
    rows = []
    for i in range(n):
        e = r"\N" if i == 4 else pt.format_vector(emb[i])
        mv = pt.format_vector_array(toks[i]) if toks[i].shape[0] else (r"\N" if i % 2 else "{}")
        # the reference's column order (orm/schema_factory.py:148-154) with the two vector columns swapped and a text with a tab
        rows.append("\t".join([str(100 + i), f"passage\\t{i}" if i == 2 else f"passage {i}", mv, e, r"\N", "f", r"\N"]))
    script = ["--", "-- PostgreSQL database dump", "--", "SET statement_timeout = 0;", "",
              "COPY public.query (id, contents, embedding) FROM stdin;", "1\tq\t[1,2]", r"\.", "",
              "COPY public.chunk (id, contents, embeddings, embedding, bm25_tokens, is_table, table_type) FROM stdin;", *rows, r"\.", "",
              "COPY public.image_chunk (id, parent_page, contents, mimetype, embedding, embeddings) FROM stdin;", r"\.", ""]
    t = pt.copy_text_to_table(pt.restore_script_rows(script, "chunk"))
    assert t.ids == list(range(100, 100 + n)) and t.contents[2] == "passage\t2"
    live = [i for i in range(n) if i != 4]
    assert np.isnan(t.embedding[4]).all() and np.array_equal(t.embedding[live], emb[live])
    assert np.array_equal(np.diff(t.mv_offsets), [x.shape[0] for x in toks]) and np.array_equal(t.mv_tokens, np.concatenate(toks))
    with pytest.raises(ValueError, match="no `COPY"):
        list(pt.restore_script_rows(script, "caption"))
    with pytest.raises(ValueError, match="no column"):
        list(pt.restore_script_rows(script, "query", ("id", "embeddings")))
    with pytest.raises(ValueError, match="does not end"):
        list(pt.restore_script_rows(script[:12], "chunk"))
    r = subprocess.run([sys.executable, "-m", "autorag_research_amd.pgtext", "--table", "chunk", "--out", str(tmp_path / "chunk")],
                       input="\n".join(script) + "\n", capture_output=True, text=True, cwd=str(GOLDEN.parent.parent), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    back = read_shard(tmp_path / "chunk")
    assert back.ids == t.ids and np.array_equal(back.mv_tokens, t.mv_tokens) and np.array_equal(back.embedding[live], emb[live])
