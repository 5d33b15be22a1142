"""GPU: the multi-vector configs (BASELINE C4 / C5) at SURVEY 8(d)'s FULL sizes inside the test suite -- the twin of
test_gpu_headline_10m.py for `@#`: 1 M ColBERT-like docs (U{32..180} token vectors each, ~106 M vectors) with 32-vector queries
and 100 k ColPali-like pages (1030 patch vectors each, 103 M vectors) with 24-vector queries, d = 128, unit-norm vectors, built
on the device chunk by chunk (bench_support.run_maxsim's generator) and handed to the index by pointer.

The oracle cannot finish at this size; the checks are the size-independent ones:
  * planted documents (the query's own vectors + noise at the head, known vector for vector, at recorded ids) come back FIRST,
    in the oracle's order, with the oracle's fp32 distances bit for bit (oracle on the planted documents only);
  * one 16-query pass == sixteen 1-query calls == the exact kernel over every document (`maxsim_screen = 0`) on a query subset,
    ids and fp32 bits;
  * the store cut into two halves by cumulative token count (row offsets), searched apart and merged by (distance, doc) == the
    whole;
  * lists sorted by (distance, doc), docs unique and in range; no query fell back to the exact full scan."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D, K, QB = 128, 10, 16


def _build(torch, pkg, dev, lens, planted, seed):
    """whole + lo / hi (cut where the cumulative token count crosses half) built from the same device chunks; `planted` =
    {doc id: [T, D] fp32 device tensor} overwrites those documents' leading vectors.  Returns (whole, lo, hi, n_lo)."""
    n_docs = len(lens)
    cum = np.concatenate([[0], np.cumsum(lens)])
    n_lo = int(np.searchsorted(cum, cum[-1] // 2))
    whole, lo, hi = pkg.Mi355Index(D), pkg.Mi355Index(D), pkg.Mi355Index(D)
    hi.set_option("row_offset", n_lo)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    docs_per_chunk = max(1, (1 << 22) // int(lens.max()))
    bounds = sorted(set(list(range(0, n_docs, docs_per_chunk)) + [n_lo, n_docs]))   # chunks never straddle the cut
    for d0, d1 in zip(bounds[:-1], bounds[1:]):
        ln = lens[d0:d1]
        off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        x = torch.randn((int(off[-1]), D), generator=g, device=dev, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        for doc, vec in planted.items():
            if d0 <= doc < d1:
                x[off[doc - d0]: off[doc - d0] + vec.shape[0]] = vec
        torch.cuda.synchronize()
        whole.add_multivec_device(x.data_ptr(), off)
        (lo if d1 <= n_lo else hi).add_multivec_device(x.data_ptr(), off)
        del x
    torch.cuda.synchronize()
    assert whole.n_docs() == n_docs and lo.n_docs() == n_lo and hi.n_docs() == n_docs - n_lo
    return whole, lo, hi, n_lo


def _check_store(torch, pkg, oracle, tokens: str):
    dev = torch.device("cuda", 0)
    if torch.cuda.mem_get_info(dev)[0] < 200 * 2**30:
        pytest.skip("needs ~170 GB of free HBM (the whole store + its two halves, fp32 + bf16 copies)")
    rng = np.random.default_rng(777)
    n_docs, nq = (1_000_000, 32) if tokens == "text" else (100_000, 24)
    lens = rng.integers(32, 181, size=n_docs) if tokens == "text" else np.full((n_docs,), 1030, dtype=np.int64)
    qtok = rng.standard_normal((QB * nq, D)).astype(np.float32)
    qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
    qoff = (np.arange(QB + 1) * nq).astype(np.int32)
    # two planted documents per query, at recorded ids, known vector for vector: the query's vectors + noise of two strengths at
    # the head, unit-norm random vectors behind them
    planted, planted_host, plant_of = {}, {}, {}
    pids = rng.choice(n_docs, size=2 * QB, replace=False)
    for b in range(QB):
        for j, sigma in enumerate((0.3, 0.8)):
            doc = int(pids[2 * b + j])
            v = rng.standard_normal((int(lens[doc]), D)).astype(np.float32)
            q = qtok[b * nq:(b + 1) * nq]
            v[:nq] = q + sigma / np.sqrt(D) * rng.standard_normal(q.shape).astype(np.float32)
            v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
            planted_host[doc] = v
            planted[doc] = torch.from_numpy(v).to(dev)
            plant_of.setdefault(b, []).append(doc)
    whole, lo, hi, n_lo = _build(torch, pkg, dev, lens, planted, 777)
    try:
        whole.reset_stats()
        dist, rows = whole.search_maxsim(qtok, qoff, K)
        assert whole.stat("maxsim_fallbacks") == 0 and whole.stat("maxsim_screened") == QB
        # lists: sorted by (distance, doc), docs unique and in range
        assert rows.min() >= 0 and rows.max() < n_docs and (np.diff(dist, axis=1) >= 0).all()
        tie = np.diff(dist, axis=1) == 0
        assert (np.diff(rows, axis=1)[tie] > 0).all() and all(len(set(r)) == K for r in rows.tolist())
        # planted documents lead their query's list, in the oracle's order, with the oracle's fp32 distances (oracle on the
        # planted documents alone: a document's distance does not depend on the rest of the store)
        for b in range(QB):
            docs = plant_of[b]
            toks = [planted_host[doc] for doc in docs]
            off = np.concatenate([[0], np.cumsum([t.shape[0] for t in toks])]).astype(np.int64)
            od, orow = oracle.maxsim_topk(np.concatenate(toks), off, qtok[b * nq:(b + 1) * nq], np.array([0, nq], np.int32), 2)
            assert rows[b, :2].tolist() == [docs[int(i)] for i in orow[0]], (b, rows[b], docs)
            assert np.array_equal(dist[b, :2].view(np.uint32), od[0].view(np.uint32))
            assert dist[b, 1] < dist[b, 2] - 1.0   # far ahead of every random document
        # exact distances of the planted documents: the subset entry point on the whole store == the search's own values
        sub = whole.maxsim_subset(qtok, qoff, np.asarray([plant_of[b] for b in range(QB)], dtype=np.int64))
        for b in range(QB):
            got = {int(r): float(x) for r, x in zip(rows[b, :2], dist[b, :2])}
            for j, doc in enumerate(plant_of[b]):
                assert np.float32(got[doc]).view(np.uint32) == np.float32(sub[b, j]).view(np.uint32)
        # one 16-query pass == sixteen 1-query calls (one wave per document, HBM-bound form), bit for bit
        for b in range(QB):
            d1, r1 = whole.search_maxsim(qtok[b * nq:(b + 1) * nq], np.array([0, nq], np.int32), K)
            assert np.array_equal(r1[0], rows[b]) and np.array_equal(d1[0].view(np.uint32), dist[b].view(np.uint32)), b
        # ... == the exact fp32 kernel over EVERY document on a query subset
        whole.set_option("maxsim_screen", 0)
        de, re_ = whole.search_maxsim(qtok[: 3 * nq], qoff[:4], K)
        whole.set_option("maxsim_screen", 1)
        assert np.array_equal(re_, rows[:3]) and np.array_equal(de.view(np.uint32), dist[:3].view(np.uint32))
        # halves (token-count split, row offsets) merged by (distance, doc) == the whole
        da, ra = lo.search_maxsim(qtok, qoff, K)
        db, rb = hi.search_maxsim(qtok, qoff, K)
        assert ra.max() < n_lo <= rb.min()
        dd, rr = np.concatenate([da, db], 1), np.concatenate([ra, rb], 1)
        order = np.lexsort((rr, dd), axis=1)[:, :K]
        assert np.array_equal(np.take_along_axis(rr, order, 1), rows)
        assert np.array_equal(np.take_along_axis(dd, order, 1).view(np.uint32), dist.view(np.uint32))
        assert lo.stat("maxsim_fallbacks") == 0 and hi.stat("maxsim_fallbacks") == 0
        # idempotent
        d2, r2 = whole.search_maxsim(qtok, qoff, K)
        assert np.array_equal(r2, rows) and np.array_equal(d2.view(np.uint32), dist.view(np.uint32))
    finally:
        for i in (whole, lo, hi):
            i.close()


def test_one_million_colbert_like_docs(native_built, oracle):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import autorag_research_amd as pkg

    _check_store(torch, pkg, oracle, "text")


def test_hundred_thousand_colpali_like_pages(native_built, oracle):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import autorag_research_amd as pkg

    _check_store(torch, pkg, oracle, "page")
