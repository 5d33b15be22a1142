"""GPU parity: the MaxSim screen over the GRANULE-PACKED bf16 copy (csrc/k_maxsim_wg8.h, option "maxsim_pack8") -- the same lists as
the padded copy's screen and as the CPU oracle, bit for bit, over document shapes that put boundaries everywhere in a block."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg(native_built):
    import autorag_research_amd as p

    return p


def _store(rng, lens, d=128):
    lens = np.asarray(lens, np.int64)
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    return tok, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def _queries(rng, lens, d=128):
    qs = [rng.standard_normal((t, d)).astype(np.float32) for t in lens]
    qs = [q / np.linalg.norm(q, axis=1, keepdims=True) for q in qs]
    return np.concatenate(qs, axis=0), np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def _same(a, b):
    (da, ra), (db, rb) = a, b
    assert np.array_equal(ra, rb)
    assert np.array_equal(np.isnan(da), np.isnan(db))
    ok = ~np.isnan(da)
    assert np.array_equal(da[ok].view(np.uint32), db[ok].view(np.uint32))


SHAPES = {
    # every residue of a document's length modulo 8 and 32, single tokens, empty documents, in an order that moves the boundaries
    # through every granule of a block
    "residues": lambda rng: [int(x) for x in rng.permutation(np.r_[np.arange(0, 70), np.arange(0, 70), [1] * 40, [0] * 25, [8, 16, 24, 32] * 10])],
    # thousands of short passages: every workgroup of the 256 owns a range, the first and last blocks of the ranges are shared
    "passages": lambda rng: [int(x) for x in rng.integers(32, 181, size=4000)],
    # tiny documents: four and more per block, empty ones in between, a long one now and then
    "tiny": lambda rng: [int(x) for x in np.where(rng.random(9000) < 0.02, 300, rng.integers(0, 12, size=9000))],
    # ONE document, a few documents (fewer than workgroups), a store that ends in empty documents
    "few": lambda rng: [57, 0, 0, 1, 200, 9, 0, 0],
}


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("nq", [16, 9, 8])
def test_pack8_equals_padded_and_oracle(pkg, oracle, shape, nq):
    """maxsim_pack8 = 1 against 0 and the oracle: doc ids and fp32 distances bit-exact; the packed launches are counted (so the
    comparison is known to be between the two forms), the candidate sets are the same size (the screen distances are bit-identical:
    same bf16 values, order-free maxima, the same per-document sums)."""
    rng = np.random.default_rng(len(shape) * 100 + nq)
    lens = SHAPES[shape](rng)
    tok, off = _store(rng, lens)
    qlens = [32] * nq
    if nq == 9:
        qlens[-1] = 20   # (the last query of an aligned pass may be shorter)
    qtok, qoff = _queries(rng, qlens)
    k = 10
    want = oracle.maxsim_topk(tok, off, qtok, qoff, k)
    with pkg.Mi355Index(128) as idx:
        idx.add_multivec(tok, off)
        if nq == 8:
            idx.set_option("maxsim_wg", 1)   # (8 column blocks take the workgroup form by document length only: force it)
        got, cands = {}, {}
        for pack in (0, 1):
            idx.set_option("maxsim_pack8", pack)
            idx.reset_stats()
            got[pack] = idx.search_maxsim(qtok, qoff, k)
            cands[pack] = idx.stat("maxsim_candidates")
            assert (idx.stat("maxsim_packed_launches") > 0) == (pack == 1), (pack, idx.stat("maxsim_packed_launches"))
            _same(got[pack], want)
        assert cands[0] == cands[1]
        assert idx.stat("maxsim_packed_blocks") > 0


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("qlens", [[32], [20, 7, 32], [24, 24], [1], [64], [31, 0, 5], [9, 20, 0, 30], [32] * 3, [24] * 5, [100, 28], [32, 31, 33, 32]])
def test_pack8_small_passes_one_wave_per_document(pkg, oracle, shape, qlens):
    """A pass of up to four column blocks -- one to four queries per call: the reference's call shape and small batches -- takes one
    wave per document (k_maxsim16_d128) over the packed copy too: a document's first and last block are shared with its neighbours, whose lanes are not
    loaded and whose accumulator quads are not looked at.  Any query layout (queries packed column after column, a query may straddle
    blocks), every document shape; pack 1 == pack 0 == oracle."""
    rng = np.random.default_rng(len(shape) * 31 + sum(qlens))
    tok, off = _store(rng, SHAPES[shape](rng))
    qtok, qoff = _queries(rng, [t for t in qlens])
    k = 7
    want = oracle.maxsim_topk(tok, off, qtok, qoff, k)
    live = np.asarray(qlens) > 0
    with pkg.Mi355Index(128) as idx:
        idx.add_multivec(tok, off)
        cands = {}
        for pack in (0, 1):
            idx.set_option("maxsim_pack8", pack)
            idx.reset_stats()
            d, r = idx.search_maxsim(qtok, qoff, k)
            cands[pack] = idx.stat("maxsim_candidates")
            assert (idx.stat("maxsim_packed_launches") > 0) == (pack == 1)
            _same((d[live], r[live]), (want[0][live], want[1][live]))
        assert cands[0] == cands[1]


def test_pack8_follows_the_store_and_the_pass_shape(pkg, oracle):
    """The packed copy is a shadow of the store: stale after an add (extended by the next pass that takes it), not taken by passes
    of the workgroup form that are not aligned (24-vector queries), taken by passes of up to 8 column blocks whatever their queries,
    and in the default mode (-1) not built for stores it would not shorten (long documents: pages)."""
    rng = np.random.default_rng(8)
    k = 5
    qa, oa = _queries(rng, [32] * 12)
    qn, on = _queries(rng, [24] * 12)
    q4, o4 = _queries(rng, [32] * 4)
    tok, off = _store(rng, rng.integers(1, 90, size=900))
    tok2, off2 = _store(rng, rng.integers(0, 50, size=500))
    with pkg.Mi355Index(128) as idx:
        idx.add_multivec(tok, off)
        idx.reset_stats()
        _same(idx.search_maxsim(qa, oa, k), oracle.maxsim_topk(tok, off, qa, oa, k))
        assert idx.stat("maxsim_packed_launches") == 1     # default mode: 1..89-token documents lose a fifth of their blocks
        b1 = idx.stat("maxsim_packed_blocks")
        _same(idx.search_maxsim(qn, on, k), oracle.maxsim_topk(tok, off, qn, on, k))
        assert idx.stat("maxsim_packed_launches") == 1     # 9 column blocks of 24-vector queries: the padded copy
        _same(idx.search_maxsim(q4, o4, k), oracle.maxsim_topk(tok, off, q4, o4, k))
        assert idx.stat("maxsim_packed_launches") == 2     # 4 column blocks: one wave per document, over the packed copy
        q5, o5 = _queries(rng, [32] * 5)
        _same(idx.search_maxsim(q5, o5, k), oracle.maxsim_topk(tok, off, q5, o5, k))
        assert idx.stat("maxsim_packed_launches") == 2     # 5 column blocks: one wave per document, over the padded copy
        idx.add_multivec(tok2, off2)                        # the store grows: the copy is rebuilt for the next aligned pass
        tok_all = np.concatenate([tok, tok2])
        off_all = np.concatenate([off, off[-1] + off2[1:]])
        _same(idx.search_maxsim(qa, oa, k), oracle.maxsim_topk(tok_all, off_all, qa, oa, k))
        assert idx.stat("maxsim_packed_launches") == 3 and idx.stat("maxsim_packed_blocks") > b1
        # documents without vectors add no granule (nothing to pack)
        empty = np.zeros((0, 128), np.float32)
        idx.add_multivec(empty, np.zeros(4, np.int64))
        off_all = np.concatenate([off_all, np.full(3, off_all[-1])])
        built0, blocks0 = idx.stat("maxsim_packed_built"), idx.stat("maxsim_packed_blocks")
        _same(idx.search_maxsim(qa, oa, k), oracle.maxsim_topk(tok_all, off_all, qa, oa, k))
        assert idx.stat("maxsim_packed_built") - built0 <= 1 and idx.stat("maxsim_packed_blocks") == blocks0
        # a store that grows is packed from the block its new granules start in (an ingest loop that searches between its adds):
        # five small adds write their own blocks and the block each shares with what was there, not the copy five times over
        for i in range(5):
            t3, o3 = _store(rng, rng.integers(0, 40, size=37))
            idx.add_multivec(t3, o3)
            tok_all = np.concatenate([tok_all, t3])
            off_all = np.concatenate([off_all, off_all[-1] + o3[1:]])
            _same(idx.search_maxsim(qa, oa, k), oracle.maxsim_topk(tok_all, off_all, qa, oa, k))
        grown = idx.stat("maxsim_packed_blocks") - blocks0
        assert 0 < grown and idx.stat("maxsim_packed_built") - built0 <= grown + 7
    tokp, offp = _store(rng, [515] * 60)                   # pages: 16.1 blocks padded to 17 -- the packed copy would save 4 %
    with pkg.Mi355Index(128) as idx:
        idx.add_multivec(tokp, offp)
        idx.reset_stats()
        _same(idx.search_maxsim(qa, oa, k), oracle.maxsim_topk(tokp, offp, qa, oa, k))
        assert idx.stat("maxsim_packed_launches") == 0 and idx.stat("maxsim_packed_blocks") == 0
        idx.set_option("maxsim_pack8", 1)                   # ... unless it is asked for
        _same(idx.search_maxsim(qa, oa, k), oracle.maxsim_topk(tokp, offp, qa, oa, k))
        assert idx.stat("maxsim_packed_launches") == 1
        with pytest.raises(pkg.NativeError):
            idx.set_option("maxsim_pack8", 2)
        with pytest.raises(pkg.NativeError):
            idx.set_option("maxsim_pack8", -2)
