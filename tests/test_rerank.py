"""ColBERT reranker MaxSim: the reference's `_maxsim_score` (rerankers/colbert.py:63-84) frozen on seeded
padded token tensors (tests/golden/rerank_golden.npz) vs the oracle restatement (CPU) and vs the GPU kernel
(`mi355dr_maxsim_subset_ex` with MI355DR_MAXSIM_CLAMP0 over a store filled by device pointer)."""

import asyncio

import numpy as np
import pytest

from helpers import GOLDEN, OracleIndex

G = np.load(GOLDEN / "rerank_golden.npz")
CASES = [0, 1, 2]


@pytest.mark.parametrize("c", CASES)
def test_oracle_restatement_matches_the_reference(oracle, c):
    got = oracle.colbert_rerank_scores(G[f"q{c}"], G[f"qm{c}"], G[f"d{c}"], G[f"dm{c}"])
    assert np.allclose(got, G[f"score{c}"], rtol=0, atol=1e-6)
    assert (G["score0"][3] == 0.0) and (G["dm0"][3].sum() == 0)  # the document without a valid token scores 0


@pytest.mark.parametrize("c", CASES)
def test_host_flow_with_the_oracle_index(oracle, c):
    from autorag_research_amd.rerank import colbert_maxsim_scores

    got = colbert_maxsim_scores(G[f"q{c}"], G[f"qm{c}"], G[f"d{c}"], G[f"dm{c}"], index_factory=OracleIndex)
    assert np.allclose(got, G[f"score{c}"], rtol=0, atol=1e-6)


def test_reranker_surface(oracle):
    from autorag_research_amd.rerank import Mi355ColBERTReranker, RandomTokenEncoder, RerankResult

    enc = RandomTokenEncoder(dim=32, device="cpu")
    r = Mi355ColBERTReranker(enc, index_factory=OracleIndex)
    docs = ["alpha beta gamma", "delta", "alpha beta", "", "gamma gamma alpha"]
    res = r.rerank("alpha beta", docs, top_k=3)
    assert [type(x) for x in res] == [RerankResult] * 3 and res[0].score >= res[1].score >= res[2].score
    assert {res[0].index, res[1].index} == {0, 2} and res[0].text == docs[res[0].index]  # both contain both query words: 1.0
    assert abs(res[0].score - 1.0) < 1e-6 and abs(res[1].score - 1.0) < 1e-6 and res[0].index < res[1].index  # stable ties
    full = r.rerank("alpha beta", docs)
    assert len(full) == 5 and full[-1].score == 0.0 or full[-1].score <= full[-2].score
    assert r.rerank("x", []) == []
    assert [x.index for x in asyncio.run(r.arerank("alpha beta", docs, 3))] == [x.index for x in res]
    assert len(r.rerank_documents(["alpha", "delta"], [docs, docs[:2]], top_k=1)) == 2
    # scores agree with the oracle restatement on what the encoder produced
    qe, qm = enc.encode(["alpha beta"])
    de, dm = enc.encode(docs)
    want = oracle.colbert_rerank_scores(qe.numpy(), qm.numpy(), de.numpy(), dm.numpy())
    assert np.allclose([x.score for x in sorted(full, key=lambda x: x.index)], want, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES)
def test_gpu_scores_match_the_reference(native_built, oracle, c):
    import torch

    from autorag_research_amd.rerank import colbert_maxsim_scores

    q, qm, d, dm = (G[f"{n}{c}"] for n in ("q", "qm", "d", "dm"))
    host = colbert_maxsim_scores(q, qm, d, dm)                                       # host arrays -> add_multivec
    dev = colbert_maxsim_scores(torch.from_numpy(q).cuda(), torch.from_numpy(qm).cuda(), torch.from_numpy(d).cuda(),
                                torch.from_numpy(dm).cuda())                          # device pointer -> add_multivec_device
    assert np.array_equal(host, dev)                                                  # both store builders agree bit for bit
    assert np.allclose(dev, G[f"score{c}"], rtol=0, atol=1e-6)
    assert np.allclose(dev, oracle.colbert_rerank_scores(q, qm, d, dm), rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_gpu_reranker_at_the_reference_models_shape(native_built, oracle):
    """colbert-ir/colbertv2.0 scored the reference's way (rerankers/colbert.py:41-84: AutoModel.last_hidden_state, hidden
    size 768, max_length 512): 768-dimensional token vectors and a query of more than 128 valid tokens -- both outside what
    one launch of the MaxSim kernel stages, both served in tiles since round 3 -- against the `_maxsim_score` restatement."""
    from autorag_research_amd.rerank import colbert_maxsim_scores

    rng = np.random.default_rng(12)
    d, lq, n_docs, ld = 768, 200, 9, 90
    q = rng.standard_normal((lq, d)).astype(np.float32) / np.sqrt(d)
    qm = np.ones((lq,), np.int64)
    qm[170:] = 0                                  # padded tail
    docs = rng.standard_normal((n_docs, ld, d)).astype(np.float32) / np.sqrt(d)
    dm = (np.arange(ld)[None, :] < rng.integers(0, ld + 1, size=(n_docs, 1))).astype(np.int64)
    dm[3] = 0                                     # a document without a valid token scores 0
    got = colbert_maxsim_scores(q, qm, docs, dm)
    want = oracle.colbert_rerank_scores(q, qm, docs, dm)
    assert np.allclose(got, want, rtol=0, atol=2e-6) and got[3] == 0.0


@pytest.mark.gpu
def test_device_built_store_equals_host_built_store(native_built):
    """mi355dr_add_multivec_device (kernel-built padded store, bf16 fragments, bound quantities) vs mi355dr_add_multivec:
    the screened MaxSim search and the exact subset scoring return identical bits, appended in two batches."""
    import torch

    import autorag_research_amd as pkg

    rng = np.random.default_rng(5)
    d = 128
    lens = rng.integers(0, 90, size=300)
    lens[7] = 1030  # a ColPali-sized page
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    qt = rng.standard_normal((3 * 24, d)).astype(np.float32)
    qoff = np.array([0, 24, 48, 72], dtype=np.int32)
    h, g = pkg.Mi355Index(d), pkg.Mi355Index(d)
    cut = 120
    h.add_multivec(tok[: off[cut]], off[: cut + 1])
    h.add_multivec(tok[off[cut]:], off[cut:] - off[cut])
    t = torch.from_numpy(tok).cuda()
    torch.cuda.synchronize()
    g.add_multivec_device(t.data_ptr(), off[: cut + 1])
    g.add_multivec_device(t[off[cut]:].data_ptr(), off[cut:] - off[cut])
    assert h.n_docs() == g.n_docs() == 300
    for screen in (1, 0):
        h.set_option("maxsim_screen", screen)
        g.set_option("maxsim_screen", screen)
        hd, hr = h.search_maxsim(qt, qoff, 10)
        gd, gr = g.search_maxsim(qt, qoff, 10)
        assert np.array_equal(hr, gr) and np.array_equal(hd.view(np.uint32), gd.view(np.uint32))
    ids = np.tile(np.arange(300, dtype=np.int64), (3, 1))
    a, b = h.maxsim_subset(qt, qoff, ids), g.maxsim_subset(qt, qoff, ids)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    h.close()
    g.close()
