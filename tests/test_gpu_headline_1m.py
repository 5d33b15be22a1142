"""GPU: BASELINE's second synthetic size -- N = 1 M rows, d = 768 (synth.gaussian_chunk, the bench's generator), one 1024-query
block -- compared with the CPU oracle IN FULL: every id and every float8 distance bit of all 1024 lists, at k = 10 (the one-wave
prune, five chunks behind the 16 k-row starter) and at k = 100 (BASELINE config 2's limit: the two-wave prune behind the
64 k-row starter), through the C ABI's device entry point; then the same block through the exact-scan path.  The largest
full-list comparison in the suite (round 5's was 300 k rows x 16 queries)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_million_rows_every_list_equals_the_oracle(native_built, oracle):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import autorag_research_amd as pkg
    from autorag_research_amd import synth

    n, d, B = 1_000_000, 768, 1024
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    Q = torch.randn((B, d), generator=g, device=dev, dtype=torch.float32)
    Q /= Q.norm(dim=1, keepdim=True)
    C = np.empty((n, d), dtype=np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.reserve(n)
        for c in range(n // synth.CHUNK_ROWS):
            x = synth.gaussian_chunk(torch, c, synth.CHUNK_ROWS, d, dev)
            torch.cuda.synchronize()
            idx.add_device(x.data_ptr(), x.shape[0])
            C[c * synth.CHUNK_ROWS:(c + 1) * synth.CHUNK_ROWS] = x.cpu().numpy()
            del x
        Qh = Q.cpu().numpy()
        s = torch.cuda.current_stream().cuda_stream
        for k in (10, 100):
            rd, rr = oracle.topk_search(C, Qh, k)
            od = torch.empty((B, k), dtype=torch.float64, device=dev)
            orr = torch.empty((B, k), dtype=torch.int64, device=dev)
            idx.reset_stats()
            idx.search_device(Q.data_ptr(), B, k, od.data_ptr(), orr.data_ptr(), s)
            torch.cuda.synchronize()
            assert idx.stat("fallback_queries") == 0 and idx.stat("retry_queries") == 0 and idx.stat("starters") == 1
            assert idx.stat("screen_dtype_active") == 2                       # the int8 screen at both limits
            assert idx.stat("chunks") <= 5, idx.stat("chunks")                # (round 5 walked 1 M rows at k = 100 in ~20 chunks)
            assert np.array_equal(orr.cpu().numpy(), rr), f"ids differ from the oracle at k={k}"
            assert np.array_equal(od.cpu().numpy().view(np.uint64), rd.view(np.uint64)), f"distances differ from the oracle at k={k}"
            if k == 100:   # the guaranteed exact path on a slice of the block: the same lists
                idx.set_option("path", "scan")
                idx.search_device(Q.data_ptr(), 96, k, od.data_ptr(), orr.data_ptr(), s)
                torch.cuda.synchronize()
                idx.set_option("path", "auto")
                assert np.array_equal(orr[:96].cpu().numpy(), rr[:96])
                assert np.array_equal(od[:96].cpu().numpy().view(np.uint64), rd[:96].view(np.uint64))
