"""CPU, world_size 2 over gloo: the row-sharded search + all-gather + merge equals the unsharded oracle."""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _maxsim_case():
    rng = np.random.default_rng(321)
    lens = rng.integers(0, 30, size=400)  # ragged, some docs without vectors
    lens[:40] = rng.integers(60, 90, size=40)  # token-heavy head: doc counts and token counts split differently
    tok = rng.standard_normal((int(lens.sum()), 24)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    qtok = rng.standard_normal((5 + 11, 24)).astype(np.float32)
    return tok, off, qtok, np.array([0, 5, 16], dtype=np.int32)


def _worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from helpers import OracleIndex
    from autorag_research_amd.sharded import ShardedSearcher, shard_bounds

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    n, d, B, k = 3001, 48, 9, 12
    C = rng.standard_normal((n, d)).astype(np.float32)
    C[5] = C[2900]        # a cross-shard exact tie: the lower global row must win on every rank
    C[17] = 0.0           # NaN distance on shard 0
    Q = rng.standard_normal((B, d)).astype(np.float32)
    lo, hi = shard_bounds(n, world, rank, granule=250)
    s = ShardedSearcher(d, "cosine", index_factory=OracleIndex)
    s.add_local(C[lo:hi], lo)
    dist_g, rows_g = s.search(Q, k)
    # the same through the overlapped pipeline: 3 blocks of <= 4 queries, the gather of block i under the search of i+1
    dist_p, rows_p = s.search(Q, k, block=4)
    assert s.overlapped_blocks == 2
    assert np.array_equal(rows_p, rows_g) and np.array_equal(dist_p.view(np.uint64), dist_g.view(np.uint64))
    # k larger than one shard's rows still merges correctly
    dist_big, rows_big = s.search(Q[:2], 40)
    # multi-vector store sharded by cumulative token count
    from autorag_research_amd.sharded import shard_bounds_by_tokens

    tok, off, qtok, qoff = _maxsim_case()
    dlo, dhi = shard_bounds_by_tokens(off, world, rank)
    m = ShardedSearcher(tok.shape[1], "cosine", index_factory=OracleIndex)
    m.add_local_multivec(tok[off[dlo]:off[dhi]], off[dlo:dhi + 1] - off[dlo], dlo)
    md, mr = m.search_maxsim(qtok, qoff, 9)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), d=dist_g, r=rows_g, db=dist_big, rb=rows_big, lo=lo, hi=hi,
             md=md, mr=mr, dlo=dlo, dhi=dhi)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_equals_unsharded(tmp_path, oracle):
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    n, d, B, k = 3001, 48, 9, 12
    C = rng.standard_normal((n, d)).astype(np.float32)
    C[5] = C[2900]
    C[17] = 0.0
    Q = rng.standard_normal((B, d)).astype(np.float32)
    rd, rr = oracle.topk_search(C, Q, k)
    rdb, rrb = oracle.topk_search(C, Q[:2], 40)
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert outs[0]["hi"] == outs[1]["lo"] and outs[0]["lo"] == 0 and outs[1]["hi"] == n
    for o in outs:
        assert np.array_equal(o["r"], rr) and np.array_equal(o["d"], rd, equal_nan=True)
        assert np.array_equal(o["rb"], rrb) and np.array_equal(o["db"], rdb, equal_nan=True)
    tok, off, qtok, qoff = _maxsim_case()
    md, mr = oracle.maxsim_topk(tok, off, qtok, qoff, 9)
    assert outs[0]["dhi"] == outs[1]["dlo"] and outs[0]["dlo"] == 0 and outs[1]["dhi"] == off.shape[0] - 1
    assert 0 < outs[0]["dhi"] < 200  # the token-heavy head makes the first shard SHORTER in docs than half
    for o in outs:
        assert np.array_equal(o["mr"], mr) and np.array_equal(o["md"].view(np.uint32), md.view(np.uint32))


def test_merge_and_bounds_unit():
    from autorag_research_amd.sharded import merge_topk_host, shard_bounds, shard_bounds_by_tokens

    off = np.array([0, 10, 10, 30, 100, 100, 120])
    cuts = [shard_bounds_by_tokens(off, 3, r) for r in range(3)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 6 and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))

    assert [shard_bounds(10_000_000, 8, r, 250_000) for r in (0, 7)] == [(0, 1_250_000), (8_750_000, 10_000_000)]
    assert shard_bounds(5, 8, 7) == (4, 5) and shard_bounds(5, 8, 0) == (0, 0)
    d = np.array([[[0.1, 0.5, np.nan]], [[0.1, 0.2, np.nan]]])      # world=2, B=1, k=3
    r = np.array([[[7, 9, 11]], [[3, 4, -1]]])
    od, orr = merge_topk_host(d, r, 4)
    assert orr.tolist() == [[3, 7, 4, 9]] and od[0, :4].tolist() == [0.1, 0.1, 0.2, 0.5]
    od, orr = merge_topk_host(d, r, 6)
    assert orr.tolist() == [[3, 7, 4, 9, 11, -1]] and np.isnan(od[0, 4]) and np.isnan(od[0, 5])


def _grid_worker(rank: int, world: int, port: int, out_dir: str):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from helpers import OracleIndex
    from autorag_research_amd.sharded import GridLayout, GridSearcher

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(77)
    n, d, B, k = 2203, 40, 23, 7
    C = rng.standard_normal((n, d)).astype(np.float32)
    C[3] = C[2100]  # cross-shard exact tie
    Q = rng.standard_normal((B, d)).astype(np.float32)
    out = {}
    for spec in ("rows", "queries", "2x2", "1x4"):
        lay = GridLayout.parse(spec, world, rank)
        g = GridSearcher(d, lay, "cosine", index_factory=OracleIndex)
        lo, hi = g.local_rows(n, granule=100)
        g.add_local(C[lo:hi], lo)
        dd, rr = g.search(Q, k, block=4)  # 6 blocks dealt to the groups; the last one is ragged (3 queries)
        out[f"d_{spec}"], out[f"r_{spec}"] = dd, rr
        out[f"lay_{spec}"] = np.array([lay.shard, lay.group, lo, hi])
        g.close()
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_grid_layouts_equal_unsharded(tmp_path, oracle):
    """world 4 over gloo: rows-only (4x1), queries-only (1x4) and the mixed 2x2 grid give every rank the unsharded result."""
    import torch.multiprocessing as mp

    from autorag_research_amd.sharded import GridLayout, auto_row_shards, resident_bytes_per_row

    world, port = 4, _free_port()
    mp.spawn(_grid_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    n, d, B, k = 2203, 40, 23, 7
    C = rng.standard_normal((n, d)).astype(np.float32)
    C[3] = C[2100]
    Q = rng.standard_normal((B, d)).astype(np.float32)
    rd, rr = oracle.topk_search(C, Q, k)
    for rank in range(world):
        z = np.load(tmp_path / f"g{rank}.npz")
        for spec in ("rows", "queries", "2x2", "1x4"):
            assert np.array_equal(z[f"r_{spec}"], rr), (rank, spec)
            assert np.array_equal(z[f"d_{spec}"].view(np.uint64), rd.view(np.uint64)), (rank, spec)
        shard, group, lo, hi = z["lay_2x2"]
        assert (shard, group) == (rank % 2, rank // 2)
        assert (lo, hi) == ((0, 1100) if shard == 0 else (1100, 2203))
        assert tuple(z["lay_queries"][2:]) == (0, n)  # one shard: every rank holds the whole corpus
    # the auto rule: fewest row shards whose shard fits in the HBM fraction
    per_row = resident_bytes_per_row(768)
    assert per_row == 4 * 768 + 2 * 768 + 768 + 5
    hbm = 288 * 10**9
    assert auto_row_shards(10_000_000, 768, 8, hbm) == 1
    assert auto_row_shards(60_000_000, 768, 8, hbm) == 2
    assert auto_row_shards(500_000_000, 768, 8, hbm) == 8
    assert GridLayout.parse("auto", 8, 5, 10_000_000, 768, hbm).describe() == "1 row shard(s) x 8 query group(s)"
    with pytest.raises(ValueError):
        GridLayout(8, 0, 3)
