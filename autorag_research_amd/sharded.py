"""Multi-GPU search: one process per GPU, a (row shards x query groups) grid of ranks.

Row sharding (ShardedSearcher): local exact top-k, one all-gather, one merge.

The reference has no counterpart (its engine is a single PostgreSQL server, SURVEY.md section 2 rows
32-34); this is the data-parallel form of `ORDER BY distance LIMIT k`: rank r owns a contiguous row range
and returns GLOBAL row ids (local + row_offset), every rank all-gathers the [B,k] (float8 distance,
int64 row) lists -- B*k*16 bytes per rank, e.g. 160 KiB at B=1024,k=10 -- and merges world*k -> k under
the same total order (distance asc, NaN last, row asc), so the result is bit-identical to the
single-GPU result.  `torch.distributed` is the transport: backend "nccl" is RCCL over xGMI on the GPU
box (tensors stay on the device and the merge runs in libmi355dr), "gloo" on CPU tests (host merge).

Query groups (GridLayout / GridSearcher): a rank holds as much of the corpus as its HBM takes -- 288 GB per MI355X is ~50 M
fp32 rows of d = 768 with both screen copies -- so the corpus is cut into only as many row shards R as it NEEDS, and the
world / R groups of R ranks each serve their own query blocks: independent units, no collective between groups (the
row-shard exchange stays inside a group).  With R = 1 there is no data-path collective at all.  Why it matters: the
per-pass costs that do not shrink with the shard (candidate appends, the exact re-score launches)
make 8 shards of 1.25 M rows cost 1.9 ms per 1024-query pass where 1/8 of the 10 M-row pass would be 1.05 ms.
"""

from __future__ import annotations

from collections.abc import Callable
from typing import Any

import numpy as np


def shard_bounds(n_rows: int, world: int, rank: int, granule: int = 1) -> tuple[int, int]:
    """Contiguous [lo, hi) of rank's rows; boundaries are multiples of `granule` (except the end)."""
    units = (n_rows + granule - 1) // granule
    lo = min(n_rows, units * rank // world * granule)
    hi = min(n_rows, units * (rank + 1) // world * granule)
    return lo, hi


def shard_bounds_by_tokens(offsets, world: int, rank: int) -> tuple[int, int]:
    """Contiguous doc range [lo, hi) of a multi-vector store with about 1/world of the TOKENS (the MaxSim pass streams
    token rows, so tokens -- not docs -- are the unit of work)."""
    off = np.asarray(offsets, dtype=np.int64)
    n_docs, total = off.shape[0] - 1, int(off[-1])
    cut = lambda r: int(np.searchsorted(off, total * r // world, side="left")) if r < world else n_docs  # noqa: E731
    lo, hi = min(cut(rank), n_docs), min(cut(rank + 1), n_docs)
    return lo, max(lo, hi)


def merge_topk_host(dist_all: np.ndarray, rows_all: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """[world,B,k] shard lists -> [B,k] under (distance asc, NaN last, row asc); pads with NaN / -1."""
    world, B, kk = dist_all.shape
    d = np.transpose(dist_all, (1, 0, 2)).reshape(B, world * kk)
    r = np.transpose(rows_all, (1, 0, 2)).reshape(B, world * kk)
    out_d = np.full((B, k), np.nan)
    out_r = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        valid = r[b] >= 0
        db, rb = d[b][valid], r[b][valid]
        nan = np.isnan(db)
        order = np.lexsort((rb, np.where(nan, 0.0, db), nan))[:k]
        out_d[b, : order.size] = db[order]
        out_r[b, : order.size] = rb[order]
    return out_d, out_r


def resident_bytes_per_row(dim: int) -> int:
    """HBM bytes one stored row costs in libmi355dr: fp32 row + bf16 shadow (64-element pad) + int8 shadow (128-element
    pad) + squared norm + int8 flag."""
    return 4 * dim + 2 * ((dim + 63) // 64 * 64) + ((dim + 127) // 128 * 128) + 5


def auto_row_shards(n_rows: int, dim: int, world: int, hbm_bytes: int, fraction: float = 0.6) -> int:
    """Fewest row shards R (a divisor of `world`) whose shard fits in `fraction` of one GPU's HBM."""
    need = n_rows * resident_bytes_per_row(dim)
    for r in range(1, world + 1):
        if world % r == 0 and need / r <= fraction * hbm_bytes:
            return r
    return world


class GridLayout:
    """rank -> (shard, group) for a world of `row_shards` x `query_groups` ranks; ranks of one group are consecutive
    (rank = group * row_shards + shard), so a group's row-shard exchange stays between neighbouring GPUs."""

    def __init__(self, world: int, rank: int, row_shards: int):
        if row_shards < 1 or world % row_shards:
            raise ValueError(f"row_shards={row_shards} does not divide the world size {world}")
        self.world, self.rank, self.row_shards = world, rank, row_shards
        self.query_groups = world // row_shards
        self.shard, self.group = rank % row_shards, rank // row_shards

    @classmethod
    def parse(cls, spec: str, world: int, rank: int, n_rows: int = 0, dim: int = 0, hbm_bytes: int = 0) -> "GridLayout":
        """'auto' (fewest row shards that fit), 'rows' (world x 1), 'queries' (1 x world) or 'RxQ'."""
        if spec == "auto":
            return cls(world, rank, auto_row_shards(n_rows, dim, world, hbm_bytes) if hbm_bytes else 1)
        if spec == "rows":
            return cls(world, rank, world)
        if spec == "queries":
            return cls(world, rank, 1)
        r, q = (int(x) for x in spec.lower().split("x"))
        if r * q != world:
            raise ValueError(f"layout {spec} does not match the world size {world}")
        return cls(world, rank, r)

    def group_ranks(self, group: int | None = None) -> list[int]:
        g = self.group if group is None else group
        return list(range(g * self.row_shards, (g + 1) * self.row_shards))

    def make_row_group(self, dist) -> Any:
        """The process group of this rank's row shards (every rank creates every group: torch.distributed's rule).
        None when the group is the whole world or a single rank."""
        if self.row_shards == self.world:
            return None
        mine = None
        for g in range(self.query_groups):
            pg = dist.new_group(self.group_ranks(g)) if self.row_shards > 1 else None
            if g == self.group:
                mine = pg
        return mine

    def describe(self) -> str:
        return f"{self.row_shards} row shard(s) x {self.query_groups} query group(s)"


class ShardedSearcher:
    """One rank's view of a row-sharded corpus."""

    def __init__(self, dim: int, metric: str = "cosine", device: int = 0, index_factory: Callable[..., Any] | None = None,
                 group: Any | None = None):
        import torch.distributed as dist

        if index_factory is None:
            from .index import Mi355Index

            index_factory = Mi355Index
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.index = index_factory(dim, metric, device)
        self.device = device
        self.row_offset = 0
        self.force_pipeline = False  # tests: run the gather + merge pipeline at world size 1 too
        # gloo with a GPU-resident index (ranks that cannot meet in RCCL -- e.g. two ranks on ONE device, which RCCL refuses --
        # or a node without xGMI between them): the shard's packed lists and the merge stay on the device exactly as under
        # nccl, only the exchange takes a host hop (D2H, gloo all-gather, H2D).  Off for index stand-ins without device entry
        # points (tests/helpers.OracleIndex): they take the host pipeline.
        self.host_hop = bool(self.backend == "gloo" and hasattr(self.index, "search_device_async")
                             and hasattr(self.index, "merge_topk_packed_device"))
        self.overlapped_blocks = 0   # blocks whose gather ran under the next block's search (host pipeline; tests)

    def add_local(self, rows, global_row0: int) -> None:
        """Add this rank's rows; `global_row0` is the global index of its first row (set once)."""
        if len(self.index) == 0:
            self.row_offset = int(global_row0)
            self.index.set_option("row_offset", self.row_offset)
        self.index.add(rows)

    def search(self, queries, k: int, block: int = 1024) -> tuple[np.ndarray, np.ndarray]:
        """Every rank passes the same queries; every rank gets the same global [B,k] result.

        Queries are served in blocks of `block`; the all-gather + merge of block i runs while block i+1 is searched
        (second stream + double-buffered packed blocks on the GPU, an asynchronous collective on CPU/gloo), so the
        collective's latency and the merge are off the critical path for every block but the last.  On the nccl
        backend the shard's list never leaves the device before the gather: the library writes it straight into the
        packed [2,B,k] block (plane 0 = float8 distance bits, plane 1 = global rows) that ONE all-gather sends."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if self.world == 1 and not self.force_pipeline:
            return self.index.search(q, k)
        if (self.backend == "nccl" or self.host_hop) and hasattr(self.index, "search_device"):
            return self._search_device_pipelined(q, k, block)
        return self._search_host_pipelined(q, k, block)

    def _search_device_pipelined(self, q: np.ndarray, k: int, block: int):
        import torch

        dist, dev = self._dist, torch.device("cuda", self.device)
        B = q.shape[0]
        qd = torch.from_numpy(q).to(dev)
        out_d = torch.empty((B, k), dtype=torch.float64, device=dev)
        out_r = torch.empty((B, k), dtype=torch.int64, device=dev)
        nbmax = min(block, B)
        packed = [torch.empty((2 * nbmax * k,), dtype=torch.int64, device=dev) for _ in range(2)]
        gathered = [torch.empty((self.world * 2 * nbmax * k,), dtype=torch.int64, device=dev) for _ in range(2)]
        # the searches run on an explicit NON-default stream: its handle is never 0, so the library uses this very stream
        # (a null handle means "the index's own stream", which torch events would not see)
        compute = torch.cuda.Stream(dev)
        comm = torch.cuda.Stream(dev)
        compute.wait_stream(torch.cuda.current_stream(dev))  # the upload of the queries
        done = [None, None]
        use_async = hasattr(self.index, "search_device_async")

        def finish(p):
            """block complete on the host (the library re-did the rare flagged queries) -> gather + merge on the second stream"""
            ticket, buf, b0, nb = p
            if ticket is not None:
                self.index.search_wait(ticket)
            pk = packed[buf][: 2 * nb * k]
            ga_host = None
            if self.host_hop:   # the block is final (search_wait / the synchronize below): host hop of the exchange
                if ticket is None:
                    compute.synchronize()
                pk_host = pk.cpu()
                ga_host = torch.empty((self.world * pk_host.numel(),), dtype=torch.int64)
                dist.all_gather_into_tensor(ga_host, pk_host, group=self.group)
            with torch.cuda.stream(comm):
                if ticket is None:
                    comm.wait_stream(compute)
                ga = gathered[buf][: self.world * 2 * nb * k]
                if ga_host is not None:
                    ga.copy_(ga_host)
                else:
                    dist.all_gather_into_tensor(ga, pk, group=self.group)
                self.index.merge_topk_packed_device(ga.data_ptr(), self.world, nb, k, out_d[b0:b0 + nb].data_ptr(),
                                                    out_r[b0:b0 + nb].data_ptr(), comm.cuda_stream)
                done[buf] = torch.cuda.Event()
                done[buf].record(comm)

        pend = None
        for i, b0 in enumerate(range(0, B, block)):
            nb, buf = min(block, B - b0), i & 1
            if done[buf] is not None:
                compute.wait_event(done[buf])  # the gather that read this packed block two blocks ago
            pk = packed[buf][: 2 * nb * k]
            args = (qd[b0:b0 + nb].data_ptr(), nb, k, pk.data_ptr(), pk[nb * k:].data_ptr(), compute.cuda_stream)
            if use_async:   # block i is on the stream before the host waits for block i - 1
                t = self.index.search_device_async(*args)
                if pend is not None:
                    finish(pend)
                    self.overlapped_blocks += 1
                pend = (t, buf, b0, nb)
            else:
                self.index.search_device(*args)
                finish((None, buf, b0, nb))
        if pend is not None:
            finish(pend)
        comm.synchronize()
        return out_d.cpu().numpy(), out_r.cpu().numpy()

    def _search_host_pipelined(self, q: np.ndarray, k: int, block: int):
        """CPU / gloo form of the same pipeline (tests): asynchronous all-gather of block i, local search of block i+1,
        then wait + host merge of block i."""
        import torch

        dist = self._dist
        B = q.shape[0]
        out_d = np.full((B, k), np.nan)
        out_r = np.full((B, k), -1, dtype=np.int64)
        pending = None

        def finish(p):
            work, ga, b0, nb = p
            work.wait()
            g = ga.numpy().reshape(self.world, 2, nb, k)
            d, r = merge_topk_host(np.ascontiguousarray(g[:, 0]).view(np.float64), np.ascontiguousarray(g[:, 1]), k)
            out_d[b0:b0 + nb], out_r[b0:b0 + nb] = d, r

        for b0 in range(0, B, block):
            nb = min(block, B - b0)
            dist_l, rows_l = self.index.search(q[b0:b0 + nb], k)
            if pending is not None:
                finish(pending)
            pk = torch.from_numpy(np.stack([np.ascontiguousarray(dist_l).view(np.int64), rows_l]).reshape(-1))
            ga = torch.empty((self.world * pk.numel(),), dtype=torch.int64)
            work = dist.all_gather_into_tensor(ga, pk, group=self.group, async_op=True)
            pending = (work, ga, b0, nb)
            self.overlapped_blocks += 1 if b0 + block < B else 0
        finish(pending)
        return out_d, out_r

    # ---- multi-vector (MaxSim): docs sharded by cumulative token count, same gather + merge ----
    def add_local_multivec(self, vecs, offsets, global_doc0: int) -> None:
        """Add this rank's docs (ragged [sum_T, d] + offsets starting at 0); `global_doc0` = global index of its first doc."""
        self.row_offset = int(global_doc0)
        self.index.set_option("row_offset", self.row_offset)
        self.index.add_multivec(vecs, offsets)

    def search_maxsim(self, qtok, q_offsets, k: int) -> tuple[np.ndarray, np.ndarray]:
        """Every rank passes the same queries; every rank gets the same global [B,k] (fp32 distance, doc) result."""
        if self.world == 1 and not self.force_pipeline:
            return self.index.search_maxsim(qtok, q_offsets, k)
        import torch

        on_gpu = self.backend == "nccl" or self.host_hop
        if on_gpu and hasattr(self.index, "search_maxsim_device"):
            # nccl: the shard's lists stay in HBM from the MaxSim kernels to the merge -- mi355dr_search_maxsim_device writes
            # them to device buffers, fp32 -> float8 (exact, order-preserving) and the packing are device ops, one
            # all-gather, k_merge_topk
            dev = torch.device("cuda", self.device)
            q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int32)
            B = q_offsets.shape[0] - 1
            # Everything on ONE real (non-default) stream.  The library works on a stream of its own, created non-blocking: torch's
            # default stream (handle 0, which the library reads as "no stream to wait for") is ordered against it by nothing.  With
            # the default stream here, the merge could start before the gathered lists' H2D copy had landed and `out_d.cpu()` could
            # read before the merge had run -- a lost race shows only when the device is contended (two ranks on one GPU: a wrong
            # list in one run of seven, round 6).  A real handle is ordered by the library itself (its stream waits for the caller's,
            # the merge is launched ON the caller's), and the host waits for this stream before it reads.
            if getattr(self, "_ms_stream", None) is None:
                self._ms_stream = torch.cuda.Stream(dev)
            side = self._ms_stream
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                qd = torch.from_numpy(np.ascontiguousarray(qtok, dtype=np.float32)).to(dev)
                d32 = torch.empty((B, k), dtype=torch.float32, device=dev)
                packed = torch.empty((2, B, k), dtype=torch.int64, device=dev)
                self.index.search_maxsim_device(qd.data_ptr(), q_offsets, k, d32.data_ptr(), packed[1].data_ptr(), side.cuda_stream)
                packed[0].view(torch.float64).copy_(d32)
                gathered = torch.empty((self.world, 2, B, k), dtype=torch.int64, device=dev)
                if self.host_hop:
                    g_host = torch.empty((self.world * packed.numel(),), dtype=torch.int64)
                    self._dist.all_gather_into_tensor(g_host, packed.view(-1).cpu(), group=self.group)
                    gathered.view(-1).copy_(g_host)
                else:
                    self._dist.all_gather_into_tensor(gathered.view(-1), packed.view(-1), group=self.group)
                out_d = torch.empty((B, k), dtype=torch.float64, device=dev)
                out_r = torch.empty((B, k), dtype=torch.int64, device=dev)
                self.index.merge_topk_packed_device(gathered.data_ptr(), self.world, B, k, out_d.data_ptr(), out_r.data_ptr(),
                                                    side.cuda_stream)
                side.synchronize()
                res = out_d.cpu().numpy().astype(np.float32), out_r.cpu().numpy()
            torch.cuda.current_stream(dev).wait_stream(side)   # (the caching allocator may hand these blocks to the default stream next)
            return res
        dist_l, rows_l = self.index.search_maxsim(qtok, q_offsets, k)
        B = dist_l.shape[0]
        # fp32 -> float8 is exact and order-preserving, so the float8 merge applies unchanged
        d64 = np.ascontiguousarray(dist_l.astype(np.float64))
        packed = torch.from_numpy(np.stack([d64.view(np.int64), rows_l]))
        gathered = torch.empty((self.world, 2, B, k), dtype=torch.int64)
        self._dist.all_gather_into_tensor(gathered.view(-1), packed.view(-1), group=self.group)
        g = gathered.numpy()
        out_d, out_r = merge_topk_host(np.ascontiguousarray(g[:, 0]).view(np.float64), np.ascontiguousarray(g[:, 1]), k)
        return out_d.astype(np.float32), out_r

    def maxsim_subset(self, qtok, q_offsets, doc_ids, clamp0: bool = False) -> np.ndarray:
        """Late-interaction distance of EXPLICIT candidates (global doc ids, [B,m]) over a token-sharded store: every rank
        scores the candidates it owns (mi355dr_maxsim_subset leaves NaN for ids outside its shard), ONE all-gather of the
        [B,m] fp32 block, and the owner's value wins.  Every rank passes the same arguments and gets the same [B,m] result --
        bit for bit what one store holding every doc returns (a candidate is scored by exactly one rank, with the same kernel).
        Callers: HEAVEN stage 2 (reference heaven.py:244-266), the ColBERT reranker form with `clamp0`."""
        ids = np.ascontiguousarray(doc_ids, dtype=np.int64)
        if ids.ndim == 1:
            ids = ids[None, :]
        kw = {"clamp0": True} if clamp0 else {}
        local = np.ascontiguousarray(self.index.maxsim_subset(qtok, q_offsets, ids, **kw), dtype=np.float32)
        if self.world == 1 and not self.force_pipeline:
            return local
        import torch

        dev = torch.device("cuda", self.device) if self.backend == "nccl" else torch.device("cpu")
        mine = torch.from_numpy(local).to(dev)
        gathered = torch.empty((self.world,) + tuple(mine.shape), dtype=torch.float32, device=dev)
        self._dist.all_gather_into_tensor(gathered.view(-1), mine.reshape(-1), group=self.group)
        g = gathered.cpu().numpy()
        out = g[0].copy()
        for r in range(1, self.world):   # NaN = "not mine" (or a genuinely undefined score, which then stays NaN)
            out = np.where(np.isnan(out), g[r], out)
        return out

    def close(self) -> None:
        self.index.close()


class GridSearcher:
    """One rank of a (row shards x query groups) grid.  Every rank passes the same queries and gets the same [B,k] result:
    query blocks are dealt round-robin to the groups, each group answers its blocks with its row-sharded searcher (a plain
    index when it has one shard), and one all-gather of the finished [B,k] lists (16 B per entry) hands every rank the whole
    result -- the only traffic between groups, after the search."""

    def __init__(self, dim: int, layout: GridLayout, metric: str = "cosine", device: int = 0,
                 index_factory: Callable[..., Any] | None = None):
        import torch.distributed as dist

        self._dist, self.layout, self.device = dist, layout, device
        self.backend = dist.get_backend() if dist.is_initialized() else None
        if layout.row_shards > 1:
            group = layout.make_row_group(dist)
            self.rows = ShardedSearcher(dim, metric, device, index_factory, group=group)
            if self.rows.world != layout.row_shards:
                raise RuntimeError("row group size does not match the layout")
        else:
            layout.make_row_group(dist)  # (no groups to create; kept for symmetry)
            self.rows = _SingleShard(dim, metric, device, index_factory)
        self.index = self.rows.index

    def local_rows(self, n_rows: int, granule: int = 1) -> tuple[int, int]:
        """Global row range [lo, hi) this rank stores."""
        return shard_bounds(n_rows, self.layout.row_shards, self.layout.shard, granule)

    def add_local(self, rows, global_row0: int) -> None:
        self.rows.add_local(rows, global_row0)

    def search(self, queries, k: int, block: int = 1024) -> tuple[np.ndarray, np.ndarray]:
        import torch

        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        B, Q = q.shape[0], self.layout.query_groups
        if Q == 1:
            return self.rows.search(q, k, block)
        n_blocks = (B + block - 1) // block
        mine = [b for b in range(n_blocks) if b % Q == self.layout.group]
        per_group = (n_blocks + Q - 1) // Q * block  # padded number of queries one group can own
        d_l = np.full((per_group, k), np.nan)
        r_l = np.full((per_group, k), -1, dtype=np.int64)
        if mine:
            sub = np.concatenate([q[b * block:(b + 1) * block] for b in mine], axis=0)
            dd, rr = self.rows.search(sub, k, block)
            d_l[: dd.shape[0]], r_l[: rr.shape[0]] = dd, rr
        dev = torch.device("cuda", self.device) if self.backend == "nccl" else torch.device("cpu")
        pk = torch.from_numpy(np.stack([d_l.view(np.int64), r_l])).to(dev)
        ga = torch.empty((self.layout.world,) + tuple(pk.shape), dtype=torch.int64, device=dev)
        self._dist.all_gather_into_tensor(ga.view(-1), pk.view(-1))
        g = ga.cpu().numpy()
        out_d = np.full((B, k), np.nan)
        out_r = np.full((B, k), -1, dtype=np.int64)
        for b in range(n_blocks):
            src = (b % Q) * self.layout.row_shards  # any rank of the owning group holds the merged list: take its first
            pos = (b // Q) * block
            n = min(block, B - b * block)
            out_d[b * block:b * block + n] = g[src, 0, pos:pos + n].view(np.float64)
            out_r[b * block:b * block + n] = g[src, 1, pos:pos + n]
        return out_d, out_r

    def close(self) -> None:
        self.rows.close()


class _SingleShard:
    """A group of one rank: the whole corpus in one index, no exchange."""

    def __init__(self, dim, metric, device, index_factory):
        if index_factory is None:
            from .index import Mi355Index

            index_factory = Mi355Index
        self.index = index_factory(dim, metric, device)
        self.world, self.rank = 1, 0

    def add_local(self, rows, global_row0: int) -> None:
        if int(global_row0) != 0 and len(self.index) == 0:
            self.index.set_option("row_offset", int(global_row0))
        self.index.add(rows)

    def search(self, q, k: int, block: int = 1024):
        return self.index.search(q, k)

    def close(self) -> None:
        self.index.close()
