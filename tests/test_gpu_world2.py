"""GPU, TWO ranks on ONE MI355X (the build's GPU boxes have one GPU; RCCL refuses two ranks on one device): the row-sharded
path with a real second producer.  Two processes, each a real `Mi355Index` over its half of the rows (`row_offset` = global
ids), meet over gloo:
  (1) `ShardedSearcher.search(block=...)` -- the overlapped two-buffer pipeline: packed [2,B,k] blocks written by the library on
      the device, exchanged (host hop), merged by `k_merge_topk` on the device -- at k = 10 and k = 100;
  (2) the LIBRARY's own sharded entry point, `mi355dr_search_sharded_device` (csrc/mi355dr_comm.hip: two packed buffers, the
      communication stream, tickets), over `mi355dr_comm_init_custom` with a gloo all-gather as the host's transport: three
      blocks per call, so block i + 1 is searched while block i is gathered and merged;
  (3) MaxSim over a store split by cumulative token count.
Every rank's result == the unsharded GPU index == the CPU oracle, ids and distance bits (BASELINE configs C3 / C4)."""

import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import autorag_research_amd as pkg
from autorag_research_amd.sharded import ShardedSearcher, shard_bounds, shard_bounds_by_tokens
from oracle import cpu_ref

rank, world = int(os.environ["RANK"]), 2
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)


def same(d, r, rd, rr):
    return np.array_equal(r, rr) and np.array_equal(np.isnan(d), np.isnan(rd)) and \
        np.array_equal(d[~np.isnan(d)].view(np.uint64), rd[~np.isnan(rd)].view(np.uint64))


rng = np.random.default_rng(77)
n, d, B = 60_000, 256, 700
C = rng.standard_normal((n, d)).astype(np.float32)
C[5] = C[n - 100]          # an exact tie ACROSS the shards: the lower global row wins on every rank
C[17] = 0.0                # a NaN distance on shard 0
Q = rng.standard_normal((B, d)).astype(np.float32)
Q[3] = C[5]                # ... and a query that looks straight at the tie
lo, hi = shard_bounds(n, world, rank, granule=256)
assert 0 < hi - lo < n
ref = {k: cpu_ref.topk_search(C, Q, k) for k in (10, 100)}
with pkg.Mi355Index(d) as whole:
    whole.add(C)
    for k in (10, 100):
        gd, gr = whole.search(Q, k)
        assert same(gd, gr, *ref[k]), f"unsharded GPU != oracle at k={k}"

# (1) ShardedSearcher: device-resident lists, host-hop exchange, device merge; 3 blocks -> 2 overlapped
s = ShardedSearcher(d, "cosine", device=0)
assert s.world == 2 and s.host_hop
s.add_local(C[lo:hi], lo)
for k in (10, 100):
    s.overlapped_blocks = 0
    sd, sr = s.search(Q, k, block=256)
    assert s.overlapped_blocks == 2
    assert same(sd, sr, *ref[k]), f"rank {rank}: sharded (host hop) != oracle at k={k}"
assert sr[3][0] == 5 and (n - 100) in sr[3][:2].tolist()
s.close()

# (2) the library's sharded entry point over the host's all-gather (gloo)
idx = pkg.Mi355Index(d)
idx.set_option("row_offset", lo)
idx.add(C[lo:hi])
calls = []


def all_gather(send_ptr, recv_ptr, nbytes, stream):
    assert nbytes %% 8 == 0 and stream != 0      # (the index's communication stream)
    mine = np.empty(nbytes // 8, dtype=np.int64)
    idx.dev_download(send_ptr, mine)
    out = torch.empty(world * mine.size, dtype=torch.int64)
    dist.all_gather_into_tensor(out, torch.from_numpy(mine))
    idx.dev_upload(recv_ptr, out.numpy())
    calls.append(nbytes)


idx.comm_init_custom(rank, world, all_gather)
assert idx.comm_world() == 2 and idx.comm_count() == 2
Bq = 2500                                       # three blocks of <= 1024: the two packed buffers are both reused
Q3 = np.concatenate([Q, rng.standard_normal((Bq - B, d)).astype(np.float32)])
for k in (10, 100):
    rd3, rr3 = cpu_ref.topk_search(C, Q3, k)
    qd = torch.from_numpy(Q3).cuda()
    od = torch.zeros((Bq, k), dtype=torch.float64, device="cuda")
    orow = torch.zeros((Bq, k), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    del calls[:]
    for rep in range(2):                        # (the second call finds the buffers armed by the first)
        od.zero_(); orow.zero_()
        side.wait_stream(torch.cuda.current_stream())
        idx.search_sharded_device(qd.data_ptr(), Bq, k, od.data_ptr(), orow.data_ptr(), side.cuda_stream)
        side.synchronize()
        assert same(od.cpu().numpy(), orow.cpu().numpy(), rd3, rr3), f"rank {rank}: mi355dr_search_sharded_device != oracle at k={k}"
    assert calls == [2 * 1024 * k * 8, 2 * 1024 * k * 8, 2 * (Bq - 2048) * k * 8] * 2
idx.close()

# (3) MaxSim: the store split by cumulative token count (a token-heavy head: the first shard is shorter in docs)
dm = 128
lens = rng.integers(1, 40, size=3000)
lens[:200] = rng.integers(150, 260, size=200)
tok = rng.standard_normal((int(lens.sum()), dm)).astype(np.float32)
tok /= np.linalg.norm(tok, axis=1, keepdims=True)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
qlens = [32] * 9 + [24, 7]
qtok = rng.standard_normal((sum(qlens), dm)).astype(np.float32)
qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
qoff = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
md, mr = cpu_ref.maxsim_topk(tok, off, qtok, qoff, 10)
dlo, dhi = shard_bounds_by_tokens(off, world, rank)
assert 0 < dhi - dlo < len(lens) and (rank == 1 or dhi < len(lens) // 2)
m = ShardedSearcher(dm, "cosine", device=0)
m.add_local_multivec(tok[off[dlo]:off[dhi]], off[dlo:dhi + 1] - off[dlo], dlo)
gd, gr = m.search_maxsim(qtok, qoff, 10)
assert np.array_equal(gr, mr) and np.array_equal(gd.view(np.uint32), md.view(np.uint32)), f"rank {rank}: sharded MaxSim != oracle"
cand = np.stack([mr[0], mr[1][::-1]])            # explicit candidates from both shards: the owner's score wins
sub = m.maxsim_subset(qtok[: qoff[2]], qoff[:3], cand)
assert np.array_equal(sub[0].view(np.uint32), md[0].view(np.uint32)) and np.array_equal(sub[1][::-1].view(np.uint32), md[1].view(np.uint32))
m.close()
dist.barrier()
dist.destroy_process_group()
print("WORLD2_OK", rank)
"""


def _run_pair():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    script = _CHILD % {"root": str(ROOT)}
    procs = []
    for rank in range(2):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank),
                   WORLD_SIZE="2", OMP_NUM_THREADS="8")
        procs.append(subprocess.Popen([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for pp in procs:
                pp.kill()
            raise
    bad = [rank for rank, (p, (so, _)) in enumerate(zip(procs, outs)) if p.returncode != 0 or f"WORLD2_OK {rank}" not in so]
    # (the rank that failed FIRST is the one to read: its peer then dies in the next collective)
    return "\n".join(f"rank {rank}: exit code {procs[rank].returncode}\n--- stdout\n{outs[rank][0][-800:]}\n--- stderr\n"
                     f"{outs[rank][1][-3500:]}" for rank in bad)


def test_two_ranks_on_one_gpu_equal_one_gpu_and_the_oracle(native_built, oracle):
    report = _run_pair()
    if report and "AssertionError" not in report and "NativeError" not in report:
        # a rank lost to the rendezvous / the transport is not a verdict on the search: one more attempt; a parity failure (an
        # assert, a library error) never retries.  (Round 6: "the peer's gloo pair was closed, once in 8 runs" WAS a parity failure
        # of the other rank -- sharded MaxSim merged on the library's stream, unordered against torch's default stream: fixed in
        # sharded.search_maxsim, 40 runs in a row clean since.)
        report = _run_pair()
    if report:
        pytest.fail(report)
