"""GPU, one rank (the build's GPU box has one GPU): the device-resident multi-GPU path end to end at world size 1 --
(1) the library's own RCCL communicator (mi355dr_comm_init + mi355dr_search_sharded_device: local search, ncclAllGather,
merge), (2) ShardedSearcher's pipelined torch.distributed path (gather + merge of block i on a second stream under the
search of block i+1).  Both must reproduce the plain search bit for bit.  The N > 1 logic itself (shard bounds, global
row ids, cross-shard ties, overlapped blocks) runs under gloo in tests/test_sharded_gloo.py."""

import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist
import autorag_research_amd as pkg
from autorag_research_amd.sharded import ShardedSearcher

rng = np.random.default_rng(11)
n, d, B, k = 40000, 256, 700, 10
C = rng.standard_normal((n, d)).astype(np.float32)
Q = rng.standard_normal((B, d)).astype(np.float32)
with pkg.Mi355Index(d) as ref:
    ref.set_option("row_offset", 1000)
    ref.add(C)
    rd, rr = ref.search(Q, k)

# (1) the library's communicator
idx = pkg.Mi355Index(d)
idx.set_option("row_offset", 1000)
idx.add(C)
assert idx.comm_world() == 0
idx.comm_init(0, 1, pkg.Mi355Index.comm_unique_id())
assert idx.comm_world() == 1 and idx.comm_count() == 1   # (comm_count: ncclCommCount, RCCL's own answer)
qd = torch.from_numpy(Q).cuda()
od = torch.empty((B, k), dtype=torch.float64, device="cuda")
orow = torch.empty((B, k), dtype=torch.int64, device="cuda")
idx.search_sharded_device(qd.data_ptr(), B, k, od.data_ptr(), orow.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.array_equal(orow.cpu().numpy(), rr) and np.array_equal(od.cpu().numpy().view(np.uint64), rd.view(np.uint64))
# more than one block per call: block i + 1 is searched while block i's all-gather + merge run on the communication stream
# (two packed buffers), on an explicit non-default stream, twice in a row (the second call reuses the buffers)
Q3 = rng.standard_normal((2500, d)).astype(np.float32)
with pkg.Mi355Index(d) as ref3:
    ref3.set_option("row_offset", 1000)
    ref3.add(C)
    rd3, rr3 = ref3.search(Q3, k)
q3 = torch.from_numpy(Q3).cuda()
o3d = torch.empty((2500, k), dtype=torch.float64, device="cuda")
o3r = torch.empty((2500, k), dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
for _ in range(2):
    o3d.zero_(); o3r.zero_()
    side.wait_stream(torch.cuda.current_stream())
    idx.search_sharded_device(q3.data_ptr(), 2500, k, o3d.data_ptr(), o3r.data_ptr(), side.cuda_stream)
    side.synchronize()   # `side` was made to wait for the last merges
    assert np.array_equal(o3r.cpu().numpy(), rr3) and np.array_equal(o3d.cpu().numpy().view(np.uint64), rd3.view(np.uint64))
idx.close()

# (2) ShardedSearcher over torch.distributed (nccl = RCCL), pipelined blocks
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = ShardedSearcher(d, "cosine", device=0)
s.force_pipeline = True
s.add_local(C, 1000)
d2, r2 = s.search(Q, k, block=256)   # 3 blocks: gathers of blocks 0 and 1 overlap the searches of blocks 1 and 2
assert np.array_equal(r2, rr) and np.array_equal(d2.view(np.uint64), rd.view(np.uint64))
assert s.overlapped_blocks == 2   # blocks 1 and 2 were on the stream before blocks 0 and 1 were waited for
s.close()

# (3) MaxSim with the query vectors and the results in device memory (mi355dr_search_maxsim_device), alone and behind
# ShardedSearcher's nccl path (lists packed, gathered and merged without leaving the device)
dm = 128
lens = rng.integers(1, 90, size=900)
tok = rng.standard_normal((int(lens.sum()), dm)).astype(np.float32)
tok /= np.linalg.norm(tok, axis=1, keepdims=True)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
qlens = [32, 24, 7, 200, 32, 32, 32, 32, 32]          # one query longer than a launch stages; > 8 queries = two passes
qtok = rng.standard_normal((sum(qlens), dm)).astype(np.float32)
qoff = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
mv = pkg.Mi355Index(dm)
mv.set_option("row_offset", 500)
mv.add_multivec(tok, off)
hd, hr = mv.search_maxsim(qtok, qoff, 7)
qt_d = torch.from_numpy(qtok).cuda()
dd = torch.empty((len(qlens), 7), dtype=torch.float32, device="cuda")
dr = torch.empty((len(qlens), 7), dtype=torch.int64, device="cuda")
mv.search_maxsim_device(qt_d.data_ptr(), qoff, 7, dd.data_ptr(), dr.data_ptr(), torch.cuda.current_stream().cuda_stream)
assert np.array_equal(dr.cpu().numpy(), hr) and np.array_equal(dd.cpu().numpy().view(np.uint32), hd.view(np.uint32))
mv.close()
sm = ShardedSearcher(dm, "cosine", device=0)
sm.force_pipeline = True
sm.add_local_multivec(tok, off, 500)
sd, sr = sm.search_maxsim(qtok, qoff, 7)
assert np.array_equal(sr, hr) and np.array_equal(sd.view(np.uint32), hd.view(np.uint32))
# explicit candidates over the sharded store (HEAVEN stage 2 behind a _World): owned ids scored, foreign ids NaN, one all-gather
cand = np.array([[500, 501, 1399, 499, 1400, 777], [900, 500, 1398, 5, 600, 601]], dtype=np.int64)   # 499 / 1400 / 5: not in this shard
sub_q, sub_off = qtok[: qlens[0] + qlens[1]], qoff[:3]
mv2 = pkg.Mi355Index(dm)
mv2.set_option("row_offset", 500)
mv2.add_multivec(tok, off)
want = mv2.maxsim_subset(sub_q, sub_off, cand)
mv2.close()
got = sm.maxsim_subset(sub_q, sub_off, cand)
assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).sum() == 3
assert np.array_equal(got[~np.isnan(got)].view(np.uint32), want[~np.isnan(want)].view(np.uint32))
sm.close()
dist.destroy_process_group()
print("SHARDED_OK")
"""


def test_device_resident_sharded_paths_at_world_one(native_built):
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": str(ROOT), "port": port}], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
