"""GPU: `bench.py --gpus 2` launched exactly as the driver launches it for N > 1 (`python -m torch.distributed.run --nnodes=1
--nproc-per-node 2 --master-addr 127.0.0.1 ...`) -- on ONE GPU, with `--rehearse-one-gpu` (gloo instead of RCCL, which refuses two
ranks on one device; device tensors of the collectives through the host).  The bench's multi-rank leg had never run with a second
rank: layout, per-rank row ranges and planted rows, the barrier / max-over-ranks timing, the packed all-gather + merge inside the
timed loop on a second stream, `identical_to_one_gpu`.  Checked: ONE JSON line from rank 0 with the contract's keys, n_gpus = 2,
strong scaling, both ranks in the all-gather, the sharded answer identical to one GPU's, no fallbacks.  (Not a measurement.)"""

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("k", [10, 100])
def test_bench_multi_rank_leg_runs_with_two_ranks_on_one_gpu(native_built, k):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rows", "1000000",
           "--k", str(k), "--rehearse-one-gpu", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["metric"] == "queries/sec" and out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert out["scaling"] == "strong" and out["higher_is_better"] is True and out["value"] > 0
    cfg = out["config"]
    assert cfg["rows_total"] == 1_000_000 and cfg["rows_per_gpu"] == 500_000 and cfg["k"] == k
    assert cfg["layout"]["row_shards"] == 2 and cfg["collective"]["ranks_in_all_gather"] == 2
    assert "REHEARSAL" in cfg["collective"]["transport"] and cfg["collective"]["search_stream_handle_nonzero"]
    assert out["extra"]["fallback_queries"] == 0
    assert out["extra"]["identical_to_one_gpu"] is True, out["extra"]
    assert out["ndcg_at_10"]["or_group"] > 0.5
