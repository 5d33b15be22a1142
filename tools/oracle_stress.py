"""Developer check: is the multi-threaded CPU oracle deterministic on this host?  (replays fuzz case 77000435)"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import cpu_ref  # noqa: E402

rng = np.random.default_rng(77000435)
rng.random()
n = int(rng.choice([1, 7, 300, 4000, 30000, 120000, 250000]))
n = max(1, int(n * rng.uniform(0.5, 1.0)))
d = int(rng.choice([5, 16, 64, 100, 128, 256, 384, 768, 1000]))
B = int(min(max(1, int(4e10 / (n * d))), rng.choice([1, 3, 33, 128, 129, 300, 777, 1024, 1500])))
k = int(rng.choice([1, 3, 10, 24, 25, 64, 100, 300, 1024]))
mode = str(rng.choice(["gauss", "scaled", "clustered", "dups", "spiky", "dirty"]))
rng.random()
C = rng.standard_normal((n, d)).astype(np.float32)
with np.errstate(all="ignore"):
    for v in (0.0, np.nan, np.inf, 1e-25, 1e25):
        C[rng.integers(0, n)] = v if v == 0.0 else C[rng.integers(0, n)] * 0 + v
Q = rng.standard_normal((B, d)).astype(np.float32)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "oracle threads", cpu_ref.num_threads(), n, d, B, k, mode)
rd0, rr0 = cpu_ref.topk_search(C, Q, k, threads=1)
n20 = np.array([cpu_ref.dot(C[i], C[i]) for i in range(n)], dtype=np.float32)
bad = badn = 0
t0 = time.time()
for it in range(400):
    junk = [np.full(int(s), 0.0, np.float32) for s in np.random.default_rng(it).integers(50, 400, size=20)]  # zeroed heap chunks
    del junk
    rd, rr = cpu_ref.topk_search(C, Q, k)
    n2 = cpu_ref.row_nrm2(C)
    if not np.array_equal(n2.view(np.uint32), n20.view(np.uint32)):
        badn += 1
        w = np.argwhere(n2.view(np.uint32) != n20.view(np.uint32))[:, 0]
        if badn < 4:
            print("NRM2 DIFF it", it, "rows", w[:6], n2[w[:6]], n20[w[:6]])
    if not np.array_equal(rr, rr0):
        bad += 1
        q = int(np.argwhere(rr != rr0)[0][0])
        if bad < 4:
            print("TOPK DIFF it", it, "query", q, rr[q][:4], rd[q][:4], "vs", rr0[q][:4], rd0[q][:4])
print("topk diffs", bad, "nrm2 diffs", badn, "of 400 in", round(time.time() - t0, 1), "s")
