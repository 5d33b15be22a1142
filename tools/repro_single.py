"""Replay one single-vector fuzz case by seed and print where the GPU and the oracle differ (developer tool)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import autorag_research_amd as pkg  # noqa: E402
import fuzz_parity as fz  # noqa: E402
from oracle import cpu_ref  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
rng.random()
# same draws as check_single
n = int(rng.choice([1, 7, 300, 4000, 30000, 120000, 250000]))
n = max(1, int(n * rng.uniform(0.5, 1.0)))
d = int(rng.choice([5, 16, 64, 100, 128, 256, 384, 768, 1000]))
Bmax = max(1, int(4e10 / (n * d)))
B = int(min(Bmax, rng.choice([1, 3, 33, 128, 129, 300, 777, 1024, 1500])))
k = int(rng.choice([1, 3, 10, 24, 25, 64, 100, 300, 1024]))
mode = str(rng.choice(["gauss", "scaled", "clustered", "dups", "spiky", "dirty"]))
metric = "ip" if rng.random() < fz.IP_PROB else "cosine"
C = fz.corpus(rng, n, d, mode)
Q = rng.standard_normal((B, d)).astype(np.float32)
if mode in ("clustered", "dups") and B > 2:
    Q[: B // 2] = C[rng.integers(0, n, size=B // 2)] + (0.01 * rng.standard_normal((B // 2, d))).astype(np.float32)
opts = {}
if rng.random() < 0.5:
    opts["screen_dtype"] = str(rng.choice(["auto", "bf16", "i8"]))
if rng.random() < 0.2:
    opts["cand_cap"] = int(rng.choice([64, 300, 1024]))
if rng.random() < 0.2:
    opts["chunk_growth"] = int(rng.choice([1, 2, 5, 7]))
if rng.random() < 0.2:
    opts["chunk0_rows"] = int(rng.choice([256, 512, 2048]))
if rng.random() < 0.25:
    opts["prefilter16"] = 1
row_offset = int(rng.choice([0, 0, 12345, 2**33]))
print(f"n={n} d={d} B={B} k={k} mode={mode} metric={metric} opts={opts} off={row_offset}")
special = [i for i in range(n) if not np.isfinite(C[i]).all() or (C[i] == 0).all() or np.abs(C[i]).max() > 1e20 or np.abs(C[i]).max() < 1e-20]
print("special rows:", [(i, float(C[i][0])) for i in special])
cut = int(rng.integers(0, n + 1))
rd, rr = cpu_ref.topk_search(C, Q, k, metric=metric)
for variant in [dict(opts), {**opts, "screen_dtype": "bf16"}, {**opts, "path": "scan"}, {k_: v for k_, v in opts.items() if k_ != "chunk0_rows"}]:
    with pkg.Mi355Index(d, metric) as idx:
        for key, val in variant.items():
            idx.set_option(key, val)
        if cut:
            idx.add(C[:cut])
        if cut < n:
            idx.add(C[cut:])
        dist, rows = idx.search(Q, k)
        stats = {s: idx.stat(s) for s in ("fallback_queries", "retry_queries", "loose_rows", "screen_dtype_active")}
    badq = sorted(set(np.argwhere(rows != rr)[:, 0].tolist()))
    print("variant", variant, "stats", stats, "bad queries", badq[:10], "of", B)
    for q in badq[:2]:
        print(" q", q, "gpu rows", rows[q][:8].tolist(), "dist", dist[q][:8].tolist())
        print(" q", q, "ref rows", rr[q][:8].tolist(), "dist", rd[q][:8].tolist())
        print("   |q|^2", float(np.dot(Q[q].astype(np.float64), Q[q].astype(np.float64))))
