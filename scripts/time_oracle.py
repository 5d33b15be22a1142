"""Dev helper: time the CPU oracle (not used by tests)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import cpu_ref as o

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(0)
C = rng.standard_normal((n, 768), dtype=np.float32)
Q = rng.standard_normal((B, 768), dtype=np.float32)
for th in (1, 8, 8):
    t = time.time(); d, r = o.topk_search(C, Q, 10, threads=th); t1 = time.time() - t
    print("threads", th, "time", round(t1, 3), "GFMA/s", round(n * B * 768 / t1 / 1e9, 1), "QPS", round(B / t1, 1))
