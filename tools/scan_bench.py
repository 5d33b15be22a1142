"""k_scan alone: the guaranteed exact path on an N x 768 Gaussian corpus for 1 / 32 / 64 queries (path = scan).
Prints wall-clock per call; under `rocprofv3 --kernel-trace --stats` the k_scan row gives the kernel's own time."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import autorag_research_amd as pkg
from autorag_research_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, k = 768, 10
dev = torch.device("cuda", 0)
idx = pkg.Mi355Index(d, "cosine", device=0)
idx.reserve(n)
for c in range((n + synth.CHUNK_ROWS - 1) // synth.CHUNK_ROWS):
    x = synth.gaussian_chunk(torch, c, min(synth.CHUNK_ROWS, n - c * synth.CHUNK_ROWS), d, dev)
    torch.cuda.synchronize()
    idx.add_device(x.data_ptr(), x.shape[0])
    del x
idx.set_option("path", "scan")
g = torch.Generator(device=dev); g.manual_seed(1)
Q = torch.randn((64, d), generator=g, device=dev)
od = torch.empty((64, k), dtype=torch.float64, device=dev); orr = torch.empty((64, k), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for B in (1, 32, 64):
    for _ in range(2):
        idx.search_device(Q.data_ptr(), B, k, od.data_ptr(), orr.data_ptr(), s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        idx.search_device(Q.data_ptr(), B, k, od.data_ptr(), orr.data_ptr(), s)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / reps
    print(f"scan path N={n} B={B}: {t*1e3:.3f} ms per call, {n*d*4*((B+31)//32)/t/1e12:.2f} TB/s of fp32 rows (end to end)")
idx.close()
