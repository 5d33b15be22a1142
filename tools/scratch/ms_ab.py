"""A/B of the MaxSim screen forms on one box: groups per pass x (one wave per doc | workgroup-cooperative)."""
import json, os, subprocess, sys
sys.path.insert(0, ".")
import numpy as np, time, torch
import autorag_research_amd as pkg
def run(tokens, n_docs, nq, qblock, wg, groups, steps=20):
    d, k = 128, 10
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(777)
    lens = rng.integers(32, 181, size=n_docs) if tokens == "text" else np.full((n_docs,), 1030, dtype=np.int64)
    idx = pkg.Mi355Index(d, "cosine", device=0)
    g = torch.Generator(device=dev); g.manual_seed(777)
    per = max(1, (1 << 22) // int(lens.max()))
    for d0 in range(0, n_docs, per):
        ln = lens[d0:d0 + per]
        x = torch.randn((int(ln.sum()), d), generator=g, device=dev, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        torch.cuda.synchronize()
        idx.add_multivec_device(x.data_ptr(), np.concatenate([[0], np.cumsum(ln)]).astype(np.int64))
        del x
    qtok = rng.standard_normal((qblock * nq * (steps + 2), d), dtype=np.float32)
    qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
    qoff = (np.arange(qblock + 1) * nq).astype(np.int32)
    out = []
    for (w, gr) in [(wg_, g_) for wg_ in wg for g_ in groups]:
        idx.set_option("maxsim_wg", w); idx.set_option("maxsim_pass_groups", gr)
        for i in range(2): idx.search_maxsim(qtok[i * qblock * nq:(i + 1) * qblock * nq], qoff, k)
        idx.reset_stats(); idx.set_option("profile", 1)
        t0 = time.perf_counter()
        for i in range(steps): r = idx.search_maxsim(qtok[(2 + i) * qblock * nq:(3 + i) * qblock * nq], qoff, k)
        el = time.perf_counter() - t0
        idx.set_option("profile", 0)
        n, ns = idx.stat("maxsim_screen_launches"), idx.stat("maxsim_screen_ns")
        cols = idx.stat("maxsim_screen_cols") / max(n, 1)
        flops = 2.0 * cols * float(((lens + 31) // 32).sum()) * 32 * d
        print(f"{tokens} docs={n_docs} wg={w} groups={gr}: {steps*qblock/el:8.1f} q/s  step {el/steps*1e3:7.3f} ms  screen launch {ns/max(n,1)*1e-6:7.3f} ms x {n/steps:.1f}/step  issued {flops/(ns/max(n,1)*1e-9)/1e12:6.0f} TF/s  checksum {int(r[1].sum())}", flush=True)
    idx.close()
if __name__ == "__main__":
    run("page", 100_000, 24, 16, [1, 2, 1, 2], [4])
    run("text", 1_000_000, 32, 16, [1, 2, 1, 2, 1, 2], [4])
