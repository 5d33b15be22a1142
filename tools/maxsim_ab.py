#!/usr/bin/env python3
"""Interleaved A/B of the MaxSim pass on ONE box (boxes differ by +-3 %: never compare across them).

Builds the two stores of SURVEY 8(d) (100 k pages x 1030 tokens / 1 M docs x U{32..180} tokens, d = 128) on the device, then runs
`--steps` steps of `--queries` queries per configuration, the configurations back to back and `--rounds` times over.  A configuration
is a comma-separated list of library options (`mi355dr_set_option`; all of them only change the time, never a result); the
checksum of the returned ids must be the same on every line of a store.

    python tools/maxsim_ab.py maxsim_wg_pipe=1 maxsim_wg_pipe=0
    python tools/maxsim_ab.py --stores text --queries 8 maxsim_wg_min=8 maxsim_wg_min=9
    python tools/maxsim_ab.py "maxsim_tighten=1,maxsim_aligned=1" "maxsim_tighten=0,maxsim_aligned=0"

`profiles/r04_maxsim_ab.txt` is this tool's output over the round (one A/B per change).
"""
import argparse
import sys
import time

import numpy as np

DEFAULTS = {"maxsim_wg": -1, "maxsim_pass_groups": 4, "maxsim_wg_bps": 4, "maxsim_wg_pipe": 1, "maxsim_wg_min": 8,
            "maxsim_aligned": 1, "maxsim_tighten": 1, "maxsim_coop": -1, "maxsim_pack8": -1}


def build(pkg, torch, tokens: str, n_docs: int, d: int = 128):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(777)
    lens = rng.integers(32, 181, size=n_docs) if tokens == "text" else np.full((n_docs,), 1030, dtype=np.int64)
    idx = pkg.Mi355Index(d, "cosine", device=0)
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    per = max(1, (1 << 22) // int(lens.max()))
    for d0 in range(0, n_docs, per):
        ln = lens[d0:d0 + per]
        x = torch.randn((int(ln.sum()), d), generator=g, device=dev, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        torch.cuda.synchronize()
        idx.add_multivec_device(x.data_ptr(), np.concatenate([[0], np.cumsum(ln)]).astype(np.int64))
        del x
    return idx, lens, rng


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("configs", nargs="*", default=["maxsim_wg_pipe=1", "maxsim_wg_pipe=0"])
    ap.add_argument("--stores", default="page,text")
    ap.add_argument("--queries", type=int, default=16, help="queries per step (one pass up to 16)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--docs-scale", type=float, default=1.0, help="shrink the stores (quick checks)")
    args = ap.parse_args()
    sys.path.insert(0, ".")
    import torch

    import autorag_research_amd as pkg

    d, k = 128, 10
    for tokens in args.stores.split(","):
        n_docs = int((100_000 if tokens == "page" else 1_000_000) * args.docs_scale)
        nq = 24 if tokens == "page" else 32
        idx, lens, rng = build(pkg, torch, tokens, n_docs, d)
        blocks = float(((lens + 31) // 32).sum())
        qb = args.queries
        qtok = rng.standard_normal((qb * nq * (args.steps + 2), d), dtype=np.float32)
        qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
        qoff = (np.arange(qb + 1) * nq).astype(np.int32)
        for rnd in range(args.rounds):
            for cfg in args.configs:
                opts = dict(DEFAULTS)
                for kv in cfg.split(","):
                    key, val = kv.split("=")
                    opts[key.strip()] = int(val)
                for key, val in opts.items():
                    idx.set_option(key, val)
                for i in range(2):
                    idx.search_maxsim(qtok[i * qb * nq:(i + 1) * qb * nq], qoff, k)
                idx.reset_stats()
                idx.set_option("profile", 1)
                t0 = time.perf_counter()
                for i in range(args.steps):
                    res = idx.search_maxsim(qtok[(2 + i) * qb * nq:(3 + i) * qb * nq], qoff, k)
                el = time.perf_counter() - t0
                idx.set_option("profile", 0)
                n, ns = idx.stat("maxsim_screen_launches"), idx.stat("maxsim_screen_ns")
                scr = ns / max(n, 1) * 1e-9
                cols = idx.stat("maxsim_screen_cols") / max(n, 1)
                issued = 2.0 * cols * blocks * 32 * d / scr / 1e12 if scr > 0 else 0.0
                cand = idx.stat("maxsim_candidates") / max(idx.stat("maxsim_screened"), 1)
                pack_ms = idx.stat("maxsim_pack_ns") * 1e-6 / args.steps
                exact_ms = idx.stat("maxsim_exact_ns") * 1e-6 / args.steps
                print(f"{tokens:4s} {n_docs} docs, {qb} x {nq}-vector queries/step, round {rnd} [{cfg}]: {args.steps * qb / el:8.1f} queries/s  "
                      f"step {el / args.steps * 1e3:7.3f} ms  screen launch {scr * 1e3:7.3f} ms ({blocks * 8192 / scr / 1e12:5.2f} TB/s of the "
                      f"bf16 copy, issued {issued:6.0f} TF/s)  exact {exact_ms:5.3f} ms  host pack {pack_ms:5.3f} ms  {cand:5.0f} docs re-scored per query  "
                      f"checksum {int(res[1].sum())}", flush=True)
        idx.close()


if __name__ == "__main__":
    main()
