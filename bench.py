#!/usr/bin/env python3
"""bench.py -- queries/sec of the Vector Search hot path on N MI355X (BASELINE.json metric).

Workload (config.workload): synthetic fp32 corpus, d=768, N=10M rows TOTAL, L2-normalised N(0,1) rows; a "step" =
one pass of the hot path over one block of 1024 queries per query group: exact cosine top-10 of every query against
the whole corpus (screen + exact re-score + select).  Queries and corpus are resident in HBM before the timed
region; outputs stay on the device.

N > 1 (--layout): the ranks form R row shards x Q query groups.  The default is 'rows' -- BASELINE.json's configuration 3 as
named: the corpus row-sharded over ALL ranks, every rank the same query block against its 1/N of the rows, ONE packed RCCL
all-gather of the per-shard top-k + k_merge_topk per step inside the timed loop (on a second stream, under the next step's
search): "scaling": "strong".  'auto' cuts the corpus into only as many row shards as it needs to fit (10 M rows x 5.4 KB =
54 GB of one GPU's 288 GB: R = 1: replicas serving their own query blocks, no data-path collective, weak scaling);
'RxQ' mixes the two (the exchange stays inside a group of R ranks).  `--replicated-leg` measures the 1 x N layout after the
main run and reports it under extra.replicated.  `value` = queries all ranks answered / max-over-ranks time in every layout.

Steps are software-pipelined through mi355dr_search_device_async / mi355dr_search_wait: block i + 1 is on the stream before
the host waits for block i (every step's work, its wait included, lies inside the timed region).

Launch: `python bench.py --gpus 1` or, for N > 1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
Prints ONE JSON line on rank 0.  torch is used for plumbing only (synthetic data, device buffers,
torch.distributed); every timed kernel is libmi355dr's hand-written HIP.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

from autorag_research_amd import synth  # noqa: E402
from autorag_research_amd.synth import CHUNK_ROWS  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak
MFMA_I8_PEAK_TOPS = 5000.0  # int8 MFMA = 2x the bf16 rate on gfx950 (2xK): the dense peak the fraction is quoted against
# cdna_hip_programming.md "MFMA ubench throughput": i8 32x32x32 (the instruction this kernel issues) 4404 TOPS,
# i8 16x16x64 3944 TOPS (the figure MI355X_MICROARCH.md's MFMA table carries); bf16 32x32x16 2382 TF
MFMA_I8_UBENCH_TOPS = {"32x32x32": 4404.0, "16x16x64": 3944.0}
MFMA_BF16_UBENCH_TF = 2382.0
# What the matrix pipe delivers ON GAUSSIAN OPERANDS with no memory traffic at all: a bare MFMA stream (operands in registers, four
# accumulators, two waves per SIMD on every CU) settles at the socket power cap -- 1.28 kW, 1.79 / 1.80 GHz -- far below the
# nominal peaks, which only zero operands reach (tools/mfma_power_probe.hip, profiles/r04_fp4_probe.txt).  The nominal dense peak
# stays the denominator of `roofline.frac`; `frac_of_power_limited_stream` is the same achieved rate over these.
MFMA_I8_POWER_LIMITED_TOPS = 3369.0
MFMA_BF16_POWER_LIMITED_TF = 1743.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000, help="TOTAL corpus rows (sharded over ranks)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--block", type=int, default=1024, help="queries per step")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--workload", choices=["single", "maxsim"], default="single",
                    help="single = headline cosine top-k (default); maxsim = multi-vector late interaction (SURVEY 8a row a2)")
    ap.add_argument("--docs", type=int, default=100_000, help="maxsim: documents (d=128)")
    ap.add_argument("--maxsim-queries", type=int, default=16, help="maxsim: queries per step (one screen pass serves up to 16)")
    ap.add_argument("--tokens", choices=["text", "page"], default="text",
                    help="maxsim: 'text' = U{32..180} vectors per doc, 32-vector queries (ColBERT-like); 'page' = 1030 patch "
                         "vectors per doc, 24-vector queries (ColPali-like)")
    ap.add_argument("--chunk0", type=int, default=0, help="override the first (emit-all) chunk size")
    ap.add_argument("--growth", type=int, default=0, help="override the chunk growth factor")
    ap.add_argument("--screen", choices=["auto", "bf16", "i8"], default="auto", help="screen element type")
    ap.add_argument("--round-a", type=int, default=None, help="k_prune: rows re-scored before the cut is known (tuning)")
    ap.add_argument("--prefilter16", type=int, default=None, help="0/1: bf16 second screen inside the prune (default: library default)")
    ap.add_argument("--metric", choices=["cosine", "ip"], default="cosine", help="cosine (headline) or inner product")
    ap.add_argument("--small-chunk", type=int, default=None, help="override small_chunk_rows (developer sweep)")
    ap.add_argument("--starter", type=int, default=None, help="0/1: pass schedule with / without the sampled threshold estimator (A/B)")
    ap.add_argument("--defer-b", type=int, default=None, help="0/1: prunes before the last carry their survivors over instead of re-scoring them (A/B)")
    ap.add_argument("--prune-companion", type=int, default=None, help="0/1: general-form prune launch behind every one-wave prune (A/B)")
    ap.add_argument("--layout", default="rows",
                    help="ranks as (row shards R) x (query groups Q): 'rows' (default) = world x 1: the corpus row-sharded over "
                         "all ranks, every rank the same block, RCCL all-gather + merge per step (BASELINE config 3); 'auto' = "
                         "fewest row shards whose shard fits in 60 %% of one GPU's HBM (N=10M, d=768 -> 1 x world: replicas, "
                         "no data-path collective); 'queries' = 1 x world; or 'RxQ'")
    ap.add_argument("--replicated-leg", action="store_true",
                    help="N > 1: also measure the replicated layout (1 x world, every rank the whole corpus and its own query "
                         "block, no collective) after the main run and report it under extra.replicated")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking mi355dr_search_device per step (no async pipelining; A/B)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the all-gather + merge path even at world size 1 (exercises the multi-GPU code on one GPU)")
    ap.add_argument("--comm", choices=["torch", "lib"], default="torch",
                    help="N > 1: all-gather through torch.distributed (RCCL, overlapped with the next step's search on a second "
                         "stream) or through the library's own RCCL communicator (mi355dr_search_sharded_device)")
    ap.add_argument("--row-sharded-leg", action="store_true",
                    help="also measure the fully row-sharded layout (world x 1) after the main run and report it under "
                         "extra.row_sharded (only meaningful when the main layout is not already that one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (planted-answer nDCG, PCIe-inclusive "
                                                           "rate, BLAS / torch / B=1 CPU baselines)")
    ap.add_argument("--data", choices=["gaussian", "anisotropic"], default="gaussian",
                    help="gaussian = BASELINE headline; anisotropic = power-law spectrum + near-duplicate clusters (the offline "
                         "stand-in for bge-base on BEIR nq, config C2: use with --metric ip --k 100)")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_500_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=3072, help="CPU baseline: queries timed (whole blocks of the pool)")
    return ap.parse_args()


def cpu_shape_baselines(Cs: np.ndarray, Qs: np.ndarray, k: int, metric: str, n_total: int, exact_rows_fn) -> list:
    """SURVEY.md 8(d) / BASELINE.md 2: the reference's engine (PostgreSQL) cannot run here, so beside the exact-chain
    oracle the same math is timed the way a numpy / torch CPU VectorSearch would issue it -- BLAS `Q_block @ C^T` in row
    chunks + argpartition, torch.mm + topk -- and in the reference's CALL SHAPE, one query at a time (B = 1).  Bounded
    samples (seconds each), scaled linearly to N; ids compared with the exact chain's on the same sample (BLAS sums in another
    order than the exact chain, so near-ties may legitimately swap: the agreement is reported, not asserted)."""
    import torch

    out = []
    S = min(Cs.shape[0], 500_000)
    C = np.ascontiguousarray(Cs[:S])
    Q = np.ascontiguousarray(Qs[:1024])
    inv = 1.0 / np.linalg.norm(C, axis=1) if metric == "cosine" else None
    # BLAS and torch size their pools from the logical CPUs they see (256 on the GPU boxes); the cgroup grants a 16-CPU quota:
    # both are held to the quota the oracle uses, so that `cores` is what the figure was measured on
    from oracle import cpu_ref as _cr

    quota = max(1, int(_cr.num_threads()))
    torch.set_num_threads(quota)
    blas_limit = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        blas_limit = threadpool_limits(limits=quota, user_api="blas")
        blas_threads = max([t.get("num_threads", 1) for t in threadpool_info() if t.get("user_api") == "blas"] or [1])
    except Exception:  # noqa: BLE001
        blas_threads = None

    def np_block(Qb):
        best = None
        for r0 in range(0, S, 100_000):
            sc = Qb @ C[r0:r0 + 100_000].T
            if inv is not None:
                sc *= inv[None, r0:r0 + 100_000]
            part = np.argpartition(-sc, min(k, sc.shape[1] - 1), axis=1)[:, :k]
            v = np.take_along_axis(sc, part, axis=1)
            cand = (v, part + r0)
            best = cand if best is None else (np.concatenate([best[0], cand[0]], 1), np.concatenate([best[1], cand[1]], 1))
        o = np.argsort(-best[0], axis=1, kind="stable")[:, :k]
        return np.take_along_axis(best[1], o, axis=1)

    np_block(Q[:8])
    t = time.perf_counter()
    rows_np = np_block(Q)
    t = time.perf_counter() - t
    agree = float(np.mean(rows_np == exact_rows_fn(C, Q)))
    out.append({"kind": "numpy-blas", "value": round(len(Q) / t * S / n_total, 3), "unit": "queries/s", "cores": blas_threads,
                "sample": f"fp32 Q_block[{len(Q)}] @ C[{S}]^T in 100k-row chunks + argpartition/argsort, {t:.2f} s, scaled to "
                          f"N={n_total}", "ids_equal_to_exact_chain": round(agree, 6)})
    Ct, Qt = torch.from_numpy(C), torch.from_numpy(Q)
    it = torch.from_numpy(inv) if inv is not None else None

    def torch_block(Qb):
        sc = Qb @ Ct.T
        if it is not None:
            sc *= it[None, :]
        return torch.topk(sc, k, dim=1).indices

    torch_block(Qt[:8])
    nqt = min(256, len(Q))
    t = time.perf_counter()
    torch_block(Qt[:nqt])
    t = time.perf_counter() - t
    out.append({"kind": "torch-cpu", "value": round(nqt / t * S / n_total, 3), "unit": "queries/s",
                "cores": torch.get_num_threads(),
                "sample": f"torch.mm(Q[{nqt}], C[{S}]^T) + torch.topk, {t:.2f} s, scaled to N={n_total}"})
    t = time.perf_counter()
    for i in range(16):
        torch_block(Qt[i:i + 1])
    t = time.perf_counter() - t
    out.append({"kind": "torch-cpu, B=1 call shape", "value": round(16 / t * S / n_total, 3), "unit": "queries/s",
                "cores": torch.get_num_threads(),
                "sample": f"the same, ONE query per call (how the reference's pipeline calls its engine: "
                          f"pipelines/retrieval/vector_search.py:157-169), 16 calls over C[{S}], {t:.2f} s, scaled to N={n_total}"})
    if blas_limit is not None:
        blas_limit.restore_original_limits()
    return out


def run_maxsim(args, n_docs: int, tokens: str, nq: int, steps: int, warmup: int, cpu_sample_docs: int = 0, probe: bool = False) -> dict:
    """MaxSim top-k (VectorChord `@#`) on a synthetic multi-vector store built ON THE DEVICE (token vectors generated in
    HBM, handed to the index by pointer: mi355dr_add_multivec_device).  `tokens` = "text" (ColBERT-like: U{32..180} vectors
    per doc) or "page" (ColPali-like: 1030 patch vectors per doc); d = 128, unit-norm vectors, seed 777.
    A step = one call with 16 queries x `nq` query vectors against every document: ONE bf16 MFMA screen pass over the bf16
    fragment copy serves all four groups of 4 (round 4; 8 queries in rounds 2-3), then per group the selection and the exact
    fp32 MFMA kernel on the candidates; wall clock includes H2D of the queries and D2H of [16,k]."""
    import torch

    import autorag_research_amd as pkg

    d, qblock, k = 128, getattr(args, "maxsim_queries", 16), args.k
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    rng = np.random.default_rng(777)
    lens = rng.integers(32, 181, size=n_docs) if tokens == "text" else np.full((n_docs,), 1030, dtype=np.int64)
    idx = pkg.Mi355Index(d, "cosine", device=dev.index)
    if os.environ.get("MI355DR_MAXSIM_PERSISTENT") is not None:   # developer A/B
        idx.set_option("maxsim_persistent", int(os.environ["MI355DR_MAXSIM_PERSISTENT"]))
    g = torch.Generator(device=dev)
    g.manual_seed(777)
    t_build = time.perf_counter()
    docs_per_chunk = max(1, (1 << 22) // int(lens.max()))
    keep = None
    for d0 in range(0, n_docs, docs_per_chunk):
        ln = lens[d0:d0 + docs_per_chunk]
        x = torch.randn((int(ln.sum()), d), generator=g, device=dev, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        torch.cuda.synchronize()
        idx.add_multivec_device(x.data_ptr(), np.concatenate([[0], np.cumsum(ln)]).astype(np.int64))
        if keep is None and cpu_sample_docs:
            S = min(cpu_sample_docs, len(ln))
            keep = (x[: int(ln[:S].sum())].cpu().numpy(), np.concatenate([[0], np.cumsum(ln[:S])]).astype(np.int64))
        del x
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    n_q = qblock * (steps + warmup)
    qtok = rng.standard_normal((n_q * nq, d), dtype=np.float32)
    qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
    qoff = (np.arange(qblock + 1) * nq).astype(np.int32)

    def step(i):
        return idx.search_maxsim(qtok[i * qblock * nq:(i + 1) * qblock * nq], qoff, k)

    for i in range(warmup):
        step(i)
    idx.reset_stats()
    idx.set_option("profile", 1)   # HIP events around the screen launch (k_maxsim16*) and the exact launch on the candidates
    t0 = time.perf_counter()
    for i in range(steps):
        res = step(warmup + i)
    el = time.perf_counter() - t0
    idx.set_option("profile", 0)
    assert (np.diff(res[0], axis=1) >= 0).all()
    probe_out = None
    if probe:  # socket power / shader clock next to ~2 s of the same steps (not timed)
        try:
            probe_out = power_probe(lambda n: [step(i % (warmup + steps)) for i in range(n)], el / steps)
        except Exception as e:  # noqa: BLE001
            probe_out = {"error": f"{type(e).__name__}: {e}"}
    blocks = int(((lens + 31) // 32).sum())
    n_tok = float(lens.sum())
    alg_bytes = n_tok * d * 4                                    # fp32 token rows read once per pass
    streamed = float(blocks) * 32 * d * 2                        # bf16 fragment store the screen streams, per pass
    screened, cands, fb = idx.stat("maxsim_screened"), idx.stat("maxsim_candidates"), idx.stat("maxsim_fallbacks")
    scr_n, scr_ns = idx.stat("maxsim_screen_launches"), idx.stat("maxsim_screen_ns")
    ex_n, ex_ns = idx.stat("maxsim_exact_launches"), idx.stat("maxsim_exact_ns")
    cols_issued = idx.stat("maxsim_screen_cols") / max(scr_n, 1)  # query columns per launch, whole blocks of 32
    scr_s = scr_ns * 1e-9 / max(scr_n, 1)                        # average duration of one screen launch (= one pass)
    # MaxSim is compute-bound from one 32-vector query up -- "the MFMA roofline is the honest one here".
    # Algorithmic flops of a pass = 2 * (query vectors of the pass) * (doc vectors) * d; the kernel issues the same on whole
    # 32-row doc blocks and whole 32-column query blocks.
    alg_flops = 2.0 * (qblock * nq) * n_tok * d
    issued_flops = 2.0 * cols_issued * blocks * 32 * d
    out = {
        "workload": f"MaxSim top-{k}: {n_docs} docs, {int(lens.sum())} doc vectors ({'U{32..180}' if tokens == 'text' else '1030'}"
                    f"/doc), d=128, {qblock} queries x {nq} vectors per step, {steps * qblock} queries timed; store built on the "
                    f"device in {t_build:.2f} s",
        "queries_per_s": round(steps * qblock / el, 2), "ms_per_step": round(el * 1e3 / steps, 3), "steps": steps,
        "queries_per_pass": qblock,
        "includes": "H2D of the query block, D2H of results",
        "roofline": {"bound": "mfma", "kernel": f"k_maxsim16_d128<{(qblock * nq + 31) // 32}>",
                     "op": "bf16 flops (v_mfma_f32_32x32x16_bf16)",
                     "achieved": round(alg_flops / scr_s / 1e12, 2) if scr_n else None,
                     "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(alg_flops / scr_s / 1e12 / MFMA_BF16_PEAK_TF, 4) if scr_n else None,
                     "issued_tflops": round(issued_flops / scr_s / 1e12, 2) if scr_n else None,
                     "frac_of_power_limited_stream": round(issued_flops / scr_s / 1e12 / MFMA_BF16_POWER_LIMITED_TF, 4) if scr_n else None,
                     "traffic": None,
                     "traffic_unit": f"HBM read bytes per launch, vs {round(streamed)} streamed (the bf16 fragment copy) and "
                                     f"algorithmic {round(alg_bytes)} (fp32 token rows, SURVEY 8d)", "traffic_source": None,
                     "launches": scr_n, "avg_launch_ms": round(scr_s * 1e3, 4),
                     "note": "algorithmic 2 * query vectors * doc vectors * d per pass over the screen kernel's average launch "
                             f"(HIP events on the launch stream, library option `profile`); one launch screens the {qblock} "
                             "queries of a step",
                     "hbm_view": {"streamed_GBps": round(streamed / scr_s / 1e9, 1) if scr_n else None,
                                  "streamed_frac": round(streamed / scr_s / 1e9 / HBM_PEAK_GBS, 4) if scr_n else None,
                                  "algorithmic_GBps": round(alg_bytes / scr_s / 1e9, 1) if scr_n else None,
                                  "note": "the pass as a stream: what the kernel reads (bf16 copy) and SURVEY 8(d)'s fp32 bytes "
                                          "over the same launch; NOT the binding roof at 16 queries per pass"},
                     "exact_rescore_ms_per_step": round(ex_ns * 1e-6 / max(steps, 1), 4), "exact_launches": ex_n,
                     "wall_clock_tflops": round(alg_flops * steps / el / 1e12, 2)},
        "queries_screened": screened, "candidates_per_query": round(cands / max(screened, 1), 1),
        "exact_full_scan_fallbacks": fb,
    }
    if keep is not None:
        from oracle import cpu_ref

        tok_s, off_s = keep
        S = off_s.shape[0] - 1
        tc = time.perf_counter()
        rd, rr = cpu_ref.maxsim_topk(tok_s, off_s, qtok[: qblock * nq], qoff, k)
        tc = time.perf_counter() - tc
        with pkg.Mi355Index(d, "cosine", device=dev.index) as s2:
            s2.add_multivec(tok_s, off_s)
            gd, gr = s2.search_maxsim(qtok[: qblock * nq], qoff, k)
        out["cpu_baseline"] = {"value": round(qblock / tc * S / n_docs, 4), "unit": "queries/s", "cores": cpu_ref.num_threads(),
                               "kind": "port", "sample": f"oracle MaxSim on the first {S} docs x {qblock} queries, scaled "
                                                         f"linearly to {n_docs} docs; {tc:.1f} s of CPU work",
                               "parity_on_sample": bool(np.array_equal(gr, rr) and np.array_equal(gd, rd))}
    if probe_out is not None:
        out["power_probe"] = probe_out
    idx.close()
    return out


def power_probe(run_n_steps, s_per_step: float, seconds: float = 2.0) -> dict:
    """Socket power and shader clock WHILE the steps run: `rocm-smi --showpower --showclocks` polled from a thread next to
    ~`seconds` of back-to-back steps.  Medians over the samples taken after the first 0.4 s (the governor's ramp)."""
    import re
    import shutil
    import statistics
    import subprocess
    import threading

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return {"error": "rocm-smi not found"}
    samples, stop = [], threading.Event()

    def poll():
        t0 = time.perf_counter()
        while not stop.is_set():
            try:
                out = subprocess.run([exe, "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True,
                                     timeout=5).stdout
            except Exception:  # noqa: BLE001
                break
            pw = re.search(r"GPU\[0\].*Socket Graphics Package Power \(W\): ([0-9.]+)", out)
            ck = re.search(r"GPU\[0\].*sclk clock level: \S+ \((\d+)Mhz\)", out)
            cap = re.search(r"GPU\[0\].*Max Graphics Package Power \(W\): ([0-9.]+)", out)
            if pw and ck:
                samples.append((time.perf_counter() - t0, float(pw.group(1)), int(ck.group(1)), float(cap.group(1)) if cap else None))

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    t0 = time.perf_counter()
    n = max(20, int(seconds / max(s_per_step, 1e-4)))
    run_n_steps(n)
    dt = time.perf_counter() - t0
    stop.set()
    th.join(timeout=6)
    used = [x for x in samples if 0.4 <= x[0] <= dt] or samples
    if not used:
        return {"error": "no rocm-smi sample landed inside the burst", "burst_s": round(dt, 4), "burst_steps": n}
    return {"socket_power_W_median": statistics.median(x[1] for x in used), "sclk_MHz_median": statistics.median(x[2] for x in used),
            "power_cap_W": next((x[3] for x in used if x[3]), None), "sclk_max_MHz": 2400, "samples": len(used),
            "burst": f"{n} steps in {dt:.2f} s ({dt / n * 1e3:.3f} ms per step)", "burst_s": round(dt, 4), "burst_steps": n,
            "note": "rocm-smi polled next to back-to-back steps of this workload; NOT part of the timed region"}


def pmc_fetch_subrun(bench_args: list, kernel_substr: str, timeout_s: int = 240):
    """HBM read bytes per launch of the dominant kernel, MEASURED in this run: the same workload re-run for 3 steps under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (counters in a pass of their own, kernel trace only: MI355X_MICROARCH.md's HBM
    recipe; FETCH_SIZE is in KiB and counts the 128-B requests of a wide stream as 64 B on gfx950: x 1024 x 2).
    Returns (bytes per launch, launches profiled, note) or (None, 0, why not).  Never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("MI355DR_BENCH_PMC", "1") == "0":
        return None, 0, "disabled (MI355DR_BENCH_PMC=0)"
    if any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ):
        return None, 0, "this process is itself running under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, 0, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="mi355dr_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp", MI355DR_BENCH_PMC="0")
        for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(var, None)
        cmd = [exe, "--kernel-trace", "--pmc", "FETCH_SIZE", "-f", "csv", "-d", out, "-o", "fetch", "--", sys.executable,
               str(ROOT / "bench.py"), *[str(a) for a in bench_args], "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
               "--no-extras"]
        p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        vals = []
        for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                        vals.append(float(r["Counter_Value"]))
        if not vals:
            return None, 0, f"no {kernel_substr} launches in the counter file (rc {p.returncode}): {p.stderr[-200:]!r}"
        return sum(vals) / len(vals) * 1024 * 2, len(vals), "ok"
    except Exception as e:  # noqa: BLE001 - a secondary figure must not take the line down
        return None, 0, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def maxsim_traffic(roof: dict, tokens: str, docs: int) -> None:
    """Fill roofline.traffic of a MaxSim leg from a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` sub-run of the same workload."""
    per_launch, n_prof, note = pmc_fetch_subrun(["--workload", "maxsim", "--tokens", tokens, "--docs", docs], "k_maxsim16",
                                                timeout_s=300)
    if per_launch is not None:
        roof["traffic"] = round(per_launch)
        roof["traffic_source"] = (
            f"MEASURED in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE sub-run of this workload (3 steps, {n_prof} screen "
            "launches; KiB x 1024 x 2: the gfx950 correction of MI355X_MICROARCH.md), mean per launch")
    else:
        roof["traffic_source"] = f"not measured [live PMC sub-run: {note}]"


def main_maxsim(args) -> None:
    """Secondary workload as the whole bench line: `python bench.py --workload maxsim [--docs N] [--tokens text|page]`."""
    r = run_maxsim(args, args.docs, args.tokens, 32 if args.tokens == "text" else 24, args.steps, args.warmup,
                   0 if args.no_cpu_baseline else 2000, probe=not args.no_extras)
    out = {"metric": "queries/sec", "value": r["queries_per_s"], "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": r["workload"], "includes": r["includes"]}, "roofline": r["roofline"],
           "extra": {kk: r[kk] for kk in ("queries_screened", "candidates_per_query", "exact_full_scan_fallbacks", "power_probe")
                     if kk in r}}
    if "cpu_baseline" in r:
        out["cpu_baseline"] = r["cpu_baseline"]
    if not args.no_extras:
        maxsim_traffic(out["roofline"], args.tokens, args.docs)
    print(json.dumps(out))


def row_sharded_leg(args, torch, pkg, dist, device, local_rank, rank, world, qpool, ref_block, make_chunk) -> dict:
    """BASELINE.json's config C3 as named -- the corpus row-sharded over ALL ranks, every rank answering the SAME query block
    against its 1/world of the rows, one packed all-gather + k_merge_topk per step (overlapped with the next step's search)
    -- measured after the main run on a second, shard-sized index, and checked against the main layout's answer."""
    n_total, d, B, k = args.rows, args.dim, args.block, args.k
    n_chunks = (n_total + CHUNK_ROWS - 1) // CHUNK_ROWS
    c_lo, c_hi = n_chunks * rank // world, n_chunks * (rank + 1) // world
    row_lo = min(n_total, c_lo * CHUNK_ROWS)
    n_local = min(n_total, c_hi * CHUNK_ROWS) - row_lo
    idx = pkg.Mi355Index(d, args.metric, device=local_rank)
    idx.reserve(n_local)
    idx.set_option("row_offset", row_lo)
    idx.set_option("screen_dtype", args.screen)
    for c in range(c_lo, c_hi):
        x = make_chunk(c, min(CHUNK_ROWS, n_total - c * CHUNK_ROWS))
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), x.shape[0])
        del x
    torch.cuda.synchronize()
    n_pool = qpool.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    comm_stream = torch.cuda.Stream(device)
    packed2 = [torch.empty((2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
    all2 = [torch.empty((world, 2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
    fin2 = [torch.empty((2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
    done = [None, None]

    def step(i: int):
        buf = i & 1
        if done[buf] is not None:
            torch.cuda.current_stream().wait_event(done[buf])
        pk = packed2[buf]
        idx.search_device(qpool[i % n_pool].data_ptr(), B, k, pk[0].data_ptr(), pk[1].data_ptr(), stream)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            dist.all_gather_into_tensor(all2[buf].view(-1), pk.view(-1))
            idx.merge_topk_packed_device(all2[buf].data_ptr(), world, B, k, fin2[buf][0].data_ptr(), fin2[buf][1].data_ptr(),
                                         comm_stream.cuda_stream)
            done[buf] = torch.cuda.Event()
            done[buf].record(comm_stream)
        return fin2[buf]

    out0 = step(0)
    torch.cuda.synchronize()
    identical = bool(torch.equal(out0, ref_block)) if ref_block is not None else None
    steps = max(4, min(args.steps, 20))
    step(1)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(2 + i)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    agree = torch.tensor([1.0 if identical in (True, None) else 0.0], device=device, dtype=torch.float64)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    idx.close()
    return {
        "layout": f"{world} row shards x 1 query group (every rank: the same {B}-query block against {n_local} of the "
                  f"{n_total} rows; one packed all-gather + k_merge_topk per step on a second stream)",
        "scaling": "strong",
        "queries_per_s": round(steps * B / elapsed, 1),
        "ms_per_step": round(elapsed * 1e3 / steps, 3),
        "steps": steps,
        "identical_to_main_layout": bool(agree.item() == 1.0) if ref_block is not None else None,
        "note": "NOT `value`: the default layout answers independent query blocks on replicas when the corpus fits one GPU "
                "(DESIGN.md section 5)",
    }


def replicated_leg(args, torch, pkg, dist, device, local_rank, rank, world, qpool, make_chunk) -> dict:
    """The 1 x world layout next to the row-sharded `value`: every rank holds the WHOLE corpus (288 GB of HBM take ~32 M rows of
    d = 768 with both screen copies) and answers its OWN query block per step -- independent units, no data-path collective,
    weak scaling in queries."""
    n_total, d, B, k = args.rows, args.dim, args.block, args.k
    idx = pkg.Mi355Index(d, args.metric, device=local_rank)
    idx.reserve(n_total)
    idx.set_option("screen_dtype", args.screen)
    for c in range((n_total + CHUNK_ROWS - 1) // CHUNK_ROWS):
        x = make_chunk(c, min(CHUNK_ROWS, n_total - c * CHUNK_ROWS))
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), x.shape[0])
        del x
    torch.cuda.synchronize()
    n_pool = qpool.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    out2 = [torch.empty((2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
    steps = max(4, min(args.steps, 20))

    def run(first, count):
        pend = None
        for i in range(first, first + count):
            o = out2[i & 1]
            t = idx.search_device_async(qpool[(i * world + rank) % n_pool].data_ptr(), B, k, o[0].data_ptr(), o[1].data_ptr(), stream)
            if pend is not None:
                idx.search_wait(pend)
            pend = t
        idx.search_wait(pend)

    run(0, 2)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(2, steps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    idx.close()
    return {"layout": f"1 row shard x {world} query groups (replicas: every rank the whole corpus and its own {B}-query block, "
                      "no data-path collective)", "scaling": "weak",
            "queries_per_s": round(steps * B * world / float(tmax.item()), 1),
            "ms_per_step": round(float(tmax.item()) * 1e3 / steps, 3), "steps": steps,
            "note": "NOT `value`: BASELINE.json's configuration 3 is the row-sharded layout"}


def block_size_table(idx, torch, qpool, d: int, k: int, n_rows: int, device) -> list:
    """the same corpus pass with 1, 32 and 128 queries per call -- the HBM-bound regime (the pass is one stream over
    the int8 shadow + the re-score launches; k_screen_stream / k_screen).  Device buffers, blocking calls."""
    out = []
    stream = torch.cuda.current_stream().cuda_stream
    flat = qpool.reshape(-1, d)
    od = torch.empty((128, k), device=device, dtype=torch.float64)
    orr = torch.empty((128, k), device=device, dtype=torch.int64)
    for b in (1, 32, 128):
        for i in range(3):
            idx.search_device(flat[i * b:(i + 1) * b].data_ptr(), b, k, od.data_ptr(), orr.data_ptr(), stream)
        torch.cuda.synchronize()
        n = 12
        t = time.perf_counter()
        for i in range(n):
            idx.search_device(flat[(3 + i) * b:(4 + i) * b].data_ptr(), b, k, od.data_ptr(), orr.data_ptr(), stream)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t) / n
        out.append({"queries_per_call": b, "ms_per_call": round(t * 1e3, 3), "queries_per_s": round(b / t, 1),
                    "algorithmic_GBps": round(n_rows * d * 4 / t / 1e9, 1)})
    return out


def other_k_line(idx, torch, qpool, B: int, k2: int, n_rows: int, device) -> dict:
    """BASELINE.json quotes k in {10, 100}: the same corpus pass at the other k as a secondary figure (device buffers,
    steps pipelined like the timed region)."""
    stream = torch.cuda.current_stream().cuda_stream
    out2 = [torch.empty((2, B, k2), device=device, dtype=torch.int64) for _ in range(2)]
    n_pool = qpool.shape[0]

    def run(first, count):
        pend = None
        for i in range(first, first + count):
            o = out2[i & 1]
            t = idx.search_device_async(qpool[i % n_pool].data_ptr(), B, k2, o[0].data_ptr(), o[1].data_ptr(), stream)
            if pend is not None:
                idx.search_wait(pend)
            pend = t
        idx.search_wait(pend)

    run(0, 3)
    torch.cuda.synchronize()
    idx.reset_stats()
    n = 10
    t = time.perf_counter()
    run(3, n)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / n
    return {"k": k2, "ms_per_step": round(t * 1e3, 3), "queries_per_s": round(B / t, 1),
            "screen": "int8" if idx.stat("screen_dtype_active") == 2 else "bf16", "retry_queries": idx.stat("retry_queries"),
            "fallback_queries": idx.stat("fallback_queries"), "rows": n_rows}


def small_corpus_line(args, torch, pkg, qpool, device, local_rank, n_rows: int) -> dict:
    """SURVEY 8(d)'s second corpus size (N = 1 M) as a secondary figure of the same run: same generator, same 1024-query blocks."""
    d, B, k = args.dim, args.block, args.k
    idx = pkg.Mi355Index(d, args.metric, device=local_rank)
    idx.reserve(n_rows)
    idx.set_option("screen_dtype", args.screen)
    for c in range((n_rows + CHUNK_ROWS - 1) // CHUNK_ROWS):
        x = synth.gaussian_chunk(torch, c, min(CHUNK_ROWS, n_rows - c * CHUNK_ROWS), d, device)
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), x.shape[0])
        del x
    stream = torch.cuda.current_stream().cuda_stream
    out2 = [torch.empty((2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
    n_pool = qpool.shape[0]

    def run(first, count):
        pend = None
        for i in range(first, first + count):
            o = out2[i & 1]
            t = idx.search_device_async(qpool[i % n_pool].data_ptr(), B, k, o[0].data_ptr(), o[1].data_ptr(), stream)
            if pend is not None:
                idx.search_wait(pend)
            pend = t
        idx.search_wait(pend)

    run(0, 3)
    torch.cuda.synchronize()
    steps = 30
    t = time.perf_counter()
    run(3, steps)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / steps
    idx.close()
    return {"rows": n_rows, "ms_per_step": round(t * 1e3, 3), "queries_per_s": round(B / t, 1),
            "algorithmic_GBps": round(n_rows * d * 4 / t / 1e9, 1)}


def main() -> None:
    args = parse_args()
    if args.workload == "maxsim":
        return main_maxsim(args)
    import torch

    import autorag_research_amd as pkg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the search path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    from autorag_research_amd.sharded import GridLayout

    n_total, d, B, k = args.rows, args.dim, args.block, args.k
    layout = GridLayout.parse(args.layout, world, rank, n_total, d, torch.cuda.mem_get_info(device)[1])
    R, QG = layout.row_shards, layout.query_groups
    have_pg = world > 1 or args.force_dist          # a process group exists (timing barrier, max over ranks)
    use_dist = R > 1 or args.force_dist             # the search itself has an exchange step (row shards)
    row_group = None
    if have_pg:
        import torch.distributed as dist  # noqa: PLC0415

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)
        row_group = layout.make_row_group(dist)      # None: the whole world (or a single rank)

    n_chunks = (n_total + CHUNK_ROWS - 1) // CHUNK_ROWS
    # contiguous chunk range per row shard
    c_lo = n_chunks * layout.shard // R
    c_hi = n_chunks * (layout.shard + 1) // R
    row_lo = min(n_total, c_lo * CHUNK_ROWS)
    row_hi = min(n_total, c_hi * CHUNK_ROWS)
    n_local = row_hi - row_lo

    idx = pkg.Mi355Index(d, args.metric, device=local_rank)
    idx.reserve(n_local)
    idx.set_option("row_offset", row_lo)
    if args.chunk0:
        idx.set_option("chunk0_rows", args.chunk0)
    if args.growth:
        idx.set_option("chunk_growth", args.growth)
    idx.set_option("screen_dtype", args.screen)
    if args.prefilter16 is not None:
        idx.set_option("prefilter16", args.prefilter16)
    if args.round_a is not None:
        idx.set_option("round_a", args.round_a)
    if args.small_chunk is not None:
        idx.set_option("small_chunk_rows", args.small_chunk)
    if args.starter is not None:
        idx.set_option("starter", args.starter)
    if args.prune_companion is not None:
        idx.set_option("prune_companion", args.prune_companion)
    if args.defer_b is not None:
        idx.set_option("defer_round_b", args.defer_b)
    aniso = synth.Anisotropic(torch, d, device) if args.data == "anisotropic" else None

    def gen_chunk(c: int, rows: int):
        return aniso.chunk(c, rows) if aniso is not None else synth.gaussian_chunk(torch, c, rows, d, device)

    # query pool in HBM: 10 blocks, cycled by the timed steps -- plus ONE extra block that is never timed: the
    # planted-answer block of SURVEY.md 8(d) (1-3 relevant rows per query written into the corpus at recorded positions;
    # ~2k rows of 10 M, invisible to the timed queries), from which nDCG@10 is computed after the timed region
    n_pool = 10
    if aniso is not None:
        qpool = aniso.queries(n_pool * B).reshape(n_pool, B, d)
        q_plant = aniso.queries(B, seed=555)
    else:
        gq = torch.Generator(device=device)
        gq.manual_seed(4321)
        qpool = torch.randn((n_pool, B, d), generator=gq, device=device, dtype=torch.float32)
        qpool /= qpool.norm(dim=2, keepdim=True)
        gq.manual_seed(555)
        q_plant = torch.randn((B, d), generator=gq, device=device, dtype=torch.float32)
        q_plant /= q_plant.norm(dim=1, keepdim=True)
    plant_window = min(n_total, args.cpu_sample_rows)
    p_pos, p_vec, p_owner, p_sigma = synth.planted_answers(torch, q_plant, plant_window)
    p_pos_t = torch.as_tensor(p_pos, device=device)

    t_build = time.time()
    keep_parts = []
    keep_rows = 0
    want_sample = rank == 0 and world == 1 and not args.no_cpu_baseline
    gworld = R if not args.force_dist else max(R, 1)  # ranks in one all-gather
    for c in range(c_lo, c_hi):
        rows = min(CHUNK_ROWS, n_total - c * CHUNK_ROWS)
        x = gen_chunk(c, rows)
        sel = (p_pos_t >= c * CHUNK_ROWS) & (p_pos_t < c * CHUNK_ROWS + rows)
        if bool(sel.any()):
            x[p_pos_t[sel] - c * CHUNK_ROWS] = p_vec[sel]
        torch.cuda.synchronize()
        idx.add_device(x.data_ptr(), rows)
        if want_sample and keep_rows < args.cpu_sample_rows:
            take = min(rows, args.cpu_sample_rows - keep_rows)
            keep_parts.append(x[:take].cpu().numpy())  # host copy of the first rows for the CPU baseline
            keep_rows += take
        del x
    keep_sample = np.concatenate(keep_parts, axis=0) if keep_parts else None
    torch.cuda.synchronize()
    t_build = time.time() - t_build

    # the shard result is written straight into the packed [2,B,k] block one all-gather sends:
    # plane 0 = float8 distance bits, plane 1 = global rows.  Two sets: the all-gather + merge of step i run on a second
    # stream under the search of step i+1.
    packed2 = [torch.empty((2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
    packed = packed2[0]
    out_dist = packed[0].view(torch.float64)
    out_rows = packed[1]
    if use_dist:
        packed_all2 = [torch.empty((gworld, 2, B, k), device=device, dtype=torch.int64) for _ in range(2)]
        packed_all = packed_all2[0]
        fin_dist2 = [torch.empty((B, k), device=device, dtype=torch.float64) for _ in range(2)]
        fin_rows2 = [torch.empty((B, k), device=device, dtype=torch.int64) for _ in range(2)]
        fin_dist, fin_rows = fin_dist2[0], fin_rows2[0]
        comm_stream = torch.cuda.Stream(device)
        gather_done = [None, None]
        if args.comm == "lib":  # the library's own communicator: the unique id travels through torch's store
            first = layout.group_ranks()[0]      # one RCCL communicator per group of row shards
            uid = [pkg.Mi355Index.comm_unique_id() if rank == first else None]
            dist.broadcast_object_list(uid, src=first, group=row_group)
            idx.comm_init(layout.shard, gworld, uid[0])
    # Every search of this process runs on an EXPLICIT non-default stream, made torch's current stream from here on: the
    # default stream's handle is 0, which the library reads as "use the index's own stream" -- a stream torch events do not
    # see, so `wait_event(gather_done[buf])` below would order the wrong stream and step i + 2's search could overwrite
    # packed2[buf] while step i's all-gather still reads it.  (sharded.py takes the same precaution.)
    search_stream = torch.cuda.Stream(device)
    search_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(search_stream)
    stream = search_stream.cuda_stream
    assert stream != 0, "the search stream must not be the null stream"

    def q_of(i: int):
        return qpool[(i * QG + layout.group) % n_pool]   # every query group serves its own block of the pool

    def finish(pend):
        """Complete one enqueued block: wait (the library re-does the rare flagged queries here), then -- row shards -- one
        all-gather of the packed [2,B,k] (distance bits, rows) block per rank + the merge kernel on the second stream."""
        ticket, buf = pend
        idx.search_wait(ticket)
        if not use_dist:
            return packed2[buf][0].view(torch.float64), packed2[buf][1]
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(packed_all2[buf].view(-1), packed2[buf].view(-1), group=row_group)
            idx.merge_topk_packed_device(packed_all2[buf].data_ptr(), gworld, B, k, fin_dist2[buf].data_ptr(),
                                         fin_rows2[buf].data_ptr(), comm_stream.cuda_stream)
            gather_done[buf] = torch.cuda.Event()
            gather_done[buf].record(comm_stream)
        return fin_dist2[buf], fin_rows2[buf]

    def run_steps(first: int, count: int):
        """`count` steps, software-pipelined: block i + 1 is put on the stream BEFORE the host waits for block i, so the GPU
        goes from one block's last kernel to the next one's first without the host's round trip; the all-gather + merge of
        block i run on the second stream under block i + 1.  Every step's wait lies inside the caller's timed region."""
        res = None
        if use_dist and args.comm == "lib":   # the library's own communicator: blocking per step
            for i in range(first, first + count):
                buf = i & 1
                idx.search_sharded_device(q_of(i).data_ptr(), B, k, fin_dist2[buf].data_ptr(), fin_rows2[buf].data_ptr(), stream)
                res = (fin_dist2[buf], fin_rows2[buf])
            return res
        pend = None
        for i in range(first, first + count):
            buf = i & 1
            if use_dist and gather_done[buf] is not None:
                torch.cuda.current_stream().wait_event(gather_done[buf])  # the gather that read this block two steps ago
            pk = packed2[buf]
            if args.sync_steps:
                idx.search_device(q_of(i).data_ptr(), B, k, pk[0].data_ptr(), pk[1].data_ptr(), stream)
                res = finish((0, buf))
                continue
            t = idx.search_device_async(q_of(i).data_ptr(), B, k, pk[0].data_ptr(), pk[1].data_ptr(), stream)
            if pend is not None:
                res = finish(pend)
            pend = (t, buf)
        if pend is not None:
            res = finish(pend)
        return res

    run_steps(0, args.warmup)
    torch.cuda.synchronize()
    if have_pg:
        dist.barrier()
    idx.reset_stats()
    idx.set_option("profile", 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    if have_pg:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    idx.set_option("profile", 0)
    if have_pg:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- dominant-kernel accounting (k_screen), measured with HIP events on the launch stream
    all_launches = idx.stat("screen_launches")
    all_screen_ns = idx.stat("screen_ns")
    # the dominant kernel's own launches (k_screen256 for B > 128; the few-thousand-row first chunks go through k_screen)
    if B > 128:
        launches = idx.stat("screen256_launches")
        screen_ns = idx.stat("screen256_ns")
        screen_rows = idx.stat("screen256_rows")
    else:
        launches, screen_ns, screen_rows = all_launches, all_screen_ns, idx.stat("screen_rows")
    fallback = idx.stat("fallback_queries")
    cand = idx.stat("candidates")
    resc = idx.stat("rescored")
    i8 = idx.stat("screen_dtype_active") == 2
    dpad = (d + 127) // 128 * 128 if i8 else (d + 63) // 64 * 64
    Bpad = (B + 255) // 256 * 256 if B > 128 else 128
    peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_BF16_PEAK_TF
    flops = 2.0 * Bpad * screen_rows * dpad          # MFMA flops actually issued by k_screen
    alg_flops = 2.0 * B * screen_rows * d             # algorithmic: 2*B*N*d per pass
    alg_bytes = float(screen_rows) * d * 4            # algorithmic HBM bytes: N*d*4 per pass
    shadow_bytes = float(screen_rows) * dpad * (1 if i8 else 2)  # bytes the screen really streams (shadow rows)
    screen_s = screen_ns * 1e-9
    # HBM traffic of the dominant kernel from the committed PMC pass of this same command (rocprofv3 cannot run
    # inside the timed region): bytes per screened row x rows per launch.  See tools/collect_traffic.sh.
    traffic = None
    traffic_src = None
    for tname in (("r04_traffic_i8.json", "r03_traffic_i8.json", "r02_traffic_i8.json") if i8 else ("r02_traffic.json", "r01_traffic.json")):
        tfile = ROOT / "profiles" / tname
        if tfile.exists() and B > 128 and d == 768 and launches:
            per_row = json.loads(tfile.read_text())["hbm_read_bytes_per_screened_row"]
            traffic = round(per_row * screen_rows / launches)
            traffic_src = f"REPLAYED, not measured in this run: {per_row:.0f} B per screened row from profiles/{tname} " \
                          "(rocprofv3 --pmc FETCH_SIZE pass of this same command, x1024 x2 gfx950 correction; " \
                          "tools/collect_traffic.sh) x the rows one launch screened here"
            break
    ubench = (MFMA_I8_UBENCH_TOPS["32x32x32"] if i8 else MFMA_BF16_UBENCH_TF)
    roof = {
        "bound": "mfma",
        "kernel": ("k_screen256c" if B > 128
                   else "k_screen") + ("<int8>" if i8 else "<bf16>"),
        "op": "int8 multiply-add ops (v_mfma_i32_32x32x32_i8)" if i8 else "bf16 flops (v_mfma_f32_32x32x16_bf16)",
        "achieved": round(alg_flops / screen_s / 1e12, 2) if screen_s > 0 else None,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": round(alg_flops / screen_s / 1e12 / peak, 4) if screen_s > 0 else None,
        "frac_of_ubench_ceiling": round(alg_flops / screen_s / 1e12 / ubench, 4) if screen_s > 0 else None,
        "frac_of_power_limited_stream": round(alg_flops / screen_s / 1e12 / (MFMA_I8_POWER_LIMITED_TOPS if i8 else MFMA_BF16_POWER_LIMITED_TF), 4) if screen_s > 0 else None,
        "power_limited_stream": {"rate": MFMA_I8_POWER_LIMITED_TOPS if i8 else MFMA_BF16_POWER_LIMITED_TF,
                                 "note": "bare MFMA stream of this instruction on Gaussian operands, no memory traffic: the chip at its "
                                         "1.4 kW socket cap (1.28 kW, 1.79 GHz); tools/mfma_power_probe.hip, profiles/r04_fp4_probe.txt"},
        "ubench_ceiling": {"this_instruction": ubench,
                           "note": "cdna_hip_programming.md MFMA ubench table: i8 32x32x32 4404 TOPS, i8 16x16x64 3944 TOPS "
                                   "(the row MI355X_MICROARCH.md quotes), bf16 32x32x16 2382 TF"},
        "traffic": traffic,
        "traffic_unit": "HBM read bytes per launch (PMC FETCH_SIZE, gfx950-corrected), vs algorithmic "
                        f"{round(alg_bytes / max(launches, 1))}",
        "traffic_source": traffic_src,
        "launches": launches,
        "avg_launch_ms": round(screen_s * 1e3 / max(launches, 1), 4),
        "kernel_ms_per_step": round(screen_s * 1e3 / max(args.steps, 1), 3),
        "all_screen_kernels_ms_per_step": round(all_screen_ns * 1e-6 / max(args.steps, 1), 3),
        "all_screen_launches": all_launches,
        "issued_tflops": round(flops / screen_s / 1e12, 2) if screen_s > 0 else None,
        # the north-star's HBM view of the same launches: algorithmic N*d*4 bytes per pass over kernel time
        "hbm_view": {
            "achieved_GBps": round(alg_bytes / screen_s / 1e9, 1) if screen_s > 0 else None,
            "peak_GBps": HBM_PEAK_GBS,
            "frac": round(alg_bytes / screen_s / 1e9 / HBM_PEAK_GBS, 4) if screen_s > 0 else None,
            "streamed_GBps": round(shadow_bytes / screen_s / 1e9, 1) if screen_s > 0 else None,
        },
    }

    result = {
        "metric": "queries/sec",
        "value": round(args.steps * B * QG / elapsed, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
        "higher_is_better": True,
        # per-GPU work is fixed when every rank holds the corpus and serves its own blocks; total work is fixed when every
        # rank serves the same block against 1/world of the rows
        "scaling": "weak" if R == 1 else ("strong" if QG == 1 else f"mixed ({layout.describe()})"),
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic fp32 d={d} N={n_total} corpus ({'L2-normalised N(0,1)' if args.data == 'gaussian' else 'anisotropic stand-in for bge-base / BEIR nq: shared mean direction, power-law spectrum, near-duplicate clusters (synth.Anisotropic)'}), {B}-query blocks, "
                        f"exact {'cosine' if args.metric == 'cosine' else 'inner-product'} top-{k}",
            "rows_total": n_total,
            "rows_per_gpu": n_local,
            "dim": d,
            "k": k,
            "queries_per_step": B * QG,
            "layout": {"row_shards": R, "query_groups": QG, "rule": args.layout},
            "parallelism": f"{layout.describe()}" + ((" + one packed RCCL all-gather of the per-shard top-k + k_merge_topk per step, inside the timed loop (" + ("library RCCL communicator" if args.comm == "lib"
                            else "torch.distributed, on a second stream under the next step's search") + ")") if use_dist else ""),
            "collective": ({"transport": "library RCCL communicator (ncclAllGather)" if args.comm == "lib"
                            else "torch.distributed backend nccl (= RCCL)",
                            "ranks_in_all_gather": int(dist.get_world_size(row_group)),
                            "rccl_comm_count": idx.comm_count() if args.comm == "lib" else None,
                            "search_stream_handle_nonzero": bool(stream != 0)} if use_dist else None),
            "stepping": "blocking call per step" if args.sync_steps else "async: block i+1 enqueued before the wait for block i (mi355dr_search_device_async)",
            "arithmetic": ("int8" if i8 else "bf16") + " MFMA screen over a normalised shadow corpus (rigorous "
                          "per-query error bound), exact fp32 chain re-score, float8 distance (results bit-exact vs "
                          "CPU oracle)",
        },
        "roofline": roof,
        "extra": {
            "index_build_s": round(t_build, 2),
            "candidates_per_query_per_step": round(cand / max(args.steps * B, 1), 1),
            "rescored_per_query_per_step": round(resc / max(args.steps * B, 1), 1),
            "fallback_queries": fallback,
            "retry_queries": idx.stat("retry_queries"),
            "loose_rows": idx.stat("loose_rows"),
            "hbm_bytes_resident": idx.stat("hbm_bytes_resident"),
        },
    }

    # ---- untimed extras -------------------------------------------------------------------------------------------
    from autorag_research_amd.metrics import MetricInput, retrieval_ndcg

    gt_or, gt_and = synth.ground_truth(p_owner, p_pos, B)

    def ndcg_at_10(rows_2d) -> dict:
        """nDCG@10 (reference semantics: evaluation/metrics/retrieval.py:71-144) of ranked row ids under both ground-truth
        shapes the reference ingests (data/beir.py:191-194)."""
        top = [[str(int(r)) for r in row[:10]] for row in rows_2d]
        v_or = retrieval_ndcg([MetricInput(retrieval_gt=g, retrieved_ids=t) for g, t in zip(gt_or, top)])
        v_and = retrieval_ndcg([MetricInput(retrieval_gt=g, retrieved_ids=t) for g, t in zip(gt_and, top)])
        return {"or_group": round(float(np.mean(v_or)), 6), "and_chain": round(float(np.mean(v_and)), 6)}

    if not args.no_extras:
        # (1) the nDCG half of the metric: the planted block against the WHOLE (sharded) corpus, same code path as a step
        kk = max(k, 10)
        if kk != k:
            pk = torch.empty((2, B, kk), device=device, dtype=torch.int64)
            pd_, pr_ = pk[0].view(torch.float64), pk[1]
            idx.search_device(q_plant.data_ptr(), B, kk, pd_.data_ptr(), pr_.data_ptr(), stream)
            if use_dist:
                pall = torch.empty((gworld, 2, B, kk), device=device, dtype=torch.int64)
                dist.all_gather_into_tensor(pall.view(-1), pk.view(-1), group=row_group)
                fd = torch.empty((B, kk), device=device, dtype=torch.float64)
                fr = torch.empty((B, kk), device=device, dtype=torch.int64)
                idx.merge_topk_packed_device(pall.data_ptr(), gworld, B, kk, fd.data_ptr(), fr.data_ptr(), stream)
                pr_ = fr
        else:
            idx.search_device(q_plant.data_ptr(), B, k, out_dist.data_ptr(), out_rows.data_ptr(), stream)
            pr_ = out_rows
            if use_dist:
                dist.all_gather_into_tensor(packed_all.view(-1), packed.view(-1), group=row_group)
                idx.merge_topk_packed_device(packed_all.data_ptr(), gworld, B, k, fin_dist.data_ptr(), fin_rows.data_ptr(),
                                             stream)
                pr_ = fin_rows
        torch.cuda.synchronize()
        plant_rows_full = pr_.cpu().numpy()
        if rank == 0:
            nd = ndcg_at_10(plant_rows_full)
            hard = float(np.mean(p_sigma >= 5.0))
            result["ndcg_at_10"] = {**nd, "queries": B, "planted_rows": int(p_pos.size),
                                    "corpus_rows": n_total,
                                    "note": f"planted-answer block (SURVEY 8d): 1-3 relevant rows per query, sigma in "
                                            f"{{0.3,0.6,1,5,7}} ({hard:.0%} of them hard: cos ~0.2/0.14); group-nDCG as the "
                                            "reference computes it (BEIR = one OR-group, hotpotqa = AND-chain)"}
    if rank == 0 and world == 1 and not args.no_extras and B > 128:
        # (1b) what the chip draws and clocks at under this workload: ~2 s of the same steps with rocm-smi sampled next to them
        # (the screen kernels run AT the socket power cap, which -- not the schedule -- sets their clock)
        try:
            pp = power_probe(lambda n: (run_steps(0, n), torch.cuda.synchronize()), elapsed / args.steps)
            result["extra"]["power_probe"] = pp
            if "burst_s" in pp and "burst_steps" in pp:
                # the timed region above is a fraction of a second; this is the same loop held for >= 2 s (the governor settles
                # at the socket power cap after ~0.4 s): the rate a long-running job sees
                result["extra"]["sustained"] = {
                    "seconds": pp["burst_s"], "steps": pp["burst_steps"],
                    "ms_per_step": round(pp["burst_s"] * 1e3 / pp["burst_steps"], 3),
                    "queries_per_s": round(pp["burst_steps"] * B * QG / pp["burst_s"], 1),
                    "socket_power_W_median": pp.get("socket_power_W_median"), "sclk_MHz_median": pp.get("sclk_MHz_median"),
                    "note": "back-to-back steps of the timed loop for >= 2 s, synchronised at both ends; NOT `value`"}
        except Exception as e:  # noqa: BLE001 - a secondary figure must not take the line down
            result["extra"]["power_probe"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_extras:
        # (2) PCIe-inclusive rate: the host entry point (H2D of the query block, D2H of [B,k]) instead of device buffers
        qh = [qpool[i % n_pool].cpu().numpy() for i in range(3)]
        idx.search(qh[0], k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(5):
            idx.search(qh[i % 3], k)
        t1 = (time.perf_counter() - t1) / 5
        # "end-to-end QPS includes H2D of the query block and D2H of [B,k]" -- a top-level sibling of `value`
        # (`value` itself keeps inputs resident in HBM: the task's bench contract)
        result["value_pcie_inclusive"] = {
            "value": round(B / t1, 1), "unit": "queries/s", "ms_per_step": round(t1 * 1e3, 3),
            "note": "mi355dr_search: pageable host queries in (H2D of the query block), float8 distances + int64 rows "
                    "out (D2H of [B,k]), one blocking call per step; corpus upload excluded"}
        result["extra"]["pcie_inclusive"] = {"ms_per_step": round(t1 * 1e3, 3), "queries_per_s": round(B / t1, 1),
                                             "note": "same figure as the top-level value_pcie_inclusive (kept for older readers)"}

    if rank == 0 and world == 1 and not args.no_extras:
        # (2b) the reference's own call shape: ONE query per call (pipelines/retrieval/vector_search.py:157-169), through the
        # host entry point (H2D of the query, D2H of its k results, one blocking call per query)
        q1 = [qpool[0, i:i + 1].cpu().numpy() for i in range(8)]
        for i in range(3):
            idx.search(q1[i], k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 24
        for i in range(n1):
            idx.search(q1[i % 8], k)
        t1 = (time.perf_counter() - t1) / n1
        result["extra"]["one_query_per_call"] = {
            "ms_per_call": round(t1 * 1e3, 3), "queries_per_s": round(1.0 / t1, 1),
            "note": "mi355dr_search with B = 1 (k_screen_stream: the int8 shadow streamed once per call, DESIGN.md 4.1d); "
                    "the CPU figure for the same call shape is cpu_baselines[kind = 'torch-cpu, B=1 call shape']; NOT `value`"}

    if rank == 0 and world == 1 and not args.no_extras:
        # (2c) SURVEY 8(d): the HBM-bound regime (1 / 32 / 128 queries per call) and the second corpus size
        result["extra"]["block_sizes"] = block_size_table(idx, torch, qpool, d, k, n_total, device)
        result["extra"]["other_k"] = other_k_line(idx, torch, qpool, B, 10 if k == 100 else 100, n_total, device)
        if args.data == "gaussian" and n_total > 1_000_000:
            result["extra"]["n_1m"] = small_corpus_line(args, torch, pkg, qpool, device, local_rank, 1_000_000)

    if rank == 0 and world == 1 and not args.no_extras and args.data == "gaussian":
        # (3) the multi-vector half of the path (configs C4 / C5) at SURVEY 8(d) sizes, as secondary figures of the same run
        # SURVEY 8(d) sizes: 1 M text docs (~106 M vectors: 54 GB fp32 + 27 GB bf16 copy) and 100 k pages (103 M vectors),
        # 1000 queries each (125 steps of 8)
        # (1000 queries each: 63 steps of 16)
        result["maxsim"] = {
            "colbert_like": run_maxsim(args, 1_000_000, "text", 32, 63, 3, 0 if args.no_cpu_baseline else 1500),
            "colpali_like": run_maxsim(args, 100_000, "page", 24, 63, 3, 0),
        }

    # ---- CPU baseline (rank 0, N=1 run only): the oracle on a bounded sample of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline and keep_sample is not None:
        from oracle import cpu_ref

        S = keep_sample.shape[0]
        nq = max(1, min(args.cpu_sample_queries, n_pool * B))
        Cs = keep_sample
        Qs = qpool.reshape(n_pool * B, d)[:nq].cpu().numpy()  # the first blocks of the query pool
        cpu_ref.topk_search(Cs[:2048], Qs[:8], k)  # warm the library / thread pool
        tc = time.perf_counter()
        rd, rr = cpu_ref.topk_search(Cs, Qs, k, metric=args.metric, verify=False)  # timed: no re-check inside
        tc = time.perf_counter() - tc
        # parity on the very same sample, through the C ABI
        with pkg.Mi355Index(d, args.metric, device=local_rank) as sidx:
            sidx.add(Cs)
            gd, gr = sidx.search(Qs, k)
        parity = bool(np.array_equal(gr, rr) and np.array_equal(gd, rd))
        qps_sample = nq / tc
        result["cpu_baseline"] = {
            "value": round(qps_sample * S / n_total, 3),
            "unit": "queries/s",
            "cores": cpu_ref.num_threads(),
            "kind": "port",
            "sample": f"oracle (C, OpenMP, exact fp32 chains) on the first {S} rows x {nq} queries of this workload: "
                      f"{qps_sample:.1f} queries/s at N={S}, scaled linearly to N={n_total}; {tc:.1f} s of CPU work",
            "parity_on_sample": parity,
        }
        if not args.no_extras:
            # the planted block on the SAME sample through both paths: nDCG from the GPU ids == nDCG from the oracle's ids
            Qp = q_plant.cpu().numpy()
            od_, or_ = cpu_ref.topk_search(Cs, Qp, max(k, 10), metric=args.metric)
            with pkg.Mi355Index(d, args.metric, device=local_rank) as sidx:
                sidx.add(Cs)
                gd_, gr_ = sidx.search(Qp, max(k, 10))
            n_gpu, n_cpu = ndcg_at_10(gr_), ndcg_at_10(or_)
            result["ndcg_at_10"]["sample_check"] = {
                "rows": S, "gpu": n_gpu, "oracle": n_cpu,
                "identical": bool(n_gpu == n_cpu and np.array_equal(gr_, or_) and np.array_equal(gd_, od_))}
            assert n_gpu == n_cpu, "nDCG@10 from the GPU ids differs from nDCG@10 from the oracle ids"
            result["cpu_baselines"] = cpu_shape_baselines(
                Cs, Qs, k, args.metric, n_total,
                lambda C_, Q_: cpu_ref.topk_search(C_, Q_, k, metric=args.metric, verify=False)[1])
    if rank == 0:
        # sanity: results are sorted and in range
        rd_, rr_ = res[0].cpu().numpy(), res[1].cpu().numpy()
        assert (np.diff(rd_, axis=1) >= 0).all(), "distances not ascending"
        assert rr_.min() >= 0 and rr_.max() < n_total
    if use_dist and (world > 1 or args.force_dist) and not args.no_extras:
        try:  # (a secondary check must not take the headline line down with it; every rank takes the same path)
            # (4) the sharded answer against ONE GPU's: rank 0 builds the whole corpus next to its shard (54 GB at N = 10 M) and
            # answers block 0 of the pool alone; every rank's merged lists must equal it bit for bit
            chk = torch.empty((2, B, k), device=device, dtype=torch.int64)
            idx.search_device(qpool[0].data_ptr(), B, k, packed[0].data_ptr(), packed[1].data_ptr(), stream)
            dist.all_gather_into_tensor(packed_all.view(-1), packed.view(-1), group=row_group)
            idx.merge_topk_packed_device(packed_all.data_ptr(), gworld, B, k, chk[0].data_ptr(), chk[1].data_ptr(), stream)
            torch.cuda.synchronize()
            one = torch.zeros((2, B, k), device=device, dtype=torch.int64)
            built = torch.zeros((1,), device=device, dtype=torch.int64)
            if rank == 0:
              try:  # (rank 0 alone builds: whatever happens here, it still joins the collectives below)
                with pkg.Mi355Index(d, args.metric, device=local_rank) as whole:
                    whole.reserve(n_total)
                    whole.set_option("screen_dtype", args.screen)
                    for c in range(n_chunks):
                        rows = min(CHUNK_ROWS, n_total - c * CHUNK_ROWS)
                        x = gen_chunk(c, rows)
                        sel = (p_pos_t >= c * CHUNK_ROWS) & (p_pos_t < c * CHUNK_ROWS + rows)
                        if bool(sel.any()):
                            x[p_pos_t[sel] - c * CHUNK_ROWS] = p_vec[sel]
                        torch.cuda.synchronize()
                        whole.add_device(x.data_ptr(), rows)
                        del x
                    whole.search_device(qpool[0].data_ptr(), B, k, one[0].data_ptr(), one[1].data_ptr(), stream)
                    torch.cuda.synchronize()
                    built += 1
              except Exception as e:  # noqa: BLE001
                result["extra"]["identical_to_one_gpu_error"] = f"{type(e).__name__}: {e}"
            dist.broadcast(one, src=0)
            dist.broadcast(built, src=0)
            same = torch.tensor([1.0 if bool(torch.equal(one, chk)) else 0.0], device=device, dtype=torch.float64)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if rank == 0:
                result["extra"]["identical_to_one_gpu"] = bool(same.item() == 1.0) if int(built.item()) == 1 else None
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                result["extra"]["identical_to_one_gpu"] = f"error: {type(e).__name__}: {e}"
    want_leg = args.row_sharded_leg and (R != world or world == 1)
    ref_block = None
    if want_leg:
        # the main layout's answer for block 0 of the pool: the row-sharded leg must reproduce it bit for bit
        ref_block = torch.empty((2, B, k), device=device, dtype=torch.int64)
        if use_dist:
            idx.search_device(qpool[0].data_ptr(), B, k, packed[0].data_ptr(), packed[1].data_ptr(), stream)
            dist.all_gather_into_tensor(packed_all.view(-1), packed.view(-1), group=row_group)
            idx.merge_topk_packed_device(packed_all.data_ptr(), gworld, B, k, ref_block[0].data_ptr(), ref_block[1].data_ptr(),
                                         stream)
        else:
            idx.search_device(qpool[0].data_ptr(), B, k, ref_block[0].data_ptr(), ref_block[1].data_ptr(), stream)
        torch.cuda.synchronize()
    idx.close()
    if want_leg:
        if not have_pg:
            import torch.distributed as dist  # noqa: PLC0415

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)
            have_pg = True

        def plant(c: int, x):
            sel = (p_pos_t >= c * CHUNK_ROWS) & (p_pos_t < c * CHUNK_ROWS + x.shape[0])
            if bool(sel.any()):
                x[p_pos_t[sel] - c * CHUNK_ROWS] = p_vec[sel]
            return x

        try:
            leg = row_sharded_leg(args, torch, pkg, dist, device, local_rank, rank, world, qpool, ref_block,
                                  lambda c, rows: plant(c, gen_chunk(c, rows)))
        except Exception as e:  # noqa: BLE001 - a secondary figure must not take the headline line down with it
            leg = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            result["extra"]["row_sharded"] = leg
    if args.replicated_leg and world > 1 and QG != world:
        try:
            leg = replicated_leg(args, torch, pkg, dist, device, local_rank, rank, world, qpool, gen_chunk)
        except Exception as e:  # noqa: BLE001 - a secondary figure must not take the headline line down with it
            leg = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            result["extra"]["replicated"] = leg
    if have_pg:
        dist.destroy_process_group()
    if rank == 0 and world == 1 and not args.no_extras and B > 128:
        # (every index of this process is closed by now: the sub-run builds its own copy of the corpus)
        sub = ["--rows", n_total, "--dim", d, "--block", B, "--k", k, "--metric", args.metric, "--data", args.data,
               "--screen", args.screen]
        per_launch, n_prof, note = pmc_fetch_subrun(sub, "k_screen256c")
        rl = result["roofline"]
        if per_launch is not None:
            rl["traffic_replayed"] = rl.get("traffic")
            rl["traffic"] = round(per_launch)
            rl["traffic_source"] = (
                f"MEASURED in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE sub-run of this workload (3 steps, {n_prof} "
                "k_screen256c launches; KiB x 1024 x 2: the gfx950 correction of MI355X_MICROARCH.md), mean per launch -- the "
                "launches of a pass differ in size exactly as in the timed region")
        else:
            rl["traffic_source"] = (rl.get("traffic_source") or "") + f" [live PMC sub-run: {note}]"
    if rank == 0 and world == 1 and not args.no_extras and "maxsim" in result:
        # the MaxSim screens' HBM traffic, measured like the single-vector kernel's: a PMC sub-run of the same store + steps
        for key, tokens, docs in (("colbert_like", "text", 1_000_000), ("colpali_like", "page", 100_000)):
            maxsim_traffic(result["maxsim"][key]["roofline"], tokens, docs)
    if rank == 0:
        # RCCL writes its version banner through C stdio; on a pipe that buffer would be flushed at exit, AFTER the
        # result.  Flush it now so that the JSON line is the last line of rank 0's output.
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001 - purely cosmetic
            pass
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
