"""bench_support.py -- the pieces of bench.py that are not the headline step loop: constants of the rooflines, the CPU call-shape
baselines, the MaxSim workload, the power and PMC sub-measurements, the secondary legs (row-sharded / replicated layouts, other
block sizes, the other k, the 1 M-row corpus).  `bench.py` keeps the driver's contract (arguments, the timed region, the JSON
line); everything here is called from it.  torch is plumbing only; every timed kernel is libmi355dr's hand-written HIP."""

from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

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
    for kv in getattr(args, "opt", []):   # (library options as KEY=VALUE, like the single-vector workload: A/B runs and tests)
        key, _, val = kv.partition("=")
        idx.set_option(key, int(val))
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
            # the CPU baseline's sample: the first docs of the store, about `cpu_sample_docs` x 106 token vectors of them (the
            # mean text doc) whatever the document length -- 5 ... 10 s of oracle work at 16 queries (the
            # 1 500-doc sample was 0.4 s, too small to be a baseline)
            S = max(1, min(int(np.searchsorted(np.cumsum(ln), cpu_sample_docs * 106)), len(ln)))
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
    # (the counters of the TIMED steps: the power probe below runs more steps)
    screened, cands, fb = idx.stat("maxsim_screened"), idx.stat("maxsim_candidates"), idx.stat("maxsim_fallbacks")
    scr_n, scr_ns = idx.stat("maxsim_screen_launches"), idx.stat("maxsim_screen_ns")
    ex_n, ex_ns = idx.stat("maxsim_exact_launches"), idx.stat("maxsim_exact_ns")
    cols_issued = idx.stat("maxsim_screen_cols") / max(scr_n, 1)  # query columns per launch, whole blocks of 32
    # the granule-packed copy (csrc/k_maxsim_wg8.h; default: taken when it has >= 5 % fewer blocks): what the launches multiplied
    packed_n, packed_blocks = idx.stat("maxsim_packed_launches"), idx.stat("maxsim_packed_blocks")
    probe_out = None
    if probe:  # socket power / shader clock next to ~2 s of the same steps (not timed)
        try:
            probe_out = power_probe(lambda n: [step(i % (warmup + steps)) for i in range(n)], el / steps)
        except Exception as e:  # noqa: BLE001
            probe_out = {"error": f"{type(e).__name__}: {e}"}
    blocks = int(((lens + 31) // 32).sum())
    packed = bool(scr_n) and packed_n == scr_n and packed_blocks > 0
    padded_blocks = blocks
    if packed:
        blocks = int(packed_blocks)
    n_tok = float(lens.sum())
    alg_bytes = n_tok * d * 4                                    # fp32 token rows read once per pass
    streamed = float(blocks) * 32 * d * 2                        # bf16 fragment store the screen streams, per pass
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
        "roofline": {"bound": "mfma", "kernel": "k_maxsim16_wg8" if packed else (lambda ncb: f"k_maxsim16_wg<{ncb}>" if ncb > 8 else f"k_maxsim16_d128<{ncb}>")((qblock * nq + 31) // 32),
                     "token_copy": (f"granule-packed bf16 copy: {blocks} blocks of 32 tokens (documents rounded up to 8 tokens; the padded "
                                    f"copy has {padded_blocks})" if packed else f"padded bf16 copy: {blocks} blocks of 32 tokens"),
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
    if probe and scr_n:
        # the denominator of frac_of_power_limited_stream measured in this run (bare bf16 MFMA stream on this chip)
        try:
            bs = bare_stream_probe(dev.index, False)
            out["roofline"]["bare_stream"] = bs
            if bs.get("tops"):
                out["roofline"]["frac_of_power_limited_stream_replayed"] = out["roofline"]["frac_of_power_limited_stream"]
                out["roofline"]["frac_of_power_limited_stream"] = round(issued_flops / scr_s / 1e12 / bs["tops"], 4)
        except Exception as e:  # noqa: BLE001
            out["roofline"]["bare_stream"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def bare_stream_probe(device: int, i8: bool, seconds: float = 2.5) -> dict:
    """The matrix pipe's rate on THIS chip, in THIS run, with nothing but the screen's MFMA instruction on Gaussian operands
    (mi355dr_diag_mfma_stream), and the socket power / clock it settles at: the denominator of
    `roofline.frac_of_power_limited_stream`."""
    from autorag_research_amd import _native

    box = {}

    def run(_n):
        box["tops"] = _native.diag_mfma_stream(device, "i8" if i8 else "bf16", seconds)

    pp = power_probe(run, seconds, seconds, min_steps=1)
    if "tops" not in box:
        return {"error": pp.get("error", "the stream did not run")}
    return {"tops": round(box["tops"], 1), "watts": pp.get("socket_power_W_median"), "mhz": pp.get("sclk_MHz_median"),
            "power_cap_W": pp.get("power_cap_W"), "seconds": seconds,
            "instruction": "v_mfma_i32_32x32x32_i8" if i8 else "v_mfma_f32_32x32x16_bf16",
            "note": "bare MFMA stream measured in this run after the timed region: operands in registers (Gaussian, as the "
                    "shadows hold), four accumulators, two waves per SIMD on every CU, no memory traffic; settled rate of the "
                    "second half of the launches; rocm-smi polled next to it"}


def power_probe(run_n_steps, s_per_step: float, seconds: float = 2.0, min_steps: int = 20) -> dict:
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
    n = max(min_steps, int(seconds / max(s_per_step, 1e-4)))
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
                   0 if args.no_cpu_baseline else 20000, probe=not args.no_extras)
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
