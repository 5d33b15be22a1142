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

from bench_support import (HBM_PEAK_GBS, MFMA_BF16_PEAK_TF, MFMA_BF16_POWER_LIMITED_TF, MFMA_BF16_UBENCH_TF,  # noqa: E402
                           MFMA_I8_PEAK_TOPS, MFMA_I8_POWER_LIMITED_TOPS, MFMA_I8_UBENCH_TOPS, block_size_table,
                           bare_stream_probe, cpu_shape_baselines, main_maxsim, other_k_line, pmc_fetch_subrun, maxsim_traffic, power_probe,
                           replicated_leg, row_sharded_leg, run_maxsim, small_corpus_line)


class _HostHopDist:
    """torch.distributed with its device collectives taken through the host (--rehearse-one-gpu: gloo between ranks that share
    ONE GPU).  Everything else is the module itself."""

    def __init__(self, dist, torch):
        self._d, self._t = dist, torch

    def __getattr__(self, name):
        return getattr(self._d, name)

    def all_gather_into_tensor(self, out, inp, group=None):
        self._t.cuda.current_stream().synchronize()
        host = self._t.empty(out.shape, dtype=out.dtype)
        self._d.all_gather_into_tensor(host, inp.cpu(), group=group)
        out.copy_(host)

    def all_reduce(self, t, op=None, group=None):
        host = t.cpu()
        self._d.all_reduce(host, op=op if op is not None else self._d.ReduceOp.SUM, group=group)
        t.copy_(host)

    def broadcast(self, t, src, group=None):
        host = t.cpu()
        self._d.broadcast(host, src=src, group=group)
        t.copy_(host)


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
    ap.add_argument("--screen-rq", type=int, default=None, help="0/1: large-block int8 screen with the query operand in registers "
                    "(k_screen_rq, default) / through the LDS (k_screen256c) (A/B)")
    ap.add_argument("--screen-rq-split-tests", type=int, default=None, help="k_screen_rq: 0 = every block test in one piece (A/B)")
    ap.add_argument("--debug-park", type=int, default=0, help="diagnostic: k_screen_rq launches of at least this many rows run with every "
                    "threshold at +inf (what such a launch costs without hits; the results of such a run are WRONG)")
    ap.add_argument("--screen-drift", type=int, default=None, help="k_screen_rq: tiles a workgroup may lead its siblings by (0 = no limiter; A/B)")
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
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="developer / test: run the N > 1 leg with every rank on cuda:0 (one-GPU boxes; RCCL refuses two ranks on one "
                         "device): gloo process group, device collectives take a host hop.  The line says so in config.collective; "
                         "its value is NOT a multi-GPU measurement")
    ap.add_argument("--traffic", action="store_true", help="run the PMC sub-run (rocprofv3 --pmc FETCH_SIZE of this workload, 3 steps) "
                                                           "that fills roofline.traffic even with --no-extras")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="library option passed to mi355dr_set_option verbatim (developer A/B; repeatable)")
    ap.add_argument("--cpu-sample-rows", type=int, default=2_500_000)
    ap.add_argument("--cpu-sample-queries", type=int, default=3072, help="CPU baseline: queries timed (whole blocks of the pool)")
    return ap.parse_args()


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
    if args.rehearse_one_gpu:
        local_rank = 0   # every rank on the one device
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
        if args.rehearse_one_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist = _HostHopDist(dist, torch)         # device tensors of the collectives travel through the host
        else:
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
    if args.screen_rq is not None:
        idx.set_option("screen_rq", args.screen_rq)
    if args.screen_drift is not None:
        idx.set_option("screen_drift", args.screen_drift)
    if args.debug_park:
        idx.set_option("debug_park_thresholds", args.debug_park)
    if args.screen_rq_split_tests is not None:
        idx.set_option("screen_rq_split_tests", args.screen_rq_split_tests)
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
    for kv in args.opt:
        key, _, val = kv.partition("=")
        idx.set_option(key, int(val))
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
    # which large-block form ran: k_screen_rq (int8 shadow <= 768 B per row: query operand in registers) or k_screen256c
    dominant = ("k_screen_rq" if idx.stat("screen_rq_launches") > 0 else "k_screen256c") if B > 128 else "k_screen"
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
    # HBM traffic of the dominant kernel: measured by a PMC sub-run of this very workload behind the timed region (below) or
    # absent -- nothing is replayed from an earlier round's profile (rocprofv3 cannot run inside the timed region)
    traffic = None
    traffic_src = "not measured (run without --no-extras, or with --traffic, for the in-run rocprofv3 --pmc FETCH_SIZE sub-run)"
    ubench = (MFMA_I8_UBENCH_TOPS["32x32x32"] if i8 else MFMA_BF16_UBENCH_TF)
    roof = {
        "bound": "mfma",
        "kernel": dominant + ("<int8>" if i8 else "<bf16>"),
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
            "collective": ({"transport": "REHEARSAL on one GPU: gloo, device tensors through the host (not a multi-GPU measurement)"
                            if args.rehearse_one_gpu else "library RCCL communicator (ncclAllGather)" if args.comm == "lib"
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
                result["sustained"] = result["extra"]["sustained"] = {   # top-level too: the driver's record keeps top-level keys
                    "seconds": pp["burst_s"], "steps": pp["burst_steps"],
                    "ms_per_step": round(pp["burst_s"] * 1e3 / pp["burst_steps"], 3),
                    "queries_per_s": round(pp["burst_steps"] * B * QG / pp["burst_s"], 1),
                    "socket_power_W_median": pp.get("socket_power_W_median"), "sclk_MHz_median": pp.get("sclk_MHz_median"),
                    "note": "back-to-back steps of the timed loop for >= 2 s, synchronised at both ends; NOT `value`"}
        except Exception as e:  # noqa: BLE001 - a secondary figure must not take the line down
            result["extra"]["power_probe"] = {"error": f"{type(e).__name__}: {e}"}
        # (1c) the denominator of `frac_of_power_limited_stream`, measured HERE: the bare stream of the screen's own MFMA
        # instruction on this chip, right after the timed region (same box, same thermal state)
        try:
            bs = bare_stream_probe(local_rank, i8)
            rl = result["roofline"]
            rl["bare_stream"] = bs
            if bs.get("tops") and rl.get("achieved"):
                rl["frac_of_power_limited_stream_replayed"] = rl.get("frac_of_power_limited_stream")
                rl["frac_of_power_limited_stream"] = round(rl["achieved"] / bs["tops"], 4)
                rl["power_limited_stream"] = {"rate": bs["tops"], "source": "roofline.bare_stream (measured in this run)"}
        except Exception as e:  # noqa: BLE001
            result["roofline"]["bare_stream"] = {"error": f"{type(e).__name__}: {e}"}
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
            "colbert_like": run_maxsim(args, 1_000_000, "text", 32, 63, 3, 0 if args.no_cpu_baseline else 20000, probe=True),
            "colpali_like": run_maxsim(args, 100_000, "page", 24, 63, 3, 0, probe=True),
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
    if rank == 0 and world == 1 and (not args.no_extras or args.traffic) and B > 128:
        # (every index of this process is closed by now: the sub-run builds its own copy of the corpus)
        sub = ["--rows", n_total, "--dim", d, "--block", B, "--k", k, "--metric", args.metric, "--data", args.data,
               "--screen", args.screen]
        dom = result["roofline"]["kernel"].split("<")[0]
        if args.screen_rq is not None:
            sub += ["--screen-rq", args.screen_rq]
        if args.screen_drift is not None:
            sub += ["--screen-drift", args.screen_drift]
        for kv in args.opt:
            sub += ["--opt", kv]
        per_launch, n_prof, note = pmc_fetch_subrun(sub, dom)
        rl = result["roofline"]
        if per_launch is not None:
            rl["traffic"] = round(per_launch)
            rl["traffic_source"] = (
                f"MEASURED in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE sub-run of this workload (3 steps, {n_prof} "
                f"{dom} launches; KiB x 1024 x 2: the gfx950 correction of MI355X_MICROARCH.md), mean per launch -- the "
                "launches of a pass differ in size exactly as in the timed region")
        else:
            rl["traffic_source"] = f"not measured [live PMC sub-run: {note}]"
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
