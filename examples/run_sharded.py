"""The retrieval pipeline on N GPUs of one node: one process per GPU, corpus row-sharded, `run()` unchanged.

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 examples/run_sharded.py

Every rank constructs the SAME pipeline over the SAME database / store (here a synthetic in-memory store, regenerated
identically on every rank; in the reference's flow the Executor does this: executor.py:383-463).  Because a torch.distributed
process group is up when the pipeline is constructed, `Mi355RetrievalService` row-shards the corpus over the ranks (this rank
keeps 1 / N of the rows on GPU `LOCAL_RANK`), answers every page of `run()` with all ranks together -- local exact top-k, ONE
all-gather of the packed [B, k] lists over RCCL / xGMI, merge: the same lists on every rank, bit-identical to the single-GPU
answer -- and lets rank 0 alone read the page's query ids and write the results.  `MI355DR_DISTRIBUTED=0` switches it off
(every process then runs the whole pipeline on its own, as before)."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from autorag_research_amd.pipelines import Mi355VectorSearchPipelineConfig  # noqa: E402
from autorag_research_amd.store import InMemoryStore  # noqa: E402

rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
if world > 1:
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))

rng = np.random.default_rng(0)                      # the same store on every rank (a real run: the same database)
n, d, nq = 400_000, 384, 2048
corpus = rng.standard_normal((n, d)).astype(np.float32)
store = InMemoryStore()
store.set_chunks([f"chunk-{i}" for i in range(n)], [None] * n, embedding=corpus)
qs = corpus[:nq] + 0.3 * rng.standard_normal((nq, d)).astype(np.float32)
store.add_queries([f"q{i}" for i in range(nq)], contents=[f"query {i}" for i in range(nq)], embedding=list(qs))

cfg = Mi355VectorSearchPipelineConfig(name="mi355_vector_search", search_mode="single", top_k=10, batch_size=1024)
p = cfg.get_pipeline_class()(session_factory=lambda: store, name=cfg.name, schema=None, **cfg.get_pipeline_kwargs())
stats = p.run(**cfg.get_run_kwargs())
u = p._service._unit("chunk")
held = len(u.single_sharded.index) if u.single_sharded is not None else len(u.single)
print(f"rank {rank}/{world}: holds {held} of {n} rows on cuda:{p._service._device}; run -> {stats['total_queries']} queries, "
      f"{stats['total_results']} results, failed {stats['failed_queries']}; rows persisted HERE: "
      f"{sum(len(v) for v in store.chunk_results.values())}")
if rank == 0:
    top = store.chunk_results[(p.pipeline_id, "q7")][0]
    print("q7 ->", top, "(its planted neighbour is chunk-7)")
p.close()
if world > 1:
    dist.destroy_process_group()
