#!/usr/bin/env python3
"""Real-weight parity of the encode side ("must be checked on a box that has weights").

The build container and the GPU boxes have no network, no checkpoints, no `sentence_transformers` and no `colpali_engine`:
every encode-side test in tests/ uses random-init stand-ins with the right SHAPES.  This script is what a provisioned box
runs once: given local checkpoint directories it compares this package's embedding wrappers with the libraries the reference
itself calls, on the same inputs, and then checks that retrieval through the GPU index ranks like the CPU math on those
embeddings.

    python tools/check_real_encoder.py --bge /ckpt/bge-base-en-v1.5 --minilm /ckpt/all-MiniLM-L6-v2 \
        --colpali /ckpt/colpali-v1.3 --colbert /ckpt/colbertv2.0 [--device cuda:0] [--images dir_with_pngs]

What is compared (reference file:line) and the tolerance each check is held to:

  bge / MiniLM     sentence_transformers.SentenceTransformer(path).encode(texts, normalize_embeddings=True)   vs
                   embeddings.load_embedding_model("mi355_bge_base" | "mi355_minilm")  (the YAML names: TorchEncoderEmbeddings
                   .from_pretrained, pooling "cls" | "mean")
                   (the reference loads such models through langchain HuggingFaceEmbeddings configs, configs/embedding/*.yaml;
                   bge-base: CLS pooling + L2 norm, MiniLM: mean pooling + L2 norm)
                   max |delta| <= 2e-5 (fp32 on both sides, different kernels), cosine >= 1 - 1e-6 per text
  ColPali          colpali_engine ColPali + ColPaliProcessor driven the way embeddings/colpali.py:109-245 drives them
                   (process_queries / process_images -> model(**inputs) -> per-item rows)   vs
                   multimodal.Mi355ColPaliEmbeddings(model_type="pali") with the SAME model object
                   bit-identical rows (the wrapper calls the same modules; the check pins batching / padding / ordering),
                   and encode_images_to_device -> Mi355Index MaxSim top-k == numpy MaxSim top-k on the reference's lists
  ColBERT rerank   rerankers/colbert.py:41-84 (`AutoModel(...).last_hidden_state`, L2 norm, `_maxsim_score`)   vs
                   rerank.Mi355ColBERTReranker with an encoder built on the same AutoModel
                   max |score delta| <= 2e-6, identical ranking

Exit code 0 when every requested check passes; each check prints one line `name: PASS|FAIL  details`.
This script cannot be exercised in the build container (see above); it only uses public APIs of the named libraries.
"""

from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

TEXTS = [
    "What is the capital of France?",
    "The mitochondrion is the powerhouse of the cell.",
    "Late interaction models keep one vector per token and score with MaxSim.",
    "short",
    "A considerably longer passage about dense retrieval: a bi-encoder embeds queries and passages into one vector space, "
    "and the top-k passages by inner product are returned to the reader model for answer extraction.",
]
QUERIES = ["capital of France", "which organelle produces energy", "how does ColBERT score documents"]

RESULTS: list[tuple[str, bool, str]] = []


def report(name: str, ok: bool, details: str) -> None:
    RESULTS.append((name, ok, details))
    print(f"{name}: {'PASS' if ok else 'FAIL'}  {details}", flush=True)


def check_single_vector(path: str, pooling: str, device: str) -> None:
    import os

    from sentence_transformers import SentenceTransformer

    from autorag_research_amd.embeddings import load_embedding_model

    ref = SentenceTransformer(path, device=device)
    want = ref.encode(TEXTS + QUERIES, normalize_embeddings=True, convert_to_numpy=True).astype(np.float32)
    # the model BY CONFIG NAME, the way a pipeline YAML's `embedding_model:` string resolves it (configs/embedding/mi355_*.yaml ->
    # TorchEncoderEmbeddings.from_pretrained), with the YAML's environment variables pointed at the given directory
    name, var = ("mi355_bge_base", "MI355_BGE_PATH") if pooling == "cls" else ("mi355_minilm", "MI355_MINILM_PATH")
    os.environ.update({var: path, "MI355_ENCODER_DEVICE": device, "MI355_ENCODER_DTYPE": "float32"})
    enc = load_embedding_model(name)
    assert enc.pooling == pooling
    got = np.asarray(enc.embed_documents(TEXTS) + [enc.embed_query(q) for q in QUERIES], dtype=np.float32)
    delta = float(np.abs(got - want).max())
    cos = float((got * want).sum(axis=1).min())
    report(f"single-vector {Path(path).name} ({pooling})", delta <= 2e-5 and cos >= 1 - 1e-6,
           f"max |delta| {delta:.2e}, min cosine {cos:.8f}, dim {got.shape[1]}")
    # retrieval through the GPU index ranks like the CPU math on these embeddings
    if device.startswith("cuda"):
        import autorag_research_amd as pkg

        with pkg.Mi355Index(got.shape[1], "cosine", device=int(device.split(":")[1]) if ":" in device else 0) as idx:
            idx.add(got[: len(TEXTS)])
            _, rows = idx.search(got[len(TEXTS):], k=len(TEXTS))
        cpu = np.argsort(-(want[len(TEXTS):] @ want[: len(TEXTS)].T), axis=1, kind="stable")
        report(f"ranking {Path(path).name}", bool(np.array_equal(rows, cpu)), f"GPU ranks {rows[0].tolist()} vs CPU {cpu[0].tolist()}")


def check_colpali(path: str, device: str, images_dir: str | None) -> None:
    import torch
    from colpali_engine.models import ColPali, ColPaliProcessor

    from autorag_research_amd.multimodal import Mi355ColPaliEmbeddings

    model = ColPali.from_pretrained(path, torch_dtype=torch.bfloat16).to(device).eval()
    proc = ColPaliProcessor.from_pretrained(path)
    ours = Mi355ColPaliEmbeddings(model_name=path, model_type="pali", device=device, torch_dtype="bfloat16", model=model,
                                  processor=proc, batch_size=4)

    def ref_queries(qs):  # embeddings/colpali.py:120-133, one text per call
        out = []
        for q in qs:
            inp = {k: v.to(device) for k, v in proc.process_queries([q]).items()}
            with torch.no_grad():
                out.append(model(**inp)[0].float().cpu().numpy())
        return out

    want_q = ref_queries(QUERIES)
    got_q = [np.asarray(ours.embed_query(q), dtype=np.float32) for q in QUERIES]
    ok = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got_q, want_q))
    report("ColPali queries", ok, f"rows per query {[a.shape[0] for a in got_q]} (reference {[a.shape[0] for a in want_q]})")
    if images_dir:
        from PIL import Image

        files = sorted(Path(images_dir).glob("*.png"))[:8]
        imgs = [Image.open(f).convert("RGB") for f in files]
        inp = {k: v.to(device) for k, v in proc.process_images(imgs).items()}  # embeddings/colpali.py:223-245
        with torch.no_grad():
            want_d = [e.float().cpu().numpy() for e in model(**inp)]
        got_d = [np.asarray(e, dtype=np.float32) for e in ours.embed_images([str(f) for f in files])]
        ok = all(a.shape == b.shape and np.allclose(a, b, atol=1e-6) for a, b in zip(got_d, want_d))
        report("ColPali pages", ok, f"{len(files)} pages, rows per page {[a.shape[0] for a in got_d]}")
        if device.startswith("cuda"):
            import autorag_research_amd as pkg

            flat, off = ours.encode_images_to_device([str(f) for f in files])
            with pkg.Mi355Index(flat.shape[1], "cosine", device=flat.device.index or 0) as idx:
                torch.cuda.synchronize()
                idx.add_multivec_device(flat.data_ptr(), off)
                qtok = np.concatenate(want_q)
                qoff = np.concatenate([[0], np.cumsum([a.shape[0] for a in want_q])]).astype(np.int32)
                dist, rows = idx.search_maxsim(qtok, qoff, k=len(files))
            cpu = np.stack([np.argsort([-(q @ d.T).max(axis=1).sum() for d in want_d], kind="stable") for q in want_q])
            report("ColPali MaxSim ranking", bool(np.array_equal(rows, cpu)), f"GPU {rows[0].tolist()} vs numpy {cpu[0].tolist()}")


def check_colbert_reranker(path: str, device: str) -> None:
    import torch
    from transformers import AutoModel, AutoTokenizer

    from autorag_research_amd.rerank import Mi355ColBERTReranker

    tok = AutoTokenizer.from_pretrained(path)
    model = AutoModel.from_pretrained(path).to(device).eval()

    def encode(texts):  # rerankers/colbert.py:45-61
        enc = tok(texts, padding=True, truncation=True, max_length=512, return_tensors="pt").to(device)
        with torch.no_grad():
            h = torch.nn.functional.normalize(model(**enc).last_hidden_state.float(), dim=-1)
        return h, enc["attention_mask"]

    def ref_scores(query, docs):  # rerankers/colbert.py:63-84
        q, qm = encode([query])
        d, dm = encode(docs)
        sim = torch.matmul(q.unsqueeze(0).squeeze(0), d.transpose(-1, -2))          # [n_docs, Lq, Ld]
        sim = sim.masked_fill(dm.unsqueeze(1) == 0, float("-inf"))
        mx = sim.max(dim=-1).values.clamp(min=0) * qm.float()
        return (mx.sum(-1) / qm.float().sum()).cpu().numpy()

    class Enc:
        def encode(self, texts):
            return encode(texts)

    want = ref_scores(QUERIES[2], TEXTS)
    res = Mi355ColBERTReranker(Enc(), model_name=path, device=int(device.split(":")[1]) if ":" in device else 0).rerank(QUERIES[2], TEXTS)
    got = np.asarray([r.score for r in sorted(res, key=lambda r: r.index)])
    report("ColBERT reranker", float(np.abs(got - want).max()) <= 2e-6 and [r.index for r in res] == list(np.argsort(-want, kind="stable")),
           f"max |score delta| {float(np.abs(got - want).max()):.2e}, order {[r.index for r in res]}")


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--bge", help="local directory of BAAI/bge-base-en-v1.5 (CLS pooling)")
    ap.add_argument("--minilm", help="local directory of sentence-transformers/all-MiniLM-L6-v2 (mean pooling)")
    ap.add_argument("--colpali", help="local directory of vidore/colpali-v1.3")
    ap.add_argument("--colbert", help="local directory of colbert-ir/colbertv2.0 (reranker)")
    ap.add_argument("--images", help="directory of page images (*.png) for the ColPali document side")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    if not any((args.bge, args.minilm, args.colpali, args.colbert)):
        ap.error("give at least one checkpoint directory")
    for name, fn in (("bge", lambda: check_single_vector(args.bge, "cls", args.device)),
                     ("minilm", lambda: check_single_vector(args.minilm, "mean", args.device)),
                     ("colpali", lambda: check_colpali(args.colpali, args.device, args.images)),
                     ("colbert", lambda: check_colbert_reranker(args.colbert, args.device))):
        if getattr(args, name):
            try:
                fn()
            except ImportError as e:
                report(name, False, f"missing library: {e}")
    return 0 if RESULTS and all(ok for _, ok, _ in RESULTS) else 1


if __name__ == "__main__":
    sys.exit(main())
