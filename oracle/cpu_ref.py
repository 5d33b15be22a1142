"""ctypes front-end of the CPU oracle (oracle/oracle.c) + small numpy float64 references.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See oracle/oracle.c for the
reference citations and the parity status ("parity unpinned" at the SQL-operator
boundary; pinned against the reference's in-process restatements through
tests/golden/).
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liboracle.so"
_lib: ctypes.CDLL | None = None


def build(force: bool = False) -> Path:
    """Compile oracle.c -> liboracle.so with the committed Makefile."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / "oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_LIB_PATH))
        f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
        i64p, i32p = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
        L.orc_dot.restype = ctypes.c_float
        L.orc_dot.argtypes = [f32p, f32p, ctypes.c_int]
        L.orc_dot_seq.restype = ctypes.c_float
        L.orc_dot_seq.argtypes = [f32p, f32p, ctypes.c_int]
        L.orc_row_nrm2.restype = None
        L.orc_row_nrm2.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p]
        L.orc_cosine_distance.restype = ctypes.c_double
        L.orc_cosine_distance.argtypes = [f32p, f32p, ctypes.c_int]
        L.orc_cosine_distance_seq.restype = ctypes.c_double
        L.orc_cosine_distance_seq.argtypes = [f32p, f32p, ctypes.c_int]
        L.orc_cosine_distance_from.restype = ctypes.c_double
        L.orc_cosine_distance_from.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.orc_topk_search.restype = ctypes.c_int
        L.orc_topk_search.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, f64p, i64p, ctypes.c_int]
        L.orc_maxsim_distance.restype = ctypes.c_float
        L.orc_maxsim_distance.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int, ctypes.c_int]
        L.orc_maxsim_topk.restype = ctypes.c_int
        L.orc_maxsim_topk.argtypes = [f32p, i64p, ctypes.c_int64, ctypes.c_int, f32p, i32p, ctypes.c_int,
                                      ctypes.c_int, f32p, i64p, ctypes.c_int]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_verify_topk.restype = ctypes.c_int64
        L.orc_verify_topk.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      f64p, i64p]
        L.orc_fpenv_fix_count.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def debug_team() -> tuple[int, int, int]:
    """(threads that entered a parallel region, team size they saw, distinct thread numbers) -- all three must agree."""
    a, b = ctypes.c_int(0), ctypes.c_int(0)
    lib().orc_debug_team.restype = ctypes.c_int
    entered = lib().orc_debug_team(ctypes.byref(a), ctypes.byref(b))
    return int(entered), int(a.value), int(b.value)


def fpenv_fix_count() -> int:
    """How many times an oracle thread was found with a non-default floating-point environment (and reset)."""
    return int(lib().orc_fpenv_fix_count())


def dot(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dot(_p(a, ctypes.c_float), _p(b, ctypes.c_float), a.shape[0]))


def row_nrm2(rows) -> np.ndarray:
    rows = _f32(rows)
    out = np.empty(rows.shape[0], dtype=np.float32)
    lib().orc_row_nrm2(_p(rows, ctypes.c_float), rows.shape[0], rows.shape[1], _p(out, ctypes.c_float))
    return out


def cosine_distance(q, c, seq: bool = False) -> float:
    q, c = _f32(q), _f32(c)
    fn = lib().orc_cosine_distance_seq if seq else lib().orc_cosine_distance
    return float(fn(_p(q, ctypes.c_float), _p(c, ctypes.c_float), q.shape[0]))


rejected_results = 0  # multi-threaded results that failed the single-threaded re-check (see topk_search)


def topk_search(C, Q, k: int, metric: str = "cosine", threads: int = 0, verify: bool = True
                ) -> tuple[np.ndarray, np.ndarray]:
    """Exact brute-force top-k.  Returns (distance float64 [B,k], rows int64 [B,k]); pads with NaN / -1.

    distance = pgvector cosine distance (metric="cosine") or negative inner product ("ip");
    order = (distance asc, NaN last, row asc).

    verify: every returned pair is re-derived single-threaded (orc_verify_topk).  A result that fails is recomputed with
    one thread and counted in `rejected_results` -- seen (rarely, never reproduced in isolation) on a 256-CPU host
    throttled to a 16-CPU quota, where a 256-thread run once returned a row with another row's distance.  The timed
    CPU baseline passes verify=False.
    """
    C, Q = _f32(C), _f32(Q)
    if Q.ndim == 1:
        Q = Q[None, :]
    n, d = (C.shape[0], C.shape[1]) if C.ndim == 2 else (0, Q.shape[1])
    B = Q.shape[0]
    dist = np.empty((B, k), dtype=np.float64)
    rows = np.empty((B, k), dtype=np.int64)
    rc = lib().orc_topk_search(_p(C, ctypes.c_float), n, d, _p(Q, ctypes.c_float), B, k,
                               0 if metric == "cosine" else 1, _p(dist, ctypes.c_double), _p(rows, ctypes.c_int64),
                               threads)
    if rc != 0:
        raise ValueError(f"orc_topk_search failed rc={rc}")
    if verify and n > 0 and threads != 1:
        m = 0 if metric == "cosine" else 1
        if lib().orc_verify_topk(_p(C, ctypes.c_float), n, d, _p(Q, ctypes.c_float), B, k, m, _p(dist, ctypes.c_double),
                                 _p(rows, ctypes.c_int64)) != 0:
            global rejected_results
            rejected_results += 1
            return topk_search(C, Q, k, metric, threads=1, verify=False)
    return dist, rows


def maxsim_distance(doc, q) -> float:
    doc, q = _f32(doc), _f32(q)
    return float(lib().orc_maxsim_distance(_p(doc, ctypes.c_float), doc.shape[0], _p(q, ctypes.c_float),
                                           q.shape[0], q.shape[1]))


def maxsim_topk(tok, offsets, qtok, q_off, k: int, threads: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """MaxSim top-k over ragged docs.  Returns (distance float32 [B,k] = -sum_i max_j <q_i,d_j>, doc rows)."""
    tok, qtok = _f32(tok), _f32(qtok)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    q_off = np.ascontiguousarray(q_off, dtype=np.int32)
    B = q_off.shape[0] - 1
    d = qtok.shape[1]
    dist = np.empty((B, k), dtype=np.float32)
    rows = np.empty((B, k), dtype=np.int64)
    rc = lib().orc_maxsim_topk(_p(tok, ctypes.c_float), _p(offsets, ctypes.c_int64), offsets.shape[0] - 1, d,
                               _p(qtok, ctypes.c_float), _p(q_off, ctypes.c_int32), B, k,
                               _p(dist, ctypes.c_float), _p(rows, ctypes.c_int64), threads)
    if rc != 0:
        raise ValueError(f"orc_maxsim_topk failed rc={rc}")
    return dist, rows


# ---- independent float64 numpy references (small cases; cross-check of the C code) ----


def np_cosine_distance_matrix(C, Q) -> np.ndarray:
    """float64 cosine distance [B,N]; NaN where a norm is zero (SQL semantics, not gqr_hybrid's epsilon)."""
    C64, Q64 = np.asarray(C, dtype=np.float64), np.asarray(Q, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        sim = (Q64 @ C64.T) / np.sqrt(np.sum(Q64 * Q64, axis=1)[:, None] * np.sum(C64 * C64, axis=1)[None, :])
    return 1.0 - np.clip(sim, -1.0, 1.0)


def np_topk_from_distance(dist: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Total order (distance asc, NaN last, row asc) on a dense [B,N] distance matrix."""
    B, N = dist.shape
    out_d = np.full((B, k), np.nan)
    out_r = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        key = np.where(np.isnan(dist[b]), np.inf, dist[b])
        order = np.lexsort((np.arange(N), key, np.isnan(dist[b])))[:k]
        out_d[b, : order.size] = dist[b, order]
        out_r[b, : order.size] = order
    return out_d, out_r


def np_maxsim_scores(doc_list, q) -> np.ndarray:
    """float64 (1/n_q) * sum_i max_j <q_i, d_j> per doc (mirrors the documented score, not the distance)."""
    q64 = np.asarray(q, dtype=np.float64)
    return np.array([np.max(q64 @ np.asarray(dm, dtype=np.float64).T, axis=1).sum() / max(q64.shape[0], 1)
                     for dm in doc_list])


def colbert_rerank_scores(query_emb, query_mask, doc_embs, doc_masks) -> np.ndarray:
    """Restatement of the reference ColBERT reranker's `_maxsim_score` (autorag_research/rerankers/colbert.py:63-84) for one
    query against n documents on padded inputs: fp32 `Q @ D^T`, document padding -> -inf, row max, clamp(min=0), query padding
    multiplied out, sum / number of valid query tokens.  Pinned by tests/golden/rerank_golden.npz (outputs of the reference's
    own method); the fp32 summation order of torch.matmul is not part of its contract, so parity is to 1e-6."""
    q = np.asarray(query_emb, dtype=np.float32).reshape(-1, np.asarray(query_emb).shape[-1])
    qm = np.asarray(query_mask).reshape(-1).astype(np.float32)
    out = np.zeros((len(doc_embs),), dtype=np.float64)
    for i, (d, m) in enumerate(zip(doc_embs, doc_masks)):
        sim = q @ np.asarray(d, dtype=np.float32).T
        sim = np.where(np.asarray(m).reshape(1, -1) == 0, -np.inf, sim).astype(np.float32)
        mx = sim.max(axis=-1) if sim.shape[1] else np.full((q.shape[0],), -np.inf, np.float32)
        mx = np.maximum(mx, np.float32(0.0)) * qm
        with np.errstate(invalid="ignore", divide="ignore"):
            out[i] = float(np.float32(mx.sum(dtype=np.float32)) / np.float32(qm.sum(dtype=np.float32)))
    return out
