"""Mi355Index -- Python handle over one libmi355dr index (one GPU, one corpus shard).

Replaces what BaseVectorRepository asked PostgreSQL to do (reference
autorag_research/orm/repository/base.py:378-426 `<=>`, :487-571 `@#`): it owns the rows in HBM and
answers exact top-k.  Rows are dense indices in insertion order; id mapping lives in store.py.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _native
from ._native import NativeError, check, f32c, ptr

METRICS = {"cosine": 0, "ip": 1}
PATHS = {"auto": 0, "screen": 1, "scan": 2}
SCREEN_DTYPES = {"auto": 0, "bf16": 1, "i8": 2}


class Mi355Index:
    def __init__(self, dim: int, metric: str = "cosine", device: int = 0):
        if metric not in METRICS:
            raise ValueError(f"metric must be one of {sorted(METRICS)}")
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        rc = self._lib.mi355dr_create(ctypes.byref(self._h), int(device), int(dim), METRICS[metric])
        if rc != 0:
            msg = self._lib.mi355dr_last_error(None)
            raise NativeError(rc, msg.decode() if msg else "")
        self.dim = int(dim)
        self.metric = metric
        self.device = int(device)

    # ---- lifetime ----
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mi355dr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # noqa: D105
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    # ---- corpus ----
    def __len__(self) -> int:
        return int(self._lib.mi355dr_size(self._h))

    def reserve(self, n_rows: int) -> None:
        check(self._h, self._lib.mi355dr_reserve(self._h, int(n_rows)))

    def add(self, rows) -> None:
        rows = f32c(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}], got {rows.shape}")
        check(self._h, self._lib.mi355dr_add_rows(self._h, ptr(rows, ctypes.c_float), rows.shape[0]))

    def add_device(self, dev_ptr: int, n: int) -> None:
        """Append n rows already resident on this device (fp32, row-major, contiguous)."""
        check(self._h, self._lib.mi355dr_add_rows_device(self._h, ctypes.c_void_p(int(dev_ptr)), int(n)))

    def get_rows(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), dtype=np.float32)
        check(self._h, self._lib.mi355dr_get_rows(self._h, int(row0), int(n), ptr(out, ctypes.c_float)))
        return out

    # ---- search ----
    def search(self, queries, k: int) -> tuple[np.ndarray, np.ndarray]:
        """Exact top-k.  Returns (distance float64 [B,k], rows int64 [B,k]); pads with NaN / -1.

        distance is pgvector's cosine distance (or negative inner product), ordered
        (distance asc, NaN last, row asc) -- the same contract the CPU checker in oracle/ states.
        """
        q = f32c(queries)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError(f"queries must be [B, {self.dim}], got {q.shape}")
        B = q.shape[0]
        dist = np.empty((B, k), dtype=np.float64)
        rows = np.empty((B, k), dtype=np.int64)
        check(self._h, self._lib.mi355dr_search(self._h, ptr(q, ctypes.c_float), B, int(k),
                                                ptr(dist, ctypes.c_double), ptr(rows, ctypes.c_int64)))
        return dist, rows

    def search_device(self, q_ptr: int, B: int, k: int, out_dist_ptr: int, out_rows_ptr: int,
                      stream: int | None = None) -> None:
        check(self._h, self._lib.mi355dr_search_device(self._h, ctypes.c_void_p(int(q_ptr)), int(B), int(k),
                                                       ctypes.c_void_p(int(out_dist_ptr)),
                                                       ctypes.c_void_p(int(out_rows_ptr)),
                                                       ctypes.c_void_p(int(stream) if stream else None)))

    def search_device_async(self, q_ptr: int, B: int, k: int, out_dist_ptr: int, out_rows_ptr: int,
                            stream: int | None = None) -> int:
        """Enqueue the search and return a ticket without synchronising (mi355dr_search_device_async): the outputs are
        valid after `search_wait(ticket)`; the query and output buffers must stay untouched until then."""
        t = ctypes.c_int64(0)
        check(self._h, self._lib.mi355dr_search_device_async(self._h, ctypes.c_void_p(int(q_ptr)), int(B), int(k),
                                                             ctypes.c_void_p(int(out_dist_ptr)),
                                                             ctypes.c_void_p(int(out_rows_ptr)),
                                                             ctypes.c_void_p(int(stream) if stream else None),
                                                             ctypes.byref(t)))
        return int(t.value)

    def search_wait(self, ticket: int) -> None:
        """Complete every block up to `ticket` (waits, then recomputes the rare queries the screen flagged)."""
        check(self._h, self._lib.mi355dr_search_wait(self._h, int(ticket)))

    def merge_topk_device(self, dist_all_ptr: int, rows_all_ptr: int, world: int, B: int, k: int, out_dist_ptr: int,
                          out_rows_ptr: int, stream: int | None = None) -> None:
        check(self._h, self._lib.mi355dr_merge_topk_device(
            self._h, ctypes.c_void_p(int(dist_all_ptr)), ctypes.c_void_p(int(rows_all_ptr)), int(world), int(B),
            int(k), ctypes.c_void_p(int(out_dist_ptr)), ctypes.c_void_p(int(out_rows_ptr)),
            ctypes.c_void_p(int(stream) if stream else None)))

    def pack_topk_device(self, dist_ptr: int, rows_ptr: int, B: int, k: int, packed_ptr: int, stream: int | None = None) -> None:
        check(self._h, self._lib.mi355dr_pack_topk_device(self._h, ctypes.c_void_p(int(dist_ptr)), ctypes.c_void_p(int(rows_ptr)),
                                                          int(B), int(k), ctypes.c_void_p(int(packed_ptr)),
                                                          ctypes.c_void_p(int(stream) if stream else None)))

    def merge_topk_packed_device(self, packed_all_ptr: int, world: int, B: int, k: int, out_dist_ptr: int, out_rows_ptr: int,
                                 stream: int | None = None) -> None:
        check(self._h, self._lib.mi355dr_merge_topk_packed_device(
            self._h, ctypes.c_void_p(int(packed_all_ptr)), int(world), int(B), int(k), ctypes.c_void_p(int(out_dist_ptr)),
            ctypes.c_void_p(int(out_rows_ptr)), ctypes.c_void_p(int(stream) if stream else None)))

    # ---- row-sharded search inside the library (RCCL, no torch): include/mi355dr.h "row-sharded search" ----
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte ncclUniqueId (call on ONE rank, hand it to every rank through the host's own channel)."""
        from ._native import load

        buf = ctypes.create_string_buffer(128)
        check(None, load().mi355dr_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p), 128))
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes) -> None:
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self._h, self._lib.mi355dr_comm_init(self._h, int(rank), int(world), ctypes.cast(buf, ctypes.c_void_p), 128))

    def comm_init_custom(self, rank: int, world: int, all_gather) -> None:
        """The sharded search over the HOST's own all-gather instead of RCCL (mi355dr_comm_init_custom).
        `all_gather(send_ptr, recv_ptr, bytes_per_rank, stream_handle) -> None` moves device memory: every rank's
        `bytes_per_rank` bytes at `send_ptr` into `recv_ptr`, rank-major, complete on return (or ordered on the stream).
        An exception inside it becomes error code 1 of the search that called it."""
        from ._native import ALLGATHER_FN

        def thunk(send, recv, nbytes, stream, _user):
            try:
                all_gather(int(send or 0), int(recv or 0), int(nbytes), int(stream or 0))
            except Exception:  # noqa: BLE001 - must not unwind through the C frame
                import traceback

                traceback.print_exc()
                return 1
            return 0

        self._allgather_thunk = ALLGATHER_FN(thunk)   # (kept alive as long as the index may call it)
        check(self._h, self._lib.mi355dr_comm_init_custom(self._h, int(rank), int(world), self._allgather_thunk, None))

    def comm_world(self) -> int:
        return int(self._lib.mi355dr_comm_world(self._h))

    def comm_count(self) -> int:
        """Ranks RCCL itself reports for this index's communicator (ncclCommCount); 0 before comm_init."""
        n = ctypes.c_int(0)
        check(self._h, self._lib.mi355dr_comm_count(self._h, ctypes.byref(n)))
        return int(n.value)

    def search_sharded_device(self, q_ptr: int, B: int, k: int, out_dist_ptr: int, out_rows_ptr: int,
                              stream: int | None = None) -> None:
        """Local search + ONE ncclAllGather + merge, on device buffers (every rank: same queries in, same result out)."""
        check(self._h, self._lib.mi355dr_search_sharded_device(
            self._h, ctypes.c_void_p(int(q_ptr)), int(B), int(k), ctypes.c_void_p(int(out_dist_ptr)),
            ctypes.c_void_p(int(out_rows_ptr)), ctypes.c_void_p(int(stream) if stream else None)))

    # ---- multi-vector ----
    def add_multivec(self, vecs, offsets) -> None:
        vecs = f32c(vecs)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if vecs.ndim != 2 or vecs.shape[1] != self.dim:
            raise ValueError(f"vecs must be [sum_T, {self.dim}]")
        if offsets.ndim != 1 or offsets.shape[0] < 1 or offsets[0] != 0 or offsets[-1] != vecs.shape[0]:
            raise ValueError("offsets must start at 0 and end at vecs.shape[0]")
        check(self._h, self._lib.mi355dr_add_multivec(self._h, ptr(vecs, ctypes.c_float),
                                                      ptr(offsets, ctypes.c_int64), offsets.shape[0] - 1))

    def add_multivec_device(self, vecs_ptr: int, offsets) -> None:
        """Docs whose token / patch vectors already sit in device memory ([sum_T, dim] fp32 at `vecs_ptr`, e.g. an
        encoder's output tensor): the padded store, its bf16 fragment copy and the screen's bound quantities are built
        by a kernel, nothing is copied to the host.  `offsets` is a host array [n_docs + 1]."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if offsets.ndim != 1 or offsets.shape[0] < 1 or offsets[0] != 0:
            raise ValueError("offsets must start at 0")
        check(self._h, self._lib.mi355dr_add_multivec_device(self._h, ctypes.c_void_p(int(vecs_ptr)),
                                                             ptr(offsets, ctypes.c_int64), offsets.shape[0] - 1))

    def n_docs(self) -> int:
        return int(self._lib.mi355dr_size_multivec(self._h))

    def search_maxsim(self, qtok, q_offsets, k: int) -> tuple[np.ndarray, np.ndarray]:
        """MaxSim top-k.  Returns (distance float32 [B,k] = -sum_i max_j <q_i,d_j>, doc rows int64 [B,k])."""
        qtok = f32c(qtok)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int32)
        B = q_offsets.shape[0] - 1
        dist = np.empty((B, k), dtype=np.float32)
        rows = np.empty((B, k), dtype=np.int64)
        check(self._h, self._lib.mi355dr_search_maxsim(self._h, ptr(qtok, ctypes.c_float),
                                                       ptr(q_offsets, ctypes.c_int32), B, int(k),
                                                       ptr(dist, ctypes.c_float), ptr(rows, ctypes.c_int64)))
        return dist, rows

    def search_maxsim_device(self, qtok_ptr: int, q_offsets, k: int, out_dist_ptr: int, out_rows_ptr: int,
                             stream: int | None = None) -> None:
        """MaxSim top-k with the query vectors ([sum_nq, dim] fp32 at `qtok_ptr`) and the results (fp32 [B,k] at
        `out_dist_ptr`, int64 [B,k] at `out_rows_ptr`) in device memory; `q_offsets` is a host array [B+1]."""
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int32)
        check(self._h, self._lib.mi355dr_search_maxsim_device(
            self._h, ctypes.c_void_p(int(qtok_ptr)), ptr(q_offsets, ctypes.c_int32), q_offsets.shape[0] - 1, int(k),
            ctypes.c_void_p(int(out_dist_ptr)), ctypes.c_void_p(int(out_rows_ptr)),
            ctypes.c_void_p(int(stream) if stream else None)))

    def maxsim_subset(self, qtok, q_offsets, doc_ids, clamp0: bool = False) -> np.ndarray:
        """Exact MaxSim distance of each query to its own list of docs: doc_ids [B, m] -> distances [B, m] (NaN = skipped).
        `clamp0`: every query vector contributes max(0, max_j <q_i, d_j>) (the ColBERT reranker's MaxSim)."""
        qtok = f32c(qtok).reshape(-1, self.dim)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int32)
        ids = np.ascontiguousarray(doc_ids, dtype=np.int64)
        B = q_offsets.shape[0] - 1
        if ids.ndim != 2 or ids.shape[0] != B:
            raise ValueError("doc_ids must be [B, m]")
        out = np.empty(ids.shape, dtype=np.float32)
        check(self._h, self._lib.mi355dr_maxsim_subset_ex(self._h, ptr(qtok, ctypes.c_float), ptr(q_offsets, ctypes.c_int32),
                                                          B, ptr(ids, ctypes.c_int64), ids.shape[1], 1 if clamp0 else 0,
                                                          ptr(out, ctypes.c_float)))
        return out

    # ---- Guided Query Refinement of candidate pools (reference gqr_hybrid.py:306-362) ----
    @staticmethod
    def _gqr_pool(pool, comp) -> tuple[np.ndarray, np.ndarray]:
        pool = np.ascontiguousarray(pool, dtype=np.int64)
        comp = np.ascontiguousarray(comp, dtype=np.float64)
        if pool.ndim != 2 or comp.shape != pool.shape:
            raise ValueError("candidate pool and complementary distribution must both be [B, P]")
        return pool, comp

    def gqr_refine(self, queries, cand_rows, comp_dist, n_steps: int, learning_rate: float, temperature: float,
                   mixture_alpha: float) -> np.ndarray:
        """Refine each query embedding against its candidate rows; returns the refined cosine scores float64 [B, P]
        (NaN at the -1 padding of a pool)."""
        pool, comp = self._gqr_pool(cand_rows, comp_dist)
        q = np.ascontiguousarray(queries, dtype=np.float64).reshape(pool.shape[0], self.dim)
        out = np.empty(pool.shape, dtype=np.float64)
        check(self._h, self._lib.mi355dr_gqr_refine(
            self._h, ptr(q, ctypes.c_double), pool.shape[0], ptr(pool, ctypes.c_int64), pool.shape[1],
            ptr(comp, ctypes.c_double), int(n_steps), float(learning_rate), float(temperature), float(mixture_alpha),
            ptr(out, ctypes.c_double)))
        return out

    def gqr_refine_maxsim(self, qtok, q_offsets, doc_ids, comp_dist, n_steps: int, learning_rate: float,
                          temperature: float, mixture_alpha: float) -> np.ndarray:
        """Multi-vector form: refine each query matrix against its candidate docs; refined mean-of-max scores [B, P]."""
        pool, comp = self._gqr_pool(doc_ids, comp_dist)
        q = np.ascontiguousarray(qtok, dtype=np.float64).reshape(-1, self.dim)
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.int32)
        if q_offsets.shape[0] != pool.shape[0] + 1 or int(q_offsets[-1]) != q.shape[0]:
            raise ValueError("q_offsets must be [B+1] and end at the number of query vectors")
        out = np.empty(pool.shape, dtype=np.float64)
        check(self._h, self._lib.mi355dr_gqr_refine_maxsim(
            self._h, ptr(q, ctypes.c_double), ptr(q_offsets, ctypes.c_int32), pool.shape[0], ptr(pool, ctypes.c_int64),
            pool.shape[1], ptr(comp, ctypes.c_double), int(n_steps), float(learning_rate), float(temperature),
            float(mixture_alpha), ptr(out, ctypes.c_double)))
        return out

    def gqr_refine_scores(self, primary_scores, counts, comp_dist, n_steps: int, learning_rate: float,
                          temperature: float, mixture_alpha: float) -> np.ndarray:
        """Score-space form (no vectors): the primary scores themselves are refined; [B, P], counts[b] live entries."""
        z = np.ascontiguousarray(primary_scores, dtype=np.float64)
        comp = np.ascontiguousarray(comp_dist, dtype=np.float64)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        if z.ndim != 2 or comp.shape != z.shape or counts.shape != (z.shape[0],):
            raise ValueError("primary_scores / comp_dist must be [B, P] and counts [B]")
        out = np.empty(z.shape, dtype=np.float64)
        check(self._h, self._lib.mi355dr_gqr_refine_scores(
            self._h, ptr(z, ctypes.c_double), ptr(counts, ctypes.c_int32), z.shape[0], z.shape[1],
            ptr(comp, ctypes.c_double), int(n_steps), float(learning_rate), float(temperature), float(mixture_alpha),
            ptr(out, ctypes.c_double)))
        return out

    # ---- options / stats / timing ----
    def set_option(self, key: str, value: int | str) -> None:
        if key == "path" and isinstance(value, str):
            value = PATHS[value]
        if key == "screen_dtype" and isinstance(value, str):
            value = SCREEN_DTYPES[value]
        check(self._h, self._lib.mi355dr_set_option(self._h, key.encode(), int(value)))

    def stat(self, key: str) -> int:
        out = ctypes.c_int64(0)
        check(self._h, self._lib.mi355dr_get_stat(self._h, key.encode(), ctypes.byref(out)))
        return int(out.value)

    def reset_stats(self) -> None:
        check(self._h, self._lib.mi355dr_reset_stats(self._h))

    def timer_start(self) -> None:
        check(self._h, self._lib.mi355dr_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = ctypes.c_double(0.0)
        check(self._h, self._lib.mi355dr_timer_stop(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def synchronize(self) -> None:
        check(self._h, self._lib.mi355dr_synchronize(self._h))

    # ---- raw device buffers (tests / bench without torch) ----
    def dev_alloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        check(self._h, self._lib.mi355dr_dev_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return int(p.value)

    def dev_free(self, p: int) -> None:
        check(self._h, self._lib.mi355dr_dev_free(self._h, ctypes.c_void_p(int(p))))

    def dev_upload(self, dst: int, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr)
        check(self._h, self._lib.mi355dr_dev_upload(self._h, ctypes.c_void_p(int(dst)),
                                                    ctypes.c_void_p(arr.ctypes.data), arr.nbytes))

    def dev_download(self, src: int, arr: np.ndarray) -> None:
        assert arr.flags["C_CONTIGUOUS"]
        check(self._h, self._lib.mi355dr_dev_download(self._h, ctypes.c_void_p(arr.ctypes.data),
                                                      ctypes.c_void_p(int(src)), arr.nbytes))

    # ---- test hooks ----
    def debug_screen_dense(self, queries, row0: int, n: int) -> np.ndarray:
        q = f32c(queries)
        out = np.empty((q.shape[0], n), dtype=np.float32)
        check(self._h, self._lib.mi355dr_debug_screen_dense(self._h, ptr(q, ctypes.c_float), q.shape[0], int(row0),
                                                            int(n), ptr(out, ctypes.c_float)))
        return out

    def debug_i8_state(self, queries, g0: int, n_groups: int) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """int8 screen: (S_q [B], kq [B], S_g [n_groups], e_g [n_groups]) -- screen value = S_q S_g (q8.c8) + e_g kq."""
        q = f32c(queries)
        sq = np.empty(q.shape[0], dtype=np.float32)
        kq = np.empty(q.shape[0], dtype=np.float32)
        st = np.empty(n_groups, dtype=np.float32)
        er = np.empty(n_groups, dtype=np.float32)
        check(self._h, self._lib.mi355dr_debug_i8_state(self._h, ptr(q, ctypes.c_float), q.shape[0], ptr(sq, ctypes.c_float),
                                                        ptr(kq, ctypes.c_float), int(g0), int(n_groups),
                                                        ptr(st, ctypes.c_float), ptr(er, ctypes.c_float)))
        return sq, kq, st, er

    def debug_screen_bound(self, queries) -> np.ndarray:
        """Per-query bound E of the active screen dtype: exact cosine <= screen value + E."""
        q = f32c(queries)
        out = np.empty(q.shape[0], dtype=np.float32)
        check(self._h, self._lib.mi355dr_debug_screen_bound(self._h, ptr(q, ctypes.c_float), q.shape[0],
                                                            ptr(out, ctypes.c_float)))
        return out

    def debug_rescore(self, queries, pair_q, pair_row) -> tuple[np.ndarray, np.ndarray]:
        q = f32c(queries)
        pq = np.ascontiguousarray(pair_q, dtype=np.int32)
        pr = np.ascontiguousarray(pair_row, dtype=np.int64)
        dot = np.empty(pq.shape[0], dtype=np.float32)
        dist = np.empty(pq.shape[0], dtype=np.float64)
        check(self._h, self._lib.mi355dr_debug_rescore(self._h, ptr(q, ctypes.c_float), q.shape[0],
                                                       ptr(pq, ctypes.c_int32), ptr(pr, ctypes.c_int64), pq.shape[0],
                                                       ptr(dot, ctypes.c_float), ptr(dist, ctypes.c_double)))
        return dot, dist
