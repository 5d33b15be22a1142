"""Randomised differential test: libmi355dr (through the Python binding) vs the CPU oracle, bit for bit.

Runs random shapes / data modes / option settings for a wall-clock budget and stops at the first mismatch, printing the
seed of the failing case.  Test infrastructure (it uses oracle/), meant for a GPU box:
    python tools/fuzz_parity.py --seconds 300 [--seed 1] [--only single|maxsim|session|gqr]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import autorag_research_amd as pkg  # noqa: E402
from oracle import cpu_ref  # noqa: E402


import os  # noqa: E402

IP_PROB = float(os.environ.get("FUZZ_IP_PROB", "0.2"))  # share of inner-product cases among the single-vector ones


def corpus(rng, n, d, mode):
    C = rng.standard_normal((n, d)).astype(np.float32)
    if mode == "scaled":
        C *= np.exp(rng.uniform(-6, 6, size=(n, 1))).astype(np.float32)
    elif mode == "clustered":
        c = rng.standard_normal((max(1, n // 200), d)).astype(np.float32)
        C = c[rng.integers(0, c.shape[0], size=n)] + (0.02 * rng.standard_normal((n, d))).astype(np.float32)
    elif mode == "dups":
        base = C[: max(1, n // 7)]
        C = base[rng.integers(0, base.shape[0], size=n)].copy()
    elif mode == "spiky":
        idx = rng.choice(n, size=max(1, n // 50), replace=False)
        C[idx, rng.integers(0, d, size=idx.size)] += 30.0
    elif mode == "dirty":
        for v in (0.0, np.nan, np.inf, 1e-25, 1e25):
            C[rng.integers(0, n)] = v if v == 0.0 else C[rng.integers(0, n)] * 0 + v
    return C


def check_single(rng, case):
    n = int(rng.choice([1, 7, 300, 4000, 30000, 120000, 250000]))
    n = max(1, int(n * rng.uniform(0.5, 1.0)))
    d = int(rng.choice([5, 16, 64, 100, 128, 256, 384, 768, 1000]))
    Bmax = max(1, int(4e10 / (n * d)))
    B = int(min(Bmax, rng.choice([1, 3, 33, 128, 129, 300, 777, 1024, 1500])))
    k = int(rng.choice([1, 3, 10, 24, 25, 33, 50, 64, 100, 128, 129, 300, 1024]))  # (33 ... 128: the two-wave prune)
    mode = str(rng.choice(["gauss", "scaled", "clustered", "dups", "spiky", "dirty"]))
    metric = "ip" if rng.random() < IP_PROB else "cosine"
    C = corpus(rng, n, d, mode)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    if mode in ("clustered", "dups") and B > 2:
        Q[: B // 2] = C[rng.integers(0, n, size=B // 2)] + (0.01 * rng.standard_normal((B // 2, d))).astype(np.float32)
    opts = {}
    if rng.random() < 0.5:
        opts["screen_dtype"] = str(rng.choice(["auto", "bf16", "i8"]))
    if rng.random() < 0.2:
        opts["cand_cap"] = int(rng.choice([64, 300, 1024]))
    if rng.random() < 0.2:
        opts["chunk_growth"] = int(rng.choice([1, 2, 5, 7]))
    if rng.random() < 0.2:
        opts["chunk0_rows"] = int(rng.choice([256, 512, 2048]))
    if rng.random() < 0.25:
        opts["prefilter16"] = 1
    if rng.random() < 0.15:
        opts["starter"] = 0
    if rng.random() < 0.15:
        opts["prune_companion"] = 1
    if rng.random() < 0.15:
        opts["defer_round_b"] = 0
    if rng.random() < 0.15:
        opts["prune_wide"] = 0
    row_offset = int(rng.choice([0, 0, 12345, 2**33]))
    desc = f"single n={n} d={d} B={B} k={k} mode={mode} metric={metric} opts={opts} row_offset={row_offset}"
    import hashlib
    h_in = hashlib.sha1(C.tobytes() + Q.tobytes()).hexdigest()
    with pkg.Mi355Index(d, metric) as idx:
        for key, val in opts.items():
            idx.set_option(key, val)
        idx.set_option("row_offset", row_offset)
        cut = int(rng.integers(0, n + 1))
        if cut:
            idx.add(C[:cut])
        if cut < n:
            idx.add(C[cut:])
        try:
            dist, rows = idx.search(Q, k)
        except pkg.NativeError as e:
            if opts.get("screen_dtype") == "i8" and "int8 screen unavailable" in str(e):
                return desc + " (i8 unavailable: skipped)"
            raise
        stats = {s: idx.stat(s) for s in ("fallback_queries", "retry_queries", "loose_rows", "screen_dtype_active")}
    rd, rr = cpu_ref.topk_search(C, Q, k, metric=metric)
    rr = np.where(rr >= 0, rr + row_offset, rr)
    ok = np.array_equal(rows, rr) and np.array_equal(np.isnan(dist), np.isnan(rd))
    m = ~np.isnan(dist)
    ok = ok and np.array_equal(dist[m].view(np.uint64), rd[m].view(np.uint64))
    if not ok:
        bad = np.argwhere((rows != rr) | ((dist != rd) & ~(np.isnan(dist) & np.isnan(rd))))
        q0 = int(bad[0][0]) if len(bad) else 0
        # who moved?  the inputs (host memory), the oracle (threads), or the GPU result
        h_now = hashlib.sha1(C.tobytes() + Q.tobytes()).hexdigest()
        rd1, rr1 = cpu_ref.topk_search(C, Q, k, metric=metric, threads=1)
        rd2, rr2 = cpu_ref.topk_search(C, Q, k, metric=metric)
        rr1 = np.where(rr1 >= 0, rr1 + row_offset, rr1)
        rr2 = np.where(rr2 >= 0, rr2 + row_offset, rr2)
        with pkg.Mi355Index(d, metric) as idx2:
            idx2.set_option("row_offset", row_offset)
            idx2.add(C)
            _, rows2 = idx2.search(Q, k)
        desc += (f" [inputs changed: {h_in != h_now}; oracle(1 thread) == oracle: {np.array_equal(rr1, rr)}; oracle again == oracle: "
                 f"{np.array_equal(rr2, rr)}; gpu == oracle(1 thread): {np.array_equal(rows, rr1)}; gpu again == gpu: {np.array_equal(rows2, rows)}; oracle fp-env resets: {cpu_ref.fpenv_fix_count()}; omp team {cpu_ref.debug_team()}; omp libs "
                 f"{sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'omp' in l.split()[-1]))}]")
        raise AssertionError(f"MISMATCH {desc} stats={stats} first bad (query,slot)={bad[:3].tolist()} "
                             f"gpu rows {rows[q0][:8].tolist()} dist {dist[q0][:8].tolist()} | "
                             f"ref rows {rr[q0][:8].tolist()} dist {rd[q0][:8].tolist()}")
    return desc + f" stats={stats}"


def check_session(rng, case):
    """one index, several operations: searches with changing k / B / options interleaved with appends (state carried
    across calls: thresholds, the int8 demotion flag, retry buffers, reallocation of every shadow)"""
    d = int(rng.choice([16, 64, 128, 384, 768]))
    mode = str(rng.choice(["gauss", "clustered", "dups", "spiky", "dirty"]))
    big = rng.random() < 0.25
    n0 = int(rng.choice([500_000, 1_000_000]) if big else rng.choice([50, 3000, 40000]))
    if big:
        d = min(d, 384)
    C = corpus(rng, n0, d, mode)
    metric = "cosine"
    desc = f"session d={d} mode={mode} n0={n0}"
    with pkg.Mi355Index(d, metric) as idx:
        idx.add(C)
        for step in range(int(rng.integers(3, 7))):
            op = rng.random()
            if op < 0.25 and not big:
                more = corpus(rng, int(rng.integers(1, 5000)), d, str(rng.choice(["gauss", "clustered", "dups"])))
                idx.add(more)
                C = np.concatenate([C, more], axis=0)
                desc += f" | add {more.shape[0]}"
                continue
            if op < 0.4:
                key, val = [("screen_dtype", str(rng.choice(["auto", "bf16"]))), ("path", "auto"),
                            ("cand_cap", int(rng.choice([128, 2048]))), ("chunk_growth", int(rng.choice([2, 3, 6])))][int(rng.integers(0, 4))]
                idx.set_option(key, val)
                desc += f" | {key}={val}"
                continue
            B = int(rng.choice([1, 2, 8])) if big else int(rng.choice([1, 5, 64, 130, 400, 1100]))
            k = int(rng.choice([1, 10, 24, 30, 40, 100, 128]))
            Q = rng.standard_normal((B, d)).astype(np.float32)
            if mode in ("clustered", "dups"):
                Q[: max(1, B // 2)] = C[rng.integers(0, C.shape[0], size=max(1, B // 2))] + (
                    0.01 * rng.standard_normal((max(1, B // 2), d))).astype(np.float32)
            desc += f" | search B={B} k={k}"
            dist, rows = idx.search(Q, k)
            rd, rr = cpu_ref.topk_search(C, Q, k, metric=metric)
            ok = np.array_equal(rows, rr) and np.array_equal(np.isnan(dist), np.isnan(rd))
            m = ~np.isnan(dist)
            ok = ok and np.array_equal(dist[m].view(np.uint64), rd[m].view(np.uint64))
            if not ok:
                st = {s: idx.stat(s) for s in ("fallback_queries", "retry_queries", "loose_rows", "screen_dtype_active", "i8_demoted")}
                raise AssertionError(f"MISMATCH {desc} stats={st} first bad={np.argwhere(rows != rr)[:3].tolist()}")
    return desc


def check_maxsim(rng, case):
    d = int(rng.choice([8, 20, 64, 96, 128, 200]))
    n_docs = int(rng.choice([1, 40, 700, 5000, 30000]))
    tmax = int(rng.choice([3, 40, 180, 1100])) if n_docs <= 700 else int(rng.choice([3, 40]))
    lens = rng.integers(0 if rng.random() < 0.3 else 1, tmax + 1, size=n_docs)
    unit = rng.random() < 0.6
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    if unit and tok.shape[0]:
        tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    mode = str(rng.choice(["plain", "dups", "dirty"]))
    if mode == "dups" and tok.shape[0] > 10:
        tok[:] = tok[rng.integers(0, max(1, tok.shape[0] // 9), size=tok.shape[0])]
    if mode == "dirty" and tok.shape[0] > 3:
        tok[rng.integers(0, tok.shape[0])] = np.inf
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nq = int(rng.choice([1, 2, 5, 9, 17, 40]))   # (round 4: up to 16 queries ride one screen pass)
    qlens = [int(x) for x in rng.choice([0, 1, 5, 24, 32, 33, 100, 128] if nq <= 9 else [0, 7, 24, 24, 32, 32, 32, 33, 100, 200], size=nq)]
    if rng.random() < 0.25:  # ColBERT-shaped batch: every query one column block of 32 vectors (the last one may be shorter)
        qlens = [32] * nq
        qlens[-1] = int(rng.integers(1, 33))
    qtok = rng.standard_normal((sum(qlens), d)).astype(np.float32)
    if unit and qtok.shape[0]:
        qtok /= np.linalg.norm(qtok, axis=1, keepdims=True)
    qoff = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
    k = int(rng.choice([1, 10, 64, 65, 200]))
    screen = int(rng.random() < 0.8)
    desc = f"maxsim d={d} docs={n_docs} tmax={tmax} qlens={qlens} k={k} unit={unit} mode={mode} screen={screen}"
    if tok.shape[0] == 0:
        return desc + " (no tokens: skipped)"
    with pkg.Mi355Index(d) as idx:
        idx.set_option("maxsim_screen", screen)
        idx.set_option("maxsim_coop", int(rng.integers(-1, 2)))
        groups, wg = int(rng.integers(1, 5)), int(rng.integers(-1, 3))
        idx.set_option("maxsim_pass_groups", groups)
        idx.set_option("maxsim_wg", wg)
        bps = int(rng.choice([2, 4]))
        idx.set_option("maxsim_wg_bps", bps)
        idx.set_option("maxsim_wg_pipe", int(rng.random() < 0.7))
        idx.set_option("maxsim_wg_min", int(rng.integers(8, 10)))
        idx.set_option("maxsim_aligned", int(rng.random() < 0.8))
        # (round 4-5 drew the token-packed copy's switch here; round 6: the granule-packed copy -- when it pays / always)
        idx.set_option("maxsim_pack8", 1 if int(rng.integers(0, 2)) else -1)
        tighten = int(rng.random() < 0.7)
        idx.set_option("maxsim_tighten", tighten)
        desc += f" groups={groups} wg={wg} bps={bps} tighten={tighten}"
        idx.add_multivec(tok, off)
        dist, rows = idx.search_maxsim(qtok, qoff, k)
        stats = {s: idx.stat(s) for s in ("maxsim_screened", "maxsim_fallbacks")}
        # candidate re-scoring entry point on a random list per query (incl. ids that are not docs)
        m_ids = int(rng.integers(1, 40))
        ids = rng.integers(-2, n_docs + 3, size=(nq, m_ids)).astype(np.int64)
        sub = idx.maxsim_subset(qtok, qoff, ids)
    for b in range(nq):
        qb = qtok[qoff[b]:qoff[b + 1]]
        for j in range(m_ids):
            i = int(ids[b, j])
            if 0 <= i < n_docs and off[i + 1] > off[i] and qb.shape[0]:
                exp = np.float32(cpu_ref.maxsim_distance(tok[off[i]:off[i + 1]], qb))
                same = (np.isnan(exp) and np.isnan(sub[b, j])) or sub[b, j].view(np.uint32) == exp.view(np.uint32)
                if not same:
                    raise AssertionError(f"SUBSET MISMATCH {desc} query {b} doc {i}: {sub[b, j]} vs {exp}")
            elif not np.isnan(sub[b, j]):
                raise AssertionError(f"SUBSET MISMATCH {desc} query {b} doc {i}: expected NaN, got {sub[b, j]}")
    rd, rr = cpu_ref.maxsim_topk(tok, off, qtok, qoff, k)
    live = [i for i, t in enumerate(qlens) if t > 0]
    ok = True
    for i in range(nq):
        if i in live:
            ok = ok and np.array_equal(rows[i], rr[i]) and np.array_equal(np.isnan(dist[i]), np.isnan(rd[i]))
            m = ~np.isnan(dist[i])
            ok = ok and np.array_equal(dist[i][m].view(np.uint32), rd[i][m].view(np.uint32))
        else:
            ok = ok and (rows[i] == -1).all()
    if not ok:
        raise AssertionError(f"MISMATCH {desc} stats={stats}")
    return desc + f" stats={stats}"


def check_gqr(rng, case):
    """Guided Query Refinement kernels vs oracle/gqr_ref.py (float64: agreement to rounding, not bit for bit)."""
    from oracle import gqr_ref

    n_steps = int(rng.integers(1, 41))
    lr = float(np.exp(rng.uniform(np.log(0.01), np.log(1.0))))
    temp = float(np.exp(rng.uniform(np.log(0.05), np.log(4.0))))
    alpha = float(rng.choice([0.0, 1.0, rng.random()]))
    prm = (n_steps, lr, temp, alpha)
    B = int(rng.choice([1, 3, 17, 64]))
    P = int(rng.choice([1, 2, 9, 40, 130, 400]))
    sizes = rng.integers(1, P + 1, size=B)
    sizes[0] = P
    comp = np.zeros((B, P))
    for b, m in enumerate(sizes):
        comp[b, :m] = rng.dirichlet(np.ones(m) * float(rng.choice([0.1, 1.0, 10.0])))
    form = str(rng.choice(["single", "multi", "scores"]))
    desc = f"gqr {form} B={B} P={P} steps={n_steps} lr={lr:.3g} T={temp:.3g} alpha={alpha:.3g}"
    tol = 1e-8
    if form == "scores":
        prim = rng.standard_normal((B, P)) * float(rng.choice([0.1, 1.0, 30.0]))
        with pkg.Mi355Index(8) as idx:
            got = idx.gqr_refine_scores(prim, sizes.astype(np.int32), comp, *prm)
        for b, m in enumerate(sizes):
            exp = gqr_ref.refine_scores(prim[b, :m], comp[b, :m], *prm)
            # conditioning of the case: the same oracle on inputs moved by one part in 1e15 and on reordered candidates.
            # With lr/T^2 >> 1 the iteration is chaotic (amplifications of 1e8..1e11 measured): any two float64
            # implementations then differ, and the kernel is held to that spread instead of a fixed tolerance.
            wiggle = 1.0 + 1e-15 * np.sign(rng.standard_normal(m))
            spread = float(np.abs(exp - gqr_ref.refine_scores(prim[b, :m] * wiggle, comp[b, :m], *prm)).max())
            perm = rng.permutation(m)
            alt = np.empty(m)
            alt[perm] = gqr_ref.refine_scores(prim[b, :m][perm], comp[b, :m][perm], *prm)
            spread = max(spread, float(np.abs(exp - alt).max()))
            lim = tol * max(1.0, np.abs(exp).max()) + 100.0 * spread
            if not (np.abs(got[b, :m] - exp).max() <= lim and np.isnan(got[b, m:]).all()):
                raise AssertionError(f"MISMATCH {desc} query {b}: {np.abs(got[b, :m] - exp).max()} > {lim}")
        return desc
    if form == "single":
        d = int(rng.choice([2, 7, 48, 384, 768, 1000]))
        n = int(rng.choice([P, 3 * P + 5, 5000]))
        C = corpus(rng, n, d, str(rng.choice(["plain", "scaled", "clustered", "dups"])))
        C[rng.integers(0, n)] = 0.0
        Q = rng.standard_normal((B, d)).astype(np.float32).astype(np.float64)
        if rng.random() < 0.2:
            Q[rng.integers(0, B)] = 0.0
        pools = np.full((B, P), -1, dtype=np.int64)
        for b, m in enumerate(sizes):
            pools[b, :m] = rng.integers(0, n, size=m)  # repeats allowed: a row may enter a pool twice
        off = int(rng.choice([0, 12345]))
        with pkg.Mi355Index(d) as idx:
            idx.add(C)
            idx.set_option("row_offset", off)
            got = idx.gqr_refine(Q, np.where(pools >= 0, pools + off, pools), comp, *prm)
        Cd = C.astype(np.float64)
        perm = rng.permutation(d)
        for b, m in enumerate(sizes):
            exp = gqr_ref.refine_single(Q[b], Cd[pools[b, :m]], comp[b, :m], *prm)
            # conditioning of the case itself: the same oracle on column-permuted data (identical mathematics, other
            # rounding).  A sharp softmax over 40 steps amplifies last-bit differences; the kernel is held to that noise.
            alt = gqr_ref.refine_single(Q[b][perm], Cd[pools[b, :m]][:, perm], comp[b, :m], *prm)
            alt2 = gqr_ref.refine_single(Q[b] * (1.0 + 1e-15 * np.sign(rng.standard_normal(d))), Cd[pools[b, :m]],
                                         comp[b, :m], *prm)
            lim = tol + 100.0 * max(np.abs(exp - alt).max(), np.abs(exp - alt2).max())
            if not (np.abs(got[b, :m] - exp).max() <= lim and np.isnan(got[b, m:]).all()):
                raise AssertionError(f"MISMATCH {desc} d={d} n={n} query {b}: {np.abs(got[b, :m] - exp).max()} > {lim}")
        return desc + f" d={d} n={n}"
    d = int(rng.choice([8, 20, 96, 128, 200]))
    n_docs = int(rng.choice([P, 2 * P + 3, 900]))
    lens = rng.integers(1, int(rng.choice([3, 40, 180])) + 1, size=n_docs)
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    if rng.random() < 0.7:
        tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    q_lens = rng.choice([1, 5, 16, 17, 32, 60], size=B)
    if (int(q_lens.max()) + 15) // 16 * 16 * (d + 9) * 8 + 3 * P * 8 > 150_000:  # LDS budget of one workgroup
        q_lens = np.minimum(q_lens, 32)
    qtok = rng.standard_normal((int(q_lens.sum()), d)).astype(np.float32).astype(np.float64)
    qoff = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    pools = np.full((B, P), -1, dtype=np.int64)
    for b, m in enumerate(sizes):
        pools[b, :m] = rng.integers(0, n_docs, size=m)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        got = idx.gqr_refine_maxsim(qtok, qoff, pools, comp, *prm)
    tokd = tok.astype(np.float64)
    for b in sorted(set([0, B - 1, int(rng.integers(0, B))])):  # the numpy loop is slow: three queries per case
        m = int(sizes[b])
        docs = [tokd[off[i]:off[i + 1]] for i in pools[b, :m]]
        exp = gqr_ref.refine_multi(qtok[qoff[b]:qoff[b + 1]], docs, comp[b, :m], *prm)
        perm = rng.permutation(d)
        alt = gqr_ref.refine_multi(qtok[qoff[b]:qoff[b + 1]][:, perm], [D[:, perm] for D in docs], comp[b, :m], *prm)
        qb = qtok[qoff[b]:qoff[b + 1]]
        alt2 = gqr_ref.refine_multi(qb * (1.0 + 1e-15 * np.sign(rng.standard_normal(qb.shape))), docs, comp[b, :m], *prm)
        err = np.abs(got[b, :m] - exp).max()
        lim = tol * max(1.0, np.abs(exp).max()) + 100.0 * max(np.abs(exp - alt).max(), np.abs(exp - alt2).max())
        # (an argmax flip mid-trajectory shows in the spread too)
        if not (err <= lim and np.isnan(got[b, m:]).all()):
            raise AssertionError(f"MISMATCH {desc} d={d} docs={n_docs} query {b}: {err} > {lim}")
    return desc + f" d={d} docs={n_docs}"


CHECKS = {"maxsim": check_maxsim, "single": check_single, "session": check_session, "gqr": check_gqr}


def pick_kind(u: float) -> str:
    return "maxsim" if u < 0.3 else "session" if u < 0.45 else "gqr" if u > 0.9 else "single"


def poison_device_memory(pattern: str, gib: int = 24) -> None:
    """Fill then free a large part of the device memory with a byte pattern, so that later allocations of this process
    come back dirty: buffers the library reads before writing show up as mismatches instead of passing on zero pages."""
    rng = np.random.default_rng(12345)
    chunk = 256 << 20
    if pattern == "random":
        img = rng.integers(0, 256, size=chunk, dtype=np.uint8)
    else:
        img = np.full(chunk, int(pattern, 16), dtype=np.uint8)
    with pkg.Mi355Index(8) as idx:
        ptrs = [idx.dev_alloc(chunk) for _ in range(gib * 4)]
        for p in ptrs:
            idx.dev_upload(p, img)
        for p in ptrs:
            idx.dev_free(p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", choices=["single", "maxsim", "session", "gqr"], default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--start", type=int, default=0, help="first case number (cases are seeded by number)")
    ap.add_argument("--poison", default=None, help="hex byte (e.g. ff, 7f) or 'random': dirty the device memory first")
    a = ap.parse_args()
    if a.poison:
        poison_device_memory(a.poison)
    t0, case, counts = time.time(), a.start, {"single": 0, "maxsim": 0, "session": 0, "gqr": 0}
    while time.time() - t0 < a.seconds:
        seed = a.seed * 1_000_003 + case
        rng = np.random.default_rng(seed)
        u = rng.random()
        kind = a.only or pick_kind(u)
        try:
            msg = CHECKS[kind](rng, case)
        except Exception as e:  # noqa: BLE001
            print(f"FAILED case {case} seed {seed} kind {kind}: {e}")
            sys.exit(1)
        counts[kind] += 1
        if a.verbose:
            print(f"ok {case} seed {seed}: {msg}", flush=True)
        case += 1
    print(f"fuzz ok: {case} cases in {time.time() - t0:.0f} s ({counts}), seed base {a.seed}; "
          f"oracle results rejected by the single-thread re-check: {cpu_ref.rejected_results}; "
          f"oracle threads found with a non-default fp environment: {cpu_ref.fpenv_fix_count()}; omp team "
          f"{cpu_ref.debug_team()}; omp libs {sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'omp' in l.split()[-1]))}")


if __name__ == "__main__":
    main()
