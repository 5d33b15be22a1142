#!/usr/bin/env python3
"""What a block-scaled FP8 / FP6 / FP4 pre-screen would have to admit -- CPU, numpy.

The int8 screen is exact because every pair's screen value is within a RIGOROUS bound of the exact cosine, built from MEASURED
residual norms (DESIGN "Screen bounds"): exact <= v + E with E ~ |e_q| + |e_c| by Cauchy-Schwarz.  The same construction for the
OCP microscaling formats of v_mfma_scale_f32_32x32x64_f8f6f4 (one power-of-two scale per 32 consecutive k-values, elements e4m3 /
e2m3 / e2m1): quantise a sample of the headline corpus (L2-normalised N(0,1) rows, d = 768) and a block of queries, measure the
residual norms, form the bound, and count the rows a query would have to re-score at N = 10 M, k = 10 -- measured on the sample
against the N = 10 M threshold, and from the Gaussian tail.  Also reported: the ACTUAL error (so the slack of the bound is
visible), and the same for the int8 scheme the library runs (one step per group of 32 rows) and for blockwise Cauchy-Schwarz over
the six 128-wide K slices.
usage: python tools/mx_bound_probe.py [rows=1000000] [queries=256]
"""
import math
import sys

import numpy as np

D = 768
E2M3 = np.array([0, .125, .25, .375, .5, .625, .75, .875, 1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,
                 2, 2.25, 2.5, 2.75, 3, 3.25, 3.5, 3.75, 4, 4.5, 5, 5.5, 6, 6.5, 7, 7.5], dtype=np.float32)
E2M1 = np.array([0, .5, 1, 1.5, 2, 3, 4, 6], dtype=np.float32)


def e4m3_grid():
    vals = [0.0]
    for f in range(1, 8):
        vals.append(f / 8 * 2.0 ** -6)
    for E in range(1, 16):
        for f in range(8):
            if E == 15 and f == 7:
                continue
            vals.append((1 + f / 8) * 2.0 ** (E - 7))
    return np.array(sorted(vals), dtype=np.float32)


E4M3 = e4m3_grid()


def round_to_grid(a: np.ndarray, grid: np.ndarray) -> np.ndarray:
    """nearest grid value of |a| (ties to the lower neighbour: immaterial here), sign restored; saturating."""
    mag = np.abs(a)
    i = np.searchsorted(grid, mag)
    i = np.clip(i, 1, len(grid) - 1)
    lo, hi = grid[i - 1], grid[i]
    pick = np.where(mag - lo <= hi - mag, lo, hi)
    return np.sign(a) * np.minimum(pick, grid[-1])


def mx_quantise(x: np.ndarray, grid: np.ndarray, emax: int) -> np.ndarray:
    """OCP MX: blocks of 32 consecutive k-values share the scale 2^(floor(log2(max|x|)) - emax); elements rounded to `grid`."""
    n, d = x.shape
    b = x.reshape(n, d // 32, 32)
    peak = np.abs(b).max(axis=2, keepdims=True)
    e = np.floor(np.log2(np.maximum(peak, 1e-30))) - emax
    scale = np.exp2(e).astype(np.float32)
    return (round_to_grid(b / scale, grid) * scale).reshape(n, d)


def int8_quantise_rows(x: np.ndarray) -> np.ndarray:
    """the library's row shadow: one step per group of 32 ROWS = the group's largest component / 127"""
    n, d = x.shape
    g = x[: n // 32 * 32].reshape(n // 32, 32, d)
    step = np.abs(g).max(axis=(1, 2), keepdims=True) / 127.0
    return (np.rint(g / step) * step).reshape(-1, d)


def int8_quantise_queries(q: np.ndarray) -> np.ndarray:
    step = np.abs(q).max(axis=1, keepdims=True) / 127.0
    return np.rint(q / step) * step


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    N_FULL, K = 10_000_000, 10
    rng = np.random.default_rng(1234)
    sigma = 1 / math.sqrt(D)
    # the k-th best cosine of N_FULL Gaussian-direction rows: z_k with N * P(z > z_k) = k
    from scipy.stats import norm

    z_k = norm.isf(K / N_FULL)
    tau = z_k * sigma
    print(f"corpus model: unit rows of d = {D} (cosines ~ N(0, {sigma:.4f}^2)); N = {N_FULL}, k = {K}: k-th best cosine tau = {tau:.4f} "
          f"(z = {z_k:.2f}); sample: {n} rows x {nq} queries\n")
    Q = rng.standard_normal((nq, D)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    schemes = [("int8 (library: step per 32 rows / per query)", None, None), ("MX fp8 e4m3", E4M3, 8), ("MX fp6 e2m3", E2M3, 2),
               ("MX fp4 e2m1", E2M1, 2)]
    res = {name: dict(ec=0.0, ec_mean=[], cand=0, err_max=0.0, err_rms=[], bw=[]) for name, *_ in schemes}
    Qh = {}
    for name, grid, emax in schemes:
        Qh[name] = int8_quantise_queries(Q) if grid is None else mx_quantise(Q, grid, emax)
    CH = 50_000
    for r0 in range(0, n, CH):
        C = rng.standard_normal((min(CH, n - r0), D)).astype(np.float32)
        C /= np.linalg.norm(C, axis=1, keepdims=True)
        C = C[: C.shape[0] // 32 * 32]
        exact = Q @ C.T
        for name, grid, emax in schemes:
            Ch = int8_quantise_rows(C) if grid is None else mx_quantise(C, grid, emax)
            ec = np.linalg.norm(C - Ch, axis=1)
            eq = np.linalg.norm(Q - Qh[name], axis=1)
            r = res[name]
            r["ec"] = max(r["ec"], float(ec.max()))
            r["ec_mean"].append(float(ec.mean()))
            t = Qh[name] @ Ch.T
            err = t - exact
            r["err_max"] = max(r["err_max"], float(np.abs(err).max()))
            r["err_rms"].append(float(np.sqrt((err ** 2).mean())))
            # the bound of a pair, per-row residual: |e_q||c_hat| + |q||e_c| (unit vectors; second-order term dropped: < 1 %)
            bound = eq[:, None] * np.linalg.norm(Ch, axis=1)[None, :] + ec[None, :]
            # candidates: rows the screen cannot rule out against the final threshold: t + bound >= tau
            r["cand"] += int((t + bound >= tau).sum())
            # blockwise Cauchy-Schwarz over the six 128-wide K slices: sum_s |e_c,s||q_s| + |e_q,s||c_hat,s|
            if r0 == 0:
                ecs = np.linalg.norm((C - Ch).reshape(-1, 6, 128), axis=2)          # [rows, 6]
                qs = np.linalg.norm(Q.reshape(nq, 6, 128), axis=2)                  # [nq, 6]
                eqs = np.linalg.norm((Q - Qh[name]).reshape(nq, 6, 128), axis=2)
                chs = np.linalg.norm(Ch.reshape(-1, 6, 128), axis=2)
                bw = qs[:64] @ ecs[:4096].T + eqs[:64] @ chs[:4096].T
                full = np.linalg.norm(Q[:64], axis=1)[:, None] * ec[None, :4096] + eq[:64, None] * np.linalg.norm(Ch[:4096], axis=1)[None, :]
                r["bw"] = [float(bw.mean()), float(full.mean())]
    print(f"{'operand format':48s} {'|e_c| mean/max':>16s} {'|e_q| mean':>11s} {'bound':>8s} {'actual err rms/max':>20s} {'bound/rms':>9s}"
          f" {'cand/query @10M (sample)':>25s} {'(Gaussian tail)':>16s} {'blockwise C-S / whole-row C-S':>30s}")
    for name, grid, emax in schemes:
        r = res[name]
        eq = float(np.linalg.norm(Q - Qh[name], axis=1).mean())
        ecm = float(np.mean(r["ec_mean"]))
        bound = eq + ecm
        rms = float(np.sqrt(np.mean(np.square(r["err_rms"]))))
        cand_sample = r["cand"] / nq * (N_FULL / (n // 32 * 32))
        cand_tail = N_FULL * norm.sf((tau - bound) / math.sqrt(sigma ** 2 + rms ** 2))
        print(f"{name:48s} {ecm:8.4f}/{r['ec']:.4f} {eq:11.4f} {bound:8.4f} {rms:10.5f}/{r['err_max']:.5f} {bound / rms:9.1f}"
              f" {cand_sample:25.0f} {cand_tail:16.0f} {r['bw'][0]:14.4f} / {r['bw'][1]:.4f}")
    print("\n(cand/query: rows with screen value + bound >= tau, i.e. what the exact re-score -- or an int8 second stage -- would have to "
          "visit per query and pass over the N = 10 M corpus, beyond the k results themselves.)")


if __name__ == "__main__":
    main()
