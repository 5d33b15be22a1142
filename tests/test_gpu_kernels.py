"""GPU parity tests of the individual kernels, through the C ABI (libmi355dr.so), against the CPU oracle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16_round(x: np.ndarray) -> np.ndarray:
    """numpy emulation of round-to-nearest-even fp32 -> bf16 -> fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


@pytest.fixture(scope="module")
def pkg(native_built):
    import autorag_research_amd as p

    return p


@pytest.mark.parametrize("d", [768, 384, 100, 7])
def test_rescore_chain_is_bit_exact(pkg, oracle, d):
    """exact fp32 chain + pgvector double distance on the GPU == oracle, bit for bit (incl. un-normalised rows)."""
    rng = np.random.default_rng(100 + d)
    n, B = 700, 5
    C = rng.standard_normal((n, d)).astype(np.float32)
    C[3] *= 1e4
    C[5] *= 1e-4
    C[9] = 0.0  # zero-norm row -> NaN distance
    Q = rng.standard_normal((B, d)).astype(np.float32)
    pq = rng.integers(0, B, size=1500).astype(np.int32)
    pr = rng.integers(0, n, size=1500).astype(np.int64)
    pr[:3] = [9, 3, 5]
    with pkg.Mi355Index(d) as idx:
        idx.add(C)
        dot, dist = idx.debug_rescore(Q, pq, pr)
    exp_dot = np.array([oracle.dot(C[r], Q[q]) for q, r in zip(pq, pr)], dtype=np.float32)
    exp_dist = np.array([oracle.cosine_distance(Q[q], C[r]) for q, r in zip(pq, pr)])
    assert np.array_equal(dot.view(np.uint32), exp_dot.view(np.uint32))
    assert np.array_equal(np.isnan(dist), np.isnan(exp_dist))  # NaN payload/sign is not part of the contract
    ok = ~np.isnan(dist)
    assert np.array_equal(dist[ok].view(np.uint64), exp_dist[ok].view(np.uint64))
    assert np.isnan(dist[0])


@pytest.mark.parametrize("d,B", [(768, 130), (384, 3), (100, 17)])
def test_screen_values_match_bf16_emulation(pkg, d, B):
    """the MFMA screen kernel (layout, swizzle, staging) == numpy emulation of the bf16 shadow product."""
    rng = np.random.default_rng(7 + d)
    n = 1500
    C = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.1, 10, size=(n, 1)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.set_option("screen_dtype", "bf16")
        idx.add(C)
        Eq = idx.debug_screen_bound(Q).astype(np.float64)  # from the MEASURED rounding residuals of queries and rows
        assert (Eq <= 2.0 ** -7 + 2.0 ** -15 + 8 * d * 2.0 ** -24).all() and (Eq > 2.0 ** -9).all()
        for row0, cnt in [(0, 1500), (256, 300), (1024, 476)]:
            t = idx.debug_screen_dense(Q, row0, cnt)
            sub = C[row0:row0 + cnt].astype(np.float64)
            ch = _bf16_round((sub / np.linalg.norm(sub, axis=1, keepdims=True)).astype(np.float32)).astype(np.float64)
            qh = _bf16_round((Q.astype(np.float64) / np.linalg.norm(Q.astype(np.float64), axis=1, keepdims=True))
                             .astype(np.float32)).astype(np.float64)
            ref = qh @ ch.T
            assert not np.isnan(t).any(), "some (query,row) pairs were never produced by the kernel"
            # shadow normalisation happens in fp32 on the device: allow bf16 ulp flips on a few elements
            assert np.abs(t - ref).max() < 2e-3
            assert np.abs(t - ref).mean() < 2e-5
            # and the screen bound itself: |t - exact cosine| <= E
            cos = (Q.astype(np.float64) @ sub.T) / (np.linalg.norm(Q.astype(np.float64), axis=1)[:, None]
                                                     * np.linalg.norm(sub, axis=1)[None, :])
            assert (np.abs(t - cos) <= Eq[:, None]).all()


@pytest.mark.parametrize("screen", ["bf16", "i8"])
@pytest.mark.parametrize("d,B,spread", [(768, 130, 0.02), (128, 17, 3.0), (5, 40, 0.5)])
def test_inner_product_screen_estimates_dot_over_query_norm(pkg, screen, d, B, spread):
    """inner-product metric (round 3): the shadows hold the rows THEMSELVES, the screen value estimates <q_hat, c> = dot / |q|
    and exact dot / |q| <= value + E with E in the rows' units (scaled by the largest row norm) -- row norms spread by 2 %
    (the C2 stand-in), by e^+-3, and low-dimensional aligned data; rows appended in two batches with a growing largest norm."""
    rng = np.random.default_rng(1000 + d)
    n = 1536
    C = rng.standard_normal((n, d)).astype(np.float32)
    if d == 5:
        C = C[rng.integers(0, 8, size=n)] + (0.01 * rng.standard_normal((n, d))).astype(np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    C *= np.exp(spread * rng.uniform(-1, 1, size=(n, 1))).astype(np.float32)
    C[900:] *= 1.7   # the second batch raises the largest norm
    Q = (C[rng.integers(0, n, size=B)] + 0.05 * rng.standard_normal((B, d)).astype(np.float32)) * 3.0
    with pkg.Mi355Index(d, "ip") as idx:
        idx.set_option("screen_dtype", screen)
        idx.add(C[:900])
        idx.add(C[900:])
        E = idx.debug_screen_bound(Q).astype(np.float64)
        cmax = float(np.linalg.norm(C.astype(np.float64), axis=1).max())
        assert (E > 0).all() and (E < 0.05 * cmax).all()
        qh = Q.astype(np.float64) / np.linalg.norm(Q.astype(np.float64), axis=1, keepdims=True)
        for row0, cnt in [(0, 1536), (512, 300)]:
            t = idx.debug_screen_dense(Q, row0, cnt).astype(np.float64)
            ref = qh @ C[row0:row0 + cnt].astype(np.float64).T
            seen = ~np.isnan(t).all(axis=0)   # (a row with an outlier component stays out of the int8 shadow: "loose")
            assert seen.sum() >= cnt - 2 and not np.isnan(t[:, seen]).any()
            t, ref = t[:, seen], ref[:, seen]
            if screen == "bf16":
                assert (np.abs(t - ref) <= E[:, None]).all()
            else:  # int8: the value already carries the pair's share of the bound: ref <= t + E, and t is not far above
                assert (ref <= t + E[:, None]).all()
                assert (t - ref <= 0.06 * cmax).all()


@pytest.mark.parametrize("d", [2, 5, 16])
def test_bf16_bound_holds_on_low_dimensional_aligned_data(pkg, d):
    """worst case of the bf16 bound: few dimensions, query and rows almost parallel, so the per-operand rounding errors
    (up to 2^-8 each, same sign) add up instead of averaging out.  The first form of the bound (2^-9 per operand) failed here."""
    rng = np.random.default_rng(d)
    n, B = 4096, 64
    base = rng.standard_normal((B, d)).astype(np.float32)
    C = base[rng.integers(0, B, size=n)] * rng.uniform(0.5, 2.0, size=(n, 1)).astype(np.float32)
    C += (0.01 * rng.standard_normal((n, d))).astype(np.float32)
    Q = base + (0.01 * rng.standard_normal((B, d))).astype(np.float32)
    with pkg.Mi355Index(d) as idx:
        idx.set_option("screen_dtype", "bf16")
        idx.add(C)
        E = idx.debug_screen_bound(Q).astype(np.float64)
        worst = 0.0
        for row0 in range(0, n, 1024):
            t = idx.debug_screen_dense(Q, row0, 1024).astype(np.float64)
            sub = C[row0:row0 + 1024].astype(np.float64)
            cos = (Q.astype(np.float64) @ sub.T) / (np.linalg.norm(Q.astype(np.float64), axis=1)[:, None]
                                                     * np.linalg.norm(sub, axis=1)[None, :])
            assert (np.abs(t - cos) <= E[:, None]).all()
            worst = max(worst, float(np.abs(t - cos).max()))
        assert worst > 2.0 ** -9 or d > 5  # (the first, wrong constant really is exceeded in the smallest dimensions)


@pytest.mark.parametrize("d,B", [(768, 130), (384, 3), (100, 17), (1000, 40)])
def test_int8_screen_values_and_bound(pkg, d, B):
    """the int8 MFMA screen (v_mfma_i32_32x32x32_i8, exact int32 accumulate; one step per group of 32 rows) == numpy
    emulation of the quantised product + the group's share of the bound, and exact cosine <= value + E (the per-query part
    of the bound the library cuts candidates with)."""
    rng = np.random.default_rng(17 + d)
    n = 1500
    C = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.1, 10, size=(n, 1)).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    Q[0] *= 1e3
    with pkg.Mi355Index(d) as idx:
        idx.set_option("screen_dtype", "i8")
        idx.add(C[:700])
        idx.add(C[700:])  # 700 = 21 groups + 28 rows: the second add() lands in a partly filled group and rebuilds it
        assert idx.stat("screen_dtype_active") == 2
        # Gaussian rows have no component beyond 6 sigma: nothing is loose
        assert idx.stat("loose_rows") == 0
        E = idx.debug_screen_bound(Q).astype(np.float64)
        n_groups = (n + 31) // 32
        sq, kq, step_g, err_g = (x.astype(np.float64) for x in idx.debug_i8_state(Q, 0, n_groups))
        ch_all = C.astype(np.float64) / np.linalg.norm(C.astype(np.float64), axis=1, keepdims=True)
        peak = np.abs(ch_all).max(axis=1)
        peak_g = np.array([peak[g * 32:(g + 1) * 32].max() for g in range(n_groups)])
        assert np.allclose(step_g, peak_g / 127.0, rtol=2e-6, atol=0)  # one step per group: its largest component / 127
        # per-query part ~ the query's own rounding residual; per-group part ~ S_g sqrt(d/12)
        assert (E > 0.003).all() and (E < 0.012).all()
        assert np.allclose(err_g, step_g * np.sqrt(d / 12.0), rtol=0.2)
        assert (kq > 1.0).all() and (kq < 1.05).all()
        qh = (Q.astype(np.float64) / np.linalg.norm(Q.astype(np.float64), axis=1, keepdims=True))
        step_q = np.abs(qh).max(axis=1, keepdims=True) / 127.0
        assert np.allclose(sq, step_q[:, 0], rtol=2e-6, atol=0)
        q8 = np.clip(np.rint(qh / step_q), -127, 127)
        for row0, cnt in [(0, 1500), (256, 300), (1024, 476)]:
            t = idx.debug_screen_dense(Q, row0, cnt).astype(np.float64)
            assert not np.isnan(t).any(), "some (query,row) pairs were never produced by the kernel"
            sub = C[row0:row0 + cnt].astype(np.float64)
            ch = ch_all[row0:row0 + cnt]
            sg = step_g[np.arange(row0, row0 + cnt) // 32]
            eg = err_g[np.arange(row0, row0 + cnt) // 32]
            c8 = np.clip(np.rint(ch / sg[:, None]), -127, 127)
            # the measured residual norm of every row is inside its group's record
            assert (np.linalg.norm(ch - c8 * sg[:, None], axis=1) <= eg * (1 + 1e-6) + 1e-7).all()
            ref = (q8 @ c8.T) * step_q * sg[None, :] + kq[:, None] * eg[None, :]
            # the device normalises in fp32, so a few components round to the neighbouring step: each flip moves
            # the integer accumulator by <= 127 units
            unit = step_q * sg[None, :]
            assert (np.abs(t - ref) <= 4 * 127 * unit + 1e-6).all()
            assert np.abs(t - ref).mean() < 1e-5
            cos = (Q.astype(np.float64) @ sub.T) / (np.linalg.norm(Q.astype(np.float64), axis=1)[:, None]
                                                     * np.linalg.norm(sub, axis=1)[None, :])
            # what the cut relies on: the value is an upper bound of the cosine up to the per-query part ...
            assert (cos <= t + E[:, None]).all()
            # ... and not a vacuous one: it never overshoots by more than the whole two-sided bound
            assert (t - cos <= E[:, None] + 2.0 * kq[:, None] * eg[None, :] + 1e-6).all()
            assert np.abs(t - kq[:, None] * eg[None, :] - cos).max() > 0.05 * (E.min() + eg.min())
