"""Synthetic workloads shared by bench.py and the GPU tests (torch tensors, any device; deterministic per chunk).

Two corpus geometries:

* ``gaussian``  -- SURVEY.md 8(d) / BASELINE headline: i.i.d. N(0,1) rows, L2-normalised.  The top-k of a query sits
  ~5 sigma out in an empty tail: the friendliest case for a screen with an additive error bound.
* ``anisotropic`` -- the offline stand-in for a real sentence-embedding corpus (config C2: BEIR nq, bge-base-en-v1.5,
  768-d, inner product, top-100): a shared mean direction (random pairs have cosine ~0.4), a power-law spectrum in a
  randomly rotated basis (no axis carries the anisotropy alone), a few "rogue" coordinates with outlier magnitude, and
  10 % of the rows in near-duplicate clusters whose sizes are heavy-tailed (up to thousands of rows within ~0.03 cosine of
  each other).  Half of the queries are drawn next to cluster centres, so their neighbourhoods are DENSE: thousands of
  rows inside the screen's 2E window -- the case the Gaussian corpus never produces.  Row norms are 1 +- 2 % so that the
  inner-product path's norm-dependent bound is exercised.

No dataset or checkpoint is reachable offline; this only reproduces the geometry that matters to the screen.
"""

from __future__ import annotations

import math

CHUNK_ROWS = 250_000  # generation granule of bench.py (shard boundaries are multiples of it for world in {1,2,4,8})
N_CENTRES = 4096


def _gen(torch, seed: int, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def gaussian_chunk(torch, chunk_index: int, rows: int, dim: int, device):
    """Deterministic chunk: N(0,1) rows, L2-normalised (seed 1234 + chunk, independent of the world size)."""
    x = torch.randn((rows, dim), generator=_gen(torch, 1234 + chunk_index, device), device=device, dtype=torch.float32)
    x /= x.norm(dim=1, keepdim=True)
    return x


class Anisotropic:
    """Generator state of the anisotropic geometry for one (dim, device)."""

    def __init__(self, torch, dim: int, device, alpha: float = 1.0, mean_weight: float = 0.8, dup_fraction: float = 0.10,
                 dup_sigma: float = 0.25, rogue: int = 4):
        self.torch, self.dim, self.device = torch, dim, device
        self.dup_fraction, self.dup_sigma, self.mean_weight = dup_fraction, dup_sigma, mean_weight
        g = _gen(torch, 4242, "cpu")
        # spectrum ~ j^(-alpha/2), a handful of rogue coordinates 6x larger, then a fixed random rotation
        s = torch.arange(1, dim + 1, dtype=torch.float64).pow(-alpha / 2.0)
        s[:rogue] *= 6.0
        s = s / s.pow(2).sum().sqrt()
        q, _ = torch.linalg.qr(torch.randn((dim, dim), generator=g, dtype=torch.float64))
        self.basis = (s[:, None] * q).to(torch.float32).to(device)          # [dim, dim]: z @ basis has the spectrum
        m = torch.randn((dim,), generator=g, dtype=torch.float64)
        self.mean = (m / m.norm()).to(torch.float32).to(device)
        self.centres = self._draw(torch.randn((N_CENTRES, dim), generator=g, dtype=torch.float32).to(device))

    def _draw(self, z):
        """Unit-norm points of the base distribution from standard-normal z."""
        x = z @ self.basis
        x = x / x.norm(dim=1, keepdim=True)
        x = x + self.mean_weight * self.mean
        return x / x.norm(dim=1, keepdim=True)

    def chunk(self, chunk_index: int, rows: int):
        torch = self.torch
        g = _gen(torch, 99_000 + chunk_index, self.device)
        z = torch.randn((rows, self.dim), generator=g, device=self.device, dtype=torch.float32)
        x = self._draw(z)
        u = torch.rand((rows, 3), generator=g, device=self.device, dtype=torch.float32)
        dup = u[:, 0] < self.dup_fraction
        # heavy-tailed cluster sizes: centre index ~ N_CENTRES * u^4 (centre 0 owns ~1/8 of all duplicates)
        c = (u[:, 1].pow(4) * N_CENTRES).long().clamp_(max=N_CENTRES - 1)
        near = self.centres[c] + self.dup_sigma / math.sqrt(self.dim) * z
        near = near / near.norm(dim=1, keepdim=True)
        x = torch.where(dup[:, None], near, x)
        return x * (1.0 + 0.04 * (u[:, 2:3] - 0.5))  # norms 1 +- 2 %

    def queries(self, n: int, seed: int = 4321):
        torch = self.torch
        g = _gen(torch, seed, self.device)
        z = torch.randn((n, self.dim), generator=g, device=self.device, dtype=torch.float32)
        x = self._draw(z)
        u = torch.rand((n, 2), generator=g, device=self.device, dtype=torch.float32)
        c = (u[:, 1].pow(4) * N_CENTRES).long().clamp_(max=N_CENTRES - 1)
        near = self.centres[c] + 2.0 * self.dup_sigma / math.sqrt(self.dim) * z
        near = near / near.norm(dim=1, keepdim=True)
        return torch.where((u[:, 0] < 0.5)[:, None], near, x)


def planted_answers(torch, queries, n_rows_window: int, seed: int = 987, sigmas=(0.3, 0.6, 1.0, 5.0, 7.0)):
    """SURVEY.md 8(d) planted-answer variant: for every query 1-3 'relevant' rows c = normalise(q + sigma * noise),
    noise ~ N(0, I/d) (so cos(q, c) ~ 1/sqrt(1 + sigma^2): 0.96, 0.86, 0.71 for the survey's sigmas -- always retrieved --
    and 0.20, 0.14 for the two hard ones, which sit at the edge of a 10 M-row Gaussian top-10), at distinct row positions
    inside [0, n_rows_window).  Returns (positions [P] int64 sorted, vectors [P, d] on queries.device, owner [P] query
    index, sigma [P])."""
    import numpy as np

    nq, d = queries.shape
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, 4, size=nq)
    owner = np.repeat(np.arange(nq), counts)
    P = int(owner.size)
    pos = np.sort(rng.choice(n_rows_window, size=P, replace=False)).astype(np.int64)
    owner = owner[rng.permutation(P)]
    sig = rng.choice(np.asarray(sigmas, dtype=np.float64), size=P)
    noise = torch.randn((P, d), generator=_gen(torch, seed + 1, queries.device), device=queries.device, dtype=torch.float32)
    q = queries[torch.as_tensor(owner, device=queries.device)]
    q = q / q.norm(dim=1, keepdim=True)
    v = q + torch.as_tensor(sig, device=queries.device, dtype=torch.float32)[:, None] * noise / math.sqrt(d)
    v = v / v.norm(dim=1, keepdim=True)
    return pos, v, owner, sig


def ground_truth(owner, positions, n_queries: int):
    """Per query: the planted row ids as one OR-group (BEIR ingestion) and as an AND-chain (hotpotqa ingestion) --
    reference data/beir.py:191-194 -- ids as strings like the reference's chunk ids."""
    rel = [[] for _ in range(n_queries)]
    for o, p in zip(owner.tolist(), positions.tolist()):
        rel[o].append(str(p))
    return [[r] for r in rel], [[[x] for x in r] for r in rel]
