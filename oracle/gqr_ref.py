"""CPU oracle of the Guided Query Refinement loops -- TEST INFRASTRUCTURE, never imported by the product.

Restates, in plain numpy float64, the arithmetic of the reference's GQR hybrid pipeline
(autorag_research/pipelines/retrieval/gqr_hybrid.py): the consensus step shared by the three optimisers
(softmax :39-51, target / logit gradient :313-316), cosine scores and their gradient w.r.t. the query (:65-92),
mean-of-max late-interaction scores and their argmax subgradient (:95-127), and the three loops (:306-362).
Pinned by tests/golden/gqr_golden.npz, which tests/golden/make_golden.py generates by importing the reference
(tests/test_oracle_golden.py); the HIP kernels (csrc/k_gqr.h) are then compared with this file.
"""

from __future__ import annotations

import numpy as np

EPS = 1e-8  # gqr_hybrid.py:36


def softmax(scores: np.ndarray, temperature: float) -> np.ndarray:
    """Max-shifted softmax of scores / max(T, eps); uniform when the normaliser is not finite or vanishes (:39-51)."""
    z = np.asarray(scores, dtype=np.float64)
    if z.size == 0:
        return z
    z = z / max(temperature, EPS)
    e = np.exp(z - z.max())
    total = float(e.sum())
    if not np.isfinite(total) or total <= EPS:
        return np.full(z.shape, 1.0 / z.size)
    return e / total


def logit_grad(scores: np.ndarray, comp: np.ndarray, temperature: float, alpha: float) -> np.ndarray:
    """(p - ((1-a) p + a comp)) / T with p = softmax(scores / T)  (:313-316, :333-336, :354-357)."""
    t = max(temperature, EPS)
    p = softmax(scores, t)
    return (p - ((1.0 - alpha) * p + alpha * comp)) / t


def missing_score_floor(score_map: dict) -> float:
    """Score given to a pool member one retriever did not return: below its minimum by max(1, spread) (:54-62)."""
    if not score_map:
        return -1.0
    lo, hi = min(score_map.values()), max(score_map.values())
    return lo - max(1.0, hi - lo)


def cosine_scores(q: np.ndarray, C: np.ndarray) -> np.ndarray:
    """cos(q, c_j) with norms floored at eps; all zeros for a zero query (:65-74)."""
    qn = np.sqrt(np.dot(q, q))
    if qn <= EPS:
        return np.zeros(C.shape[0])
    cn = np.maximum(np.sqrt((C * C).sum(axis=1)), EPS)
    return (C @ q) / (cn * qn)


def cosine_grads(q: np.ndarray, C: np.ndarray, cos: np.ndarray) -> np.ndarray:
    """d cos_j / d q = c_j / (|c_j| |q|) - cos_j q / |q|^2   (:77-92)."""
    qn = np.sqrt(np.dot(q, q))
    if qn <= EPS:
        return np.zeros_like(C)
    cn = np.maximum(np.sqrt((C * C).sum(axis=1)), EPS)
    return C / (cn[:, None] * qn) - (cos[:, None] * q[None, :]) / (qn * qn)


def refine_single(q0, C, comp, n_steps: int, lr: float, temperature: float, alpha: float) -> np.ndarray:
    """Gradient steps on the query vector, then the cosine scores of the refined query (:321-340)."""
    q = np.array(q0, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    comp = np.asarray(comp, dtype=np.float64)
    for _ in range(n_steps):
        cos = cosine_scores(q, C)
        g = logit_grad(cos, comp, temperature, alpha)
        q -= lr * (g[:, None] * cosine_grads(q, C, cos)).sum(axis=0)
    return cosine_scores(q, C)


def maxsim_scores(Q: np.ndarray, docs: list[np.ndarray]) -> np.ndarray:
    """(1/n_q) sum_i max_j <q_i, d_j> per doc; 0 for a doc without vectors (:95-110)."""
    if Q.size == 0:
        return np.zeros(len(docs))
    n_q = max(Q.shape[0], 1)
    return np.array([0.0 if D.size == 0 else float((Q @ D.T).max(axis=1).sum() / n_q) for D in docs])


def maxsim_grads(Q: np.ndarray, docs: list[np.ndarray]) -> np.ndarray:
    """Subgradient: row i of doc j's gradient is its FIRST best-matching vector / n_q (:113-127)."""
    out = np.zeros((len(docs), *Q.shape))
    if Q.size == 0:
        return out
    n_q = max(Q.shape[0], 1)
    for j, D in enumerate(docs):
        if D.size:
            out[j] = D[np.argmax(Q @ D.T, axis=1)] / n_q
    return out


def refine_multi(Q0, docs, comp, n_steps: int, lr: float, temperature: float, alpha: float) -> np.ndarray:
    """Gradient steps on the query matrix, then the late-interaction scores of the refined matrix (:342-362)."""
    Q = np.array(Q0, dtype=np.float64)
    docs = [np.asarray(D, dtype=np.float64) for D in docs]
    comp = np.asarray(comp, dtype=np.float64)
    for _ in range(n_steps):
        s = maxsim_scores(Q, docs)
        g = logit_grad(s, comp, temperature, alpha)
        Q -= lr * (g[:, None, None] * maxsim_grads(Q, docs)).sum(axis=0)
    return maxsim_scores(Q, docs)


def refine_scores(primary, comp, n_steps: int, lr: float, temperature: float, alpha: float) -> np.ndarray:
    """No vectors available: the same consensus steps taken on the primary scores themselves (:306-319)."""
    z = np.array(primary, dtype=np.float64)
    comp = np.asarray(comp, dtype=np.float64)
    for _ in range(n_steps):
        z -= lr * logit_grad(z, comp, temperature, alpha)
    return z
