import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import autorag_research_amd as pkg
from oracle import cpu_ref as oracle
rng = np.random.default_rng(41)
d = 128
def ragged(n_docs, tmin, tmax):
    lens = rng.integers(tmin, tmax + 1, size=n_docs)
    tok = rng.standard_normal((int(lens.sum()), d)).astype(np.float32)
    tok /= np.linalg.norm(tok, axis=1, keepdims=True)
    return tok, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
def queries(lens):
    qs = []
    for t in lens:
        m = rng.standard_normal((t, d)).astype(np.float32)
        if t: m /= np.linalg.norm(m, axis=1, keepdims=True)
        qs.append(m)
    return np.concatenate(qs, axis=0), np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
tok, off = ragged(3000, 20, 150)
for name, lens in [("uniform32x16", [32] * 16), ("uniform24x16", [24] * 16), ("uniform32x12", [32] * 12), ("uniform32x9", [32] * 9),
                   ("ragged", [32, 24, 7, 32, 31, 1, 33, 32, 32, 32, 32, 32, 24, 24, 24, 24, 24, 24, 0, 32])]:
    qtok, qoff = queries(lens)
    rd, rr = oracle.maxsim_topk(tok, off, qtok, qoff, 10)
    with pkg.Mi355Index(d) as idx:
        idx.add_multivec(tok, off)
        for groups in (1, 2, 3, 4):
            idx.set_option("maxsim_pass_groups", groups)
            idx.reset_stats()
            dist, rows = idx.search_maxsim(qtok, qoff, 10)
            bad = [i for i in range(len(lens)) if not np.array_equal(rows[i], rr[i])]
            print(name, "groups", groups, "bad queries", bad, "screened", idx.stat("maxsim_screened"), "fallbacks", idx.stat("maxsim_fallbacks"),
                  "cols", idx.stat("maxsim_screen_cols"))
            for i in bad[:2]:
                print("   q", i, "got", rows[i][:6], "exp", rr[i][:6], dist[i][:3], rd[i][:3])
