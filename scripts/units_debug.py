import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lp_amd as lp
from lp_amd import synth
os.environ["LPIPM_OVERLAP"] = "0"
for (m, n) in ((100, 130), (128, 256), (256, 512)):
    A, b, c, _ = synth.planted_lp(0, m, n)
    d = np.random.default_rng(1).uniform(0.1, 3.0, n)
    cx = lp.Context(0)
    cx.upload_arrays(A, b, c)
    M, _ = cx.k_adat(d, 1)
    M2, _ = cx.k_adat(d, 1)
    ref = (A * d) @ A.T
    il = np.tril_indices(m)
    E = np.abs(M - ref); E[np.triu_indices(m, 1)] = 0
    print(m, n, "max err", E[il].max(), "repeat-identical", np.array_equal(M[il], M2[il]), "ref max", np.abs(ref).max())
    bad = np.argwhere(E > 1e-9 * np.abs(ref).max())
    print(" bad count", len(bad), "of", len(il[0]), "first", bad[:6].tolist(), "rows", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist()))[:20])
    if len(bad):
        i, j = bad[0]
        # which partial sums would explain it?
        kc = 128
        parts = [(A[i, k:k + kc] * d[k:k + kc]) @ A[j, k:k + kc] for k in range(0, n, kc)]
        print("  got", M[i, j], "ref", ref[i, j], "chunk partials", parts)
    cx.close()
