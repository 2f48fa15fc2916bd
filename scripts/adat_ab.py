"""A.D.A^T kernel variants side by side in ONE process, interleaved rounds (cdna guide rule 24): child processes per variant
because the variant is chosen by an environment variable read once."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, lp_amd
    from lp_amd import synth
    m, n = int(sys.argv[2]), int(sys.argv[3])
    A, b, c, xs = synth.planted_lp(0, m, n)
    ctx = lp_amd.Context(0); ctx.upload_arrays(A, b, c)
    d = np.random.default_rng(0).uniform(0.5, 2.0, n)
    M, ms = ctx.k_adat(d, repeats=20)
    ref = (A * d) @ A.T
    err = np.abs(np.tril(M) - np.tril(ref)).max() / np.abs(ref).max()
    for _ in range(3):
        M, ms = ctx.k_adat(d, repeats=300)
        print(f"  {m}x{n}: {ms:.4f} ms per launch = {m*(m+1)*n/ms/1e9:.2f} TFLOP/s (rel err vs numpy {err:.1e})", flush=True)
else:
    for rnd in range(2):
        for name, env in (("4 waves/WG (2 per SIMD)", {"LPIPM_ADAT_W4": "1"}), ("8 waves/WG (4 per SIMD)", {})):
            print(name, flush=True)
            for (m, n) in ((4096, 8192), (1024, 2048)):
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(m), str(n)], env=dict(os.environ, **env))
