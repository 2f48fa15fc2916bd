"""The vector stage at iteration granularity: ONE loop body of solve_normal_form from a GIVEN
iterate -- Rhat::predictor / ::corrector (rhat.rs:17-75, both ip arms), Delta::compute (delta.rs:21-49), update_gamma
(feasible_point.rs:156-165), get_step_size (:53-72: minima over x, z, tau, kappa with alpha0 applied after the min),
do_step (:76-106, incl. the ip clamp) -- device (lpipm_k_iteration) against the oracle (oracle_iteration) on RANDOM
interior iterates, not only along solver trajectories: directions with negative, zero and positive entries, the step
blocked by x, by z, by tau or by kappa."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _iterate(rng, m, n, spread):
    x = np.exp(rng.uniform(-spread, spread, n)); z = np.exp(rng.uniform(-spread, spread, n))
    return x, rng.standard_normal(m), z, float(np.exp(rng.uniform(-1, 1))), float(np.exp(rng.uniform(-1, 1)))


def _compare(dev, ref, x0, z0):
    assert ref["status"] == 0 and dev["info"] == 0
    scale = lambda a: max(1.0, np.abs(a).max())
    for k in ("d_x", "d_y", "d_z"):
        assert np.abs(dev[k] - ref[k]).max() <= 1e-8 * scale(ref[k]), k
    for k in ("d_tau", "d_kappa"):
        assert abs(dev[k] - ref[k]) <= 1e-8 * max(1.0, abs(ref[k])), k
    assert abs(dev["alpha"] - ref["alpha"]) <= 1e-8
    for k in ("x", "y", "z"):
        assert np.abs(dev[k] - ref[k]).max() <= 1e-8 * scale(ref[k]), k
    assert abs(dev["tau"] - ref["tau"]) <= 1e-8 * max(1.0, abs(ref["tau"]))
    assert abs(dev["kappa"] - ref["kappa"]) <= 1e-8 * max(1.0, abs(ref["kappa"]))


@pytest.mark.parametrize("m,n,seed,spread", [(3, 4, 0, 1.0), (40, 100, 1, 1.0), (100, 333, 2, 2.0), (256, 512, 3, 1.5),
                                            (200, 420, 4, 3.0)])
@pytest.mark.parametrize("ip", [False, True])
def test_one_iteration_from_random_iterates(ctx, m, n, seed, spread, ip):
    import lp_amd as lp
    from lp_amd import synth
    from oracle import capi as oracle
    A, b, c, _ = synth.planted_lp(seed, m, n)
    ctx.upload_arrays(A, b, c)
    rng = np.random.default_rng(100 * seed + int(ip))
    o = lp.InteriorPoint.default().opts()
    blocked_by = set()
    for trial in range(4):
        x, y, z, tau, kappa = _iterate(rng, m, n, spread)
        ref = oracle.iteration(A, b, c, x, y, z, tau, kappa, ip=ip)
        dev = ctx.k_iteration(o, x, y, z, tau, kappa, ip=ip)
        _compare(dev, ref, x, z)
        if not ip:                                   # which quantity blocked the step (feasible_point.rs:61-71)
            d = ref
            cand = {"x": np.min(np.where(d["d_x"] < 0, x / -np.where(d["d_x"] < 0, d["d_x"], -1.0), np.inf)),
                    "z": np.min(np.where(d["d_z"] < 0, z / -np.where(d["d_z"] < 0, d["d_z"], -1.0), np.inf)),
                    "tau": tau / -d["d_tau"] if d["d_tau"] < 0 else np.inf,
                    "kappa": kappa / -d["d_kappa"] if d["d_kappa"] < 0 else np.inf}
            blocked_by.add(min(cand, key=cand.get) if min(cand.values()) < 1.0 else "none")
            assert (d["d_x"] < 0).any() and (d["d_x"] > 0).any()
        else:
            assert dev["alpha"] == 1.0 and dev["x"].min() >= 1.0 and dev["z"].min() >= 1.0      # ip arm: alpha = 1, clamp
            assert dev["tau"] >= 1.0 and dev["kappa"] >= 1.0
    if not ip:
        assert len(blocked_by) >= 1


def test_ratio_test_blocked_by_tau_and_by_kappa(ctx):
    """Iterates constructed so that tau (resp. kappa) is what limits the step: a tiny tau (kappa) next to large x, z."""
    import lp_amd as lp
    from lp_amd import synth
    from oracle import capi as oracle
    m, n = 30, 80
    A, b, c, _ = synth.planted_lp(7, m, n)
    ctx.upload_arrays(A, b, c)
    o = lp.InteriorPoint.default().opts()
    rng = np.random.default_rng(5)
    seen = set()
    for tau, kappa in ((1e-3, 1.0), (1.0, 1e-3), (1e-4, 1e-4), (50.0, 1e-2), (1e-2, 50.0)):
        x, y, z = np.full(n, 5.0) + rng.uniform(0, 1, n), 0.1 * rng.standard_normal(m), np.full(n, 5.0) + rng.uniform(0, 1, n)
        ref = oracle.iteration(A, b, c, x, y, z, tau, kappa, ip=False)
        dev = ctx.k_iteration(o, x, y, z, tau, kappa, ip=False)
        _compare(dev, ref, x, z)
        rt = tau / -ref["d_tau"] if ref["d_tau"] < 0 else np.inf
        rk = kappa / -ref["d_kappa"] if ref["d_kappa"] < 0 else np.inf
        rx = np.min(np.where(ref["d_x"] < 0, x / -np.where(ref["d_x"] < 0, ref["d_x"], -1.0), np.inf))
        rz = np.min(np.where(ref["d_z"] < 0, z / -np.where(ref["d_z"] < 0, ref["d_z"], -1.0), np.inf))
        if min(rt, rk, rx, rz) < 1.0:
            seen.add(["tau", "kappa", "x", "z"][int(np.argmin([rt, rk, rx, rz]))])
    assert "tau" in seen or "kappa" in seen, seen
