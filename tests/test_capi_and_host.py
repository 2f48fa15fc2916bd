"""C-ABI library loads and exports every symbol include/lpipm.h declares; host-side logic (problem
assembly, option validation, error mapping, synthetic generator) -- no compute calls, no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(built):
    from lp_amd import _capi
    hdr = open(os.path.join(ROOT, "include", "lpipm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only, not prose
    declared = set(re.findall(r"\b(lpipm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "header parse failed"
    lib = C.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/lpipm.h but not exported"
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    _capi.lib()


def test_struct_layouts_match_header(built):
    from lp_amd import _capi
    assert C.sizeof(_capi.Opts) == 40 and C.sizeof(_capi.IterRow) == 56 and C.sizeof(_capi.PhaseTimes) == 72


def test_default_opts_and_strerror(built):
    from lp_amd import _capi
    o = _capi.Opts()
    _capi.lib().lpipm_default_opts(C.byref(o))
    # interior_point/mod.rs:52-59
    assert (o.tol, o.alpha0, o.max_iter, o.ip, o.solver_type, o.disp) == (1e-8, 0.99995, 1000, 1, 0, 0)
    assert "infeasible" in _capi.strerror(_capi.INFEASIBLE)          # error.rs:23-24
    assert "unbounded" in _capi.strerror(_capi.UNBOUNDED)            # error.rs:26-27
    assert "unconstrained" in _capi.strerror(_capi.UNCONSTRAINED)    # error.rs:11-12
    assert _capi.strerror(12345) == "unknown status"


def test_problem_builder_matches_oracle_and_reference_shape(built):
    """linear_program.rs:125-169 via lp_amd.Problem (C host code) vs the oracle's restatement."""
    import lp_amd as lp
    from oracle import capi as oracle
    rng = np.random.default_rng(0)
    for (n, mub, meq) in [(2, 2, 1), (5, 0, 3), (4, 6, 0), (7, 3, 2)]:
        c = rng.standard_normal(n)
        Aub, bub = rng.standard_normal((mub, n)), rng.standard_normal(mub)
        Aeq, beq = rng.standard_normal((meq, n)), rng.standard_normal(meq)
        b = lp.Problem.target(c)
        if mub:
            b = b.ub(Aub, bub)
        if meq:
            b = b.eq(Aeq, beq)
        prob = b.build()
        st, A, bb, cc, ns = oracle.problem_build(c, Aub if mub else None, bub if mub else None,
                                                 Aeq if meq else None, beq if meq else None)
        assert st == 0
        assert np.array_equal(prob.A(), A) and np.array_equal(prob.b(), bb) and np.array_equal(prob.c(), cc)
        assert prob.n_slack() == ns == mub and prob.c0() == 0.0
        assert prob.A().shape == (mub + meq, n + mub)
        assert prob.denormalize_x_into(np.arange(n + mub, dtype=float)).shape == (n,)


def test_problem_builder_errors(built):
    """linear_program.rs:134-143."""
    import lp_amd as lp
    with pytest.raises(lp.Unconstrained):
        lp.Problem.target([1.0, 2.0]).build()
    with pytest.raises(lp.IncompatibleInputDimensions):
        lp.Problem.target([1.0, 2.0]).ub([[1.0]], [1.0]).build()
    with pytest.raises(lp.IncompatibleInputDimensions):
        lp.Problem.target([1.0, 2.0]).ub([[1.0, 1.0]], [1.0, 2.0]).build()
    with pytest.raises(lp.IncompatibleInputDimensions):
        lp.Problem.target([1.0, 2.0]).ub([[1.0, 1.0]], [1.0]).eq([[1.0, 1.0, 1.0]], [1.0]).build()


def test_interior_point_builder(built):
    """interior_point/mod.rs:118-137, :249-254 (default_builder_doesnt_panic)."""
    import lp_amd as lp
    assert lp.InteriorPoint.default() == lp.InteriorPoint.custom().build()
    for bad in (0.0, 1.0, -0.1, 1.5):
        with pytest.raises(lp.InvalidParameter):
            lp.InteriorPoint.custom().alpha0(bad).build()
    for bad in (0.0, -1e-8):
        with pytest.raises(lp.InvalidParameter):
            lp.InteriorPoint.custom().tol(bad).build()
    s = (lp.InteriorPoint.custom().tol(1e-6).disp(True).ip(False).solver_type(lp.EquationSolverType.Inverse)
         .alpha0(0.9).max_iter(7).build())
    o = s.opts()
    assert (o.tol, o.alpha0, o.max_iter, o.ip, o.solver_type, o.disp) == (1e-6, 0.9, 7, 0, 1, 1)


def test_synth_generator_c_vs_python_mirror(built):
    from lp_amd import synth
    for seed, m, n in [(0, 5, 9), (7, 16, 40)]:
        A, b, c, xs = synth.planted_lp(seed, m, n)
        A2, b2, c2, xs2 = synth.planted_lp_py(seed, m, n)
        assert np.abs(A - A2).max() < 1e-14 and np.array_equal(xs, xs2)
        assert np.abs(b - b2).max() < 1e-12 and np.abs(c - c2).max() < 1e-12
        assert (xs > 0).sum() == m and np.abs(A @ xs - b).max() < 1e-12
    A3, *_ = synth.planted_lp(1, 5, 9)
    assert not np.array_equal(A3, synth.planted_lp(0, 5, 9)[0])


def test_no_gpu_means_loud_failure(built):
    """The product path must fail loudly when there is no usable device -- never fall back."""
    import lp_amd as lp
    from lp_amd import _capi
    if _capi.lib().lpipm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(lp.BackendError):
        lp.Context(0)
    prob = lp.Problem.target([-1.0, 4.0]).ub([[-3.0, 1.0], [1.0, 2.0]], [6.0, 4.0]).build()
    with pytest.raises(lp.BackendError):
        lp.InteriorPoint.default().solve(prob)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under lp_amd/ or include/ may reference it."""
    bad = []
    for base in ("lp_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    for line in txt.splitlines():
                        s = line.strip()
                        if re.search(r"^\s*(from|import)\s+oracle\b", s) or re.search(r"#include.*oracle", s) \
                                or "liboracle" in s:
                            bad.append((fn, s))
    assert not bad, bad


def test_units_kernel_has_no_spills_and_four_waves_per_simd():
    """Build-time property of the A.D.A^T units kernel (kernels_gemm.hip, gemm_nt_units_kernel): 4 waves per SIMD (two
    512-thread workgroups per CU) with NO VGPR / SGPR spill and no scratch -- at 128 VGPRs a spill inside the main loop puts
    an s_waitcnt vmcnt(0) in front of the prefetch it has just issued.  Read from hipcc's own resource-usage remarks."""
    import re
    import subprocess
    src = os.path.join(ROOT, "lp_amd", "csrc", "kernels_gemm.hip")
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950", "--cuda-device-only", "-c",
                          "-Rpass-analysis=kernel-resource-usage", src, "-o", os.devnull],
                         capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
    blocks = re.split(r"remark: Function Name: ", out)
    units = [b for b in blocks if "gemm_nt_units_kernel" in b.splitlines()[0]]
    assert len(units) == 2, [b.splitlines()[0] for b in blocks[1:]]          # <GRP = false>, <GRP = true>
    for b in units:
        get = lambda key: int(re.search(key + r": (\d+)", b).group(1))
        assert get(r"VGPRs Spill") == 0 and get(r"SGPRs Spill") == 0 and get(r"ScratchSize \[bytes/lane\]") == 0, b
        assert get(r"Occupancy \[waves/SIMD\]") == 4 and get(r"VGPRs") <= 128, b
