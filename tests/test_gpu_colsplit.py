"""One LP split by columns over ranks (BASELINE config C5 shape class): N gloo ranks, all on
cuda:0, each holding a column block; the library's cross-rank reductions go through the callback of
lpipm_set_collective.  Checked against the oracle (same iteration count, |dx| <= 1e-6) and against the
single-context solve of the same LP."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, seed, m, n, align):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import lp_amd
    from lp_amd import synth
    from lp_amd.colsplit import column_range, solve_column_split
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, c = synth.planted_lp(seed, m, n)[:3]
    cols = column_range(n, world, rank, align)
    opts = lp_amd.InteriorPoint.default().opts()
    rc, x, fun, it, rows, coll = solve_column_split(np.ascontiguousarray(A[:, cols.start:cols.stop]), b,
                                                    c[cols.start:cols.stop], n, 0.0, opts, want_log=True,
                                                    on_stream=True if os.environ.get("LPIPM_TEST_ON_STREAM") == "1" else None)
    q.put((rank, rc, cols.start, x.tolist(), fun, it, rows, coll.calls, coll.bytes))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, seed, m, n, align=128):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 131 + m) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, seed, m, n, align)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,seed,m,n,align", [(2, 0, 256, 512, 128), (3, 1, 100, 333, 64), (2, 3, 512, 1280, 128)])
def test_column_split_matches_oracle_and_single(ctx, world, seed, m, n, align):
    import lp_amd
    from lp_amd import synth
    from oracle import capi as oracle
    got = _run(world, seed, m, n, align)
    A, b, c = synth.planted_lp(seed, m, n)[:3]
    ref = oracle.solve(A, b, c)
    ctx.upload_arrays(A, b, c)
    rc1, x1, fun1, it1, _ = ctx.solve_raw(lp_amd.InteriorPoint.default().opts())
    assert rc1 == 0 and ref["status"] == 0
    x = np.full(n, np.nan)
    for rank, rc, lo, xs, fun, it, rows, calls, nbytes in got:
        assert rc == 0
        assert it == ref["iterations"] == it1                      # same trajectory length on every rank
        assert abs(fun - ref["fun"]) <= 1e-6 * max(1.0, abs(ref["fun"]))
        assert fun == got[0][4] and rows == got[0][6]              # replicated scalars are bit-identical across ranks
        x[lo:lo + len(xs)] = xs
        # per iteration: M, 2 x (A w), 6 scalar groups; plus the residual / final ones
        assert calls >= 9 * it
    assert not np.isnan(x).any()
    assert np.abs(x - ref["x_slack"]).max() <= 1e-6
    assert np.abs(x - x1).max() <= 1e-6


def test_column_split_world_one_is_the_plain_solve(ctx):
    """world = 1 through the n-split entry points (no callback needed) walks the same code with gs reductions
    local: must reproduce the plain solve bit for bit."""
    import lp_amd
    from lp_amd import synth
    A, b, c = synth.planted_lp(5, 192, 448)[:3]
    opts = lp_amd.InteriorPoint.default().opts()
    ctx.upload_arrays(A, b, c)
    rc0, x0, f0, it0, _ = ctx.solve_raw(opts)
    ctx.set_collective(0, 1, None)
    ctx.upload_column_block(A, b, c, A.shape[1])
    rc1, x1, f1, it1, _ = ctx.solve_raw(opts)
    ctx.upload_arrays(A, b, c)                                     # back to the plain mode for later tests
    assert (rc0, it0) == (rc1, it1) == (0, it0)
    assert np.array_equal(x0, x1) and f0 == f1


def test_column_split_infeasible_agrees_on_all_ranks(ctx):
    """Control flow (status decisions) must be identical on every rank or the collectives would deadlock."""
    import lp_amd
    A = np.array([[1.0, 1.0, 1.0, 1.0]]); b = np.array([-1.0]); c = np.ones(4)
    ctx.set_collective(0, 1, None)
    ctx.upload_column_block(A, b, c, 4)
    rc, _, _, _, _ = ctx.solve_raw(lp_amd.InteriorPoint.default().opts())
    ctx.upload_arrays(A, b, c)
    assert rc == lp_amd._capi.INFEASIBLE


def test_column_split_on_stream_contract(ctx):
    """lpipm_set_collective_on_stream(1): the library does not drain before the callback (which here orders itself on
    the stream it is handed).  Two gloo ranks on cuda:0, against the drained contract: bit-identical."""
    import lp_amd
    got_a = _run(2, 0, 256, 512, 128)
    os.environ["LPIPM_TEST_ON_STREAM"] = "1"
    try:
        got_b = _run(2, 0, 256, 512, 128)
    finally:
        del os.environ["LPIPM_TEST_ON_STREAM"]
    for a, b in zip(got_a, got_b):
        assert a[1] == b[1] == 0 and a[5] == b[5] and a[3] == b[3] and a[4] == b[4]


def test_split_over_too_many_ranks_is_refused_everywhere():
    from lp_amd.colsplit import check_split, column_range
    check_split(512, 4)
    with pytest.raises(ValueError):
        check_split(512, 8)                      # 4 column groups, 8 ranks: ranks 4..7 would hold nothing
    assert len(column_range(512, 8, 7)) == 0


def test_c5_two_ranks_on_one_gpu_at_c3_size_matches_golden(built):
    """scripts/bench_c5.py: the strong-scaling harness of BASELINE config 5 with 2 ranks (gloo) sharing cuda:0 at
    4096 x 8192, against the committed oracle vector of the headline LP."""
    import json
    import subprocess
    port = 29500 + (os.getpid() * 11) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", "bench_c5.py"), "--rows", "4096", "--cols", "8192",
           "--steps", "1", "--warmup", "0", "--backend", "gloo", "--one-gpu", "--seed", "0",
           "--golden", os.path.join(ROOT, "tests", "golden", "planted_4096x8192_s0.npz")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["config"]["max_abs_err_vs_golden"] <= 1e-6
    assert r["config"]["allreduce_MB_per_iteration"] > 60.0          # the packed lower block-triangle of M: 67.6 MB


def test_bench_distributed_code_path_on_one_rank(built):
    """bench.py's N > 1 code path (init_process_group nccl = RCCL, barrier, all-gather, max-over-ranks) on ONE rank."""
    import json
    import subprocess
    env = dict(os.environ, LPIPM_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29500 + (os.getpid() * 13) % 2000))
    for extra in (["--m", "512", "--n", "1024", "--steps", "3", "--no-cpu-baseline"], ["--workload", "c4", "--steps", "1"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--warmup", "1"] + extra,
                             capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert r["n_gpus"] == 1 and r["value"] > 0


def _worker_two_ways(rank, world, port, q, seed, m, n):
    """The same column-split solve twice: group-by-group reduction of M behind the running A.D.A^T launch (default) and the
    one-block reduction after it (LPIPM_ADAT_UNITS=0: the round-2 path)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import lp_amd
    from lp_amd import synth
    from lp_amd.colsplit import column_range, solve_column_split
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, c = synth.planted_lp(seed, m, n)[:3]
    cols = column_range(n, world, rank, 128)
    opts = lp_amd.InteriorPoint.default().opts()
    out = []
    for env in (None, "0"):
        if env is not None:
            os.environ["LPIPM_EXPERIMENTAL"] = "1"          # knobs are read only with the master switch on
            os.environ["LPIPM_ADAT_UNITS"] = env
        ctx = lp_amd.Context(0)
        os.environ.pop("LPIPM_ADAT_UNITS", None)
        os.environ.pop("LPIPM_EXPERIMENTAL", None)
        rc, x, fun, it, rows, coll = solve_column_split(np.ascontiguousarray(A[:, cols.start:cols.stop]), b, c[cols.start:cols.stop], n,
                                                        0.0, opts, ctx=ctx, want_log=True)
        out.append((rc, x.tolist(), fun, it, coll.calls))
        ctx.close()
    q.put((rank, cols.start, out))
    dist.barrier()
    dist.destroy_process_group()


def test_group_by_group_reduction_of_M_is_bit_identical_to_the_one_block_reduction(built):
    """m = 1536 (12 tile rows = 3 column groups of the normal equations), two ranks: the groups of M are summed over the
    ranks one by one on the communication stream while the A.D.A^T launch is still producing the later ones
    (solver.hip enqueue_head, lpipm_ctx::grouped_reduce).  An element-wise sum of two terms: the same bits as reducing
    the whole packed triangle in one block after the launch, and more collective calls (one per group and iteration)."""
    import torch.multiprocessing as mp
    from lp_amd import synth
    from oracle import capi as oracle
    world, seed, m, n = 2, 5, 1536, 3072
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29500 + (os.getpid() * 7 + 977) % 2000
    procs = [mpc.Process(target=_worker_two_ways, args=(r, world, port, q, seed, m, n)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    A, b, c = synth.planted_lp(seed, m, n)[:3]
    ref = oracle.solve(A, b, c)
    x = np.full(n, np.nan)
    for rank, lo, (grouped, block) in got:
        assert grouped[0] == block[0] == 0 and grouped[3] == block[3] == ref["iterations"]
        assert grouped[1] == block[1] and grouped[2] == block[2]          # bit-identical x and objective
        assert grouped[4] == block[4] + 2 * grouped[3]                      # 3 groups instead of 1 block per iteration
        x[lo:lo + len(grouped[1])] = grouped[1]
    assert np.abs(x - ref["x_slack"]).max() <= 1e-6
