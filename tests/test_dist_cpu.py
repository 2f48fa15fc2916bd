"""The N > 1 path on CPU ranks: world_size-2 gloo, the sharding / packing / single all-gather logic of
lp_amd.batch with an injected solver (the oracle -- test infrastructure, allowed here; the product's
default solve_fn is the HIP path).  Also the partition properties of shard_range."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from lp_amd.batch import shard_range
    for count in (0, 1, 5, 8, 255, 256, 257):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_range(count, world, r)]
            assert got == list(range(count))
            sizes = [len(shard_range(count, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from lp_amd import synth
    from lp_amd.batch import solve_batch_sharded
    from oracle import capi as oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    probs = [synth.planted_lp(s, 24 + 4 * (s % 3), 60 + 8 * (s % 2))[:3] + (0.0,) for s in range(5)]
    probs.append((np.array([[1.0, 1.0]]), np.array([-1.0]), np.array([1.0, 1.0]), 0.0))       # infeasible
    calls = []

    def solve_fn(A, b, c, c0, row):
        calls.append(A.shape)
        r = oracle.solve(A, b, c, c0)
        return r["status"], r["x_slack"], r["fun"], r["iterations"]

    res = solve_batch_sharded(probs, solve_fn=solve_fn, device=torch.device("cpu"))
    q.put((rank, len(calls), [(r["status"], r["iterations"], None if r["x_slack"] is None else r["x_slack"].tolist(),
                               r["fun"]) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_two_gloo_ranks(built):
    import torch.multiprocessing as mp
    from lp_amd import synth
    from oracle import capi as oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [3, 3]                  # 6 LPs -> 3 per rank, each solved exactly once
    assert got[0][2] == got[1][2]                         # every rank holds the whole gathered batch
    probs = [synth.planted_lp(s, 24 + 4 * (s % 3), 60 + 8 * (s % 2))[:3] for s in range(5)]
    for i, (A, b, c) in enumerate(probs):
        ref = oracle.solve(A, b, c)
        st, it, x, fun = got[0][2][i]
        assert st == ref["status"] == 0 and it == ref["iterations"]
        assert np.abs(np.array(x) - ref["x_slack"]).max() == 0.0 and fun == ref["fun"]
    assert got[0][2][5][0] == oracle.INFEASIBLE and got[0][2][5][2] is None


def test_column_range_partitions():
    from lp_amd.colsplit import column_range
    for n in (1, 127, 128, 333, 8192, 32768 + 5):
        for world in (1, 2, 3, 8):
            for align in (64, 128):
                got = [j for r in range(world) for j in column_range(n, world, r, align)]
                assert got == list(range(n))
                starts = [column_range(n, world, r, align).start for r in range(world)]
                assert all(s % align == 0 or s == n for s in starts)
