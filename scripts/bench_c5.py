#!/usr/bin/env python3
"""BASELINE config C5: ONE dense fp64 LP (default m=16384 n=32768) column-split over the ranks, the partial
A.D.A^T panels summed by an all-reduce every iteration (lp_amd.colsplit).  Strong scaling: total work fixed.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      scripts/bench_c5.py [--rows M --cols N --steps K --warmup W --backend nccl|gloo --one-gpu]

`--backend gloo --one-gpu` rehearses the N-rank path with every rank on cuda:0 (what the GPU tests do).
Prints one JSON line on rank 0: iterations/s, bytes all-reduced per iteration, max |x - x*| of the gathered x.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--cols", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--one-gpu", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--golden", default=None, help="committed oracle vector (tests/golden/*.npz) to compare the gathered x with")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist
    import lp_amd
    from lp_amd import synth
    from lp_amd.colsplit import TorchCollective, check_split, column_range

    torch.cuda.set_device(local_rank)
    if world > 1:
        kw = {"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world, **kw)
    m, n = args.rows, args.cols
    check_split(n, world)
    A, b, c, xstar = synth.planted_lp(args.seed, m, n)               # same LP on every rank; keep only this rank's columns
    cols = column_range(n, world, rank)
    A_loc = np.ascontiguousarray(A[:, cols.start:cols.stop])
    c_loc = c[cols.start:cols.stop].copy()
    del A
    ctx = lp_amd.Context(local_rank)
    coll = TorchCollective(local_rank)
    ctx.set_collective(rank, world, coll if world > 1 else None)
    ctx.upload_column_block(A_loc, b, c_loc, n)
    opts = lp_amd.InteriorPoint.default().opts()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rc, x, fun, its, _ = ctx.solve_raw(opts)
    coll.calls = coll.bytes = 0
    iters = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rc, x, fun, its, _ = ctx.solve_raw(opts)
        if rc != 0 or coll.error is not None:
            raise RuntimeError(f"solve failed: status {rc}, collective error {coll.error}")
        iters += its
    barrier()
    dt = time.perf_counter() - t0
    err = float(np.abs(x - xstar[cols.start:cols.stop]).max())
    gold_err = None
    if args.golden:
        g = np.load(args.golden)
        gold_err = float(np.abs(x - g["x_slack"][cols.start:cols.stop]).max())
        if its != int(g["iterations"]):
            gold_err = float("inf")
    if world > 1:
        t = torch.tensor([dt, err, gold_err if gold_err is not None else 0.0], dtype=torch.float64,
                         device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, err = float(t[0]), float(t[1])
        if gold_err is not None:
            gold_err = float(t[2])
    if rank == 0:
        print(json.dumps({
            "metric": "IPM iterations/sec, one dense fp64 LP column-split over ranks", "value": iters / dt,
            "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "scaling": "strong", "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C5: one planted dense LP m={m} n={n}, columns split over {world} rank(s), "
                                   f"backend {args.backend if world > 1 else 'none'}",
                       "iterations_per_solve": iters / args.steps, "max_abs_err_vs_planted_optimum": err,
                       "max_abs_err_vs_golden": gold_err, "collective": "on-stream" if coll.on_stream else "drained",
                       "allreduce_calls_per_iteration": coll.calls / max(iters, 1),
                       "allreduce_MB_per_iteration": coll.bytes / max(iters, 1) / 1e6}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
