#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec on the headline workload of BASELINE.json (dense 4096x8192 fp64 LP).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one complete `InteriorPoint::solve` (mod.rs:199-240) of the rank's LP: the whole hot path
(A.D.A^T, Cholesky, triangular solves, GEMVs, vector kernels) for as many IPM iterations as the
solver needs, with A already resident in HBM (uploaded once before the timed region).
Weak scaling: every rank owns ONE independent LP of the same shape (seed = rank) -- the batch shards
one per GPU with no data-path collective -- and the timed region ends with the single RCCL
all-gather of the solutions.  value = IPM iterations of all ranks / max-over-ranks wall time.

`--workload c4` (optional, not the default): BASELINE config 4 instead -- 32 independent 1024x2048 LPs per GPU as one
lockstep batch with resident inputs, value in LP/s; same timing protocol, no roofline / cpu_baseline objects.

Extra objects on the JSON line:
  roofline     : the dominant kernel (A.D.A^T, MFMA-bound): algorithmic flops m(m+1)n per launch /
                 its average launch duration from HIP events recorded on the solver's own stream
                 INSIDE the timed region (2 events per iteration, bracketing that kernel and its fix-up);
                 peak = 78.6 TFLOP/s dense fp64 MFMA.  The per-phase breakdown comes from one extra,
                 untimed solve with every phase bracketed.
  cpu_baseline : rank 0, N == 1 only: the single-threaded C restatement of the reference
                 (oracle/, kind "port") timed on this box's host for ONE IPM iteration of the same LP
                 (every iteration performs the same operations, so 1 / t is its iterations/sec).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_RESULT_FD = 1


def emit(obj):
    os.write(_RESULT_FD, (json.dumps(obj) + "\n").encode())


PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (probe: lpipm_k_mfma_f64_probe ~76 TF/s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=("c3", "c4"), default="c3",
                    help="c3 (default, the BASELINE metric): one 4096x8192 LP per GPU; c4: a shard of 32 independent "
                         "1024x2048 LPs per GPU as one lockstep batch (BASELINE config 4: 256 LPs over 8 GPUs)")
    args = ap.parse_args()

    # STDOUT carries exactly one line, the JSON result.  Libraries that print to file descriptor 1 on their own (RCCL's
    # version banner under NCCL_DEBUG=VERSION, warnings) are sent to stderr for the lifetime of the process.
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import lp_amd
    from lp_amd import synth

    dist = None
    if world > 1 or os.environ.get("LPIPM_BENCH_FORCE_DIST") == "1":   # the latter: rehearse the N>1 path on one GPU
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")          # only matter for the one-rank rehearsal:
        os.environ.setdefault("MASTER_PORT", "29517")              # torch.distributed.run sets both
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    if args.workload == "c4":
        return bench_c4(args, rank, local_rank, world, dist, dev, np, torch, lp_amd, synth)
    m, n = args.m, args.n
    A, b, c, xstar = synth.planted_lp(rank, m, n)        # one independent LP per rank (seed = rank)
    ctx = lp_amd.Context(local_rank)
    ctx.upload_arrays(A, b, c)                           # one-time H2D, outside the timed region
    solver = lp_amd.InteriorPoint.default()              # reference defaults (mod.rs:52-59)
    opts = solver.opts()
    x_dev = torch.zeros(n, dtype=torch.float64, device=dev)
    gathered = torch.zeros(world * n, dtype=torch.float64, device=dev) if dist is not None else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        rc, _, fun, its, _ = ctx.solve_raw(opts, x_dev_ptr=x_dev.data_ptr())
        if rc != 0:
            raise RuntimeError(f"solve failed with status {rc}")
        if dist is not None:                             # the single RCCL gather of the batch's solutions
            dist.all_gather_into_tensor(gathered, x_dev)
        return its, fun

    for _ in range(args.warmup):
        one_step()
    ctx.set_profiling(2)                                 # HIP events around the dominant kernel only (2 per
                                                         # iteration), recorded on the solver's own stream
    adat_ms = 0.0
    adat_launches = 0
    phase = {k: 0.0 for k in ("adat_ms", "potrf_ms", "trsv_ms", "gemv_ms", "vec_ms", "total_ms")}
    iters_local = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        its, fun = one_step()
        iters_local += its
        pt = ctx.phase_times()
        adat_ms += pt["adat_ms"]
        adat_launches += pt["adat_launches"]
    barrier()
    dt = time.perf_counter() - t0
    # per-phase breakdown: one more solve with every phase bracketed by events, OUTSIDE the timed region
    ctx.set_profiling(1)
    ctx.solve_raw(opts, x_dev_ptr=x_dev.data_ptr())
    pt = ctx.phase_times()
    phase = {k: pt[k] for k in phase}
    phase_iters = max(int(pt["iterations"]), 1)
    ctx.set_profiling(0)

    # parity guard on what was just timed: the planted vertex is the known answer
    err = float((x_dev.cpu().numpy() - xstar).__abs__().max())

    if dist is not None:
        t = torch.tensor([dt, float(iters_local)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, iters_total = float(tmax[0]), float(tsum[1])
    else:
        dt_max, iters_total = dt, float(iters_local)

    if rank == 0:
        flops_per_launch = float(m) * (m + 1) * n                     # lower triangle of A.D.A^T
        avg_ms = adat_ms / max(adat_launches, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # HBM-side bytes per launch of the dominant kernel come from a separate rocprofv3 --pmc run (counters
        # cannot be read from inside this process); the committed summary is profiles/r01_adat_pmc.json.
        traffic = None
        mfma_busy = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_adat_pmc.json")
        if (m, n) == (4096, 8192) and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            traffic = pmc.get("traffic_bytes_per_launch")
            mfma_busy = pmc.get("mfma_busy_fraction")
        out = {
            "metric": "IPM iterations/sec, dense 4096x8192 fp64 LP",
            "value": iters_total / dt_max,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{'C3' if (m, n) == (4096, 8192) else 'custom'}: random dense planted LP m={m} n={n} fp64, one independent LP per GPU "
                                   f"(seed = rank), reference default options, A resident in HBM",
                       "m": m, "n": n, "iterations_per_solve": iters_local / args.steps,
                       "max_abs_err_vs_planted_optimum": err},
            "roofline": {"kernel": "gemm_nt_streamk_w8_kernel<true> + fix-up (A.diag(x/z).A^T, lower 128x128 tiles, "
                                   "v_mfma_f64_16x16x4_f64)",
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, "
                                                             "profiles/r01_adat_pmc.json)",
                         "algorithmic_bytes_per_launch": 8.0 * m * n + 4.0 * m * m,
                         "mfma_busy_pmc": mfma_busy,   # SQ_VALU_MFMA_BUSY_CYCLES share of the kernel's SIMD-cycles (same JSON)
                         "avg_launch_ms": avg_ms, "launches": adat_launches,
                         "flops_per_launch": flops_per_launch},
            "phase_ms_per_iteration": {k: v / phase_iters for k, v in phase.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import capi as oracle                        # checker / baseline only
            o = oracle.default_opts(max_iter=1)
            t1 = time.perf_counter()
            r = oracle.solve(A, b, c, 0.0, o, want_log=False)
            tc = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": 1.0 / r["timing"]["total"], "unit": "iterations/s", "cores": 1, "kind": "port",
                "sample": f"1 IPM iteration (the first) of the same {m}x{n} LP with the single-threaded C "
                          f"restatement of the reference (as-written op counts); {tc:.1f} s of CPU work; "
                          f"every iteration performs the same operations",
                "phase_s": r["timing"],
            }
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_c4(args, rank, local_rank, world, dist, dev, np, torch, lp_amd, synth):
    """BASELINE config 4: 32 independent 1024x2048 LPs per GPU (256 over 8), solved as ONE lockstep batch with the inputs
    resident in HBM, then the single gather of the solutions.  value = LPs of all ranks / max-over-ranks wall time."""
    import time
    per_rank, m, n = 32, 1024, 2048
    probs = [synth.planted_lp(rank * per_rank + s, m, n) for s in range(per_rank)]
    ctx = lp_amd.Context(local_rank)
    ctx.upload_lockstep([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs])   # untimed H2D
    opts = lp_amd.InteriorPoint.default().opts()
    gathered = torch.zeros(world * per_rank * n, dtype=torch.float64, device=dev) if dist is not None else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step():
        res = ctx.solve_lockstep(opts)
        if any(r[0] != 0 for r in res):
            raise RuntimeError("a member of the shard did not solve")
        if dist is not None:                              # the single RCCL gather of the batch's solutions
            xs = torch.from_numpy(np.stack([r[1] for r in res]).reshape(-1)).to(dev)
            dist.all_gather_into_tensor(gathered, xs)
        return res

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_step()
    barrier()
    dt = time.perf_counter() - t0
    its = sum(r[3] for r in res)
    err = max(float(np.abs(r[1] - p[3]).max()) for r, p in zip(res, probs))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    if rank == 0:
        emit({
            "metric": "independent LPs solved per second, batch of 1024x2048 fp64 LPs sharded 32 per GPU", "value": world * per_rank * args.steps / dt,
            "unit": "LP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C4: {world * per_rank} independent planted LPs m={m} n={n} fp64, {per_rank} per GPU as one lockstep batch, "
                                   "inputs resident in HBM, one all-gather of the solutions",
                       "iterations_per_lp": its / per_rank, "max_abs_err_vs_planted_optimum": err}})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
