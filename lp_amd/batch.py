"""Batches of independent LPs sharded one-shard-per-GPU (BASELINE config C4 / SURVEY.md 8e).

LPs are independent, so the path shards with NO data-path collective: rank r solves the contiguous
block `shard_range(count, world, r)` on its own GPU (its own lpipm_ctx / stream), and the batch ends
with exactly ONE collective -- an all-gather (RCCL over xGMI when the backend is "nccl") of a packed
[shard_max, n_max + 3] DEVICE block per rank holding x / tau, fun, iterations and status of each LP.
On the GPU the shard goes to lpipm_solve_batch_device in one call: LPs of equal shape advance as lockstep
batches (one kernel launch covers all of them), odd shapes one at a time, and every member's x / tau is
copied device to device into its row of the packed block -- no solution vector visits the host before the
gather (only the 3 scalars per LP do).
One process per GPU, `torch.distributed` for the plumbing; nothing here computes on the CPU.
`solve_fn` exists so that the sharding / packing / gather logic can be unit-tested on CPU ranks
(gloo) with an injected solver; the default is the HIP path and it fails loudly without a GPU.
"""
from __future__ import annotations

import numpy as np

from . import _capi


def shard_range(count: int, world: int, rank: int) -> range:
    """Static block partition: the first `count % world` ranks get one extra LP."""
    base, extra = divmod(count, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))



def solve_batch_sharded(problems, opts=None, ctx=None, group=None, device=None, solve_fn=None):
    """problems: sequence of (A, b, c, c0) -- every rank passes the same list (or at least its shard
    at the right indices).  Returns, on EVERY rank, a list of dicts {status, x_slack, fun, iterations}
    in problem order.  `solve_fn(A, b, c, c0, None) -> (status, x | None, fun, iterations)` replaces the
    library call in the CPU-rank tests."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    count = len(problems)
    mine = shard_range(count, world, rank)
    shard_max = -(-count // world) if count else 0
    n_max = max((np.asarray(p[2]).shape[0] for p in problems), default=0)
    rows = max(shard_max, 1)
    meta = np.zeros((rows, 3))
    meta[:, 2] = -1.0                                 # status -1: padding slot, no LP here
    if solve_fn is None:                              # the product path: the whole shard in one library call
        import lp_amd
        ctx = ctx or lp_amd.default_context(device.index if device is not None and device.index is not None else 0)
        opts = opts or lp_amd.InteriorPoint.default().opts()
        device = device or torch.device("cuda", ctx.device)
        packed = torch.zeros((rows, n_max + 3), dtype=torch.float64, device=device)
        torch.cuda.synchronize(device)                # the zero fill runs on torch's stream, the solver on its own
        res = ctx.solve_batch_device([problems[i] for i in mine], opts, packed.data_ptr(), n_max + 3)
        for slot, (rc, fun, it) in enumerate(res):
            meta[slot] = (fun if fun is not None else float("nan"), float(it), float(rc))
        packed[:, n_max:] = torch.from_numpy(meta).to(device)       # 3 scalars per LP; x rows never left the device
    else:                                             # CPU-rank tests: injected solver, host rows
        device = device or torch.device("cpu")
        host = np.zeros((rows, n_max + 3))
        for slot, i in enumerate(mine):
            A, b, c, c0 = problems[i]
            n = np.asarray(c).shape[0]
            rc, x, fun, it = solve_fn(A, b, c, c0, None)
            if x is not None:
                host[slot, :n] = np.asarray(x, dtype=np.float64)
            ok = rc in (_capi.OK, _capi.ITERATION_LIMIT)
            meta[slot] = (fun if ok and fun is not None else float("nan"), float(it), float(rc))
        host[:, n_max:] = meta
        packed = torch.from_numpy(host).to(device)
    if world > 1:
        flat = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(flat, packed, group=group)         # the single collective of the batch
        gathered = flat.view(world, packed.shape[0], packed.shape[1])
    else:
        gathered = packed.unsqueeze(0)
    g = gathered.cpu().numpy()
    out = []
    for i in range(count):
        r = next(rr for rr in range(world) if i in shard_range(count, world, rr))
        slot = i - shard_range(count, world, r).start
        row = g[r, slot]
        n = np.asarray(problems[i][2]).shape[0]
        status = int(row[n_max + 2])
        has_x = status in (_capi.OK, _capi.ITERATION_LIMIT)
        out.append(dict(status=status, x_slack=row[:n].copy() if has_x else None,
                        fun=float(row[n_max]) if has_x else None, iterations=int(row[n_max + 1])))
    return out
