"""One LP split by columns over ranks (BASELINE config C5 / SURVEY.md 8e "n-split").

Rank g holds A[:, J_g], c[J_g] and the matching slices of x and z; b and y are replicated.  Per
iteration the ranks sum their partial normal equations M_g = A_g D_g A_g^T (the all-reduce on the
A.D.A^T panels), the m-vectors A_g w_g, and a handful of scalars (dots over n, ratio-test minima);
the Cholesky factorisation and the m-sized solves run replicated.  liblpipm.so does all the device
work and calls back for each reduction (include/lpipm.h `lpipm_set_collective`); this module supplies
that callback from `torch.distributed` (backend "nccl" = RCCL over xGMI on a GPU node; "gloo" in the
tests) -- plumbing only, nothing here computes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


def column_range(n: int, world: int, rank: int, align: int = 128) -> range:
    """Contiguous column block of `rank`: whole `align`-column groups (the GEMM k-tile) dealt as evenly
    as possible, the ragged tail going to the last rank that has any columns."""
    groups = -(-n // align)
    base, extra = divmod(groups, world)
    lo_g = rank * base + min(rank, extra)
    hi_g = lo_g + base + (1 if rank < extra else 0)
    return range(min(lo_g * align, n), min(hi_g * align, n))


class _DevPtr:
    """A raw device pointer dressed as a __cuda_array_interface__ array so torch can alias it."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class TorchCollective:
    """lpipm_allreduce_fn over a torch.distributed process group.

    Default: the DRAINED contract (the library drains its stream, the callback returns with the result in place).
    on_stream=True (or LPIPM_COLLECTIVE_ON_STREAM=1 in the environment): the library does not drain its stream before
    calling (lpipm_set_collective_on_stream); the all-reduce is issued with the SOLVER's stream as torch's current stream,
    so RCCL orders it behind the kernels that produced the operand and the kernels that follow wait for it on the device
    -- the host never blocks.  OPT-IN because it has so far only been exercised with gloo ranks (where the callback
    synchronises the stream itself) and with ONE RCCL rank; no run with >= 2 RCCL ranks has been recorded
    (INTEGRATION.md)."""

    def __init__(self, device: int, group=None, on_stream=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, int(device)
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        import os
        env = os.environ.get("LPIPM_COLLECTIVE_ON_STREAM") == "1"
        self.on_stream = (backend == "nccl" and env) if on_stream is None else bool(on_stream)
        self.calls = 0
        self.bytes = 0
        self.error = None
        self.cfn = _capi.ALLREDUCE_FN(self._call)     # keep the thunk alive as long as the ctx uses it

    def _call(self, _user, ptr, count, op, stream):
        try:
            torch, dist = self.torch, self.dist
            t = torch.as_tensor(_DevPtr(ptr, count), device=torch.device("cuda", self.device))
            rop = dist.ReduceOp.MIN if op == 1 else dist.ReduceOp.SUM
            if self.on_stream:
                ext = torch.cuda.ExternalStream(int(stream), device=torch.device("cuda", self.device))
                if dist.get_backend(self.group) != "nccl":
                    ext.synchronize()                              # a host-staged backend reads the operand now
                with torch.cuda.stream(ext):
                    dist.all_reduce(t, op=rop, group=self.group)   # RCCL: enqueued behind / ahead of the solver's kernels
                self.calls += 1
                self.bytes += 8 * int(count)
                return 0
            dist.all_reduce(t, op=rop, group=self.group)
            torch.cuda.current_stream(self.device).synchronize()   # result in place before the library resumes
            self.calls += 1
            self.bytes += 8 * int(count)
            return 0
        except Exception as e:  # never unwind through the C frames
            self.error = e
            return 1


def check_split(n: int, world: int, align: int = 128):
    """Every rank needs at least one column group: with fewer groups than ranks some ranks would hold nothing, refuse
    their upload, and leave the others waiting in the first all-reduce.  Raises the same error on every rank."""
    groups = -(-int(n) // align)
    if world > groups:
        raise ValueError(f"cannot split {n} columns ({groups} groups of {align}) over {world} ranks: use at most {groups}")


def solve_column_split(A_local, b, c_local, n_total: int, c0: float = 0.0, opts=None, ctx=None, group=None,
                       want_log: bool = False, on_stream=None):
    """Solve min c.x s.t. A x = b, x >= 0 with this rank holding the column block (A_local, c_local).
    Every rank must call this with the same b / n_total / opts.  Returns
    (status, x_local, fun, iterations, log rows, collective) -- x_local is this rank's slice of x / tau."""
    import torch
    import torch.distributed as dist
    import lp_amd

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    check_split(n_total, world)                       # the same verdict on every rank, before anything is uploaded
    ctx = ctx or lp_amd.Context(torch.cuda.current_device())
    opts = opts or lp_amd.InteriorPoint.default().opts()
    coll = TorchCollective(ctx.device, group, on_stream=on_stream)
    ctx.set_collective(rank, world, coll)
    ctx.upload_column_block(A_local, b, c_local, n_total, c0)
    rc, x, fun, it, rows = ctx.solve_raw(opts, want_log=want_log)
    if coll.error is not None:
        raise coll.error
    return rc, x, fun, it, rows, coll
