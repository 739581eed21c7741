"""Multi-GPU plumbing for the NDT core (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" on CPU).

Two ways the path spans GPUs:
  * scan-sharding  -- registrations of different scans are independent: every rank takes a slice of
    the scans against its own replica of the target grid; no collective in the data path.
  * lock-step / point-sharding -- ranks evaluate disjoint parts of the same evaluation rows and
    SUM the packed [rows][32] f64 buffer once per evaluation (ndt_set_allreduce hook).
"""
import ctypes as C

import numpy as np

EVAL_STRIDE = 32  # score, g[6], H upper triangle[21], neighbour count, 3 spare


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_row(score, g, H, n_neighbors=0.0):
    """(score, g[6], H[6x6]) -> the packed 32-f64 row the kernels produce."""
    row = np.zeros(EVAL_STRIDE)
    row[0] = score
    row[1:7] = g
    H = np.asarray(H).reshape(6, 6)
    k = 7
    for i in range(6):
        for j in range(i, 6):
            row[k] = H[i, j]
            k += 1
    row[28] = n_neighbors
    return row


def unpack_row(row):
    H = np.zeros((6, 6))
    k = 7
    for i in range(6):
        for j in range(i, 6):
            H[i, j] = H[j, i] = row[k]
            k += 1
    return float(row[0]), np.array(row[1:7]), H, float(row[28])


def make_allreduce(group=None):
    """Callback for NormalDistributionsTransform.setAllreduce(): in-place SUM over `group`.

    Host buffers are reduced through a CPU tensor (gloo); device buffers are wrapped zero-copy via
    __cuda_array_interface__ and reduced by RCCL."""
    import torch
    import torch.distributed as dist

    class _DevView:
        def __init__(self, addr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (addr, False), "version": 2}

    def fn(addr, n, on_device):
        if on_device:
            t = torch.as_tensor(_DevView(addr, n), device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize()
        else:
            buf = (C.c_double * n).from_address(addr)
            t = torch.from_numpy(np.ctypeslib.as_array(buf))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    return fn
