"""View sharding across the GPUs of one node (one process per GPU, torch.distributed).

Views are independent through render, network and maxima; the only
cross-view step is the per-landmark consensus, which needs every view's
(row, col, score).  So each rank handles a contiguous slice of the pose table and
one all-gather (RCCL over xGMI when the backend is "nccl") of the small
``[NL, n_local, 3]`` maxima precedes fusion.  View order is preserved because the
quantile filter and the RANSAC draws address views by position.
"""
from __future__ import annotations

import numpy as np


def is_distributed() -> bool:
    try:
        import torch.distributed as dist
    except Exception:  # noqa: BLE001
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world() -> tuple[int, int]:
    if not is_distributed():
        return 0, 1
    import torch.distributed as dist

    return dist.get_rank(), dist.get_world_size()


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) of ``n`` items for ``rank``; sizes differ by at most one."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_array(arr: np.ndarray | None, src: int = 0) -> np.ndarray:
    """Broadcast a small host array (pose table, RANSAC draws) from ``src`` to all ranks."""
    if not is_distributed():
        return arr
    import torch.distributed as dist

    box = [arr]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_int32(arr: np.ndarray | None, shape: tuple, device, src: int = 0) -> np.ndarray:
    """Broadcast a small int32 table (the RANSAC draws) as ONE tensor collective - no pickling."""
    if not is_distributed():
        return arr
    import torch
    import torch.distributed as dist

    on_gpu = dist.get_backend() != "gloo"
    t = torch.zeros(shape, dtype=torch.int32, device=device if on_gpu else "cpu")
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def all_gather_views(local, n_total: int):
    """local: torch tensor [NL, n_local, 3] on this rank's device -> [NL, n_total, 3] in view order."""
    if not is_distributed():
        return local
    import torch
    import torch.distributed as dist

    rank, world = rank_world()
    nl = local.shape[0]
    n_max = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    # fixed-size slots: [n_max, NL, 3] per rank, padded; one collective
    send = torch.zeros((n_max, nl, 3), dtype=local.dtype, device=local.device)
    send[: local.shape[1]] = local.permute(1, 0, 2)
    recv = torch.empty((world * n_max, nl, 3), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo" and send.is_cuda:
        # rehearsal on one GPU (several ranks share a device, which RCCL refuses): stage through the host
        r_cpu = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r_cpu, send.cpu().contiguous())
        recv.copy_(r_cpu)
    else:
        dist.all_gather_into_tensor(recv, send.contiguous())
    parts = []
    for r in range(world):
        s, e = shard_range(n_total, r, world)
        parts.append(recv[r * n_max: r * n_max + (e - s)])
    return torch.cat(parts, 0).permute(1, 0, 2).contiguous()
