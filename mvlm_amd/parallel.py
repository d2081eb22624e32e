"""View sharding across the GPUs of one node (one process per GPU, torch.distributed).

Views are independent through render, network and maxima; the only
cross-view step is the per-landmark consensus, which needs every view's
(row, col, score).  So each rank handles a contiguous slice of the pose table and
one all-gather (RCCL over xGMI when the backend is "nccl") of the small
``[NL, n_local, 3]`` maxima precedes fusion.  View order is preserved because the
quantile filter and the RANSAC draws address views by position.
"""
from __future__ import annotations

import os

import numpy as np


# Optional timing of the two collectives of a sharded step (bench.py's scaling breakdown): enable_timing(True) makes
# all_gather_views / broadcast_int32 bracket their collective - HIP events on the rank's current stream under RCCL (torch
# orders the collective's stream against it on both sides), wall clock around a drained GPU under gloo - and
# timing_summary() reports milliseconds per call.  Off by default: the product path records nothing.
_TIMING = {"on": False, "all_gather": [], "broadcast": []}


def enable_timing(on: bool) -> None:
    _TIMING["on"] = bool(on)
    _TIMING["all_gather"].clear()
    _TIMING["broadcast"].clear()


class _Timed:
    def __init__(self, kind: str, on_device: bool):
        self.kind, self.on_device, self.active = kind, on_device, _TIMING["on"]

    def __enter__(self):
        if not self.active:
            return self
        import time

        import torch

        if self.on_device:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            if torch.cuda.is_available():
                torch.cuda.synchronize()  # the host-staged form would otherwise count the GPU work it waits for
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if not self.active or exc[0] is not None:
            return False
        import time

        if self.on_device:
            self.e1.record()
            _TIMING[self.kind].append((self.e0, self.e1))
        else:
            _TIMING[self.kind].append(1e3 * (time.perf_counter() - self.t0))
        return False


def timing_summary() -> dict:
    """Mean milliseconds per call of the collectives recorded since enable_timing(True) (synchronises the device)."""
    out = {}
    for kind in ("all_gather", "broadcast"):
        vals = []
        for rec in _TIMING[kind]:
            if isinstance(rec, tuple):
                rec[1].synchronize()
                vals.append(rec[0].elapsed_time(rec[1]))
            else:
                vals.append(rec)
        out[kind] = {"calls": len(vals), "mean_ms": (sum(vals) / len(vals)) if vals else None,
                     "clock": "hip events on the rank's stream" if any(isinstance(r, tuple) for r in _TIMING[kind]) else "host wall clock"}
    return out


def rccl_environment(env=None) -> dict:
    """Environment a process needs BEFORE it initialises the GPU to take part in a multi-process RCCL job on this
    platform: HSA_ENABLE_IPC_MODE_LEGACY=0 selects dmabuf IPC for ROCr's inter-process memory handles.  Source: the
    deployment image this package is built for exports exactly that (its host kernel driver supports only dmabuf IPC;
    with the legacy mode RCCL's and torch's cross-process buffer sharing fail with ``hipIpcGetMemHandle: invalid
    argument``) - so a launcher that builds its children's environment from scratch must carry it over, and one that
    inherits the image's environment gets the same value.  A value the caller has already set is kept.
    Used by every place that starts RCCL ranks: bench.py's launcher, the nccl workers of the tests, INTEGRATION.md's recipe."""
    env = os.environ if env is None else env
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def is_distributed() -> bool:
    """A process group with more than one rank exists.  MVLM_DIST_WORLD_OF_ONE=1 also counts a group of ONE rank: the
    rehearsal of the sharded path (same collectives, tensors and devices) over RCCL on a one-GPU box."""
    try:
        import torch.distributed as dist
    except Exception:  # noqa: BLE001
        return False
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MVLM_DIST_WORLD_OF_ONE") == "1"


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


def _collective_device(device):
    """Where a tensor collective's payload must live: the rank's own GPU under RCCL ("nccl"), the host under gloo."""
    import torch
    import torch.distributed as dist

    if dist.get_backend() == "gloo":
        return torch.device("cpu")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)


def broadcast_array(arr: np.ndarray | None, shape: tuple, device=None, src: int = 0) -> np.ndarray:
    """Broadcast a small host table of known shape (the pose table) from ``src`` as ONE tensor collective on
    ``device`` (this rank's GPU under RCCL) - no pickling, no staging on whatever GPU happens to be current.
    float32 / float64 tables keep their dtype (the 8-view table is float32 in the reference, render3d.py:94-111,
    and the estimator's rotation arithmetic follows that dtype)."""
    if not is_distributed():
        return arr
    import torch
    import torch.distributed as dist

    n = int(np.prod(shape))
    t = torch.zeros(n + 1, dtype=torch.float64, device=_collective_device(device))
    if dist.get_rank() == src:
        a = np.asarray(arr)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"broadcast_array: table is {a.shape}, announced {tuple(shape)}")
        host = np.empty(n + 1, np.float64)
        host[:n] = a.astype(np.float64).ravel()
        host[n] = 1.0 if a.dtype == np.float32 else 0.0
        t.copy_(torch.from_numpy(host))
    dist.broadcast(t, src=src)
    host = t.cpu().numpy()
    out = host[:n].reshape(shape)
    return out.astype(np.float32) if host[n] == 1.0 else out


def broadcast_int32(arr: np.ndarray | None, shape: tuple, device, src: int = 0, keep_on_device: bool = False):
    """Broadcast a small int32 table (the RANSAC draws) as ONE tensor collective - no pickling.
    ``keep_on_device``: under RCCL return the device tensor itself (the consumer is a kernel on this rank's GPU: no
    copy back to the host, no wait for the collective on the host); under gloo the host array as always."""
    if not is_distributed():
        return arr
    import torch
    import torch.distributed as dist

    t = torch.zeros(shape, dtype=torch.int32, device=_collective_device(device))
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)))
    with _Timed("broadcast", t.is_cuda):
        dist.broadcast(t, src=src)
    if keep_on_device and t.is_cuda:
        return t
    return t.cpu().numpy()


def any_rank(flag: bool, device=None) -> bool:
    """Is ``flag`` set on ANY rank?  (one MAX all-reduce of a scalar; every rank must call it.)  Decisions that change which
    collectives a rank will issue next - repeating a pass - have to be the same on every rank."""
    if not is_distributed():
        return bool(flag)
    import torch
    import torch.distributed as dist

    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_collective_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def all_gather_valid(valid_local: np.ndarray | None, n_local: int, n_total: int, device=None) -> np.ndarray:
    """Per-view validity of a sharded step (a detector that found nothing in some views, mediapipepredictor.py:38-41):
    every rank's host mask [n_local] (None = all valid) -> the job's mask [n_total] in view order, one small tensor
    collective.  Every rank must call it."""
    mine = np.ones(n_local, bool) if valid_local is None else np.asarray(valid_local, bool)
    if not is_distributed():
        return mine
    import torch
    import torch.distributed as dist

    rank, world = rank_world()
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    n_max = max(sizes)
    dev = _collective_device(device)
    send = torch.ones(n_max, dtype=torch.int32)
    send[: len(mine)] = torch.from_numpy(mine.astype(np.int32))
    send = send.to(dev)
    recv = torch.empty(world * n_max, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    host = recv.cpu().numpy()
    return np.concatenate([host[r * n_max: r * n_max + sizes[r]] for r in range(world)]).astype(bool)


def all_gather_views(local, n_total: int):
    """local: torch tensor [NL, n_local, 3] on this rank's device -> [NL, n_total, 3] in view order."""
    if not is_distributed():
        return local
    import torch
    import torch.distributed as dist

    rank, world = rank_world()
    nl = local.shape[0]
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    n_max = max(sizes)
    even = min(sizes) == n_max          # every rank holds the same number of views: no padding, no per-rank slicing
    # fixed-size slots: [n_max, NL, 3] per rank (padded when the shards are uneven); one collective
    if even:
        send = local.permute(1, 0, 2).contiguous()
    else:
        send = torch.zeros((n_max, nl, 3), dtype=local.dtype, device=local.device)
        send[: local.shape[1]] = local.permute(1, 0, 2)
    recv = torch.empty((world * n_max, nl, 3), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "gloo" and send.is_cuda:
        # rehearsal on one GPU (several ranks share a device, which RCCL refuses): stage through the host
        with _Timed("all_gather", False):
            r_cpu = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(r_cpu, send.cpu().contiguous())
            recv.copy_(r_cpu)
    else:
        with _Timed("all_gather", send.is_cuda):
            dist.all_gather_into_tensor(recv, send)
    if even:
        return recv.permute(1, 0, 2).contiguous()
    parts = []
    for r in range(world):
        parts.append(recv[r * n_max: r * n_max + sizes[r]])
    return torch.cat(parts, 0).permute(1, 0, 2).contiguous()
