"""World-size-2 gloo test of the view-sharding exchange (CPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, nl, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mvlm_amd import parallel

    full = torch.arange(nl * n_total * 3, dtype=torch.float32).reshape(nl, n_total, 3)
    lo, hi = parallel.shard_range(n_total, rank, world)
    got = parallel.all_gather_views(full[:, lo:hi].contiguous(), n_total)
    poses = np.arange(12.0).reshape(4, 3) if rank == 0 else None
    poses = parallel.broadcast_array(poses, (4, 3))
    p32 = parallel.broadcast_array(np.arange(48, dtype=np.float32).reshape(8, 6) if rank == 0 else None, (8, 6))
    assert p32.dtype == np.float32 and poses.dtype == np.float64  # the 8-view table stays float32 (render3d.py:94-111)
    assert np.array_equal(p32, np.arange(48, dtype=np.float32).reshape(8, 6))
    draws = np.arange(nl * 8, dtype=np.int32).reshape(nl, 8) * 3 if rank == 0 else None
    draws = parallel.broadcast_int32(draws, (nl, 8), torch.device("cpu"))  # the RANSAC index table of rank 0
    ok_draws = draws.dtype == np.int32 and np.array_equal(draws, np.arange(nl * 8).reshape(nl, 8) * 3)
    q.put((rank, bool(torch.equal(got, full)) and ok_draws, poses.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [12, 13])
def test_all_gather_views_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(poses == np.arange(12.0).reshape(4, 3).tolist() for _, _, poses in res)
