"""gloo tests of the view-sharding exchange on the CPU: world size 2, and the 8-rank worlds of BASELINE configs[3] / [4]."""
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


def _worker8(rank, world, port, n_total, nl, invalid, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mvlm_amd import parallel

    rs = np.random.RandomState(7)
    full = torch.from_numpy(rs.standard_normal((nl, n_total, 3)).astype(np.float32))
    lo, hi = parallel.shard_range(n_total, rank, world)
    got = parallel.all_gather_views(full[:, lo:hi].contiguous(), n_total)
    valid = np.ones(n_total, bool)
    valid[list(invalid)] = False
    mine = valid[lo:hi] if (hi > lo and not valid[lo:hi].all()) else None   # ranks without an invalid view pass None
    got_valid = parallel.all_gather_valid(mine, hi - lo, n_total, torch.device("cpu"))
    draws = rs.randint(0, n_total // 2, (nl, 8)).astype(np.int32) if rank == 0 else None
    draws = parallel.broadcast_int32(draws, (nl, 8), torch.device("cpu"))
    want_draws = np.random.RandomState(7)
    want_draws.standard_normal((nl, n_total, 3))
    ok = (torch.equal(got, full) and np.array_equal(got_valid, valid)
          and np.array_equal(draws, want_draws.randint(0, n_total // 2, (nl, 8)).astype(np.int32))
          and parallel.any_rank(rank == 5) and not parallel.any_rank(False))
    q.put((rank, bool(ok), hi - lo))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,nl,invalid", [(96, 73, ()), (128, 478, (17, 90)), (100, 84, (0, 99)), (5, 73, (2,))])
def test_collectives_in_a_world_of_eight(n_total, nl, invalid):
    """The exchange of a sharded step with EIGHT ranks, at the sizes the driver's 8-GPU run has: configs[3] (96 views, 12 per
    rank, 73 landmarks), configs[4] (128 views, 16 per rank, 478 landmarks, two views without a detection), an uneven
    split (100 views: 13 + 12 ...), and more ranks than views (5 views: three ranks hold nothing and still join)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, n_total, nl, invalid, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sum(k for _, _, k in res) == n_total
