"""Round 6, -m gpu: what changed in the render / fusion step for configs[4] (0.55 -> 0.44 ms per 128-view mesh) keeps every result."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _face(grid, seed=1, tex=64):
    from mvlm_amd.utils.synthetic import face_like_mesh

    return face_like_mesh(grid=grid, tex_size=tex, seed=seed)


def test_key_plane_is_left_clean_between_renders():
    """The tile kernel hands the depth-key plane back EMPTY instead of a fill in front of every render: a scene that covers
    much, then one that covers little, more views, fewer views, another mesh - each bit-equal to the oracle, which starts from
    a fresh z-buffer every time."""
    from mvlm_amd.utils import HipRenderer3D, Mesh
    from oracle import raster

    big = Mesh(np.array([[-400, -400, 0], [400, -400, 0], [0, 600, 0]], np.float32), np.array([[0, 1, 2]], np.int32))
    small = Mesh(np.array([[-10, -10, 50], [10, -10, 50], [0, 12, 50]], np.float32), np.array([[0, 1, 2]], np.int32))
    face = _face(60)
    r = HipRenderer3D(n_views=8, verbose=False)
    rs = np.random.RandomState(5)
    for mesh, n in ((big, 8), (small, 8), (face, 24), (small, 3), (face, 8), (big, 40), (small, 40)):
        poses = np.stack([rs.randint(-40, 40, n), rs.randint(-80, 80, n), rs.randint(-20, 20, n)], 1).astype(np.float64)
        got = r.render_device(mesh, poses).cpu().numpy()
        r.check()
        want = raster.multiview_render(mesh.verts, mesh.tris, mesh.uvs, mesh.texture, poses)
        np.testing.assert_array_equal(got, want)


def test_rays_from_the_rasterisers_own_rotation_table():
    """lines_device with the table the last render left on the device == with a table uploaded for it (bit for bit), and the
    borrowed table follows the next render."""
    from mvlm_amd.utils import HipEstimator3D, HipRenderer3D

    r = HipRenderer3D(n_views=96, verbose=False)
    e = HipEstimator3D(verbose=False)
    assert r.ctx is e.ctx
    mesh = _face(30)
    rs = np.random.RandomState(2)
    for seed in (4, 9):
        np.random.seed(seed)
        poses = r.generate_3d_transformations()
        maxima = torch.from_numpy(rs.uniform(0, 255, (73, 96, 3)).astype(np.float32)).cuda()
        r.render_device(mesh, poses)
        s1, e1 = e.lines_device(maxima, poses, 256, rot_dev=r.rotations_device())
        s2, e2 = e.lines_device(maxima, poses, 256)
        assert torch.equal(s1, s2) and torch.equal(e1, e2)
        assert float(s1.abs().max()) > 100.0


def test_fused_step_equals_the_slot_protocol_after_the_reordering(tmp_path):
    """predict_mesh_device (render first, rotations borrowed from the rasteriser, results through pinned memory, stream bound
    once) against the reference's slot-by-slot protocol on host arrays, twice in a row (buffers and staging slots reused)."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.mesh_io import load_obj
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "f.obj", grid=41, tex_size=64, seed=2)
    pipe = pipeline.create_pipeline("dtu3d", n_views=12, weights="synthetic:5", verbose=False)
    mesh = load_obj(obj)
    for seed in (3, 4):
        np.random.seed(seed)
        poses = pipe.renderer_3d.generate_3d_transformations()
        state = np.random.get_state()
        fused, err = pipe.predict_mesh_device(mesh, poses)
        np.random.set_state(state)
        images = pipe.renderer_3d.render_device(mesh, poses).cpu().numpy()
        lms, valid = pipe.predictor_2d.predict_landmarks_from_images(images)
        starts, ends = pipe.estimator_3d.estimate_landmark_lines(images, lms, poses)
        raw, err2 = pipe.estimator_3d.estimate_landmarks_from_lines(lms, starts, ends)
        slots = pipe.estimator_3d.project_landmarks_to_surface(mesh, raw)
        np.testing.assert_array_equal(fused, slots)
        assert err == err2


# ---- the sharded path's exchange through the C ABI (mvlm_allgather_maxima) ------------------------------------------------
def _gather(ctx, comm, rank, world, local, n_total, nl):
    import ctypes as C

    out = torch.full((nl, n_total, 3), -7.0, dtype=torch.float32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rc = ctx.lib.mvlm_allgather_maxima(ctx.handle, comm, rank, world, C.c_void_p(local.data_ptr()) if local is not None else None,
                                       n_total, nl, C.c_void_p(out.data_ptr()))
    return rc, out


def test_allgather_maxima_world_of_one_without_a_communicator():
    from mvlm_amd import _lib

    ctx = _lib.get_context(0)
    rs = np.random.RandomState(3)
    for nl, n in ((73, 12), (478, 128), (84, 1)):
        local = torch.from_numpy(rs.standard_normal((nl, n, 3)).astype(np.float32)).cuda()
        rc, out = _gather(ctx, None, 0, 1, local, n, nl)
        assert rc == 0 and torch.equal(out, local)
    rc, _ = _gather(ctx, None, 0, 2, local, 2, 84)      # more than one rank needs a communicator
    assert rc != 0 and b"communicator" in ctx.lib.mvlm_last_error(ctx.handle)


@pytest.mark.parametrize("n_total,nl,world", [(96, 73, 8), (128, 478, 8), (100, 84, 8), (5, 73, 8), (96, 84, 2), (7, 84, 3)])
def test_gather_pack_and_unpack_for_every_rank_of_a_world(n_total, nl, world):
    """The layout of the exchange with EVERY rank played on this GPU: each rank's shard packed into its slot, the slots unpacked
    = the full tensor in view order - even, uneven and more-ranks-than-views splits; the same answer as the Python host's
    torch.distributed path (mvlm_amd/parallel.py all_gather_views, whose shard_range this follows)."""
    import ctypes as C

    from mvlm_amd import _lib, parallel

    ctx = _lib.get_context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rs = np.random.RandomState(n_total + world)
    full = torch.from_numpy(rs.standard_normal((nl, n_total, 3)).astype(np.float32)).cuda()
    n_max = -(-n_total // world)
    slots = torch.full((world, n_max, nl, 3), 9.0, dtype=torch.float32, device="cuda")
    for r in range(world):
        lo, hi = parallel.shard_range(n_total, r, world)
        local = full[:, lo:hi].contiguous()
        assert ctx.lib.mvlm_gather_pack(ctx.handle, C.c_void_p(local.data_ptr()) if hi > lo else None, hi - lo, n_max, nl,
                                        C.c_void_p(slots[r].data_ptr())) == 0
        assert torch.equal(slots[r, :hi - lo], local.permute(1, 0, 2)) and not slots[r, hi - lo:].any()
    out = torch.empty_like(full)
    assert ctx.lib.mvlm_gather_unpack(ctx.handle, C.c_void_p(slots.data_ptr()), world, n_total, nl, C.c_void_p(out.data_ptr())) == 0
    assert torch.equal(out, full)


def test_allgather_maxima_over_rccl_in_a_world_of_one():
    """A real ncclComm_t (RCCL's own library through ctypes, one rank on this GPU) handed to the C entry point: pack,
    ncclAllGather on the context's stream, unpack = the identity on this rank's views."""
    import ctypes as C
    import os

    from mvlm_amd import _lib

    path = os.environ.get("MVLM_RCCL_LIB", "/opt/rocm/lib/librccl.so.1")
    if not os.path.exists(path):
        pytest.skip("no RCCL library")
    os.environ["MVLM_RCCL_LIB"] = path       # the library the communicator comes from is the one the entry point must call
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        ctx = _lib.get_context(0)
        rs = np.random.RandomState(5)
        local = torch.from_numpy(rs.standard_normal((84, 96, 3)).astype(np.float32)).cuda()
        rc, out = _gather(ctx, comm, 0, 1, local, 96, 84)
        assert rc == 0, ctx.lib.mvlm_last_error(ctx.handle)
        torch.cuda.synchronize()
        assert torch.equal(out, local)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_render_overflow_is_reported_and_the_next_render_is_clean():
    """A mesh of window-filling triangles overflows the per-view tile lists (4 T + 16 384 entries: 100 triangles x 256 tiles):
    mvlm_render_check says so - the flag now reaches the host inside the tile kernel - and the render after it is bit-equal to
    the oracle again (counters, key plane and flag all start clean)."""
    from mvlm_amd import _lib
    from mvlm_amd.utils import HipRenderer3D, Mesh
    from oracle import raster

    n = 100
    rs = np.random.RandomState(2)
    verts = np.concatenate([np.array([[-400, -400, z], [400, -400, z], [0, 600, z]], np.float32) for z in rs.uniform(-50, 50, n)])
    huge = Mesh(verts, np.arange(3 * n, dtype=np.int32).reshape(n, 3))
    r = HipRenderer3D(n_views=8, verbose=False)
    poses = r.generate_3d_transformations()
    r.render_device(huge, poses)
    with pytest.raises(_lib.MvlmHipError, match="overflowed"):
        r.check()
    face = _face(40)
    got = r.render_device(face, poses).cpu().numpy()
    r.check()
    np.testing.assert_array_equal(got, raster.multiview_render(face.verts, face.tris, face.uvs, face.texture, poses))
