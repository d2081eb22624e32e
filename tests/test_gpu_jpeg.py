"""Texture JPEGs decoded on the device (mvlm_amd/csrc/jpeg.hip) against libjpeg: the committed fixtures, a seeded matrix of
fresh files decoded by Pillow's libjpeg-turbo, the bench's 2048 x 2048 texture, damaged streams, and the mesh path end to
end (the decoder replaces vtkJPEGReader in obj_to_actor, src/mvlm/utils/utils3d.py:28-34; bar: every byte equal)."""
import ctypes as C
import io

import numpy as np
import pytest
import torch
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
JPEG = GOLDEN / "jpeg"
NAMES = sorted(p.stem for p in JPEG.glob("*.jpg"))


@pytest.fixture(scope="module")
def ctx():
    from mvlm_amd import _lib

    return _lib.Context(0)


def _decode(ctx, data):
    """-> (pixels or None, return code, message, synchronisation rounds)"""
    from mvlm_amd import _lib

    raw = np.frombuffer(data, np.uint8)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    why = C.create_string_buffer(256)
    rc = ctx.lib.mvlm_jpeg_info(_lib.as_ptr(raw, C.c_uint8), raw.size, C.byref(w), C.byref(h), C.byref(c), why, 256)
    if rc != 0:
        return None, rc, why.value.decode(), -1
    out = torch.full((h.value, w.value, 3), 77, dtype=torch.uint8, device="cuda")
    rounds = C.c_int(-1)
    rc = ctx.lib.mvlm_jpeg_decode(ctx.handle, _lib.as_ptr(raw, C.c_uint8), raw.size, C.c_void_p(out.data_ptr()), C.byref(rounds))
    if rc != 0:
        msg = ctx.lib.mvlm_last_error(ctx.handle)
        return None, rc, msg.decode() if msg else "", rounds.value
    return out.cpu().numpy(), 0, "", rounds.value


def _encode(img, **opts):
    from PIL import Image, ImageFile

    ImageFile.MAXBLOCK = 1 << 25
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **opts)
    return buf.getvalue()


def _pillow(data):
    from PIL import Image

    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def _picture(h, w, seed, noise=12):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 9.0 + yy / 17.0), 128 + 90 * np.cos(xx / 5.0 - yy / 11.0),
                    128 + 80 * np.sin(xx * yy / 900.0)], -1) + rng.normal(0, noise, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("name", NAMES)
def test_fixtures_byte_for_byte(ctx, name):
    data = (JPEG / f"{name}.jpg").read_bytes()
    got, rc, why, rounds = _decode(ctx, data)
    if name.startswith("progressive"):
        assert got is None and rc == 2 and "progressive" in why
        return
    want = np.load(JPEG / "expected.npz")[name]
    assert rc == 0 and rounds >= 0, why
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("sub", [0, 1, 2])
def test_matrix_of_fresh_files_against_pillow(ctx, sub):
    """Sizes around the MCU and subsequence boundaries, all qualities, optimised tables, restart intervals of 1 MCU / 3 MCUs /
    one and two MCU rows - every byte as Pillow's libjpeg-turbo decodes it."""
    n = 0
    for (h, w) in [(8, 8), (1, 1), (3, 5), (17, 4), (5, 3), (16, 16), (37, 53), (64, 64), (100, 130), (255, 257), (300, 260), (513, 1025)]:
        for q in (20, 60, 90, 100):
            for extra in ({}, dict(optimize=True), dict(restart_marker_blocks=1), dict(restart_marker_blocks=3), dict(restart_marker_rows=2)):
                if h * w > 100000 and extra and q != 90:
                    continue
                data = _encode(_picture(h, w, seed=h * w + q + sub), quality=q, subsampling=sub, **extra)
                got, rc, why, _ = _decode(ctx, data)
                assert rc == 0, (h, w, q, extra, why)
                assert np.array_equal(got, _pillow(data)), (h, w, q, extra)
                n += 1
    assert n > 150
    grey = _encode(_picture(75, 91, seed=5)[..., 1], quality=88)
    got, rc, why, _ = _decode(ctx, grey)
    assert rc == 0 and np.array_equal(got, _pillow(grey))


@pytest.mark.parametrize("noise,q", [(12, 95), (30, 98), (2, 70)])
def test_full_size_texture(ctx, noise, q):
    """2048 x 2048 (the bench's texture size): 15-40 thousand subsequences, tens of synchronisation rounds."""
    data = _encode(_picture(2048, 2048, seed=q, noise=noise), quality=q)
    got, rc, why, rounds = _decode(ctx, data)
    assert rc == 0 and rounds >= 1, why
    assert np.array_equal(got, _pillow(data))
    again, rc, _, rounds2 = _decode(ctx, data)  # the scratch is reused: same bytes, same number of rounds
    assert rc == 0 and rounds2 == rounds and np.array_equal(again, got)


def test_damaged_streams_never_take_the_gpu_down(ctx):
    """Flipped and missing bytes in the entropy-coded segment: the call answers 0 (a picture libjpeg would also produce
    something for) or 2 (not taken), every access stays inside its buffers, and a good file decodes afterwards."""
    rng = np.random.default_rng(7)
    good = (JPEG / "c420_80x96_restart3.jpg").read_bytes()
    plain = (JPEG / "c422_100x130_q95.jpg").read_bytes()
    sos = {id(d): d.index(b"\xff\xda") + 14 for d in (good, plain)}
    seen = set()
    for trial in range(400):
        src = good if trial % 2 else plain
        d = bytearray(src)
        kind = trial % 4
        if kind == 0:  # a few flipped bytes
            for _ in range(1 + trial % 5):
                d[rng.integers(sos[id(src)], len(d) - 2)] ^= int(rng.integers(1, 256))
            if trial % 16 == 0:  # ... and in the tables too (the header still parses or the file is refused up front)
                d[rng.integers(20, sos[id(src)] - 14)] ^= int(rng.integers(1, 256))
        elif kind == 1:  # truncated
            d = d[:rng.integers(sos[id(src)], len(d) - 2)]
        elif kind == 2:  # a run of zeros
            a = int(rng.integers(sos[id(src)], len(d) - 40))
            d[a:a + 32] = bytes(32)
        else:  # a run of ones (looks like markers / fill bytes)
            a = int(rng.integers(sos[id(src)], len(d) - 40))
            d[a:a + 8] = b"\xff" * 8
        got, rc, why, _ = _decode(ctx, bytes(d))
        assert rc in (0, 2), (trial, rc, why)
        seen.add(rc)
        if rc == 0:
            assert got.shape == _pillow(src).shape
    assert seen == {0, 2}
    got, rc, _, _ = _decode(ctx, plain)
    assert rc == 0 and np.array_equal(got, np.load(JPEG / "expected.npz")["c422_100x130_q95"])


def test_mesh_path_device_decode_equals_host_decode(tmp_path):
    """load_obj keeps the JPEG bytes, the upload decodes them on the device: the rendered views (RGB and depth planes) and
    the landmarks of a whole predict_one_file are what the host-decoded texture gives, bit for bit."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.mesh_io import load_obj
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=61, tex_size=512, seed=2)
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="RGB", verbose=False)
    poses = pipe.renderer_3d.generate_3d_transformations()
    host = load_obj(obj, decode="host")
    dev = load_obj(obj)
    assert dev.texture_jpeg is not None and dev._texture is None
    img_host = pipe.renderer_3d.render_device(host, poses)
    img_dev = pipe.renderer_3d.render_device(dev, poses)
    assert dev._texture is None, "the device path must not have decoded on the host"
    assert torch.equal(img_dev, img_host)
    assert float(img_dev[..., :3].std()) > 0.05  # (a textured render, not a white one)
    np.random.seed(3)
    a = pipe.predict_one_file(obj)
    np.random.seed(3)
    lm_host, _ = pipe.predict_mesh_device(host, pipe.renderer_3d.generate_3d_transformations())
    assert a is not None and np.array_equal(np.asarray(a), np.asarray(lm_host))


def test_a_jpeg_the_device_does_not_take_is_decoded_by_libjpeg(tmp_path):
    from mvlm_amd import pipeline
    from mvlm_amd.utils.mesh_io import load_obj
    from mvlm_amd.utils.synthetic import write_face_like_obj
    from PIL import Image

    obj = write_face_like_obj(tmp_path / "face.obj", grid=41, tex_size=256, seed=5)
    with Image.open(obj.with_suffix(".jpg")) as im:
        im.save(obj.with_suffix(".jpg"), "JPEG", quality=90, progressive=True)
    pipe = pipeline.create_pipeline("dtu3d", n_views=4, weights="synthetic:3", image_mode="RGB", verbose=False)
    poses = pipe.renderer_3d.generate_3d_transformations()
    dev = load_obj(obj)
    img_dev = pipe.renderer_3d.render_device(dev, poses)
    assert dev._texture is not None  # fell back to the host decoder
    assert torch.equal(img_dev, pipe.renderer_3d.render_device(load_obj(obj, decode="host"), poses))
    # a texture nobody can decode is ignored: the white mesh of utils3d.py:58-64
    obj.with_suffix(".jpg").write_bytes(obj.with_suffix(".jpg").read_bytes()[:300])
    broken = load_obj(obj)
    img = pipe.renderer_3d.render_device(broken, poses)
    assert broken.texture is None
    fg = img[..., 3] < 1.0
    assert bool(fg.any()) and bool((img[..., :3][fg] == 1.0).all())


def test_folder_with_reader_threads_decodes_on_the_upload_stream(tmp_path):
    """predict_files: reader threads load and upload (and so decode) the next scans while the launch thread works - the
    results are those of the files one by one."""
    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    files = []
    for i in range(5):
        f = write_face_like_obj(tmp_path / f"s{i}.obj", grid=41 + 4 * i, tex_size=256 + 64 * i, seed=10 + i)
        files.append(f)
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="RGB", verbose=False)
    np.random.seed(11)
    single = [np.asarray(pipe.predict_one_file(f)) for f in files]
    for readers in (1, 3):
        np.random.seed(11)
        out = dict((str(p), lm) for p, lm in pipe.predict_files(files, readers=readers))
        for f, want in zip(files, single):
            assert np.array_equal(np.asarray(out[str(f)]), want), (readers, f.name)


def test_texture_decoded_ahead_of_the_mesh(tmp_path):
    """HipRenderer3D.load_mesh decodes the texture on the device on a second thread while the geometry is parsed
    (mvlm_texture_from_jpeg); the mesh upload takes the buffer over (mvlm_mesh_upload_texture).  Same views as the host
    decode; a handle nobody uploads is given back; a mesh without texture coordinates does not use (or consume) it."""
    import gc

    from mvlm_amd import pipeline
    from mvlm_amd.utils.mesh_io import load_obj
    from mvlm_amd.utils.render3d import decode_texture_ahead, upload_mesh
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=61, tex_size=640, seed=8)
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="RGB", verbose=False)
    r3 = pipe.renderer_3d
    poses = r3.generate_3d_transformations()
    want = r3.render_device(load_obj(obj, decode="host"), poses).clone()
    for _ in range(3):  # (buffers go through the mesh pool and come back)
        mesh = r3.load_mesh(obj)
        ahead = mesh._texture_ahead
        assert ahead is not None and ahead.handle and mesh._texture is None and mesh.texture_jpeg is not None
        got = r3.render_device(mesh, poses)
        assert ahead.handle is None and mesh._texture_ahead is None and mesh._texture is None  # consumed, never decoded on the host
        assert torch.equal(got, want)
        del mesh
    # never uploaded: the owner frees the device buffer
    orphan = r3.load_mesh(obj)
    assert orphan._texture_ahead.handle
    del orphan
    gc.collect()
    # host decode asked for: no handle
    r3.texture_decode = "host"
    m = r3.load_mesh(obj)
    assert getattr(m, "_texture_ahead", None) is None and m._texture is not None
    assert torch.equal(r3.render_device(m, poses), want)
    r3.texture_decode = "device"
    # a texture for a mesh without texture coordinates: decoded, not used, still the caller's
    plain = tmp_path / "plain.obj"
    plain.write_text("v -50 -50 0\nv 50 -50 0\nv 0 50 10\nf 1 2 3\n")
    loose = load_obj(plain)
    loose._texture_ahead = decode_texture_ahead(r3.ctx, obj.with_suffix(".jpg").read_bytes())
    keep = loose._texture_ahead
    upload_mesh(r3.ctx, loose)
    assert keep.handle, "a mesh without texture coordinates must not consume the texture"
    img = r3.render_device(loose, poses)
    fg = img[..., 3] < 1.0
    assert bool(fg.any()) and bool((img[..., :3][fg] == 1.0).all())  # the white mesh of utils3d.py:58-64
    del keep, loose
    gc.collect()
    # a progressive file: nothing ahead, the upload falls back to libjpeg
    from PIL import Image

    with Image.open(obj.with_suffix(".jpg")) as im:
        im.save(obj.with_suffix(".jpg"), "JPEG", quality=90, progressive=True)
    prog = r3.load_mesh(obj)
    assert prog._texture_ahead is None and prog.texture_jpeg is not None
    assert torch.equal(r3.render_device(prog, poses), r3.render_device(load_obj(obj, decode="host"), poses))


def test_server_threads_with_device_decoded_textures(tmp_path):
    """The reference's server calls predict_one_file from a thread pool without locks (3DMD_server.py:26-31).  Four threads x
    six RGB scans: every call reads, decodes its texture on the device (a helper thread per call), uploads outside the pipeline
    lock and takes its turn on the GPU - the landmarks are those of the calls one after another."""
    import threading

    from mvlm_amd import pipeline
    from mvlm_amd.utils.synthetic import write_face_like_obj

    files = [write_face_like_obj(tmp_path / f"s{i}.obj", grid=51 + 6 * i, tex_size=256 + 128 * (i % 3), seed=40 + i) for i in range(6)]
    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="RGB+depth", verbose=False)
    want = {f.name: np.asarray(pipe.predict_one_file(f)) for f in files}  # (8 views: the fixed pose table)
    # RANSAC draws come from the global numpy RNG, whose order across threads is not defined: compare the landmarks that do not
    # depend on it (fewer than three surviving views -> plain least squares) exactly and the others within the consensus spread
    got: dict = {}
    errors: list = []

    def work(k):
        try:
            for j in range(len(files)):
                f = files[(j + k) % len(files)]
                got[(k, f.name)] = np.asarray(pipe.predict_one_file(f))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(got) == 24
    for (k, name), lm in got.items():
        assert lm.shape == want[name].shape and np.isfinite(lm).all()
        # same views, same maxima, same surface: whatever the draws, a landmark lands on the same spot of the mesh or a
        # neighbouring consensus - never on another scan's geometry (a mixed-up texture or mesh would move all of them)
        d = np.linalg.norm(lm - want[name], axis=1)
        assert np.median(d) < 1e-6, (k, name, float(np.median(d)))


def test_restart_markers_that_do_not_match_the_header(ctx):
    """Restart intervals that are empty, missing or too many: not taken (2), never a picture with silently missing blocks."""
    src = (JPEG / "c420_80x96_restart3.jpg").read_bytes()
    first = src.index(b"\xff\xd0")
    second = src.index(b"\xff\xd1", first)
    empty = src[:first + 2] + src[second:]                       # RST0 RST1 back to back
    missing = src[:first] + src[first + 2:]                      # one marker gone: the two intervals run together
    extra = src[:first] + b"\xff\xd0" + src[first:]              # one marker too many
    tail = src[:src.rindex(b"\xff\xd9")]
    last_rst = max(tail.rfind(bytes([0xFF, 0xD0 + i])) for i in range(8))
    empty_last = src[:last_rst + 2] + b"\xff\xd9"                # the last interval holds nothing
    for name, data in (("empty", empty), ("missing", missing), ("extra", extra), ("empty_last", empty_last)):
        got, rc, why, _ = _decode(ctx, data)
        assert rc == 2 and got is None and why, (name, rc, why)
    got, rc, _, _ = _decode(ctx, src)
    assert rc == 0 and np.array_equal(got, np.load(JPEG / "expected.npz")["c420_80x96_restart3"])
