"""The reference's own scanner textures (assets/Normal-Emotion/*.jpg; three of them committed as data under
tests/golden/jpeg/real/, tests/golden/make_real_jpeg_golden.py) - the only real inputs the reference holds for obj_to_actor's
vtkJPEGReader path (src/mvlm/utils/utils3d.py:26-36).  Not written by Pillow's encoder, 3496 x 2275 ... 3562 x 2359, 4:2:0,
2.2 MB of entropy-coded data without restart markers.

CPU: oracle/jpeg.py == libjpeg-turbo (Pillow) on every one of them (all 15 where /root/reference exists, else the three).
-m gpu: the device decoder byte for byte, and the rasteriser with such a 24 MB non-power-of-two texture."""
import ctypes as C
import hashlib
import io
import json
import os
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

REAL = GOLDEN / "jpeg" / "real"
EXPECTED = json.loads((REAL / "expected.json").read_text())
REFERENCE = Path("/root/reference/assets/Normal-Emotion")


def _pillow(data):
    from PIL import Image

    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def _oracle_equals_pillow(path: str):
    from oracle import jpeg as ojpeg

    data = Path(path).read_bytes()
    want = _pillow(data)
    got = ojpeg.decode(data)
    return Path(path).stem, got.shape == want.shape and bool(np.array_equal(got, want)), hashlib.sha256(want.tobytes()).hexdigest()


def test_oracle_decodes_the_reference_textures_like_libjpeg():
    files = sorted(REFERENCE.glob("*.jpg")) if REFERENCE.is_dir() else sorted(REAL.glob("*.jpg"))
    assert len(files) in (15, 3)
    workers = max(1, min(8, len(os.sched_getaffinity(0)), len(files)))      # ~18 s of pure-Python entropy decoding per file
    with ProcessPoolExecutor(workers) as pool:
        results = list(pool.map(_oracle_equals_pillow, [str(f) for f in files]))
    for name, equal, sha in results:
        assert equal, name
        if name in EXPECTED:
            assert sha == EXPECTED[name]["sha256_rgb"], name   # (the committed checksum is this libjpeg's decode)


def test_committed_textures_are_the_reference_files():
    for name, e in EXPECTED.items():
        data = (REAL / f"{name}.jpg").read_bytes()
        assert len(data) == e["file_bytes"]
        if REFERENCE.is_dir():
            assert data == (REFERENCE / f"{name}.jpg").read_bytes()
        assert b"\xff\xc0" in data and b"\xff\xc2" not in data[:data.index(b"\xff\xda")]      # baseline, one scan
        assert not any(bytes([0xFF, 0xD0 + k]) in data for k in range(8))                       # no restart markers


# ---------------------------------------------------------------------------------------------------------------- -m gpu
@pytest.fixture(scope="module")
def ctx():
    from mvlm_amd import _lib

    return _lib.Context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_device_decoder_on_a_real_texture(ctx, name):
    import torch

    from mvlm_amd import _lib

    data = (REAL / f"{name}.jpg").read_bytes()
    e = EXPECTED[name]
    raw = np.frombuffer(data, np.uint8)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    why = C.create_string_buffer(256)
    assert ctx.lib.mvlm_jpeg_info(_lib.as_ptr(raw, C.c_uint8), raw.size, C.byref(w), C.byref(h), C.byref(c), why, 256) == 0, why.value
    assert (h.value, w.value, c.value) == (e["height"], e["width"], 3)
    out = torch.full((h.value, w.value, 3), 77, dtype=torch.uint8, device="cuda")
    rounds = C.c_int(-1)
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ctx.lib.mvlm_jpeg_decode(ctx.handle, _lib.as_ptr(raw, C.c_uint8), raw.size, C.c_void_p(out.data_ptr()), C.byref(rounds))
        times.append(time.perf_counter() - t0)       # (the call returns with the image complete)
        assert rc == 0, ctx.lib.mvlm_last_error(ctx.handle)
    got = out.cpu().numpy()
    t0 = time.perf_counter()
    want = _pillow(data)
    t_host = time.perf_counter() - t0
    print(f"{name}: {w.value} x {h.value}, {rounds.value} synchronisation rounds, device decode {min(times) * 1e3:.2f} ms "
          f"(first call {times[0] * 1e3:.2f}), libjpeg-turbo on one core {t_host * 1e3:.1f} ms")
    assert hashlib.sha256(want.tobytes()).hexdigest() == e["sha256_rgb"]
    assert np.array_equal(got, want)
    assert 1 <= rounds.value < 200


@pytest.mark.gpu
def test_render_with_a_real_texture(tmp_path):
    """mvlm_mesh_upload_jpeg with the 24 MB, 3546 x 2282 texture (nothing a power of two, rows of 10 638 bytes) under the
    224-grid face: the rendered views are bit for bit the oracle rasteriser's with libjpeg's decode of the same file."""
    import torch

    from mvlm_amd.utils import HipRenderer3D
    from mvlm_amd.utils.mesh_io import load_obj, write_obj
    from mvlm_amd.utils.synthetic import face_like_mesh
    from oracle import raster

    m = face_like_mesh(grid=224, tex_size=8, seed=5)
    obj = tmp_path / "scan.obj"
    write_obj(obj, m.verts, m.tris, m.uvs, None)
    jpg = (REAL / "angry_01.jpg").read_bytes()
    obj.with_suffix(".jpg").write_bytes(jpg)
    mesh = load_obj(obj)
    assert mesh.texture_jpeg == jpg and mesh._texture is None
    r = HipRenderer3D(n_views=8, verbose=False)
    poses = r.generate_3d_transformations()
    got = r.render_device(mesh, poses).cpu().numpy()
    r.check()
    assert mesh._texture is None, "the texture was decoded on the host"
    want = raster.multiview_render(mesh.verts, mesh.tris, mesh.uvs, _pillow(jpg), poses)
    assert np.array_equal(got, want)
    cov = want[..., 3] != np.float32(1 / 255)
    assert cov.mean() > 0.2 and len(np.unique((want[cov][:, :3] * 255).round().astype(np.uint8), axis=0)) > 5000   # a real picture
    # and the same through the whole product path: predict_one_file on that file runs and returns finite landmarks
    from mvlm_amd import pipeline

    pipe = pipeline.create_pipeline("dtu3d", n_views=8, weights="synthetic:3", image_mode="RGB", verbose=False)
    np.random.seed(3)
    lm = pipe.predict_one_file(obj)
    assert lm is not None and lm.shape == (73, 3) and np.isfinite(lm).all()
