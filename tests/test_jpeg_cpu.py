"""The texture decoder's CPU side: oracle/jpeg.py (numpy restatement of libjpeg's baseline decode) against the committed
fixtures and against Pillow's libjpeg-turbo, the host-only header parser of the C-ABI, and the loaders' lazy texture.
Reference boundary: vtkJPEGReader in obj_to_actor / multi_read_surface (src/mvlm/utils/utils3d.py:28-34, :42-48, :457-462)."""
import ctypes as C
import io

import numpy as np
import pytest
from conftest import GOLDEN

JPEG = GOLDEN / "jpeg"
NAMES = sorted(p.stem for p in JPEG.glob("*.jpg"))
TAKEN = [n for n in NAMES if not n.startswith("progressive")]


def _expected():
    return np.load(JPEG / "expected.npz")


def test_fixture_set_is_complete():
    exp = _expected()
    assert set(exp.files) == set(NAMES) and len(TAKEN) >= 11


@pytest.mark.parametrize("name", TAKEN)
def test_oracle_decodes_the_fixtures_byte_for_byte(name):
    from oracle import jpeg

    got = jpeg.decode((JPEG / f"{name}.jpg").read_bytes())
    want = _expected()[name]
    assert got.dtype == np.uint8 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_oracle_refuses_what_is_out_of_scope():
    from oracle import jpeg

    with pytest.raises(jpeg.Unsupported, match="progressive"):
        jpeg.decode((JPEG / "progressive_64x64.jpg").read_bytes())
    with pytest.raises(jpeg.Unsupported):
        jpeg.decode(b"not a jpeg at all")


def _encode(img, **opts):
    from PIL import Image, ImageFile

    ImageFile.MAXBLOCK = 1 << 24
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **opts)
    return buf.getvalue()


def _pillow(data):
    from PIL import Image

    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


@pytest.mark.parametrize("h,w,sub,q,extra", [
    (33, 47, 2, 85, {}), (33, 47, 1, 85, {}), (33, 47, 0, 85, {}), (24, 24, 2, 100, {}), (9, 70, 2, 50, dict(optimize=True)),
    (48, 40, 2, 92, dict(restart_marker_blocks=1)), (48, 40, 1, 92, dict(restart_marker_rows=2)), (2, 2, 2, 90, {}), (7, 6, 1, 90, {})])
def test_oracle_is_pinned_to_pillow_on_fresh_files(h, w, sub, q, extra):
    """The restatement against the decoder itself (libjpeg-turbo through Pillow), on files made now - not only the fixtures."""
    from oracle import jpeg

    rs = np.random.RandomState(h * 1000 + w + sub)
    base = rs.randint(0, 256, size=(-(-h // 4), -(-w // 4), 3)).astype(np.float32)
    img = np.repeat(np.repeat(base, 4, axis=0), 4, axis=1)[:h, :w] + rs.randint(-15, 16, size=(h, w, 3))
    data = _encode(np.clip(img, 0, 255).astype(np.uint8), quality=q, subsampling=sub, **extra)
    assert np.array_equal(jpeg.decode(data), _pillow(data))


def _info(data):
    from mvlm_amd import _lib

    lib = _lib.load()
    raw = np.frombuffer(data, np.uint8)
    w, h, c = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    why = C.create_string_buffer(256)
    rc = lib.mvlm_jpeg_info(_lib.as_ptr(raw, C.c_uint8), raw.size, C.byref(w), C.byref(h), C.byref(c), why, 256)
    return rc, (h.value, w.value, c.value), why.value.decode()


def test_header_parser_of_the_library():
    """mvlm_jpeg_info needs no GPU: sizes of what the device decoder takes, the reason for what it does not."""
    exp = _expected()
    for name in TAKEN:
        rc, (h, w, c), why = _info((JPEG / f"{name}.jpg").read_bytes())
        assert rc == 0 and why == "", (name, why)
        assert (h, w) == exp[name].shape[:2] and c == (1 if name.startswith("grey") else 3)
    rc, _, why = _info((JPEG / "progressive_64x64.jpg").read_bytes())
    assert rc == 2 and "progressive" in why
    rc, _, why = _info(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    assert rc == 2 and "not a JPEG" in why
    good = (JPEG / "c420_37x53_q75.jpg").read_bytes()
    for cut in (2, 3, 10, 30, 100, 200, 400):  # truncated inside the headers: refused, never read past the end
        rc, _, why = _info(good[:cut])
        assert rc == 2 and why, cut
    from PIL import Image

    cmyk = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 4), np.uint8), "CMYK").save(cmyk, "JPEG")
    rc, _, why = _info(cmyk.getvalue())
    assert rc == 2 and "component" in why


def test_loaders_keep_the_jpeg_bytes_and_decode_lazily(tmp_path):
    from mvlm_amd.utils.mesh_io import load_mesh, load_obj
    from mvlm_amd.utils.prealign import apply_prealign
    from mvlm_amd.utils.synthetic import write_face_like_obj

    obj = write_face_like_obj(tmp_path / "face.obj", grid=15, tex_size=48, seed=4)
    host = load_obj(obj, decode="host")
    assert host.texture_jpeg is None and host.texture.shape == (48, 48, 3)
    for loader in (load_obj, load_mesh):
        lazy = loader(obj)
        assert lazy.texture_jpeg == obj.with_suffix(".jpg").read_bytes() and lazy._texture is None
        moved, _ = apply_prealign(lazy, dict(rot_x=10.0))
        assert moved.texture_jpeg is lazy.texture_jpeg and moved._texture is None  # still bytes: the upload decodes them
        assert np.array_equal(lazy.texture, host.texture) and lazy._texture is not None
    with pytest.raises(ValueError, match="decode mode"):
        load_obj(obj, decode="gpu")
    # a texture that does not decode is ignored (utils3d.py:35-36), lazily too
    obj.with_suffix(".jpg").write_bytes(b"\xff\xd8 broken")
    broken = load_obj(obj)
    assert broken.texture_jpeg is not None and broken.texture is None and broken.texture_jpeg is None
    # no texture coordinates -> no texture (utils3d.py:26)
    plain = tmp_path / "plain.obj"
    plain.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "plain.jpg").write_bytes((JPEG / "c444_16x16_q30.jpg").read_bytes())
    assert load_obj(plain).texture_jpeg is None and load_obj(plain).texture is None


def test_header_parser_survives_mutated_headers():
    """The header parser reads untrusted bytes on the host: 3 000 files with random bytes, deletions and duplications in the
    marker segments (and 64 random byte strings behind an SOI) - every call answers 0 or 2 and, when 0, a plausible size."""
    rng = np.random.default_rng(11)
    sources = [(JPEG / f"{n}.jpg").read_bytes() for n in ("c420_80x96_restart3", "c422_100x130_q95", "grey_40x44_q90", "c420_80x96_optimised")]
    seen = {0: 0, 2: 0}
    for trial in range(3000):
        src = sources[trial % len(sources)]
        head = src.index(b"\xff\xda") + 14
        d = bytearray(src)
        kind = trial % 5
        if kind == 0:
            for _ in range(1 + trial % 4):
                d[int(rng.integers(2, head))] = int(rng.integers(0, 256))
        elif kind == 1:
            a = int(rng.integers(2, head))
            del d[a:a + int(rng.integers(1, 40))]
        elif kind == 2:
            a = int(rng.integers(2, head))
            d[a:a] = d[a:a + int(rng.integers(1, 40))]
        elif kind == 3:
            d = d[:int(rng.integers(2, head + 8))]
        else:
            a = int(rng.integers(2, head - 2))
            d[a:a + 2] = int(rng.integers(0, 65536)).to_bytes(2, "big")  # a segment length, a size, a table id ...
        rc, (h, w, c), why = _info(bytes(d))
        assert rc in (0, 2), (trial, rc)
        seen[rc] += 1
        if rc == 0:
            assert 0 < h <= 65535 and 0 < w <= 65535 and c in (1, 3) and why == ""
        else:
            assert why
    assert seen[0] > 100 and seen[2] > 100, seen
    for _ in range(64):
        junk = b"\xff\xd8" + rng.integers(0, 256, int(rng.integers(0, 600)), dtype=np.uint8).tobytes()
        assert _info(junk)[0] in (0, 2)
