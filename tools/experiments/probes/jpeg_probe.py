"""GPU probe: the device JPEG decoder against Pillow on a matrix of files + timing on a 2048 x 2048 texture."""
import ctypes as C
import io
import sys
import time
from pathlib import Path

import numpy as np
import torch
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 25  # (optimize=True needs the whole file in one encoder buffer)

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from mvlm_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
lib = ctx.lib
rng = np.random.default_rng(0)


def photo(h, w, noise=12):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 9.0 + yy / 17.0), 128 + 90 * np.cos(xx / 5.0 - yy / 11.0), 128 + 80 * np.sin(xx * yy / 900.0)], -1)
    img += rng.normal(0, noise, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def decode(data):
    raw = np.frombuffer(data, np.uint8)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    why = C.create_string_buffer(200)
    rc = lib.mvlm_jpeg_info(_lib.as_ptr(raw, C.c_uint8), raw.size, C.byref(w), C.byref(h), C.byref(c), why, 200)
    if rc != 0:
        return None, why.value.decode(), -1
    out = torch.empty((h.value, w.value, 3), dtype=torch.uint8, device="cuda")
    rounds = C.c_int(-1)
    rc = lib.mvlm_jpeg_decode(ctx.handle, _lib.as_ptr(raw, C.c_uint8), raw.size, C.c_void_p(out.data_ptr()), C.byref(rounds))
    if rc != 0:
        return None, lib.mvlm_last_error(ctx.handle).decode(), rounds.value
    return out.cpu().numpy(), "", rounds.value


bad = 0
n = 0
for (h, w) in [(64, 64), (37, 53), (16, 16), (8, 8), (1, 1), (3, 5), (100, 130), (17, 4), (5, 3), (300, 260), (513, 1025)]:
    for sub in (0, 1, 2):
        for q in (30, 75, 95, 100):
            for kw in ({}, dict(optimize=True), dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):
                buf = io.BytesIO()
                Image.fromarray(photo(h, w)).save(buf, "JPEG", quality=q, subsampling=sub, **kw)
                data = buf.getvalue()
                want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
                got, why, rounds = decode(data)
                n += 1
                if got is None or not np.array_equal(got, want):
                    bad += 1
                    if bad < 12:
                        d = None if got is None else np.abs(got.astype(int) - want.astype(int))
                        print("MISMATCH", h, w, sub, q, kw, why, rounds, None if d is None else (d.max(), int((d > 0).sum()), np.argwhere(d.max(-1) > 0)[:3].tolist()))
buf = io.BytesIO()
Image.fromarray(photo(40, 44)[..., 0]).save(buf, "JPEG", quality=90)
got, why, rounds = decode(buf.getvalue())
want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
print("grey", got is not None and np.array_equal(got, want), why)
buf = io.BytesIO()
Image.fromarray(photo(64, 64)).save(buf, "JPEG", quality=90, progressive=True)
print("progressive ->", decode(buf.getvalue())[1])
print("cases", n, "bad", bad)

for noise, q in ((12, 95), (4, 95), (12, 75), (30, 98)):
    big = photo(2048, 2048, noise)
    buf = io.BytesIO()
    Image.fromarray(big).save(buf, "JPEG", quality=q)
    data = buf.getvalue()
    t0 = time.perf_counter()
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    t_pil = time.perf_counter() - t0
    got, why, rounds = decode(data)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        got, why, rounds = decode(data)
        ts.append(time.perf_counter() - t0)
    print(f"2048^2 noise {noise} q {q}: {len(data) / 1e6:.2f} MB, equal {got is not None and np.array_equal(got, want)}, rounds {rounds}, "
          f"device decode (incl. D2H of 12 MB) {min(ts) * 1e3:.2f} ms, libjpeg {t_pil * 1e3:.2f} ms {why}")
