#!/usr/bin/env python3
"""Writes tests/golden/jpeg/: small JPEG files of the kinds the texture decoder takes (and one it does not) and
expected.npz = the pixels libjpeg-turbo (Pillow, this image: the decoder family VTK's vtkJPEGReader bundles) decodes them
to with its defaults - JDCT_ISLOW, fancy upsampling, RGB out.  The files are DATA (encoded here from a seeded synthetic
picture); oracle/jpeg.py and the device decoder are both checked against expected.npz.

usage: python tests/golden/make_jpeg_golden.py"""
import io
from pathlib import Path

import numpy as np
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 24
OUT = Path(__file__).resolve().parent / "jpeg"


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 9.0 + yy / 17.0), 128 + 90 * np.cos(xx / 5.0 - yy / 11.0),
                    128 + 80 * np.sin(xx * yy / 900.0)], -1) + rng.normal(0, 12, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, height, width, save options
    ("c420_37x53_q75", 37, 53, dict(quality=75, subsampling=2)),
    ("c422_100x130_q95", 100, 130, dict(quality=95, subsampling=1)),
    ("c444_16x16_q30", 16, 16, dict(quality=30, subsampling=0)),
    ("c420_80x96_restart3", 80, 96, dict(quality=90, subsampling=2, restart_marker_blocks=3)),
    ("c422_80x96_restart_rows", 80, 96, dict(quality=90, subsampling=1, restart_marker_rows=1)),
    ("c420_80x96_optimised", 80, 96, dict(quality=90, subsampling=2, optimize=True)),
    ("c420_1x1", 1, 1, dict(quality=90, subsampling=2)),
    ("c420_5x3_narrow", 5, 3, dict(quality=90, subsampling=2)),      # downsampled width <= 2: libjpeg replicates, no filter
    ("c422_17x4_narrow", 17, 4, dict(quality=90, subsampling=1)),
    ("c420_64x64_q100", 64, 64, dict(quality=100, subsampling=2)),
    ("grey_40x44_q90", 40, 44, dict(quality=90)),
    ("progressive_64x64", 64, 64, dict(quality=90, progressive=True)),  # not taken by the device decoder
]


def main():
    OUT.mkdir(exist_ok=True)
    expected = {}
    for i, (name, h, w, opts) in enumerate(CASES):
        img = picture(h, w, 100 + i)
        if name.startswith("grey"):
            img = img[..., 0]
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **opts)
        (OUT / f"{name}.jpg").write_bytes(buf.getvalue())
        with Image.open(io.BytesIO(buf.getvalue())) as im:
            expected[name] = np.asarray(im.convert("RGB"), dtype=np.uint8)
    np.savez_compressed(OUT / "expected.npz", **expected)
    print("wrote", len(CASES), "files,", sum(p.stat().st_size for p in OUT.iterdir()), "bytes")


if __name__ == "__main__":
    main()
