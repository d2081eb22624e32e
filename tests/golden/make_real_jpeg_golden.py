#!/usr/bin/env python3
"""Writes tests/golden/jpeg/real/: three of the reference's own scanner textures (assets/Normal-Emotion/*.jpg - the only real
inputs the reference holds for obj_to_actor's vtkJPEGReader path, src/mvlm/utils/utils3d.py:26-36) as DATA fixtures, plus
expected.json = what libjpeg-turbo (Pillow, this image) decodes each of them to (size + SHA-256 of the RGB bytes; 24 MB of pixels
per file are not committed).  Unlike everything under tests/golden/jpeg/ these files were NOT written by Pillow's encoder:
3546 x 2282 ... 3562 x 2359 (no multiple of 16), 4:2:0, 2.2 MB of entropy-coded data without restart markers, JFIF with two
quantisation and four Huffman tables of another encoder.

BUILD-CONTAINER TOOL (needs /root/reference):  python tests/golden/make_real_jpeg_golden.py"""
import hashlib
import json
import shutil
from pathlib import Path

import numpy as np
from PIL import Image

SRC = Path("/root/reference/assets/Normal-Emotion")
OUT = Path(__file__).resolve().parent / "jpeg" / "real"
PICK = ("angry_01", "happy_01", "sad_03")   # the file the round-5 review decoded; the largest; width a multiple of 8 only, odd height


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    expected = {}
    for name in PICK:
        shutil.copyfile(SRC / f"{name}.jpg", OUT / f"{name}.jpg")
        (OUT / f"{name}.jpg").chmod(0o644)
        rgb = np.asarray(Image.open(OUT / f"{name}.jpg").convert("RGB"))
        expected[name] = {"height": int(rgb.shape[0]), "width": int(rgb.shape[1]), "file_bytes": (OUT / f"{name}.jpg").stat().st_size,
                          "sha256_rgb": hashlib.sha256(rgb.tobytes()).hexdigest()}
        print(name, expected[name])
    (OUT / "expected.json").write_text(json.dumps(expected, indent=1) + "\n")


if __name__ == "__main__":
    main()
