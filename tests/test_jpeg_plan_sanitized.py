"""ASan + UBSan run of the JPEG host parser (mvlm_amd/csrc/jpeg_plan.cpp) - the code of the device JPEG decoder that reads
untrusted file bytes: headers, Huffman tables, the un-stuffing into the staging buffer.  CPU build (the GPU pool has no
sanitizers); same shape as tests/test_obj_reader_sanitized.py.  The reference swallows texture failures (utils3d.py:35-36);
a silent out-of-bounds read is not that."""
import shutil
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import REPO

JPEG = Path(__file__).resolve().parent / "golden" / "jpeg"


def _sof(data: bytes) -> int:
    return data.index(b"\xff\xc0")


def _inputs():
    rng = np.random.default_rng(11)
    out = []
    sources = [(JPEG / f"{n}.jpg").read_bytes() for n in ("c420_80x96_restart3", "c422_100x130_q95", "grey_40x44_q90", "c420_80x96_optimised")]
    out += sources
    for trial in range(3000):      # mutated headers (the generator of tests/test_jpeg_cpu.py's header test)
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
            d[a:a + 2] = int(rng.integers(0, 65536)).to_bytes(2, "big")
        out.append(bytes(d))
    rng = np.random.default_rng(7)
    good, plain = sources[0], sources[1]
    for trial in range(400):       # damaged entropy-coded segments (the generator of tests/test_gpu_jpeg.py)
        src = good if trial % 2 else plain
        sos = src.index(b"\xff\xda") + 14
        d = bytearray(src)
        kind = trial % 4
        if kind == 0:
            for _ in range(1 + trial % 5):
                d[rng.integers(sos, len(d) - 2)] ^= int(rng.integers(1, 256))
        elif kind == 1:
            d = d[:rng.integers(sos, len(d) - 2)]
        elif kind == 2:
            a = int(rng.integers(sos, len(d) - 40))
            d[a:a + 32] = bytes(32)
        else:
            a = int(rng.integers(sos, len(d) - 40))
            d[a:a + 8] = b"\xff" * 8
        out.append(bytes(d))
    src = good                     # restart markers that do not match the header
    first = src.index(b"\xff\xd0")
    second = src.index(b"\xff\xd1", first)
    tail = src[:src.rindex(b"\xff\xd9")]
    last_rst = max(tail.rfind(bytes([0xFF, 0xD0 + i])) for i in range(8))
    out += [src[:first + 2] + src[second:], src[:first] + src[first + 2:], src[:first] + b"\xff\xd0" + src[first:],
            src[:last_rst + 2] + b"\xff\xd9", src[:first] + b"\xff\xd0" * 300 + src[first:]]
    # header bombs: a few hundred bytes that declare a huge frame (nothing may be sized from them)
    for h, w in ((65535, 65535), (20000, 20000), (16385, 8), (8, 16385), (16384, 16384), (0, 8), (8, 0)):
        d = bytearray(plain)
        s = _sof(plain)
        d[s + 5:s + 9] = struct.pack(">HH", h, w)
        out.append(bytes(d))
    # a stream that is nothing but markers / fill bytes / stuffing behind a valid header
    sos = plain.index(b"\xff\xda") + 14
    out += [plain[:sos] + b"\xff" * 5000, plain[:sos] + b"\xff\x00" * 5000, plain[:sos] + b"\xff\xd0" * 5000, plain[:sos],
            plain[:sos] + b"\xff", plain[:sos - 1], b"\xff\xd8", b"", b"\xff\xd8\xff"]
    for _ in range(64):
        out.append(b"\xff\xd8" + rng.integers(0, 256, int(rng.integers(0, 600)), dtype=np.uint8).tobytes())
    real = sorted((JPEG / "real").glob("*.jpg")) if (JPEG / "real").is_dir() else []
    out += [p.read_bytes() for p in real]          # the reference's own scanner textures (3546 x 2282, no restart markers)
    return out, len(real)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_jpeg_host_parser_under_sanitizers(tmp_path):
    exe = tmp_path / "harness"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        str(REPO / "mvlm_amd/csrc/jpeg_plan.cpp"), str(REPO / "tests/native/jpeg_plan_harness.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    inputs, n_real = _inputs()
    pack = tmp_path / "inputs.pack"
    with open(pack, "wb") as f:
        for d in inputs:
            f.write(struct.pack("<I", len(d)))
            f.write(d)
    r = subprocess.run([str(exe), str(pack)], capture_output=True, text=True,
                       env={"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(inputs)
    assert not any("INVARIANT" in ln for ln in lines)
    for k in range(4):                                   # the four sources themselves parse and stage
        assert " rc=0 fill=0 " in lines[k], lines[k]
    taken = sum(" rc=0 fill=0 " in ln for ln in lines)
    refused = sum(" rc=2 " in ln for ln in lines) + sum(" fill=2 " in ln for ln in lines)
    assert taken + refused == len(lines) and taken > 300 and refused > 300, (taken, refused)
    bombs = [ln for ln in lines if "16384 pixels a side" in ln or "too short for the frame" in ln]
    assert len(bombs) >= 5, bombs
    for ln in lines[len(lines) - n_real:] if n_real else []:
        assert " rc=0 fill=0 " in ln and " 3 " in ln, ln       # real textures: taken, three components
