"""ASan + UBSan run of the native OBJ reader on mutated files (CPU build; the GPU pool has no sanitizers)."""
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REPO


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_obj_reader_under_sanitizers(tmp_path):
    exe = tmp_path / "harness"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                        "-x", "c++", str(REPO / "mvlm_amd/csrc/obj_reader.hip"), str(REPO / "tests/native/obj_reader_harness.cpp"),
                        "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    from mvlm_amd.utils.synthetic import write_face_like_obj

    base = write_face_like_obj(tmp_path / "base.obj", grid=12, tex_size=8).read_bytes()
    rs = np.random.RandomState(3)
    alphabet = np.frombuffer(b"0123456789 .-+e/\n\r\tvtfn#x", np.uint8)
    files = [tmp_path / "base.obj"]
    for i in range(150):
        b = bytearray(base)
        kind = i % 5
        if kind == 0:      # random byte substitutions from the format's alphabet
            for p in rs.randint(0, len(b), 40):
                b[p] = alphabet[rs.randint(len(alphabet))]
        elif kind == 1:    # truncation in the middle of a token
            b = b[: rs.randint(1, len(b))]
        elif kind == 2:    # arbitrary binary noise
            for p in rs.randint(0, len(b), 40):
                b[p] = rs.randint(256)
        elif kind == 3:    # huge / negative / zero indices and very long numbers
            extra = [b"f 1 2 99999999999999999999\n", b"f -99999999 1 2\n", b"f 0 0 0\n", b"f 1/ 2/ 3/\n", b"f / / /\n",
                     b"v " + b"9" * 400 + b" 1 1\n", b"v 1e99999 -1e-99999 0." + b"0" * 300 + b"1\n", b"f 1//// 2 3\n",
                     b"f " + b" ".join(b"%d" % (k % 5 + 1) for k in range(200)) + b"\n", b"vt\n", b"v\n", b"f\n"]
            b += extra[rs.randint(len(extra))] * 3
        else:              # lines glued together / no trailing newline
            b = b.replace(b"\n", b" ", 5)[:-1]
        f = tmp_path / f"m{i}.obj"
        f.write_bytes(bytes(b))
        files.append(f)
    (tmp_path / "empty.obj").write_bytes(b"")
    files += [tmp_path / "empty.obj", tmp_path / "does_not_exist.obj"]
    r = subprocess.run([str(exe)] + [str(f) for f in files], capture_output=True, text=True,
                       env={"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    assert lines[0].endswith("bad_indices=0") and " ok " in lines[0]
    assert all(("rc=" in ln) or ln.endswith("bad_indices=0") for ln in lines)   # parsed output is always in range
