"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host-side
logic (OBJ ingest, configs, weight packing, view sharding) behaves like the reference."""
import json
import os
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import REPO


def test_library_exports_every_declared_symbol():
    from mvlm_amd import _lib

    assert _lib.LIB_PATH.exists(), "run __graft_entry__.build() first"
    lib = _lib.load()
    header = (REPO / "include" / "mvlm_hip.h").read_text()
    declared = set(re.findall(r"\b(mvlm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mvlm_build_arch() == b"gfx950"


def test_kernel_occupancy_table():
    """Every kernel of the built objects holds at least the waves per SIMD and spills at most the registers that
    tests/golden/kernel_occupancy.json records (tools/kernel_occupancy.py --write after a deliberate change): a tile that
    silently crosses a register boundary loses a resident workgroup per CU - a third of its rate at small batches."""
    import importlib.util

    build = REPO / "mvlm_amd" / "csrc" / "build"
    if not any(build.glob("*.o")):
        pytest.skip("no object files (the library was not built from source here)")
    spec = importlib.util.spec_from_file_location("kernel_occupancy", REPO / "tools" / "kernel_occupancy.py")
    ko = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ko)
    assert ko.waves_per_simd(166) == 3 and ko.waves_per_simd(191) == 2 and ko.waves_per_simd(64) == 8 and ko.waves_per_simd(256) == 2
    want = json.loads(ko.TABLE.read_text())
    got = ko.build_table()
    assert len(got) >= 80  # conv variants (single + pair), split kernels, rasteriser, fusion, surface, misc
    worse = {k: (want[k], {f: v[f] for f in ("waves_per_simd", "spilled", "vgprs")}) for k, v in got.items()
             if k in want and (v["waves_per_simd"] < want[k]["waves_per_simd"] or v["spilled"] > want[k]["spilled"])}
    assert not worse, worse
    unknown = sorted(set(got) - set(want))
    assert not unknown, f"kernels missing from the table (tools/kernel_occupancy.py --write): {unknown[:5]}"


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mvlm_amd import _lib

    with pytest.raises(_lib.MvlmHipError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline() may touch it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for p in list((REPO / "mvlm_amd").rglob("*.py")) + list((REPO / "tools").rglob("*.py")):
        assert not pat.search(p.read_text()), p
    for name, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        src = (REPO / name).read_text()
        for m in pat.finditer(src):
            # the enclosing top-level function of every oracle import
            head = src[:m.start()]
            fn = re.findall(r"^def\s+(\w+)", head, re.M)
            assert fn and fn[-1] == allowed and m.group(0).strip("\n").startswith((" ", "\t")), (name, m.group(0).strip())


def test_obj_ingest(tmp_path):
    from mvlm_amd.utils.mesh_io import load_obj

    (tmp_path / "a.obj").write_text(
        "# comment\nmtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
        "vn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4/5 -3/2 -2/3\n")
    m = load_obj(tmp_path / "a.obj")
    # quad -> 2 triangles (fan) + 1 triangle; vertex 1 appears with vt 1 and vt 5 -> duplicated point
    assert m.n_tris == 3 and m.n_verts == 5
    np.testing.assert_array_equal(m.tris[0], [0, 1, 2])
    np.testing.assert_array_equal(m.tris[1], [0, 2, 3])
    np.testing.assert_array_equal(m.verts[m.tris[2]], [[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    np.testing.assert_array_equal(m.uvs[m.tris[2][0]], [0.5, 0.5])
    assert m.texture is None
    (tmp_path / "empty.obj").write_text("# nothing\n")
    with pytest.raises(ValueError, match="does not contain any points"):
        load_obj(tmp_path / "empty.obj")
    with pytest.raises(ValueError, match="does not exist"):
        load_obj(tmp_path / "nope.obj")
    (tmp_path / "a.jpg").write_bytes(b"not a jpeg")  # unreadable texture is ignored (utils3d.py:35-36)
    assert load_obj(tmp_path / "a.obj").texture is None


def _same_mesh(a, b):
    np.testing.assert_array_equal(a.verts.view(np.uint32), b.verts.view(np.uint32))  # bit patterns: -0.0, nan
    np.testing.assert_array_equal(a.tris, b.tris)
    assert (a.uvs is None) == (b.uvs is None)
    if a.uvs is not None:
        np.testing.assert_array_equal(a.uvs.view(np.uint32), b.uvs.view(np.uint32))


def test_native_obj_reader_equals_python_statement(tmp_path):
    """mvlm_obj_read (C++, in the library) against mesh_io._parse_obj on the corner cases of the format."""
    from mvlm_amd.utils.mesh_io import load_obj

    tricky = tmp_path / "t.obj"
    tricky.write_text(
        "# comment\nmtllib x.mtl\nv 0 0 0\nv 1.5e0 0 0 1.0\nv  1 1 0.1234567890123456789\n\tv -1.25E-3 +2 .5\n"
        "vt 0 0\nvt 1 0\nvt 0.5 1 0\nvn 0 0 1\ng grp\nusemtl m\ns 1\n"
        "f 1/1 2/2 3/3 4/1\nf -1/-1 -2 -3//1\nf 1/3/1 2/2/1 3/1/1\r\nf 1 2\nf 1/9 2/2 3/3\nf 1/-9 2/2 3/3\n"
        "v 5 5 5\nf -1 1 2\nv 1e400 -1e-400 nan\nv inf -inf -0.0\nf 6 7 1 2 3\n\n   \n#f 1 2 3\n # f 1 2 3\n")
    a, b = load_obj(tricky, reader="native"), load_obj(tricky, reader="python")
    _same_mesh(a, b)
    assert a.n_tris == 10 and a.uvs is not None
    for name, text, msg in [("idx", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n", "vertex that does not exist"),
                            ("zero", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "vertex that does not exist"),
                            ("neg", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -4 1 2\n", "vertex that does not exist"),
                            ("num", "v 0 a 0\n", None), ("face", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 x\n", None),
                            ("hex", "v 0x10 0 0\n", None)]:
        p = tmp_path / f"{name}.obj"
        p.write_text(text)
        for reader in ("native", "python"):
            with pytest.raises(ValueError, match=msg):
                load_obj(p, reader=reader)
    cloud = tmp_path / "cloud.obj"
    cloud.write_text("v 0 0 0\nv 1 2 3\nvt 0 0\n")
    for reader in ("native", "python"):
        m = load_obj(cloud, reader=reader)
        assert m.n_tris == 0 and m.n_verts == 2 and m.uvs is None


def test_chunk_parallel_obj_parse_equals_one_thread(tmp_path, monkeypatch):
    """The native reader cuts the text into one chunk per thread (round 3): for 1, 2, 3, 5 and 8 chunks the mesh is the
    single-threaded one bit for bit - points, corner numbering in order of first use, negative indices that count back
    across chunk borders, a point used with several texture coordinates, CR/LF line ends - and a syntax error is
    reported for the same line."""
    from mvlm_amd.utils.mesh_io import load_obj

    rs = np.random.RandomState(9)
    lines, n_v, n_vt = [], 0, 0
    for block in range(40):                      # points, texture coordinates and faces interleaved through the file
        for _ in range(rs.randint(3, 30)):
            lines.append("v %.6f %.6f %.6f" % tuple(rs.standard_normal(3) * 50))
            n_v += 1
        for _ in range(rs.randint(0, 20)):
            lines.append("vt %.5f %.5f" % tuple(rs.rand(2)))
            n_vt += 1
        if block % 7 == 3:
            lines += ["# a comment", "", "g group%d" % block, "vn 0 0 1"]
        for _ in range(rs.randint(5, 40)):
            k = rs.randint(3, 6)
            corners = []
            for _ in range(k):
                vi = rs.randint(1, n_v + 1) if rs.rand() < 0.7 else -rs.randint(1, n_v + 1)
                if n_vt and rs.rand() < 0.8:
                    ti = rs.randint(1, n_vt + 1) if rs.rand() < 0.7 else -rs.randint(1, n_vt + 1)
                    corners.append(f"{vi}/{ti}" + ("/1" if rs.rand() < 0.2 else ""))
                else:
                    corners.append(str(vi) + ("//1" if rs.rand() < 0.2 else ""))
            lines.append("f " + " ".join(corners))
    text = "\n".join(lines[:200]) + "\r\n" + "\r\n".join(lines[200:400]) + "\n" + "\n".join(lines[400:]) + "\n"
    p = tmp_path / "mixed.obj"
    p.write_bytes(text.encode())
    monkeypatch.setenv("MVLM_OBJ_THREADS", "1")
    one = load_obj(p, reader="native")
    _same_mesh(one, load_obj(p, reader="python"))
    assert one.n_tris > 500 and one.uvs is not None
    for n in (2, 3, 5, 8):
        monkeypatch.setenv("MVLM_OBJ_THREADS", str(n))
        _same_mesh(load_obj(p, reader="native"), one)
    # the first bad line wins, whichever chunk it falls into, with its line number in the whole file
    bad = lines[:]
    bad[len(bad) * 2 // 3] = "v 1 oops 3"
    bad[len(bad) * 5 // 6] = "f 1 2 x"
    q = tmp_path / "bad.obj"
    q.write_text("\n".join(bad) + "\n")
    messages = set()
    for n in (1, 2, 4, 8):
        monkeypatch.setenv("MVLM_OBJ_THREADS", str(n))
        with pytest.raises(ValueError) as e:
            load_obj(q, reader="native")
        messages.add(str(e.value))
    assert len(messages) == 1 and f"line {len(bad) * 2 // 3 + 1}:" in messages.pop()
    # a big scan takes the threaded path by default
    monkeypatch.delenv("MVLM_OBJ_THREADS")
    from mvlm_amd.utils.synthetic import face_like_mesh, write_face_like_obj

    big = write_face_like_obj(tmp_path / "big.obj", grid=120, tex_size=8)
    assert big.stat().st_size > 1_000_000
    m, ref = load_obj(big), face_like_mesh(120, 8)
    monkeypatch.setenv("MVLM_OBJ_THREADS", "1")
    _same_mesh(m, load_obj(big))
    assert m.n_tris == ref.n_tris


def test_native_obj_reader_rounds_numbers_like_python(tmp_path):
    """Every decimal spelling must land on the same float32 as Python's float() -> np.float32."""
    from mvlm_amd.utils.mesh_io import load_obj

    rs = np.random.RandomState(5)
    vals = np.concatenate([rs.standard_normal(3000) * 10.0 ** rs.randint(-30, 30, 3000),
                           rs.standard_normal(3000) * 100, np.float64(rs.standard_normal(3000).astype(np.float32)),
                           [0.0, 1e22, 1e23, 9007199254740993.0, 8.5e-46, 1.4e-45, 3.4028235e38, 3.4028236e38, 5e-324]])
    fmts = ["%r", "%.6f", "%.17g", "%.25e", "%.3E", "%+.9f", "%.40f"]
    lines = []
    for i in range(0, len(vals) - 2, 3):
        f = fmts[(i // 3) % len(fmts)]
        tok = [repr(float(v)) if f == "%r" else f % v for v in vals[i:i + 3]]
        lines.append("v " + " ".join(tok))
    n = len(lines)
    lines += [f"f {i + 1} {i + 2} {i + 3}" for i in range(n - 2)]
    p = tmp_path / "numbers.obj"
    p.write_text("\n".join(lines) + "\n")
    with np.errstate(over="ignore"):
        _same_mesh(load_obj(p, reader="native"), load_obj(p, reader="python"))


def test_synthetic_mesh_round_trips_through_obj(tmp_path):
    from mvlm_amd.utils.mesh_io import load_obj
    from mvlm_amd.utils.synthetic import face_like_mesh, write_face_like_obj

    p = write_face_like_obj(tmp_path / "f.obj", grid=20, tex_size=32)
    m, ref = load_obj(p), face_like_mesh(20, 32)
    assert m.n_tris == ref.n_tris == 2 * 19 * 19
    np.testing.assert_allclose(m.verts[m.tris], ref.verts[ref.tris], atol=1e-5)
    assert m.texture.shape == (32, 32, 3)
    assert face_like_mesh(224, 8).n_tris == 99458  # the ~100k-triangle benchmark mesh


@pytest.mark.parametrize("dataset,mode,nl,c", [("DTU3D", "RGB", 73, 3), ("BU_3DFE", "RGB+depth", 84, 4),
                                               ("DTU3D", "geometry+depth", 73, 2), ("BU_3DFE", "depth", 84, 1)])
def test_config_parser(tmp_path, dataset, mode, nl, c):
    from mvlm_amd import config

    d = config.default_config(dataset, mode, n_views=64)
    d["trainer"] = {"epochs": 100}  # training keys are ignored
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(d))
    cfg = config.load_config(p)
    assert (cfg.n_landmarks, cfg.in_channels, cfg.n_views) == (nl, c, 64)
    assert cfg.filter_view_lines == "quantile" and cfg.heatmap_max_quantile == 0.5
    d["process_3d"]["filter_view_lines"] = "bogus"
    with pytest.raises(ValueError, match="Unknown mode"):
        config.load_config(d)


def test_config_parser_reads_reference_files_when_present():
    """Every configs/*.json of the reference parses, and ``default_config(<stem>)`` reproduces each file's inference
    keys one by one (view count, batch size, line filter, pose ranges, pre-align block, write_renderings)."""
    import dataclasses

    from mvlm_amd import config

    ref = Path("/root/reference/configs")
    if not ref.exists():
        pytest.skip("reference checkout not present")
    files = sorted(ref.glob("*.json"))
    assert sorted(f.stem for f in files) == sorted(config.REFERENCE_CONFIGS)
    for f in files:
        cfg = config.load_config(f)
        assert cfg.n_landmarks in (73, 84) and cfg.in_channels in (1, 2, 3, 4)
        mine = config.load_config(config.default_config(f.stem))
        assert dataclasses.asdict(mine) == dataclasses.asdict(cfg), f.name
        raw = json.loads(f.read_text())
        d = config.default_config(f.stem)
        for block in ("process_3d", "pre-align"):
            assert d[block] == raw[block], (f.name, block)
        assert d["data_loader"]["args"]["n_views"] == raw["data_loader"]["args"]["n_views"]


def test_default_config_carries_each_files_pre_align():
    from mvlm_amd import config
    from mvlm_amd.utils.prealign import is_active

    d = config.default_config("BU_3DFE", "depth")           # BASELINE configs[0]
    assert d["pre-align"]["align_center_of_mass"] is True and d["pre-align"]["scale"] == 20
    assert d["data_loader"]["args"]["n_views"] == 96
    assert config.default_config("BU_3DFE", "depth", n_views=8)["data_loader"]["args"]["n_views"] == 8
    assert config.default_config("BU_3DFE-RGB+depth")["data_loader"]["args"]["n_views"] == 8
    assert not is_active(config.default_config("DTU3D", "RGB")["pre-align"])
    assert is_active(config.default_config("DTU3D-RGB_BU3DFE_RAW")["pre-align"])
    cfg = config.load_config(config.default_config("DTU3D-RGB_Artec3D"))
    assert cfg.write_renderings and cfg.pre_align["write_pre_aligned"] and cfg.pre_align["rot_x"] == -90
    with pytest.raises(ValueError, match="no reference config"):
        config.default_config("DTU3D", "bogus")


def test_unaligned_copy_is_undone_by_the_pre_align_block():
    from mvlm_amd import config
    from mvlm_amd.utils.prealign import aligned, is_active, landmarks_to_original_space
    from mvlm_amd.utils.synthetic import face_like_mesh, unaligned_copy

    mesh = face_like_mesh(20, 8)
    for stem in ("BU_3DFE-depth", "DTU3D-depth-MRI", "DTU3D-RGB_Artec3D", "DTU3D-RGB_infinite"):
        block = config.default_config(stem)["pre-align"]
        raw = unaligned_copy(mesh, block)
        back = aligned(raw, block)
        assert back is not raw and back.to_original is not None and is_active(block)
        centre = mesh.verts.astype(np.float64).mean(0) if block["align_center_of_mass"] else 0.0
        np.testing.assert_allclose(back.verts, mesh.verts - centre, atol=2e-3)
        np.testing.assert_allclose(landmarks_to_original_space(back.verts, back.to_original), raw.verts,
                                   atol=2e-3 / float(block["scale"]) + 1e-5)
        assert aligned(back, block) is back          # never applied twice
    assert aligned(mesh, config.default_config("DTU3D", "RGB")["pre-align"]) is mesh


def test_packed_weights_reproduce_conv(tmp_path):
    """The [tap][cin_pad][cout_pad] layout + folded BN equal torch's conv/BN."""
    from mvlm_amd import arch, weights

    sd = weights.synthetic_state_dict(73, 3, seed=2)
    blob, desc = weights.pack_for_device(sd, 73, 3)
    assert desc.shape == (arch.N_CONV_SLOTS, weights.DESC_INTS)
    slots = arch.conv_slots(73, 3)
    assert sum(s.present and s.ksize != 2 for s in slots) == 138  # + 4 parity restatements of conv11
    s = next(s for s in slots if s.name == "conv4.conv1")
    row = desc[s.index]
    cin_pad, cout_pad = int(row[4]), int(row[5])
    w = blob[row[6]: row[6] + 9 * cin_pad * cout_pad].reshape(9, cin_pad, cout_pad)
    x = torch.randn(1, s.cin, 8, 8)
    scale, shift = blob[row[8]: row[8] + s.cin], blob[row[9]: row[9] + s.cin]
    act = torch.relu(x * torch.from_numpy(scale)[None, :, None, None] + torch.from_numpy(shift)[None, :, None, None])
    wt = torch.from_numpy(w[:, : s.cin, : s.cout].reshape(3, 3, s.cin, s.cout).transpose(3, 2, 0, 1).copy())
    mine = torch.nn.functional.conv2d(act, wt, None, 1, 1)
    ref_act = torch.relu(torch.nn.functional.batch_norm(
        x, torch.from_numpy(sd["conv4.bn1.running_mean"]), torch.from_numpy(sd["conv4.bn1.running_var"]),
        torch.from_numpy(sd["conv4.bn1.weight"]), torch.from_numpy(sd["conv4.bn1.bias"]), False, 0.1, 1e-5))
    ref = torch.nn.functional.conv2d(ref_act, torch.from_numpy(sd["conv4.conv1.weight"]), None, 1, 1)
    assert torch.allclose(mine, ref, atol=1e-5)
    assert (w[:, s.cin:, :] == 0).all() and (w[:, :, s.cout:] == 0).all()
    torch.save({"state_dict": {("module." + k): torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, tmp_path / "ck.pth")
    back = weights.load_state_dict_file(tmp_path / "ck.pth")  # full checkpoint saved from DataParallel
    np.testing.assert_array_equal(back["conv11.bias"], sd["conv11.bias"])
    bad = dict(sd)
    bad.pop("bn3.weight")
    with pytest.raises(KeyError):
        weights.pack_for_device(bad, 73, 3)


def test_pose_table_and_rotations_match_oracle():
    from mvlm_amd.utils.render3d import HipRenderer3D, view_rotations
    from oracle import estimator as oest
    from oracle import poses as oposes

    r = HipRenderer3D.__new__(HipRenderer3D)  # pose logic needs no GPU
    r.__dict__.update(dict(n_views=64, min_x_angle=-40, max_x_angle=40, min_y_angle=-80, max_y_angle=80,
                           min_z_angle=-20, max_z_angle=20, min_scale=1.4, max_scale=1.9, min_tx=-20, max_tx=20,
                           min_ty=-20, max_ty=20))
    np.random.seed(0)
    mine = r.generate_3d_transformations()
    np.random.seed(0)
    np.testing.assert_array_equal(mine, oposes.generate_3d_transformations(64))
    r.n_views = 8
    np.testing.assert_array_equal(r.generate_3d_transformations(), oposes.generate_3d_transformations(8))
    rot = view_rotations(mine)
    for i in range(64):
        np.testing.assert_array_equal(rot[i].reshape(3, 3), oest.view_rotation(*mine[i, :3]))


def test_shard_range_covers_views_in_order():
    from mvlm_amd.parallel import shard_range

    for n in (8, 12, 96, 97, 128):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_create_pipeline_names():
    from mvlm_amd import pipeline

    with pytest.raises(ValueError, match="Unknown pipeline"):
        pipeline.create_pipeline("nope")
    assert {"MediaPipePipeline", "DlibPipeline", "FaceAlignmentPipeline"} <= set(pipeline.__all__)


def test_third_party_detector_adapters(tmp_path, monkeypatch):
    """The wrappers around the reference's third-party detectors: loud ImportError without the library, and with a
    stand-in library the output conventions of mediapipepredictor.py:35-48 / face_alignmentpredictor.py:37-52."""
    import sys
    import types

    from mvlm_amd.prediction import DlibPredictor, FaceAlignmentPredictor, MediaPipePredictor

    for cls in (MediaPipePredictor, DlibPredictor, FaceAlignmentPredictor):
        with pytest.raises(ImportError, match="not installed"):
            cls()
    rs = np.random.RandomState(0)
    stack = rs.rand(3, 256, 256, 4).astype(np.float32)

    # face_alignment stand-in: view 1 has no face
    xy = rs.uniform(-5, 260, (68, 2)).astype(np.float32)   # some points outside the image: the depth lookup clamps
    fa = types.ModuleType("face_alignment")
    fa.LandmarksType = types.SimpleNamespace(TWO_D=1)

    class FaceAlignment:
        def __init__(self, *a, **k):
            self.calls = 0

        def get_landmarks_from_image(self, img, **k):
            assert img.dtype == np.uint8 and img.shape == (256, 256, 3)
            self.calls += 1
            return None if self.calls == 2 else [xy]

    fa.FaceAlignment = FaceAlignment
    monkeypatch.setitem(sys.modules, "face_alignment", fa)
    lms, valid = FaceAlignmentPredictor(device="cpu").predict_landmarks_from_images(stack)
    assert lms.shape == (68, 3, 3) and valid.tolist() == [True, False, True]
    assert np.isnan(lms[:, 1]).all()
    np.testing.assert_array_equal(lms[:, 0, 0], xy[:, 1])
    np.testing.assert_array_equal(lms[:, 0, 1], xy[:, 0])
    r, c = np.clip(xy[:, 1], 0, 255).astype(int), np.clip(xy[:, 0], 0, 255).astype(int)
    np.testing.assert_array_equal(lms[:, 2, 2], stack[2, r, c, 3])

    # mediapipe stand-in
    mp = types.ModuleType("mediapipe")
    mp.ImageFormat = types.SimpleNamespace(SRGB=1)
    mp.Image = lambda image_format, data: data
    pt = [types.SimpleNamespace(x=rs.rand(), y=rs.rand(), z=rs.rand() - 0.5) for _ in range(478)]
    python = types.ModuleType("mediapipe.tasks.python")
    python.BaseOptions = lambda model_asset_path: model_asset_path
    vision = types.ModuleType("mediapipe.tasks.python.vision")
    vision.RunningMode = types.SimpleNamespace(IMAGE=0)
    vision.FaceLandmarkerOptions = lambda **k: k

    class FaceLandmarker:
        @staticmethod
        def create_from_options(options):
            return types.SimpleNamespace(detect=lambda img: types.SimpleNamespace(face_landmarks=[pt] if img[0, 0, 0] < 250 else []))

    vision.FaceLandmarker = FaceLandmarker
    for name, mod in (("mediapipe", mp), ("mediapipe.tasks", types.ModuleType("mediapipe.tasks")),
                      ("mediapipe.tasks.python", python), ("mediapipe.tasks.python.vision", vision)):
        monkeypatch.setitem(sys.modules, name, mod)
    (tmp_path / "m.task").write_bytes(b"x")
    stack[2, 0, 0, 0] = 1.0   # this view: "no face"
    lms, valid = MediaPipePredictor(model_asset_path=tmp_path / "m.task").predict_landmarks_from_images(stack)
    assert lms.shape == (478, 3, 3) and valid.tolist() == [True, True, False]
    want = np.array([[p.y * 256, p.x * 256, -p.z * 256] for p in pt], np.float32)
    np.testing.assert_array_equal(lms[:, 0], want)
    with pytest.raises(FileNotFoundError):
        MediaPipePredictor(model_asset_path=tmp_path / "missing.task")


def test_rng_draw_shortcut_is_the_same_stream():
    """np.random.choice(range(k), 8, replace=True) (estimator3d.py:105) == randint(0, k, size=8)."""
    for k in (3, 4, 5, 17, 32, 33, 48, 64, 100, 478):
        np.random.seed(k)
        a = [np.random.choice(range(k), 8, replace=True) for _ in range(3)]
        sa = np.random.get_state()[1][:8].copy()
        np.random.seed(k)
        b = [np.random.randint(0, k, size=8) for _ in range(3)]
        sb = np.random.get_state()[1][:8].copy()
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and np.array_equal(sa, sb)


def test_one_call_for_all_landmarks_is_the_same_stream():
    """draw_ransac_indices with equal survivor counts: ONE randint(0, k, size=(NL, 8)) == NL calls of the reference's
    np.random.choice(range(k), 8, replace=True) (estimator3d.py:105), values and the RNG state afterwards - for every
    survivor count a view count of the BASELINE configs can produce, and landmark counts 73 / 84 / 478."""
    from mvlm_amd.utils.estimator3d import HipEstimator3D

    draw = HipEstimator3D.draw_ransac_indices
    est = HipEstimator3D.__new__(HipEstimator3D)
    est.verbose = False
    for nl in (73, 84, 478):
        for k in (3, 4, 5, 6, 7, 8, 12, 16, 24, 32, 33, 47, 48, 63, 64, 65, 100, 127, 128, 255, 256, 257, 1000):
            np.random.seed(1000 * nl + k)
            want = np.stack([np.random.choice(range(k), 8, replace=True) for _ in range(nl)]).astype(np.int32)
            tail_want = np.random.randint(0, 1 << 30, size=4)
            np.random.seed(1000 * nl + k)
            got = draw(est, np.full(nl, k))
            tail_got = np.random.randint(0, 1 << 30, size=4)
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(tail_got, tail_want)
    # unequal counts keep the per-landmark path (and its skipping of landmarks with fewer than three lines)
    counts = np.array([5, 2, 9, 0, 3])
    np.random.seed(3)
    want = np.zeros((5, 8), np.int32)
    for i, k in enumerate(counts):
        if k >= 3:
            want[i] = np.random.choice(range(k), 8, replace=True)
    np.random.seed(3)
    np.testing.assert_array_equal(draw(est, counts), want)


def test_vectorised_rotations_equal_scalar_formulation():
    from mvlm_amd.utils.render3d import _view_rotation_scalar, view_rotations

    rs = np.random.RandomState(0)
    t = np.stack([rs.randint(-90, 90, 200), rs.randint(-180, 180, 200), rs.randint(-45, 45, 200)], 1).astype(np.float64)
    got = view_rotations(t)
    want = np.stack([_view_rotation_scalar(*row).ravel() for row in t])
    np.testing.assert_array_equal(got, want)
    t8 = np.array([[30, 15, 0, 0, 0, 0], [-30, -45, 0, 0, 0, 0]], np.float32)
    np.testing.assert_array_equal(view_rotations(t8), np.stack([_view_rotation_scalar(*r[:3]).ravel() for r in t8]))


def test_collapsed_upsampled_conv_equals_conv_of_upsampled():
    """conv3x3(nearest_upsample2x(x)) == four 2x2 convolutions of x, one per output parity."""
    from mvlm_amd.weights import collapse_upsampled_3x3

    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((2, 5, 6, 7))).double()
    w = rs.standard_normal((4, 5, 3, 3)).astype(np.float32)
    want = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"),
                                      torch.from_numpy(w).double(), None, 1, 1)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    got = torch.zeros_like(want)
    for a in (0, 1):
        for b in (0, 1):
            w2 = torch.from_numpy(collapse_upsampled_3x3(w, a, b)).double()
            # window rows (i-1+a, i+a), cols (j-1+b, j+b) of x == rows (i+a, i+a+1) of the padded tensor
            y = torch.nn.functional.conv2d(xp[:, :, a:a + 7, b:b + 8], w2)
            got[:, :, a::2, b::2] = y
    assert torch.allclose(got, want, atol=1e-5)


def test_prealign_round_trip():
    from mvlm_amd.utils.mesh_io import Mesh
    from mvlm_amd.utils.prealign import apply_prealign, landmarks_to_original_space, prealign_matrix

    rs = np.random.RandomState(2)
    verts = (rs.standard_normal((50, 3)) * 3 + [10, -4, 7]).astype(np.float32)
    mesh = Mesh(verts, np.array([[0, 1, 2]], np.int32))
    cfg = dict(align_center_of_mass=True, rot_x=-35, rot_y=10, rot_z=180, scale=20)
    out, m = apply_prealign(mesh, cfg)
    # centre of mass goes to the origin, distances scale by 20
    np.testing.assert_allclose(out.verts.astype(np.float64).mean(0), 0, atol=1e-4)
    d0 = np.linalg.norm(verts[0].astype(np.float64) - verts[1])
    np.testing.assert_allclose(np.linalg.norm(out.verts[0].astype(np.float64) - out.verts[1]), 20 * d0, rtol=1e-5)
    back = landmarks_to_original_space(out.verts.astype(np.float64), m)
    np.testing.assert_allclose(back, verts, atol=1e-4)
    assert np.array_equal(prealign_matrix(verts, dict(scale=1)), np.eye(4))


def test_prealign_equals_the_oracles_call_sequence():
    """Product (closed form S*Ry*Rx*Rz, utils/prealign.py) against the oracle's statement-by-statement vtkTransform
    sequence (oracle/prealign.py, utils3d.py:465-527) for every pre-align block the reference's configs hold."""
    from mvlm_amd import config
    from mvlm_amd.utils.mesh_io import Mesh
    from mvlm_amd.utils.prealign import apply_prealign, landmarks_to_original_space
    from oracle import prealign as opre

    rs = np.random.RandomState(5)
    verts = (rs.standard_normal((200, 3)) * 4 + [6, -3, 2]).astype(np.float32)
    blocks = [config.default_config(stem)["pre-align"] for stem in sorted(config.REFERENCE_CONFIGS)]
    blocks.append(dict(align_center_of_mass=True, rot_x=-35, rot_y=10, rot_z=180, scale=20, write_pre_aligned=False))
    for block in blocks:
        out, m = apply_prealign(Mesh(verts, np.array([[0, 1, 2]], np.int32)), block)
        want, t = opre.pre_transformation(verts, block)
        np.testing.assert_allclose(m, t, rtol=0, atol=1e-12 * max(1.0, float(block["scale"])))
        scale = float(np.abs(want).max()) + 1.0
        np.testing.assert_allclose(out.verts, want, rtol=0, atol=scale * 2.5e-7)   # float32 points on both sides
        lms = rs.standard_normal((7, 3)) * 50
        np.testing.assert_allclose(landmarks_to_original_space(lms, m), opre.landmarks_to_original_space(lms, t), atol=1e-9)


def test_cli_shards_scans_across_ranks():
    from mvlm_amd.__main__ import shard_files

    files = [f"s{i}.obj" for i in range(10)]
    parts = [shard_files(files, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == sorted(files) and parts[0] == ["s0.obj", "s4.obj", "s8.obj"]
    assert shard_files(files, 0, 1) == files


def test_visualize_image_stack_png_dump(tmp_path):
    """general_pipeline.py:133-146: `<stem>_<ii>.png` per view from the RGB planes, ValueError for a missing folder."""
    import types

    from PIL import Image

    from mvlm_amd.pipeline.general_pipeline import Pipeline

    rs = np.random.RandomState(0)
    stack = (rs.randint(0, 256, (3, 256, 256, 4)) / 255).astype(np.float32)
    me = types.SimpleNamespace(render_image_folder=tmp_path)
    Pipeline.visualize_image_stack(me, stack, tmp_path / "scan.obj", first_index=4)
    for i in range(3):
        png = np.asarray(Image.open(tmp_path / f"scan_{4 + i:02d}.png"))
        np.testing.assert_array_equal(png, np.uint8(stack[i, :, :, 0:3] * 255))
    me.render_image_folder = tmp_path / "missing"
    with pytest.raises(ValueError, match="does not exist"):
        Pipeline.visualize_image_stack(me, stack, tmp_path / "scan.obj")


def test_host_allocator_hint_is_idempotent_and_optional(monkeypatch):
    """utils/hostmem.py: applied once per process; MVLM_HOST_MALLOC_TUNING=0 switches it off."""
    from mvlm_amd.utils import hostmem

    monkeypatch.setattr(hostmem, "_applied", None)
    monkeypatch.setenv("MVLM_HOST_MALLOC_TUNING", "0")
    assert hostmem.retain_freed_host_memory() is False
    monkeypatch.setattr(hostmem, "_applied", None)
    monkeypatch.delenv("MVLM_HOST_MALLOC_TUNING")
    first = hostmem.retain_freed_host_memory()
    assert first in (True, False) and hostmem.retain_freed_host_memory() is first
    a = np.ones(4 << 20, np.uint8)  # allocations keep working either way
    assert int(a.sum()) == 4 << 20


def test_bench_parent_refuses_without_enough_gpus():
    """`python bench.py --gpus N` as a plain process is the launcher of its N ranks; on a machine that shows fewer GPUs
    it exits non-zero with a clear message, prints nothing on stdout and never touches a GPU (this container has none)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MVLM_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "needs 2 visible GPUs" in r.stderr
