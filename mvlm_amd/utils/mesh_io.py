"""Mesh ingest on the host: Wavefront OBJ + same-stem JPEG texture.

Mirrors what ``obj_to_actor`` obtains from vtkOBJReader / vtkJPEGReader
(reference src/mvlm/utils/utils3d.py:10-85): geometry as float32 points, one
texture coordinate per point (points are duplicated where a vertex is used with
several ``vt`` indices), polygons as triangle fans, the ``.mtl`` ignored, the
texture looked up as ``<stem>.jpg`` next to the file and silently dropped if it
cannot be read (:26-36).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Union

import numpy as np


@dataclass
class Mesh:
    """Host copy of a triangle mesh plus lazily created device handles.

    This object is the opaque ``pd`` handle that ``multiview_render`` returns and
    ``project_landmarks_to_surface`` receives (general_pipeline.py:83, :107).
    """

    verts: np.ndarray                 # [V,3] float32
    tris: np.ndarray                  # [T,3] int32
    uvs: np.ndarray | None = None     # [V,2] float32
    texture: np.ndarray | None = None  # [H,W,3] uint8, row 0 = top of the image
    path: Path | None = None
    _device: dict = field(default_factory=dict, repr=False)
    # 4x4 matrix of the config's "pre-align" block this mesh has been through (utils/prealign.py); landmarks found on
    # it go back to the file's own coordinates through the inverse (utils3d.py:505-527)
    to_original: np.ndarray | None = None
    # The texture as the bytes of its JPEG file, not decoded yet (load_obj / load_mesh with decode="device").  The renderer's
    # upload hands them to the device decoder (mvlm_mesh_upload_jpeg); ``texture`` decodes them with libjpeg on first use.
    texture_jpeg: bytes | None = field(default=None, repr=False)

    @property
    def n_verts(self) -> int:
        return int(self.verts.shape[0])

    @property
    def n_tris(self) -> int:
        return int(self.tris.shape[0])


def _lazy_texture(self):
    if self._texture is None and self.texture_jpeg is not None:
        self._texture = decode_texture_bytes(self.texture_jpeg)
        if self._texture is None:
            self.texture_jpeg = None  # "if we cannot load the texture, we just ignore it" (utils3d.py:35-36)
    return self._texture


def _set_texture(self, value):
    self._texture = value


Mesh.texture = property(_lazy_texture, _set_texture)  # (the dataclass' __init__ assigns through the setter)


def decode_texture_bytes(data: bytes):
    """Image file bytes -> [H,W,3] uint8 with libjpeg (Pillow), None if it does not decode."""
    try:
        import io

        from PIL import Image

        with Image.open(io.BytesIO(data)) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")
            return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
    except Exception:  # noqa: BLE001
        return None


def _parse_obj(text: str):
    pos: list[tuple[float, float, float]] = []
    tex: list[tuple[float, float]] = []
    corners: dict[tuple[int, int], int] = {}
    out_v: list[int] = []
    out_t: list[int] = []
    tris: list[tuple[int, int, int]] = []
    for line in text.splitlines():
        if not line or line[0] == "#":
            continue
        parts = line.split()
        if not parts:
            continue
        tag = parts[0]
        if tag == "v" and len(parts) >= 4:
            pos.append((float(parts[1]), float(parts[2]), float(parts[3])))
        elif tag == "vt" and len(parts) >= 3:
            tex.append((float(parts[1]), float(parts[2])))
        elif tag == "f" and len(parts) >= 4:
            ids = []
            for tok in parts[1:]:
                f = tok.split("/")
                vi = int(f[0])
                vi = vi - 1 if vi > 0 else len(pos) + vi
                ti = -1
                if len(f) > 1 and f[1]:
                    ti = int(f[1])
                    ti = ti - 1 if ti > 0 else len(tex) + ti
                key = (vi, ti)
                idx = corners.get(key)
                if idx is None:
                    idx = corners[key] = len(out_v)
                    out_v.append(vi)
                    out_t.append(ti)
                ids.append(idx)
            for k in range(1, len(ids) - 1):  # polygon -> fan
                tris.append((ids[0], ids[k], ids[k + 1]))
    return pos, tex, out_v, out_t, tris


def _read_texture(jpg: Path):
    try:
        from PIL import Image

        with Image.open(jpg) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")
            return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))  # one copy out of the decoder's buffer
    except Exception:  # noqa: BLE001 - "if we cannot load the texture, we just ignore it" (utils3d.py:35-36)
        return None


def _read_obj_native(path: Path, any_format: bool = False):
    """Parse with the library's readers: mvlm_obj_read (mvlm_amd/csrc/obj_reader.hip) or, with ``any_format``,
    mvlm_mesh_read (mesh_readers.hip: .obj .ply .stl .vtk .wrl by extension)."""
    import ctypes as C

    from .. import _lib

    lib = _lib.load()
    handle = C.c_void_p()
    err = C.create_string_buffer(512)
    read = lib.mvlm_mesh_read if any_format else lib.mvlm_obj_read
    rc = read(str(path).encode(), C.byref(handle), err, len(err))
    if rc != 0:
        raise ValueError(err.value.decode(errors="replace") or f"File {path}: mesh reader failed ({rc})")
    try:
        nv, nt, has_uv = C.c_int64(), C.c_int64(), C.c_int()
        lib.mvlm_obj_info(handle, C.byref(nv), C.byref(nt), C.byref(has_uv))
        verts = np.empty((nv.value, 3), np.float32)
        tris = np.empty((nt.value, 3), np.int32)
        uvs = np.empty((nv.value, 2), np.float32) if has_uv.value else None
        lib.mvlm_obj_copy(handle, _lib.as_ptr(verts, C.c_float), _lib.as_ptr(uvs, C.c_float) if uvs is not None else None,
                          _lib.as_ptr(tris, C.c_int32))
    finally:
        lib.mvlm_obj_free(handle)
    return verts, tris, uvs


def _read_obj_python(path: Path):
    """The same rules in plain Python (the statement tests check the native reader against)."""
    pos, tex, out_v, out_t, tris = _parse_obj(path.read_text(errors="replace"))
    if len(pos) == 0:
        raise ValueError(f"File {path} does not contain any points.")  # utils3d.py:20-21
    pos_a = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    if len(tris) == 0:
        # a point cloud: keep the points, nothing to render or to snap to
        return pos_a, np.zeros((0, 3), np.int32), None
    v_idx = np.asarray(out_v, dtype=np.int64)
    if v_idx.min() < 0 or v_idx.max() >= len(pos):
        raise ValueError(f"File {path} references a vertex that does not exist.")
    verts = pos_a[v_idx]
    uvs = None
    t_idx = np.asarray(out_t, dtype=np.int64)
    if len(tex) > 0 and (t_idx >= 0).any():
        tex_a = np.asarray(tex, dtype=np.float32).reshape(-1, 2)
        safe = np.clip(t_idx, 0, len(tex) - 1)
        uvs = np.where((t_idx >= 0)[:, None], tex_a[safe], np.float32(0)).astype(np.float32)
    return np.ascontiguousarray(verts), np.asarray(tris, dtype=np.int32).reshape(-1, 3), uvs


def load_obj(path: Union[Path, str], load_texture: bool = True, reader: str = "native", decode: str = "device") -> Mesh:
    """OBJ + same-stem ``.jpg`` -> Mesh.

    ``decode="device"`` (default): the JPEG file is only read; the Mesh carries its bytes (``texture_jpeg``) and the renderer's
    upload decodes them on the GPU (``mvlm_mesh_upload_jpeg``, byte for byte libjpeg's result).  ``mesh.texture`` still
    gives the pixels - decoded with libjpeg on first use.  ``decode="host"``: libjpeg on a second thread while the
    geometry is parsed (both release the GIL), as in rounds 1-4."""
    path = Path(path)
    if not path.is_file():
        raise ValueError(f"File {path} does not exist.")  # utils3d.py:13-14
    if decode not in ("device", "host"):
        raise ValueError(f"unknown texture decode mode: {decode}")
    jpg = path.with_suffix(".jpg")
    tex_job = None
    jpeg_bytes = None
    if load_texture and jpg.exists():
        if decode == "device":
            try:
                jpeg_bytes = jpg.read_bytes()
            except OSError:
                jpeg_bytes = None
        else:
            import threading

            box: list = [None]
            tex_job = threading.Thread(target=lambda: box.__setitem__(0, _read_texture(jpg)), daemon=True)
            tex_job.start()
    try:
        if reader == "native":
            verts, tris, uvs = _read_obj_native(path)
        elif reader == "python":
            verts, tris, uvs = _read_obj_python(path)
        else:
            raise ValueError(f"unknown OBJ reader: {reader}")
    finally:
        if tex_job is not None:
            tex_job.join()
    texture = box[0] if (tex_job is not None and uvs is not None) else None  # utils3d.py:26: only with tcoords
    return Mesh(verts, tris, uvs, texture, path, texture_jpeg=jpeg_bytes if uvs is not None else None)


SURFACE_SUFFIXES = (".obj", ".wrl", ".vtk", ".stl", ".ply")  # Utils3D.multi_read_surface, utils3d.py:389-423


def find_texture(path: Union[Path, str], texture_file_name=None) -> Path | None:
    """Texture file of a scan by the rules of ``Utils3D.multi_read_texture`` (utils3d.py:425-441): same stem
    with ``.bmp``, then ``.png``, then ``.jpg`` - each one found replaces the earlier candidate, so .jpg wins
    over .png over .bmp - and for BU-3DFE raw scans ``*RAW.wrl`` the ``*F3D.bmp`` next to it wins over all."""
    if texture_file_name is not None:
        return Path(texture_file_name)
    path = Path(path)
    found = None
    for suffix in (".bmp", ".png", ".jpg"):
        cand = path.with_suffix(suffix)
        if cand.is_file():
            found = cand
    name = str(path)
    if name.find("RAW.wrl") > 0:  # "BU-3DFE RAW file hack" (utils3d.py:438-441)
        cand = Path(name.replace("RAW.wrl", "F3D.bmp"))
        if cand.is_file():
            found = cand
    return found


def load_mesh(path: Union[Path, str], load_texture: bool = True, texture_file_name=None, decode: str = "device") -> Mesh:
    """Any surface format of the reference's legacy reader (``.obj .wrl .vtk .stl .ply``, utils3d.py:389-423)
    with the texture looked up by ``find_texture`` (``.bmp / .png / .jpg``, :425-462) -> Mesh.  ``load_obj`` is
    the live pipeline's stricter OBJ + same-stem JPEG ingest (utils3d.py:10-36).  A ``.jpg`` texture is decoded on the
    device at upload time unless ``decode="host"`` (see ``load_obj``); ``.bmp`` / ``.png`` always on the host."""
    path = Path(path)
    if not path.is_file():
        raise ValueError(f"File {path} does not exist.")
    if path.suffix.lower() not in SURFACE_SUFFIXES:
        raise ValueError(f"Can not read files with extension {path.suffix}")  # utils3d.py:421-422
    if decode not in ("device", "host"):
        raise ValueError(f"unknown texture decode mode: {decode}")
    tex_path = find_texture(path, texture_file_name) if load_texture else None
    tex_job, box = None, [None]
    jpeg_bytes = None
    if tex_path is not None and tex_path.suffix.lower() in (".bmp", ".png", ".jpg") and tex_path.is_file():
        if decode == "device" and tex_path.suffix.lower() == ".jpg":
            try:
                jpeg_bytes = tex_path.read_bytes()
            except OSError:
                jpeg_bytes = None
        else:
            import threading

            tex_job = threading.Thread(target=lambda: box.__setitem__(0, _read_texture(tex_path)), daemon=True)
            tex_job.start()
    try:
        verts, tris, uvs = _read_obj_native(path, any_format=True)
    finally:
        if tex_job is not None:
            tex_job.join()
    texture = box[0] if uvs is not None else None
    return Mesh(verts, tris, uvs, texture, path, texture_jpeg=jpeg_bytes if uvs is not None else None)


def write_obj(path: Union[Path, str], verts: np.ndarray, tris: np.ndarray, uvs: np.ndarray | None = None,
              texture: np.ndarray | None = None, jpeg_quality: int = 95) -> None:
    """Small writer used by the synthetic-mesh generator and the tests."""
    path = Path(path)
    lines = [f"v {x:.6f} {y:.6f} {z:.6f}" for x, y, z in np.asarray(verts, dtype=np.float64)]
    if uvs is not None:
        lines += [f"vt {u:.6f} {v:.6f}" for u, v in np.asarray(uvs, dtype=np.float64)]
        lines += [f"f {a + 1}/{a + 1} {b + 1}/{b + 1} {c + 1}/{c + 1}" for a, b, c in np.asarray(tris)]
    else:
        lines += [f"f {a + 1} {b + 1} {c + 1}" for a, b, c in np.asarray(tris)]
    path.write_text("\n".join(lines) + "\n")
    if texture is not None:
        from PIL import Image

        Image.fromarray(np.asarray(texture, dtype=np.uint8)).save(path.with_suffix(".jpg"), quality=jpeg_quality)
