"""Multi-view renderer slot: all camera poses of a mesh rasterised on the MI355X.

Drop-in for the reference's ``ObjVTKRenderer3D`` (src/mvlm/utils/render3d.py):
same constructor arguments, same pose generation (global numpy RNG, same draw
order), same ``multiview_render(path) -> (image_stack, transform_stack, mesh)``
contract, with the per-pose VTK offscreen loop (:139-170) replaced by one
batched HIP launch set (mvlm_amd/csrc/raster.hip) through ``mvlm_render``.
"""
from __future__ import annotations

import ctypes as C
import time
from pathlib import Path

import numpy as np

from .. import _lib
from .mesh_io import Mesh, load_obj

__all__ = ["HipRenderer3D", "view_rotations", "upload_mesh"]


def _view_rotation_scalar(rx, ry, rz) -> np.ndarray:
    """One view, evaluated exactly like the reference's estimator (estimator3d.py:8-15, :57)."""
    rx, ry, rz = (np.deg2rad(v) for v in (rx, ry, rz))
    mx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    my = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    mz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    return (my @ mx) @ mz


def view_rotations(transform_stack: np.ndarray) -> np.ndarray:
    """[N,>=3] (rx, ry, rz) degrees -> [N,9] float64 row-major M = Ry @ Rx @ Rz.

    The order is VTK's RotateY / RotateX / RotateZ pre-multiplication
    (render3d.py:140-144), which the estimator inverts (estimator3d.py:57).
    Evaluated for all views at once; the first and last view are re-evaluated with the
    reference's scalar formulation and, should numpy's array kernels ever round differently
    from its scalar ones on this machine, the whole table falls back to the scalar path.
    """
    t = np.asarray(transform_stack)
    n = t.shape[0]
    if n == 0:
        return np.zeros((0, 9))
    # the fixed 8-view table (render3d.py:94-111) is the same for every scan, and a caller that repeats a pose table gets the
    # same answer: the last few tables are kept (read-only), keyed by the angles' bytes
    key = (t.dtype.str, n, t[:, :3].tobytes())
    hit = _ROTATION_MEMO.get(key)
    if hit is not None:
        return hit
    out = _view_rotations_uncached(t, n)
    out.setflags(write=False)
    if len(_ROTATION_MEMO) >= 8:
        _ROTATION_MEMO.pop(next(iter(_ROTATION_MEMO)))
    _ROTATION_MEMO[key] = out
    return out


_ROTATION_MEMO: dict = {}


def _view_rotations_uncached(t: np.ndarray, n: int) -> np.ndarray:
    r = np.deg2rad(t[:, :3].astype(np.float64))
    c, s = np.cos(r), np.sin(r)
    mx, my, mz = np.zeros((n, 3, 3)), np.zeros((n, 3, 3)), np.zeros((n, 3, 3))
    mx[:, 0, 0] = 1; mx[:, 1, 1] = c[:, 0]; mx[:, 1, 2] = -s[:, 0]; mx[:, 2, 1] = s[:, 0]; mx[:, 2, 2] = c[:, 0]
    my[:, 0, 0] = c[:, 1]; my[:, 0, 2] = s[:, 1]; my[:, 1, 1] = 1; my[:, 2, 0] = -s[:, 1]; my[:, 2, 2] = c[:, 1]
    mz[:, 0, 0] = c[:, 2]; mz[:, 0, 1] = -s[:, 2]; mz[:, 1, 0] = s[:, 2]; mz[:, 1, 1] = c[:, 2]; mz[:, 2, 2] = 1
    out = np.matmul(np.matmul(my, mx), mz).reshape(n, 9)
    for i in {0, n - 1}:
        if not np.array_equal(out[i], _view_rotation_scalar(*t[i, :3]).ravel()):
            return np.stack([_view_rotation_scalar(*t[k, :3]).ravel() for k in range(n)])
    return out


def upload_mesh(ctx: "_lib.Context", mesh: Mesh) -> C.c_void_p:
    """Copy a host mesh to the context's device once; cached on the Mesh object."""
    key = id(ctx)
    ent = mesh._device.get(key)
    if ent is not None:
        return ent[0]
    if mesh.n_tris == 0:
        raise ValueError("mesh has no triangles to render")
    verts = np.ascontiguousarray(mesh.verts, dtype=np.float32)
    tris = np.ascontiguousarray(mesh.tris, dtype=np.int32)
    uvs = None if mesh.uvs is None else np.ascontiguousarray(mesh.uvs, dtype=np.float32)
    handle = C.c_void_p()
    uploaded = False
    ahead = getattr(mesh, "_texture_ahead", None)
    if ahead is not None and ahead.ctx is ctx and ahead.handle:
        # the texture was decoded on the device while the geometry was parsed (HipRenderer3D.load_mesh): its buffer
        # becomes the mesh's
        consumed = C.c_int(0)
        ctx.check(ctx.lib.mvlm_mesh_upload_texture(
            ctx.handle, _lib.as_ptr(verts, C.c_float), None if uvs is None else _lib.as_ptr(uvs, C.c_float), mesh.n_verts,
            _lib.as_ptr(tris, C.c_int32), mesh.n_tris, ahead.handle, C.byref(consumed), C.byref(handle)), ValueError)
        if consumed.value:
            ahead.handle = None
        mesh._texture_ahead = None  # (an unused handle is freed with its owner)
        uploaded = True
    elif uvs is not None and mesh.texture_jpeg is not None and getattr(mesh, "_texture", None) is None:
        # the texture is still the JPEG file's bytes: entropy decoding, inverse DCT, upsampling and colour conversion on
        # the device (csrc/jpeg.hip).  2 = a kind of JPEG that decoder does not take: libjpeg on the host below
        raw = np.frombuffer(mesh.texture_jpeg, dtype=np.uint8)
        rc = ctx.lib.mvlm_mesh_upload_jpeg(
            ctx.handle, _lib.as_ptr(verts, C.c_float), _lib.as_ptr(uvs, C.c_float), mesh.n_verts, _lib.as_ptr(tris, C.c_int32),
            mesh.n_tris, _lib.as_ptr(raw, C.c_uint8), raw.size, C.byref(handle))
        if rc == 0:
            uploaded = True
        elif rc != 2:
            ctx.check(rc, ValueError)
    if not uploaded:
        tex = None if mesh.texture is None else np.ascontiguousarray(mesh.texture, dtype=np.uint8)
        ctx.check(ctx.lib.mvlm_mesh_upload(
            ctx.handle, _lib.as_ptr(verts, C.c_float), None if uvs is None else _lib.as_ptr(uvs, C.c_float), mesh.n_verts,
            _lib.as_ptr(tris, C.c_int32), mesh.n_tris, None if tex is None else _lib.as_ptr(tex, C.c_uint8),
            0 if tex is None else tex.shape[0], 0 if tex is None else tex.shape[1], C.byref(handle)), ValueError)

    class _Owner:  # frees the device copy with the Mesh
        def __init__(self, ctx, h):
            self.ctx, self.h = ctx, h

        def __del__(self):
            try:
                self.ctx.lib.mvlm_mesh_free(self.ctx.handle, self.h)
            except Exception:  # noqa: BLE001
                pass

    mesh._device[key] = (handle, _Owner(ctx, handle))
    return handle


class _DevicePointer:
    """A borrowed device address with the one method the C-ABI wrappers ask of a tensor."""

    def __init__(self, ptr: int):
        self._ptr = int(ptr)

    def data_ptr(self) -> int:
        return self._ptr


class _TextureAhead:
    """A texture decoded on the device ahead of its mesh (mvlm_texture_from_jpeg); gives the buffer back unless a mesh
    upload has taken it over."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def __del__(self):
        try:
            if self.handle:
                self.ctx.lib.mvlm_texture_free(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001
            pass


def decode_texture_ahead(ctx: "_lib.Context", jpeg_bytes: bytes):
    """JPEG bytes -> _TextureAhead on the context's device, or None when the device decoder does not take the file."""
    raw = np.frombuffer(jpeg_bytes, dtype=np.uint8)
    handle = C.c_void_p()
    rc = ctx.lib.mvlm_texture_from_jpeg(ctx.handle, _lib.as_ptr(raw, C.c_uint8), raw.size, C.byref(handle))
    if rc == 0:
        return _TextureAhead(ctx, handle)
    if rc != 2:
        ctx.check(rc, ValueError)
    return None


class HipRenderer3D:
    def __init__(self, n_views: int = 8, image_size: tuple = (256, 256), offscreen: bool = True,
                 min_x_angle: int = -40, max_x_angle: int = 40, min_y_angle: int = -80, max_y_angle: int = 80,
                 min_z_angle: int = -20, max_z_angle: int = 20, min_scale: float = 1.4, max_scale: float = 1.9,
                 min_tx: int = -20, max_tx: int = 20, min_ty: int = -20, max_ty: int = 20, device: int = 0,
                 verbose: bool = True, shading: str = "texture", subpixel_bits: int = 8):
        if tuple(image_size) != (256, 256):
            raise ValueError("the HIP renderer is built for 256x256 views (general_pipeline.py:57)")
        self.n_views = n_views
        self.image_size = tuple(image_size)
        self.offscreen = offscreen  # accepted for signature compatibility; there is no window
        self.min_x_angle, self.max_x_angle = min_x_angle, max_x_angle
        self.min_y_angle, self.max_y_angle = min_y_angle, max_y_angle
        self.min_z_angle, self.max_z_angle = min_z_angle, max_z_angle
        self.min_scale, self.max_scale = min_scale, max_scale
        self.min_tx, self.max_tx = min_tx, max_tx
        self.min_ty, self.max_ty = min_ty, max_ty
        self.slack = 5
        self.side_length = max([150 - (-150), 150 - (-150)]) * 1.0 / 2  # render3d.py:50
        self.verbose = verbose
        if shading not in ("texture", "geometry"):
            raise ValueError("shading must be 'texture' (the reference's unlit render) or 'geometry'")
        # "geometry": build-defined shaded plane for models trained on geometry renderings
        self.shading = shading
        # vertex snap of the rasteriser, 2^-bits pixel: GL_SUBPIXEL_BITS of the OpenGL whose images are to be matched (the
        # reference's pixels depend on it; 8 = GPUs, 4 = the software GL behind tests/golden/gl_raster.npz)
        if subpixel_bits not in (4, 5, 6, 7, 8):
            raise ValueError("subpixel_bits must be 4..8")
        self.subpixel_bits = int(subpixel_bits)
        # "pre-align" block of a Deep-MVLM config (utils/prealign.py; utils3d.py:465-503): applied to every mesh this
        # renderer loads, the mesh handle it returns carries the matrix (Mesh.to_original)
        self.pre_align: dict | None = None
        # where load_mesh's JPEG texture is decoded: "device" (the upload decodes the file's bytes on the GPU, csrc/jpeg.hip)
        # or "host" (libjpeg through Pillow at load time); the pixels are the same bytes either way
        self.texture_decode = "device"
        self.ctx = _lib.get_context(device)

    # ---- pose table (render3d.py:79-112) ----------------------------------------------
    def random_transform(self, size=1):
        rx = np.random.randint(self.min_x_angle, self.max_x_angle, size=size)
        ry = np.random.randint(self.min_y_angle, self.max_y_angle, size=size)
        rz = np.random.randint(self.min_z_angle, self.max_z_angle, size=size)
        # drawn but unused, kept so the global RNG advances exactly as in the reference
        scale = np.random.uniform(self.min_scale, self.max_scale, size=size)
        tx = np.random.randint(self.min_tx, self.max_tx, size=size)
        ty = np.random.randint(self.min_ty, self.max_ty, size=size)
        return np.stack((rx, ry, rz, scale, tx, ty), axis=1)

    def generate_3d_transformations(self):
        if self.n_views == 8:
            table = [[rx, ry, 0, 0, 0, 0] for rx in (30, -30) for ry in (15, -15, 45, -45)]
            return np.array(table, dtype=np.float32)
        return self.random_transform(size=self.n_views)

    # ---- rendering --------------------------------------------------------------------
    def render_device(self, mesh: Mesh, transform_stack: np.ndarray, rot: np.ndarray | None = None, out=None):
        """Poses -> torch.float32 [N,256,256,4] on the device (RGB + depth, /255, flipped).
        Only enqueues work; ``check()`` reports a deferred failure after the caller's sync.
        ``out``: an existing stack of that shape to render into (the pipeline reuses one buffer per view count)."""
        import torch

        n = int(transform_stack.shape[0])
        dev = torch.device("cuda", self.ctx.device)
        if out is None:
            out = torch.empty((n, 256, 256, 4), dtype=torch.float32, device=dev)
        elif tuple(out.shape) != (n, 256, 256, 4) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("render_device: out must be a contiguous float32 [N,256,256,4] tensor on the renderer's GPU")
        rot = np.ascontiguousarray(view_rotations(transform_stack) if rot is None else rot, dtype=np.float64)
        handle = upload_mesh(self.ctx, mesh)
        self.ctx.bind_current_stream(torch, dev)
        mode = (1 if self.shading == "geometry" else 0, self.subpixel_bits)
        if getattr(self.ctx, "_render_mode", None) != mode:  # (renderers of one GPU share the context)
            self.ctx.check(self.ctx.lib.mvlm_set_render_shading(self.ctx.handle, mode[0]))
            self.ctx.check(self.ctx.lib.mvlm_set_render_subpixel_bits(self.ctx.handle, mode[1]))
            self.ctx._render_mode = mode
        self.ctx.check(self.ctx.lib.mvlm_render(self.ctx.handle, handle, _lib.as_ptr(rot, C.c_double), n,
                                                C.c_void_p(out.data_ptr())))
        return out

    def rotations_device(self):
        """The last render's rotation table where the rasteriser keeps it on the device (f64[N,9]; valid until this context's
        next render): an object with ``data_ptr()`` that ``HipEstimator3D.lines_device(rot_dev=...)`` takes - the rays of the
        same views need the same matrices (estimator3d.py:57), so the table crosses PCIe once per mesh."""
        p = C.c_void_p()
        self.ctx.check(self.ctx.lib.mvlm_render_rotations_dev(self.ctx.handle, C.byref(p)))
        return _DevicePointer(p.value)

    def check(self):
        """Wait for the stream and raise if an enqueued render failed."""
        self.ctx.check(self.ctx.lib.mvlm_render_check(self.ctx.handle))

    def render_3d_multi_rgb_geometry_depth(self, transform_stack, file_name):
        """Signature of render3d.py:114; returns the *unscaled* 0..255 stack like the reference."""
        tt = time.time()
        mesh = file_name if isinstance(file_name, Mesh) else load_obj(file_name)
        if self.verbose:
            print("Render [1] - Setup time: ", f"{time.time() - tt:08.6f} s")
        tt = time.time()
        stack = self.render_device(mesh, np.asarray(transform_stack))
        image_stack = (stack * 255.0).round().cpu().numpy()
        self.check()
        if self.verbose:
            print("Render [2] - Render", f"{time.time() - tt:08.6f} s")
        return image_stack, mesh

    def _check_file(self, file_name: Path):
        file_name = Path(file_name)
        if not file_name.exists():
            raise FileNotFoundError(f"File {file_name} does not exist")
        if not file_name.is_file():
            raise FileNotFoundError(f"File {file_name} is not a file")
        if not file_name.suffix == ".obj":
            raise ValueError(f"File {file_name} is not an .obj file. Only .obj files are supported.")
        return file_name

    def multiview_render(self, file_name: Path, load_texture: bool = True):
        """render3d.py:179-193 - numpy results, as the slot contract requires.  ``load_texture``: see ``load_mesh``."""
        t = time.time()
        file_name = self._check_file(file_name)
        if self.verbose:
            print("Render [0] - Prepare", f"{time.time() - t:08.6f} s")
        transformation_stack = self.generate_3d_transformations()
        mesh = self.load_mesh(file_name, load_texture=load_texture)
        image_stack = self.render_device(mesh, transformation_stack).cpu().numpy()
        self.check()
        return image_stack, transformation_stack, mesh

    def load_mesh(self, file_name: Path, load_texture: bool = True) -> Mesh:
        """OBJ (+ texture) from disk, through the ``pre_align`` block when one is set.
        ``load_texture=False`` leaves the JPEG alone (not even read), and the consumer of a depth / geometry model's views
        reads no texture-shaded plane.  It is
        an argument of the call, decided by the caller that knows the consumer (``Pipeline._texture_needed``) - the
        renderer keeps no such state, so its public entry points load the texture like the reference (utils3d.py:26-36)."""
        from .prealign import aligned

        jpg = Path(file_name).with_suffix(".jpg")
        if not (load_texture and self.texture_decode == "device" and jpg.exists()):
            return aligned(load_obj(file_name, load_texture=load_texture, decode=self.texture_decode), self.pre_align)
        # The texture is decoded on the device by a second thread (read the .jpg, unstuff, GPU decode: 1.8 ms at 2048^2)
        # while this one parses the geometry (2.7 ms): the scan is ready when the slower of the two is.
        import threading

        box: list = [None, None, None, None]  # bytes, device texture, exception, host-decoded pixels (device refused)

        def texture_job():
            try:
                box[0] = jpg.read_bytes()
                box[1] = decode_texture_ahead(self.ctx, box[0])
                if box[1] is None:
                    # not a JPEG the device takes (progressive ones are common among exported textures): libjpeg on the host,
                    # HERE, beside the geometry parse - not afterwards on the caller's thread inside upload_mesh
                    from .mesh_io import decode_texture_bytes

                    box[3] = decode_texture_bytes(box[0])
            except Exception as e:  # noqa: BLE001 - "if we cannot load the texture, we just ignore it" (utils3d.py:35-36) ...
                box[2] = e        # ... but a failing GPU call is not a texture problem: raised below

        job = threading.Thread(target=texture_job, daemon=True)
        job.start()
        try:
            mesh = load_obj(file_name, load_texture=False)
        finally:
            job.join()
        if isinstance(box[2], ValueError):
            raise box[2]
        if mesh.uvs is not None and box[0] is not None:  # utils3d.py:26: only with tcoords
            mesh.texture_jpeg = box[0]
            mesh._texture_ahead = box[1]
            if box[1] is None:
                # decoded on the host by the texture thread (or not decodable at all: ignored, utils3d.py:35-36): the upload
                # goes straight to mvlm_mesh_upload, without a second look at the header
                mesh._texture = box[3]
                if box[3] is None:
                    mesh.texture_jpeg = None
        return aligned(mesh, self.pre_align)

    def multiview_render_device(self, file_or_mesh, transformation_stack=None):
        """Same, but the image stack stays in HBM (used by the fused pipeline path)."""
        mesh = file_or_mesh if isinstance(file_or_mesh, Mesh) else self.load_mesh(self._check_file(file_or_mesh))
        if transformation_stack is None:
            transformation_stack = self.generate_3d_transformations()
        return self.render_device(mesh, transformation_stack), transformation_stack, mesh
