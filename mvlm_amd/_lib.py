"""ctypes binding of libmvlm_hip.so (the C ABI declared in include/mvlm_hip.h).

There is no CPU fallback: if the library is missing or the GPU is not a gfx950
the import of the product path fails loudly.  ``build()`` (re)compiles the
library in-tree with hipcc (cross-compiles without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# MVLM_HIP_LIB: load an alternative build of the same ABI (kernel ablation experiments)
LIB_PATH = Path(os.environ["MVLM_HIP_LIB"]) if os.environ.get("MVLM_HIP_LIB") else _HERE / "lib" / "libmvlm_hip.so"
CSRC = _HERE / "csrc"

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); mirrors include/mvlm_hip.h one to one
SIGNATURES = {
    "mvlm_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mvlm_ctx_destroy": (None, [C.c_void_p]),
    "mvlm_last_error": (C.c_char_p, [C.c_void_p]),
    "mvlm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mvlm_synchronize": (C.c_int, [C.c_void_p]),
    "mvlm_build_arch": (C.c_char_p, []),
    "mvlm_obj_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]),
    "mvlm_mesh_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]),
    "mvlm_obj_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "mvlm_obj_copy": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_int32_p]),
    "mvlm_obj_free": (None, [C.c_void_p]),
    "mvlm_mesh_upload": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, c_int32_p, C.c_int, c_uint8_p,
                                   C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mvlm_mesh_free": (None, [C.c_void_p, C.c_void_p]),
    "mvlm_jpeg_info": (C.c_int, [c_uint8_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "mvlm_mesh_upload_jpeg": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, c_int32_p, C.c_int, c_uint8_p, C.c_size_t,
                                        C.POINTER(C.c_void_p)]),
    "mvlm_texture_from_jpeg": (C.c_int, [C.c_void_p, c_uint8_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "mvlm_texture_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvlm_texture_free": (None, [C.c_void_p, C.c_void_p]),
    "mvlm_mesh_upload_texture": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, c_int32_p, C.c_int, C.c_void_p,
                                           C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "mvlm_jpeg_decode": (C.c_int, [C.c_void_p, c_uint8_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]),
    "mvlm_render": (C.c_int, [C.c_void_p, C.c_void_p, c_double_p, C.c_int, C.c_void_p]),
    "mvlm_render_check": (C.c_int, [C.c_void_p]),
    "mvlm_render_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_render_get_profile": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, c_float_p, C.c_int]),
    "mvlm_render_rotations_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mvlm_gather_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mvlm_gather_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mvlm_allgather_maxima": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mvlm_set_render_shading": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_set_render_subpixel_bits": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_cnn_load": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, c_int32_p, C.c_int, C.c_int, C.c_int]),
    "mvlm_cnn_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "mvlm_cnn_maxima": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_int32_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_int]),
    "mvlm_cnn_heatmaps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_int32_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_int]),
    "mvlm_cnn_set_execution": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mvlm_cnn_set_selection": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_cnn_set_pairing": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_cnn_execution_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64)]),
    "mvlm_pack_fast_weights": (C.c_size_t, [c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16)]),
    "mvlm_cnn_load_fast": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint16), C.c_size_t, C.POINTER(C.c_int64), C.c_int]),
    "mvlm_cnn_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_pack_fast_weights16": (C.c_size_t, [c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16), c_float_p]),
    "mvlm_cnn_load_fast16": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint16), C.c_size_t, C.POINTER(C.c_int64), c_float_p, C.c_int]),
    "mvlm_cnn_fast16_overflowed": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "mvlm_conv2d_fast16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, c_float_p,
                                     c_float_p, c_float_p, c_float_p, c_float_p, C.c_void_p, C.c_void_p]),
    "mvlm_conv2d_fast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, c_float_p,
                                   c_float_p, c_float_p, c_float_p, c_float_p, C.c_void_p, C.c_void_p]),
    "mvlm_cnn_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_cnn_get_profile": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_double_p, c_float_p, C.c_int]),
    "mvlm_conv_variant_name": (C.c_char_p, [C.c_int]),
    "mvlm_cnn_get_profile_shapes": (C.c_int, [C.c_void_p, c_int32_p, C.c_int]),
    "mvlm_conv_variant_serves": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvlm_conv_set_override": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mvlm_heatmap_maxima": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mvlm_conv2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_int, C.c_int,
                              c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mvlm_conv2d_pair": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, c_float_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int, c_float_p, C.c_void_p, C.c_void_p, C.c_void_p, c_float_p, c_float_p,
                                   C.c_int]),
    "mvlm_conv_force_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "mvlm_conv_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  c_float_p, C.POINTER(C.c_int)]),
    "mvlm_conv_pair_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p]),
    "mvlm_estimate_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p]),
    "mvlm_consensus_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_void_p, C.c_void_p]),
    "mvlm_consensus_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    "mvlm_project_to_surface": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mvlm_clip_rays_to_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p]),
}

_lib = None


class MvlmHipError(RuntimeError):
    pass


def build(force: bool = False) -> Path:
    """Compile libmvlm_hip.so for gfx950 with hipcc (mvlm_amd/csrc/Makefile)."""
    if force and LIB_PATH.exists():
        LIB_PATH.unlink()
    jobs = str(min(8, os.cpu_count() or 1))
    r = subprocess.run(["make", "-C", str(CSRC), "-j", jobs], capture_output=True, text=True)
    if r.returncode != 0 or not LIB_PATH.exists():
        raise MvlmHipError(f"building libmvlm_hip.so failed:\n{r.stdout[-4000:]}\n{r.stderr[-4000:]}")
    return LIB_PATH


def load():
    """dlopen the library and attach the prototypes; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MvlmHipError(
            f"{LIB_PATH} not found - the MI355X HIP library is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mvlm_amd/csrc`.")
    # torch ships its own ROCm runtime; import it first so this library binds to the same
    # libamdhip64 instance the tensors we are handed live in (two runtimes in one process
    # do not see each other's devices or allocations)
    import torch  # noqa: F401

    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_ptr(arr, ctype):
    """Host numpy array -> typed ctypes pointer (array must stay alive)."""
    return arr.ctypes.data_as(C.POINTER(ctype))


class Context:
    """One mvlm_ctx bound to a CUDA/HIP device index; owns nothing else."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.handle = C.c_void_p()
        rc = self.lib.mvlm_ctx_create(int(device), C.byref(self.handle))
        if rc != 0:
            reasons = {2: "no such GPU device", 3: "hipSetDevice failed", 4: "hipGetDeviceProperties failed",
                       5: "device is not a gfx950 (MI355X)"}
            raise MvlmHipError(f"mvlm_ctx_create(device={device}) failed: {reasons.get(rc, rc)}")
        self.device = int(device)

    def check(self, rc: int, exc=MvlmHipError):
        if rc != 0:
            msg = self.lib.mvlm_last_error(self.handle)
            raise exc(msg.decode() if msg else f"mvlm call failed ({rc})")

    def set_stream(self, stream_ptr: int):
        self.check(self.lib.mvlm_set_stream(self.handle, C.c_void_p(stream_ptr)))
        self._bound_stream = stream_ptr

    def bind_current_stream(self, torch, dev):
        """Make torch's current stream on ``dev`` this context's launch stream - unless THIS thread has bound it for a whole
        step already (``hold_stream``) and nobody has re-bound the context since: asking torch for the stream and telling the
        library costs ~8 us, and a step of the fused pipeline would do it nine times."""
        held = getattr(self, "_held_stream", None)
        if held is not None and held[0] == threading.get_ident() and getattr(self, "_bound_stream", None) == held[1]:
            return
        self.set_stream(torch.cuda.current_stream(dev).cuda_stream)

    def hold_stream(self, torch, dev):
        """Context manager: bind torch's current stream once and keep it bound until exit (nested holds are no-ops)."""
        return _HeldStream(self, torch, dev)

    def synchronize(self):
        self.check(self.lib.mvlm_synchronize(self.handle))

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.mvlm_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class _HeldStream:
    def __init__(self, ctx, torch, dev):
        self.ctx, self.torch, self.dev, self.mine = ctx, torch, dev, False

    def __enter__(self):
        if getattr(self.ctx, "_held_stream", None) is None:  # (held by another thread: this one binds call by call)
            ptr = self.torch.cuda.current_stream(self.dev).cuda_stream
            self.ctx.set_stream(ptr)
            self.ctx._held_stream = (threading.get_ident(), ptr)
            self.mine = True
        return self

    def __exit__(self, *exc):
        if self.mine:
            self.ctx._held_stream = None
        return False


_contexts: dict[int, Context] = {}


def get_context(device: int = 0) -> Context:
    """Process-wide context per device (pipelines on one GPU share scratch and weights cache)."""
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = Context(device)
    return ctx
