"""Landmark-heatmap predictor slot running on the MI355X.

Drop-in for the reference's ``PaulsenModel`` / ``BU3DFEPredictor`` /
``DTU3DPredictor`` (src/mvlm/prediction/paulsenpredictor.py:42-243): same
constructor arguments and ``predict_landmarks_from_images`` /
``get_lm_count`` contract.  The stacked-hourglass forward (:404-432), the
batching loop (:189-212) and the heatmap maxima (:112-165) run as HIP kernels
behind ``mvlm_cnn_maxima`` (mvlm_amd/csrc/cnn_graph.hip, conv_mfma.hip); the
[N,NL,256,256] heatmaps are never materialised, for the default "simple"
selection method and for "moment" alike.
"""
from __future__ import annotations

import abc
import ctypes as C
import os
from pathlib import Path

import numpy as np

from .. import _lib, arch, weights as W
from .predictor2d import Predictor2D

__all__ = ["HipPaulsenModel", "BU3DFEPredictor", "DTU3DPredictor", "checkpoint_basename"]

# file names of the reference's published state dicts (paulsenpredictor.py:15-26); the
# loader looks for them on disk instead of downloading
_CHECKPOINTS = {
    "MVLMModel_DTU3D-RGB": "MVLMModel_DTU3D_RGB_07092019_only_state_dict-c0255a70.pth",
    "MVLMModel_DTU3D-depth": "MVLMModel_DTU3D_Depth_19092019_only_state_dict-95b89b63.pth",
    "MVLMModel_DTU3D-geometry": "MVLMModel_DTU3D_geometry_only_state_dict-41851074.pth",
    "MVLMModel_DTU3D-geometry+depth": "MVLMModel_DTU3D_geometry+depth_20102019_15epoch_only_state_dict-73b20e31.pth",
    "MVLMModel_DTU3D-RGB+depth": "MVLMModel_DTU3D_RGB+depth_20092019_only_state_dict-e3c12463a9.pth",
    "MVLMModel_BU_3DFE-RGB": "MVLMModel_BU_3DFE_RGB_24092019_6epoch_only_state_dict-eb652074.pth",
    "MVLMModel_BU_3DFE-depth": "MVLMModel_BU_3DFE_depth_10102019_4epoch_only_state_dict-e2318093.pth",
    "MVLMModel_BU_3DFE-geometry": "MVLMModel_BU_3DFE_geometry_02102019_4epoch-only_state_dict-f85518fa.pth",
    "MVLMModel_BU_3DFE-RGB+depth": "MVLMModel_BU_3DFE_RGB+depth_05102019_5epoch_only_state_dict-297955f6.pth",
    "MVLMModel_BU_3DFE-geometry+depth": "MVLMModel_BU_3DFE_geometry+depth_17102019_13epoch_only_state_dict-aa34a6d68.pth",
}
_URL_ROOT = "https://shapeml.compute.dtu.dk/Deep-MVLM/models/"


def checkpoint_basename(model_type: str, image_mode: str) -> str:
    return _CHECKPOINTS[f"{model_type}-{image_mode}"]


class HipPaulsenModel(Predictor2D):
    """weights: None -> look for the reference checkpoint file in ``model_dir`` /
    ``$MVLM_MODEL_DIR`` / this package's ``models`` folder, then try the reference's
    URL; a path -> that checkpoint; a dict -> a state dict; ``"synthetic"`` or
    ``"synthetic:<seed>"`` -> seeded random weights (benchmarks, tests)."""

    def __init__(self, model_type: str, image_mode: str, n_gpus=1, batch_size=2, selection_method="simple",
                 weights=None, device: int = 0, device_batch: int | None = None, model_dir=None, verbose: bool = True,
                 precision: str = "exact"):
        super().__init__()
        if image_mode not in arch.IMAGE_CHANNELS:
            raise ValueError("Image channels should be: geometry, RGB, depth, RGB+depth or geometry+depth")
        if selection_method not in ("simple", "moment"):
            raise ValueError(f"unknown selection_method {selection_method!r}")
        self.batch_size = batch_size
        self.selection_method = selection_method
        self.model_type = model_type
        self.image_mode = image_mode
        self.n_gpus = n_gpus
        self.verbose = verbose
        self.in_channels = arch.IMAGE_CHANNELS[image_mode]
        self.chan_sel = np.asarray(arch.CHANNEL_SELECT[image_mode], dtype=np.int32)
        # views pushed through the network together (default: all, up to 128; ~113 MB of HBM
        # scratch per view); the reference's batch_size=2 is a CPU memory knob and does not
        # change any per-view result
        self.device_batch = device_batch
        self.ctx = _lib.Context(device)  # own context: it holds this model's weights
        self._workspace = None
        state_dict = self._resolve_weights(weights, model_dir)
        blob, desc = W.pack_for_device(state_dict, self.get_lm_count(), self.in_channels)
        self.ctx.check(self.ctx.lib.mvlm_cnn_load(
            self.ctx.handle, _lib.as_ptr(blob, C.c_float), blob.size, _lib.as_ptr(desc, C.c_int32), desc.shape[0],
            self.get_lm_count(), self.in_channels))
        self._state_dict, self._desc = state_dict, desc
        # n_gpus > 1: the reference's single-process form (paulsenpredictor.py:100-105, nn.DataParallel over
        # device_ids): one replica of the weights per further device, the views of a call split contiguously over
        # the devices, the maxima gathered on this model's own device
        self._replicas: list[_Replica] = []
        for d in self._replica_devices(n_gpus, device):
            rctx = _lib.Context(d)
            rctx.check(rctx.lib.mvlm_cnn_load(
                rctx.handle, _lib.as_ptr(blob, C.c_float), blob.size, _lib.as_ptr(desc, C.c_int32), desc.shape[0],
                self.get_lm_count(), self.in_channels))
            self._replicas.append(_Replica(rctx))
        self._fast_loaded = False
        self._fast16_loaded = False
        self.fast16_fallbacks = 0  # passes repeated with "fast" because an activation left fp16's range
        self._fast16_seen = False  # an overflow met in an earlier library call of the current pass (see _maxima_on)
        self.precision = "exact"
        self.set_precision(precision)

    @staticmethod
    def _replica_devices(n_gpus, device: int) -> list[int]:
        """Device indices of the replicas beside ``device``.  Like ``_prepare_device`` (paulsenpredictor.py:72-87) a
        request for more GPUs than the machine has is clamped - with a message instead of silently.
        MVLM_REPLICA_DEVICES="0,0" names the devices explicitly (rehearsal of the N-device path on one GPU)."""
        import torch

        n_gpus = int(n_gpus or 1)
        if n_gpus <= 1:
            return []
        forced = os.environ.get("MVLM_REPLICA_DEVICES")
        if forced:
            ids = [int(v) for v in forced.split(",") if v.strip() != ""]
            return ids[1:n_gpus]
        have = torch.cuda.device_count()
        if n_gpus > have:
            print(f"Warning: n_gpus={n_gpus} requested, {have} GPU(s) visible - using {have}")
            n_gpus = have
        return [d for d in range(have) if d != device][: max(0, n_gpus - 1)]

    def set_precision(self, precision: str):
        """"exact" (default): every convolution in exact fp32 on the matrix cores - the path all parity claims are
        about.  Opt-in, fp32-accurate but not bit-identical (argmax near-ties may flip; bench.py reports both separately):
        "fast" - the big 3x3 layers multiply bf16x3-split operands (6 cross products, mvlm_amd/csrc/conv_fast.hip);
        "fast16" - f16x2-split operands (3 cross products: half the matrix work again).  fp16 ends at 65504: a pass that meets
        a larger activation is flagged by the kernel (its maxima carry NaN scores), and ``predict_landmarks_from_images`` / the
        pipeline repeat it with "fast"."""
        if precision not in ("exact", "fast", "fast16"):
            raise ValueError("precision must be 'exact', 'fast' or 'fast16'")
        holders = [self.ctx] + [r.ctx for r in self._replicas]
        if precision == "fast" and not self._fast_loaded:
            blob16, offsets = W.pack_fast_for_device(self._state_dict, self.get_lm_count(), self.in_channels, self._desc)
            for ctx in holders:
                ctx.check(ctx.lib.mvlm_cnn_load_fast(
                    ctx.handle, blob16.ctypes.data_as(C.POINTER(C.c_uint16)), blob16.size,
                    offsets.ctypes.data_as(C.POINTER(C.c_int64)), offsets.shape[0]))
            self._fast_loaded = True
        if precision == "fast16" and not self._fast16_loaded:
            blob16, offsets, unscale = W.pack_fast16_for_device(self._state_dict, self.get_lm_count(), self.in_channels, self._desc)
            for ctx in holders:
                ctx.check(ctx.lib.mvlm_cnn_load_fast16(
                    ctx.handle, blob16.ctypes.data_as(C.POINTER(C.c_uint16)), blob16.size,
                    offsets.ctypes.data_as(C.POINTER(C.c_int64)), _lib.as_ptr(unscale, C.c_float), offsets.shape[0]))
            self._fast16_loaded = True
        for ctx in holders:
            ctx.check(ctx.lib.mvlm_cnn_set_precision(ctx.handle, {"exact": 0, "fast": 1, "fast16": 2}[precision]))
        self.precision = precision
        # what the CALLER asked for (repeat_without_fp16 restores it): the same on every rank of a sharded job, unlike
        # ``precision``, which a rank's own fallback changes - collectives are decided from this one
        self.configured_precision = precision

    @abc.abstractmethod
    def get_lm_count(self) -> int:
        pass

    # ---- weights ----------------------------------------------------------------------
    def _resolve_weights(self, weights, model_dir):
        nl, c = self.get_lm_count(), self.in_channels
        if isinstance(weights, dict):
            return {k: np.asarray(v) for k, v in weights.items()}
        if isinstance(weights, str) and weights.startswith("synthetic"):
            seed = int(weights.split(":", 1)[1]) if ":" in weights else 0
            return W.synthetic_state_dict(nl, c, seed=seed)
        if weights is not None:
            return W.load_state_dict_file(weights)
        name = checkpoint_basename(self.model_type, self.image_mode)
        candidates = [Path(d) / name for d in (model_dir, os.environ.get("MVLM_MODEL_DIR"),
                                                Path(__file__).parent / "models") if d]
        for p in candidates:
            if p.is_file():
                return W.load_state_dict_file(p)
        # same fallback as the reference (:94-101): fetch into the package's models folder.  The file
        # names carry the sha256 prefix torch.hub verifies (check_hash); the downloaded file then goes
        # through the same safe loader (weights_only, "module." prefix handling) as a local checkpoint
        from torch.hub import download_url_to_file

        if self.verbose:
            print("Loading checkpoint")
        target = Path(__file__).parent / "models" / name
        target.parent.mkdir(parents=True, exist_ok=True)
        hash_prefix = name.rsplit("-", 1)[1].split(".")[0]
        download_url_to_file(_URL_ROOT + name, str(target), hash_prefix=hash_prefix, progress=self.verbose)
        return W.load_state_dict_file(target)

    # ---- inference --------------------------------------------------------------------
    def _batch_for(self, n_views: int) -> int:
        return max(1, min(n_views, self.device_batch or 128))

    def _get_workspace(self, batch: int, holder=None):
        import torch

        holder = holder or self
        ctx = holder.ctx
        need = int(ctx.lib.mvlm_cnn_workspace_bytes(ctx.handle, batch))
        if need == 0:
            raise _lib.MvlmHipError("mvlm_cnn_workspace_bytes returned 0 (weights not loaded?)")
        if holder._workspace is None or holder._workspace.numel() < need:
            holder._workspace = None
            holder._workspace = torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", ctx.device))
        return holder._workspace

    def _maxima_on(self, holder, x, maxima):
        """The network + maxima of the views ``x`` (on ``holder``'s device) into ``maxima`` [NL,n,3] there."""
        import torch

        ctx = holder.ctx
        dev = torch.device("cuda", ctx.device)
        n, nl = int(x.shape[0]), self.get_lm_count()
        batch = self._batch_for(n)
        ws = self._get_workspace(batch, holder)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        # "moment" (:129-156) runs fused too: the 31x31 window around each peak is recomputed from conv10's output with
        # conv11's own arithmetic (mvlm_cnn_set_selection) - no [N,NL,256,256] tensor.  MVLM_MOMENT_MATERIALISED=1 (tests) or
        # packed weights without conv11's parity form take the heatmaps through HBM instead; the results are the same.
        moment = self.selection_method == "moment"
        fused = moment and os.environ.get("MVLM_MOMENT_MATERIALISED") != "1"
        if fused and ctx.lib.mvlm_cnn_set_selection(ctx.handle, 1) != 0:
            fused = False  # conv11 without its parity slots
        if not fused:
            ctx.check(ctx.lib.mvlm_cnn_set_selection(ctx.handle, 0))
        if fused or not moment:
            ctx.check(ctx.lib.mvlm_cnn_maxima(
                ctx.handle, C.c_void_p(x.data_ptr()), n, _lib.as_ptr(self.chan_sel, C.c_int32),
                C.c_void_p(maxima.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), batch))
            return
        for s in range(0, n, batch):
            nb = min(batch, n - s)
            heat = torch.empty((nb, nl, 256, 256), dtype=torch.float32, device=dev)
            part = torch.empty((nl, nb, 3), dtype=torch.float32, device=dev)
            ctx.check(ctx.lib.mvlm_cnn_heatmaps(
                ctx.handle, C.c_void_p(x[s:s + nb].data_ptr()), nb, _lib.as_ptr(self.chan_sel, C.c_int32),
                C.c_void_p(heat.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), batch))
            ctx.check(ctx.lib.mvlm_heatmap_maxima(ctx.handle, C.c_void_p(heat.data_ptr()), nb, nl, 256,
                                                  1, C.c_void_p(part.data_ptr())))
            maxima[:, s:s + nb] = part
            if self.precision == "fast16":
                # every mvlm_cnn_heatmaps call starts with a clear range flag: an overflow in THIS slice has to be noted
                # before the next slice's call erases it
                v = C.c_int(0)
                ctx.check(ctx.lib.mvlm_cnn_fast16_overflowed(ctx.handle, C.byref(v)))
                self._fast16_seen = self._fast16_seen or bool(v.value)

    def predict_device(self, image_stack_dev, out=None):
        """torch f32 [N,256,256,4] on this GPU -> maxima torch f32 [NL,N,3] on the GPU (written into ``out``
        when given).  A pass over the same buffers is captured as a hipGraph on its second use and replayed
        afterwards (mvlm_cnn_set_execution), so callers in a loop should hand over stable buffers.
        With ``n_gpus`` > 1 the views are split contiguously over the replicas' devices (the scatter / gather of
        nn.DataParallel, paulsenpredictor.py:104-105): slices travel device to device, every device's pass is
        enqueued before any result is fetched, and the maxima come back into ``out`` in view order."""
        import torch

        dev = torch.device("cuda", self.ctx.device)
        if image_stack_dev.dtype != torch.float32 or tuple(image_stack_dev.shape[1:]) != (256, 256, 4):
            raise RuntimeError(f"Unexpected image stack shape: {tuple(image_stack_dev.shape)} {image_stack_dev.dtype}")
        self._fast16_seen = False  # range flag of THIS pass (slices of a materialised "moment" pass add to it)
        x = image_stack_dev.contiguous()
        n = int(x.shape[0])
        nl = self.get_lm_count()
        if out is None:
            maxima = torch.empty((nl, n, 3), dtype=torch.float32, device=dev)
        else:
            maxima = out
            if tuple(out.shape) != (nl, n, 3) or out.dtype != torch.float32 or not out.is_contiguous():
                raise ValueError("predict_device: out must be a contiguous float32 [NL,N,3] tensor")
        if not self._replicas or n < 2:
            self._maxima_on(self, x, maxima)
            return maxima
        from ..parallel import shard_range

        holders = [self] + self._replicas
        world = len(holders)
        parts = []
        # the scatter copies run on THIS device's stream (torch issues a device-to-device copy on the source's
        # stream), so every replica gets its slice and its pass before this device's own pass is enqueued
        for r in list(range(1, world)) + [0]:
            h = holders[r]
            lo, hi = shard_range(n, r, world)
            if hi == lo:
                continue
            rdev = torch.device("cuda", h.ctx.device)
            if r == 0:
                xin = x[lo:hi]
            else:
                xin = self._slice_buffer(h, "images", (hi - lo, 256, 256, 4))
                xin.copy_(x[lo:hi], non_blocking=True)   # device to device (xGMI), ordered by torch's stream events
            with torch.cuda.device(rdev):
                # [NL,N,3] is landmark-major: every device computes its [NL,n_r,3] slice into a buffer of its own
                rmax = self._slice_buffer(h, "maxima", (nl, hi - lo, 3))
                self._maxima_on(h, xin, rmax)
            parts.append((lo, hi, rmax))
        for lo, hi, part in parts:                        # gather: [NL, n_r, 3] slices into view order
            maxima[:, lo:hi].copy_(part, non_blocking=True)
        return maxima

    @staticmethod
    def _slice_buffer(holder, name: str, shape: tuple):
        """A float32 tensor on ``holder``'s device kept from call to call (stable addresses: graph replay)."""
        import torch

        bufs = holder.__dict__.setdefault("_slice_buffers", {})
        buf = bufs.get(name)
        if buf is None or tuple(buf.shape) != tuple(shape):
            buf = bufs[name] = torch.empty(shape, dtype=torch.float32, device=torch.device("cuda", holder.ctx.device))
        return buf

    def set_execution(self, graphs: bool = True, concurrency: bool = False, pairing: int | None = None):
        """How the forward pass is issued (mvlm_cnn_set_execution): replayed hipGraphs / launch by launch, and
        (experiment only, measured slower) whether small batches run the lower hourglass pyramid on a second stream.
        ``pairing`` (mvlm_cnn_set_pairing; None = leave as it is): 0 one launch per convolution, 1 (the default) independent
        residual blocks of a hourglass level share launches where the measured table says that is faster, 2 wherever one
        kernel variant can serve both.  graphs / concurrency never change a result; pairing may change the kernel variant
        a convolution runs on, i.e. the last bits of its fp32 sums."""
        for ctx in [self.ctx] + [r.ctx for r in self._replicas]:
            ctx.check(ctx.lib.mvlm_cnn_set_execution(ctx.handle, int(bool(graphs)), int(bool(concurrency))))
            if pairing is not None:
                ctx.check(ctx.lib.mvlm_cnn_set_pairing(ctx.handle, int(pairing)))

    def execution_stats(self) -> dict:
        v = [C.c_int64() for _ in range(4)]
        self.ctx.check(self.ctx.lib.mvlm_cnn_execution_stats(self.ctx.handle, *[C.byref(x) for x in v]))
        return dict(zip(("eager_runs", "graph_captures", "graph_replays", "graph_failures"), (int(x.value) for x in v)))

    def heatmaps_device(self, image_stack_dev):
        """Final-stage heatmaps torch f32 [N,NL,256,256] (tests / diagnostics)."""
        import torch

        dev = torch.device("cuda", self.ctx.device)
        x = image_stack_dev.contiguous()
        n, nl = int(x.shape[0]), self.get_lm_count()
        batch = self._batch_for(n)
        ws = self._get_workspace(batch)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        heat = torch.empty((n, nl, 256, 256), dtype=torch.float32, device=dev)
        self.ctx.check(self.ctx.lib.mvlm_cnn_heatmaps(
            self.ctx.handle, C.c_void_p(x.data_ptr()), n, _lib.as_ptr(self.chan_sel, C.c_int32),
            C.c_void_p(heat.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), batch))
        return heat

    def predict_landmarks_from_images(self, image_stack: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        import torch

        n_views = image_stack.shape[0]
        valid = np.ones((n_views), dtype=bool)
        x = torch.from_numpy(np.ascontiguousarray(image_stack, dtype=np.float32)).to(torch.device("cuda", self.ctx.device))
        lms = self.predict_device(x).cpu().numpy()
        if self.precision == "fast16" and self.fast16_overflowed():
            lms = self.repeat_without_fp16(lambda: self.predict_device(x).cpu().numpy())
        return lms, valid

    def fast16_overflowed(self) -> bool:
        """Did the last "fast16" pass meet an activation outside fp16's range (on any replica)?  Waits for the passes; ask
        after the results have been fetched."""
        hit = self._fast16_seen
        for ctx in [self.ctx] + [r.ctx for r in self._replicas]:
            v = C.c_int(0)
            ctx.check(ctx.lib.mvlm_cnn_fast16_overflowed(ctx.handle, C.byref(v)))
            hit = hit or bool(v.value)
        return hit

    def repeat_without_fp16(self, run):
        """An activation left fp16's range in a "fast16" pass (non-finite maxima): repeat ``run`` with the bf16x3 form, whose
        exponent range is fp32's, and stay there for this predictor's remaining calls (the scan that overflows once will
        again)."""
        print('Warning: an activation exceeded the fp16 range of precision="fast16" - repeating the pass with precision="fast"')
        self.fast16_fallbacks += 1
        asked = self.configured_precision
        self.set_precision("fast")
        self.configured_precision = asked
        return run()


class _Replica:
    """Weights of the model on one further device (``n_gpus`` > 1): its context, workspace and slice buffers."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._workspace = None


class BU3DFEPredictor(HipPaulsenModel):
    def __init__(self, batch_size=2, selection_method="simple", n_gpus=1, image_mode="RGB+depth", **kw):
        super().__init__(model_type="MVLMModel_BU_3DFE", image_mode=image_mode, n_gpus=n_gpus, batch_size=batch_size,
                         selection_method=selection_method, **kw)

    def get_lm_count(self) -> int:
        return 84


class DTU3DPredictor(HipPaulsenModel):
    def __init__(self, batch_size=2, selection_method="simple", n_gpus=1, image_mode="RGB+depth", **kw):
        super().__init__(model_type="MVLMModel_DTU3D", image_mode=image_mode, n_gpus=n_gpus, batch_size=batch_size,
                         selection_method=selection_method, **kw)

    def get_lm_count(self) -> int:
        return 73
