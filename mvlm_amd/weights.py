"""Weights for the landmark network: synthetic generator, checkpoint loader, packer.

* ``synthetic_state_dict`` - framework-independent seeded weights over the 817
  reference keys (no pretrained checkpoints are reachable offline; the reference
  downloads them, paulsenpredictor.py:94-101).
* ``load_state_dict_file`` - local-path replacement for the reference's URL
  loader; accepts both full checkpoints (``["state_dict"]``) and state-dict-only
  files (paulsenpredictor.py:101-102).
* ``pack_for_device`` - folds every BatchNorm (eval mode, running stats) into a
  per-channel scale/shift and lays the conv weights out as
  ``[tap][cin_pad][cout_pad]`` f32, the order the HIP implicit-GEMM kernel
  stages them into LDS (mvlm_amd/csrc/conv_mfma.hip).
"""
from __future__ import annotations

import numpy as np

from . import arch

CIN_ALIGN = 4    # smallest K-chunk of the conv kernel (channels per LDS stage)
COUT_ALIGN = 32  # one MFMA 32x32 tile of output channels
COUT_EXACT_84 = 84  # conv6 / conv10 of the 84-landmark network run unpadded on the 64 + 16 + 4-row tile
COUT_TAIL_PADS = (80,)  # plain conv+bias layers may run as 64 rows + one 16-row MFMA strip (73 landmarks -> 80, not 96)
DESC_INTS = 12   # int32 fields per conv slot in the descriptor table


def synthetic_state_dict(n_landmarks: int, in_channels: int, seed: int = 0, gain: float = 0.7) -> dict[str, np.ndarray]:
    """Seeded random weights keyed by the reference's state-dict names.

    Values are drawn from ``np.random.RandomState(seed)`` walking the keys in
    sorted order, so any consumer (the oracle, the packer, the golden generator)
    reproduces them without torch.
    """
    rs = np.random.RandomState(seed)
    shapes = arch.state_dict_shapes(n_landmarks, in_channels)
    sd: dict[str, np.ndarray] = {}
    for key in sorted(shapes):
        shp = shapes[key]
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[key] = np.array(1, dtype=np.int64)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            sd[key] = (rs.standard_normal(shp) * (gain * np.sqrt(2.0 / fan_in))).astype(np.float32)
        elif leaf == "running_mean":
            sd[key] = (rs.standard_normal(shp) * 0.1).astype(np.float32)
        elif leaf == "running_var":
            sd[key] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif leaf == "weight":  # BN gamma
            sd[key] = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif leaf == "bias":
            sd[key] = (rs.standard_normal(shp) * 0.1).astype(np.float32)
        else:  # pragma: no cover
            raise KeyError(key)
    return sd


def load_state_dict_file(path) -> dict[str, np.ndarray]:
    """Read a reference checkpoint from a local file into numpy arrays."""
    import torch

    ckpt = torch.load(str(path), map_location="cpu", weights_only=True)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):  # saved from nn.DataParallel (paulsenpredictor.py:104-105)
            k = k[len("module."):]
        out[k] = v.detach().cpu().numpy()
    return out


def check_state_dict(sd: dict[str, np.ndarray], n_landmarks: int, in_channels: int) -> None:
    shapes = arch.state_dict_shapes(n_landmarks, in_channels)
    missing = [k for k in shapes if k not in sd]
    if missing:
        raise KeyError(f"state dict is missing {len(missing)} keys, first: {missing[:3]}")
    for k, shp in shapes.items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"state dict key {k}: shape {tuple(sd[k].shape)} != expected {shp}")


def fold_bn(sd: dict[str, np.ndarray], prefix: str) -> tuple[np.ndarray, np.ndarray]:
    """Eval-mode BatchNorm as y = x*scale + shift, all in float32."""
    gamma = sd[f"{prefix}.weight"].astype(np.float32)
    beta = sd[f"{prefix}.bias"].astype(np.float32)
    mean = sd[f"{prefix}.running_mean"].astype(np.float32)
    var = sd[f"{prefix}.running_var"].astype(np.float32)
    invstd = (np.float32(1.0) / np.sqrt(var + np.float32(arch.BN_EPS))).astype(np.float32)
    scale = (gamma * invstd).astype(np.float32)
    shift = (beta - mean * scale).astype(np.float32)
    return scale, shift


def collapse_upsampled_3x3(w: np.ndarray, a: int, b: int) -> np.ndarray:
    """3x3 kernel applied after nearest 2x upsampling == 2x2 kernel on the low-res tensor.

    Output pixel (2i+a, 2j+b) reads upsampled rows 2i+a-1..2i+a+1, i.e. low-res rows
    {i-1, i, i} for a = 0 and {i, i, i+1} for a = 1 (same for columns), so the taps that land on
    the same low-res pixel are summed (in float64, rounded once).  w [cout,cin,3,3] ->
    [cout,cin,2,2] over low-res rows (i-1+a, i+a), columns (j-1+b, j+b).
    """
    groups = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    w64 = w.astype(np.float64)
    out = np.zeros(w.shape[:2] + (2, 2), np.float64)
    for ty, rows in enumerate(groups[a]):
        for tx, cols in enumerate(groups[b]):
            for r in rows:
                for c in cols:
                    out[:, :, ty, tx] += w64[:, :, r, c]
    return out.astype(np.float32)


def _round_up(x: int, a: int) -> int:
    return (x + a - 1) // a * a


def pack_for_device(sd: dict[str, np.ndarray], n_landmarks: int, in_channels: int) -> tuple[np.ndarray, np.ndarray]:
    """Return (blob f32[total], desc int32[N_CONV_SLOTS, DESC_INTS]).

    desc row: present, cin, cout, ksize, cin_pad, cout_pad, w_off, bias_off,
    pre_scale_off, pre_shift_off, post_scale_off, post_shift_off (offsets in
    floats into blob, -1 = absent).  Padding channels carry zero weights, zero
    bias, zero scale and zero shift so they contribute exactly 0.
    """
    check_state_dict(sd, n_landmarks, in_channels)
    slots = arch.conv_slots(n_landmarks, in_channels)
    desc = np.full((len(slots), DESC_INTS), -1, dtype=np.int32)
    parts: list[np.ndarray] = []
    cursor = 0

    def push(a: np.ndarray) -> int:
        nonlocal cursor
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        # keep every sub-array 16-byte aligned for float4 loads
        pad = (-a.size) % 4
        if pad:
            a = np.concatenate([a, np.zeros(pad, np.float32)])
        off = cursor
        parts.append(a)
        cursor += a.size
        return off

    for s in slots:
        row = desc[s.index]
        row[0] = int(s.present)
        row[1:4] = (s.cin, s.cout, s.ksize)
        cin_pad = _round_up(s.cin, CIN_ALIGN)
        cout_pad = _round_up(s.cout, COUT_ALIGN)
        plain = s.has_bias and s.pre_bn is None and s.post_bn is None  # conv6, conv7, conv10, conv11(+parity)
        if plain and _round_up(s.cout, 16) in COUT_TAIL_PADS:
            cout_pad = _round_up(s.cout, 16)
        elif plain and s.cout == COUT_EXACT_84 and (s.name in ("conv6", "conv10") or s.ksize == 2):
            # 64 + 16 + 4 rows: conv6 / conv10 (conv3x3_c84_t8x32) and, since round 3, conv11's four parity convolutions
            # (conv2x2_c84_t8x32, fused argmax on all three row groups); the un-collapsed 3x3 conv11 slot keeps 96
            cout_pad = s.cout
        row[4:6] = (cin_pad, cout_pad)
        if not s.present:
            continue
        if s.ksize == 2:  # "conv11.parityAB"
            base, par = s.name.rsplit(".parity", 1)
            w = collapse_upsampled_3x3(sd[f"{base}.weight"].astype(np.float32), int(par[0]), int(par[1]))
            bias_key = f"{base}.bias"
        else:
            w = sd[f"{s.name}.weight"].astype(np.float32)  # [cout, cin, k, k]
            bias_key = f"{s.name}.bias"
        k = s.ksize
        wp = np.zeros((k * k, cin_pad, cout_pad), np.float32)
        wp[:, : s.cin, : s.cout] = w.transpose(2, 3, 1, 0).reshape(k * k, s.cin, s.cout)
        row[6] = push(wp)
        if s.has_bias:
            b = np.zeros(cout_pad, np.float32)
            b[: s.cout] = sd[bias_key]
            row[7] = push(b)
        if s.pre_bn is not None:
            sc, sh = fold_bn(sd, s.pre_bn)
            a = np.zeros(cin_pad, np.float32)
            b = np.zeros(cin_pad, np.float32)
            a[: s.cin], b[: s.cin] = sc, sh
            row[8], row[9] = push(a), push(b)
        if s.post_bn is not None:
            sc, sh = fold_bn(sd, s.post_bn)
            a = np.zeros(cout_pad, np.float32)
            b = np.zeros(cout_pad, np.float32)
            a[: s.cout], b[: s.cout] = sc, sh
            row[10], row[11] = push(a), push(b)
    blob = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    return blob, desc


def pack_fast_for_device(sd: dict[str, np.ndarray], n_landmarks: int, in_channels: int, desc: np.ndarray):
    """Weights of the layers the opt-in "fast" precision serves (mvlm_amd/csrc/conv_fast.hip): every present 3x3 conv
    with 16..256 input channels whose output channels fill at least 5/8 of their 64-channel tiles, split into three
    bf16 terms and laid out by the library's own packer (mvlm_pack_fast_weights).  ``desc`` is the descriptor table of
    ``pack_for_device`` (it fixes the paddings).  Returns (blob uint16[total], offsets int64[N_CONV_SLOTS], -1 = the
    layer stays on the exact kernel)."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    slots = arch.conv_slots(n_landmarks, in_channels)
    offsets = np.full(len(slots), -1, dtype=np.int64)
    parts, cursor = [], 0
    for s in slots:
        # the kernel's own paddings and limits (mvlm_fast_cin_pad / mvlm_fast_cout_pad / mvlm_fast_channels_ok, common.h)
        cin_pad, cout_pad = _round_up(s.cin, 16), _round_up(s.cout, 64)
        if not s.present or s.ksize != 3 or not 16 <= s.cin <= 256 or s.cout * 8 < cout_pad * 5:
            continue  # (the 32-channel layers: the f16x2 form only, pack_fast16_for_device)
        w = np.ascontiguousarray(sd[f"{s.name}.weight"], dtype=np.float32)
        n = int(lib.mvlm_pack_fast_weights(_lib.as_ptr(w, C.c_float), s.cout, s.cin, cout_pad, cin_pad, None))
        if n == 0:
            raise ValueError(f"mvlm_pack_fast_weights refused layer {s.name}")
        buf = np.empty(n, dtype=np.uint16)
        if int(lib.mvlm_pack_fast_weights(_lib.as_ptr(w, C.c_float), s.cout, s.cin, cout_pad, cin_pad,
                                          buf.ctypes.data_as(C.POINTER(C.c_uint16)))) != n:
            raise ValueError(f"mvlm_pack_fast_weights failed on layer {s.name}")
        offsets[s.index] = cursor
        parts.append(buf)
        cursor += n
    blob = np.concatenate(parts) if parts else np.zeros(0, np.uint16)
    return blob, offsets


def pack_fast16_for_device(sd: dict[str, np.ndarray], n_landmarks: int, in_channels: int, desc: np.ndarray):
    """The same layers for the second opt-in form, "fast16" (f16x2-split operands, conv_fast.hip): every weight scaled by
    the layer's power of two and split into two fp16 terms by the library's packer (mvlm_pack_fast_weights16).
    Returns (blob uint16[total], offsets int64[N_CONV_SLOTS] (-1: the layer stays exact), unscale float32[N_CONV_SLOTS])."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    slots = arch.conv_slots(n_landmarks, in_channels)
    offsets = np.full(len(slots), -1, dtype=np.int64)
    unscale = np.ones(len(slots), dtype=np.float32)
    parts, cursor = [], 0
    for s in slots:
        # (mvlm_fast_cout_pad: this form has a 32-channel tile, so layers with up to 32 output channels pad to 32)
        cin_pad, cout_pad = _round_up(s.cin, 16), (32 if s.cout <= 32 else _round_up(s.cout, 64))
        if not s.present or s.ksize != 3 or not 16 <= s.cin <= 256 or s.cout * 8 < cout_pad * 5:
            continue
        w = np.ascontiguousarray(sd[f"{s.name}.weight"], dtype=np.float32)
        n = int(lib.mvlm_pack_fast_weights16(_lib.as_ptr(w, C.c_float), s.cout, s.cin, cout_pad, cin_pad, None, None))
        if n == 0:
            raise ValueError(f"mvlm_pack_fast_weights16 refused layer {s.name}")
        buf = np.empty(n, dtype=np.uint16)
        inv = C.c_float(1.0)
        if int(lib.mvlm_pack_fast_weights16(_lib.as_ptr(w, C.c_float), s.cout, s.cin, cout_pad, cin_pad,
                                            buf.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(inv))) != n:
            raise ValueError(f"mvlm_pack_fast_weights16 failed on layer {s.name} (non-finite weights?)")
        offsets[s.index] = cursor
        unscale[s.index] = inv.value
        parts.append(buf)
        cursor += n
    blob = np.concatenate(parts) if parts else np.zeros(0, np.uint16)
    return blob, offsets, unscale
