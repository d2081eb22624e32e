"""Oracle: landmark-heatmap network and heatmap maxima on the CPU (torch fp32).

Restates src/mvlm/prediction/paulsenpredictor.py:
  MVLMModel.forward :404-432, ResidualBlock.forward :267-273,
  HourGlassModule.forward :301-361, find_heat_map_maxima :112-158,
  predict_landmarks_from_images :167-217.
The network is evaluated functionally over a plain ``{key: ndarray}`` state dict
(the reference's 817 keys), eval mode: dropout = identity, BatchNorm uses
running statistics.  TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def _bn(sd, prefix, x):
    return F.batch_norm(x, _t(sd, f"{prefix}.running_mean"), _t(sd, f"{prefix}.running_var"),
                        _t(sd, f"{prefix}.weight"), _t(sd, f"{prefix}.bias"), False, 0.1, BN_EPS)


def _conv(sd, prefix, x, pad):
    b = sd.get(f"{prefix}.bias")
    return F.conv2d(x, _t(sd, f"{prefix}.weight"), None if b is None else _t(sd, f"{prefix}.bias"), 1, pad)


def residual_block(sd, p, x):
    """paulsenpredictor.py:267-273 (pre-activation, concat of 3 conv outputs + residual)."""
    out1 = _conv(sd, f"{p}.conv1", F.relu(_bn(sd, f"{p}.bn1", x)), 1)
    out2 = _conv(sd, f"{p}.conv2", F.relu(_bn(sd, f"{p}.bn2", out1)), 1)
    out3 = _conv(sd, f"{p}.conv3", F.relu(_bn(sd, f"{p}.bn3", out2)), 1)
    residual = x
    if f"{p}.resample.2.weight" in sd:
        residual = _conv(sd, f"{p}.resample.2", F.relu(_bn(sd, f"{p}.resample.0", x)), 0)
    return torch.cat((out1, out2, out3), 1) + residual


def hourglass(sd, p, x):
    """paulsenpredictor.py:301-361."""
    rb = lambda i, t: residual_block(sd, f"{p}.rb{i}", t)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    up1 = rb(1, x)
    low1 = rb(2, F.max_pool2d(x, 2))
    up11 = rb(3, low1)
    low11 = rb(4, F.max_pool2d(low1, 2))
    up12 = rb(5, low11)
    low12 = rb(6, F.max_pool2d(low11, 2))
    up13 = rb(7, low12)
    low13 = rb(8, F.max_pool2d(low12, 2))
    up14 = rb(9, low13)
    low14 = rb(10, F.max_pool2d(low13, 2))
    low3 = rb(12, rb(11, low14))
    add1 = up(low3) + up14
    add2 = up(rb(14, rb(13, add1))) + up13
    add3 = up(rb(16, rb(15, add2))) + up12
    add4 = up(rb(18, rb(17, add3))) + up11
    add5 = up(rb(20, rb(19, add4))) + up1
    return add5


def forward_last_stage(sd, x: torch.Tensor) -> torch.Tensor:
    """[B,C,256,256] -> final-stage heatmaps [B,NL,256,256] (``up_out2``).

    paulsenpredictor.py:404-432; only ``outputs[-1]`` is consumed downstream
    (:204-205) so the dead ``conv8`` branch is not evaluated.
    """
    with torch.no_grad():
        x = F.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, 1)))
        x = residual_block(sd, "conv2", x)
        x = F.max_pool2d(x, 2)
        x = residual_block(sd, "conv3", x)
        r3 = residual_block(sd, "conv4", x)
        x = hourglass(sd, "hg1", r3)
        ll1 = F.relu(_bn(sd, "bn2", _conv(sd, "conv5", x, 1)))
        x = _conv(sd, "conv6", ll1, 1)
        x = _conv(sd, "conv7", x, 1)
        sum_temp = r3 + ll1 + x
        x = hourglass(sd, "hg2", sum_temp)
        x = F.relu(_bn(sd, "bn3", _conv(sd, "conv9", x, 1)))
        x = _conv(sd, "conv10", x, 1)
        up_temp2 = F.interpolate(x, scale_factor=2, mode="nearest")
        return _conv(sd, "conv11", up_temp2, 1)


def find_heat_map_maxima(heatmaps: np.ndarray, selection_method: str = "simple") -> np.ndarray:
    """[NL,S,S] -> [NL,3] = (row-1, col-0.5, value); paulsenpredictor.py:112-158."""
    nl, s = heatmaps.shape[0], heatmaps.shape[1]
    out = np.zeros((nl, 3), dtype=np.float32)
    for k in range(nl):
        hm = heatmaps[k]
        flat = int(np.argmax(hm))  # first maximum in row-major order (:123)
        px, py = divmod(flat, s)
        value = hm[px, py]
        if selection_method == "moment":  # :129-156
            sz = 15
            if px > sz and s - px > sz and py > sz and s - py > sz:
                slc = hm[px - sz: px + sz + 1, py - sz: py + sz + 1]
                ar = np.arange(2 * sz + 1)
                sum_x = np.sum(slc, axis=1)
                px = px + (np.sum(np.multiply(ar, sum_x)) / np.sum(sum_x) - sz)
                sum_y = np.sum(slc, axis=0)
                py = py + (np.sum(np.multiply(ar, sum_y)) / np.sum(sum_y) - sz)
        out[k] = (px - 1, py - 0.5, value)
    return out


def maxima_from_heatmaps(heatmaps: np.ndarray, selection_method: str = "simple") -> np.ndarray:
    """[N,NL,S,S] -> [NL,N,3]; paulsenpredictor.py:160-165."""
    n, nl = heatmaps.shape[:2]
    out = np.empty((nl, n, 3), dtype=np.float32)
    for i in range(n):
        out[:, i, :] = find_heat_map_maxima(heatmaps[i], selection_method)
    return out


def maxima_fast(heatmaps: torch.Tensor) -> np.ndarray:
    """Vectorised equal of ``maxima_from_heatmaps(..., "simple")`` for big stacks
    (torch.argmax also returns the first maximal index)."""
    n, nl, s, _ = heatmaps.shape
    flat = heatmaps.reshape(n, nl, s * s)
    idx = torch.argmax(flat, dim=2)
    val = torch.gather(flat, 2, idx[..., None])[..., 0]
    out = np.empty((nl, n, 3), dtype=np.float32)
    out[:, :, 0] = ((idx // s).to(torch.float32) - 1).numpy().T
    out[:, :, 1] = ((idx % s).to(torch.float32) - 0.5).numpy().T
    out[:, :, 2] = val.numpy().T
    return out


def predict_landmarks_from_images(sd, image_stack: np.ndarray, chan_sel, batch_size: int = 2,
                                  selection_method: str = "simple", return_heatmaps: bool = False):
    """[N,256,256,4] f32 -> (landmarks [NL,N,3] f32, valid [N] bool).

    paulsenpredictor.py:167-217.  ``chan_sel`` picks the planes of the 4-plane
    stack the model was built for (the reference predictors always take all 4,
    :227/:240; other image modes exist in MVLMModel :371-383).
    """
    n = image_stack.shape[0]
    # BHWC -> BCHW as a *view* (:184-185): the tensor keeps channels-last strides,
    # which selects oneDNN's channels-last kernels exactly as in the reference.
    x = torch.from_numpy(np.ascontiguousarray(image_stack[..., list(chan_sel)])).permute(0, 3, 1, 2)
    hms = []
    for s in range(0, n, batch_size):
        hms.append(forward_last_stage(sd, x[s: s + batch_size]))
    heat = torch.cat(hms, 0)
    lms = maxima_fast(heat) if selection_method == "simple" else maxima_from_heatmaps(heat.numpy(), selection_method)
    valid = np.ones(n, dtype=bool)
    if return_heatmaps:
        return lms, valid, heat
    return lms, valid
