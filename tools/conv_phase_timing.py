#!/usr/bin/env python3
"""Where a convolution workgroup spends its cycles: prologue / K loop / epilogue of wave 0, summed over
the workgroups of one launch (diagnostic build of the library with -DMVLM_CONV_TIMING, see
mvlm_amd/csrc/conv_kernel.h).  Run on the GPU box:

  make -C mvlm_amd/csrc timing        # builds mvlm_amd/lib/libmvlm_hip_timing.so
  MVLM_HIP_LIB=mvlm_amd/lib/libmvlm_hip_timing.so python tools/conv_phase_timing.py
"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mvlm_amd import _lib  # noqa: E402

CASES = [  # name, cin, cout, size, k, batch, pre-BN, residual, bias+post-BN  (MVLM_PHASE_BATCH overrides the batch)
    ("block conv1  256->128 @128 (c128)", 256, 128, 128, 3, 64, True, True, False),
    ("conv5        256->256 @128 (c128)", 256, 256, 128, 3, 64, False, False, True),
    ("block conv2  128-> 64 @128 (c64)", 128, 64, 128, 3, 64, True, True, False),
    ("block conv3   64-> 64 @128 (c64)", 64, 64, 128, 3, 64, True, True, False),
    ("stem conv1    64-> 64 @256 (c64)", 64, 64, 256, 3, 64, True, True, False),
    ("block conv1  256->128 @ 64 (c128)", 256, 128, 64, 3, 64, True, True, False),
    ("resample 1x1  64->128 @256", 64, 128, 256, 1, 64, True, False, False),
    ("conv6        256-> 73 @128 (c80)", 256, 73, 128, 3, 64, False, False, False),
]


SMALL = [  # the latency-bound levels of a 12-view batch (MVLM_PHASE_CASES=small)
    ("block conv1  256->128 @ 32 B12", 256, 128, 32, 3, 12, True, True, False),
    ("block conv2  128-> 64 @ 32 B12", 128, 64, 32, 3, 12, True, True, False),
    ("block conv1  256->128 @ 16 B12", 256, 128, 16, 3, 12, True, True, False),
    ("block conv2  128-> 64 @ 16 B12", 128, 64, 16, 3, 12, True, True, False),
    ("block conv3   64-> 64 @ 16 B12", 64, 64, 16, 3, 12, True, True, False),
    ("block conv1  256->128 @  8 B12", 256, 128, 8, 3, 12, True, True, False),
    ("block conv2  128-> 64 @  8 B12", 128, 64, 8, 3, 12, True, True, False),
    ("block conv3   64-> 64 @  8 B12", 64, 64, 8, 3, 12, True, True, False),
    ("block conv1  256->128 @  4 B12", 256, 128, 4, 3, 12, True, True, False),
    ("block conv3   64-> 64 @  4 B12", 64, 64, 4, 3, 12, True, True, False),
    ("block conv1  256->128 @ 64 B12", 256, 128, 64, 3, 12, True, True, False),
    ("block conv2  128-> 64 @ 64 B12", 128, 64, 64, 3, 12, True, True, False),
]


def main():
    global CASES
    if os.environ.get("MVLM_PHASE_CASES") == "small":
        CASES = SMALL
    ctx = _lib.get_context(0)
    buf = torch.zeros(4, dtype=torch.int64, device="cuda")
    os.environ["MVLM_CONV_TIMING_BUF"] = hex(buf.data_ptr())
    rs = np.random.RandomState(0)
    f = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    print(f"{'layer':40s} {'ms':>7s} {'TFLOP/s':>8s} | per workgroup, wave 0: prologue / K loop / epilogue (cycles, share)")
    for name, cin, cout, size, k, batch, pre, res, post in CASES:
        batch = int(os.environ.get("MVLM_PHASE_BATCH", batch))
        x = torch.randn(batch, cin, size, size, device="cuda")
        y = torch.empty(batch, cout, size, size, device="cuda")
        r = torch.randn(batch, cout, size, size, device="cuda") if res else None
        w = (rs.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
        b = rs.standard_normal(cout).astype(np.float32) if (post or not pre) else None
        ps, pt = (rs.rand(cin).astype(np.float32) + 0.5, rs.standard_normal(cin).astype(np.float32)) if pre else (None, None)
        qs, qt = (rs.rand(cout).astype(np.float32) + 0.5, rs.standard_normal(cout).astype(np.float32)) if post else (None, None)
        times = []
        for it in range(3):
            buf.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            e0.record()
            ctx.check(ctx.lib.mvlm_conv2d(ctx.handle, C.c_void_p(x.data_ptr()), batch, cin, size, size, f(w), cout, k, f(b),
                                          f(ps), f(pt), f(qs), f(qt), C.c_void_p(r.data_ptr()) if res else None, 0,
                                          C.c_void_p(y.data_ptr())))
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        t = buf.cpu().numpy().astype(np.float64)
        n = max(t[3], 1)
        tot = t[:3].sum()
        ms = min(times)  # includes the hook's host-side packing only before the launch; event pair brackets the kernel + H2D
        flops = 2.0 * cin * cout * k * k * size * size * batch
        print(f"{name:40s} {ms:7.3f} {flops / ms / 1e9:8.1f} | {t[0] / n:9.0f} {t[1] / n:9.0f} {t[2] / n:9.0f}   "
              f"{100 * t[0] / tot:4.1f}% {100 * t[1] / tot:4.1f}% {100 * t[2] / tot:4.1f}%   ({int(n)} workgroups)")
        del x, y, r


if __name__ == "__main__":
    main()
