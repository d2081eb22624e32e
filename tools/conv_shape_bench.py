#!/usr/bin/env python3
"""Time single convolution layers through mvlm_conv_bench (zero data: optimistic clocks, good for A/B only).
usage: conv_shape_bench.py B,cin,cout,size,flags[,variant[/variant...]] ...   (flags: 1 pre-BN, 2 residual+raw, 4 bias,
8 post-BN; variant 62 = the opt-in bf16x3 kernel, -2 = rule-based exact, -1 = tuned exact, + 256 log2(parts) = a split-K
variant with the input channels divided over workgroups)"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mvlm_amd import _lib  # noqa: E402

ctx = _lib.get_context(0)
for spec in sys.argv[1:]:
    f = spec.split(",")
    b, cin, cout, size, flags = (int(v) for v in f[:5])
    ks = 1 if flags & 256 else 3  # flag 256 (this tool only): a 1x1 convolution
    flags &= 255
    for v in ([int(v) for v in f[5].split("/")] if len(f) > 5 else [62, -1]):
        ms, used = C.c_float(), C.c_int()
        rc = ctx.lib.mvlm_conv_bench(ctx.handle, b, cin, cout, ks, size, flags, v, 20, C.byref(ms), C.byref(used))
        fl = 2.0 * cin * cout * ks * ks * size * size * b
        print(f"B{b} {ks}x{ks} {cin}->{cout} @{size} flags {flags} variant {v:3d} {ctx.lib.mvlm_conv_variant_name(used.value if v < 0 else v).decode():24s}: rc {rc} {ms.value * 1e3:9.1f} us "
              f"{fl / (ms.value * 1e-3) / 1e12 if rc == 0 and ms.value > 0 else 0:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
