#!/bin/bash
# Timing experiments on the opt-in split kernels (62 bf16x3, 61 f16x2) (run on the GPU box): builds of conv_fast.hip with parts switched off
# (wrong results, timing only) against the real one, on the dominant layer shapes.  usage: tools/experiments/fast_ablate.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
OTHERS=$(ls build/*.o | grep -v conv_fast.o)
for V in ${ABLATE_VARIANTS:-NONE NO_STAGING ONE_TAP}; do
  D=""; [ $V != NONE ] && D="-DMVLM_FAST_ABLATE_$V"
  /opt/rocm/bin/hipcc $FLAGS $D -c conv_fast.hip -o /tmp/conv_fast_$V.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_$V.so /tmp/conv_fast_$V.o $OTHERS || exit 1
  echo "== $V"
  MVLM_HIP_LIB=/tmp/libmvlm_$V.so python3 - <<'PY'
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from mvlm_amd import _lib
ctx = _lib.get_context(0)
lib = ctx.lib
for (b, cin, cout, size, flags) in ((96, 256, 128, 128, 3), (96, 256, 256, 128, 12), (96, 128, 64, 128, 3), (96, 64, 64, 128, 1)):
    for v in (62, 61):
        ms, used = C.c_float(), C.c_int()
        rc = lib.mvlm_conv_bench(ctx.handle, b, cin, cout, 3, size, flags, v, 6, C.byref(ms), C.byref(used))
        fl = 2.0 * cin * cout * 9 * size * size * b
        print(f"  B{b} {cin}->{cout} @{size} variant {v:3d}: rc {rc} {ms.value * 1e3:9.1f} us  {fl / (ms.value * 1e-3) / 1e12 if rc == 0 else 0:7.1f} TF-equiv", flush=True)
PY
done
