#!/bin/bash
# Registers, scratch and LDS of every kernel in an object file or the library (gfx950 code object inside .hip_fatbin).
# usage: tools/kernel_resources.sh mvlm_amd/csrc/build/conv_inst_g0.o
set -e
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $tmp/fat.bin
# the fat binary is a clang offload bundle; the code object starts at the ELF magic
python3 - "$tmp/fat.bin" "$tmp" <<'PY'
import sys
data = open(sys.argv[1], "rb").read()
i, n = 0, 0
while True:
    j = data.find(b"\x7fELF", i)
    if j < 0:
        break
    k = data.find(b"\x7fELF", j + 4)
    open(f"{sys.argv[2]}/co{n}.elf", "wb").write(data[j:k if k > 0 else len(data)])
    n += 1
    i = j + 4
PY
for f in $tmp/co*.elf; do
    $LLVM/llvm-readelf --notes $f 2>/dev/null | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count" \
        | sed 's/^ *//' | paste -sd' ' | sed 's/\.name:/\n.name:/g' | sed 's/_ZN[0-9a-zA-Z_]*conv_mfma_kernel/conv_mfma_kernel/' | cut -c1-230
done
rm -rf $tmp
