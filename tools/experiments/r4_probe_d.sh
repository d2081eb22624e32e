#!/bin/bash
# round-4 probe: the f16x2 kernel's 128-channel x 16-row tile against its 8-row tile
out=gpurun_out/${1:-r4h}; mkdir -p $out
specs=""
for b in 96 24; do
  specs="$specs $b,256,128,128,3,59/60/62/-1 $b,256,256,128,12,59/60/62/-1 $b,256,128,64,3,59/60/62/-1 $b,128,128,128,3,59/60"
done
timeout -k 10 300 python tools/conv_shape_bench.py $specs > $out/fast16_tall.txt 2>&1
echo "rc $?" >> $out/fast16_tall.txt
cat $out/fast16_tall.txt
