#!/bin/bash
# On the GPU box: tools/wgrad_probe.py over the ablated builds of wgrad.hip (bash tools/t192_ablate.sh "0 1 2 4 6 8" wgrad WG_ABL, built
# in the container) and over the kernel's environment switches.  Output: gpurun_out/wgrad_abl.txt
R=$PWD; O=$R/gpurun_out/wgrad_abl.txt; : > $O
for v in 0 1 2 4 6 8; do
  echo "== WG_ABL=$v (1 no MFMA, 2 no fragment reads, 4 no DMA, 8 no slab stores)" >> $O
  M3L_LIB_PATH=$R/build_abl/libwgrad_$v.so python3 tools/wgrad_probe.py 20 >> $O 2>&1 || exit 1
done
for w in 1 3 4; do
  echo "== M3L_WGRAD_WAVES=$w" >> $O
  M3L_WGRAD_WAVES=$w M3L_LIB_PATH=$R/build_abl/libwgrad_0.so python3 tools/wgrad_probe.py 20 >> $O 2>&1 || exit 1
done
echo "== M3L_WGRAD_NT=1" >> $O
M3L_WGRAD_NT=1 M3L_LIB_PATH=$R/build_abl/libwgrad_0.so python3 tools/wgrad_probe.py 20 >> $O 2>&1
cat $O
