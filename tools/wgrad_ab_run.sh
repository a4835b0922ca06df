#!/bin/bash
# On the GPU box: the grouped weight-gradient launch alone over tile heights and split counts.  Output: gpurun_out/wgrad_ab.txt
R=$PWD; O=$R/gpurun_out/wgrad_ab.txt; : > $O
python3 -m pytest tests/test_kernels_gpu.py -q -k "gemm_tn" >> $O 2>&1 || { tail -30 $O; exit 1; }
for na in 4 6; do for w in 1 2 3; do
  echo "== M3L_WGRAD_NA=$na M3L_WGRAD_WAVES=$w" >> $O
  M3L_WGRAD_NA=$na M3L_WGRAD_WAVES=$w python3 tools/wgrad_probe.py 20 >> $O 2>&1 || exit 1
done; done
grep -v amdgpu.ids $O
