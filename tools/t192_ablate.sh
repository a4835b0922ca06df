#!/bin/bash
# Diagnostic: builds libm3l_amd variants with pieces of mlp_t192_fwd_kernel switched off (-DT192_ABL=bits) under build_abl/ (CPU side),
# to be timed on the GPU with tools/t192_probe.py <lib>.   usage: bash tools/t192_ablate.sh "0 1 2 4 8 16 ..."
set -e
R=$(cd $(dirname $0)/.. && pwd)
O=$R/build_abl; mkdir -p $O/obj
cd $R/m3l_amd/csrc
for f in *.hip; do
  [ $O/obj/${f%.hip}.o -nt $f ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c $f -o $O/obj/${f%.hip}.o &
done
wait
F=${2:-t192}; MAC=${3:-T192_ABL}          # file and macro to vary (e.g. mlp_block MB_ABL)
for b in ${1:-0 1 2 4 8 16}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -D$MAC=$b -c $F.hip -o $O/obj/${F}_v$b.o && \
    hipcc --offload-arch=gfx950 -fPIC -shared -o $O/lib${F}_$b.so $O/obj/${F}_v$b.o $(ls $O/obj/*.o | grep -v "_v[0-9]*.o" | grep -v "/$F.o") ) &
done
wait
ls -la $O/*.so
