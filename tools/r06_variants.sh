#!/bin/bash
# usage (GPU box): bash tools/r06_variants.sh <tag> <lib names...>: bench_conv on the stage masks with each tools/instrumented/libpnx_<name>.so (PNX_CONV_PC=3)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
P="timeout 300 python tools/bench_conv.py --batch 12 --tiles"
PCS=${PCS:-3}
for pc in $PCS; do export PNX_CONV_PC=$pc; for v in "$@"; do
  echo "## $v PNX_CONV_PC=$pc"
  if [ $v = default ]; then unset PNX_LIB; else export PNX_LIB=$PWD/tools/instrumented/libpnx_$v.so; fi
  timeout 200 python tools/conv_pc_check.py 2>&1 | tail -1
  $P --cin 64 --cout 64 --lidar 0 --dilate 2>&1 | grep -v amdgpu.ids | tail -2
  $P --cin 64 --cout 64 --lidar 0 --dilate --res  2>&1 | grep -v amdgpu.ids | tail -2
  $P --cin 128 --cout 128 --lidar 1 --dilate --res  2>&1 | grep -v amdgpu.ids | tail -2
  $P --cin 256 --cout 256 --lidar 2 --dilate --res  2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/bench_conv.py --cin 64 --cout 64 --batch 2  2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/bench_conv.py --cin 256 --cout 256 --hw 360 --batch 8  2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/bench_conv.py --cin 64 --cout 384 --hw 360 --batch 8 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/bench_conv.py --cin 64 --cout 128 --hw 720 --stride 2 --batch 8 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 300 python tools/bench_conv.py --cin 128 --cout 256 --hw 360 --stride 2 --batch 8 2>&1 | grep -v amdgpu.ids | tail -2
done; done > $OUT/variants.txt 2>&1
cat $OUT/variants.txt
