#!/bin/bash
# What bounds the masked convolution kernels on the LiDAR masks (C2 sweep, 12 frames): the same launches with one cost removed at a time
# (instrumented builds, results of the altered kernels are WRONG by construction; timing only):
#   SAMEW    every tap re-reads tap 0's weight fragments (L1-resident): the cost of streaming the weights from L2
#   NOSTAGE  the input tile is not staged: the cost of the HBM -> LDS staging (and its barriers' waits)
# (round 5 also built a no-store variant: the compiler deleted the arithmetic with the stores, its numbers were void; removed)
# usage (GPU box): bash tools/conv_limits.sh > gpurun_out/<tag>/conv_limits.txt
P="python tools/bench_conv.py --batch 12 --tiles --dilate"
for v in default SAMEW NOSTAGE; do
  echo "## $v"
  if [ $v = default ]; then unset PNX_LIB; else export PNX_LIB=$PWD/tools/instrumented/libpnx_dbg_$v.so; fi
  $P --cin 64 --cout 64 --lidar 0 2>/dev/null | tail -1
  $P --cin 128 --cout 128 --lidar 1 --res 2>/dev/null | tail -1
  $P --cin 256 --cout 256 --lidar 2 --res 2>/dev/null | tail -1
done
