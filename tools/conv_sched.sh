#!/bin/bash
# Scheduling experiments on the tap loop of the masked convolution kernels (compile-time variants under tools/instrumented/, same results as the
# default build): NOSB = no sched_barrier between k-steps, PRIO = s_setprio 2 around each k-step's MFMA cluster.
# usage (GPU box): bash tools/conv_sched.sh > gpurun_out/<tag>/conv_sched.txt
P="python tools/bench_conv.py --batch 12 --tiles --dilate"
for v in default NOSB PRIO NOSB_PRIO; do
  echo "## $v"
  if [ $v = default ]; then unset PNX_LIB; else export PNX_LIB=$PWD/tools/instrumented/libpnx_sched_$v.so; fi
  $P --cin 64 --cout 64 --lidar 0 2>/dev/null | tail -1
  $P --cin 64 --cout 64 --lidar 0 --res 2>/dev/null | tail -1
  $P --cin 128 --cout 128 --lidar 1 --res 2>/dev/null | tail -1
  $P --cin 256 --cout 256 --lidar 2 --res 2>/dev/null | tail -1
  python tools/bench_conv.py --cin 256 --cout 256 --hw 360 --batch 8 2>/dev/null | tail -1
done
