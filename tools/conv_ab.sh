#!/bin/bash
# Round-5 A/B of the convolution kernels on the LiDAR masks of the C2 sweep (12 frames, persistent workspaces + tile lists) and end to end:
#   default    residual joined in the epilogue (conv_rows / conv_rows_x), 16 x 32 tiles for 64 -> 64
#   res_early  the round 2-4 form (-DPNX_CONV_RES_EARLY: residual starts the accumulators), tools/instrumented/libpnx_res_early.so
#   (round 5 also measured 64 -> 64 on 8 x 32 tiles with three workgroups per CU: 855 / 1032 us against 742 / 849, profiles/r05_conv_ab.txt; removed)
# plus the in-kernel section timers (tools/instrumented/libpnx_timers.so).  usage (GPU box): bash tools/conv_ab.sh > gpurun_out/<tag>/conv_ab.txt
P="python tools/bench_conv.py --batch 12 --tiles"
run_set() {
  $P --cin 64 --cout 64 --lidar 0 --dilate | tail -1
  $P --cin 64 --cout 64 --lidar 0 --dilate --res | tail -1
  $P --cin 128 --cout 128 --lidar 1 --dilate --res | tail -1
  $P --cin 256 --cout 256 --lidar 2 --dilate --res | tail -1
  $P --cin 256 --cout 256 --lidar 3 --dilate --res | tail -1
  python bench.py --steps 10 --warmup 5 --no-extras --no-back-to-back | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py --no-extras: value', d['value'], 'ms_per_step', d['ms_per_step'])"
}
echo "## default"; run_set
echo "## res_early"; PNX_LIB=$PWD/tools/instrumented/libpnx_res_early.so run_set
echo "## section timers (default kernels)"
export PNX_LIB=$PWD/tools/instrumented/libpnx_timers.so
$P --cin 64 --cout 64 --lidar 0 --dilate | tail -2
$P --cin 64 --cout 64 --lidar 0 --dilate --res | tail -2
python tools/bench_conv.py --cin 64 --cout 64 --batch 2 | tail -2
