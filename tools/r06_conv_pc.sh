#!/bin/bash
# Round-6 A/B of the producer / consumer convolution kernel (csrc/conv_pc.h, PNX_CONV_PC bit 0: 64-channel layers, bit 1: 128 / 256) against the
# row-split kernels of rounds 2-5 (PNX_CONV_PC=0).  usage (GPU box): bash tools/r06_conv_pc.sh <tag>
TAG=${1:-r06a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "## quick parity" > $OUT/quick.txt
timeout 300 python tools/conv_pc_check.py >> $OUT/quick.txt 2>&1; echo "rc $?" >> $OUT/quick.txt
tail -30 $OUT/quick.txt
if grep -q "rc 0" $OUT/quick.txt; then
  PNX_CONV_PC=3 timeout 1200 python -m pytest tests/test_gpu_dense_ops.py tests/test_gpu_conv_f16.py -x -q -m gpu > $OUT/pytest_pc3.txt 2>&1; tail -5 $OUT/pytest_pc3.txt
fi
P="timeout 300 python tools/bench_conv.py --batch 12 --tiles"
for pc in 0 3; do
  export PNX_CONV_PC=$pc
  echo "## PNX_CONV_PC=$pc"
  $P --cin 64 --cout 64 --lidar 0 --dilate | tail -1
  $P --cin 64 --cout 64 --lidar 0 --dilate --res | tail -1
  $P --cin 128 --cout 128 --lidar 1 --dilate --res | tail -1
  $P --cin 256 --cout 256 --lidar 2 --dilate --res | tail -1
  $P --cin 256 --cout 256 --lidar 3 --dilate --res | tail -1
  timeout 300 python tools/bench_conv.py --cin 64 --cout 64 --batch 2 | tail -1
  timeout 300 python tools/bench_conv.py --cin 64 --cout 384 --hw 360 --batch 8 | tail -1
  timeout 300 python tools/bench_conv.py --cin 256 --cout 256 --hw 360 --batch 8 | tail -1
done > $OUT/conv_ab.txt 2>&1
cat $OUT/conv_ab.txt
echo "## section timers (producer / consumer kernel)" > $OUT/timers.txt
export PNX_CONV_PC=3 PNX_LIB=$PWD/tools/instrumented/libpnx_timers.so
{ $P --cin 64 --cout 64 --lidar 0 --dilate | tail -2
  $P --cin 64 --cout 64 --lidar 0 --dilate --res | tail -2
  $P --cin 128 --cout 128 --lidar 1 --dilate --res | tail -2
  $P --cin 256 --cout 256 --lidar 2 --dilate --res | tail -2
  timeout 300 python tools/bench_conv.py --cin 64 --cout 64 --batch 2 | tail -2; } >> $OUT/timers.txt 2>&1
cat $OUT/timers.txt
unset PNX_LIB
for pc in 0 1 3; do
  PNX_CONV_PC=$pc timeout 600 python bench.py --steps 10 --warmup 5 --no-extras --no-back-to-back 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py --no-extras PNX_CONV_PC=$pc: value', d['value'], 'ms_per_step', d['ms_per_step'], d.get('sections_us'))"
done > $OUT/bench_ab.txt 2>&1
cat $OUT/bench_ab.txt
