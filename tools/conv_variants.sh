#!/bin/bash
# Achieved TFLOP/s of every convolution kernel variant: dense (all sites active) and on the LiDAR masks of the C2 sweep (8 frames).
# usage (GPU box): bash tools/conv_variants.sh > gpurun_out/conv_variants.txt
P="python tools/bench_conv.py"
echo "## dense (mask = None)"
$P --cin 64 --cout 64 --batch 2 | tail -1
$P --cin 64 --cout 64 --batch 2 --res | tail -1
$P --cin 128 --cout 128 --hw 720 --batch 4 | tail -1
$P --cin 256 --cout 256 --hw 360 --batch 8 | tail -1
$P --cin 64 --cout 384 --hw 360 --batch 8 | tail -1
$P --cin 64 --cout 128 --hw 720 --batch 4 --stride 2 | tail -1
$P --cin 128 --cout 256 --hw 360 --batch 8 --stride 2 | tail -1
$P --cin 256 --cout 256 --hw 180 --batch 8 --stride 2 | tail -1
echo "## MIOpen/CK + epilogue pass, dense, same shapes"
$P --cin 64 --cout 64 --batch 2 --miopen | tail -1
$P --cin 128 --cout 128 --hw 720 --batch 4 --miopen | tail -1
$P --cin 256 --cout 256 --hw 360 --batch 8 --miopen | tail -1
$P --cin 64 --cout 384 --hw 360 --batch 8 --miopen | tail -1
echo "## LiDAR masks (C2 sweep, 8 frames, residual, persistent workspace + tile list); TFLOP/s are dense-equivalent"
for st in 0 1 2 3; do c=$((64 << st)); if [ $c -gt 256 ]; then c=256; fi
  $P --cin $c --cout $c --batch 8 --lidar $st --res --tiles | tail -3 | grep -v "^tile list" | tr '\n' ' '; echo
done
$P --cin 64 --cout 128 --batch 8 --lidar 1 --stride 2 | tail -2 | tr '\n' ' '; echo
$P --cin 128 --cout 256 --batch 8 --lidar 2 --stride 2 | tail -2 | tr '\n' ' '; echo
$P --cin 256 --cout 256 --batch 8 --lidar 3 --stride 2 | tail -2 | tr '\n' ' '; echo
