#!/bin/bash
# Instrumented / variant build of the convolution kernels only: recompiles csrc/conv3x3.hip (bf16 + f16) with extra defines and links it with the
# objects of the last full build.  usage: tools/build_conv_variant.sh <name> [-DPC_PD=3 ...]   ->   tools/instrumented/libpnx_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/.."
C=pillarnext_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result -fno-honor-nans"
mkdir -p tools/instrumented
/opt/rocm/bin/hipcc $F "$@" -x hip -c $C/conv3x3.hip -o $C/conv3x3_v_$name.o &
/opt/rocm/bin/hipcc $F "$@" -DPNX_CONV_F16 -x hip -c $C/conv3x3.hip -o $C/conv3x3_f16_v_$name.o &
wait
objs=$(ls $C/*.o | grep -v "_v_\|_tm\.o\|conv3x3\.o\|conv3x3_f16\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/instrumented/libpnx_$name.so $objs $C/conv3x3_v_$name.o $C/conv3x3_f16_v_$name.o
rm -f $C/conv3x3_v_$name.o $C/conv3x3_f16_v_$name.o
echo tools/instrumented/libpnx_$name.so
