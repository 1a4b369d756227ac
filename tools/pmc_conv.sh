#!/bin/bash
# One SQ counter pass (8 slots, nothing but --kernel-trace beside it) over the convolution kernels: where do the wave cycles go?
# usage (GPU box): bash tools/pmc_conv.sh <outdir>   -> <outdir>/summary.md
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$(realpath -m ${1:-$R/gpurun_out/pmc_conv})
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {  # tag, bench_conv args
  local tag=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/$tag -o p --output-format csv -- python $R/tools/bench_conv.py "$@" --iters 5 > $O/$tag.log 2>&1
}
run head_64_384 --cin 64 --cout 384 --hw 360 --batch 8
run dense_256 --cin 256 --cout 256 --hw 360 --batch 8
run lidar_64 --cin 64 --cout 64 --batch 8 --lidar 0 --res --tiles
cd $R
python tools/pmc_conv_summary.py $O > $O/summary.md 2>&1
cat $O/summary.md
