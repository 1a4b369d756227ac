#!/bin/bash
# value on the uniform (worst-case) cloud with and without the producer / consumer kernel: its one workgroup per CU holds all of the CU's LDS, so the decoder
# kernels on the side stream can only run between convolution launches.  usage (GPU box): bash tools/r06_uniform_ab.sh <tag>
OUT=gpurun_out/${1:-r06u}; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do for pc in 0 1; do for d in uniform sweep; do
  PNX_CONV_PC=$pc timeout 600 python bench.py --steps 10 --warmup 4 --no-extras --no-back-to-back --dist $d 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PNX_CONV_PC=$pc dist=$d: value', d['value'], 'ms_per_step', d['ms_per_step'])"
done; done; done > $OUT/uniform_ab.txt 2>&1
cat $OUT/uniform_ab.txt
