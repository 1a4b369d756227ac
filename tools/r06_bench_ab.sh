#!/bin/bash
# usage (GPU box): bash tools/r06_bench_ab.sh <tag> <lib name|default> <PNX_CONV_PC values...>: end-to-end bench.py per setting, twice
TAG=$1; LIBN=$2; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ $LIBN != default ]; then export PNX_LIB=$PWD/tools/instrumented/libpnx_$LIBN.so; fi
for rep in 1 2; do for pc in "$@"; do
  PNX_CONV_PC=$pc timeout 600 python bench.py --steps 12 --warmup 5 --no-extras --no-back-to-back 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib $LIBN PNX_CONV_PC=$pc: value', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v) for k, v in (d.get('sections_us') or {}).items()})"
done; done > $OUT/bench_ab.txt 2>&1
cat $OUT/bench_ab.txt
