#!/bin/bash
# kernel table of the reader alone: tools/prof_reader.sh <out-name> <variant> [batch]   -> gpurun_out/<out-name>.md
set -u
NAME=$1; VAR=$2; B=${3:-12}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/profr_$$
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $D -o p -- python $R/tools/reader_ab.py --batch $B --exact "$VAR" > $R/gpurun_out/$NAME.log 2>&1
cd $R
python tools/prof_summary.py $(find $D -name "*.db" | head -1) 16 "k_|Kernel|fill" > gpurun_out/$NAME.md 2>&1
rm -rf $D
