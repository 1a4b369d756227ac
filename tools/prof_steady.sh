#!/bin/bash
# rocprofv3 kernel trace of a command that runs reader-delimited steps (bench.py, tools/train_step.py); per-kernel us/step of the last N steps.
# usage: tools/prof_steady.sh <out-name> <n-steps> <top> <command...>   -> gpurun_out/<out-name>.md
set -u
NAME=$1; N=$2; TOP=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/profs_$$; mkdir -p $(dirname $R/gpurun_out/$NAME)
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace -d $D -o p -- "$@" > $R/gpurun_out/$NAME.log 2>&1
cd $R
python tools/steady_trace.py $D $N $TOP > gpurun_out/$NAME.md 2>&1
rm -rf $D
