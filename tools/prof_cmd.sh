#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command, summarised per kernel: tools/prof_cmd.sh <out-name> <top-n> <command...>
# PROF_ONLY=<regex> lists only matching kernels.  -> gpurun_out/<out-name>.md (raw database deleted: gpurun copies at most 64 MiB back)
set -u
NAME=$1; TOP=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=/tmp/prof_$$; mkdir -p $(dirname $R/gpurun_out/$NAME)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $D -o p -- "$@" > $R/gpurun_out/$NAME.log 2>&1
cd $R
python tools/prof_summary.py $(find $D -name "*.db" | head -1) $TOP "${PROF_ONLY:-}" > gpurun_out/$NAME.md 2>&1
rm -rf $D
