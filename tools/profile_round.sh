#!/bin/bash
# Profiles of the round (run on the GPU box via gpurun): kernel traces of bench.py and of the reader alone, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, each in its own run with nothing but --kernel-trace) behind bench.py's roofline.traffic.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
set -u
TAG=${1:-r04}
BATCH=${PNX_BENCH_BATCH:-12}   # frames per reader launch: what bench.py runs (profiles/pmc_traffic.json key C2_b<BATCH>_sweep)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/bench_trace -o p -- python $R/bench.py --steps 6 --warmup 4 --no-extras > $O/bench_trace.json 2> $O/bench_trace.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/reader_trace -o p -- python $R/tools/reader_ab.py --exact "spans default" --batch $BATCH --iters 20 > $O/reader_trace.log 2>&1
# PMC passes in the condition bench.py's roofline.frac is measured in: the reader calls of the detector's loop (bench.py --no-extras --no-back-to-back:
# every reader call of the process follows a step's convolutions), each counter in its own run with nothing but --kernel-trace
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 4 --no-extras --no-back-to-back > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 4 --no-extras --no-back-to-back > $O/pmc_write.log 2>&1
cd $R
python tools/steady_trace.py $O/bench_trace 3 45 > $O/bench_steady_trace.md 2>&1
python tools/prof_summary.py $(find $O/reader_trace -name "*.db" | head -1) 30 > $O/reader_kernels.md 2>&1
python tools/span_spread.py $O/reader_trace 4 > $O/span_spread.md 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
echo "pmc files: $F $W"
cp profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
python tools/pmc_traffic.py "$F" "$W" C2_b${BATCH}_sweep $O/pmc_traffic.json "in the detector's loop: every reader call of bench.py --steps 6 --warmup 4 --no-extras --no-back-to-back (one rocprofv3 --pmc pass per counter)"
tail -3 $O/reader_trace.log; head -40 $O/reader_kernels.md
# gpurun copies at most 64 MiB back: keep the summaries and the per-kernel statistics, drop the raw databases / counter dumps
for d in bench_trace reader_trace; do find $O/$d -name "*kernel_stats.csv" -exec cp {} $O/${d}_kernel_stats.csv \; ; done
rm -rf $O/bench_trace $O/reader_trace $O/pmc_fetch $O/pmc_write
du -sh $O
