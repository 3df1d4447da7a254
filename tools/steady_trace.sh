#!/bin/bash
# kernel timeline of the steady-state iterations of bench.py (every dispatch with the idle gap before it); $1 = tag
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 bench.py --no-cpu-baseline --no-schedule --no-moving --no-voronoi --no-locations --reps 1 --steps 40 > $out/steady_t.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python tools/trace_dump.py $out/trace > $out/dispatches.csv
rm -rf $out/trace
python tools/schedule_iters.py $out/dispatches.csv 6 | cut -c1-600
