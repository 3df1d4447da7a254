#!/bin/bash
# kernel trace of the faithful schedule run as one run() call; $1 = tag
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/schedule_probe.py > $out/schedule.json 2> $out/schedule.err || { tail -5 $out/schedule.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 tools/schedule_probe.py 200000 2 > $out/schedule_t.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python tools/trace_dump.py $out/trace > $out/dispatches.csv
rm -rf $out/trace
cat $out/schedule.json | head -c 1500
