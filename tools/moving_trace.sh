#!/bin/bash
# kernel timeline of the first iterations after set_transform() at the finest level (the moving-transform regime); $1 = tag
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 tools/moving_probe.py > $out/moving_t.txt 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python tools/trace_dump.py $out/trace > $out/dispatches.csv
rm -rf $out/trace
cat $out/moving_t.txt
