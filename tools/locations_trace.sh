#!/bin/bash
# kernel statistics and timeline of the config-5 workload (bench.py --workload locations); $1 = tag
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --workload locations --no-cpu-baseline > $out/locations_t.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python tools/prof_summary.py kernels $out/trace > $out/loc_kernels.csv
python tools/trace_dump.py $out/trace > $out/dispatches.csv
rm -rf $out/trace
head -24 $out/loc_kernels.csv
