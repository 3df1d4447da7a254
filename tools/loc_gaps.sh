#!/bin/bash
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/loctrace -o trace -- python3 bench.py --workload locations > $out/loc_t.json 2> $out/loc_t.err || exit 1
python tools/trace_gaps.py $out/loctrace k_roi_count > $out/gaps.txt
rm -rf $out/loctrace
cat $out/gaps.txt
