#!/bin/bash
out=gpurun_out/$1
mkdir -p $out
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/loctrace -o trace -- python3 bench.py --workload locations > $out/loc_t.json 2> $out/loc_t.err || exit 1
python tools/trace_gaps.py $out/loctrace k_roi_count > $out/gaps.txt
rm -rf $out/loctrace
cat $out/gaps.txt
