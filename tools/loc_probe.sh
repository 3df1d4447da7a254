#!/bin/bash
# locations workload: batch tests, phase timing, per-dispatch series of the batch kernels; $1 = tag
out=gpurun_out/$1
mkdir -p $out
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
LRR_BATCH_TIMING=1 timeout -k 10 300 python bench.py --workload locations > $out/loc.json 2> $out/loc.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/loctrace -o trace -- python3 bench.py --workload locations > $out/loc_t.json 2> $out/loc_t.err || exit 1
python tools/prof_summary.py kernels $out/loctrace > $out/loc_kernels.csv
python tools/trace_series.py $out/loctrace k_b_match k_match_hard2 k_b_residual k_b_accumulate k_b_select k_b_weights k_b_stats k_b_errors > $out/series.txt
rm -rf $out/loctrace
tail -1 $out/tests.log; tail -2 $out/loc.err; head -16 $out/loc_kernels.csv
