#!/bin/bash
# quick A/B of a K1 change: batch tests, locations bench, main bench; $1 = tag
out=gpurun_out/$1
mkdir -p $out
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -20 $out/tests.log; exit 1; }
timeout -k 10 300 python bench.py --workload locations > $out/loc.json 2> $out/loc.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/loctrace -o trace -- python3 bench.py --workload locations > $out/loc_t.json 2> $out/loc_t.err || exit 1
python tools/prof_summary.py kernels $out/loctrace > $out/loc_kernels.csv
rm -rf $out/loctrace
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || exit 1
python - <<PY
import json
b=json.load(open('$out/loc.json')); print('loc/s', round(b['locations_per_s']), [round(x*1e3,2) for x in b['seconds_all']])
b=json.load(open('$out/bench.json')); print('main', b['ms_per_step'], b.get('schedule',{}).get('run_call_ms'), b.get('schedule',{}).get('k1_ms_per_direction'))
PY
grep -E "k_b_match|k_match_hard2" $out/loc_kernels.csv
