#!/bin/bash
# the GPU tests in one process; $1 = tag (outputs in gpurun_out/<tag>/), further arguments = the test files / pytest options
# (none: the whole suite)
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ $# -eq 0 ]; then set -- tests; fi
timeout -k 10 1000 python -m pytest -m gpu -q --durations=8 "$@" > $out/tests.log 2>&1
rc=$?
tail -25 $out/tests.log
exit $rc
