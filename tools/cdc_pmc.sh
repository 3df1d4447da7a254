#!/bin/bash
# PMC passes over the CDC-EM workload; $1 = tag
out=gpurun_out/$1
mkdir -p $out
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $out/pmc1 -o p -- python3 bench.py --workload cdc > $out/pmc1.json 2> $out/pmc1.err || exit 1
python tools/prof_summary.py pmc $out/pmc1 --last 20 > $out/pmc1_summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $out/pmc2 -o p -- python3 bench.py --workload cdc > $out/pmc2.json 2> $out/pmc2.err || exit 1
python tools/prof_summary.py pmc $out/pmc2 --last 20 > $out/pmc2_summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/tr -o t -- python3 bench.py --workload cdc > $out/tr.json 2> $out/tr.err || exit 1
python tools/prof_summary.py kernels $out/tr > $out/kernels.csv
rm -rf $out/pmc1 $out/pmc2 $out/tr
