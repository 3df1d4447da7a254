#!/bin/bash
# PMC passes over the locations workload; $1 = tag
out=gpurun_out/$1
mkdir -p $out
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $out/pmc1 -o p -- python3 bench.py --workload locations > $out/pmc1.json 2> $out/pmc1.err || exit 1
python tools/prof_summary.py pmc $out/pmc1 --last 200 > $out/pmc1_summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU -d $out/pmc2 -o p -- python3 bench.py --workload locations > $out/pmc2.json 2> $out/pmc2.err || exit 1
python tools/prof_summary.py pmc $out/pmc2 --last 200 > $out/pmc2_summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc3 -o p -- python3 bench.py --workload locations > $out/pmc3.json 2> $out/pmc3.err || exit 1
python tools/prof_summary.py pmc $out/pmc3 --last 200 > $out/pmc3_summary.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM -d $out/pmc4 -o p -- python3 bench.py --workload locations > $out/pmc4.json 2> $out/pmc4.err
python tools/prof_summary.py pmc $out/pmc4 --last 200 > $out/pmc4_summary.json
rm -rf $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4
