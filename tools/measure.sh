#!/bin/bash
# the measurement set behind profiles/: bench line, kernel trace, two PMC passes; $1 = tag (run on the GPU box from the repo root; outputs in gpurun_out/<tag>/)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 100 > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --no-cpu-baseline > $out/trace_bench.json 2> $out/trace.err || exit 1
python tools/prof_summary.py kernels $out/trace > $out/kernel_stats.csv || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --steps 10 > $out/pmc_fetch.json 2> $out/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o write -- python3 bench.py --no-cpu-baseline --steps 10 > $out/pmc_write.json 2> $out/pmc_write.err || exit 1
python tools/prof_summary.py pmc $out/pmc_fetch > $out/pmc_fetch_summary.json; echo "rc fetch $?"
python tools/prof_summary.py pmc $out/pmc_write > $out/pmc_write_summary.json; echo "rc write $?"
cat $out/bench.json; tail -n 18 $out/kernel_stats.csv
