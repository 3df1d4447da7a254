#!/bin/bash
# the measurement set behind profiles/: bench line, kernel trace, two PMC traffic passes, one PMC utilisation pass, locations bench;
# $1 = tag (run on the GPU box from the repo root; outputs in gpurun_out/<tag>/)
tag=$1
out=gpurun_out/$tag
mkdir -p $out
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --no-cpu-baseline --no-schedule --no-moving --no-voronoi --no-locations --reps 1 > $out/trace_bench.json 2> $out/trace.err || exit 1
python tools/prof_summary.py kernels $out/trace > $out/kernel_stats.csv || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-schedule --no-moving --no-voronoi --no-locations --reps 1 --steps 40 > $out/pmc_fetch.json 2> $out/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o write -- python3 bench.py --no-cpu-baseline --no-schedule --no-moving --no-voronoi --no-locations --reps 1 --steps 40 > $out/pmc_write.json 2> $out/pmc_write.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $out/pmc_util -o util -- python3 bench.py --no-cpu-baseline --no-schedule --no-moving --no-voronoi --no-locations --reps 1 --steps 40 > $out/pmc_util.json 2> $out/pmc_util.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU -d $out/pmc_util2 -o util -- python3 bench.py --no-cpu-baseline --no-schedule --no-moving --no-voronoi --no-locations --reps 1 --steps 40 > $out/pmc_util2.json 2> $out/pmc_util2.err || exit 1
python tools/prof_summary.py pmc $out/pmc_fetch > $out/pmc_fetch_summary.json; echo "rc fetch $?"
python tools/prof_summary.py pmc $out/pmc_write > $out/pmc_write_summary.json; echo "rc write $?"
python tools/prof_summary.py pmc $out/pmc_util > $out/pmc_util_summary.json; echo "rc util $?"
python tools/prof_summary.py pmc $out/pmc_util2 > $out/pmc_util2_summary.json; echo "rc util2 $?"
timeout -k 10 300 python bench.py --workload locations > $out/locations.json 2> $out/locations.err; echo "rc locations $?"
timeout -k 10 300 python bench.py --workload cdc > $out/cdc.json 2> $out/cdc.err; echo "rc cdc $?"
LRR_BENCH_K1_SAMPLE=4 timeout -k 10 300 python bench.py --n 2000000 --steps 40 --no-cpu-baseline --no-schedule > $out/bench_2m.json 2> $out/bench_2m.err; echo "rc 2m $?"
python -c "
import json
b=json.load(open('$out/bench.json')); print(b['ms_per_step'], b['value'], b['roofline']['kernel_ms'], b['roofline']['frac'])
print(json.dumps(b.get('schedule',{}).get('k1_ms_per_direction')))
print(json.dumps({k:(v if not isinstance(v,dict) else v.get('value')) for k,v in b.get('cpu_baseline',{}).items() if k in ('value','all_cores','brute','voronoi_lookup','ms_per_iteration')}))
b=json.load(open('$out/bench_2m.json')); print('2M', b['ms_per_step'], b['value'], b['roofline']['kernel_ms'], b['roofline']['frac'])
b=json.load(open('$out/locations.json')); print('loc', b['locations_per_s'])
"
tail -n 16 $out/kernel_stats.csv
