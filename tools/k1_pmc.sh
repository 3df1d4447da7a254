#!/bin/bash
# PMC evidence for the search kernels real runs execute: the faithful schedule of the BASELINE workload
# (k_slots_coarse2, k_match_hard2, k_match_ball2, k_match_tile) and the locations workload (k_b_match) under rocprofv3 --pmc,
# one counter group per pass, plus a kernel trace for the durations; $1 = tag.  Summaries: gpurun_out/<tag>/*.json
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name, counters..., then the python script and its arguments after --
  name=$1; shift
  ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d $out/$name -o p -- python3 "$@" > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; return 1; }
  python tools/prof_summary.py pmc $out/$name --last 40 > $out/${name}_summary.json
  rm -rf $out/$name
}
for w in sched loc; do
  if [ $w = sched ]; then cmd="tools/schedule_probe.py 200000 3"; else cmd="bench.py --workload locations --reps 2 --no-cpu-baseline"; fi
  run ${w}_util1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -- $cmd || exit 1
  run ${w}_util2 SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU -- $cmd || exit 1
  run ${w}_fetch FETCH_SIZE -- $cmd || exit 1
  run ${w}_write WRITE_SIZE -- $cmd || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${w}_trace -o t -- python3 $cmd > $out/${w}_trace.json 2> $out/${w}_trace.err || exit 1
  python tools/prof_summary.py kernels $out/${w}_trace > $out/${w}_kernels.csv
  rm -rf $out/${w}_trace
done
python tools/k1_pmc_table.py $out > $out/k1_pmc_table.json
head -c 3000 $out/k1_pmc_table.json
