#!/bin/bash
# memory-pipeline counters of the dense-level search kernels in the moving-transform regime (tools/moving_probe.py):
# texture-addresser busy and stall cycles, L1 -> L2 read requests and their latency; $1 = tag
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {
  name=$1; shift
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc "$@" -d $out/$name -o p -- python3 tools/moving_probe.py > $out/$name.txt 2> $out/$name.err || { tail -3 $out/$name.err; return 1; }
  python tools/prof_summary.py pmc $out/$name --last 12 > $out/${name}_summary.json
  rm -rf $out/$name
}
# (one or two counters per pass: more than the hardware's counter slots of a block aborts the profiler)
run ta_busy TA_BUSY_avr || exit 1
run tcp_lat TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum || exit 1
run tcp_pend TCP_PENDING_STALL_CYCLES_sum || exit 1
run ta_stall TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 1
run tcp_read TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ || exit 1
python - <<PY
import json
for n in ("ta_busy", "tcp_lat", "tcp_pend", "ta_stall", "tcp_read"):
    d = json.load(open("$out/%s_summary.json" % n))
    for k in ("k_match_ball2", "k_match_wave2", "k_match_hard2", "k_match_cert_inline2", "k_f_residual"):
        if k in d: print(n, k, json.dumps(d[k]))
PY
