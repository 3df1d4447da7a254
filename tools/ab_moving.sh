#!/bin/bash
# builds the HIP library with extra hipcc flags in a scratch copy on the GPU box and prints the kernel durations of the
# moving-transform regime (tools/moving_probe.py under rocprofv3 --kernel-trace); $1 = tag, then the flags
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out /tmp/abm
cp -r $GRAFT_REPO_ROOT/lrr_amd $GRAFT_REPO_ROOT/include $GRAFT_REPO_ROOT/tools $GRAFT_REPO_ROOT/oracle $GRAFT_REPO_ROOT/bench.py /tmp/abm/ 2>/dev/null
cd /tmp/abm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -Iinclude -o lrr_amd/liblrr_icp_hip.so lrr_amd/csrc/lrr_icp.hip 2> $out/build.err || { tail -5 $out/build.err; exit 1; }
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/trace -o trace -- python3 tools/moving_probe.py > $out/moving_t.txt 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python tools/trace_dump.py $out/trace > $out/dispatches.csv
rm -rf $out/trace
tail -2 $out/moving_t.txt
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/dispatches.csv")))
d = collections.defaultdict(list)
for r in rows[-150:]:
    d[r["kernel"][:28]].append(float(r["dur_us"]))
for k in ("k_match_wave2", "k_match_ball2", "k_match_hard2"):
    if k in d: print("$*", k, len(d[k]), "avg %.1f min %.1f max %.1f us" % (sum(d[k]) / len(d[k]), min(d[k]), max(d[k])))
PY
