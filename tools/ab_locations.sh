#!/bin/bash
# builds the HIP library with extra hipcc flags in a scratch copy on the GPU box and prints config 5's rate and kernel durations
# (bench.py --workload locations, then the same under rocprofv3 --kernel-trace); $1 = tag, then the flags
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf /tmp/abl; mkdir -p $out /tmp/abl
cp -r $GRAFT_REPO_ROOT/lrr_amd $GRAFT_REPO_ROOT/include $GRAFT_REPO_ROOT/tools $GRAFT_REPO_ROOT/oracle $GRAFT_REPO_ROOT/bench.py $GRAFT_REPO_ROOT/profiles /tmp/abl/ 2>/dev/null
cd /tmp/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -Iinclude -o lrr_amd/liblrr_icp_hip.so lrr_amd/csrc/lrr_icp.hip 2> $out/build.err || { tail -5 $out/build.err; exit 1; }
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --workload locations --no-cpu-baseline > $out/loc.json 2> $out/loc.err || { tail -5 $out/loc.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 bench.py --workload locations --no-cpu-baseline > $out/loc_t.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
python tools/prof_summary.py kernels $out/trace > $out/loc_kernels.csv
rm -rf $out/trace
python - <<PY
import json
d = json.loads(open("$out/loc.json").read().strip().splitlines()[-1])
print("$*", "locations/s %.0f  total %.2f ms  iterate %.2f ms  voronoi %.0f" % (d["locations_per_s"], d["ms_total"], d["phases_ms"]["iterate"], d.get("voronoi", {}).get("locations_per_s", 0)))
PY
grep "^k_b_\|^k_run\|^k_match" $out/loc_kernels.csv | head -14
