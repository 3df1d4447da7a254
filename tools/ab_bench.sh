#!/bin/bash
# builds the HIP library with extra hipcc flags in a scratch copy on the GPU box and prints the steady-state bench figures at
# 200k and 2M features (bench.py --no-cpu-baseline --no-schedule); $1 = tag, then the flags
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf /tmp/abb; mkdir -p $out /tmp/abb
cp -r $GRAFT_REPO_ROOT/lrr_amd $GRAFT_REPO_ROOT/include $GRAFT_REPO_ROOT/tools $GRAFT_REPO_ROOT/oracle $GRAFT_REPO_ROOT/bench.py $GRAFT_REPO_ROOT/profiles /tmp/abb/ 2>/dev/null
cd /tmp/abb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -Iinclude -o lrr_amd/liblrr_icp_hip.so lrr_amd/csrc/lrr_icp.hip 2> $out/build.err || { tail -5 $out/build.err; exit 1; }
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-schedule > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-schedule --n 2000000 > $out/b2m.json 2> $out/b2m.err || { tail -5 $out/b2m.err; exit 1; }
python - <<PY
import json
for f in ("b", "b2m"):
    d = json.loads(open("$out/" + f + ".json").read().strip().splitlines()[-1])
    r = d["roofline"]; s = r["samples"]["per_direction"]
    print("$*", f, "pairs/s %.3e  ms/step %.4f  K1 median %.2f us (p10 %.2f p90 %.2f) frac %.3f  moving %.3e" % (d["value"], d["ms_per_step"], 1e3 * r["kernel_ms"], 1e3 * s["p10_ms"], 1e3 * s["p90_ms"], r["frac"], d["regimes"]["moving_transform_finest_level"]))
PY
