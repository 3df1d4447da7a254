#!/bin/bash
# A/B of a compile-time switch: builds the HIP library with the extra hipcc flags $2.. in a scratch copy on the GPU box (the product
# build in the tree is left alone) and runs bench.py there and in the tree; $1 = tag
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out /tmp/ab
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cp -r $GRAFT_REPO_ROOT/lrr_amd $GRAFT_REPO_ROOT/include $GRAFT_REPO_ROOT/tools $GRAFT_REPO_ROOT/oracle $GRAFT_REPO_ROOT/bench.py $GRAFT_REPO_ROOT/profiles /tmp/ab/ 2>/dev/null
cd /tmp/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" -Iinclude -o lrr_amd/liblrr_icp_hip.so lrr_amd/csrc/lrr_icp.hip 2> $out/build.err || { tail -5 $out/build.err; exit 1; }
export TMPDIR=/tmp
for where in $GRAFT_REPO_ROOT /tmp/ab; do
  cd $where
  name=$([ "$where" = /tmp/ab ] && echo variant || echo tree)
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-locations --no-voronoi > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
done
python - <<PY
import json
for name in ("tree", "variant"):
    b = json.load(open("$out/%s.json" % name))
    m = b.get("moving_transform", {})
    s = b.get("schedule", {})
    print(name, "steady ms/step %.4f value %.3e K1 us/dir %.2f | moving ms %.3f value %.3e K1 mean us/dir %.1f cert skipped %.3f | schedule run_call %.3f ms" % (
        b["ms_per_step"], b["value"], 1e3 * b["roofline"]["kernel_ms"], m.get("ms_total", 0), m.get("value", 0),
        1e3 * m["k1"]["per_direction"]["mean_ms"], m["k1"]["certificate"]["skipped_frac"] or 0, s.get("run_call_ms", 0)))
    print("   moving K1 per iteration (ms/dir):", m["k1"]["per_iteration_ms_per_direction"], m["k1"]["paths"])
    print("   schedule k1:", s.get("k1_ms_per_direction"))
PY
