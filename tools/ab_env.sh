#!/bin/bash
# A/B of a run-time switch on the BASELINE workload: bench.py (no CPU baseline, no locations, no Voronoi run) with and without
# the environment assignment $2 (e.g. LRR_ICP_NO_KNN=1); $1 = tag.  Prints the three regimes and the K1 figures of both runs.
out=gpurun_out/$1
mkdir -p $out
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --no-cpu-baseline --no-locations --no-voronoi > $out/a.json 2> $out/a.err || { tail -5 $out/a.err; exit 1; }
env "$2" timeout -k 10 300 python bench.py --no-cpu-baseline --no-locations --no-voronoi > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
python - <<PY
import json
for name, f in (("default", "$out/a.json"), ("$2", "$out/b.json")):
    b = json.load(open(f))
    m = b.get("moving_transform", {})
    s = b.get("schedule", {})
    print(name, "steady ms/step %.4f value %.3e K1 us/dir %.2f | moving ms %.3f its %s value %.3e K1 mean us/dir %.1f cert skipped %.3f | schedule run_call %.3f ms" % (
        b["ms_per_step"], b["value"], 1e3 * b["roofline"]["kernel_ms"], m.get("ms_total", 0), m.get("iterations"), m.get("value", 0),
        1e3 * m["k1"]["per_direction"]["mean_ms"], m["k1"]["certificate"]["skipped_frac"] or 0, s.get("run_call_ms", 0)))
    print("   moving K1 per iteration (ms/dir):", m["k1"]["per_iteration_ms_per_direction"], m["k1"]["paths"])
    print("   schedule k1:", s.get("k1_ms_per_direction"))
PY
