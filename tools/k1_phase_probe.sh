#!/bin/bash
# builds the HIP library with the K1 phase counters on the GPU box (the product build in the tree is left alone), runs the probe
out=gpurun_out/$1
mkdir -p $out /tmp/prof
# which code these outputs belong to (.git does not travel to the GPU box): sha256 of the kernel sources + the time
( cd ${GRAFT_REPO_ROOT:-.} && echo "code_sha $(python bench.py --code-stamp) $(date -u +%FT%TZ) $0 $*" ) > $out/code_stamp.txt
cp -r lrr_amd include tools oracle /tmp/prof/ 2>/dev/null
cd /tmp/prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DLRR_PROFILE_K1 -Iinclude -o lrr_amd/liblrr_icp_hip.so lrr_amd/csrc/lrr_icp.hip 2> $GRAFT_REPO_ROOT/$out/build.err || { tail -5 $GRAFT_REPO_ROOT/$out/build.err; exit 1; }
timeout -k 10 300 python ${PROBE:-tools/k1_phase_probe.py} > $GRAFT_REPO_ROOT/$out/phases.txt 2> $GRAFT_REPO_ROOT/$out/phases.err || { grep -v "^[WIE]2026" $GRAFT_REPO_ROOT/$out/phases.err | tail -5; exit 1; }
cat $GRAFT_REPO_ROOT/$out/phases.txt
