"""In-tree build of the HIP library and the host C++ library (no cmake/ninja needed).

    liblrr_icp_hip.so  <- lrr_amd/csrc/lrr_icp.hip       (hipcc --offload-arch=gfx950; kernels + C-ABI)
    liblrr_est.so      <- lrr_amd/csrc/lrr_est_capi.cpp  (g++; the C++ estimator classes + flat view)

hipcc cross-compiles for gfx950 without a GPU; the .so files travel to the GPU box.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
HIP_SRC = os.path.join(HERE, "csrc", "lrr_icp.hip")
HIP_INCS = [os.path.join(HERE, "csrc", f) for f in ("lrr_icp_k1.inc", "lrr_icp_solve.inc", "lrr_icp_passes.inc", "lrr_icp_fused.inc", "lrr_icp_grid.inc",
                                                  "lrr_icp_batch.inc", "lrr_icp_cdc.inc")]  # sections of the one translation unit
HIP_SO = os.path.join(HERE, "liblrr_icp_hip.so")
EST_SRC = os.path.join(HERE, "csrc", "lrr_est_capi.cpp")
EST_SO = os.path.join(HERE, "liblrr_est.so")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _headers():
    out = []
    for base, _, files in os.walk(INC):
        out += [os.path.join(base, f) for f in files if f.endswith(".h")]
    return out


def build(force: bool = False, verbose: bool = False) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = _headers()
    if force or not _newer(HIP_SO, [HIP_SRC] + HIP_INCS + hdrs):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-I" + INC, "-o", HIP_SO, HIP_SRC]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if force or not _newer(EST_SO, [EST_SRC, HIP_SO] + hdrs):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I" + INC,
               "-pthread", "-o", EST_SO, EST_SRC, "-L" + HERE, "-llrr_icp_hip", "-lz", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


if __name__ == "__main__":
    build(force=True, verbose=True)
