"""The drop-in boundary against the reference's own headers (build container only).

INTEGRATION.md says that inside the reference tree the drop-in classes use the reference's value types unchanged
(`-DLRR_AMD_USE_REFERENCE_TYPES`: cdcl_feature*.h, cdcl_match.h, cdcl_trans*.h, cdcl_estimation_abs.h).  VXL is not
in this image, so the reference cannot be BUILT; what can be checked is that every member of every drop-in class
— explicitly instantiated in tests/cpp/dropin_example.cpp and tests/cpp/headers_check.cpp — type-checks against those
headers: `g++ -fsyntax-only`, with a test-only include directory (tests/cpp/vxl_shim) that maps the VXL header names
the reference includes onto include/lrr_amd/vxl_compat.h.  No object code is produced and nothing of the reference
runs.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/cdcl"


def _syntax(src, extra):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(ROOT, "include")] + extra + [os.path.join(ROOT, "tests", "cpp", src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


@pytest.mark.parametrize("src", ["dropin_example.cpp", "headers_check.cpp"])
def test_headers_compile_stand_alone(src):
    _syntax(src, [])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("src", ["dropin_example.cpp", "headers_check.cpp"])
def test_drop_in_classes_type_check_against_reference_headers(src):
    _syntax(src, ["-DLRR_AMD_USE_REFERENCE_TYPES", "-I" + os.path.join(ROOT, "tests", "cpp", "vxl_shim"), "-I" + REF])


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_reference_types_are_really_the_ones_used():
    """with the switch on, the preprocessed unit pulls the type headers from the reference tree, not the look-alikes"""
    cmd = ["g++", "-std=c++17", "-E", "-DLRR_AMD_USE_REFERENCE_TYPES", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "tests", "cpp", "vxl_shim"), "-I" + REF, os.path.join(ROOT, "tests", "cpp", "headers_check.cpp")]
    out = subprocess.run(cmd, capture_output=True, text=True).stdout
    for h in ("cdcl_feature_ICP.h", "cdcl_match.h", "cdcl_trans_affine.h", "cdcl_trans_rigid3d.h", "cdcl_estimation_abs.h"):
        assert f'"{REF}/{h}"' in out, h
    assert "class cdcl_feature_ICP : public vbl_ref_count" not in out.split('lrr_amd/cdcl_types.h"')[1].split("# ")[0]
