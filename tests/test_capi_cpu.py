"""CPU-side checks of the product libraries: they build, load, export every symbol that
include/lrr_icp.h declares, and fail loudly (no CPU fallback) when no GPU is usable."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from lrr_amd import estimation as E
    return E


def test_header_symbols_all_exported(built):
    E = built
    hdr = open(os.path.join(ROOT, "include", "lrr_icp.h")).read()
    declared = sorted(set(re.findall(r"\b(lrr_icp_[a-z_0-9]+)\s*\(", hdr)))
    assert sorted(E.ICP_SYMBOLS) == declared, "estimation.ICP_SYMBOLS out of sync with include/lrr_icp.h"
    L = E.hip_lib()
    for s in declared:
        assert hasattr(L, s), f"liblrr_icp_hip.so does not export {s}"


def test_subset_perm_matches_oracle(built):
    from oracle import binding as ob
    E = built
    for n in (1, 2, 3, 17, 1000, 4097):
        a, sa = E.subset_perm(n, 1234 + n)
        b, sb = ob.shuffle(n, 1234 + n)
        assert np.array_equal(a, b) and sa == sb
        assert np.array_equal(np.sort(a), np.arange(n))


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    E = built
    from lrr_amd import synth
    pair = synth.make_pair(50, 50, 1)
    with pytest.raises(RuntimeError, match="no CPU path|HIP"):
        E.Estimator(E.CDCL_FORWARD, 6, pair.moving, pair.fixed, np.eye(3), np.zeros(3), np.zeros(3))


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under lrr_amd/ or include/ may reference it"""
    bad = []
    for base in ("lrr_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".inc")):
                    txt = open(os.path.join(d, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle|lrr_oracle|liblrr_oracle", txt, flags=re.M):
                        bad.append(os.path.join(d, f))
    assert not bad, bad
