"""The C++ drop-in classes used directly, the way location_registration uses the reference
classes (tests/cpp/dropin_example.cpp): compiles on CPU; on the GPU its result must equal the
oracle's run of the same problem."""
import os
import struct
import subprocess

import numpy as np
import pytest

from lrr_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_example.cpp")


def _compile(out):
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "lrr_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-o", out, SRC, "-L" + lib, "-llrr_icp_hip", "-Wl,-rpath," + lib])


def _write_problem(path, pair, T0):
    with open(path, "wb") as f:
        f.write(np.concatenate([T0.A.ravel(), T0.t, T0.c]).astype(np.float64).tobytes())
        for fs in (pair.moving, pair.fixed):
            f.write(struct.pack("q", fs.n))
            rec = np.concatenate([fs.loc, fs.proj, fs.strength[:, None], fs.shape[:, None].astype(np.float64), fs.dirs], axis=1)
            f.write(np.ascontiguousarray(rec, dtype=np.float64).tobytes())


def test_dropin_example_compiles(tmp_path):
    exe = str(tmp_path / "dropin_example")
    _compile(exe)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_dropin_example_matches_oracle(tmp_path):
    from oracle import binding as ob
    exe = str(tmp_path / "dropin_example")
    _compile(exe)
    pair = synth.make_pair(6000, 5500, 20248, kind="affine")
    T0 = synth.perturbed(pair.planted, 21)
    prob = str(tmp_path / "problem.bin")
    _write_problem(prob, pair, T0)
    out = subprocess.check_output([exe, prob], text=True)
    lines = dict((l.split()[0], l.split()[1:]) for l in out.strip().splitlines())
    # the class seeds its subset permutation with 1 by default
    c = ob.OracleEstimator(ob.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                           moving_inside=pair.moving, fixed_inside=pair.fixed, seed=1)
    conv, its = c.run(70)
    assert int(lines["converged"][0]) == int(conv) and int(lines["converged"][2]) == its
    assert int(lines["converged"][4]) == 1  # the display callback saw one-to-one matches once
    A, t, cc, _ = c.get_transform()
    got_A = np.array([float(x) for x in lines["A"]]).reshape(3, 3)
    got_t = np.array([float(x) for x in lines["t"]])
    assert np.abs(got_A - A).max() < 1e-6 * np.abs(A).max() and np.abs(got_t - t).max() < 1e-6 * np.abs(t).max()
    assert abs(float(lines["weighted_error"][0]) - c.weighted_error(0)) < 1e-7
    assert abs(float(lines["weighted_error"][2]) - c.weighted_error(1)) < 1e-7
    tb = np.array([float(x) for x in lines["tb"]])
    assert np.abs(tb - c.get_transform(1)[1]).max() < 1e-6 * np.abs(tb).max()
