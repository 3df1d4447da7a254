"""Host linear algebra of the C++ classes (include/lrr_amd/linalg.h): the inverse of the normal
matrix that stands in for vnl_svd::inverse() (cdcl/cdcl_estimation_ICP_matching_all.txx:1073-1074).
Cholesky route for well-conditioned positive definite input, Jacobi pseudo-inverse otherwise."""
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, mats):
    exe = str(tmp_path / "linalg_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "linalg_check.cpp")])
    path = str(tmp_path / "mats.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("i", len(mats)))
        for A in mats:
            f.write(struct.pack("i", A.shape[0]))
            f.write(np.ascontiguousarray(A, dtype=np.float64).tobytes())
    out = []
    for line, A in zip(subprocess.check_output([exe, path], text=True).strip().splitlines(), mats):
        tok = line.split()
        n = int(tok[0])
        assert n == A.shape[0]
        out.append((bool(int(tok[1])), np.array([float(x) for x in tok[2:]]).reshape(n, n)))
    return out


def test_sym_inverse_against_numpy(tmp_path):
    rng = np.random.default_rng(5)
    mats, kinds = [], []
    for n in (6, 12):
        for grade in (1.0, 30.0, 1e3):  # affine normal equations are graded: l^2 against 1
            J = rng.normal(size=(400, n)) * np.where(np.arange(n) < (n * 3) // 4, grade, 1.0)
            mats.append(J.T @ J)
            kinds.append("spd")
        J = rng.normal(size=(400, n))
        J[:, n - 1] = J[:, 0] + J[:, 1]  # rank n-1: exactly dependent columns
        mats.append(J.T @ J)
        kinds.append("singular")
        D = np.diag(np.concatenate([[2.0, 4.0], np.zeros(n - 2)]))
        mats.append(D)
        kinds.append("zeros")
        Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        mats.append(Q @ np.diag(np.concatenate([[-3.0], np.linspace(1, 5, n - 1)])) @ Q.T)
        kinds.append("indefinite")
    res = _run(tmp_path, mats)
    for (chol, inv), A, kind in zip(res, mats, kinds):
        n = A.shape[0]
        assert np.array_equal(inv, inv.T) or np.allclose(inv, inv.T, rtol=1e-12, atol=0)
        if kind == "spd":
            assert chol, "well-conditioned positive definite input must take the Cholesky route"
            ref = np.linalg.inv(A)
            assert np.abs(inv - ref).max() <= 1e-9 * np.abs(ref).max()
            assert np.abs(A @ inv - np.eye(n)).max() < 1e-8
        elif kind == "zeros":
            assert not chol
            ref = np.zeros((n, n)); ref[0, 0] = 0.5; ref[1, 1] = 0.25
            assert np.allclose(inv, ref, atol=1e-15)  # zero eigenvalues are left out
        elif kind == "indefinite":
            assert not chol
            ref = np.linalg.inv(A)
            assert np.abs(inv - ref).max() <= 1e-9 * np.abs(ref).max()
        else:  # numerically singular: Cholesky must refuse, the Jacobi route decides
            assert not chol
            assert np.all(np.isfinite(inv))
