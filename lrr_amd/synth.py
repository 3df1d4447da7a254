"""Seeded synthetic 3-D feature clouds for the ICP hot path.

The reference ships no data (the lung-CT pair of its README is a download link), so
every test and benchmark runs on clouds generated here: points sampled on a union of
smooth surfaces (SHEET: planar / spherical patches, TUBE: line segments with radial
jitter, CORNER: uniform), with the per-feature fields of ``cdcl_feature_with_shape`` /
``cdcl_feature_ICP`` (reference ``cdcl/cdcl_feature_with_shape.h:22-89``,
``cdcl/cdcl_feature_ICP.h:22-96``): location, strength, shape, error projector
(``nn^T`` / ``I - tt^T`` / ``I``, built as in ``features/itkFeatureImageFilter.txx:1053-1072``
and rounded to float as the on-disk format does) and directions.

numpy only; the same arrays are handed to the CPU oracle and to the HIP path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

CORNER, TUBE, SHEET = 0, 1, 2


@dataclass
class FeatureSet:
    """Flat (SoA-of-rows) view of a vector of features."""

    loc: np.ndarray       # (n, 3) float64  location_
    proj: np.ndarray      # (n, 9) float64  error_projector_, row-major 3x3
    strength: np.ndarray  # (n,)   float64  strength_
    shape: np.ndarray     # (n,)   uint8    shape_
    dirs: np.ndarray      # (n, 6) float64  directions_[0], directions_[1]
    cov: np.ndarray = None  # (n, 9) float64  covariance_ of cdcl_feature_with_shape (None: not set)

    @property
    def n(self) -> int:
        return int(self.loc.shape[0])

    def subset(self, idx: np.ndarray) -> "FeatureSet":
        idx = np.asarray(idx)
        return FeatureSet(
            np.ascontiguousarray(self.loc[idx]),
            np.ascontiguousarray(self.proj[idx]),
            np.ascontiguousarray(self.strength[idx]),
            np.ascontiguousarray(self.shape[idx]),
            np.ascontiguousarray(self.dirs[idx]),
            None if self.cov is None else np.ascontiguousarray(self.cov[idx]),
        )

    def in_box(self, x0, x1) -> np.ndarray:
        """Indices of features inside [x0, x1] (inclusive), the ROI test of
        ``rrl/itkLocationRegistration.cxx:1340-1346``."""
        m = np.all((self.loc >= np.asarray(x0)) & (self.loc <= np.asarray(x1)), axis=1)
        return np.nonzero(m)[0]


@dataclass
class Transform:
    """x -> A (x - c) + t  (``cdcl_trans_rigid3d`` / ``cdcl_trans_affine``)."""

    A: np.ndarray
    t: np.ndarray
    c: np.ndarray

    def map(self, x: np.ndarray) -> np.ndarray:
        return (x - self.c) @ self.A.T + self.t

    def inverse(self) -> "Transform":
        Ai = np.linalg.inv(self.A)
        return Transform(Ai, -Ai @ self.t + self.c, np.zeros(3))

    def params(self) -> np.ndarray:
        return np.concatenate([self.A.ravel(), self.t])


def rotation_xyz(ax: float, ay: float, az: float) -> np.ndarray:
    """Rx(ax) Ry(ay) Rz(az), the composition of ``cdcl_trans_rigid3d.cxx:41-63``."""
    sa, ca = np.sin(ax), np.cos(ax)
    sb, cb = np.sin(ay), np.cos(ay)
    sg, cg = np.sin(az), np.cos(az)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]], dtype=np.float64)
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]], dtype=np.float64)
    Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]], dtype=np.float64)
    return Rx @ Ry @ Rz


def planted_transform(kind: str = "rigid", extent: float = 300.0) -> Transform:
    """The planted moving->fixed transform of SURVEY.md §8d: rotation 3/-2/4 degrees about
    x/y/z around the cube centre, translation (2, -1.5, 1) mm; ``affine`` adds 2 %
    anisotropic scale."""
    R = rotation_xyz(np.deg2rad(3.0), np.deg2rad(-2.0), np.deg2rad(4.0))
    A = R.copy()
    if kind == "affine":
        A = R @ np.diag([1.02, 0.99, 1.01])
    c0 = np.full(3, extent / 2.0)
    t = c0 + np.array([2.0, -1.5, 1.0])
    return Transform(A, t, c0)


def covariances(fs: "FeatureSet") -> np.ndarray:
    """covariance_ of cdcl_feature_with_shape for a synthetic feature set: a SHEET is tight along its
    normal (0.25 mm^2) and loose in its plane (1.0 and 1.5), a TUBE loose along its tangent (1.0) and tight across (0.25,
    0.3), a CORNER isotropic-ish (0.2, 0.25, 0.3 on the coordinate axes) — distinct eigenvalues, so the eigenvectors the
    cdcl angle diagnostics compare (cdcl .txx:273-382) are well defined."""
    n = fs.n
    C = np.zeros((n, 3, 3))
    d0, d1 = fs.dirs[:, 0:3], fs.dirs[:, 3:6]

    def unit(v):
        nr = np.linalg.norm(v, axis=1, keepdims=True)
        return v / np.where(nr > 0, nr, 1.0)
    sheet, tube, corner = fs.shape == SHEET, fs.shape == TUBE, fs.shape == CORNER
    if sheet.any():
        nrm = unit(d0[sheet])
        helper = np.where(np.abs(nrm[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
        u = unit(np.cross(nrm, helper))
        v = np.cross(nrm, u)
        C[sheet] = 0.25 * nrm[:, :, None] * nrm[:, None, :] + 1.0 * u[:, :, None] * u[:, None, :] + 1.5 * v[:, :, None] * v[:, None, :]
    if tube.any():
        n1 = unit(d0[tube])
        n2 = unit(d1[tube] - np.sum(d1[tube] * n1, axis=1, keepdims=True) * n1)
        t = np.cross(n1, n2)
        C[tube] = 0.25 * n1[:, :, None] * n1[:, None, :] + 0.3 * n2[:, :, None] * n2[:, None, :] + 1.0 * t[:, :, None] * t[:, None, :]
    C[corner] = np.diag([0.2, 0.25, 0.3])
    C = 0.5 * (C + np.transpose(C, (0, 2, 1)))
    return np.ascontiguousarray(C.reshape(n, 9))


def _orthonormal_frame(rng: np.random.Generator, m: int):
    a = rng.normal(size=(m, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(m, 3))
    b -= np.sum(a * b, axis=1, keepdims=True) * a
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    c = np.cross(a, b)
    return a, b, c


class Scene:
    """A union of smooth primitives inside an ``extent`` mm cube.  Sampling it twice gives
    two independent point sets of the same surfaces (fixed and moving clouds)."""

    def __init__(self, seed: int, n_nominal: int, extent: float = 300.0):
        rng = np.random.default_rng(seed)
        self.extent = extent
        k = max(8, n_nominal // 1000)
        self.n_sheet = k
        self.n_tube = max(4, k // 2)
        margin = 0.12 * extent
        # sheets: half planar, half spherical caps
        self.sheet_center = rng.uniform(margin, extent - margin, size=(k, 3))
        self.sheet_n, self.sheet_u, self.sheet_v = _orthonormal_frame(rng, k)
        self.sheet_half = rng.uniform(15.0, 30.0, size=k)
        self.sheet_spherical = rng.random(k) < 0.5
        self.sheet_radius = rng.uniform(30.0, 80.0, size=k)
        # tubes
        self.tube_p0 = rng.uniform(margin, extent - margin, size=(self.n_tube, 3))
        self.tube_t, self.tube_n1, self.tube_n2 = _orthonormal_frame(rng, self.n_tube)
        self.tube_len = rng.uniform(25.0, 60.0, size=self.n_tube)

    def sample(self, n: int, rng: np.random.Generator, noise: float = 0.0,
               outlier_frac: float = 0.0) -> FeatureSet:
        n_sheet = int(round(0.6 * n))
        n_tube = int(round(0.3 * n))
        n_corner = n - n_sheet - n_tube
        loc = np.empty((n, 3))
        dirs = np.zeros((n, 6))
        shape = np.empty(n, dtype=np.uint8)
        proj = np.empty((n, 3, 3))
        # --- sheets
        s = rng.integers(0, self.n_sheet, size=n_sheet)
        a = rng.uniform(-1.0, 1.0, size=n_sheet) * self.sheet_half[s]
        b = rng.uniform(-1.0, 1.0, size=n_sheet) * self.sheet_half[s]
        c0, nn, uu, vv = self.sheet_center[s], self.sheet_n[s], self.sheet_u[s], self.sheet_v[s]
        flat = c0 + a[:, None] * uu + b[:, None] * vv
        R = self.sheet_radius[s]
        # spherical cap tangent to the plane at c0, centre at c0 - R n
        d = flat - (c0 - R[:, None] * nn)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        sph = (c0 - R[:, None] * nn) + R[:, None] * d
        is_sph = self.sheet_spherical[s]
        p = np.where(is_sph[:, None], sph, flat)
        nrm = np.where(is_sph[:, None], d, nn)
        loc[:n_sheet] = p
        dirs[:n_sheet, 0:3] = nrm
        shape[:n_sheet] = SHEET
        proj[:n_sheet] = nrm[:, :, None] * nrm[:, None, :]
        # --- tubes
        k = rng.integers(0, self.n_tube, size=n_tube)
        along = rng.uniform(0.0, 1.0, size=n_tube) * self.tube_len[k]
        jit = rng.normal(scale=1.0, size=(n_tube, 2))
        tt, n1, n2 = self.tube_t[k], self.tube_n1[k], self.tube_n2[k]
        loc[n_sheet:n_sheet + n_tube] = (self.tube_p0[k] + along[:, None] * tt
                                         + jit[:, 0:1] * n1 + jit[:, 1:2] * n2)
        # directions_: two normals whose cross product is the tangent (rrl .txx:351)
        dirs[n_sheet:n_sheet + n_tube, 0:3] = n1
        dirs[n_sheet:n_sheet + n_tube, 3:6] = n2
        shape[n_sheet:n_sheet + n_tube] = TUBE
        proj[n_sheet:n_sheet + n_tube] = np.eye(3)[None] - tt[:, :, None] * tt[:, None, :]
        # --- corners
        lo = n_sheet + n_tube
        loc[lo:] = rng.uniform(0.0, self.extent, size=(n_corner, 3))
        dirs[lo:, 0] = 1.0
        shape[lo:] = CORNER
        proj[lo:] = np.eye(3)[None]
        if noise > 0.0:
            loc += rng.normal(scale=noise, size=loc.shape)
        if outlier_frac > 0.0:
            m = rng.random(n) < outlier_frac
            loc[m] = rng.uniform(0.0, self.extent, size=(int(m.sum()), 3))
        strength = rng.lognormal(mean=4.0, sigma=0.7, size=n)
        # on-disk projectors are float (features/itkFeatureImageFilter.txx:1053-1072): round, then widen
        proj = proj.astype(np.float32).astype(np.float64)
        proj = 0.5 * (proj + np.transpose(proj, (0, 2, 1)))  # exactly symmetric after rounding
        perm = rng.permutation(n)  # interleave the shapes
        return FeatureSet(
            np.ascontiguousarray(loc[perm]),
            np.ascontiguousarray(proj.reshape(n, 9)[perm]),
            np.ascontiguousarray(strength[perm]),
            np.ascontiguousarray(shape[perm]),
            np.ascontiguousarray(dirs[perm]),
        )


def _pull_back(fs: FeatureSet, T: Transform) -> FeatureSet:
    """Express a fixed-space feature set in moving space: x_m = T^-1(x_f); normals
    transform as covectors (A^T n), tangents as vectors (A^-1 t)."""
    Ti = T.inverse()
    loc = Ti.map(fs.loc)
    Ai = np.linalg.inv(T.A)
    n = fs.n
    dirs = np.zeros_like(fs.dirs)
    proj = np.empty((n, 3, 3))
    d0 = fs.dirs[:, 0:3] @ T.A  # (A^T n)^T = n^T A
    nrm0 = np.linalg.norm(d0, axis=1, keepdims=True)
    d0 = d0 / np.where(nrm0 > 0, nrm0, 1.0)
    d1 = fs.dirs[:, 3:6] @ T.A
    nrm1 = np.linalg.norm(d1, axis=1, keepdims=True)
    d1 = d1 / np.where(nrm1 > 0, nrm1, 1.0)
    dirs[:, 0:3], dirs[:, 3:6] = d0, d1
    eye = np.eye(3)[None]
    sheet = fs.shape == SHEET
    tube = fs.shape == TUBE
    proj[:] = eye
    proj[sheet] = d0[sheet][:, :, None] * d0[sheet][:, None, :]
    tang = np.cross(d0[tube], d1[tube])
    tn = np.linalg.norm(tang, axis=1, keepdims=True)
    tang = tang / np.where(tn > 0, tn, 1.0)
    proj[tube] = eye - tang[:, :, None] * tang[:, None, :]
    proj = proj.astype(np.float32).astype(np.float64)
    proj = 0.5 * (proj + np.transpose(proj, (0, 2, 1)))
    return FeatureSet(np.ascontiguousarray(loc), np.ascontiguousarray(proj.reshape(n, 9)),
                      fs.strength.copy(), fs.shape.copy(), np.ascontiguousarray(dirs))


@dataclass
class VolumePair:
    fixed: FeatureSet
    moving: FeatureSet
    planted: Transform  # moving -> fixed
    extent: float


def make_pair(n_fixed: int, n_moving: int, seed: int, kind: str = "rigid",
              noise: float = 0.3, outlier_frac: float = 0.10,
              extent: float = 300.0) -> VolumePair:
    """Fixed cloud + independently resampled, noisy moving cloud expressed in moving space
    through the inverse of the planted transform."""
    scene = Scene(seed, max(n_fixed, n_moving), extent)
    rng_f = np.random.default_rng([seed, 1])
    rng_m = np.random.default_rng([seed, 2])
    fixed = scene.sample(n_fixed, rng_f)
    moving_in_fixed_space = scene.sample(n_moving, rng_m, noise=noise, outlier_frac=outlier_frac)
    T = planted_transform(kind, extent)
    moving = _pull_back(moving_in_fixed_space, T)
    return VolumePair(fixed, moving, T, extent)


def perturbed(T: Transform, seed: int, rot_deg: float = 0.5, trans_mm: float = 0.5) -> Transform:
    """planted (+) small perturbation: the 'initial transform' of configs C2/C3."""
    rng = np.random.default_rng([seed, 3])
    r = np.deg2rad(rng.uniform(-rot_deg, rot_deg, size=3))
    dR = rotation_xyz(*r)
    dt = rng.uniform(-trans_mm, trans_mm, size=3)
    return Transform(dR @ T.A, T.t + dt, T.c.copy())


def query_rois(pair: VolumePair, n_queries: int, seed: int, half: float = 25.0,
               min_features: int = 1000):
    """Config C5: query centres drawn from fixed features; each query is a 50 mm cube ROI in
    the fixed volume and the planted-mapped cube in the moving volume
    (``rrl/itkLocationRegistration.cxx:1324-1360``; need >= 1000 features each, :1632)."""
    rng = np.random.default_rng([seed, 4])
    Ti = pair.planted.inverse()
    out = []
    tries = 0
    while len(out) < n_queries and tries < 50 * n_queries:
        tries += 1
        cf = pair.fixed.loc[rng.integers(0, pair.fixed.n)]
        cm = Ti.map(cf[None])[0]
        fi = pair.fixed.in_box(cf - half, cf + half)
        mi = pair.moving.in_box(cm - half, cm + half)
        if fi.size >= min_features and mi.size >= min_features:
            out.append((cf, cm, mi, fi))
    return out
