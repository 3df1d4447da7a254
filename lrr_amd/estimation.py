"""ctypes view of the C++ drop-in estimator classes (include/lrr_amd/*.h) and of the C-ABI.

Python is plumbing here (tests, bench, the multi-GPU driver): the product is the HIP library
``liblrr_icp_hip.so`` (C-ABI: include/lrr_icp.h) and the C++ classes above it.  There is no CPU
fallback: loading fails loudly when the libraries are missing, and creating an estimator fails
loudly when no GPU is usable.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(_HERE, "liblrr_icp_hip.so")
EST_SO = os.path.join(_HERE, "liblrr_est.so")

CDCL_FORWARD, RRL_FORWARD, CDCL_SYMMETRIC, RRL_SYMMETRIC = 0, 1, 2, 3
NN_AUTO, NN_BRUTE, NN_GRID, NN_VORONOI = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)

# every symbol include/lrr_icp.h declares
ICP_SYMBOLS = [
    "lrr_icp_create", "lrr_icp_destroy", "lrr_icp_last_error", "lrr_icp_synchronize",
    "lrr_icp_timer_start", "lrr_icp_timer_stop", "lrr_icp_match_profile", "lrr_icp_match_profile_samples", "lrr_icp_event_overhead", "lrr_icp_cert_stats", "lrr_icp_k1_path_counts",
    "lrr_icp_cloud_create", "lrr_icp_cloud_destroy", "lrr_icp_cloud_size", "lrr_icp_cloud_build_voronoi",
    "lrr_icp_dir_create", "lrr_icp_dir_destroy", "lrr_icp_dir_set_from", "lrr_icp_dir_num_matches",
    "lrr_icp_perm_keys", "lrr_icp_subset_perm", "lrr_icp_match", "lrr_icp_match_pair", "lrr_icp_premap",
    "lrr_icp_residual_pass", "lrr_icp_muse_scale", "lrr_icp_assign_weights",
    "lrr_icp_normalize_stats", "lrr_icp_normalize_stats_merged", "lrr_icp_clone_weights", "lrr_icp_weight_by_strength",
    "lrr_icp_weight_spatially", "lrr_icp_accumulate", "lrr_icp_errors", "lrr_icp_errors_launch", "lrr_icp_errors_collect", "lrr_icp_median_error", "lrr_icp_cloud_set_shapes", "lrr_icp_cloud_set_shape_vectors", "lrr_icp_angles", "lrr_icp_angles_unmapped", "lrr_icp_transfer_covariance", "lrr_icp_get_matches",
    "lrr_icp_estimate_symmetric", "lrr_icp_estimate_forward",
    "lrr_icp_nearest", "lrr_icp_debug_set_match", "lrr_icp_cloud_set_covariances", "lrr_icp_cdc_compute_weights",
    "lrr_icp_cdc_set_create", "lrr_icp_cdc_set_destroy", "lrr_icp_cdc_set_compute_weights", "lrr_icp_cdc_set_get", "lrr_icp_cdc_set_stats", "lrr_icp_cdc_set_row_sums", "lrr_icp_cdc_set_renormalize", "lrr_icp_cdc_set_premap", "lrr_icp_cdc_set_closest_euclidean", "lrr_icp_cdc_objective_par", "lrr_icp_cdc_objective_cov", "lrr_icp_cdc_normalize_stats",
    "lrr_icp_batch_create", "lrr_icp_batch_destroy", "lrr_icp_batch_size", "lrr_icp_batch_set_nn_method", "lrr_icp_batch_set_rois", "lrr_icp_batch_get_roi",
    "lrr_icp_batch_match", "lrr_icp_batch_estimate", "lrr_icp_batch_errors", "lrr_icp_batch_angles", "lrr_icp_batch_iterate",
    "lrr_icp_batch_num_matches", "lrr_icp_batch_get_matches",
]


class _Features(C.Structure):
    _fields_ = [("n", C.c_int64), ("loc", _dp), ("proj", _dp), ("strength", _dp),
                ("shape", C.POINTER(C.c_uint8)), ("dirs", _dp), ("cov", _dp)]


class Xform(C.Structure):
    _fields_ = [("A", C.c_double * 9), ("t", C.c_double * 3), ("c", C.c_double * 3)]


class Norm(C.Structure):
    _fields_ = [("cm", C.c_double * 3), ("rm", C.c_double), ("cf", C.c_double * 3), ("rf", C.c_double)]


_hip = None
_est = None


def hip_lib() -> C.CDLL:
    """The HIP library (C-ABI of include/lrr_icp.h)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise RuntimeError(f"{HIP_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the ICP hot path has no CPU fallback)")
        L = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        L.lrr_icp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.lrr_icp_destroy.argtypes = [C.c_void_p]
        L.lrr_icp_last_error.argtypes = [C.c_void_p]
        L.lrr_icp_last_error.restype = C.c_char_p
        L.lrr_icp_synchronize.argtypes = [C.c_void_p]
        L.lrr_icp_timer_start.argtypes = [C.c_void_p]
        L.lrr_icp_timer_stop.argtypes = [C.c_void_p, _dp]
        L.lrr_icp_match_profile.argtypes = [C.c_void_p, C.c_int, _dp, _ip, C.c_int]
        L.lrr_icp_event_overhead.argtypes = [C.c_void_p, C.c_int, _dp]
        L.lrr_icp_cert_stats.argtypes = [C.c_void_p, _ip, _ip, C.c_int]
        L.lrr_icp_cloud_create.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp, C.c_int, C.POINTER(C.c_void_p)]
        L.lrr_icp_cloud_destroy.argtypes = [C.c_void_p]
        L.lrr_icp_cloud_size.argtypes = [C.c_void_p]
        L.lrr_icp_cloud_size.restype = C.c_int64
        L.lrr_icp_perm_keys.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.lrr_icp_subset_perm.argtypes = [C.c_int64, C.c_int64, C.POINTER(C.c_uint32)]
        L.lrr_icp_subset_perm.restype = C.c_int64
        L.lrr_icp_nearest.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_int, _ip]
        L.lrr_icp_batch_num_matches.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.lrr_icp_batch_num_matches.restype = C.c_int64
        L.lrr_icp_batch_get_matches.argtypes = [C.c_void_p, C.c_int64, C.c_int, _ip, _ip, _dp]
        L.lrr_icp_batch_get_roi.argtypes = [C.c_void_p, C.c_int64, C.c_int, _ip]
        _hip = L
    return _hip


def est_lib() -> C.CDLL:
    """The host C++ estimator classes behind their flat ctypes view."""
    global _est
    if _est is None:
        hip_lib()
        if not os.path.exists(EST_SO):
            raise RuntimeError(f"{EST_SO} is missing: run __graft_entry__.build()")
        L = C.CDLL(EST_SO)
        L.lrr_est_last_error.restype = C.c_char_p
        L.lrr_est_set_device.argtypes = [C.c_int]
        L.lrr_est_context.restype = C.c_void_p
        L.lrr_est_create.restype = C.c_void_p
        L.lrr_est_create.argtypes = [C.c_int, C.c_int, C.POINTER(_Features), C.POINTER(_Features),
                                     C.POINTER(_Features), C.POINTER(_Features), _dp, _dp, _dp,
                                     C.c_uint64, C.c_int]
        L.lrr_est_destroy.argtypes = [C.c_void_p]
        L.lrr_est_set_transform.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.lrr_est_set_features.argtypes = [C.c_void_p, C.POINTER(_Features), C.POINTER(_Features)]
        L.lrr_est_weight_by_strength.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_normalize_matches.argtypes = [C.c_void_p]
        L.lrr_est_set_number_points.argtypes = [C.c_void_p, C.c_int64]
        L.lrr_est_inject_initial_scale.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.lrr_est_set_subset_order.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int64]
        L.lrr_est_set_nn_method.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_set_voronoi_maps.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_find_closest_euclidean.argtypes = [C.c_void_p]
        L.lrr_est_one_iteration.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_int)]
        L.lrr_est_iterate.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_int)]
        L.lrr_est_run.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
        L.lrr_est_run_counted.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_int), _ip]
        L.lrr_est_set_subset_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.lrr_est_estimate_LS.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_estimate_LS_backward.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_get_transform.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        L.lrr_est_RMS_error.argtypes = [C.c_void_p, _dp]
        L.lrr_est_median_error.argtypes = [C.c_void_p, _dp]
        for name in ("weighted_error", "sheet_angles", "tube_angles", "last_scale"):
            getattr(L, "lrr_est_" + name).argtypes = [C.c_void_p, C.c_int, _dp]
        L.lrr_est_compute_transfer_error_covariance.argtypes = [C.c_void_p, _dp, _dp]
        L.lrr_est_oscillated.argtypes = [C.c_void_p]
        L.lrr_est_finest_level.argtypes = [C.c_void_p]
        L.lrr_est_num_matches.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_num_matches.restype = C.c_int64
        L.lrr_est_get_matches.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp]
        L.lrr_est_get_number_points.argtypes = [C.c_void_p, C.c_int]
        L.lrr_est_get_number_points.restype = C.c_int64
        L.lrr_est_last_normal_equations.argtypes = [C.c_void_p, _dp, _dp]
        L.lrr_est_batch_create.restype = C.c_void_p
        L.lrr_est_batch_create.argtypes = [C.POINTER(_Features), C.POINTER(_Features)]
        L.lrr_est_batch_destroy.argtypes = [C.c_void_p]
        L.lrr_est_batch_handle.restype = C.c_void_p
        L.lrr_est_batch_handle.argtypes = [C.c_void_p]
        L.lrr_est_batch_run.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_uint64), C.c_uint,
                                        C.c_int, C.c_int, _dp]
        _est = L
    return _est


def _d(a):
    return a.ctypes.data_as(_dp)


def _features(fs, keep):
    if fs is None:
        return None
    loc = np.ascontiguousarray(fs.loc, dtype=np.float64)
    proj = np.ascontiguousarray(fs.proj, dtype=np.float64)
    st = np.ascontiguousarray(fs.strength, dtype=np.float64)
    sh = np.ascontiguousarray(fs.shape, dtype=np.uint8)
    di = np.ascontiguousarray(fs.dirs, dtype=np.float64)
    cov = np.ascontiguousarray(fs.cov, dtype=np.float64) if getattr(fs, "cov", None) is not None else None
    keep.extend([loc, proj, st, sh, di, cov])
    return _Features(loc.shape[0], _d(loc), _d(proj), _d(st), sh.ctypes.data_as(C.POINTER(C.c_uint8)), _d(di),
                     _d(cov) if cov is not None else None)


def set_device(index: int) -> None:
    if est_lib().lrr_est_set_device(int(index)) != 0:
        raise RuntimeError(est_lib().lrr_est_last_error().decode())


def context() -> int:
    """lrr_icp_ctx* of the calling thread (for the timing helpers)."""
    h = est_lib().lrr_est_context()
    if not h:
        raise RuntimeError(est_lib().lrr_est_last_error().decode())
    return h


class DeviceTimer:
    """HIP-event stopwatch on the estimator stream + accumulated match-kernel time."""

    def __init__(self):
        self.ctx = context()
        self.L = hip_lib()

    def start(self):
        self.L.lrr_icp_timer_start(self.ctx)

    def stop(self) -> float:
        ms = C.c_double(0.0)
        if self.L.lrr_icp_timer_stop(self.ctx, C.byref(ms)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return ms.value

    def match_profile(self, enable=True, reset=False):
        ms, n = C.c_double(0.0), C.c_int64(0)
        if self.L.lrr_icp_match_profile(self.ctx, int(enable), C.byref(ms), C.byref(n), int(reset)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return ms.value, n.value

    def match_samples(self, cap=1 << 16):
        """elapsed ms of every timed K1 submission since the last reset of match_profile"""
        buf = np.zeros(cap)
        n = C.c_int64(0)
        self.L.lrr_icp_match_profile_samples.argtypes = [C.c_void_p, _dp, C.c_int64, _ip]
        if self.L.lrr_icp_match_profile_samples(self.ctx, _d(buf), cap, C.byref(n)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return buf[:min(cap, n.value)].copy()

    K1_PATHS = ("cert_inline", "cert", "ball", "ball_growth", "tile", "coarse_pair", "direct")

    def k1_path_counts(self, reset=False):
        """searches (directions) served by each state of the grid matcher since the last reset"""
        v = (C.c_int64 * 7)()
        self.L.lrr_icp_k1_path_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        if self.L.lrr_icp_k1_path_counts(self.ctx, v, int(reset)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return dict(zip(self.K1_PATHS, [int(x) for x in v]))

    def event_overhead(self, pairs=64):
        """mean elapsed ms of an empty HIP-event pair on the context's stream"""
        ms = C.c_double(0.0)
        if self.L.lrr_icp_event_overhead(self.ctx, int(pairs), C.byref(ms)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return ms.value

    def synchronize(self):
        self.L.lrr_icp_synchronize(self.ctx)

    def cert_stats(self, reset=False):
        """(sources seen by the K1 certificate kernel, sources it had to search) since the last reset"""
        q, f = C.c_int64(0), C.c_int64(0)
        if self.L.lrr_icp_cert_stats(self.ctx, C.byref(q), C.byref(f), int(reset)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return q.value, f.value


class DeviceCloud:
    """A feature cloud in HBM (tests of the stand-alone nearest-neighbour entry point)."""

    def __init__(self, fs, float_strength=False):
        self.L = hip_lib()
        self.ctx = context()
        keep = []
        f = _features(fs, keep)
        h = C.c_void_p()
        if self.L.lrr_icp_cloud_create(self.ctx, f.n, f.loc, f.proj, f.strength, int(float_strength), C.byref(h)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        self.h = h

    def set_covariances(self, cov9):
        c = np.ascontiguousarray(cov9, dtype=np.float64)
        if self.L.lrr_icp_cloud_set_covariances(self.h, _d(c)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())

    def nearest(self, query, method=NN_GRID):
        q = np.ascontiguousarray(query, dtype=np.float64)
        out = np.empty(q.shape[0], dtype=np.int64)
        if self.L.lrr_icp_nearest(self.h, _d(q), q.shape[0], method, out.ctypes.data_as(_ip)) != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.ctx).decode())
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.lrr_icp_cloud_destroy(self.h)
            self.h = None

    __del__ = close


def cdc_compute_weights(moving: "DeviceCloud", fixed: "DeviceCloud", dof, A, t, c, param_covar, seed, max_matches=100):
    """cdcl_estimation::compute_weights at one resolution on the device (lrr_icp_cdc_compute_weights); the clouds need
    set_covariances().  Returns (row_count, to_idx (nm, max_matches), w, overflow_rows)."""
    L = hip_lib()
    x = Xform()
    x.A[:] = np.asarray(A, dtype=np.float64).ravel()
    x.t[:] = np.asarray(t, dtype=np.float64)
    x.c[:] = np.asarray(c, dtype=np.float64)
    cov = np.ascontiguousarray(param_covar, dtype=np.float64)
    st = C.c_uint64(seed)
    keys = (C.c_uint32 * 4)()
    L.lrr_icp_perm_keys(C.byref(st), keys)
    nm = L.lrr_icp_cloud_size(moving.h)
    rc = np.zeros(nm, dtype=np.int64)
    ti = np.zeros((nm, max_matches), dtype=np.int64)
    w = np.zeros((nm, max_matches))
    over = C.c_int64(0)
    L.lrr_icp_cdc_compute_weights.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Xform), C.c_int, _dp, C.POINTER(C.c_uint32), C.c_int,
                                              _ip, _ip, _dp, _ip]
    if L.lrr_icp_cdc_compute_weights(moving.h, fixed.h, C.byref(x), dof, _d(cov), keys, max_matches, rc.ctypes.data_as(_ip),
                                     ti.ctypes.data_as(_ip), _d(w), C.byref(over)) != 0:
        raise RuntimeError(L.lrr_icp_last_error(moving.ctx).decode())
    return rc, ti, w, over.value


class CdcSet:
    """Match rows of the covariance-driven EM estimator, resident on the device: compute_weights() = the E-step
    (cdcl_estimation::compute_weights), objective_par() = cdcl_obj_fun_par::compute over those rows (the M-step's
    parameter objective and gradient; the L-BFGS driver stays with the caller)."""

    def __init__(self, moving: "DeviceCloud", fixed: "DeviceCloud", max_matches=100):
        self.L = hip_lib()
        self.moving, self.fixed, self.max_matches = moving, fixed, int(max_matches)
        self.h = C.c_void_p()
        self.L.lrr_icp_cdc_set_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        self.L.lrr_icp_cdc_set_destroy.argtypes = [C.c_void_p]
        self.L.lrr_icp_cdc_set_destroy.restype = None
        self.L.lrr_icp_cdc_set_compute_weights.argtypes = [C.c_void_p, C.POINTER(Xform), C.c_int, _dp, C.POINTER(C.c_uint32), _ip]
        self.L.lrr_icp_cdc_set_get.argtypes = [C.c_void_p, _ip, _ip, _dp]
        self.L.lrr_icp_cdc_objective_par.argtypes = [C.c_void_p, C.POINTER(Xform), C.c_int, _dp, C.POINTER(Norm), _dp, _dp, _ip]
        self._ck(self.L.lrr_icp_cdc_set_create(moving.h, fixed.h, self.max_matches, C.byref(self.h)))

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.lrr_icp_last_error(self.moving.ctx).decode())

    @staticmethod
    def _xform(A, t, c):
        x = Xform()
        x.A[:] = np.asarray(A, dtype=np.float64).ravel()
        x.t[:] = np.asarray(t, dtype=np.float64)
        x.c[:] = np.asarray(c, dtype=np.float64)
        return x

    def compute_weights(self, dof, A, t, c, param_covar, seed):
        x = self._xform(A, t, c)
        cov = np.ascontiguousarray(param_covar, dtype=np.float64)
        st = C.c_uint64(seed)
        keys = (C.c_uint32 * 4)()
        self.L.lrr_icp_perm_keys(C.byref(st), keys)
        over = C.c_int64(0)
        self._ck(self.L.lrr_icp_cdc_set_compute_weights(self.h, C.byref(x), dof, _d(cov), keys, C.byref(over)))
        return over.value

    def rows(self):
        nm = self.L.lrr_icp_cloud_size(self.moving.h)
        rc = np.zeros(nm, dtype=np.int64)
        ti = np.zeros((nm, self.max_matches), dtype=np.int64)
        w = np.zeros((nm, self.max_matches))
        self._ck(self.L.lrr_icp_cdc_set_get(self.h, rc.ctypes.data_as(_ip), ti.ctypes.data_as(_ip), _d(w)))
        return rc, ti, w

    def objective_par(self, dof, A, t, c, param_covar, norm=None, want_grad=True):
        """(f, g or None, rows that took part); norm = (cm[3], rm, cf[3], rf) or None"""
        x = self._xform(A, t, c)
        cov = np.ascontiguousarray(param_covar, dtype=np.float64)
        nrm = None
        if norm is not None:
            nrm = Norm()
            v = np.asarray(norm, dtype=np.float64)
            nrm.cm[:] = v[0:3]; nrm.rm = v[3]; nrm.cf[:] = v[4:7]; nrm.rf = v[7]
        f = C.c_double(0.0)
        g = np.zeros(dof)
        rows = C.c_int64(0)
        self._ck(self.L.lrr_icp_cdc_objective_par(self.h, C.byref(x), dof, _d(cov), C.byref(nrm) if nrm is not None else None, C.byref(f),
                                                  _d(g) if want_grad else None, C.byref(rows)))
        return f.value, (g if want_grad else None), rows.value

    def normalize_stats(self):
        """cdcl_normalize_matches over the rows: (cm[3], rm, cf[3], rf) as one array of 8"""
        nrm = Norm()
        self.L.lrr_icp_cdc_normalize_stats.argtypes = [C.c_void_p, C.POINTER(Norm)]
        self._ck(self.L.lrr_icp_cdc_normalize_stats(self.h, C.byref(nrm)))
        return np.r_[np.array(nrm.cm[:]), nrm.rm, np.array(nrm.cf[:]), nrm.rf]

    def objective_cov(self, dof, A, t, c, x_upper, norm=None, want_grad=True):
        """cdcl_obj_fun_cov::compute: (f, g or None, pairs); x_upper = upper triangle of the Cholesky factor, row by row"""
        x = self._xform(A, t, c)
        xu = np.ascontiguousarray(x_upper, dtype=np.float64)
        nrm = None
        if norm is not None:
            nrm = Norm()
            v = np.asarray(norm, dtype=np.float64)
            nrm.cm[:] = v[0:3]; nrm.rm = v[3]; nrm.cf[:] = v[4:7]; nrm.rf = v[7]
        f = C.c_double(0.0)
        g = np.zeros(dof * (dof + 1) // 2)
        pairs = C.c_int64(0)
        self.L.lrr_icp_cdc_objective_cov.argtypes = [C.c_void_p, C.POINTER(Xform), C.c_int, _dp, C.POINTER(Norm), _dp, _dp, _ip]
        self._ck(self.L.lrr_icp_cdc_objective_cov(self.h, C.byref(x), dof, _d(xu), C.byref(nrm) if nrm is not None else None, C.byref(f),
                                                  _d(g) if want_grad else None, C.byref(pairs)))
        return f.value, (g if want_grad else None), pairs.value

    def close(self):
        if self.h:
            self.L.lrr_icp_cdc_set_destroy(self.h)
            self.h = C.c_void_p()


class CdcEm:
    """cdcl_estimation's EM loop (include/lrr_amd/cdcl_estimation.h: lrr_amd::cdcl_estimation_em_t<dof>): one_iteration() =
    compute_weights, from iteration 6 on estimate_parameters + compute_weights, estimate_covariance
    (cdcl/cdcl_estimation.txx:95-187).  dof 12: the affine transform; dof 6: the rigid one with its small-angle increment.
    moving / fixed: one feature set, or a list of them ([0] the finest resolution, with `spacing` = the fixed sets' point
    spacings): the estimator starts at the coarsest and switches resolution as compute_weights decides (:491-517)."""

    def __init__(self, moving, fixed, A, t, c, covar, seed=1, max_matches=100, dof=12, spacing=None):
        L = est_lib()
        self.L = L
        self._keep = []
        mv = list(moving) if isinstance(moving, (list, tuple)) else [moving]
        fx = list(fixed) if isinstance(fixed, (list, tuple)) else [fixed]
        if len(mv) != len(fx) or (len(mv) > 1 and (spacing is None or len(spacing) != len(mv))):
            raise ValueError("one fixed set and one spacing per moving set")
        levels = len(mv)
        fms = [_features(m, self._keep) for m in mv]
        ffs = [_features(f, self._keep) for f in fx]
        FP = C.POINTER(_Features)
        pm = (FP * levels)(*[C.pointer(f) for f in fms])
        pf = (FP * levels)(*[C.pointer(f) for f in ffs])
        sp = np.ascontiguousarray(spacing if spacing is not None else [1.0] * levels, dtype=np.float64)
        A, t, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (A, t, c))
        cv = np.ascontiguousarray(covar, dtype=np.float64)
        if cv.size != dof * dof:
            raise ValueError("covar must be dof x dof")
        self.dof, self.sizes, self.max_matches = int(dof), [m.n for m in mv], int(max_matches)
        self.resolution, self.matches_size, self.fell_back = levels - 1, 0, False
        self.h = C.c_void_p()
        L.lrr_cdc_em_create_pyramid.argtypes = [C.c_int, C.c_int, C.POINTER(FP), C.POINTER(FP), _dp, _dp, _dp, _dp, _dp, C.c_uint64, C.c_int,
                                                C.POINTER(C.c_void_p)]
        L.lrr_cdc_em_destroy.argtypes = [C.c_void_p]
        L.lrr_cdc_em_destroy.restype = None
        L.lrr_cdc_em_one_iteration_state.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_int), _dp, _dp]
        L.lrr_cdc_em_get_transform.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.lrr_cdc_em_rows.argtypes = [C.c_void_p, _ip, _ip, _dp]
        L.lrr_cdc_em_normalize_matches.argtypes = [C.c_void_p, C.c_int]
        L.lrr_cdc_em_initialize.argtypes = [C.c_void_p, _dp]
        L.lrr_cdc_em_run.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint)]
        L.lrr_cdc_em_resolution.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._ck(L.lrr_cdc_em_create_pyramid(self.dof, levels, pm, pf, _d(sp), _d(A), _d(t), _d(c), _d(cv), seed, self.max_matches, C.byref(self.h)))

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.lrr_est_last_error().decode())

    def one_iteration(self, iteration):
        """(converged, info[8]: par f start/end, cov f start/end, optimiser codes, evaluations); afterwards .resolution,
        .matches_size (matches_.size()) and .fell_back (find_closest_euclidean() supplied the matches)"""
        conv = C.c_int(0)
        info, state = np.zeros(8), np.zeros(6)
        self._ck(self.L.lrr_cdc_em_one_iteration_state(self.h, int(iteration), C.byref(conv), _d(info), _d(state)))
        self.resolution, self.matches_size, self.fell_back, self.avg_largest_eval = int(state[0]), int(state[1]), bool(state[2]), float(state[3])
        self.switches = (int(state[4]), int(state[5]))  # to a finer, to a coarser resolution, this iteration
        return bool(conv.value), info

    def normalize_matches(self, on=True):
        """normalize_matches() of the base class (cdcl/cdcl_estimation_ICP.h:76); on unless switched off here"""
        self._ck(self.L.lrr_cdc_em_normalize_matches(self.h, int(bool(on))))

    def initialize(self):
        """cdcl_estimation::initialize (cdcl/cdcl_estimation.txx:70-84): closest points of all moving features, the robust
        point-to-plane fit of the base class, its inverse normal matrix as the parameter covariance; returns the scale (2 x MUSE)"""
        scale = np.zeros(1)
        self._ck(self.L.lrr_cdc_em_initialize(self.h, _d(scale)))
        return float(scale[0])

    def run(self):
        """cdcl_estimation::run (cdcl/cdcl_estimation.txx:190-204): (converged, iterations done)"""
        conv, its, res = C.c_int(0), C.c_uint(0), C.c_int(0)
        self._ck(self.L.lrr_cdc_em_run(self.h, C.byref(conv), C.byref(its)))
        self._ck(self.L.lrr_cdc_em_resolution(self.h, C.byref(res)))
        self.resolution = int(res.value)
        return bool(conv.value), int(its.value)

    def get_transform(self):
        A, t, c, cov = np.zeros((3, 3)), np.zeros(3), np.zeros(3), np.zeros((self.dof, self.dof))
        self._ck(self.L.lrr_cdc_em_get_transform(self.h, _d(A), _d(t), _d(c), _d(cov)))
        return A, t, c, cov

    def rows(self):
        """the match rows of the current resolution"""
        nm = self.sizes[self.resolution]
        rc = np.zeros(nm, dtype=np.int64)
        ti = np.zeros((nm, self.max_matches), dtype=np.int64)
        w = np.zeros((nm, self.max_matches))
        self._ck(self.L.lrr_cdc_em_rows(self.h, rc.ctypes.data_as(_ip), ti.ctypes.data_as(_ip), _d(w)))
        return rc, ti, w

    def close(self):
        if self.h:
            self.L.lrr_cdc_em_destroy(self.h)
            self.h = C.c_void_p()


def subset_perm(n: int, seed: int):
    """The keyed subset permutation the match kernel evaluates (host copy, for tests)."""
    L = hip_lib()
    st = C.c_uint64(seed)
    keys = (C.c_uint32 * 4)()
    L.lrr_icp_perm_keys(C.byref(st), keys)
    return np.array([L.lrr_icp_subset_perm(i, n, keys) for i in range(n)], dtype=np.int64), st.value


class Estimator:
    """The reference estimator classes' public methods, running on the GPU.

    kind: CDCL_FORWARD / RRL_FORWARD -> cdcl_/rrl_estimation_ICP_matching_all<3,dof>,
          CDCL_SYMMETRIC / RRL_SYMMETRIC -> cdcl_/rrl_estimation_symmetric_ICP_matching_all<3,dof>.
    """

    def __init__(self, kind, dof, moving, fixed, A, t, c, moving_inside=None, fixed_inside=None,
                 seed=1, nn_method=NN_AUTO):
        self.kind, self.dof = kind, dof
        L = est_lib()
        keep = []
        fm, ff = _features(moving, keep), _features(fixed, keep)
        fmi, ffi = _features(moving_inside, keep), _features(fixed_inside, keep)
        A = np.ascontiguousarray(A, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        self._h = L.lrr_est_create(kind, dof, C.byref(fm), C.byref(ff),
                                   C.byref(fmi) if fmi is not None else None,
                                   C.byref(ffi) if ffi is not None else None,
                                   _d(A), _d(t), _d(c), seed, nn_method)
        if not self._h:
            raise RuntimeError("lrr_est_create: " + L.lrr_est_last_error().decode())

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(est_lib().lrr_est_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            est_lib().lrr_est_destroy(self._h)
            self._h = None

    __del__ = close

    def set_transform(self, A, t, c):
        A, t, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (A, t, c))
        self._ck(est_lib().lrr_est_set_transform(self._h, _d(A), _d(t), _d(c)))

    def set_features(self, moving_inside, fixed_inside=None):
        keep = []
        a, b = _features(moving_inside, keep), _features(fixed_inside, keep)
        self._ck(est_lib().lrr_est_set_features(self._h, C.byref(a), C.byref(b) if b is not None else None))

    def weight_by_strength(self, flag=True):
        self._ck(est_lib().lrr_est_weight_by_strength(self._h, int(flag)))

    def normalize_matches(self):
        self._ck(est_lib().lrr_est_normalize_matches(self._h))

    def set_number_points(self, n):
        self._ck(est_lib().lrr_est_set_number_points(self._h, int(n)))

    def inject_initial_scale(self, fwd, bwd=-1.0):
        self._ck(est_lib().lrr_est_inject_initial_scale(self._h, float(fwd), float(bwd)))

    def set_subset_order(self, direction, order):
        o = np.ascontiguousarray(order, dtype=np.int64)
        self._ck(est_lib().lrr_est_set_subset_order(self._h, direction, o.ctypes.data_as(_ip), o.shape[0]))

    def set_nn_method(self, method):
        self._ck(est_lib().lrr_est_set_nn_method(self._h, int(method)))

    def set_voronoi_maps(self, lookup=None):
        """The class's set_voronoi_maps() / set_fixed_voronoi_map() as the reference's caller issues it (the images are never
        read).  lookup: None = the process default (the maps' lookup semantics unless LRR_ICP_VORONOI=0), True / False =
        lrr_amd::voronoi_lookup::enable() first."""
        self._ck(est_lib().lrr_est_set_voronoi_maps(self._h, -1 if lookup is None else int(bool(lookup))))

    def find_closest_euclidean(self):
        self._ck(est_lib().lrr_est_find_closest_euclidean(self._h))

    def one_iteration(self, iteration=0):
        conv = C.c_int(0)
        self._ck(est_lib().lrr_est_one_iteration(self._h, iteration, C.byref(conv)))
        return bool(conv.value)

    def iterate(self, first, count):
        """`count` iterations numbered from `first` in one call; returns what the last one reported"""
        conv = C.c_int(0)
        self._ck(est_lib().lrr_est_iterate(self._h, int(first), int(count), C.byref(conv)))
        return bool(conv.value)

    def run(self, max_iterations=30):
        n, conv = C.c_uint(0), C.c_int(0)
        self._ck(est_lib().lrr_est_run(self._h, max_iterations, C.byref(n), C.byref(conv)))
        return bool(conv.value), int(n.value)

    def run_counted(self, max_iterations=70, check_valid=False):
        """run() + the number of matched pairs formed over its iterations; check_valid: the caller's
        `if (!validTransform(...)) break;` after every iteration (rrl/itkLocationRegistration.cxx:2482) — whether it
        ended the loop is kept in self.last_valid"""
        n, conv, pairs, valid = C.c_uint(0), C.c_int(0), C.c_int64(0), C.c_int(1)
        L = est_lib()
        L.lrr_est_run_caller.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_int), _ip, C.POINTER(C.c_int)]
        self._ck(L.lrr_est_run_caller(self._h, max_iterations, int(check_valid), C.byref(n), C.byref(conv), C.byref(pairs), C.byref(valid)))
        self.last_valid = bool(valid.value)
        return bool(conv.value), int(n.value), int(pairs.value)

    def set_subset_seed(self, seed):
        self._ck(est_lib().lrr_est_set_subset_seed(self._h, int(seed)))

    def estimate_LS(self, estimate_parameters=True):
        self._ck(est_lib().lrr_est_estimate_LS(self._h, int(estimate_parameters)))

    def estimate_LS_backward(self, estimate_parameters=False):
        self._ck(est_lib().lrr_est_estimate_LS_backward(self._h, int(estimate_parameters)))

    def get_transform(self, direction=0):
        A, t, c = np.empty((3, 3)), np.empty(3), np.empty(3)
        cov = np.empty((self.dof, self.dof))
        self._ck(est_lib().lrr_est_get_transform(self._h, direction, _d(A), _d(t), _d(c), _d(cov)))
        return A, t, c, cov

    def _scalar(self, fn, *args):
        out = C.c_double(0.0)
        self._ck(fn(self._h, *args, C.byref(out)))
        return out.value

    def RMS_error(self):
        return self._scalar(est_lib().lrr_est_RMS_error)

    def median_error(self):
        return self._scalar(est_lib().lrr_est_median_error)

    def weighted_error(self, direction=0):
        return self._scalar(est_lib().lrr_est_weighted_error, direction)

    def sheet_angles(self, direction=0):
        return self._scalar(est_lib().lrr_est_sheet_angles, direction)

    def tube_angles(self, direction=0):
        return self._scalar(est_lib().lrr_est_tube_angles, direction)

    def last_scale(self, direction=0):
        return self._scalar(est_lib().lrr_est_last_scale, direction)

    def compute_transfer_error_covariance(self):
        a, b = np.zeros((3, 3)), np.zeros((3, 3))
        self._ck(est_lib().lrr_est_compute_transfer_error_covariance(self._h, _d(a), _d(b)))
        return a, b

    def oscillated(self):
        return bool(est_lib().lrr_est_oscillated(self._h))

    def finest_level(self):
        return bool(est_lib().lrr_est_finest_level(self._h))

    def matches(self, direction=0):
        n = est_lib().lrr_est_num_matches(self._h, direction)
        f, t = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
        w = np.empty(n, dtype=np.float64)
        self._ck(est_lib().lrr_est_get_matches(self._h, direction, f.ctypes.data_as(_ip), t.ctypes.data_as(_ip), _d(w)))
        return f, t, w

    def number_points(self, direction=0):
        return est_lib().lrr_est_get_number_points(self._h, direction)

    def debug_set_match(self, direction, from_idx, value):
        """test hook (lrr_icp_debug_set_match): the stored match of from-feature from_idx becomes `value` as it is"""
        L, H = est_lib(), hip_lib()
        L.lrr_est_device_dir.restype = C.c_void_p
        L.lrr_est_device_dir.argtypes = [C.c_void_p, C.c_int]
        H.lrr_icp_debug_set_match.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        if H.lrr_icp_debug_set_match(L.lrr_est_device_dir(self._h, direction), int(from_idx), int(value)) != 0:
            raise RuntimeError("lrr_icp_debug_set_match failed")

    def normal_equations(self):
        AtA, Atb = np.empty((self.dof, self.dof)), np.empty(self.dof)
        self._ck(est_lib().lrr_est_last_normal_equations(self._h, _d(AtA), _d(Atb)))
        return AtA, Atb


class LocationBatch:
    """include/lrr_amd/location_batch.h: the two volumes resident on the device, many query locations
    (ROI boxes + initial transforms) registered per call, every device phase one submission for all."""

    # row layout of lrr_est_batch_run (lrr_est_capi.cpp)
    COL = dict(converged=0, iterations=1, A=slice(2, 11), t=slice(11, 14), c=slice(14, 17), weighted_error=17, rms=18, index=19,
               enough_features=20, oscillated=21, valid=22, n_moving_inside=23, n_fixed_inside=24, matched_pairs=25,
               initial_rms=26, weighted_error_backward=27, sheet_angles=28, tube_angles=29, sheet_angles_backward=30,
               tube_angles_backward=31, measurements=slice(32, 40), backward=slice(40, 55), covariance=slice(55, 199),
               covariance_backward=slice(199, 343))

    def __init__(self, moving, fixed):
        L = est_lib()
        keep = []
        fm, ff = _features(moving, keep), _features(fixed, keep)
        self._h = L.lrr_est_batch_create(C.byref(fm), C.byref(ff))
        if not self._h:
            raise RuntimeError("lrr_est_batch_create: " + L.lrr_est_last_error().decode())
        self.width = L.lrr_est_batch_row_width()

    def close(self):
        if getattr(self, "_h", None):
            est_lib().lrr_est_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def set_nn_method(self, method):
        """NN_AUTO / NN_GRID: exact search (default); NN_VORONOI: the lookup semantics of the 1 mm Voronoi maps"""
        L = est_lib()
        L.lrr_est_batch_set_matcher.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if L.lrr_est_batch_set_matcher(self._h, int(method), -1) != 0:
            raise RuntimeError(L.lrr_est_last_error().decode())

    def set_voronoi_maps(self, lookup=None):
        """rrl_location_batch::set_voronoi_maps() as the caller of the reference issues it per estimator (the images are never
        read); lookup: None = the process default (the maps' semantics unless LRR_ICP_VORONOI=0), True / False =
        lrr_amd::voronoi_lookup::enable() first"""
        L = est_lib()
        L.lrr_est_batch_set_matcher.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if L.lrr_est_batch_set_matcher(self._h, -1, -1 if lookup is None else int(bool(lookup))) != 0:
            raise RuntimeError(L.lrr_est_last_error().decode())

    def run(self, A0, t0, c0, moving_box, fixed_box, seeds, max_iterations=70, diagnostics=True, check_valid=True):
        """A0 (B,3,3), t0 (B,3), c0 (B,3): initial transforms; *_box (B,6): x0, x1 of the ROIs; seeds (B,).
        Returns one row of `width` doubles per location (COL)."""
        A0 = np.ascontiguousarray(A0, dtype=np.float64).reshape(-1, 9)
        B = A0.shape[0]
        t0 = np.ascontiguousarray(t0, dtype=np.float64).reshape(B, 3)
        c0 = np.ascontiguousarray(c0, dtype=np.float64).reshape(B, 3)
        mb = np.ascontiguousarray(moving_box, dtype=np.float64).reshape(B, 6)
        fb = np.ascontiguousarray(fixed_box, dtype=np.float64).reshape(B, 6)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(B)
        rows = np.zeros((B, self.width))
        if est_lib().lrr_est_batch_run(self._h, B, _d(A0), _d(t0), _d(c0), _d(mb), _d(fb),
                                       sd.ctypes.data_as(C.POINTER(C.c_uint64)), int(max_iterations), int(diagnostics),
                                       int(check_valid), _d(rows)) != 0:
            raise RuntimeError(est_lib().lrr_est_last_error().decode())
        return rows

    def last_timing(self):
        """wall-clock ms of the phases of the last run() inside the C++ class: {'roi': .., 'iterate': .., ...}"""
        buf = C.create_string_buffer(2048)
        L = est_lib()
        L.lrr_est_batch_last_timing.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.lrr_est_batch_last_timing(self._h, buf, 2048)
        out = {}
        for item in buf.value.decode().split(";"):
            if "=" in item:
                k, v = item.split("=")
                out[k] = out.get(k, 0.0) + float(v)
        return out

    def matches(self, b, direction=0):
        """matches_ of location b after run(): from = index in the ROI list, to = index in the full volume"""
        L, h = hip_lib(), est_lib().lrr_est_batch_handle(self._h)
        n = L.lrr_icp_batch_num_matches(h, b, direction)
        f, t = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
        w = np.empty(n, dtype=np.float64)
        if L.lrr_icp_batch_get_matches(h, b, direction, f.ctypes.data_as(_ip), t.ctypes.data_as(_ip), _d(w)) != 0:
            raise RuntimeError("lrr_icp_batch_get_matches failed")
        return f, t, w

    def roi(self, b, direction, n):
        L, h = hip_lib(), est_lib().lrr_est_batch_handle(self._h)
        idx = np.empty(n, dtype=np.int64)
        if L.lrr_icp_batch_get_roi(h, b, direction, idx.ctypes.data_as(_ip)) != 0:
            raise RuntimeError("lrr_icp_batch_get_roi failed")
        return idx
