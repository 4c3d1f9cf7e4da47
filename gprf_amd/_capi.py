"""ctypes binding of libgprf_hip.so (include/gprf_hip.h).  The library is built in-tree by
``gprf_amd.build``; there is no CPU fallback — if the HIP library cannot be loaded the import of the
product path fails loudly."""
import ctypes
import os

import numpy as np

from . import build as _build

GPRF_OK, GPRF_NOT_PD, GPRF_RETRY = 0, 1, 2
N_STAGES = 7
STAGE_NAMES = ("gather", "fill", "potrf", "solve", "at", "grad", "assemble")   # grad = k_mgrad + k_gx_finalize
DIST_IDS = {"euclidean": 0, "lld": 1}
KERN_IDS = {"se": 0, "matern32": 1}
MAX_UNIT = 16384
YPAD, XPAD = 64, 4

_dp = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
_i32 = ctypes.c_int32

# every entry point include/gprf_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "gprf_create": (ctypes.c_int, [ctypes.POINTER(_vp), _i32, _i32, _i32, _i32, _i32, _i32]),
    "gprf_create_multi": (ctypes.c_int, [ctypes.POINTER(_vp), _i32, _i32, _i32, _i32, _i32, _i32, _i32p]),
    "gprf_destroy": (ctypes.c_int, [_vp]),
    "gprf_last_error": (ctypes.c_char_p, [_vp]),
    "gprf_set_Y": (ctypes.c_int, [_vp, _dp]),
    "gprf_set_theta": (ctypes.c_int, [_vp, _dp, _i32]),
    "gprf_set_blocks": (ctypes.c_int, [_vp, _i32, _i64p, _i32p]),
    "gprf_set_neighbors": (ctypes.c_int, [_vp, _i32, _i32p]),
    "gprf_pair_kernel_max": (ctypes.c_int, [_vp, _dp, _i32, _i64p, _i32p, ctypes.c_double, _i32, _i32p, _i32p, _dp]),
    "gprf_nearest_center": (ctypes.c_int, [_i32, _i32, _dp, _i32, _dp, _i32p]),
    "gprf_set_block_assignment": (ctypes.c_int, [_vp, _i32, _i32p]),
    "gprf_set_centers": (ctypes.c_int, [_vp, _i32, _dp]),
    "gprf_assign_blocks": (ctypes.c_int, [_vp, _dp, _i32p, _i32p]),
    "gprf_get_block_assignment": (ctypes.c_int, [_vp, _i32p]),
    "gprf_set_split_tree": (ctypes.c_int, [_vp, _i32, _i32, _i32, _dp, _dp, _dp, _i32p, _i32p, _i32p]),
    "gprf_set_shard": (ctypes.c_int, [_vp, _i32, _i32]),
    "gprf_partition_units": (ctypes.c_int, [_i32, _i32p, _i32, _i32, _i32p]),
    "gprf_set_unit_jitter": (ctypes.c_int, [_vp, _i32, _dp]),
    "gprf_eval": (ctypes.c_int, [_vp, _dp, _i32, _i32, _dp, _dp, _dp, _i32p]),
    "gprf_update_eval": (ctypes.c_int, [_vp, _dp, _i32, _i32, _dp, _dp, _dp, _i32p, _i32p]),
    "gprf_eval_device": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "gprf_update_eval_device": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "gprf_eval_status": (ctypes.c_int, [_vp, _i32p]),
    "gprf_last_reblocked": (ctypes.c_int, [_vp, _i32p]),
    "gprf_num_units": (ctypes.c_int, [_vp, _i32p, _i32p]),
    "gprf_work_estimate": (ctypes.c_int, [_vp, _dp, _dp]),
    "gprf_table_builds": (ctypes.c_int, [_vp, _i32p]),
    "gprf_set_timing": (ctypes.c_int, [_vp, _i32]),
    "gprf_set_stream_pipelines": (ctypes.c_int, [_vp, _i32]),
    "gprf_get_timing": (ctypes.c_int, [_vp, _i32, _dp]),
    "gprf_set_x_prior": (ctypes.c_int, [_vp, _dp, ctypes.c_double]),
    "gprf_set_hyper_param": (ctypes.c_int, [_vp, _i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_double]),
    "gprf_objective": (ctypes.c_int, [_vp, _dp, _i32, _dp, _i32, _dp, _dp, _dp, _i32p, _i32p]),
    "gprf_objective_device": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _i32]),
    "gprf_x_prior": (ctypes.c_int, [ctypes.c_int64, _dp, _dp, ctypes.c_double, _dp, _dp]),
    "gprf_hyper_unpack": (ctypes.c_int, [_i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, _i32, _dp, _dp]),
    "gprf_hyper_grad": (ctypes.c_int, [_i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, _i32, _dp, _dp, _dp, _dp]),
    "gprf_build_flags": (ctypes.c_char_p, []),
    "gprf_runtime_config": (ctypes.c_char_p, []),
    "gprf_group_info": (ctypes.c_int, [_vp, _i32p, _i32p, _i32, _i32p, _i32p]),
    "gprf_debug_run": (ctypes.c_int, [_vp, _dp, _i32]),
    "gprf_debug_fetch": (ctypes.c_int, [_vp, _i32, _i32, _dp, ctypes.c_int64]),
    "gprf_debug_unit_shape": (ctypes.c_int, [_vp, _i32, _i32p, _i32p, _i32p]),
}

_lib = None


class GprfHipError(RuntimeError):
    pass


class NotPositiveDefinite(np.linalg.LinAlgError):
    """A unit's kernel matrix failed Cholesky (the reference raises LinAlgError from jitchol,
    gpy_linalg.py:85,97)."""

    def __init__(self, msg, unit):
        super().__init__(msg)
        self.unit = unit


def library_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building first if the .so is absent or stale and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if build_if_missing and _build._stale() and _build.have_compiler():
        _build.build()
    if not os.path.exists(path):
        raise GprfHipError("libgprf_hip.so is missing (%s); run `python -m gprf_amd.build`" % path)
    _preload_torch_hip_runtime()
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _preload_torch_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.  If this library
    pulled in /opt/rocm's copy first, a later ``import torch`` in the same process would find a second,
    mismatched runtime already bound to the SONAME and fail to see the GPU.  So when a torch wheel with a
    bundled runtime is installed, load that copy first (RTLD_GLOBAL); libgprf_hip.so then binds to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def build_flags():
    """diagnostic defines the loaded library was compiled with ("" = the product build)"""
    return load().gprf_build_flags().decode().split()


def runtime_config():
    """the launch structure the loaded library runs with in this environment (gprf_runtime_config) as a dict of strings"""
    return dict(kv.split("=", 1) for kv in load().gprf_runtime_config().decode().split())


def partition_units(m, dy, world):
    """owner rank of every unit (gprf_partition_units; host only, no GPU needed)."""
    m = np.ascontiguousarray(m, dtype=np.int32)
    owner = np.zeros(len(m), dtype=np.int32)
    rc = load().gprf_partition_units(len(m), m.ctypes.data_as(_i32p), int(dy), int(world), owner.ctypes.data_as(_i32p))
    if rc != GPRF_OK:
        raise GprfHipError("gprf_partition_units failed (%d)" % rc)
    return owner


def dptr(a):
    return a.ctypes.data_as(_dp)


def nearest_center(X, centers):
    """block id of every point (gprf_nearest_center; host only)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    C = np.ascontiguousarray(centers, dtype=np.float64)
    out = np.empty(X.shape[0], dtype=np.int32)
    rc = load().gprf_nearest_center(X.shape[0], X.shape[1], dptr(X), C.shape[0], dptr(C), out.ctypes.data_as(_i32p))
    if rc != GPRF_OK:
        raise GprfHipError("gprf_nearest_center failed (%d)" % rc)
    return out


HYPER_NONE, HYPER_TIED, HYPER_FULL = 0, 1, 2


def x_prior(x, x_obs, obs_std, want_grad=True):
    """Gaussian location prior N(x_obs, obs_std^2) (gprf_x_prior; host only): -> (log density, gradient or None)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    x_obs = np.ascontiguousarray(x_obs, dtype=np.float64).ravel()
    if x.shape != x_obs.shape:
        raise ValueError("x and x_obs differ in size")
    ll = ctypes.c_double(0.0)
    g = np.empty_like(x) if want_grad else None
    rc = load().gprf_x_prior(x.size, dptr(x), dptr(x_obs), float(obs_std), ctypes.byref(ll), dptr(g) if want_grad else None)
    if rc != GPRF_OK:
        raise GprfHipError("gprf_x_prior failed (%d)" % rc)
    return ll.value, g


def hyper_unpack(mode, cov_scale, fixed_nv, fixed_sv, ntheta, zh):
    """theta = [noise_var, signal_var, lengthscales...] from the optimiser's hyper variables (gprf_hyper_unpack)."""
    zh = np.ascontiguousarray(zh, dtype=np.float64).ravel()
    theta = np.empty(ntheta)
    rc = load().gprf_hyper_unpack(mode, float(cov_scale), float(fixed_nv), float(fixed_sv), ntheta, dptr(zh), dptr(theta))
    if rc != GPRF_OK:
        raise GprfHipError("gprf_hyper_unpack failed (%d)" % rc)
    return theta


def hyper_grad(mode, cov_scale, prior_mean, prior_std, zh, gradC):
    """-> (hyper prior log density, d(ll + prior)/d zh) from gradC = d ll / d theta (gprf_hyper_grad)."""
    zh = np.ascontiguousarray(zh, dtype=np.float64).ravel()
    gradC = np.ascontiguousarray(gradC, dtype=np.float64).ravel()
    pll = ctypes.c_double(0.0)
    g = np.empty_like(zh)
    rc = load().gprf_hyper_grad(mode, float(cov_scale), float(prior_mean), float(prior_std), gradC.size, dptr(zh), dptr(gradC),
                                ctypes.byref(pll), dptr(g))
    if rc != GPRF_OK:
        raise GprfHipError("gprf_hyper_grad failed (%d)" % rc)
    return pll.value, g


class Context(object):
    """Thin RAII wrapper over gprf_ctx*."""

    def __init__(self, n, dx, dy, dist_id, kern_id, device=0, devices=None):
        """``devices``: a list of HIP device ordinals -> one context over several devices, driven from this one process
        (gprf_create_multi); otherwise ``device``."""
        self.lib = load()
        self.h = _vp()
        if devices is not None:
            devs = np.ascontiguousarray(devices, dtype=np.int32)
            rc = self.lib.gprf_create_multi(ctypes.byref(self.h), n, dx, dy, dist_id, kern_id, len(devs), devs.ctypes.data_as(_i32p))
        else:
            rc = self.lib.gprf_create(ctypes.byref(self.h), n, dx, dy, dist_id, kern_id, device)
        if rc != GPRF_OK:
            self.h = None
            raise GprfHipError("gprf_create%s failed (%d): %s (device(s) %s, n=%d dx=%d dy=%d dist=%d kern=%d)" % (
                "_multi" if devices is not None else "", rc, self.lib.gprf_last_error(None).decode(),
                devices if devices is not None else device, n, dx, dy, dist_id, kern_id))
        self.n, self.dx, self.dy = n, dx, dy
        self.ncov = 2 + (dx if dist_id == 0 else 2)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gprf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise GprfHipError("%s failed (%d): %s" % (what, rc, self.lib.gprf_last_error(self.h).decode()))
        return rc

    def set_Y(self, Y):
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        assert Y.shape == (self.n, self.dy)
        self._check(self.lib.gprf_set_Y(self.h, dptr(Y)), "gprf_set_Y")

    def set_theta(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64).ravel()
        self._check(self.lib.gprf_set_theta(self.h, dptr(theta), len(theta)), "gprf_set_theta")

    def set_blocks(self, block_ptr, point_idx):
        block_ptr = np.ascontiguousarray(block_ptr, dtype=np.int64)
        point_idx = np.ascontiguousarray(point_idx, dtype=np.int32)
        self._check(self.lib.gprf_set_blocks(self.h, len(block_ptr) - 1, block_ptr.ctypes.data_as(_i64p),
                                             point_idx.ctypes.data_as(_i32p)), "gprf_set_blocks")

    def set_block_assignment(self, n_blocks, block_of):
        block_of = np.ascontiguousarray(block_of, dtype=np.int32)
        assert block_of.shape == (self.n,)
        self._check(self.lib.gprf_set_block_assignment(self.h, int(n_blocks), block_of.ctypes.data_as(_i32p)),
                    "gprf_set_block_assignment")

    def set_centers(self, centers):
        centers = np.ascontiguousarray(centers, dtype=np.float64)
        assert centers.ndim == 2 and centers.shape[1] == self.dx
        self._check(self.lib.gprf_set_centers(self.h, centers.shape[0], dptr(centers)), "gprf_set_centers")

    def set_split_tree(self, vec, center, split, left, right, leaf_block, lon_wrap):
        """Route gprf_assign_blocks through a binary split tree (see include/gprf_hip.h)."""
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        center = np.ascontiguousarray(center, dtype=np.float64)
        split = np.ascontiguousarray(split, dtype=np.float64)
        left, right, leaf_block = (np.ascontiguousarray(a, dtype=np.int32) for a in (left, right, leaf_block))
        n_nodes, dim = vec.shape
        assert center.shape == (n_nodes, dim) and split.shape == left.shape == right.shape == leaf_block.shape == (n_nodes,)
        self._check(self.lib.gprf_set_split_tree(self.h, n_nodes, dim, 1 if lon_wrap else 0, dptr(vec), dptr(center),
                                                 dptr(split), left.ctypes.data_as(_i32p), right.ctypes.data_as(_i32p),
                                                 leaf_block.ctypes.data_as(_i32p)), "gprf_set_split_tree")

    def assign_blocks(self, X):
        """Device re-blocking: -> (changed, block_of or None).  When ``changed`` the context has already installed
        the new partition."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.shape == (self.n, self.dx)
        changed = ctypes.c_int32(0)
        out = np.empty(self.n, dtype=np.int32)
        self._check(self.lib.gprf_assign_blocks(self.h, dptr(X), ctypes.byref(changed), out.ctypes.data_as(_i32p)),
                    "gprf_assign_blocks")
        return (True, out) if changed.value else (False, None)

    def pair_kernel_max(self, X, block_ptr, point_idx, threshold, cand, want_max=False):
        """-> keep flags (and the exact maxima if ``want_max``) for the candidate block pairs (gprf.py:119-150)"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        block_ptr = np.ascontiguousarray(block_ptr, dtype=np.int64)
        point_idx = np.ascontiguousarray(point_idx, dtype=np.int32)
        cand = np.ascontiguousarray(np.asarray(cand, dtype=np.int32).reshape(-1, 2))
        keep = np.zeros(len(cand), dtype=np.int32)
        mx = np.zeros(len(cand)) if want_max else None
        self._check(self.lib.gprf_pair_kernel_max(self.h, dptr(X), len(block_ptr) - 1, block_ptr.ctypes.data_as(_i64p),
                                                  point_idx.ctypes.data_as(_i32p), float(threshold), len(cand),
                                                  cand.ctypes.data_as(_i32p), keep.ctypes.data_as(_i32p),
                                                  dptr(mx) if want_max else None), "gprf_pair_kernel_max")
        return (keep, mx) if want_max else keep

    def set_neighbors(self, pairs):
        pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        self._check(self.lib.gprf_set_neighbors(self.h, pairs.shape[0], pairs.ctypes.data_as(_i32p)),
                    "gprf_set_neighbors")

    def set_shard(self, rank, world):
        self._check(self.lib.gprf_set_shard(self.h, rank, world), "gprf_set_shard")

    def set_unit_jitter(self, jitter):
        if jitter is None:
            self._check(self.lib.gprf_set_unit_jitter(self.h, 0, None), "gprf_set_unit_jitter")
        else:
            jitter = np.ascontiguousarray(jitter, dtype=np.float64)
            self._check(self.lib.gprf_set_unit_jitter(self.h, len(jitter), dptr(jitter)), "gprf_set_unit_jitter")

    def eval(self, X, want_gx, want_gc):
        """-> (rc, ll, gradX or None, gradC or None, first_bad_unit)"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.shape == (self.n, self.dx)
        ll = ctypes.c_double(0.0)
        gx = np.empty((self.n, self.dx)) if want_gx else None
        gc = np.empty((self.ncov,)) if want_gc else None
        bad = _i32(-1)
        rc = self.lib.gprf_eval(self.h, dptr(X), 1 if want_gx else 0, 1 if want_gc else 0, ctypes.byref(ll),
                                dptr(gx) if want_gx else None, dptr(gc) if want_gc else None, ctypes.byref(bad))
        self._check(rc, "gprf_eval")
        return rc, ll.value, gx, gc, bad.value

    def update_eval(self, X, want_gx, want_gc):
        """update_X + llgrad in one call (device re-blocking): -> (rc, ll, gradX, gradC, first_bad_unit, reblocked)"""
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.shape == (self.n, self.dx)
        ll = ctypes.c_double(0.0)
        gx = np.empty((self.n, self.dx)) if want_gx else None
        gc = np.empty((self.ncov,)) if want_gc else None
        bad, reb = _i32(-1), _i32(0)
        rc = self.lib.gprf_update_eval(self.h, dptr(X), 1 if want_gx else 0, 1 if want_gc else 0, ctypes.byref(ll),
                                       dptr(gx) if want_gx else None, dptr(gc) if want_gc else None, ctypes.byref(bad),
                                       ctypes.byref(reb))
        self._check(rc, "gprf_update_eval")
        return rc, ll.value, gx, gc, bad.value, bool(reb.value)

    def set_x_prior(self, X_obs, obs_std):
        if X_obs is None:
            self._check(self.lib.gprf_set_x_prior(self.h, None, 0.0), "gprf_set_x_prior")
            return
        X_obs = np.ascontiguousarray(X_obs, dtype=np.float64)
        assert X_obs.size == self.n * self.dx
        self._check(self.lib.gprf_set_x_prior(self.h, dptr(X_obs), float(obs_std)), "gprf_set_x_prior")

    def set_hyper_param(self, mode, cov_scale=1.0, prior_mean=0.0, prior_std=1.0, fixed_nv=0.0, fixed_sv=1.0):
        self._check(self.lib.gprf_set_hyper_param(self.h, mode, float(cov_scale), float(prior_mean), float(prior_std),
                                                  float(fixed_nv), float(fixed_sv)), "gprf_set_hyper_param")

    def objective(self, z, X_fixed=None, reblock=False):
        """One optimiser callback in the library (gprf_objective): -> (rc, f, grad, parts, first_bad_unit, reblocked)"""
        z = np.ascontiguousarray(z, dtype=np.float64).ravel()
        Xf = None if X_fixed is None else np.ascontiguousarray(X_fixed, dtype=np.float64)
        f = ctypes.c_double(0.0)
        g = np.empty_like(z)
        parts = np.zeros(3)
        bad, reb = _i32(-1), _i32(0)
        rc = self.lib.gprf_objective(self.h, dptr(z), z.size, dptr(Xf) if Xf is not None else None, 1 if reblock else 0,
                                     ctypes.byref(f), dptr(g), dptr(parts), ctypes.byref(bad), ctypes.byref(reb))
        self._check(rc, "gprf_objective")
        return rc, f.value, g, parts, bad.value, bool(reb.value)

    def objective_device(self, d_X_ptr, want_gx, want_gc, d_out_ptr, stream_ptr=None, reblock=False):
        rc = self.lib.gprf_objective_device(self.h, _vp(d_X_ptr), 1 if want_gx else 0, 1 if want_gc else 0, _vp(d_out_ptr),
                                            _vp(stream_ptr) if stream_ptr else None, 1 if reblock else 0)
        self._check(rc, "gprf_objective_device")
        return rc

    def get_block_assignment(self):
        out = np.empty(self.n, dtype=np.int32)
        self._check(self.lib.gprf_get_block_assignment(self.h, out.ctypes.data_as(_i32p)), "gprf_get_block_assignment")
        return out

    def table_builds(self):
        b = _i32(0)
        self._check(self.lib.gprf_table_builds(self.h, ctypes.byref(b)), "gprf_table_builds")
        return b.value

    def eval_device(self, d_X_ptr, want_gx, want_gc, d_out_ptr, stream_ptr=None, reblock=False):
        fn = self.lib.gprf_update_eval_device if reblock else self.lib.gprf_eval_device
        rc = fn(self.h, _vp(d_X_ptr), 1 if want_gx else 0, 1 if want_gc else 0,
                _vp(d_out_ptr), _vp(stream_ptr) if stream_ptr else None)
        self._check(rc, "gprf_update_eval_device" if reblock else "gprf_eval_device")
        return rc

    def eval_status(self):
        bad = _i32(-1)
        rc = self._check(self.lib.gprf_eval_status(self.h, ctypes.byref(bad)), "gprf_eval_status")
        return rc, bad.value

    def last_reblocked(self):
        r = _i32(0)
        self._check(self.lib.gprf_last_reblocked(self.h, ctypes.byref(r)), "gprf_last_reblocked")
        return bool(r.value)

    def group_info(self):
        """-> (members, slots_on_host, [device ordinal per member], [units per member]) of a multi-device group"""
        nm, oh = _i32(0), _i32(0)
        devs, units = np.zeros(64, dtype=np.int32), np.zeros(64, dtype=np.int32)
        self._check(self.lib.gprf_group_info(self.h, ctypes.byref(nm), ctypes.byref(oh), 64, devs.ctypes.data_as(_i32p),
                                             units.ctypes.data_as(_i32p)), "gprf_group_info")
        return nm.value, bool(oh.value), devs[:nm.value].tolist(), units[:nm.value].tolist()

    def num_units(self):
        a, b = _i32(0), _i32(0)
        self._check(self.lib.gprf_num_units(self.h, ctypes.byref(a), ctypes.byref(b)), "gprf_num_units")
        return a.value, b.value

    def work_estimate(self):
        f, b = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self.lib.gprf_work_estimate(self.h, ctypes.byref(f), ctypes.byref(b)), "gprf_work_estimate")
        return f.value, b.value

    def set_timing(self, on, reset=False):
        self._check(self.lib.gprf_set_timing(self.h, 2 if (on and reset) else (1 if on else 0)), "gprf_set_timing")

    def get_timing(self):
        """average ms per stage over the evaluations timed since the last reset, plus 'count'"""
        ms = np.zeros(2 * N_STAGES + 1)
        self._check(self.lib.gprf_get_timing(self.h, len(ms), dptr(ms)), "gprf_get_timing")
        d = dict(zip(STAGE_NAMES, ms[:N_STAGES].tolist()))
        d["count"] = int(ms[2 * N_STAGES])
        return d

    # ---- per-stage parity hooks (tests) ----
    def set_stream_pipelines(self, on):
        """the by-class pipelines on a caller's stream too (gprf_set_stream_pipelines): for one evaluation at a time"""
        self._check(self.lib.gprf_set_stream_pipelines(self.h, 1 if on else 0), "gprf_set_stream_pipelines")

    def debug_run(self, X, stop_after=6):
        X = np.ascontiguousarray(X, dtype=np.float64)
        self._check(self.lib.gprf_debug_run(self.h, dptr(X), stop_after), "gprf_debug_run")

    def debug_unit_shape(self, l):
        m, mp, g = _i32(0), _i32(0), _i32(0)
        self._check(self.lib.gprf_debug_unit_shape(self.h, l, ctypes.byref(m), ctypes.byref(mp), ctypes.byref(g)),
                    "gprf_debug_unit_shape")
        return m.value, mp.value, g.value

    def max_T(self):
        """16-row tiles per edge of the largest local unit (stride of the per-block partial pools)."""
        _, nl = self.num_units()
        return max([self.debug_unit_shape(l)[1] // 16 for l in range(nl)] + [0])

    def debug_fetch(self, l, what):
        m, mp, _ = self.debug_unit_shape(l)
        tbm = max((self.max_T() + 3) // 4, 1) if what in (7, 8) else 0
        shape = {0: (mp, mp), 1: (mp, mp), 2: (mp, YPAD), 3: (YPAD, mp), 4: (mp, XPAD), 5: (4,), 6: (8,),
                 7: (mp, tbm, XPAD), 8: (mp, tbm, XPAD), 9: (self.ncov,), 10: (mp,)}[what]
        out = np.zeros(shape)
        self._check(self.lib.gprf_debug_fetch(self.h, l, what, dptr(out), out.size), "gprf_debug_fetch")
        return out
