"""Adapter between ``scipy.optimize`` and the library's objective entry point.

The reference wraps ``GPRF.llgrad`` in a closure before handing it to L-BFGS-B (gprfopt.py:320-432): it unpacks the
optimiser's flat vector, adds a Gaussian prior around the observed locations and a wide prior on the log
hyper-parameters, applies the chain rule and flips the signs.  Here that arithmetic lives in ``libgprf_hip.so``
(``gprf_objective``, include/gprf_hip.h): the location prior is added by the assembly kernel and the result comes down
already in the optimiser's layout, in the evaluation's one download.  What is left in Python is bookkeeping:

* ``VectorLayout``  — which slice of the flat vector is what, and the constants of the parametrisation;
* ``RunLog``        — the ``log.txt`` lines / optional ``.npy`` checkpoints of a run and the in-memory trace;
* ``Objective``     — the callable scipy sees; it routes to the library (a ``gprf_amd.GPRF``) or, for any other object
                      with the ``update_X / update_covs / llgrad`` surface, to the library's host-side prior functions.
"""
import os
import time

import numpy as np
import scipy.optimize

from . import _capi


class OutOfTimeError(Exception):
    pass


# the reference's constants: the optimiser sees log-parameters stretched by 5 (gprfopt.py:364) under a N(-1, 10^2)
# prior (gprfopt.py:325)
COV_SCALE = 5.0
HYPER_PRIOR = (-1.0, 10.0)


def cov_prior(c):
    """log density and gradient of the hyper-parameter prior at the log parameters ``c`` (gprfopt.py:324-331)"""
    c = np.atleast_1d(np.asarray(c, dtype=np.float64)).ravel()
    return _capi.hyper_grad(_capi.HYPER_FULL, 1.0, HYPER_PRIOR[0], HYPER_PRIOR[1], c, np.zeros_like(c))


class VectorLayout(object):
    """z = [locations, row-major | cov_scale * log(free hyper-parameters)].

    ``C0`` of shape (1, 1) frees one lengthscale shared by every input dimension, with the noise variance pinned to the
    data's and the signal variance to 1 (gprfopt.py:333-355); (1, 2 + n_lengthscales) frees everything."""

    def __init__(self, X0, C0, noise_var, ntheta):
        self.x_shape = None if X0 is None else tuple(np.shape(X0))
        self.nx = 0 if X0 is None else int(np.prod(self.x_shape))
        self.ntheta = ntheta
        self.noise_var = noise_var
        if C0 is None:
            self.mode, self.nh, self.c_shape = _capi.HYPER_NONE, 0, None
        else:
            C0 = np.asarray(C0, dtype=np.float64)
            self.c_shape = C0.shape
            if C0.shape[1] == 1:
                self.mode, self.nh = _capi.HYPER_TIED, 1
            elif C0.shape[1] == ntheta:
                self.mode, self.nh = _capi.HYPER_FULL, ntheta
            else:
                raise Exception("unrecognized cov param shape")
        self.start = np.concatenate([np.zeros(0) if X0 is None else np.asarray(X0, dtype=np.float64).ravel(),
                                     np.zeros(0) if C0 is None else COV_SCALE * np.log(C0.ravel())])

    def locations(self, z):
        return z[:self.nx].reshape(self.x_shape)

    def theta_row(self, z):
        """the (1, ntheta) row ``GPRF.update_covs`` takes"""
        return _capi.hyper_unpack(self.mode, COV_SCALE, self.noise_var, 1.0, self.ntheta, z[self.nx:]).reshape(1, -1)


class RunLog(object):
    """``log.txt`` ("step seconds objective", gprfopt.py:411-413), optional per-step checkpoints (gprfopt.py:388,394)."""

    def __init__(self, log_dir, checkpoint):
        self.dir, self.checkpoint = log_dir, bool(checkpoint and log_dir)
        self.fh = open(os.path.join(log_dir, "log.txt"), "w") if log_dir else None
        self.t0 = time.time()
        self.rows = []

    def elapsed(self):
        return time.time() - self.t0

    def record(self, ll):
        step, secs = len(self.rows), self.elapsed()
        self.rows.append((step, secs, ll))
        if self.fh:
            self.fh.write("%d %.2f %.2f\n" % (step, secs, ll))
            self.fh.flush()

    def save(self, what, array):
        if self.checkpoint:
            np.save(os.path.join(self.dir, "step_%05d_%s.npy" % (len(self.rows), what)), array)

    def close(self):
        if self.fh:
            self.fh.write("optimization finished after %.fs\n" % self.elapsed())
            self.fh.close()
            self.fh = None


class Objective(object):
    """``obj(z) -> (-log posterior, -gradient)`` for ``scipy.optimize.minimize(..., jac=True)``."""

    cov_scale = COV_SCALE

    def __init__(self, gprf, X0, C0, sdata, maxsec=None, log_dir=None, checkpoint=False, parallel=False):
        self.gprf, self.sdata, self.maxsec = gprf, sdata, maxsec
        # theta = [noise_var, signal_var, dfn_params...]: the MODEL says how many (euclidean: one lengthscale per input
        # dimension; lld: two, whatever dx — gprf.py:578 against :278); a foreign model with the
        # reference's surface is asked for its dfn_params, the synthetic drivers' 2 + dx is the last resort
        ctx = getattr(gprf, "_ctx", None)
        if ctx is not None and hasattr(ctx, "ncov"):
            ntheta = int(ctx.ncov)
        elif hasattr(getattr(gprf, "cov", None), "dfn_params"):
            ntheta = 2 + len(np.ravel(gprf.cov.dfn_params))
        else:
            ntheta = 2 + np.shape(sdata.X_obs)[1]
        self.layout = VectorLayout(X0, C0, sdata.noise_var, ntheta)
        self.log = RunLog(log_dir, checkpoint)
        self.parts = None                   # (GPRF terms, location prior, hyper prior) of the last call
        self._native = hasattr(gprf, "objective_call")
        if self._native:
            gprf.objective_setup(sdata.X_obs if self.layout.nx else None, sdata.obs_std, self.layout.mode, COV_SCALE,
                                 HYPER_PRIOR, sdata.noise_var, 1.0)

    # names the reference's callers (and this repository's tests) read
    full0 = property(lambda self: self.layout.start)
    nx = property(lambda self: self.layout.nx)
    trace = property(lambda self: self.log.rows)
    step = property(lambda self: len(self.log.rows))
    t0 = property(lambda self: self.log.t0)

    def __call__(self, z):
        if self.maxsec is not None and self.log.elapsed() > self.maxsec:
            raise OutOfTimeError
        z = np.ascontiguousarray(z, dtype=np.float64)
        lay = self.layout
        if lay.nx:
            self.log.save("X", lay.locations(z))
        if lay.nh:
            self.log.save("cov", lay.theta_row(z))
        if self._native:
            f, grad, self.parts = self.gprf.objective_call(z, lay)
        else:
            f, grad, self.parts = self._through_llgrad(z)
        self.log.record(-f)
        return f, grad

    def _through_llgrad(self, z):
        """any model with the reference's surface: the model's llgrad, then the library's host-side prior terms"""
        lay, g = self.layout, self.gprf
        if lay.nx:
            g.update_X(lay.locations(z))
        if lay.nh:
            g.update_covs(lay.theta_row(z))
        ll, gX, gC = g.llgrad(local=True, grad_X=lay.nx > 0, grad_cov=lay.nh > 0)
        grad = np.empty(lay.nx + lay.nh)
        xp = hp = 0.0
        if lay.nx:
            xp, xg = _capi.x_prior(z[:lay.nx], self.sdata.X_obs, self.sdata.obs_std)
            grad[:lay.nx] = -(np.asarray(gX).ravel() + xg)
        if lay.nh:
            hp, hg = _capi.hyper_grad(lay.mode, COV_SCALE, HYPER_PRIOR[0], HYPER_PRIOR[1], z[lay.nx:], gC)
            grad[lay.nx:] = -hg
        return -(ll + xp + hp), grad, (ll, xp, hp)

    def close(self):
        self.log.close()


def do_optimization(gprf, X0, C0, sdata, method="l-bfgs-b", maxsec=3600, log_dir=None, maxiter=200, **kw):
    """The optimiser run of the reference's drivers (gprfopt.py:419-432: ftol 1e-6, no bounds; a ``finished`` marker in
    the run directory).  Returns (final vector, or None when the time budget ran out; the Objective with its trace)."""
    obj = Objective(gprf, X0, C0, sdata, maxsec=maxsec, log_dir=log_dir, **kw)
    z_final = None
    try:
        z_final = scipy.optimize.minimize(obj, obj.full0, jac=True, method=method, bounds=None,
                                          options={"ftol": 1e-6, "maxiter": maxiter}).x
    except OutOfTimeError:
        pass
    obj.close()
    if log_dir:
        open(os.path.join(log_dir, "finished"), "w").close()
    return z_final, obj
