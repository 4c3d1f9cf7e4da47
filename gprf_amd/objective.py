"""The L-BFGS-B objective callback around the GPRF path — counterpart of
``gprfopt.do_optimization`` (gprfopt.py:320-432): ``lgpllgrad`` with the X prior (gprfopt.py:172-182),
the log-space covariance parametrisation scaled by ``cov_scale = 5`` and its near-uniform prior
(gprfopt.py:324-331, 365-368, 383, 403-407), ``full_cov`` / ``collapse_cov_grad`` (gprfopt.py:333-355)
and the ``log.txt`` line format (gprfopt.py:411-413).  Per-evaluation ``np.save`` checkpoints
(gprfopt.py:388,394) are optional."""
import os
import time

import numpy as np
import scipy.optimize


class OutOfTimeError(Exception):
    pass


def cov_prior(c):
    """gprfopt.py:324-331"""
    mean, std = -1, 10
    r = (c - mean) / std
    ll = -.5 * np.sum(r ** 2) - .5 * len(c) * np.log(2 * np.pi * std ** 2)
    return ll, -(c - mean) / (std ** 2)


class Objective(object):
    """``obj(x) -> (-ll, -grad)`` for ``scipy.optimize.minimize(..., jac=True)``."""

    cov_scale = 5.  # gprfopt.py:364

    def __init__(self, gprf, X0, C0, sdata, maxsec=None, log_dir=None, checkpoint=False, parallel=False):
        self.gprf, self.X0, self.C0, self.sdata = gprf, X0, C0, sdata
        self.gradX, self.gradC = (X0 is not None), (C0 is not None)
        x0 = X0.flatten() if self.gradX else np.array(())
        c0 = np.log(C0.flatten()) * self.cov_scale if self.gradC else np.array(())
        self.nx = len(x0)
        self.full0 = np.concatenate([x0, c0])
        self.maxsec, self.parallel = maxsec, parallel
        self.log_dir, self.checkpoint = log_dir, checkpoint
        self.f_log = open(os.path.join(log_dir, "log.txt"), "w") if log_dir else None
        self.step = 0
        self.t0 = time.time()
        self.trace = []  # (step, secs, ll)

    def full_cov(self, C):
        """gprfopt.py:333-345"""
        if C.shape[1] == 1:
            FC = np.empty((self.C0.shape[0], 2 + self.sdata.X_obs.shape[1]))
            FC[:, 0] = self.sdata.noise_var
            FC[:, 1] = 1.0
            FC[:, 2:3] = C
            FC[:, 3:4] = C
            return FC
        if C.shape[1] == 4:
            return C
        raise Exception("unrecognized cov param shape")

    def collapse_cov_grad(self, grad_FC):
        """gprfopt.py:347-355"""
        if self.C0.shape[1] == 1:
            return grad_FC[:, 2:3] + grad_FC[:, 3:4]
        if self.C0.shape[1] == 4:
            return grad_FC
        raise Exception("unrecognized cov param shape")

    def __call__(self, x):
        if self.maxsec is not None and time.time() - self.t0 > self.maxsec:
            raise OutOfTimeError
        xx = x[:self.nx]
        xc = x[self.nx:] / self.cov_scale
        if self.gradX:
            XX = xx.reshape(self.X0.shape)
            self.gprf.update_X(XX)
            if self.checkpoint and self.log_dir:
                np.save(os.path.join(self.log_dir, "step_%05d_X.npy" % self.step), XX)
        if self.gradC:
            C = np.exp(xc.reshape(self.C0.shape))
            FC = self.full_cov(C)
            self.gprf.update_covs(FC)
            if self.checkpoint and self.log_dir:
                np.save(os.path.join(self.log_dir, "step_%05d_cov.npy" % self.step), FC)
        ll, gX, gC = self.gprf.llgrad(local=True, grad_X=self.gradX, grad_cov=self.gradC, parallel=self.parallel)
        if self.gradX:
            prior_ll, prior_grad = self.sdata.x_prior(xx)
            ll += prior_ll
            gX = gX.flatten() + prior_grad
        if self.gradC:
            prior_ll, prior_grad = cov_prior(xc)
            ll += prior_ll
            gC = (np.array(self.collapse_cov_grad(gC)) * C).flatten() + prior_grad
            gC /= self.cov_scale
        grad = np.concatenate([gX.flatten(), gC.flatten()])
        secs = time.time() - self.t0
        self.trace.append((self.step, secs, ll))
        if self.f_log:
            self.f_log.write("%d %.2f %.2f\n" % (self.step, secs, ll))
            self.f_log.flush()
        self.step += 1
        return -ll, -grad

    def close(self):
        if self.f_log:
            self.f_log.write("optimization finished after %.fs\n" % (time.time() - self.t0))
            self.f_log.close()
            self.f_log = None


def do_optimization(gprf, X0, C0, sdata, method="l-bfgs-b", maxsec=3600, log_dir=None, maxiter=200, **kw):
    """gprfopt.py:320-432.  Returns (x_final or None if timed out, Objective)."""
    obj = Objective(gprf, X0, C0, sdata, maxsec=maxsec, log_dir=log_dir, **kw)
    rx = None
    try:
        r = scipy.optimize.minimize(obj, obj.full0, jac=True, method=method, bounds=None,
                                    options={"ftol": 1e-6, "maxiter": maxiter})
        rx = r.x
    except OutOfTimeError:
        pass
    obj.close()
    if log_dir:
        open(os.path.join(log_dir, "finished"), "w").close()
    return rx, obj
