"""ORACLE — restatement of the four dense-LAPACK helpers the GPRF path uses from the reference's
vendored GPy slice (``/root/reference/gpy_linalg.py``): jitchol :77-97, dpotrs :139-148,
dpotri :150-171 (+ symmetrify :410-483), pdinv :219-240, dtrtri :243-253.  Same LAPACK routines
(scipy.linalg.lapack), same order of operations, including the ``dtrtri`` whose result the path
never uses.  Test infrastructure only."""
import numpy as np
from scipy import linalg
from scipy.linalg import lapack


def jitchol(A, maxtries=5):
    """gpy_linalg.py:77-97.  Lower Cholesky; on failure retry on A + jitter*I with
    jitter = mean(diag)*1e-6 * 10^k, k = 0..maxtries-1.  NB (reference quirk):
    the factor of the *jittered* matrix is returned and the caller is not told."""
    A = np.ascontiguousarray(A)
    L, info = lapack.dpotrf(A, lower=1)
    if info == 0:
        return L
    diagA = np.diag(A)
    if np.any(diagA <= 0.):
        raise linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-6
    num_tries = 0
    while num_tries < maxtries and np.isfinite(jitter):
        try:
            return linalg.cholesky(A + np.eye(A.shape[0]) * jitter, lower=True)
        except Exception:
            jitter *= 10
        finally:
            num_tries += 1
    raise linalg.LinAlgError("not positive definite, even with jitter.")


def jitter_schedule(diag_mean, maxtries=5):
    """The jitter values jitchol tries, in order (used by the HIP wrapper's parity tests)."""
    return [diag_mean * 1e-6 * 10.0 ** k for k in range(maxtries)]


def symmetrify(A):
    """gpy_linalg.py:410-483 (default upper=False): copy the lower triangle onto the upper, in place."""
    il = np.tril_indices(A.shape[0], -1)
    A.T[il] = A[il]
    return A


def dtrtri(L):
    """gpy_linalg.py:243-253"""
    return lapack.dtrtri(np.asfortranarray(L), lower=1)[0]


def dpotri(L):
    """gpy_linalg.py:150-171: inverse from the lower Cholesky factor, then mirrored.  (The reference
    works around an old scipy argument bug by passing lower=0 on an F-ordered array; the routine it
    reaches is LAPACK DPOTRI on the lower factor, which is what is called here.)"""
    R, info = lapack.dpotri(np.asfortranarray(L), lower=1)
    symmetrify(R)
    return R, info


def dpotrs(L, B, lower=1):
    """gpy_linalg.py:139-148"""
    return lapack.dpotrs(np.asfortranarray(L), B, lower=lower)


def pdinv(A):
    """gpy_linalg.py:219-240 -> (Ai, L, Li, logdet).  Li is computed and never used by the path."""
    L = jitchol(A)
    logdet = 2. * np.sum(np.log(np.diag(L)))
    Li = dtrtri(L)
    Ai, _ = dpotri(L)
    symmetrify(Ai)
    return Ai, L, Li, logdet
