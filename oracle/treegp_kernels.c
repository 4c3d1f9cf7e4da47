/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under gprf_amd/ may link, import or call this.
 *
 * CPU restatement (plain C, scalar, fp64) of the arithmetic the reference obtains from the
 * un-vendored third-party extension  treegp.cover_tree.VectorTree  (treegp @
 * a0aa7ae65a4b9144a499016bbf0ccaf0c611cc0d, pinned by /root/reference/README.md:4).  treegp's
 * sources are not under /root/reference and there is no network, so the formulas below restate
 * its published algorithm from (i) the reference's own call sites and conventions and (ii)
 * recollection of treegp's cover_tree/vector_mult.cc; the SE/euclidean branch is PINNED by
 * reproducing the reference's published objective traces to every printed digit
 * (tests/test_oracle_kat.py).  The lld/matern32 branch is **parity unpinned**
 * (dataset and treegp both absent): it is anchored only on the in-tree haversine
 * (/root/reference/run_seismic.py:19-63, 230-233) and on finite differences.
 *
 * Reference call sites restated here:
 *   VectorTree.kernel_matrix(X1, X2, distance_only)      gprf.py:339, 342, 373
 *   VectorTree.kernel_deriv_wrt_xi_row(X, p, i, out)     gprf.py:353   (row q -> d k(x_p,x_q) / d x_p[i])
 *   VectorTree.kernel_deriv_wrt_i(X1, X2, i, 1, dists)   gprf.py:374   (d K / d dfn_params[i])
 *
 * dist_id: 0 = "euclidean" (scaled L2, dfn_params = one lengthscale per input dim)
 *          1 = "lld"       (lon deg, lat deg, depth km; dfn_params = [l_horiz_km, l_depth_km])
 * kern_id: 0 = "se"        k = sv * exp(-d^2)          (NO 1/2 factor: gprfopt.py:238-239)
 *          1 = "matern32"  k = sv * (1 + sqrt3 d) exp(-sqrt3 d)
 */
#include <math.h>
#include <stddef.h>

#define TG_EARTH_R_KM 6371.0 /* run_seismic.py:52 */
#define TG_DEG2RAD (M_PI / 180.0)

/* great-circle distance in km, haversine form of run_seismic.py:19-63 (np.radians -> sin/cos ->
 * 2*arcsin(sqrt(.)) -> degrees -> radians*R; the degrees/radians round trip is kept). */
static double tg_dist_km(const double *p1, const double *p2) {
    double rlon1 = p1[0] * TG_DEG2RAD, rlat1 = p1[1] * TG_DEG2RAD;
    double rlon2 = p2[0] * TG_DEG2RAD, rlat2 = p2[1] * TG_DEG2RAD;
    double s1 = sin((rlat1 - rlat2) / 2.0);
    double s2 = sin((rlon1 - rlon2) / 2.0);
    double a = s1 * s1 + cos(rlat1) * cos(rlat2) * s2 * s2;
    if (a > 1.0) a = 1.0;
    double dist_rad = 2.0 * asin(sqrt(a));
    double deg = dist_rad * (180.0 / M_PI);
    return (deg * TG_DEG2RAD) * TG_EARTH_R_KM;
}

/* d(great-circle km)/d(lon1 deg) and d/d(lat1 deg) */
static void tg_dist_km_grad(const double *p1, const double *p2, double *dlon, double *dlat) {
    double rlon1 = p1[0] * TG_DEG2RAD, rlat1 = p1[1] * TG_DEG2RAD;
    double rlon2 = p2[0] * TG_DEG2RAD, rlat2 = p2[1] * TG_DEG2RAD;
    double hl = (rlat1 - rlat2) / 2.0, hn = (rlon1 - rlon2) / 2.0;
    double s1 = sin(hl), c1 = cos(hl), s2 = sin(hn), c2 = cos(hn);
    double cl1 = cos(rlat1), cl2 = cos(rlat2);
    double a = s1 * s1 + cl1 * cl2 * s2 * s2;
    if (a <= 0.0 || a >= 1.0) { *dlon = 0.0; *dlat = 0.0; return; }
    double dg_da = TG_EARTH_R_KM / sqrt(a * (1.0 - a));
    double da_dlat = s1 * c1 - sin(rlat1) * cl2 * s2 * s2;
    double da_dlon = cl1 * cl2 * s2 * c2;
    *dlon = dg_da * da_dlon * TG_DEG2RAD;
    *dlat = dg_da * da_dlat * TG_DEG2RAD;
}

/* treegp distance functions [recollection]: euclidean -> sqrt(sum(((a-b)/scale)^2));
 * lld -> sqrt((km/scale0)^2 + (ddepth/scale1)^2) (unscaled form in-tree: run_seismic.py:230-233) */
static double tg_dist(int dist_id, const double *p1, const double *p2, int dx, const double *scales) {
    if (dist_id == 0) {
        double sq = 0.0;
        for (int i = 0; i < dx; ++i) {
            double diff = (p1[i] - p2[i]) / scales[i];
            sq += diff * diff;
        }
        return sqrt(sq);
    } else {
        double dk = tg_dist_km(p1, p2) / scales[0];
        double dd = (p1[2] - p2[2]) / scales[1];
        return sqrt(dk * dk + dd * dd);
    }
}

/* d dist / d p1[i] */
static double tg_dist_deriv_xi(int dist_id, const double *p1, const double *p2, int dx, const double *scales,
                               int i, double d) {
    if (d == 0.0) return 0.0;
    if (dist_id == 0) {
        return (p1[i] - p2[i]) / (scales[i] * scales[i] * d);
    } else {
        if (i == 2) return (p1[2] - p2[2]) / (scales[1] * scales[1] * d);
        double dlon, dlat;
        tg_dist_km_grad(p1, p2, &dlon, &dlat);
        double g = tg_dist_km(p1, p2);
        return g * (i == 0 ? dlon : dlat) / (scales[0] * scales[0] * d);
    }
}

/* d dist / d scales[i] */
static double tg_dist_deriv_scale(int dist_id, const double *p1, const double *p2, int dx, const double *scales,
                                  int i, double d) {
    if (d == 0.0) return 0.0;
    if (dist_id == 0) {
        double diff = p1[i] - p2[i];
        return -(diff * diff) / (scales[i] * scales[i] * scales[i] * d);
    } else {
        double num = (i == 0) ? tg_dist_km(p1, p2) : (p1[2] - p2[2]);
        return -(num * num) / (scales[i] * scales[i] * scales[i] * d);
    }
}

static double tg_w(int kern_id, double d, double sv) {
    if (kern_id == 0) return sv * exp(-1.0 * d * d);
    double s3d = sqrt(3.0) * d;
    return sv * (1.0 + s3d) * exp(-s3d);
}

/* dk/dd */
static double tg_w_deriv(int kern_id, double d, double sv) {
    if (kern_id == 0) return -2.0 * d * sv * exp(-1.0 * d * d);
    return -3.0 * sv * d * exp(-sqrt(3.0) * d);
}

/* gprf.py:339,342,373 — out is n1 x n2 row-major; distance_only -> the scaled distance matrix */
void tg_kernel_matrix(const double *X1, int n1, const double *X2, int n2, int dx, int dist_id,
                      const double *dfn_params, int kern_id, const double *wfn_params, int distance_only,
                      double *out) {
    for (int p = 0; p < n1; ++p)
        for (int q = 0; q < n2; ++q) {
            double d = tg_dist(dist_id, X1 + (size_t)p * dx, X2 + (size_t)q * dx, dx, dfn_params);
            out[(size_t)p * n2 + q] = distance_only ? d : tg_w(kern_id, d, wfn_params[0]);
        }
}

/* gprf.py:353 — out[q] = d k(x_p, x_q) / d x_p[i]  (entry q == p is whatever the formula gives; the
 * caller zeroes it, gprf.py:354) */
void tg_kernel_deriv_wrt_xi_row(const double *X, int n, int dx, int p, int i, int dist_id,
                                const double *dfn_params, int kern_id, const double *wfn_params, double *out) {
    const double *xp = X + (size_t)p * dx;
    for (int q = 0; q < n; ++q) {
        const double *xq = X + (size_t)q * dx;
        double d = tg_dist(dist_id, xp, xq, dx, dfn_params);
        double dd = tg_dist_deriv_xi(dist_id, xp, xq, dx, dfn_params, i, d);
        out[q] = tg_w_deriv(kern_id, d, wfn_params[0]) * dd;
    }
}

/* Convenience for the oracle's mode="matrix": the routine above for every p, diagonal zeroed as the
 * caller does at gprf.py:354.  Entry-wise identical arithmetic; only the per-row call is hoisted. */
void tg_kernel_deriv_wrt_xi_allrows(const double *X, int n, int dx, int i, int dist_id,
                                    const double *dfn_params, int kern_id, const double *wfn_params,
                                    double *out) {
    for (int p = 0; p < n; ++p) {
        tg_kernel_deriv_wrt_xi_row(X, n, dx, p, i, dist_id, dfn_params, kern_id, wfn_params, out + (size_t)p * n);
        out[(size_t)p * n + p] = 0.0;
    }
}

/* gprf.py:374 — d K / d dfn_params[i]; dists = precomputed distance matrix (n1 x n2) */
void tg_kernel_deriv_wrt_i(const double *X1, int n1, const double *X2, int n2, int dx, int i, int dist_id,
                           const double *dfn_params, int kern_id, const double *wfn_params, const double *dists,
                           double *out) {
    for (int p = 0; p < n1; ++p)
        for (int q = 0; q < n2; ++q) {
            double d = dists[(size_t)p * n2 + q];
            double dd = tg_dist_deriv_scale(dist_id, X1 + (size_t)p * dx, X2 + (size_t)q * dx, dx, dfn_params, i, d);
            out[(size_t)p * n2 + q] = tg_w_deriv(kern_id, d, wfn_params[0]) * dd;
        }
}

/* exposed for the haversine doctest restatement (run_seismic.py:24-33) */
double tg_dist_km_pub(double lon1, double lat1, double lon2, double lat2) {
    double a[3] = {lon1, lat1, 0.0}, b[3] = {lon2, lat2, 0.0};
    return tg_dist_km(a, b);
}
