/*
 * batotp_oracle.c -- TEST INFRASTRUCTURE ONLY (see batotp_oracle.h).
 *
 * CPU restatement of the batotp hot path in plain C99.  Compile with
 *     gcc -std=c99 -O2 -ffp-contract=off
 * All citations are file:line under /root/reference.
 */
#include "batotp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* std::min / std::max semantics of libstdc++: min(a,b) = (b<a)?b:a ; max(a,b) = (a<b)?b:a */
static inline double dmin(double a, double b) { return (b < a) ? b : a; }
static inline double dmax(double a, double b) { return (a < b) ? b : a; }
/* util.h:94-96 */
static inline int sgn_d(double v) { return (0.0 < v) - (v < 0.0); }

/* ------------------------------------------------------------------------------------------ */
/* path container                                                                              */
/* ------------------------------------------------------------------------------------------ */
static int prob_dyn_dim(const batotp_problem *prob)
{
    if (!(prob->flags & BATOTP_F_TRQ_ON)) return 0;
    /* ba.cpp:876-888: _dynDim = nCart for a parallel mechanism, nJoints otherwise */
    return (prob->flags & BATOTP_F_PARALLEL) ? prob->n_cart : prob->n_joints;
}

bo_path *bo_path_new(const batotp_problem *prob, int64_t n)
{
    bo_path *p = (bo_path *)calloc(1, sizeof(bo_path));
    if (!p) return NULL;
    p->n = n;
    p->n_theta = prob->n_joints;
    p->n_cart = prob->n_cart;
    p->dyn_dim = prob_dyn_dim(prob);
    p->n_ch = p->n_theta + p->n_cart + 4 * p->dyn_dim;
    p->parallel_now = (prob->flags & BATOTP_F_PARALLEL) ? 1 : 0;
    p->sC = (double *)calloc((size_t)n, sizeof(double));
    p->sMVC = (double *)calloc((size_t)n, sizeof(double));
    p->coef = (double *)calloc((size_t)n * 4 * (size_t)p->n_ch, sizeof(double));
    p->samp = (double *)calloc((size_t)n * 3 * (size_t)(p->n_theta + p->n_cart), sizeof(double));
    p->dyn = (double *)calloc((size_t)n * 4 * (size_t)(p->dyn_dim ? p->dyn_dim : 1), sizeof(double));
    p->mvc = (double *)calloc((size_t)n * 3, sizeof(double));
    if (!p->sC || !p->sMVC || !p->coef || !p->samp || !p->dyn || !p->mvc) {
        bo_path_free(p);
        return NULL;
    }
    return p;
}

void bo_path_free(bo_path *p)
{
    if (!p) return;
    free(p->sC); free(p->sMVC); free(p->coef); free(p->samp); free(p->dyn); free(p->mvc);
    free(p);
}

/* ------------------------------------------------------------------------------------------ */
/* spline                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* Spline::solveTriDiagNatural, spline.cpp:252-276.  d has n+1 entries (d[0] untouched = 0). */
static void tridiag_natural(double *d, int64_t npts)
{
    int64_t n = npts - 1;
    double a = 1.0, b = 4.0;
    double *c = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t i;
    for (i = 0; i < n; i++) c[i] = 1.0;
    c[1] /= b;
    d[1] /= b;
    for (i = 2; i < n; i++) {
        c[i] /= b - a * c[i - 1];
        d[i] = (d[i] - a * d[i - 1]) / (b - a * c[i - 1]);
    }
    /* spline.cpp:269: the last unknown is eliminated once more instead of being forced to 0 */
    d[n] = (d[n] - a * d[n - 1]) / (b - a * c[n - 1]);
    for (i = n; i > 1; --i) d[i - 1] -= c[i - 1] * d[i];
    free(c);
}

/* Spline::solveTriDiagClamped, spline.cpp:225-243 */
static void tridiag_clamped(double *d, int64_t n)
{
    double a = 1.0;
    double *c = (double *)malloc(sizeof(double) * (size_t)n);
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t i;
    for (i = 0; i < n; i++) { c[i] = 1.0; b[i] = 4.0; }
    b[0] = 2.0; b[n - 1] = 2.0;
    c[0] /= b[0];
    d[0] /= b[0];
    for (i = 1; i < n; i++) {
        c[i] /= b[i] - a * c[i - 1];
        d[i] = (d[i] - a * d[i - 1]) / (b[i] - a * c[i - 1]);
    }
    /* spline.cpp:240: "for (i = n-2; i-- > 0;)" starts at n-3 */
    for (i = n - 2; i-- > 0;) d[i] -= c[i] * d[i + 1];
    free(c); free(b);
}

/* Spline::getSplineCoeffs, spline.cpp:168-211.  c = [4][n] (c0,c1,c2,c3); row n-1 is not written. */
void bo_spline_coeffs(const double *y, int64_t n, double *c, int clamped)
{
    double *sol = (double *)calloc((size_t)n, sizeof(double));
    double *c0 = c, *c1 = c + n, *c2 = c + 2 * n, *c3 = c + 3 * n;
    int64_t i;
    for (i = 1; i < n - 1; i++) sol[i] = 6 * (y[i - 1] - 2 * y[i] + y[i + 1]);
    if (clamped) tridiag_clamped(sol, n);
    else tridiag_natural(sol, n);
    for (i = 0; i < n - 1; i++) {
        c3[i] = (sol[i + 1] - sol[i]) / 6.0;
        c2[i] = sol[i] / 2.0;
        c1[i] = y[i + 1] - y[i] - (sol[i + 1] + 2 * sol[i]) / 6.0;
        c0[i] = y[i];
    }
    free(sol);
}

/* the second derivatives Spline::getSplineCoeffs forms its rows from (spline.cpp:168-200, "natural"): sol[0 .. n-1] */
void bo_spline_sol(const double *y, int64_t n, double *sol)
{
    int64_t i;
    for (i = 0; i < n; i++) sol[i] = 0.0;
    for (i = 1; i < n - 1; i++) sol[i] = 6 * (y[i - 1] - 2 * y[i] + y[i + 1]);
    tridiag_natural(sol, n);
}

/* Spline::findInterpSegs, spline.cpp:56-99 */
int bo_find_interp_segs(const double *a_in, int64_t n_in, const double *a_out, int64_t n_out,
                        int32_t *seg, double *tau)
{
    int64_t i;
    int64_t cur = 0;
    for (i = 0; i < n_out; i++) {
        double ao = a_out[i];
        for (;;) {
            if (ao < a_in[cur + 1] || cur == n_in - 2) { seg[i] = (int32_t)cur; break; }
            cur++;
        }
    }
    for (i = 0; i < n_in - 1; i++) {
        double den = a_in[i + 1] - a_in[i];
        if (den < 1e-20) return -1;
    }
    for (i = 0; i < n_out; i++) {
        int32_t s = seg[i];
        tau[i] = (a_out[i] - a_in[s]) / (a_in[s + 1] - a_in[s]);
    }
    return 0;
}

/* Spline::interp1spline, spline.cpp:129-155 */
void bo_interp1_spline(const double *c, int64_t n_c, const int32_t *seg, const double *tau,
                       int64_t n_out, double tfact, double *b, double *bD, double *bD2)
{
    const double *C0 = c, *C1 = c + n_c, *C2 = c + 2 * n_c, *C3 = c + 3 * n_c;
    double vfact = 1.0 / tfact;
    double afact = vfact * vfact;
    int64_t i;
    for (i = 0; i < n_out; i++) {
        int32_t j = seg[i];
        double t = tau[i];
        double t2 = t * t, t3 = t2 * t;
        double c3 = C3[j], c2 = C2[j], c1 = C1[j], c0 = C0[j];
        b[i] = c3 * t3 + c2 * t2 + c1 * t + c0;
        bD[i] = (3 * c3 * t2 + 2 * c2 * t + c1) * vfact;
        bD2[i] = (6 * c3 * t + 2 * c2) * afact;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* small dense algebra                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* solveLinSys (util.cpp:413-442), isSVD=0: x = A.lu().solve(b) with Eigen::PartialPivLU.
 * Eigen is not vendored in the reference (CMake find_package(Eigen3), README names 3.3.4); the
 * published algorithm (Eigen/src/LU/PartialPivLU.h, unblocked path for small matrices):
 *   for k: pivot = first row of max |A(i,k)|, i>=k; swap rows; A(i,k) /= A(k,k);
 *          A(i,j) -= A(i,k)*A(k,j)  (i,j > k)
 *   solve: permute b; unit-lower forward substitution and upper back substitution, both
 *          column-oriented (rhs[i] (/= diag); rhs[rest] -= rhs[i]*col(i)) as in
 *          Eigen/src/Core/products/TriangularSolverVector.h (ColMajor). */
void bo_solve_lin_sys(int dim, const double *A, const double *b, double *x)
{
    double lu[64];
    double r[8];
    int perm[8];
    int i, j, k;
    for (i = 0; i < dim; i++) {
        for (j = 0; j < dim; j++) lu[i * 8 + j] = A[i * dim + j];
        r[i] = b[i];
        perm[i] = i;
    }
    for (k = 0; k < dim; k++) {
        int piv = k;
        double best = fabs(lu[k * 8 + k]);
        for (i = k + 1; i < dim; i++) {
            double v = fabs(lu[i * 8 + k]);
            if (v > best) { best = v; piv = i; }
        }
        perm[k] = piv;
        if (best != 0.0) {
            if (piv != k) {
                for (j = 0; j < dim; j++) {
                    double t = lu[k * 8 + j];
                    lu[k * 8 + j] = lu[piv * 8 + j];
                    lu[piv * 8 + j] = t;
                }
            }
            for (i = k + 1; i < dim; i++) lu[i * 8 + k] /= lu[k * 8 + k];
        }
        for (i = k + 1; i < dim; i++)
            for (j = k + 1; j < dim; j++)
                lu[i * 8 + j] -= lu[i * 8 + k] * lu[k * 8 + j];
    }
    /* apply the row transpositions to the right-hand side in order */
    for (k = 0; k < dim; k++) {
        if (perm[k] != k) { double t = r[k]; r[k] = r[perm[k]]; r[perm[k]] = t; }
    }
    /* unit lower, column oriented */
    for (i = 0; i < dim; i++) {
        if (r[i] != 0.0)
            for (j = i + 1; j < dim; j++) r[j] -= r[i] * lu[j * 8 + i];
    }
    /* upper, column oriented, from the last row */
    for (i = dim - 1; i >= 0; i--) {
        if (r[i] != 0.0) {
            r[i] /= lu[i * 8 + i];
            for (j = 0; j < i; j++) r[j] -= r[i] * lu[j * 8 + i];
        }
    }
    for (i = 0; i < dim; i++) x[i] = r[i];
}

/* solveQuadratic, util.cpp:361-383 */
int bo_solve_quadratic(double A, double B, double C, double *sol1, double *sol2)
{
    double rad, den, F1, F2;
    if (fabs(A) < 1e-308) {
        if (fabs(B) < 1e-308) return -2;
        *sol1 = -C / B; *sol2 = *sol1;
        return 0;
    }
    rad = B * B - 4 * A * C;
    if (rad < 0) return -1;
    den = 2 * A;
    F1 = -B / den; F2 = sqrt(rad) / den;
    *sol1 = F1 + F2;
    *sol2 = F1 - F2;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* robot models on the path                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* Robot::findCSPR3DOFpmat, robot.cpp:291-322 */
void bo_cspr_pmat(double pmat[9])
{
    double cible1[3] = {1.0941, -4.9074, 2.5542};
    double delta1[3] = {-0.765, 0.112, 3.74};
    double cible3[3] = {0.2098, 5.3409, 2.6236};
    double delta2[3] = {0.43, 0.125, 3.615};
    double p1[3], p2[3], p3[3] = {-5.9751, 0.1399, 6.1543};
    int ind[3] = {1, 0, 2};
    int i, j;
    for (i = 0; i < 3; i++) { p1[i] = cible1[i] + delta1[i]; p2[i] = cible3[i] + delta2[i]; }
    for (i = 0; i < 3; i++) {
        int it = ind[i];
        pmat[i * 3 + 0] = -p1[it];
        pmat[i * 3 + 1] = -p2[it];
        pmat[i * 3 + 2] = -p3[it];
    }
    for (i = 0; i < 3; i++) {
        double centroid = 1 / 3.0 * (pmat[i * 3 + 0] + pmat[i * 3 + 1] + pmat[i * 3 + 2]);
        for (j = 0; j < 3; j++) pmat[i * 3 + j] -= centroid;
    }
}

/* Robot::setA for CSPR3DOF, robot.cpp:534-558: A[i][j] = (cart[i]-pmat[i][j])/theta[j] */
void bo_cspr_setA(const double *pmat, const double *theta, const double *cart, double *A)
{
    int i, j;
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++) A[i * 3 + j] = (cart[i] - pmat[i * 3 + j]) / theta[j];
}

static const double kPI = 3.14159265358979323846;      /* config.h:27 */
#define BO_DEG2RAD (kPI / 180.0)                       /* config.h:28 */
static const double kG = 9.81;                         /* config.h:30 */

/* ------------------------------------------------------------------------------------------ */
/* per-knot precompute                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* BA::evalSplineFullTraj with oldRes == newRes == sres, ba.cpp:790-863 */
int bo_precompute_kin(const batotp_problem *prob, bo_path *p, const double *y, double sres)
{
    int64_t n = p->n, i;
    int nch = p->n_theta + p->n_cart, ch;
    int64_t n_new;
    double new_res, s_scale;
    int32_t *seg;
    double *tau;
    (void)prob;

    /* ba.cpp:796-798 */
    n_new = (int64_t)ceil(sres / sres * (double)(n - 1)) + 1;
    if (n_new < 4) n_new = 4;
    new_res = sres * (double)(n - 1) / (double)(n_new - 1);
    if (n_new != n) return -1; /* oracle covers the N->N call only */

    /* ba.cpp:800-806: sC = sres * iota */
    for (i = 0; i < n; i++) p->sC[i] = sres * (double)i;
    /* ba.cpp:809-813 */
    s_scale = p->sC[n - 1] / (double)(n_new - 1);
    for (i = 0; i < n; i++) p->sMVC[i] = s_scale * (double)i;
    /* ba.cpp:815-819 */
    p->sres_c = sres;
    p->vfact = 1 / p->sres_c;
    p->afact = p->vfact * p->vfact;
    p->sres = new_res;

    for (ch = 0; ch < nch; ch++)
        bo_spline_coeffs(y + (int64_t)ch * n, n, p->coef + (int64_t)ch * 4 * n, 0);

    seg = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    tau = (double *)malloc(sizeof(double) * (size_t)n);
    if (bo_find_interp_segs(p->sC, n, p->sMVC, n, seg, tau) != 0) { free(seg); free(tau); return -1; }
    for (ch = 0; ch < nch; ch++) {
        double *o = p->samp + (int64_t)ch * 3 * n;
        bo_interp1_spline(p->coef + (int64_t)ch * 4 * n, n, seg, tau, n, sres, o, o + n, o + 2 * n);
    }
    free(seg); free(tau);
    return 0;
}

/* Robot::dynRR, robot.cpp:377-431 */
void bo_dyn_rr(const bo_path *p, const double *trig, double *a1, double *a2, double *a3, double *a4)
{
    int64_t n = p->n, i;
    const double *th0 = p->samp, *thD0 = p->samp + n, *thDD0 = p->samp + 2 * n;
    const double *th1s = p->samp + 3 * n, *thD1 = p->samp + 4 * n, *thDD1 = p->samp + 5 * n;
    double A1 = .4, A2 = .6, m1 = 4, m2 = 8;
    for (i = 0; i < n; i++) {
        double th1 = BO_DEG2RAD * th0[i];
        double th2 = BO_DEG2RAD * th1s[i];
        double dth1 = BO_DEG2RAD * thD0[i];
        double dth2 = BO_DEG2RAD * thD1[i];
        double ddth1 = BO_DEG2RAD * thDD0[i];
        double ddth2 = BO_DEG2RAD * thDD1[i];
        double c1, c2, c12, s2, A11, A12, A22, ccFact;
        if (trig) { c1 = trig[i]; c2 = trig[n + i]; c12 = trig[2 * n + i]; s2 = trig[3 * n + i]; }
        else { double tr[4]; bo_rr_dyn_trig(th1, th2, tr); c1 = tr[0]; c2 = tr[1]; c12 = tr[2]; s2 = tr[3]; }

        A11 = .25 * m1 * A1 * A1 + m2 * (A1 * A1 + .25 * A2 * A2 + A1 * A2 * c2);
        A12 = .5 * m2 * (.5 * A2 * A2 + A1 * A2 * c2);
        A22 = .25 * m2 * A2 * A2;

        a1[i] = A11 * dth1 + A12 * dth2;
        a1[n + i] = A12 * dth1 + A22 * dth2;

        ccFact = m2 * A1 * A2 * s2;
        a2[i] = A11 * ddth1 + A12 * ddth2 - ccFact * dth2 * (dth1 + .5 * dth2);
        a2[n + i] = A12 * ddth1 + A22 * ddth2 - .5 * ccFact * dth1 * dth1;

        a3[i] = 10 * dth1;
        a3[n + i] = 10 * dth2;

        a4[i] = .5 * kG * (m1 * A1 * c1 + m2 * (2.0 * A1 * c1 + A2 * c12));
        a4[n + i] = .5 * kG * m2 * A2 * c12;
    }
}

/* BA::findDynModel, ba.cpp:873-949 */
int bo_precompute_dyn(const batotp_problem *prob, bo_path *p, const double *trig)
{
    int64_t n = p->n, i;
    int d = p->dyn_dim, j, k;
    double *a1, *a2, *a3, *a4;
    if (d == 0) return 0;
    if (d != prob->n_joints) return -1; /* ba.cpp:940-946 loops over _nJoints */
    a1 = p->dyn; a2 = p->dyn + (int64_t)d * n; a3 = p->dyn + 2 * (int64_t)d * n; a4 = p->dyn + 3 * (int64_t)d * n;

    if (prob->flags & BATOTP_F_PARALLEL) {
        /* Robot::dynCSPR3DOF, robot.cpp:487-517 (cartD, cartD2 = knot samples) */
        const double *cs = p->samp + (int64_t)p->n_theta * 3 * n;
        if (prob->robot_type != BATOTP_ROBOT_CSPR3DOF) return -1;
        for (i = 0; i < n; i++) {
            for (j = 0; j < 3; j++) {
                a1[(int64_t)j * n + i] = -cs[(int64_t)j * 3 * n + n + i];
                a2[(int64_t)j * n + i] = -cs[(int64_t)j * 3 * n + 2 * n + i];
                a3[(int64_t)j * n + i] = 0.0;
                a4[(int64_t)j * n + i] = 0.0;
            }
            a4[2 * n + i] = kG;
        }
    } else {
        if (p->serial) {
            /* one more case of Robot::dynSerial's switch (robot.cpp:349-360): a table-driven chain */
            if (bo_dyn_serial(p->serial, p, trig, a1, a2, a3, a4) != 0) return -1;
        } else {
            if (prob->robot_type != BATOTP_ROBOT_RR) return -1; /* robot.cpp:349-360 */
            bo_dyn_rr(p, trig, a1, a2, a3, a4);
        }
    }

    if ((prob->flags & BATOTP_F_PARALLEL) && (prob->flags & BATOTP_F_PAR2SER)) {
        /* ba.cpp:916-938 */
        const double *cs = p->samp + (int64_t)p->n_theta * 3 * n;
        for (i = 0; i < n; i++) {
            double A[9], th[3], ca[3], bs[3], xs[3];
            double *ak[4];
            ak[0] = a1; ak[1] = a2; ak[2] = a3; ak[3] = a4;
            for (j = 0; j < 3; j++) { ca[j] = cs[(int64_t)j * 3 * n + i]; th[j] = p->samp[(int64_t)j * 3 * n + i]; }
            bo_cspr_setA(prob->pmat, th, ca, A);
            for (k = 0; k < 4; k++) {
                for (j = 0; j < 3; j++) bs[j] = ak[k][(int64_t)j * n + i];
                bo_solve(prob, 3, A, bs, xs);
                for (j = 0; j < 3; j++) ak[k][(int64_t)j * n + i] = xs[j];
            }
        }
        p->parallel_now = 0; /* ba.cpp:937 */
    }

    /* ba.cpp:940-946 */
    for (k = 0; k < 4; k++)
        for (j = 0; j < d; j++) {
            int ch = p->n_theta + p->n_cart + k * d + j;
            bo_spline_coeffs(p->dyn + ((int64_t)k * d + j) * n, n, p->coef + (int64_t)ch * 4 * n, 0);
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* the sweep                                                                                   */
/* ------------------------------------------------------------------------------------------ */

typedef struct sweep_ctx {
    const batotp_problem *prob;
    const bo_path *p;
    int dir;
    /* reverse-sweep curve (forward sweep only) */
    const double *mvc_s, *mvc_sdot;
    int64_t n_mvc;
    /* Traj cursor state, ba.h:94-103,144-145 */
    int64_t cur_seg_c, cur_seg_mvc;
    double tau_c, tau_mvc;
    double s_cur, sdot_cur, sddot_l, sddot_h;
    int sdot_lim_type_t, is_on_sdot;
    /* point buffers, ba.h:119-134 */
    double thetapt[BATOTP_MAX_JOINTS], thetaDpt[BATOTP_MAX_JOINTS], thetaD2pt[BATOTP_MAX_JOINTS];
    double cartpt[BATOTP_MAX_CART], cartDpt[BATOTP_MAX_CART], cartD2pt[BATOTP_MAX_CART];
    double cart_acc[3];
    double a1pt[BATOTP_MAX_JOINTS], a2pt[BATOTP_MAX_JOINTS], a3pt[BATOTP_MAX_JOINTS], a4pt[BATOTP_MAX_JOINTS];
    double Apt[9];
    double sdot_min;     /* BA::_sdotMin, ba.h:319 */
    uint32_t status;
    int32_t n_fail;
} sweep_ctx;

/* BA::updateCurSeg, ba.cpp:1617-1652.  The reference loops forever when sCur compares neither
 * above nor below the segment start without lying inside it (NaN); here that sets a status bit. */
static void update_cur_seg(const double *s, int64_t n, double sCur, int64_t *curSeg, double *tau,
                           uint32_t *status)
{
    double sSeg;
    const int64_t lastSeg = n - 2;
    for (;;) {
        int moved = 0;
        sSeg = s[*curSeg];
        if (sCur >= sSeg && sCur <= s[*curSeg + 1]) break;
        if (sCur > sSeg) {
            if (*curSeg >= lastSeg) { *curSeg = lastSeg; break; }
            (*curSeg)++; moved = 1;
        }
        if (sCur < sSeg) {
            if (*curSeg <= 0) { *curSeg = 0; break; }
            (*curSeg)--; moved = 1;
        }
        if (!moved) { *status |= BATOTP_ST_NONFINITE; break; }
    }
    *tau = (sCur - sSeg) / (s[*curSeg + 1] - sSeg);
}

/* BA::evalCartQuadCoeffs, ba.cpp:1423-1439 */
static void eval_cart_quad(sweep_ctx *c)
{
    double vx = c->cartDpt[0], vy = c->cartDpt[1], vz = c->cartDpt[2];
    double ax = c->cartD2pt[0], ay = c->cartD2pt[1], az = c->cartD2pt[2];
    c->cart_acc[0] = vx * vx + vy * vy + vz * vz;
    c->cart_acc[1] = 2 * (vx * ax + vy * ay + vz * az);
    c->cart_acc[2] = ax * ax + ay * ay + az * az;
}

/* BA::evalSplinePartials, ba.cpp:1341-1413 */
static void eval_spline_partials(sweep_ctx *c)
{
    const batotp_problem *prob = c->prob;
    const bo_path *p = c->p;
    int64_t n = p->n, seg;
    int i;
    double tau, tau2, tau3;

    update_cur_seg(p->sC, n, c->s_cur, &c->cur_seg_c, &c->tau_c, &c->status);
    seg = c->cur_seg_c;
    tau = c->tau_c; tau2 = tau * tau; tau3 = tau2 * tau;

    for (i = 0; i < p->n_theta; i++) {
        const double *co = p->coef + (int64_t)i * 4 * n;
        double c0 = co[seg], c1 = co[n + seg], c2 = co[2 * n + seg], c3 = co[3 * n + seg];
        c->thetapt[i] = c3 * tau3 + c2 * tau2 + c1 * tau + c0;
        c->thetaDpt[i] = (3 * c3 * tau2 + 2 * c2 * tau + c1) * p->vfact;
        c->thetaD2pt[i] = (6 * c3 * tau + 2 * c2) * p->afact;
    }
    if (prob->flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)) {
        for (i = 0; i < p->n_cart; i++) {
            const double *co = p->coef + (int64_t)(p->n_theta + i) * 4 * n;
            double c0 = co[seg], c1 = co[n + seg], c2 = co[2 * n + seg], c3 = co[3 * n + seg];
            c->cartpt[i] = c3 * tau3 + c2 * tau2 + c1 * tau + c0;
            c->cartDpt[i] = (3 * c3 * tau2 + 2 * c2 * tau + c1) * p->vfact;
            c->cartD2pt[i] = (6 * c3 * tau + 2 * c2) * p->afact;
        }
        eval_cart_quad(c);
    }
    if ((prob->flags & BATOTP_F_TRQ_ON) && p->dyn_dim > 0) {
        int d = p->dyn_dim;
        int base = p->n_theta + p->n_cart;
        for (i = 0; i < p->n_theta; i++) { /* ba.cpp:1385 loops over _nJoints (== dynDim) */
            const double *k1 = p->coef + (int64_t)(base + i) * 4 * n;
            const double *k2 = p->coef + (int64_t)(base + d + i) * 4 * n;
            const double *k3 = p->coef + (int64_t)(base + 2 * d + i) * 4 * n;
            const double *k4 = p->coef + (int64_t)(base + 3 * d + i) * 4 * n;
            c->a1pt[i] = k1[3 * n + seg] * tau3 + k1[2 * n + seg] * tau2 + k1[n + seg] * tau + k1[seg];
            c->a2pt[i] = k2[3 * n + seg] * tau3 + k2[2 * n + seg] * tau2 + k2[n + seg] * tau + k2[seg];
            c->a3pt[i] = k3[3 * n + seg] * tau3 + k3[2 * n + seg] * tau2 + k3[n + seg] * tau + k3[seg];
            c->a4pt[i] = k4[3 * n + seg] * tau3 + k4[2 * n + seg] * tau2 + k4[n + seg] * tau + k4[seg];
        }
        if (p->parallel_now) bo_cspr_setA(prob->pmat, c->thetapt, c->cartpt, c->Apt); /* ba.cpp:1407-1410 */
    }
}

/* BA::evalsdot ("linear"), ba.cpp:1590-1607 */
static double eval_sdot(sweep_ctx *c)
{
    int64_t seg;
    double v;
    update_cur_seg(c->mvc_s, c->n_mvc, c->s_cur, &c->cur_seg_mvc, &c->tau_mvc, &c->status);
    seg = c->cur_seg_mvc;
    v = c->mvc_sdot[seg] + c->tau_mvc * (c->mvc_sdot[seg + 1] - c->mvc_sdot[seg]);
    return dmax(v, c->sdot_min);
}

/* BA::sdotLim, ba.cpp:1204-1236 */
static void sdot_lim(sweep_ctx *c, double *sdot)
{
    const batotp_problem *prob = c->prob;
    const bo_path *p = c->p;
    double sdoti = *sdot;
    int i;
    if (c->dir == 1) {
        double sdotMVC = eval_sdot(c);
        if (*sdot > sdotMVC) { c->is_on_sdot = 1; *sdot = sdotMVC; }
        else c->is_on_sdot = 0;
    }
    *sdot = dmin(*sdot, p->sC[p->n - 1] / prob->integ_res);
    *sdot = dmax(*sdot, c->sdot_min);
    /* joint velocity limits are applied unconditionally (ba.cpp:1219-1225) with the theta'
     * of the previous evalSplinePartials call */
    for (i = 0; i < p->n_theta; i++) {
        if (fabs(c->thetaDpt[i]) > prob->jnt_thresh * p->vfact)
            *sdot = dmin(*sdot, fabs(prob->jnt_vel_max[i] / c->thetaDpt[i]));
    }
    if ((prob->flags & BATOTP_F_CART_VEL_ON) && c->cart_acc[0] > prob->quad_rad_thresh * p->afact)
        *sdot = dmin(*sdot, prob->cart_vel_max / sqrt(c->cart_acc[0]));
    if (*sdot < sdoti) c->sdot_lim_type_t = 1;
}

/* BA::verifySecondOrderConstraints, ba.cpp:1449-1581.  Returns 1 when violated. */
static int verify_second_order(sweep_ctx *c, double sdotCur, double sddotMax)
{
    const batotp_problem *prob = c->prob;
    const bo_path *p = c->p;
    double sdotSQ = sdotCur * sdotCur;
    double CartAccMaxSQ = prob->cart_acc_max * prob->cart_acc_max;
    int nJ = p->n_theta, nC = p->n_cart;
    int j;

    c->sddot_l = -sddotMax;
    c->sddot_h = sddotMax;

    if (prob->flags & BATOTP_F_TRQ_ON) {
        if (p->parallel_now) {
            /* ba.cpp:1463-1491 */
            double cStar1[BATOTP_MAX_CART], bStar[BATOTP_MAX_CART], xStar[BATOTP_MAX_CART];
            double Astar[64];
            int i, k, ii;
            for (i = 0; i < nC; i++) cStar1[i] = sdotSQ * c->a2pt[i] + sdotCur * c->a3pt[i] + c->a4pt[i];
            for (j = 0; j < nJ; j++) {
                double trqLim[2], sol[2];
                trqLim[0] = prob->jnt_trq_min[j]; trqLim[1] = prob->jnt_trq_max[j];
                for (ii = 0; ii < 2; ii++) {
                    for (i = 0; i < nC * nJ; i++) Astar[i] = c->Apt[i];
                    for (k = 0; k < nC; k++) {
                        bStar[k] = cStar1[k] - c->Apt[k * nJ + j] * trqLim[ii];
                        Astar[k * nJ + j] = -c->a1pt[k];
                    }
                    bo_solve(prob, nC, Astar, bStar, xStar);
                    sol[ii] = xStar[j];
                }
                c->sddot_h = dmin(c->sddot_h, dmax(sol[0], sol[1]));
                c->sddot_l = dmax(c->sddot_l, dmin(sol[0], sol[1]));
                if (c->sddot_l > c->sddot_h) return 1;
            }
        } else {
            /* ba.cpp:1495-1509 */
            for (j = 0; j < nJ; j++) {
                double a1pt = c->a1pt[j];
                double tmp1 = c->a3pt[j] * sdotCur + c->a4pt[j];
                double tmp2, s0, s1;
                if (fabs(a1pt) < prob->jnt_thresh * p->vfact) continue;
                tmp2 = c->a2pt[j] * sdotSQ + tmp1;
                s0 = (prob->jnt_trq_max[j] - tmp2) / a1pt;
                s1 = (prob->jnt_trq_min[j] - tmp2) / a1pt;
                c->sddot_h = dmin(c->sddot_h, dmax(s0, s1));
                c->sddot_l = dmax(c->sddot_l, dmin(s0, s1));
                if (c->sddot_l > c->sddot_h) return 1;
            }
        }
    }

    if (prob->flags & BATOTP_F_JNT_ACC_ON) {
        /* ba.cpp:1514-1534 */
        for (j = 0; j < nJ; j++) {
            double vpt = c->thetaDpt[j];
            double vTerm;
            int svpt;
            if (fabs(vpt) < prob->jnt_thresh * p->vfact) {
                if (fabs(c->thetaD2pt[j]) < prob->jnt_thresh * p->afact) continue;
                if (sdotSQ > prob->jnt_acc_max[j] / fabs(c->thetaD2pt[j])) return 1;
                else continue;
            }
            svpt = sgn_d(vpt);
            vTerm = c->thetaD2pt[j] * sdotSQ;
            c->sddot_h = dmin(c->sddot_h, (svpt * prob->jnt_acc_max[j] - vTerm) / vpt);
            c->sddot_l = dmax(c->sddot_l, (-svpt * prob->jnt_acc_max[j] - vTerm) / vpt);
            if (c->sddot_l > c->sddot_h) return 1;
        }
    }

    if (prob->flags & BATOTP_F_CART_ACC_ON) {
        /* ba.cpp:1535-1579 */
        double A = c->cart_acc[0];
        if (A > prob->quad_rad_thresh * p->afact) {
            double B = c->cart_acc[1] * sdotSQ;
            double C = c->cart_acc[2] * sdotSQ * sdotSQ - CartAccMaxSQ;
            double sol1 = 0, sol2 = 0, cmax, cmin;
            int ef = bo_solve_quadratic(A, B, C, &sol1, &sol2);
            if (ef == -1) return 1;
            cmax = dmax(sol1, sol2);
            cmin = dmin(sol1, sol2);
            c->sddot_h = dmin(c->sddot_h, cmax);
            c->sddot_l = dmax(c->sddot_l, cmin);
            if (c->sddot_l > c->sddot_h) return 1;
        } else {
            double C = c->cart_acc[2];
            if (C < prob->quad_rad_thresh * prob->quad_rad_thresh * p->afact * p->afact) return 0;
            if (sdotSQ * sdotSQ > CartAccMaxSQ / C) return 1;
            else return 0;
        }
    }
    return 0;
}

/* BA::applyAccelConstraintsBisectionPt, ba.cpp:1248-1332 */
static int apply_accel_bisection(sweep_ctx *c, double *sddot, int *nIter)
{
    const batotp_problem *prob = c->prob;
    const bo_path *p = c->p;
    double sdotErr, sdotGoodLast;
    const double sdotErrThresh = .001;
    double lowFact = .01;
    double sdotMin = 0;
    double sdotGood = sdotMin;
    int isViol;
    int anyGoodIter = 0;
    double sddotmax = 2 * p->sC[p->n - 1] / (prob->integ_res * prob->integ_res);
    double sdotL = sdotGood;
    double sdotH = c->sdot_cur;
    double sdotCur = sdotH;
    *nIter = 0;

    eval_spline_partials(c);

    for (;;) {
        isViol = verify_second_order(c, sdotCur, sddotmax);
        if (isViol) {
            sdotH = sdotCur;
            if (!anyGoodIter) {
                lowFact *= 2.0;
                sdotL = dmax(.999 * sdotMin, (1.0 - lowFact) * sdotH);
            }
        } else {
            if (*nIter == 0) break;
            anyGoodIter = 1;
            sdotGoodLast = sdotGood;
            sdotGood = sdotCur;
            sdotErr = fabs(sdotGood - sdotGoodLast) / sdotGood;
            if (sdotErr < sdotErrThresh || sdotCur < sdotMin) {
                c->sdot_cur = sdotCur;
                break;
            }
            sdotL = sdotCur;
        }
        (*nIter)++;
        if (*nIter > 100) return -1;
        if (sdotCur < 0 || ((sdotH - sdotL) / sdotH < 1e-20 && !anyGoodIter)) return -1;
        sdotCur = .5 * (sdotH + sdotL);
    }
    if (c->dir == 1) *sddot = c->sddot_h;
    else *sddot = c->sddot_l;
    return 0;
}

static void accel_pt(sweep_ctx *c, double *sddot)
{
    int nIter;
    if (apply_accel_bisection(c, sddot, &nIter) != 0) {
        /* ba.cpp:1024,1038,1091: the caller ignores the -1; sddot keeps its previous value */
        c->status |= BATOTP_ST_BISECT_FAIL;
        c->n_fail++;
    }
}

/* Butcher tableau, ba.cpp:58-63: _B[k][j]; stage j+1 uses column j */
static const double kB[6][6] = {
    {1. / 5, 3. / 40, 44. / 45, 19372. / 6561, 9017. / 3168, 35. / 384},
    {0, 9. / 40, -56. / 15, -25360. / 2187, -355. / 33, 0},
    {0, 0, 32. / 9, 64448. / 6561, 46732. / 5247, 500. / 1113},
    {0, 0, 0, -212. / 729, 49. / 176, 125. / 192},
    {0, 0, 0, 0, -5103. / 18656, -2187. / 6784},
    {0, 0, 0, 0, 0, 11. / 84}};

static void ctx_init(sweep_ctx *c, const batotp_problem *prob, const bo_path *p, int dir)
{
    memset(c, 0, sizeof(*c));
    c->prob = prob; c->p = p; c->dir = dir;
}

/* BA::sweep, ba.cpp:979-1195 */
int bo_sweep(const batotp_problem *prob, const bo_path *p, int dir,
             const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
             double *out_s, double *out_sdot, int64_t cap,
             int64_t *n_out, int64_t *n_steps, double *t_total, uint32_t *status, int32_t *n_bisect_fail)
{
    return bo_sweep_ex(prob, p, dir, mvc_s, mvc_sdot, n_mvc, out_s, out_sdot, cap, n_out, n_steps, t_total, status, n_bisect_fail, 0);
}

/* in_place != 0 mirrors the PRODUCT's storage rule for BATOTP_F_CURVES_IN_PLACE (include/batotp_hip.h; no counterpart in the
 * reference, whose arrays grow): the forward curve shares a buffer of cap points with the right-aligned reverse curve and the
 * path ends with BATOTP_ST_CAPACITY as soon as point i would come within 64 points of the reverse points still to be read.
 * The arithmetic of the sweep is untouched. */
int bo_sweep_ex(const batotp_problem *prob, const bo_path *p, int dir,
                const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
                double *out_s, double *out_sdot, int64_t cap,
                int64_t *n_out, int64_t *n_steps, double *t_total, uint32_t *status, int32_t *n_bisect_fail, int in_place)
{
    sweep_ctx cx;
    sweep_ctx *c = &cx;
    const int64_t maxIntegSteps = (int64_t)floor(prob->max_integ_time / prob->integ_res) + 1;
    double absh = prob->integ_res;
    double h, sLast, tElapsed = 0;
    double sArr[7] = {0, 0, 0, 0, 0, 0, 0}, sdotArr[7] = {0, 0, 0, 0, 0, 0, 0}, sddotArr[7] = {0, 0, 0, 0, 0, 0, 0};
    double sdotT, sddotT;
    double dsMin, dsMinV[6];
    /* ba.cpp:48-54 */
    const double kA[6] = {1. / 5, 3. / 10, 4. / 5, 8. / 9, 1.0, 1.0};
    double dA[6];
    int64_t nPts = 0, i, n = p->n;
    int j, k;

    ctx_init(c, prob, p, dir);
    c->mvc_s = mvc_s; c->mvc_sdot = mvc_sdot; c->n_mvc = n_mvc;
    *n_out = 0; *n_steps = 0; *t_total = 0; *status = 0; *n_bisect_fail = 0;
    if (cap < 2) { *status = BATOTP_ST_CAPACITY; return -1; }

    dA[0] = kA[0];
    for (j = 1; j < 6; j++) dA[j] = kA[j] - kA[j - 1];
    dA[5] = 1.0e-6;

    /* ba.cpp:998: getSplineCoeffs(traj.sdot) feeds only the dead "cubic" branch of evalsdot */

    if (dir == 1) { /* ba.cpp:1000-1009 */
        c->cur_seg_c = 0; c->tau_c = 0; sArr[0] = 0;
        c->cur_seg_mvc = 0; c->tau_mvc = 0;
        sLast = p->sC[n - 1];
    } else { /* ba.cpp:1010-1019 */
        c->cur_seg_c = n - 2; c->tau_c = 1; sArr[0] = p->sC[n - 1];
        c->cur_seg_mvc = n - 2; c->tau_mvc = 1;
        sLast = 0;
    }
    c->s_cur = sArr[0];
    c->sdot_cur = 0;
    c->sdot_min = 1.7976931348623157e308; /* ba.h:319; overwritten below before use in rev; see note */

    /* ba.cpp:1024-1041 bootstrap */
    accel_pt(c, &sddotArr[0]);
    h = dir * absh;
    sdotArr[0] = .1 * h * sddotArr[0];
    c->sdot_min = sdotArr[0];
    sdot_lim(c, &sdotArr[0]);
    c->sdot_min = sdotArr[0];
    c->sdot_cur = sdotArr[0];
    out_s[0] = sArr[0];
    accel_pt(c, &sddotArr[0]);
    sdotArr[0] = c->sdot_cur;
    sdot_lim(c, &sdotArr[0]);
    out_sdot[0] = sdotArr[0];

    h = dir * absh;
    /* ba.cpp:1050-1051: sArr.back() is still 0 here, so every dsMinV[j] is 0 */
    dsMin = 1.0e-6 * sArr[6] / 7;
    for (j = 0; j < 6; j++) dsMinV[j] = dsMin * dA[j];

    for (i = 1;; i++) { /* ba.cpp:1053: the array grows by nChunk, the loop is unbounded */
        double s0 = c->s_cur;
        if (i >= cap || (in_place && dir == 1 && i + 64 >= (cap - n_mvc) + c->cur_seg_mvc)) {
            *status = c->status | BATOTP_ST_CAPACITY; *n_bisect_fail = c->n_fail; *n_steps = i; return -1;
        }
        c->sdot_lim_type_t = 0;
        sdotT = sdotArr[0];
        sddotT = sddotArr[0];
        sArr[6] = sArr[0] + h * sdotT;
        sdotArr[6] = sdotArr[0] + h * sddotT;

        c->s_cur = sArr[6];
        sdot_lim(c, &sdotArr[6]); /* ba.cpp:1063: only the MVC cursor side effect survives */
        c->s_cur = s0;

        for (j = 0; j < 6; j++) {
            c->sdot_lim_type_t = 0;
            sdotT = 0;
            sddotT = 0;
            for (k = 0; k < j + 1; k++) {
                sdotT += kB[k][j] * sdotArr[k];
                sddotT += kB[k][j] * sddotArr[k];
            }
            sArr[j + 1] = sArr[0] + h * sdotT;
            sdotArr[j + 1] = sdotArr[0] + h * sddotT;
            sdotArr[j + 1] = dmax(sdotArr[j + 1], dsMinV[j] / absh); /* ba.cpp:1085 */
            c->s_cur = sArr[j + 1];
            sdot_lim(c, &sdotArr[j + 1]);
            c->sdot_cur = sdotArr[j + 1];
            accel_pt(c, &sddotArr[j + 1]);
            sdotArr[j + 1] = c->sdot_cur;
        }
        sArr[0] = sArr[6];
        sdotArr[0] = sdotArr[6];
        sddotArr[0] = sddotArr[6];
        out_s[i] = sArr[0];
        out_sdot[i] = sdotArr[0];

        if (c->s_cur * dir > sLast) { /* ba.cpp:1109-1115 */
            tElapsed = absh * (double)i;
            nPts = i + 1;
            break;
        }
        if (i > maxIntegSteps) { /* ba.cpp:1117-1122 */
            *status = c->status | BATOTP_ST_MAX_INTEG_TIME;
            *n_bisect_fail = c->n_fail;
            *n_steps = i;
            return -1;
        }
    }

    /* ba.cpp:1132-1134 */
    {
        double sRat = (sLast - out_s[nPts - 2]) / (out_s[nPts - 1] - out_s[nPts - 2]);
        out_sdot[nPts - 1] = out_sdot[nPts - 2] + sRat * (out_sdot[nPts - 1] - out_sdot[nPts - 2]);
        out_s[nPts - 1] = sLast;
    }
    if (dir == 1) {
        out_sdot[nPts - 1] = mvc_sdot[n_mvc - 1]; /* ba.cpp:1140 */
    } else {
        /* ba.cpp:1145-1146 */
        for (i = 0; i < nPts / 2; i++) {
            double t = out_s[i]; out_s[i] = out_s[nPts - 1 - i]; out_s[nPts - 1 - i] = t;
            t = out_sdot[i]; out_sdot[i] = out_sdot[nPts - 1 - i]; out_sdot[nPts - 1 - i] = t;
        }
    }
    *t_total = tElapsed; /* ba.cpp:1156 */
    *n_steps = nPts - 1;

    if (nPts < 4) { /* ba.cpp:1171-1184 */
        double tInteg[4], tNew[4], sN[4], sdN[4], tau4[4];
        int32_t seg4[4];
        double tResNew;
        int64_t m;
        if (cap < 4) { *status = c->status | BATOTP_ST_CAPACITY; return -1; }
        for (m = 0; m < nPts; m++) tInteg[m] = absh * (double)m;
        tResNew = tInteg[nPts - 1] / 3.;
        for (m = 0; m < 4; m++) tNew[m] = tResNew * (double)m;
        bo_find_interp_segs(tInteg, nPts, tNew, 4, seg4, tau4);
        for (m = 0; m < 4; m++) { /* Spline::interp1linear, spline.cpp:108-120 */
            int32_t sg = seg4[m];
            sN[m] = out_s[sg] + (out_s[sg + 1] - out_s[sg]) * tau4[m];
            sdN[m] = out_sdot[sg] + (out_sdot[sg + 1] - out_sdot[sg]) * tau4[m];
        }
        for (m = 0; m < 4; m++) { out_s[m] = sN[m]; out_sdot[m] = sdN[m]; }
        nPts = 4;
        c->status |= BATOTP_ST_SHORT;
    }
    *n_out = nPts;
    *status = c->status;
    *n_bisect_fail = c->n_fail;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* pointwise evaluation (K3) and the per-point KAT hook                                        */
/* ------------------------------------------------------------------------------------------ */

/* K3 definition: at knot i the cursor sits on segment min(i, n-2); theta', theta''
 * (and cart / a1..a4) are evaluated there; sdot starts from the sweep's upper clamp
 * sC.back()/integRes (ba.cpp:1216), is cut by the velocity limits of ba.cpp:1219-1229 using this
 * knot's own derivatives, then by the bisection of ba.cpp:1248-1332 with dir=+1. */
void bo_pointwise_mvc(const batotp_problem *prob, bo_path *p)
{
    int64_t n = p->n, i;
    sweep_ctx cx;
    sweep_ctx *c = &cx;
    for (i = 0; i < n; i++) {
        double sdot, sddot = 0;
        int nIter;
        ctx_init(c, prob, p, -1); /* dir=-1: no reverse-curve lookup inside sdotLim */
        c->sdot_min = 0;
        c->cur_seg_c = (i < n - 1) ? i : n - 2;
        c->s_cur = p->sC[i];
        eval_spline_partials(c);
        sdot = p->sC[n - 1] / prob->integ_res;
        sdot_lim(c, &sdot);
        c->sdot_cur = sdot;
        c->sddot_l = 0; c->sddot_h = 0;
        if (apply_accel_bisection(c, &sddot, &nIter) != 0) {
            /* no admissible sdot at this knot (ba.cpp:1307-1319): the interval is empty.  The reference leaves whatever its
             * early loop exits last wrote; the K3 definition publishes NaN bounds instead */
            c->sddot_l = NAN; c->sddot_h = NAN;
        }
        p->mvc[i] = c->sdot_cur;
        p->mvc[n + i] = c->sddot_l;
        p->mvc[2 * n + i] = c->sddot_h;
    }
}

void bo_point_eval(const batotp_problem *prob, const bo_path *p, int dir, double s, double sdot_in,
                   double *sdot_out, double *sddot_l, double *sddot_h, int32_t *n_iter, int32_t *rc)
{
    sweep_ctx cx;
    sweep_ctx *c = &cx;
    double sdot = sdot_in, sddot = 0;
    int nIter = 0;
    ctx_init(c, prob, p, -1);
    (void)dir;
    c->sdot_min = 0;
    c->cur_seg_c = 0;
    c->s_cur = s;
    eval_spline_partials(c);
    sdot_lim(c, &sdot);
    c->sdot_cur = sdot;
    *rc = apply_accel_bisection(c, &sddot, &nIter);
    *sdot_out = c->sdot_cur;
    *sddot_l = c->sddot_l;
    *sddot_h = c->sddot_h;
    *n_iter = nIter;
}

/* ------------------------------------------------------------------------------------------ */
/* Per-point known-answer hooks for the fp64 vectors read out of the reference binary with a    */
/* debugger (oracle/make_golden_f64.py, tests/golden/<case>/ref_point_kats.npz): ONE call of     */
/* the reference routine from the cursor state the reference had at that call.                  */
/* ------------------------------------------------------------------------------------------ */

/* BA::applyAccelConstraintsBisectionPt (ba.cpp:1248-1332) entered with traj.sCur, traj.sdotCur, traj.curSegC and the
 * caller's sddot as given; returns what the reference leaves in traj.sdotCur, sddot, traj.sddotL/H, nIter, the int it
 * returns and traj.curSegC */
void bo_kat_accel(const batotp_problem *prob, const bo_path *p, int dir, double s_cur, double sdot_cur, int64_t cur_seg_c,
                  double sddot_in, double *sdot_out, double *sddot_out, double *sddot_l, double *sddot_h, int32_t *n_iter,
                  int32_t *rc, int64_t *cur_seg_out)
{
    sweep_ctx cx;
    sweep_ctx *c = &cx;
    double sddot = sddot_in;
    int nIter = 0;
    ctx_init(c, prob, p, dir);
    c->cur_seg_c = cur_seg_c;
    c->s_cur = s_cur;
    c->sdot_cur = sdot_cur;
    *rc = apply_accel_bisection(c, &sddot, &nIter);
    *sdot_out = c->sdot_cur; *sddot_out = sddot; *sddot_l = c->sddot_l; *sddot_h = c->sddot_h;
    *n_iter = nIter; *cur_seg_out = c->cur_seg_c;
}

/* BA::sdotLim (ba.cpp:1204-1236) entered with traj.sCur, the caller's sdot, BA::_sdotMin, the point buffers
 * traj.thetaDpt / traj.CartAccCoeffs[0] of the PREVIOUS evaluation point and, in the forward sweep, the reverse curve
 * with its cursor traj.curSegMVC; returns sdot and the cursor */
void bo_kat_sdot_lim(const batotp_problem *prob, const bo_path *p, int dir, double s_cur, double sdot_in, double sdot_min,
                     const double *theta_d_pt, double cart_coeff0, const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
                     int64_t cur_seg_mvc, double *sdot_out, int64_t *cur_seg_mvc_out)
{
    sweep_ctx cx;
    sweep_ctx *c = &cx;
    double sdot = sdot_in;
    int i;
    ctx_init(c, prob, p, dir);
    c->s_cur = s_cur;
    c->sdot_min = sdot_min;
    for (i = 0; i < p->n_theta; i++) c->thetaDpt[i] = theta_d_pt[i];
    c->cart_acc[0] = cart_coeff0;
    c->mvc_s = mvc_s; c->mvc_sdot = mvc_sdot; c->n_mvc = n_mvc; c->cur_seg_mvc = cur_seg_mvc;
    sdot_lim(c, &sdot);
    *sdot_out = sdot; *cur_seg_mvc_out = c->cur_seg_mvc;
}
