/*
 * batotp_oracle_output.c -- TEST INFRASTRUCTURE ONLY (see batotp_oracle.h).
 *
 * Plain-C restatement of the output stage behind the hot path:
 * BA::interpOutputData (ba.cpp:1661-1931) for the configurations the device output stage covers --
 * JOINT paths (robots without kinematic model; KUKA and the two-link arm with their forward kinematics, SURVEY.md 8 f-3;
 * serial robots with the torque recomputation of ba.cpp:1791-1827) and CART paths of the 3-cable robot: the optimised s(t) is
 * re-sampled at constant time steps, the joint splines are evaluated there, the result is smoothed
 * and down-sampled (_outSmoothFact) and, when the output resolution is finer than the integration
 * step, re-interpolated.  It is the checker of batotp_hip_output.
 *
 * Pinning: tests/test_oracle_output.py -- the trajectories it produces, rounded to float32 the way
 * BA::trajWriteBIN writes them, are byte-identical to the reference binary's traj_out.dat of the
 * covered golden cases.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "batotp_oracle.h"

/* util.cpp:263-290 smooth(): centred moving average, shrinking windows at both ends */
static void out_smooth(double *x, int64_t n, int w)
{
    if (w > n) w = (int)n;
    const int half = w / 2 + w % 2 - 1;
    w = 2 * half + 1;
    double *y = (double *)malloc(sizeof(double) * (size_t)n);
    y[0] = x[0];
    y[n - 1] = x[n - 1];
    for (int i = 1; i < half; ++i) {
        const int span = 2 * i + 1;
        double head = 0, tail = 0;
        for (int j = 0; j < span; ++j) { head += x[j]; tail += x[n - j - 1]; }
        y[i] = head / span;
        y[n - i - 1] = tail / span;
    }
    for (int64_t i = half; i < n - half; ++i) {
        double acc = 0;
        for (int64_t j = i - half; j < i + half + 1; ++j) acc += x[j];
        y[i] = acc / w;
    }
    memcpy(x, y, sizeof(double) * (size_t)n);
    free(y);
}

/* resample every row of x[C][nIn]: smooth + linear down-sampling (ba.cpp:1838-1871) */
static double *out_smooth_down(double *x, int C, int64_t nIn, double smoothFact, int64_t *nDownOut)
{
    int64_t nDown = (int64_t)(int)((double)(nIn - 1) / smoothFact) + 1;
    if (nDown < 4) nDown = 4;
    const double sc = (double)(nIn - 1) / (double)(nDown - 1);
    const int window = (int)smoothFact;
    double *dn = (double *)malloc(sizeof(double) * (size_t)C * (size_t)nDown);
    for (int j = 0; j < C; ++j) {
        double *b = x + (size_t)j * nIn;
        out_smooth(b, nIn, window);
        int64_t cursor = 0;
        for (int64_t i = 0; i < nDown; ++i) { /* findInterpSegs(0..nIn-1, sc*i) + interp1linear */
            const double site = sc * (double)i;
            while (!(site < (double)(cursor + 1) || cursor == nIn - 2)) ++cursor;
            const double width = (double)(cursor + 1) - (double)cursor;
            const double t = (site - (double)cursor) / width;
            dn[(size_t)j * nDown + i] = b[cursor] + (b[cursor + 1] - b[cursor]) * t;
        }
    }
    *nDownOut = nDown;
    return dn;
}

/* natural splines of every row of x[C][n] evaluated at nUser uniform sites of the unit interval (ba.cpp:1873-1919) */
static double *out_reinterp(const double *x, int C, int64_t n, int64_t nUser, double tfact)
{
    const double c1 = 1. / (double)(n - 1), c2 = 1. / (double)(nUser - 1);
    double *g1 = (double *)malloc(sizeof(double) * (size_t)n), *g2 = (double *)malloc(sizeof(double) * (size_t)nUser);
    for (int64_t i = 0; i < n; ++i) g1[i] = c1 * (double)i;
    for (int64_t i = 0; i < nUser; ++i) g2[i] = c2 * (double)i;
    int32_t *sg = (int32_t *)malloc(sizeof(int32_t) * (size_t)nUser);
    double *tu = (double *)malloc(sizeof(double) * (size_t)nUser);
    double *e1 = (double *)malloc(sizeof(double) * (size_t)nUser), *e2 = (double *)malloc(sizeof(double) * (size_t)nUser);
    bo_find_interp_segs(g1, n, g2, nUser, sg, tu);
    double *up = (double *)malloc(sizeof(double) * (size_t)C * (size_t)nUser);
    double *c = (double *)calloc((size_t)4 * (size_t)n, sizeof(double));
    for (int j = 0; j < C; ++j) {
        memset(c, 0, sizeof(double) * 4 * (size_t)n);
        bo_spline_coeffs(x + (size_t)j * n, n, c, 0);
        bo_interp1_spline(c, n, sg, tu, nUser, tfact, up + (size_t)j * nUser, e1, e2);
    }
    free(c); free(g1); free(g2); free(sg); free(tu); free(e1); free(e2);
    return up;
}

int bo_output(const batotp_problem *prob, const batotp_output_params *prm, const bo_path *p, const double *fwd_s, int64_t n_fwd,
              double t_step, double **out, int32_t *n_cart_out, int32_t *n_trq_out, int64_t *n_out, double *sres_out)
{
    const int nJ = prm->n_joints;
    const int cable = prm->path_type == BATOTP_PATH_CART && prob->robot_type == BATOTP_ROBOT_CSPR3DOF && nJ == 3 && p->n_cart == 3 &&
                      (prob->flags & BATOTP_F_TRQ_ON) && (prob->flags & BATOTP_F_PARALLEL);
    const int jointPath = prm->path_type == BATOTP_PATH_JOINT || prm->path_type == 0;
    /* JOINT path of a robot with forward kinematics (KUKA, RR): three Cartesian rows from the joints */
    const int kin = jointPath && bo_fwdkin_trig_rows(prob->robot_type, nJ) != 0 && p->n_cart >= 3;
    /* torque recomputation of a serial robot, ba.cpp:1791-1827: the two-link arm's closed form or a chain model */
    const int serialTrq = jointPath && (prob->flags & BATOTP_F_TRQ_ON) && !(prob->flags & BATOTP_F_PARALLEL) &&
                          (p->serial != NULL || prob->robot_type == BATOTP_ROBOT_RR) && nJ == p->n_theta;
    const int joint = jointPath && (!(prob->flags & BATOTP_F_TRQ_ON) || serialTrq);
    /* joints and Cartesian rows taught together (path type BOTH): the Cartesian rows are evaluated like the joints (ba.cpp:1726-1736);
     * seven of them are position + quaternion (the UR5 example after aa2qVect) and are turned back into axis-angle at the very end
     * (ba.cpp:1920-1927) */
    const int both = prm->path_type == BATOTP_PATH_BOTH && p->n_cart >= 3 && !(prob->flags & BATOTP_F_TRQ_ON);
    const int pose = both && p->n_cart == 7;
    if (nJ < 1 || nJ > p->n_theta || n_fwd < 4 || (!cable && !joint && !both)) return -1;
    const int nC = cable ? 3 : (both ? p->n_cart : (kin ? 3 : 0)), nT = cable ? 3 : (serialTrq ? nJ : 0), C = nJ + nC + nT;
    double outRes = prm->out_res, smoothFact = prm->out_smooth_fact;
    const double outResUser = outRes;
    int reinterp = 0;
    if (outRes < prm->integ_res) { /* ba.cpp:1668-1675 */
        reinterp = 1;
        outRes = prm->integ_res;
        const double r = outResUser / outRes;
        smoothFact *= (r > 1. ? r : 1.);
    }

    /* tMVC[i] = t_step*i */
    double *tMVC = (double *)malloc(sizeof(double) * (size_t)n_fwd);
    for (int64_t i = 0; i < n_fwd; ++i) tMVC[i] = t_step * (double)i;
    double tLast = tMVC[n_fwd - 1];
    int64_t nOut = (int64_t)(int)(smoothFact * ceil(tMVC[n_fwd - 1] / outRes + 1.));
    if (nOut < 4) nOut = 4;

    /* output times (ba.cpp:1691-1699) */
    double *tOut = (double *)malloc(sizeof(double) * (size_t)nOut);
    for (int64_t i = 0; i < nOut; ++i) tOut[i] = (double)(i - 1);
    tOut[0] = 0;
    tOut[1] = 1.0 / 3.0;
    tOut[nOut - 1] = tOut[nOut - 2];
    tOut[nOut - 2] = tOut[nOut - 2] - 1.0 / 3.0;
    {
        const double c = tMVC[n_fwd - 1] / tOut[nOut - 1];
        for (int64_t i = 0; i < nOut; ++i) tOut[i] = c * tOut[i];
    }

    /* s at the output times: natural spline of sMVC over the step index */
    int32_t *seg = (int32_t *)malloc(sizeof(int32_t) * (size_t)nOut);
    double *tau = (double *)malloc(sizeof(double) * (size_t)nOut);
    double *sOut = (double *)malloc(sizeof(double) * (size_t)nOut);
    double *d1 = (double *)malloc(sizeof(double) * (size_t)nOut), *d2 = (double *)malloc(sizeof(double) * (size_t)nOut);
    double *cS = (double *)calloc((size_t)4 * (size_t)n_fwd, sizeof(double));
    bo_find_interp_segs(tMVC, n_fwd, tOut, nOut, seg, tau);
    bo_spline_coeffs(fwd_s, n_fwd, cS, 0);
    bo_interp1_spline(cS, n_fwd, seg, tau, nOut, p->sres / smoothFact, sOut, d1, d2);
    free(cS);

    /* path samples at those s: joint rows (JOINT path) or Cartesian rows + cable lengths (CART path) */
    bo_find_interp_segs(p->sC, p->n, sOut, nOut, seg, tau);
    int64_t n = nOut;
    double *x = (double *)calloc((size_t)C * (size_t)n, sizeof(double));
    if (!cable) {
        for (int j = 0; j < nJ; ++j)
            bo_interp1_spline(p->coef + (size_t)j * 4 * (size_t)p->n, p->n, seg, tau, nOut, outRes, x + (size_t)j * n, d1, d2);
        if (both)
            for (int j = 0; j < nC; ++j)
                bo_interp1_spline(p->coef + (size_t)(p->n_theta + j) * 4 * (size_t)p->n, p->n, seg, tau, nOut, outRes, x + (size_t)(nJ + j) * n, d1, d2);
    } else {
        for (int j = 0; j < nC; ++j)
            bo_interp1_spline(p->coef + (size_t)(p->n_theta + j) * 4 * (size_t)p->n, p->n, seg, tau, nOut, outRes, x + (size_t)(nJ + j) * n, d1, d2);
        for (int64_t i = 0; i < n; ++i) /* Robot::invKinCSPR3DOF, robot.cpp:243-278 */
            for (int k = 0; k < 3; ++k) {
                double sumSQ = 0.0;
                for (int r = 0; r < 3; ++r) {
                    const double dlt = x[(size_t)(nJ + r) * n + i] - prob->pmat[r * 3 + k];
                    sumSQ += dlt * dlt;
                }
                x[(size_t)k * n + i] = sqrt(sumSQ);
            }
    }

    if (kin) {
        /* Robot::fwdKin at the output points (ba.cpp:1722-1725).  The two-link arm's routine leaves the third row alone
         * (robot.cpp:192-193): it keeps what the Traj held -- the knot samples of that channel -- cut or zero-extended to
         * the new length */
        double *trig = (double *)malloc(sizeof(double) * (size_t)bo_fwdkin_trig_rows(prob->robot_type, nJ) * (size_t)n);
        if (prob->robot_type == BATOTP_ROBOT_RR) {
            const double *old = p->samp + (size_t)(p->n_theta + 2) * 3 * (size_t)p->n;
            for (int64_t i = 0; i < n; ++i) x[(size_t)(nJ + 2) * n + i] = i < p->n ? old[i] : 0.0;
        }
        bo_fwdkin(prob->robot_type, nJ, x, n, x + (size_t)nJ * n, trig);
        free(trig);
    }
    if (serialTrq) {
        /* ba.cpp:1744-1750, 1791-1827: "clamped" splines through the joint samples, value and derivatives at the END of the
         * previous segment (the values are replaced too), Robot::dynSerial there, torque = a2 + a3 + a4 */
        const double tfact = outRes / smoothFact;
        for (int64_t i = 0; i < n; ++i) { seg[i] = (int32_t)(i - 1); tau[i] = 1; }
        seg[0] = 0; tau[0] = 0;
        bo_path q;
        memset(&q, 0, sizeof(q));
        q.n = n; q.n_theta = nJ;
        q.samp = (double *)malloc(sizeof(double) * 3 * (size_t)nJ * (size_t)n);
        double *c = (double *)calloc((size_t)4 * (size_t)n, sizeof(double));
        for (int j = 0; j < nJ; ++j) {
            memset(c, 0, sizeof(double) * 4 * (size_t)n);
            bo_spline_coeffs(x + (size_t)j * n, n, c, 1);
            double *sp = q.samp + (size_t)j * 3 * n;
            bo_interp1_spline(c, n, seg, tau, n, tfact, sp, sp + n, sp + 2 * n);
            memcpy(x + (size_t)j * n, sp, sizeof(double) * (size_t)n);
        }
        free(c);
        double *a = (double *)malloc(sizeof(double) * 4 * (size_t)nJ * (size_t)n);
        double *a1 = a, *a2 = a + (size_t)nJ * n, *a3 = a + 2 * (size_t)nJ * n, *a4 = a + 3 * (size_t)nJ * n;
        if (p->serial) bo_dyn_serial(p->serial, &q, NULL, a1, a2, a3, a4); /* cos / sin: separate libm calls, as the host twin */
        else bo_dyn_rr(&q, NULL, a1, a2, a3, a4);
        for (int j = 0; j < nJ; ++j)
            for (int64_t i = 0; i < n; ++i)
                x[(size_t)(nJ + nC + j) * n + i] = a2[(size_t)j * n + i] + a3[(size_t)j * n + i] + a4[(size_t)j * n + i];
        free(a); free(q.samp);
    }

    if (cable) {
        /* torque recomputation, parallel mechanism (ba.cpp:1744-1790): every site looks at the END of the previous
         * segment of a natural spline through the output samples themselves */
        const double tfact = outRes / smoothFact, vfact = 1.0 / tfact, afact = vfact * vfact;
        for (int64_t i = 0; i < n; ++i) { seg[i] = (int32_t)(i - 1); tau[i] = 1; }
        seg[0] = 0; tau[0] = 0;
        double *c = (double *)calloc((size_t)4 * (size_t)n, sizeof(double));
        double *cD2 = (double *)malloc(sizeof(double) * (size_t)nC * (size_t)n);
        for (int j = 0; j < nJ + nC; ++j) {
            memset(c, 0, sizeof(double) * 4 * (size_t)n);
            bo_spline_coeffs(x + (size_t)j * n, n, c, 0);
            bo_interp1_spline(c, n, seg, tau, n, tfact, x + (size_t)j * n, d1, d2); /* the values are re-evaluated too */
            if (j >= nJ) memcpy(cD2 + (size_t)(j - nJ) * n, d2, sizeof(double) * (size_t)n);
        }
        (void)vfact; (void)afact;
        free(c);
        for (int64_t i = 0; i < n; ++i) {
            /* Robot::dynCSPR3DOF (robot.cpp:487-517): a2 = -cartD2, a3 = 0, a4 = (0, 0, g) */
            double b[3], th[3], ca[3], A[9], xs[3];
            for (int j = 0; j < 3; ++j) {
                const double a2 = -cD2[(size_t)j * n + i], a3 = 0.0, a4 = (j == 2) ? 9.81 : 0.0;
                b[j] = a2 + a3 + a4;
                ca[j] = x[(size_t)(nJ + j) * n + i];
                th[j] = x[(size_t)j * n + i];
            }
            bo_cspr_setA(prob->pmat, th, ca, A);
            bo_solve(prob, 3, A, b, xs);
            for (int j = 0; j < 3; ++j) x[(size_t)(nJ + nC + j) * n + i] = xs[j];
        }
        free(cD2);
    }
    free(seg); free(tau); free(sOut); free(d1); free(d2); free(tOut); free(tMVC);

    if (smoothFact > 1.5) { /* ba.cpp:1838-1871: joints, torques, Cartesian rows */
        int64_t nDown;
        double *dn = out_smooth_down(x, C, n, smoothFact, &nDown);
        free(x);
        x = dn;
        n = nDown;
    }
    if (reinterp) { /* ba.cpp:1873-1919 */
        int64_t nUser = (int64_t)(int)ceil(tLast / outResUser);
        if (nUser < 4) nUser = 4;
        double *up = out_reinterp(x, C, n, nUser, outResUser);
        free(x);
        x = up;
        n = nUser;
        outRes = outResUser;
    }
    int nCout = nC;
    if (pose) {
        /* BA::q2aaVect (ba.cpp:384-403): rows nJ+3 .. nJ+6 (quaternion) -> nJ+3 .. nJ+5 (axis-angle), the seventh row goes */
        bo_q2aa_rows(x + (size_t)(nJ + 3) * n, n, n);
        nCout = 6;
    }
    *out = x;
    *n_cart_out = nCout;
    *n_trq_out = nT;
    *n_out = n;
    *sres_out = outRes;
    return 0;
}
