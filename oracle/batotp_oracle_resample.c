/*
 * batotp_oracle_resample.c -- TEST INFRASTRUCTURE ONLY (see batotp_oracle.h).
 *
 * Plain-C restatement of the path resampling that precedes the hot path, for the
 * path kinds the device resampler covers: JOINT paths of a robot without kinematic model, JOINT paths of the
 * robots with forward kinematics (KUKA, RR: batotp_oracle_kin.c) and CART paths of the 3-cable
 * robot.  It is the checker of batotp_hip_resample.
 *
 * Pinning: tests/test_oracle_resample.py compares the knots it produces with the knots.npz
 * fixtures under tests/golden/ -- the knots behind the s-sdot / trajectory outputs that are
 * byte-identical to the reference binary's (oracle/README.md).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "batotp_oracle.h"

/* stage observer of the calling thread (batotp_oracle.h) */
static __thread bo_resample_stage_fn rs_obs = NULL;
static __thread void *rs_obs_user = NULL;
void bo_resample_set_stage_observer(bo_resample_stage_fn fn, void *user) { rs_obs = fn; rs_obs_user = user; }
#define RS_STAGE(k, ptr, cnt) do { if (rs_obs) rs_obs(rs_obs_user, (k), (ptr), (int64_t)(cnt)); } while (0)

typedef struct rs_traj {
    int     nJ, nC, C;
    int64_t n;
    double *x;    /* [C][n] channel-major: theta rows then cart rows */
    double  sres;
} rs_traj;

/* remClosePts (util.cpp:452-524) on the driving channels [c0, c0+cN) */
static void rs_rem_close(rs_traj *t, int c0, int cN, double thresh)
{
    const int64_t n0 = t->n;
    int64_t n = n0;
    char *rem = (char *)calloc((size_t)n0, 1);
    const double t2 = thresh * thresh;
    for (;;) {
        int found = 0;
        for (int64_t i = 1; i < n; ++i) {
            double sumsq = 0;
            for (int j = 0; j < cN; ++j) {
                const double dx = t->x[(c0 + j) * n0 + i] - t->x[(c0 + j) * n0 + i - 1];
                sumsq += dx * dx;
            }
            if (sumsq < t2 && !rem[i - 1]) { rem[i] = 1; found = 1; }
        }
        if (rem[n - 1] && n > 2) { rem[n - 1] = 0; rem[n - 2] = 1; rem[n - 3] = 0; }
        if (!found) break;
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (rem[i]) continue;
            for (int c = 0; c < t->C; ++c) t->x[c * n0 + k] = t->x[c * n0 + i];
            ++k;
        }
        n = k;
        memset(rem, 0, (size_t)n);
    }
    free(rem);
    if (n != n0) { /* repack rows to stride n */
        for (int c = 1; c < t->C; ++c) memmove(t->x + c * n, t->x + c * n0, sizeof(double) * (size_t)n);
        t->n = n;
    }
}

/* BA::interpTrajLinear (ba.cpp:2768-2794) for a stage of 2 or 3 points: every row onto 4 evenly spaced nodes by linear
 * interpolation (findInterpSegs, spline.cpp:56-99; interp1linear, spline.cpp:108-120), the spacing stretched accordingly */
static void rs_stretch_to_four(rs_traj *t)
{
    const int64_t nOld = t->n, nNew = 4;
    double gOld[3], gNew[4], *y = (double *)calloc((size_t)t->C * (size_t)nNew, sizeof(double));
    const double cOld = 1.0 / (double)(nOld - 1), cNew = 1.0 / (double)(nNew - 1);
    for (int64_t k = 0; k < nOld; ++k) gOld[k] = cOld * (double)k;
    for (int64_t k = 0; k < nNew; ++k) gNew[k] = cNew * (double)k;
    int64_t cur = 0;
    for (int64_t i = 0; i < nNew; ++i) {
        while (!(gNew[i] < gOld[cur + 1] || cur == nOld - 2)) ++cur;
        const double tau = (gNew[i] - gOld[cur]) / (gOld[cur + 1] - gOld[cur]);
        for (int c = 0; c < t->C; ++c) {
            const double a = t->x[c * nOld + cur], b = t->x[c * nOld + cur + 1];
            y[c * nNew + i] = a + (b - a) * tau;
        }
    }
    free(t->x);
    t->x = y;
    t->sres = t->sres * (double)(nOld - 1) / (double)(nNew - 1);
    t->n = nNew;
}

/* smooth (util.cpp:263-290): centred moving average, shrinking windows at both ends */
static void rs_smooth(double *x, int64_t n, int w)
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

/* input decimation and smoothing (ba.cpp:195-242) of the driving rows [c0, c0+cN); the other rows only shrink (they are
 * zero or recomputed by the kinematics).  Note: both smoothing calls of the reference pass _inputDecimFact as window. */
static void rs_decimate_smooth(rs_traj *t, int c0, int cN, int decim, int smooth_window)
{
    if (decim > 1) {
        const int64_t nIn = t->n, nOut = (nIn - 1) / decim + 1;
        for (int j = 0; j < cN; ++j) rs_smooth(t->x + (c0 + j) * nIn, nIn, decim);
        double *y = (double *)calloc((size_t)t->C * (size_t)nOut, sizeof(double));
        for (int c = 0; c < t->C; ++c) { /* decimate (util.cpp:347-356): every decim-th sample, always the last one */
            const double *src = t->x + c * nIn;
            double *dst = y + c * nOut;
            for (int64_t i = 0; i < nOut; ++i) dst[i] = src[decim * i];
            if (decim * (nOut - 1) + 1 != nIn) dst[nOut - 1] = src[nIn - 1];
        }
        free(t->x);
        t->x = y;
        t->n = nOut;
        t->sres *= decim;
    }
    if (smooth_window > 1)
        for (int j = 0; j < cN; ++j) rs_smooth(t->x + (c0 + j) * t->n, t->n, decim);
}

/* Robot::invKinCSPR3DOF (robot.cpp:243-278) */
static void rs_invkin_cspr(rs_traj *t, const double pmat[9])
{
    const int64_t n = t->n;
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            double sumSQ = 0.0;
            for (int r = 0; r < 3; ++r) {
                const double dlt = t->x[(t->nJ + r) * n + i] - pmat[r * 3 + k];
                sumSQ += dlt * dlt;
            }
            t->x[k * n + i] = sqrt(sumSQ);
        }
}

/* Robot::fwdKin (robot.cpp:73-96) on the stage's joint rows: KUKA writes the three Cartesian rows, the two-link arm the first two */
static void rs_fwdkin(rs_traj *t, int robot_type)
{
    double *trig = (double *)malloc(sizeof(double) * (size_t)bo_fwdkin_trig_rows(robot_type, t->nJ) * (size_t)t->n);
    bo_fwdkin(robot_type, t->nJ, t->x, t->n, t->x + (size_t)t->nJ * t->n, trig);
    free(trig);
}

typedef struct rs_scale { double sLast, sResNew, teach, thetaF, cartF, sResi; } rs_scale;

/* what the automatic integration resolution rewrites per path and carries from the first adjust_s pass into the second
 * (BA::_integRes, _sWeights[1..2], _scaleType are members of the BA object; the Cartesian resolution of a pass is a local) */
typedef struct rs_auto { int on; double integRes, sw1, sw2; int scaleType; } rs_auto;

static double rs_min(double a, double b) { return (b < a) ? b : a; } /* std::min / std::max of libstdc++ (NaN behaviour included) */
static double rs_max(double a, double b) { return (a < b) ? b : a; }

/* ba.cpp:493-556 after the arc lengths are known: path-resolution integration step, s weights, scale type, Cartesian resolution */
static void rs_auto_rule(const bo_resample_params *prm, rs_auto *au, double thetaLast, double cartLast, double minCartPerTheta, double *cartRes)
{
    const double kDeg = 180.0 / 3.14159265358979323846; /* _RAD2DEG, config.h:29 */
    if (cartLast < *cartRes && au->scaleType == 2) { /* no Cartesian motion to speak of: s follows the joints */
        au->sw1 = au->sw1 + au->sw2;
        au->sw2 = 0;
        au->scaleType = 1;
    }
    const double weightIn = au->sw1 + au->sw2;
    double cartRat = 500.0 * cartLast;
    double thetaRat = thetaLast;
    if (!prm->degrees) thetaRat *= kDeg;
    const double lo = 0.004, hi = 0.2, K = 0.0003;
    double step = K * prm->cart_acc_max / prm->cart_vel_max;
    for (int j = 0; j < prm->n_joints; ++j) step = rs_max(step, K * prm->jnt_acc_max[j] / prm->jnt_vel_max[j]);
    step = rs_min(step, hi);
    const double ratio = cartRat / thetaRat;
    double byJoints = hi * ratio * ratio;
    double byWindow = hi * minCartPerTheta * minCartPerTheta;
    byWindow = rs_max(byWindow, 0.016);
    byJoints = rs_min(byJoints, byWindow);
    if (byJoints < step) step = byJoints; /* orientation movement dominates */
    step = rs_max(step, lo);
    au->integRes = step;
    const double rescale = weightIn / (cartRat + thetaRat);
    cartRat *= rescale;
    thetaRat *= rescale;
    if (thetaRat > au->sw1) { au->sw1 = thetaRat; au->sw2 = cartRat; }
    if (au->sw2 > 0) *cartRes = rs_min(*cartRes, *cartRes * au->sw2 / au->sw1);
}

/* first half of adjust_s (ba.cpp:430-590): arc lengths, factors, sC.  0 ok / status bit */
static unsigned rs_arclen(const bo_resample_params *prm, const rs_traj *t, int special, double *sC, rs_scale *sc, rs_auto *au)
{
    const int64_t n = t->n;
    double cartRes = special ? prm->cart_norm_res : prm->cart_norm_res2;
    const double thetaRes = special ? prm->theta_norm_res : prm->theta_norm_res2;
    double *th = (double *)calloc((size_t)n, sizeof(double)), *ca = (double *)calloc((size_t)n, sizeof(double));
    /* ba.cpp:441-446, 462-470: the smallest Cartesian advance per joint-space advance over windows of 5 degrees */
    double minCartPerTheta = 1.0 / prm->quad_rad_thresh, window = 5.0, thMark = 0, caMark = 0;
    if (!prm->degrees) window *= 3.14159265358979323846 / 180.0;
    for (int64_t i = 0; i < n - 1; ++i) {
        double sq = 0;
        for (int j = 0; j < t->nJ; ++j) {
            const double d = t->x[j * n + i + 1] - t->x[j * n + i];
            sq += d * d;
        }
        th[i + 1] = th[i] + sqrt(sq);
        sq = 0;
        for (int j = 0; j < 3; ++j) {
            const double d = t->x[(t->nJ + j) * n + i + 1] - t->x[(t->nJ + j) * n + i];
            sq += d * d;
        }
        ca[i + 1] = ca[i] + sqrt(sq);
        if (au->on) {
            const double dTh = th[i + 1] - thMark, dCa = ca[i + 1] - caMark;
            if (dTh > window) {
                minCartPerTheta = rs_min(minCartPerTheta, 3.0 * dCa / dTh);
                thMark = th[i + 1];
                caMark = ca[i + 1];
            }
        }
    }
    unsigned st = 0;
    if (th[n - 1] < thetaRes) st = BATOTP_RS_IDENTICAL; /* ba.cpp:484-488 */
    if (!st) {
        const double sResi = t->sres, last = (double)(n - 1);
        double sLast = 0, sResNew = 0;
        if (au->on) rs_auto_rule(prm, au, th[n - 1], ca[n - 1], minCartPerTheta, &cartRes);
        switch (au->scaleType) { /* ba.cpp:558-572 */
        case 0: sLast = sResi * last; sResNew = sResi; break;
        case 1: sLast = th[n - 1]; sResNew = thetaRes; break;
        default: sLast = ca[n - 1]; sResNew = cartRes; break;
        }
        double cartF = 0;
        if (ca[n - 1] >= cartRes) cartF = au->sw2 * sLast / ca[n - 1];
        const double teach = prm->s_weights[0] * sLast / (sResi * last);
        const double thetaF = au->sw1 * sLast / th[n - 1];
        for (int64_t i = 0; i < n; ++i) sC[i] = teach * sResi * (double)i + thetaF * th[i] + cartF * ca[i];
        sc->sLast = sLast; sc->sResNew = sResNew; sc->teach = teach; sc->thetaF = thetaF; sc->cartF = cartF; sc->sResi = sResi;
    }
    free(th); free(ca);
    return st;
}

/* spline coefficients of every channel: coef[c] -> [4][n] */
static double *rs_all_coeffs(const rs_traj *t)
{
    double *coef = (double *)calloc((size_t)t->C * 4 * (size_t)t->n, sizeof(double));
    for (int c = 0; c < t->C; ++c) bo_spline_coeffs(t->x + c * t->n, t->n, coef + (size_t)c * 4 * t->n, 0);
    return coef;
}

/* interpSpecial (ba.cpp:651-781) */
static unsigned rs_special(const bo_resample_params *prm, rs_traj *t, const double *sC, const rs_scale *sc)
{
    const int64_t n = t->n;
    const int nJ = t->nJ, nC = t->nC, C = t->C;
    const int cartEval = (prm->flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)) != 0;
    double *coef = rs_all_coeffs(t);
    RS_STAGE(2, coef, (int64_t)C * 4 * n);
    int64_t chunk = (int64_t)ceil(sc->sLast / sc->sResNew) + 1; /* ba.cpp:666-667 */
    if (chunk < 4) chunk = 4;
    int64_t cap = chunk;
    double *out = (double *)calloc((size_t)cap * C, sizeof(double)); /* point-major rows */
    double cartpt[BATOTP_MAX_CART] = {0};
    for (int c = 0; c < C; ++c) out[c] = t->x[c * n];
    double sPrv = 0, prvDs = 0;
    int64_t newPt = 1, oldPt = 1, seg = 0;
    int done = 0;
    while (!done) {
        double thSq = 0, caSq = 0;
        for (int j = 0; j < nJ; ++j) {
            const double d = t->x[j * n + oldPt] - out[(newPt - 1) * C + j];
            thSq += d * d;
        }
        for (int j = 0; j < 3; ++j) {
            const double d = t->x[(nJ + j) * n + oldPt] - out[(newPt - 1) * C + nJ + j];
            caSq += d * d;
        }
        const double ds = sc->teach * sc->sResi * (double)oldPt + sc->thetaF * sqrt(thSq) + sc->cartF * sqrt(caSq);
        if (ds > sc->sResNew) {
            const double sNew = sPrv + sc->sResNew - prvDs;
            prvDs = 0;
            sPrv = sNew;
            if (sNew > sC[n - 1]) done = 1;
            if (!done) {
                /* evalSplinePartials -> updateCurSeg (ba.cpp:1617-1652) from the cached segment */
                double s0;
                for (;;) {
                    s0 = sC[seg];
                    if (sNew >= s0 && sNew <= sC[seg + 1]) break;
                    int moved = 0;
                    if (sNew > s0) { if (seg >= n - 2) { seg = n - 2; break; } ++seg; moved = 1; }
                    if (sNew < s0) { if (seg <= 0) { seg = 0; break; } --seg; moved = 1; }
                    if (!moved) break;
                }
                const double tau = (sNew - s0) / (sC[seg + 1] - s0);
                const double tau2 = tau * tau, tau3 = tau2 * tau;
                double *o = out + newPt * C;
                for (int j = 0; j < nJ; ++j) {
                    const double *k = coef + (size_t)j * 4 * n;
                    o[j] = k[3 * n + seg] * tau3 + k[2 * n + seg] * tau2 + k[1 * n + seg] * tau + k[seg];
                }
                if (cartEval)
                    for (int j = 0; j < nC; ++j) {
                        const double *k = coef + (size_t)(nJ + j) * 4 * n;
                        cartpt[j] = k[3 * n + seg] * tau3 + k[2 * n + seg] * tau2 + k[1 * n + seg] * tau + k[seg];
                    }
                for (int j = 0; j < nC; ++j) o[nJ + j] = cartpt[j];
                oldPt = seg + 1;
                ++newPt;
                if (newPt == cap) {
                    cap += chunk;
                    out = (double *)realloc(out, sizeof(double) * (size_t)cap * C);
                }
            }
        } else if (oldPt == n - 1) {
            done = 1;
        } else {
            prvDs = ds;
            sPrv = sC[oldPt];
            ++oldPt;
        }
    }
    for (int c = 0; c < C; ++c) out[newPt * C + c] = t->x[c * n + n - 1];
    const int64_t nNew = newPt + 1;
    free(coef);
    free(t->x);
    t->x = (double *)malloc(sizeof(double) * (size_t)nNew * C);
    for (int64_t i = 0; i < nNew; ++i)
        for (int c = 0; c < C; ++c) t->x[c * nNew + i] = out[i * C + c];
    free(out);
    t->n = nNew;
    t->sres = sc->sResNew;
    if (nNew < 4) rs_stretch_to_four(t); /* ba.cpp:773-774 */
    return 0;
}

/* second half of adjust_s "regularInterp" + evalSplineFullTraj (ba.cpp:601-613, 790-863) */
static unsigned rs_regular(rs_traj *t, const double *sC, const rs_scale *sc)
{
    const int64_t n = t->n;
    const int C = t->C;
    const double oldRes = sc->sLast / (double)(n - 1); /* traj.sres, ba.cpp:585 */
    for (int64_t i = 1; i < n; ++i)
        if (sC[i] - sC[i - 1] < 1e-12 * oldRes) return BATOTP_RS_SMALL_STEP;
    int64_t nNew = (int64_t)ceil(oldRes / sc->sResNew * (double)(n - 1)) + 1;
    if (nNew < 4) nNew = 4;
    const double newRes = oldRes * (double)(n - 1) / (double)(nNew - 1);
    double *sites = (double *)malloc(sizeof(double) * (size_t)nNew);
    const double sScale = sC[n - 1] / (double)(nNew - 1);
    for (int64_t i = 0; i < nNew; ++i) sites[i] = sScale * (double)i;
    int32_t *seg = (int32_t *)malloc(sizeof(int32_t) * (size_t)nNew);
    double *tau = (double *)malloc(sizeof(double) * (size_t)nNew);
    unsigned st = 0;
    if (bo_find_interp_segs(sC, n, sites, nNew, seg, tau) != 0) st = BATOTP_RS_SEG_ERROR;
    if (!st) {
        double *coef = rs_all_coeffs(t);
        RS_STAGE(6, coef, (int64_t)C * 4 * n);
        double *y = (double *)malloc(sizeof(double) * (size_t)nNew * C);
        double *d1 = (double *)malloc(sizeof(double) * (size_t)nNew), *d2 = (double *)malloc(sizeof(double) * (size_t)nNew);
        for (int c = 0; c < C; ++c) bo_interp1_spline(coef + (size_t)c * 4 * n, n, seg, tau, nNew, oldRes, y + c * nNew, d1, d2);
        free(d1); free(d2); free(coef);
        free(t->x);
        t->x = y;
        t->n = nNew;
        t->sres = newRes;
    }
    free(sites); free(seg); free(tau);
    return st;
}

int bo_resample(const bo_resample_params *prm, int64_t n_in, const double *x, double sres_in, double **y_out, int64_t *n_out,
                double *sres_out, uint32_t *status)
{
    double au[5];
    return bo_resample_auto(prm, n_in, x, sres_in, y_out, n_out, sres_out, status, au);
}

/* the same, also reporting what the automatic integration resolution left: auto_out = integ_res (NaN-able; 0 when the rule
 * is off), s_weights[0..2], scale_type */
int bo_resample_auto(const bo_resample_params *prm, int64_t n_in, const double *x, double sres_in, double **y_out, int64_t *n_out,
                     double *sres_out, uint32_t *status, double auto_out[5])
{
    const int joint = prm->path_type == BATOTP_PATH_JOINT && prm->robot_type == BATOTP_ROBOT_GENJNT;
    const int cable = prm->path_type == BATOTP_PATH_CART && prm->robot_type == BATOTP_ROBOT_CSPR3DOF && prm->n_joints == 3;
    /* JOINT path of a robot with forward kinematics (KUKA, RR): SURVEY.md 8 f-3 */
    const int kin = prm->path_type == BATOTP_PATH_JOINT && bo_fwdkin_trig_rows(prm->robot_type, prm->n_joints) != 0;
    const int cartOn = (prm->flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)) != 0;
    /* joints and Cartesian rows taught together (ba.cpp:184-192, 245-262: nothing is recomputed, both channel sets are resampled as
     * taught).  With 6 Cartesian rows they are tool poses (the UR5 example): 6 rows in, 7 out (aa2qVect: quaternions); any other count
     * (tool positions without orientations, ...) goes through unchanged */
    const int both = prm->path_type == BATOTP_PATH_BOTH;
    const int pose = both && prm->n_cart == 6;
    if ((!joint && !cable && !kin && !both) || n_in < 4 || prm->n_cart < 3) return -1;
    if (prm->s_weights[1] + prm->s_weights[2] < 1e-8) return -1; /* ba.cpp:416: nothing to do */
    rs_traj t;
    t.nJ = prm->n_joints; t.nC = prm->n_cart + (pose ? 1 : 0); t.C = t.nJ + t.nC; t.n = n_in; t.sres = sres_in;
    t.x = (double *)calloc((size_t)n_in * t.C, sizeof(double));
    memcpy(t.x, x, sizeof(double) * (size_t)n_in * (t.nJ + prm->n_cart));
    if (joint) memset(t.x + (size_t)t.nJ * n_in, 0, sizeof(double) * (size_t)t.nC * n_in); /* ba.cpp:258-262 */
    unsigned st = 0;
    if (cable) rs_rem_close(&t, t.nJ, t.nC, prm->cart_thresh); /* ba.cpp:166-175 */
    else rs_rem_close(&t, 0, t.nJ, prm->jnt_thresh);
    if (t.n < 2) st |= BATOTP_RS_TOO_SHORT; /* "less than one site after remClosePts", ba.cpp:176-181 */
    else if (t.n < 4) rs_stretch_to_four(&t); /* ba.cpp:182-183 */
    if (!st && pose) bo_aa2q_rows(t.x + (size_t)(t.nJ + 3) * t.n, t.n, t.n); /* ba.cpp:186-193 */
    if (!st && (prm->input_decim_fact > 1 || prm->smooth_window > 1)) {
        /* driving rows: joints (JOINT), Cartesian rows (CART), both sets (BOTH), ba.cpp:197-241 */
        rs_decimate_smooth(&t, cable ? t.nJ : 0, cable ? t.nC : (both ? t.C : t.nJ), prm->input_decim_fact > 1 ? prm->input_decim_fact : 1, prm->smooth_window);
        if (t.n < 4) st |= BATOTP_RS_TOO_SHORT;
    }
    if (!st && cable) rs_invkin_cspr(&t, prm->pmat);
    if (!st && kin && cartOn) rs_fwdkin(&t, prm->robot_type); /* ba.cpp:247-256; otherwise the Cartesian rows stay as loaded */
    rs_auto au;
    au.on = (prm->flags & BATOTP_RS_AUTO_INTEG_RES) != 0;
    au.integRes = 0; au.sw1 = prm->s_weights[1]; au.sw2 = prm->s_weights[2]; au.scaleType = prm->scale_type;
    for (int pass = 0; pass < 2 && !st; ++pass) {
        double *sC = (double *)malloc(sizeof(double) * (size_t)t.n);
        rs_scale sc;
        RS_STAGE(pass ? 4 : 0, t.x, (int64_t)t.C * t.n);
        st |= rs_arclen(prm, &t, pass == 0, sC, &sc, &au);
        if (!st) RS_STAGE(pass ? 5 : 1, sC, t.n);
        if (!st) st |= pass == 0 ? rs_special(prm, &t, sC, &sc) : rs_regular(&t, sC, &sc);
        free(sC);
        if (!st && cable) rs_invkin_cspr(&t, prm->pmat); /* ba.cpp:630 */
        if (!st && kin) rs_fwdkin(&t, prm->robot_type);   /* ba.cpp:626-628: after either pass, whatever the constraints */
        if (!st) RS_STAGE(pass ? 7 : 3, t.x, (int64_t)t.C * t.n);
    }
    auto_out[0] = au.integRes; auto_out[1] = prm->s_weights[0]; auto_out[2] = au.sw1; auto_out[3] = au.sw2; auto_out[4] = (double)au.scaleType;
    *status = st;
    if (st) {
        free(t.x);
        *y_out = (double *)calloc((size_t)4 * t.C, sizeof(double));
        *n_out = 4;
        *sres_out = 0.0;
    } else {
        *y_out = t.x;
        *n_out = t.n;
        *sres_out = t.sres;
    }
    return 0;
}
