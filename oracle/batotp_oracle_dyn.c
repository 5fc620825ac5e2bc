/*
 * batotp_oracle_dyn.c -- TEST INFRASTRUCTURE ONLY (see batotp_oracle.h).
 *
 * Serial-chain dynamics coefficients for BASELINE config 3 (7-DOF arm with torque limits).
 *
 * PARITY UNPINNED for the arm model itself: the reference has no dynamics model for any serial robot
 * but the two-link arm (Robot::dynSerial, /root/reference/batotp/robot.cpp:349-360; KUKA with
 * isTrqConOn = 1 segfaults).  What IS pinned: the machinery around it -- the form
 * tau = a1 sddot + a2 sdot^2 + a3 sdot + a4 (robot.cpp:368-372), the spline build of a1..a4
 * (ba.cpp:940-946) and the serial torque branch of the sweep (ba.cpp:1495-1509) through the RR
 * golden cases -- and this file's recursion itself against Robot::dynRR (robot.cpp:377-431): with
 * the two point masses of that model as the link table, bo_dyn_serial reproduces dynRR's a1..a4
 * (tests/test_oracle_dyn.py; tolerance, the operation order differs).
 *
 * Algorithm (published: Luh, Walker, Paul 1980; Featherstone, "Rigid Body Dynamics Algorithms", ch. 5):
 * recursive Newton-Euler in link coordinates, revolute joints, zero-aligned frames (every link frame
 * coincides with the base frame at q = 0; joint i turns about `axis`).  With q = q(s):
 *   qdot = q' sdot, qddot = q' sddot + q'' sdot^2   =>
 *   tau = M q' sddot + (M q'' + C(q, q') q') sdot^2 + fv q' sdot + g(q)
 *   a1 = RNEA(q, 0, q', no gravity);  a2 = RNEA(q, q', q'', no gravity);  a3 = fv .* q';
 *   a4 = RNEA(q, 0, 0, gravity)
 * Arithmetic contract as everywhere: binary64, no contraction, the expression order written here
 * (the HIP kernel k_dyn_serial restates exactly this order).
 */
#include "batotp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cos and sin through separate libm calls: side by side the compiler fuses them into sincos(), whose results are
 * not bit-identical to cos() / sin() in this glibc; the trig policy of the chain model is libm cos and libm sin */
static double __attribute__((noinline)) libm_cos(double x) { return cos(x); }
static double __attribute__((noinline)) libm_sin(double x) { return sin(x); }

static void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* rotation by the angle with cosine c and sine s about the unit axis a (Rodrigues), row-major,
 * maps link coordinates to parent coordinates */
static void rot_axis(const double *a, double c, double s, double *R)
{
    const double omc = 1.0 - c;
    R[0] = omc * a[0] * a[0] + c;
    R[1] = omc * a[0] * a[1] - s * a[2];
    R[2] = omc * a[0] * a[2] + s * a[1];
    R[3] = omc * a[1] * a[0] + s * a[2];
    R[4] = omc * a[1] * a[1] + c;
    R[5] = omc * a[1] * a[2] - s * a[0];
    R[6] = omc * a[2] * a[0] - s * a[1];
    R[7] = omc * a[2] * a[1] + s * a[0];
    R[8] = omc * a[2] * a[2] + c;
}

static void mat_v(const double *R, const double *v, double *o)
{
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

static void matT_v(const double *R, const double *v, double *o)
{
    o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}

static void inertia_v(const double *I, const double *v, double *o)
{
    /* I = Ixx Iyy Izz Ixy Ixz Iyz */
    o[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
    o[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
    o[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
}

/* One pass of the recursion at one configuration: joint cosines / sines cq, sq, joint rates qd and
 * accelerations qdd (radians), base acceleration a0 (= -gravity, or 0).  tau[n_links]. */
void bo_rnea(const batotp_serial_model *m, const double *cq, const double *sq, const double *qd,
             const double *qdd, const double *a0, double *tau)
{
    const int n = m->n_links;
    double F[BATOTP_MAX_LINKS][3], Nn[BATOTP_MAX_LINKS][3];
    double w[3] = {0, 0, 0}, wd[3] = {0, 0, 0}, a[3];
    double fc[3] = {0, 0, 0}, nc[3] = {0, 0, 0}; /* force / moment handed down by the child, in this link's coordinates */
    int i, k;
    a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2];

    /* outward: velocities and accelerations, link forces */
    for (i = 0; i < n; i++) {
        const batotp_serial_link *L = &m->link[i];
        double R[9], t1[3], t2[3], t3[3], wp[3], wdp[3], ap[3], zq[3], ac[3], Iw[3], Iwd[3];
        rot_axis(L->axis, cq[i], sq[i], R);
        /* acceleration of this joint's origin, parent coordinates: a + wd x off + w x (w x off) */
        cross3(wd, L->off, t1);
        cross3(w, L->off, t2);
        cross3(w, t2, t3);
        for (k = 0; k < 3; k++) t1[k] = a[k] + t1[k] + t3[k];
        matT_v(R, t1, ap);
        matT_v(R, w, wp);
        matT_v(R, wd, wdp);
        for (k = 0; k < 3; k++) zq[k] = L->axis[k] * qd[i];
        cross3(wp, zq, t2);
        for (k = 0; k < 3; k++) {
            w[k] = wp[k] + zq[k];
            wd[k] = wdp[k] + L->axis[k] * qdd[i] + t2[k];
            a[k] = ap[k];
        }
        /* centre of mass: a + wd x com + w x (w x com) */
        cross3(wd, L->com, t1);
        cross3(w, L->com, t2);
        cross3(w, t2, t3);
        for (k = 0; k < 3; k++) ac[k] = a[k] + t1[k] + t3[k];
        for (k = 0; k < 3; k++) F[i][k] = L->mass * ac[k];
        inertia_v(L->inertia, wd, Iwd);
        inertia_v(L->inertia, w, Iw);
        cross3(w, Iw, t1);
        for (k = 0; k < 3; k++) Nn[i][k] = Iwd[k] + t1[k];
    }

    /* inward: joint forces and moments */
    for (i = n - 1; i >= 0; i--) {
        const batotp_serial_link *L = &m->link[i];
        double f[3], nn[3], t1[3], t2[3];
        cross3(L->com, F[i], t1);
        if (i + 1 < n) cross3(m->link[i + 1].off, fc, t2);
        else { t2[0] = 0; t2[1] = 0; t2[2] = 0; }
        for (k = 0; k < 3; k++) {
            f[k] = F[i][k] + fc[k];
            nn[k] = Nn[i][k] + nc[k] + t1[k] + t2[k];
        }
        tau[i] = L->axis[0] * nn[0] + L->axis[1] * nn[1] + L->axis[2] * nn[2];
        /* hand down to the parent: rotate into its coordinates */
        {
            double R[9];
            rot_axis(L->axis, cq[i], sq[i], R);
            mat_v(R, f, fc);
            mat_v(R, nn, nc);
        }
    }
}

/* a1..a4 [n_links][n] at the knots of p (theta, theta', theta'' = p->samp); trig: optional
 * [2*n_links][n] host cosines then sines of the joint angles in radians (NULL -> libm here). */
int bo_dyn_serial(const batotp_serial_model *m, const bo_path *p, const double *trig,
                  double *a1, double *a2, double *a3, double *a4)
{
    const int64_t n = p->n;
    const int nl = m->n_links;
    const double kDeg2Rad = 3.14159265358979323846 / 180.0; /* config.h:27-28 */
    const double unit = m->degrees ? kDeg2Rad : 1.0;
    const double zero[3] = {0, 0, 0};
    double g0[3];
    int64_t i;
    int j;
    if (nl < 1 || nl > BATOTP_MAX_LINKS || nl != p->n_theta) return -1;
    g0[0] = -m->gravity[0]; g0[1] = -m->gravity[1]; g0[2] = -m->gravity[2];
    for (i = 0; i < n; i++) {
        double cq[BATOTP_MAX_LINKS], sq[BATOTP_MAX_LINKS], q1[BATOTP_MAX_LINKS], q2[BATOTP_MAX_LINKS], z[BATOTP_MAX_LINKS];
        double t1[BATOTP_MAX_LINKS], t2[BATOTP_MAX_LINKS], t4[BATOTP_MAX_LINKS];
        for (j = 0; j < nl; j++) {
            const double *sp = p->samp + (int64_t)j * 3 * n;
            const double q = unit * sp[i];
            q1[j] = unit * sp[n + i];
            q2[j] = unit * sp[2 * n + i];
            z[j] = 0.0;
            if (trig) { cq[j] = trig[(int64_t)j * n + i]; sq[j] = trig[(int64_t)(nl + j) * n + i]; }
            else { cq[j] = libm_cos(q); sq[j] = libm_sin(q); }
        }
        bo_rnea(m, cq, sq, z, q1, zero, t1);
        bo_rnea(m, cq, sq, q1, q2, zero, t2);
        bo_rnea(m, cq, sq, z, z, g0, t4);
        for (j = 0; j < nl; j++) {
            a1[(int64_t)j * n + i] = t1[j];
            a2[(int64_t)j * n + i] = t2[j];
            a3[(int64_t)j * n + i] = m->link[j].fv * q1[j];
            a4[(int64_t)j * n + i] = t4[j];
        }
    }
    return 0;
}
