/*
 * batotp_oracle_kin.c -- TEST INFRASTRUCTURE ONLY (see batotp_oracle.h).
 *
 * Plain-C restatement of the reference's forward kinematics:
 *   Robot::fwdKinKuka  batotp/robot.cpp:105-174   KUKA LWR IV+ tool point
 *   Robot::fwdKinRR    batotp/robot.cpp:188-202   planar two-link arm
 * split the way the product splits it: the trigonometry of a point (bo_fwdkin_trig) and the arithmetic on those values
 * (bo_fwdkin_from_trig).  The trigonometry is glibc's sincos(): that is what the reference's optimised build calls for
 * "c1=cos(t1); s1=sin(t1);" (robot.cpp:130-136) and for cos(th1) / sin(th1) of the two-link arm, and in this glibc sincos() is
 * not bit-identical to separate cos() / sin() calls.
 *
 * Pinning: the knots and trajectories of the KUKA-LWR-IV, KUKA_cartacc, RR and RR_acc golden cases -- outputs of the
 * reference's prebuilt binary -- are reproduced through these routines (tests/test_oracle_resample.py, test_oracle_output.py).
 * The 3x3 products follow the summation order of the reference binary's Eigen (read off its instruction stream in round 1:
 * rows 0 and 1 of a product left to right, row 2 and the row-times-vector products a0 + (a1 + a2)).
 */
#define _GNU_SOURCE
#include <math.h>
#include <string.h>

#include "batotp_oracle.h"

#define KIN_DEG2RAD (3.14159265358979323846 / 180.0) /* config.h:28 */

int bo_fwdkin_trig_rows(int robot_type, int n_joints)
{
    if (robot_type == BATOTP_ROBOT_KUKA && n_joints == 7) return 14;
    if (robot_type == BATOTP_ROBOT_RR && n_joints == 2) return 4;
    return 0;
}

/* trig[rows][n]: KUKA cos(t_k) k = 0..6, then sin(t_k); RR cos(th1), cos(th1+th2), sin(th1), sin(th1+th2) */
int bo_fwdkin_trig(int robot_type, int n_joints, const double *theta, int64_t n, double *trig)
{
    const int rows = bo_fwdkin_trig_rows(robot_type, n_joints);
    if (!rows) return -1;
    for (int64_t i = 0; i < n; ++i) {
        if (robot_type == BATOTP_ROBOT_KUKA) {
            for (int k = 0; k < 7; ++k) {
                double s, c;
                sincos(KIN_DEG2RAD * theta[(size_t)k * n + i], &s, &c);
                trig[(size_t)k * n + i] = c;
                trig[(size_t)(7 + k) * n + i] = s;
            }
        } else {
            const double th1 = KIN_DEG2RAD * theta[i], th2 = KIN_DEG2RAD * theta[(size_t)n + i];
            double s, c;
            sincos(th1, &s, &c);
            trig[i] = c; trig[(size_t)2 * n + i] = s;
            sincos(th1 + th2, &s, &c);
            trig[(size_t)n + i] = c; trig[(size_t)3 * n + i] = s;
        }
    }
    return 0;
}

static double sum_seq(double a0, double a1, double a2) { return (a0 + a1) + a2; }
static double sum_tree(double a0, double a1, double a2) { return a0 + (a1 + a2); }

static void mul3(const double L[3][3], const double R[3][3], double out[3][3])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const double t0 = L[r][0] * R[0][c], t1 = L[r][1] * R[1][c], t2 = L[r][2] * R[2][c];
            out[r][c] = (r == 2) ? sum_tree(t0, t1, t2) : sum_seq(t0, t1, t2);
        }
}

/* cart rows 0..2 (KUKA) or 0..1 (RR: the third row is only sized by the reference, robot.cpp:192-193) of cart[.][n] */
int bo_fwdkin_from_trig(int robot_type, int n_joints, const double *trig, int64_t n, double *cart)
{
    if (!bo_fwdkin_trig_rows(robot_type, n_joints)) return -1;
    for (int64_t i = 0; i < n; ++i) {
        if (robot_type == BATOTP_ROBOT_RR) {
            const double a1 = .4, a2 = .6;
            cart[i] = a1 * trig[i] + a2 * trig[(size_t)n + i];
            cart[(size_t)n + i] = a1 * trig[(size_t)2 * n + i] + a2 * trig[(size_t)3 * n + i];
            continue;
        }
        double c[7], s[7];
        for (int k = 0; k < 7; ++k) { c[k] = trig[(size_t)k * n + i]; s[k] = trig[(size_t)(7 + k) * n + i]; }
        const double c1 = c[0], c2 = c[1], c3 = c[2], c4 = c[3], c5 = c[4], c6 = c[5], c7 = c[6];
        const double s1 = s[0], s2 = s[1], s3 = s[2], s4 = s[3], s5 = s[4], s6 = s[5], s7 = s[6];
        const double Q12[3][3] = {{c1 * c2, -s1, -c1 * s2}, {c2 * s1, c1, -s1 * s2}, {s2, 0, c2}};
        const double Q34[3][3] = {{c3 * c4, -s3, c3 * s4}, {c4 * s3, c3, s3 * s4}, {-s4, 0, c4}};
        const double Q567[3][3] = {{c5 * c6 * c7 - s5 * s7, -c7 * s5 - c5 * c6 * s7, -c5 * s6},
                                   {c5 * s7 + c6 * c7 * s5, c5 * c7 - c6 * s5 * s7, -s5 * s6},
                                   {c7 * s6, -s6 * s7, c6}};
        double Q1234[3][3], Q[3][3];
        mul3(Q12, Q34, Q1234);
        mul3(Q1234, Q567, Q);
        const double tool[3] = {0, -.08, .545};
        const double a0 = .3105, a1 = .4, a2 = .39;
        const double x1 = a1 * Q12[0][2], y1 = a1 * Q12[1][2], z1 = a1 * Q12[2][2] + a0;
        const double x2 = x1 + a2 * Q1234[0][2], y2 = y1 + a2 * Q1234[1][2], z2 = z1 + a2 * Q1234[2][2];
        cart[i] = x2 + sum_tree(Q[0][0] * tool[0], Q[0][1] * tool[1], Q[0][2] * tool[2]);
        cart[(size_t)n + i] = y2 + sum_tree(Q[1][0] * tool[0], Q[1][1] * tool[1], Q[1][2] * tool[2]);
        cart[(size_t)2 * n + i] = z2 + sum_tree(Q[2][0] * tool[0], Q[2][1] * tool[1], Q[2][2] * tool[2]);
    }
    return 0;
}

int bo_fwdkin(int robot_type, int n_joints, const double *theta, int64_t n, double *cart, double *trig_scratch)
{
    if (bo_fwdkin_trig(robot_type, n_joints, theta, n, trig_scratch) != 0) return -1;
    return bo_fwdkin_from_trig(robot_type, n_joints, trig_scratch, n, cart);
}

/* the trigonometry of Robot::dynRR (robot.cpp:408-419) as the reference's optimised build evaluates it: cos(th1) and
 * cos(th1+th2) are calls of cos(), th2's cosine and sine one sincos().  out = cos(th1), cos(th2), cos(th1+th2), sin(th2) */
static double __attribute__((noinline)) plain_cos(double x) { return cos(x); }
void bo_rr_dyn_trig(double th1, double th2, double out[4])
{
    out[0] = plain_cos(th1);
    sincos(th2, &out[3], &out[1]);
    out[2] = plain_cos(th1 + th2);
}

/* ---- tool poses: axis-angle <-> quaternion (SURVEY.md 8 f-3 / pose paths) ---------------------------------------------
 * aa2q util.cpp:534-555, q2aa util.cpp:562-581, BA::aa2qVect ba.cpp:327-369 (successive quaternions kept on one hemisphere),
 * BA::q2aaVect ba.cpp:384-403.  Trigonometry as the reference's optimised build calls it: sin and cos of the half angle are
 * one sincos(), the inverse is one atan2(). */
void bo_aa2q(const double aa[3], double q[4])
{
    const double theta = sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    if (theta < 1e-6) { q[0] = 1.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0; return; }
    double sh, ch;
    sincos(0.5 * theta, &sh, &ch);
    q[0] = ch;
    for (int i = 0; i < 3; ++i) q[i + 1] = aa[i] * sh / theta;
}

void bo_q2aa(const double q[4], double aa[3])
{
    const double norme = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (norme < 1e-6) { aa[0] = 0.0; aa[1] = 0.0; aa[2] = 0.0; return; }
    const double theta = 2.0 * atan2(norme, q[0]) / norme;
    for (int i = 0; i < 3; ++i) aa[i] = theta * q[i + 1];
}

/* rows[4][n] hold (rx, ry, rz, -) on entry (row stride `stride`) and the quaternions on return */
void bo_aa2q_rows(double *rows, int64_t stride, int64_t n)
{
    double prev[4], q[4], aa[3];
    aa[0] = rows[0]; aa[1] = rows[stride]; aa[2] = rows[2 * stride];
    bo_aa2q(aa, prev);
    for (int64_t i = 0; i < n; ++i) {
        aa[0] = rows[i]; aa[1] = rows[stride + i]; aa[2] = rows[2 * stride + i];
        bo_aa2q(aa, q);
        double qdir = 0;
        for (int j = 0; j < 4; ++j) qdir += q[j] * prev[j];
        if (qdir < 0.0) for (int j = 0; j < 4; ++j) q[j] = -q[j];
        for (int j = 0; j < 4; ++j) { prev[j] = q[j]; rows[(size_t)j * stride + i] = q[j]; }
    }
}

/* rows[4][n] hold quaternions on entry and (rx, ry, rz) in their first three rows on return */
void bo_q2aa_rows(double *rows, int64_t stride, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        const double q[4] = {rows[i], rows[stride + i], rows[2 * stride + i], rows[3 * stride + i]};
        double aa[3];
        bo_q2aa(q, aa);
        for (int j = 0; j < 3; ++j) rows[(size_t)j * stride + i] = aa[j];
    }
}
