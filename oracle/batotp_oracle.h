/*
 * batotp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the batotp hot path (per-knot precompute, per-knot bisection and the
 * reverse/forward sweep), written from scratch against the reference sources, every function
 * citing the reference file:line it follows.  It exists to CHECK the HIP kernels:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call it.
 * The product (batotp_amd/csrc, batotp_amd/host) never includes or links anything in oracle/.
 *
 * Pinning: see oracle/README.md -- the restatement is checked against the reference itself run in
 * the build container (prebuilt /root/reference/bin/batest, real Eigen) on the five shipped
 * examples and on synthetic inputs; fixtures are committed under tests/golden/.
 *
 * Arithmetic contract: IEEE-754 binary64, no FMA contraction (-ffp-contract=off), the reference's
 * operation order.
 */
#ifndef BATOTP_ORACLE_H
#define BATOTP_ORACLE_H

#include <stdint.h>
#include "batotp_hip.h" /* POD batotp_problem / batotp_path_result / flag macros only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bo_path {
    int64_t n;          /* knots (traj.nPtsC == traj.nPts at sweep entry)                 */
    int32_t n_theta;    /* n_joints                                                       */
    int32_t n_cart;     /* cart channels                                                  */
    int32_t dyn_dim;    /* 0 if torque constraints off                                    */
    int32_t n_ch;       /* n_theta + n_cart + 4*dyn_dim                                   */
    int32_t parallel_now; /* _isParallelMech as seen by the sweep (cleared by par2ser)    */
    double  sres_c;     /* traj.sresC                                                     */
    double  sres;       /* traj.sres after evalSplineFullTraj (ba.cpp:798,818)            */
    double  vfact;      /* 1/sresC                                                        */
    double  afact;      /* vfact^2                                                        */
    double *sC;         /* [n]  knot sites                                                */
    double *sMVC;       /* [n]  output sites (ba.cpp:809-813)                             */
    double *coef;       /* [n_ch][4][n]  c0,c1,c2,c3; row n-1 left 0 (spline.cpp:203-209) */
    double *samp;       /* [n_theta+n_cart][3][n] value, d/ds, d2/ds2 at the sMVC sites   */
    double *dyn;        /* [4][dyn_dim][n] a1..a4                                         */
    double *mvc;        /* [3][n] pointwise sdot_max, sddotL, sddotH (bo_pointwise_mvc)   */
    const batotp_serial_model *serial; /* serial-chain dynamics model (not owned; NULL: Robot::dynSerial's built-in cases) */
} bo_path;

/* allocate / free a path for problem prob with n knots */
bo_path *bo_path_new(const batotp_problem *prob, int64_t n);
void     bo_path_free(bo_path *p);

/* Spline::getSplineCoeffs, "natural" and "clamped" (spline.cpp:168-211,225-243,252-276).
 * c points to [4][n]. */
void bo_spline_coeffs(const double *y, int64_t n, double *c, int clamped);
/* the second derivatives those rows are formed from (natural spline): sol[0 .. n-1] */
void bo_spline_sol(const double *y, int64_t n, double *sol);
/* Spline::findInterpSegs (spline.cpp:56-99); returns -1 on the division-by-zero error */
int  bo_find_interp_segs(const double *a_in, int64_t n_in, const double *a_out, int64_t n_out,
                         int32_t *seg, double *tau);
/* Spline::interp1spline (spline.cpp:129-155) */
void bo_interp1_spline(const double *c, int64_t n_c, const int32_t *seg, const double *tau,
                       int64_t n_out, double tfact, double *b, double *bD, double *bD2);

/* Robot::findCSPR3DOFpmat (robot.cpp:291-322), row-major [3][3] */
void bo_cspr_pmat(double pmat[9]);
/* solveLinSys, LU branch (util.cpp:413-442 -> Eigen PartialPivLU), dim <= 8 */
void bo_solve_lin_sys(int dim, const double *A /* [dim][dim] row-major */, const double *b, double *x);
/* solveQuadratic (util.cpp:361-383) */
int  bo_solve_quadratic(double A, double B, double C, double *sol1, double *sol2);

/* BA::evalSplineFullTraj as called at ba.cpp:299 (oldRes == newRes == sres):
 * y = [n_theta+n_cart][n] knot values.  Fills sC, sMVC, coef (theta, cart), samp.  -1 on error. */
int  bo_precompute_kin(const batotp_problem *prob, bo_path *p, const double *y, double sres);
/* BA::findDynModel (ba.cpp:873-949): fills dyn and the a1..a4 channels of coef.
 * trig: optional [4][n] host trig for RR (NULL -> libm). */
int  bo_precompute_dyn(const batotp_problem *prob, bo_path *p, const double *trig);
/* Serial-chain dynamics (BASELINE config 3; no reference model: batotp_oracle_dyn.c says what pins it).
 * One pass of the recursive Newton-Euler algorithm, and a1..a4 [n_links][n] at the knots of p;
 * trig: optional [2*n_links][n] host cosines, then sines, of the joint angles (radians). */
void bo_rnea(const batotp_serial_model *m, const double *cq, const double *sq, const double *qd,
             const double *qdd, const double *a0, double *tau);
int  bo_dyn_serial(const batotp_serial_model *m, const bo_path *p, const double *trig,
                   double *a1, double *a2, double *a3, double *a4);

/* BA::sweep (ba.cpp:979-1195).  dir=-1 reverse / +1 forward.  For dir=+1, (mvc_s, mvc_sdot, n_mvc)
 * is the curve published by the reverse sweep.  Output arrays have capacity cap points; ascending s.
 * Returns 0, or -1 when the reference would return -1 (max integration time). */
int  bo_sweep(const batotp_problem *prob, const bo_path *p, int dir,
              const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
              double *out_s, double *out_sdot, int64_t cap,
              int64_t *n_out, int64_t *n_steps, double *t_total, uint32_t *status, int32_t *n_bisect_fail);
/* the same sweep under the product's storage rule for BATOTP_F_CURVES_IN_PLACE (in_place != 0, forward sweep: the path ends
 * with BATOTP_ST_CAPACITY when its curve comes within 64 points of the reverse points still to be read) */
int  bo_sweep_ex(const batotp_problem *prob, const bo_path *p, int dir,
                 const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
                 double *out_s, double *out_sdot, int64_t cap,
                 int64_t *n_out, int64_t *n_steps, double *t_total, uint32_t *status, int32_t *n_bisect_fail, int in_place);

/* sdotLim + applyAccelConstraintsBisectionPt at every knot (K3 definition) */
void bo_pointwise_mvc(const batotp_problem *prob, bo_path *p);

/* per-point known-answer hooks used by the tests: evaluate the constraint set at (s, sdot_in)
 * from a cold cursor.  Outputs: sdot after sdotLim+bisection, sddotL, sddotH, nIter, rc. */
void bo_point_eval(const batotp_problem *prob, const bo_path *p, int dir, double s, double sdot_in,
                   double *sdot_out, double *sddot_l, double *sddot_h, int32_t *n_iter, int32_t *rc);

/* one call of applyAccelConstraintsBisectionPt / sdotLim from a given cursor state: replays of the fp64 known-answer
 * vectors read out of the reference binary (oracle/make_golden_f64.py) */
void bo_kat_accel(const batotp_problem *prob, const bo_path *p, int dir, double s_cur, double sdot_cur, int64_t cur_seg_c,
                  double sddot_in, double *sdot_out, double *sddot_out, double *sddot_l, double *sddot_h, int32_t *n_iter,
                  int32_t *rc, int64_t *cur_seg_out);
void bo_kat_sdot_lim(const batotp_problem *prob, const bo_path *p, int dir, double s_cur, double sdot_in, double sdot_min,
                     const double *theta_d_pt, double cart_coeff0, const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
                     int64_t cur_seg_mvc, double *sdot_out, int64_t *cur_seg_mvc_out);

/* Path resampling before the hot path: remClosePts (util.cpp:452-524), the two
 * BA::adjust_s passes (ba.cpp:412-638), BA::interpSpecial (ba.cpp:651-781), the resampling
 * BA::evalSplineFullTraj (ba.cpp:790-863) and Robot::invKinCSPR3DOF (robot.cpp:243-278) for one path.
 * x: [n_joints+n_cart][n_in].  *y_out is malloc'd [n_joints+n_cart][*n_out] (caller frees).
 * Returns -1 for path kinds outside the device resampler's scope (see batotp_hip.h). */
typedef batotp_resample_params bo_resample_params;
int  bo_resample(const bo_resample_params *prm, int64_t n_in, const double *x, double sres_in,
                 double **y_out, int64_t *n_out, double *sres_out, uint32_t *status);
int  bo_resample_auto(const bo_resample_params *prm, int64_t n_in, const double *x, double sres_in, double **y_out, int64_t *n_out,
                      double *sres_out, uint32_t *status, double auto_out[5]);
/* Observer of the resampler's intermediate stages, for the CALLING THREAD (NULL: none): while set, bo_resample / bo_resample_auto hand it
 * the arrays the stage trace of the product names (batotp_hip_set_resample_trace, include/batotp_hip.h) -- 0: taught points after
 * close-point removal, 1: their sites, 2: their spline coefficients ([C][4][n]; the product keeps second derivatives), 3: the points
 * interpSpecial emitted, 4-6: the same three of the second pass, 7: the knots.  The checker's ABI shim uses it to exercise the reporting
 * path of the host library's two-evaluations guard on the CPU. */
typedef void (*bo_resample_stage_fn)(void *user, int stage, const double *data, int64_t count);
void bo_resample_set_stage_observer(bo_resample_stage_fn fn, void *user);

/* Output stage behind the hot path: BA::interpOutputData (ba.cpp:1661-1931) for JOINT paths
 * without kinematic model and without torque constraints.  p: the path as precomputed for the sweep;
 * fwd_s[n_fwd]: s of the forward curve; t_step: its time step.  Also CART paths of the 3-cable robot with the torque
 * recomputation of ba.cpp:1744-1790.  *out is malloc'd: joint rows, then (cable robot) Cartesian and torque rows. */
int  bo_output(const batotp_problem *prob, const batotp_output_params *prm, const bo_path *p, const double *fwd_s, int64_t n_fwd,
               double t_step, double **out /* [n_theta + n_cart + n_trq][*n_out] */, int32_t *n_cart_out, int32_t *n_trq_out,
               int64_t *n_out, double *sres_out);
/* Forward kinematics: Robot::fwdKinKuka robot.cpp:105-174, Robot::fwdKinRR robot.cpp:188-202, split into the
 * trigonometry of the points (glibc sincos) and the arithmetic on it.  theta [n_joints][n]; trig [bo_fwdkin_trig_rows][n]: KUKA
 * cos(t_k) k = 0..6 then sin(t_k), RR cos(th1), cos(th1+th2), sin(th1), sin(th1+th2); cart rows 0..2 (KUKA) / 0..1 (RR) of
 * cart[.][n] are written.  -1 for robots without a model. */
int  bo_fwdkin_trig_rows(int robot_type, int n_joints);
int  bo_fwdkin_trig(int robot_type, int n_joints, const double *theta, int64_t n, double *trig);
int  bo_fwdkin_from_trig(int robot_type, int n_joints, const double *trig, int64_t n, double *cart);
int  bo_fwdkin(int robot_type, int n_joints, const double *theta, int64_t n, double *cart, double *trig_scratch);
/* tool poses: aa2q util.cpp:534-555, q2aa util.cpp:562-581, BA::aa2qVect ba.cpp:327-369, BA::q2aaVect ba.cpp:384-403 */
void bo_aa2q(const double aa[3], double q[4]);
void bo_q2aa(const double q[4], double aa[3]);
void bo_aa2q_rows(double *rows, int64_t stride, int64_t n);
void bo_q2aa_rows(double *rows, int64_t stride, int64_t n);
/* trigonometry of Robot::dynRR (robot.cpp:408-419): out = cos(th1), cos(th2), cos(th1+th2), sin(th2); th2 through one sincos() */
void bo_rr_dyn_trig(double th1, double th2, double out[4]);
/* Robot::dynRR (robot.cpp:377-431) on the samples of p (value, d/ds, d2/ds2 of the two joints); trig: optional [4][n] table */
void bo_dyn_rr(const bo_path *p, const double *trig, double *a1, double *a2, double *a3, double *a4);
/* solveLinSys with isSVD = 1 (util.cpp:421-438): Eigen's two-sided Jacobi SVD and its solve, restated in batotp_oracle_svd.c.
 * A row-major [n][n]; returns 1 (x untouched) if the system is reported ill-conditioned.  bo_solve picks by BATOTP_F_SVD. */
int  bo_solve_lin_sys_svd(int n, const double *A, const double *b, double *x);
void bo_solve(const batotp_problem *prob, int n, const double *A, const double *b, double *x);
extern int bo_svd_order[2];
/* Robot::setA for CSPR3DOF (robot.cpp:534-558), A row-major [3][3] */
void bo_cspr_setA(const double *pmat, const double *theta, const double *cart, double *A);

#ifdef __cplusplus
}
#endif
#endif
