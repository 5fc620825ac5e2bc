/*
 * batotp_hip.h -- C-ABI of the MI355X (gfx950) implementation of batotp's hot path:
 * per-knot spline/dynamics precompute, per-knot max-admissible-sdot evaluation and the
 * backward/forward constant-step sweep of (s, sdot) over a batch of independent paths.
 *
 * The reference (ebarnett2/batotp) has no FFI layer: its boundary is the C++ class BATOTP::BA
 * (reference batotp/ba.h:168-255).  The host C++11 drop-in (batotp_amd/host/ba.h) keeps that
 * class and calls the entry points below; nothing above this line of the stack knows about HIP.
 * Every entry point takes plain pointers / sizes, returns an int status (BATOTP_OK == 0),
 * never throws and never prints.
 *
 * Which reference routine each entry point replaces (file:line in /root/reference):
 *   batotp_hip_precompute      BA::evalSplineFullTraj   batotp/ba.cpp:790-863  (final call, ba.cpp:299)
 *                              Spline::getSplineCoeffs  batotp/spline.cpp:168-211
 *                              Spline::solveTriDiagNatural batotp/spline.cpp:252-276
 *                              Spline::findInterpSegs / interp1spline  spline.cpp:56-99,129-155
 *                              BA::findDynModel         batotp/ba.cpp:873-949
 *                              BA::dynCoeffs2Ser        batotp/ba.cpp:958-967
 *                              Robot::dynRR / dynCSPR3DOF / setA  batotp/robot.cpp:377-431,487-517,534-558
 *                              solveLinSys              batotp/util.cpp:413-442
 *   batotp_hip_pointwise_mvc   BA::sdotLim + BA::applyAccelConstraintsBisectionPt evaluated at
 *                              every knot               batotp/ba.cpp:1204-1236,1248-1332
 *   batotp_hip_resample        BA::adjust_s / interpSpecial / remClosePts  batotp/ba.cpp:412-781, util.cpp:452-524
 *   batotp_hip_output          BA::interpOutputData     batotp/ba.cpp:1661-1931
 *   batotp_hip_sweep           BA::sweep and everything it calls  batotp/ba.cpp:979-1195,
 *                              1204-1236,1248-1332,1341-1413,1423-1439,1449-1581,1590-1652
 *                              solveQuadratic           batotp/util.cpp:361-383
 */
#ifndef BATOTP_HIP_H
#define BATOTP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BATOTP_MAX_JOINTS 8   /* lanes per path group; nJoints, dynDim <= 8            */
#define BATOTP_MAX_CART   8   /* cart channels carried (only x,y,z enter constraints)   */

/* status codes returned by every entry point */
#define BATOTP_OK              0
#define BATOTP_ERR_ARG        -1
#define BATOTP_ERR_HIP        -2   /* a HIP runtime call failed; see batotp_hip_last_error */
#define BATOTP_ERR_NO_DEVICE  -3
#define BATOTP_ERR_STATE      -4   /* call sequence violated (e.g. sweep before precompute) */
#define BATOTP_ERR_ALLOC      -5

/* robot ids: reference batotp/robot.h:33-37 */
#define BATOTP_ROBOT_KUKA     1
#define BATOTP_ROBOT_UR       2
#define BATOTP_ROBOT_RR       3
#define BATOTP_ROBOT_CSPR3DOF 4
#define BATOTP_ROBOT_GENJNT   5

/* problem flags (reference BA private members, batotp/ba.h:262-302) */
#define BATOTP_F_JNT_ACC_ON     (1u<<0)  /* _isJntAccConOn  */
#define BATOTP_F_TRQ_ON         (1u<<1)  /* _isTrqConOn     */
#define BATOTP_F_CART_VEL_ON    (1u<<2)  /* _isCartVelConOn */
#define BATOTP_F_CART_ACC_ON    (1u<<3)  /* _isCartAccConOn */
#define BATOTP_F_PARALLEL       (1u<<4)  /* _isParallelMechOrig: dynamics are A*tau = a1 sddot+... */
#define BATOTP_F_PAR2SER        (1u<<5)  /* _isPar2Ser: convert a1..a4 with A^-1 per knot (ba.cpp:916-938) */
#define BATOTP_F_HOST_TRIG      (1u<<6)  /* RR dynamics use host-supplied cos/sin (bit parity with glibc) */
#define BATOTP_F_NO_SAMPLES     (1u<<7)  /* do not keep traj.theta/thetaD/thetaD2-style knot samples (saves 24 B/knot/channel;
                                            without torque constraints, or together with BATOTP_F_COMPACT_SPLINES;
                                            download_samples then fails) */
#define BATOTP_F_COMPACT_SPLINES (1u<<8) /* keep every spline as (knot value, second derivative) instead of the four
                                            coefficients per segment: 16 instead of 32 bytes per knot and channel, the
                                            coefficients are formed where they are used (same arithmetic, same results).
                                            Needs BATOTP_F_NO_SAMPLES.  Joint velocity/acceleration-only problems keep their
                                            n_joints + n_cart input channels that way; problems with Cartesian or torque limits
                                            (constraints in serial form: not the parallel-mechanism torque branch) keep ALL
                                            channels that way, a1..a4 of every dynamics row included (round 4): the dynamics
                                            stage forms the samples it needs from the pairs and writes its values into their
                                            channels' slots, the per-knot evaluation and the one-path-per-wavefront sweep kernel
                                            form their coefficient rows from the pairs (the sweep in an LDS window); there is no
                                            sample, dynamics or coefficient array -- for the cable robot 288 instead of 992
                                            bytes per knot.  Uniform knot sites only (a sweep after batotp_hip_upload_path_sites
                                            with other sites fails with BATOTP_ERR_STATE); host trig tables of a serial model
                                            need samples from elsewhere.  upload_coeffs and download_dynamics then fail,
                                            download_coeffs still works for every channel (a_k of row r is c0 of its rows) */
#define BATOTP_F_CURVES_IN_PLACE (1u<<9) /* the forward sweep writes its curve over the reverse curve, as the reference does
                                            with traj.sMVC / traj.sdot (ba.cpp:1143-1160): one curve buffer of max_steps points
                                            per path instead of two.  The forward curve has at least as many points up to any
                                            s as the reverse curve (its speed is capped by it), so it only ever overwrites
                                            reverse points its cursor has left behind; a path whose forward curve comes within
                                            64 points of the reverse points still to be read ends with BATOTP_ST_CAPACITY
                                            (max_steps >= forward points + 72 suffices).  After batotp_hip_sweep(b, +1)
                                            the reverse curve is gone: download / pack of curve -1 and a second forward sweep
                                            return BATOTP_ERR_STATE until the reverse sweep has run again.  Same results. */
#define BATOTP_F_MVC_IN_CURVES   (1u<<10) /* batotp_hip_pointwise_mvc writes its three values per knot into the paths' curve
                                            buffers instead of an array of its own (24 bytes per knot less; needs max_steps >=
                                            1.5 * knots of every path).  They are valid until the next sweep starts: afterwards
                                            batotp_hip_download_mvc returns BATOTP_ERR_STATE, and a pointwise evaluation after a
                                            sweep invalidates the curves (the reverse sweep has to run again).  The overlap
                                            mode (batotp_hip_set_overlap) is ignored for such a batch.  Same results. */

#define BATOTP_F_SVD             (1u<<11) /* _isSVD: solveLinSys (the 3x3 wrench systems of a parallel mechanism: per-knot conversion with
                                            BATOTP_F_PAR2SER, every constraint check without it, the tensions of the output stage)
                                            uses Eigen's two-sided Jacobi SVD instead of its partial-pivot LU (util.cpp:421-438) */

/* per-path status bits written by the sweep kernel (the reference only printf()s these) */
#define BATOTP_ST_MAX_INTEG_TIME (1u<<0) /* ba.cpp:1117-1122 (MAX_INTEGRATION_TIME)        */
#define BATOTP_ST_CAPACITY       (1u<<1) /* output capacity max_steps exhausted            */
#define BATOTP_ST_BISECT_FAIL    (1u<<2) /* >=1 bisection failure, ba.cpp:1307-1319         */
#define BATOTP_ST_NONFINITE      (1u<<3) /* NaN reached the segment search (reference would spin) */
#define BATOTP_ST_SHORT          (1u<<4) /* nPts<4: re-interpolated to 4 points, ba.cpp:1171-1184 */
#define BATOTP_ST_SEG_ERROR      (1u<<5) /* findInterpSegs division-by-zero error, spline.cpp:84-88 */

/* Limits and switches shared by every path of a batch.
 * Mirrors the subset of BA's private configuration the hot path reads (ba.h:262-302). */
typedef struct batotp_problem {
    int32_t  n_joints;                 /* _nJoints                                   */
    int32_t  n_cart;                   /* cart channels uploaded (0 if none)         */
    int32_t  robot_type;               /* BATOTP_ROBOT_*                             */
    uint32_t flags;                    /* BATOTP_F_*                                 */
    double   jnt_vel_max[BATOTP_MAX_JOINTS];  /* _JntVelMax */
    double   jnt_acc_max[BATOTP_MAX_JOINTS];  /* _JntAccMax */
    double   jnt_trq_max[BATOTP_MAX_JOINTS];  /* _JntTrqMax */
    double   jnt_trq_min[BATOTP_MAX_JOINTS];  /* _JntTrqMin */
    double   cart_vel_max;             /* _CartVelMax                                */
    double   cart_acc_max;             /* _CartAccMax                                */
    double   jnt_thresh;               /* _jntThresh                                 */
    double   quad_rad_thresh;          /* _quadraticRadThresh = _cartThresh^2        */
    double   integ_res;                /* _integRes  [s]                             */
    double   max_integ_time;           /* _maxIntegTime [s]                          */
    double   pmat[9];                  /* CSPR cable anchors, row-major [3][3] (robot.cpp:291-322) */
} batotp_problem;

/* per-path outcome of the two sweeps */
typedef struct batotp_path_result {
    double   t_rev;        /* traj.tTotalTraj after sweep(-1): integRes * steps (ba.cpp:1112,1156) */
    double   t_total;      /* traj.tTotalTraj after sweep(+1)                               */
    int64_t  n_rev;        /* points of the reverse curve (traj.nPts after sweep(-1))      */
    int64_t  n_fwd;        /* points of the forward curve                                   */
    int64_t  steps_rev;    /* integration steps taken (nPts-1 before the nPts<4 fix-up)     */
    int64_t  steps_fwd;
    uint32_t status_rev;   /* BATOTP_ST_*                                                   */
    uint32_t status_fwd;
    int32_t  n_bisect_fail_rev;
    int32_t  n_bisect_fail_fwd;
} batotp_path_result;

/* Dynamics model of a serial chain of revolute joints (BASELINE config 3: 7-DOF arm with torque limits).
 * The reference dispatches Robot::dynSerial on the robot type and only knows the two-link arm
 * (batotp/robot.cpp:349-360, dynRR robot.cpp:377-431); this table is how a further `case` is supplied without a
 * kernel per robot: tau = a1 sddot + a2 sdot^2 + a3 sdot + a4 (robot.cpp:368-372) with
 *   a1 = M(q) q',  a2 = M(q) q'' + C(q, q') q',  a3 = fv .* q',  a4 = g(q)
 * evaluated per knot by three passes of the recursive Newton-Euler algorithm.
 * Frames: every link frame coincides with the base frame at q = 0 ("zero-aligned"); joint i turns link i about
 * `axis` (unit vector, the same in link and parent coordinates) by q_i. */
#define BATOTP_MAX_LINKS 8
typedef struct batotp_serial_link {
    double axis[3];     /* joint axis                                                            */
    double off[3];      /* joint origin relative to the parent's joint origin, parent coordinates */
    double com[3];      /* centre of mass relative to the joint origin, link coordinates         */
    double mass;
    double inertia[6];  /* about the centre of mass, link coordinates: Ixx Iyy Izz Ixy Ixz Iyz    */
    double fv;          /* viscous friction coefficient (dynRR uses 10, robot.cpp:423-424)       */
} batotp_serial_link;
typedef struct batotp_serial_model {
    int32_t n_links;    /* == n_joints of the problem                                            */
    int32_t degrees;    /* joint values are degrees: scaled by _DEG2RAD first (robot.cpp:401-406) */
    double  gravity[3]; /* gravitational acceleration in base coordinates, e.g. (0, 0, -9.81)    */
    batotp_serial_link link[BATOTP_MAX_LINKS];
} batotp_serial_model;

typedef struct batotp_ctx   batotp_ctx;    /* one per GPU (device + stream)        */
typedef struct batotp_batch batotp_batch;  /* B independent paths resident in HBM  */

/* ---- device context --------------------------------------------------------------------- */
int  batotp_hip_device_count(int *count);
int  batotp_hip_ctx_create(int device, batotp_ctx **out);
int  batotp_hip_ctx_destroy(batotp_ctx *ctx);
/* release the workspaces the context caches between batotp_hip_resample calls */
int  batotp_hip_ctx_trim(batotp_ctx *ctx);
/* message of the last HIP failure on this thread ("" if none) */
const char *batotp_hip_last_error(void);
/* device-side known-answer test of fp64 div / sqrt rounding (no contraction):
 * computes q[i]=a[i]/b[i], r[i]=sqrt(a[i]), p[i]=a[i]*b[i]+q[i] on the GPU for n host values */
int  batotp_hip_fp64_kat(batotp_ctx *ctx, int64_t n, const double *a, const double *b,
                         double *q, double *r, double *p);
/* q[i] = a[i] / 6.0 computed the way the compact spline form (BATOTP_F_COMPACT_SPLINES) divides:
 * through the reciprocal with one exact residual correction; must equal the IEEE quotient */
int  batotp_hip_div6_kat(batotp_ctx *ctx, int64_t n, const double *a, double *q);
/* known-answer test of the natural-spline solves outside the hot path (reference batotp/spline.cpp:252-276; resampler and output
 * stage): second derivatives of ONE series of n values (n >= 4), (a) sol: by the wavefront-per-series kernel of
 * batotp_amd/csrc/spline_lanes.hip.h -- the series in 64 chunks whose warm-ups are compared bit for bit -- followed by the
 * lane-per-series kernel if it flagged the series (too short, or a comparison failed), exactly as the two stages run them, and
 * (b) sol_seq: by the lane-per-series kernel alone.  *redone = 1: the series took the sequential kernel in (a). */
int  batotp_hip_spline_lanes_kat(batotp_ctx *ctx, int64_t n, const double *y, double *sol, double *sol_seq, int32_t *redone);
/* known-answer test of the division through a shared refined reciprocal (batotp_amd/csrc/sweep8.hip.h: the last three
 * operations of hipcc's own fp64 division sequence, valid for operands in [2^-350, 2^350]; the sweep kernel k_sweep8 uses it
 * for the quotients by theta' of reference batotp/ba.cpp:1223 and :1526-1531): q[i] = the kernel's a[i] / b[i] -- through the
 * reciprocal when both operands lie in the window (in_window[i] = 1), by the hardware sequence otherwise. */
int  batotp_hip_sdiv_kat(batotp_ctx *ctx, int64_t n, const double *a, const double *b, double *q, int32_t *in_window);

/* ---- batch lifetime --------------------------------------------------------------------- */
/* n_knots[b] = number of uniform-s knots of path b (>=4); max_steps = per-path capacity of each
 * integrated curve (points).  Channels per path: n_joints theta + n_cart cart (+ 4*dynDim built
 * on the device when BATOTP_F_TRQ_ON). */
int  batotp_hip_batch_create(batotp_ctx *ctx, const batotp_problem *prob, int32_t n_paths,
                             const int64_t *n_knots, int64_t max_steps, batotp_batch **out);
int  batotp_hip_batch_destroy(batotp_batch *batch);

/* Upload the knot values of paths [path0, path0+n): for each path, channel-major doubles
 * theta[n_joints][N] then cart[n_cart][N], paths concatenated; sres[n] = knot spacing
 * (traj.sres at ba.cpp:299). */
int  batotp_hip_upload_knots(batotp_batch *batch, int32_t path0, int32_t n,
                             const double *y, const double *sres);
/* same, with the knot values already resident in HBM (y_dev is a device pointer; sres stays a
 * host array) */
int  batotp_hip_upload_knots_device(batotp_batch *batch, int32_t path0, int32_t n,
                                    const double *y_dev, const double *sres);
/* same for paths that carry MORE rows than the batch keeps: every path of the source is [src_rows][N] (src_rows >= n_joints +
 * n_cart of the batch) and its first n_joints + n_cart rows are taken -- the knots of batotp_hip_resample (joint rows, then the
 * Cartesian rows traj.cart of reference batotp/ba.cpp:247-262) into a batch of a problem without Cartesian limits, which carries no
 * Cartesian channels (BA::deviceSweep): one call and one kernel for a block of paths instead of a call per path */
int  batotp_hip_upload_knots_device_rows(batotp_batch *batch, int32_t path0, int32_t n,
                                         const double *y_dev, int32_t src_rows, const double *sres);
/* RR only (BATOTP_F_HOST_TRIG): trig[4][N] = cos(th1), cos(th2), cos(th1+th2), sin(th2) of the
 * knot samples of path p, evaluated with the host libm (robot.cpp:408-419). */
int  batotp_hip_upload_rr_trig(batotp_batch *batch, int32_t path, const double *trig);
/* Serial-chain dynamics (BATOTP_F_TRQ_ON on a serial robot): with a model set, batotp_hip_precompute stage 2
 * evaluates a1..a4 with it instead of the built-in two-link arm -- Robot::dynSerial's switch
 * (batotp/robot.cpp:349-360) with one more case.  Needed for every serial robot other than BATOTP_ROBOT_RR. */
int  batotp_hip_set_serial_model(batotp_batch *batch, const batotp_serial_model *model);
/* BATOTP_F_HOST_TRIG with a serial model: trig[2*n_joints][N] = cos(q_j) rows, then sin(q_j) rows, of the knot
 * samples of path p (q_j in radians), evaluated with the host libm; without the flag the device libm is used
 * (not bit-identical to glibc). */
int  batotp_hip_upload_joint_trig(batotp_batch *batch, int32_t path, const double *trig);
/* the built-in model table of a robot type (include/batotp_models.h: BATOTP_ROBOT_KUKA = LWR IV+ with nominal
 * inertial parameters, BATOTP_ROBOT_RR = the point masses of Robot::dynRR as a chain); BATOTP_ERR_ARG if none */
int  batotp_hip_builtin_serial_model(int32_t robot_type, batotp_serial_model *out);

/* Marshalling entry points used by BA::sweep (single path, B = 1), which like the reference reads
 * whatever the caller left in the public Traj arrays (reference ba.h:140-152):
 *   sites[N] = traj.sC, vfact/afact = traj.vFact/aFact, parallel_now = BA::_isParallelMech */
int  batotp_hip_upload_path_sites(batotp_batch *batch, int32_t path, const double *sites,
                                  double vfact, double afact, int32_t parallel_now);
/* c[4][N] = c0,c1,c2,c3 of one channel (channel order as in batotp_hip_download_coeffs) */
int  batotp_hip_upload_coeffs(batotp_batch *batch, int32_t path, int32_t channel, const double *c);
/* the curve published by the reverse sweep (traj.sMVC / traj.sdot, n = traj.nPts) */
int  batotp_hip_upload_curve(batotp_batch *batch, int32_t path, const double *s, const double *sdot,
                             int64_t n);
/* the curve published by the FORWARD sweep (traj.sMVC / traj.sdot after sweep(+1), n = traj.nPts) with its traversal
 * time traj.tTotalTraj: what the output stage (batotp_hip_output) needs of a path whose sweeps ran elsewhere -- the
 * step-by-step host API BA::sweep / BA::interpOutputData (reference ba.cpp:979-1195, 1661-1931) moves one Traj at a time
 * through batches of one path.  Marks both sweeps as done, for the WHOLE batch: the call is only valid on a batch of one path
 * (BATOTP_ERR_STATE otherwise) that does not share curve slots (no BATOTP_F_CURVES_IN_PLACE / BATOTP_F_MVC_IN_CURVES). */
int  batotp_hip_upload_forward_curve(batotp_batch *batch, int32_t path, const double *s, const double *sdot,
                                     int64_t n, double t_total);
/* the integration step _integRes of paths [path0, path0 + n) (default: batotp_problem.integ_res for every path).  The
 * automatic integration resolution of the reference (ba.cpp:493-556, the class default ba.h:309) derives it from each path;
 * the sweeps, the pointwise evaluation and the output stage read it per path.  Call before batotp_hip_precompute /
 * batotp_hip_sweep; integ_res[k] must be positive and finite, or NaN (the rule's own result for a robot without Cartesian
 * limits: such a path takes no step and ends with BATOTP_ST_MAX_INTEG_TIME); zero, negative and infinite steps are
 * BATOTP_ERR_ARG. */
int  batotp_hip_set_path_integ_res(batotp_batch *batch, int32_t path0, int32_t n, const double *integ_res);

/* ---- the hot path ----------------------------------------------------------------------- */
/* stage 1 = theta/cart spline coefficients + knot samples (evalSplineFullTraj);
 * stage 2 = dynamics coefficients a1..a4 and their splines (findDynModel); stage 0 = both. */
int  batotp_hip_precompute(batotp_batch *batch, int32_t stage);
/* max admissible sdot and sddot interval at every knot (K3) */
int  batotp_hip_pointwise_mvc(batotp_batch *batch);
/* dir = -1: reverse sweep; dir = +1: forward sweep capped by the reverse curve */
int  batotp_hip_sweep(batotp_batch *batch, int32_t dir);
/* precompute + sweep(-1) + sweep(+1) */
int  batotp_hip_optimize(batotp_batch *batch);
int  batotp_hip_synchronize(batotp_ctx *ctx);

/* ---- results ---------------------------------------------------------------------------- */
int  batotp_hip_get_results(batotp_batch *batch, batotp_path_result *out /* [n_paths] */);
/* which = -1 reverse curve, +1 forward curve; copies min(n, cap) points, *n = points available */
int  batotp_hip_download_curve(batotp_batch *batch, int32_t path, int32_t which,
                               double *s, double *sdot, int64_t cap, int64_t *n);
/* spline coefficients of one channel of one path: c[4][N] (c0,c1,c2,c3).  Channel order:
 * theta[0..nJ) cart[0..nC) a1[0..d) a2[0..d) a3[0..d) a4[0..d) */
int  batotp_hip_download_coeffs(batotp_batch *batch, int32_t path, int32_t channel, double *c);
/* knot samples value/first/second derivative of theta and cart channels: out[3][N] */
int  batotp_hip_download_samples(batotp_batch *batch, int32_t path, int32_t channel, double *out);
/* dynamics coefficients a_k (k=1..4) row r at the knots: out[N] */
int  batotp_hip_download_dyn(batotp_batch *batch, int32_t path, int32_t k, int32_t row, double *out);
/* K3 outputs of one path: sdot_max[N], sddot_l[N], sddot_h[N] (any pointer may be NULL).
 * Deviation from the reference, stated here on purpose: at a knot where the bisection finds no admissible sdot
 * (ba.cpp:1307-1319 returns -1) sddot_l / sddot_h are published as NaN.  The reference has no per-knot output at all (K3
 * is this implementation's own product); what its early loop exits last wrote into traj.sddotL/H at such a
 * point is an artefact of the loop order and differed between correct implementations, so the K3 definition replaces it
 * by NaN (the sweep itself keeps the reference's stale-value behaviour: that IS observable in its curves). */
int  batotp_hip_download_mvc(batotp_batch *batch, int32_t path, double *sdot_max,
                             double *sddot_l, double *sddot_h);

/* ---- plumbing for the Python / torch.distributed layer ---------------------------------- */
/* The integrated curves of paths [path0, path0 + n_paths) -- which = -1 reverse, +1 forward -- packed path after path
 * into dst_dev (DEVICE memory with room for dst_points (s, sdot) pairs of doubles); *total_points = pairs written.  This
 * is the send buffer of the multi-GPU curve gather (traj.sMVC / traj.sdot after sweep, reference ba.cpp:1154-1190): the
 * curves have different lengths, so a size exchange + grouped send/recv moves them (batotp_amd/dist.py). */
int  batotp_hip_pack_curves(batotp_batch *batch, int32_t which, int32_t path0, int32_t n_paths, void *dst_dev,
                            int64_t dst_points, int64_t *total_points);
/* device pointer + element count of the per-path result table (batotp_path_result[n_paths]) */
int  batotp_hip_results_device_ptr(batotp_batch *batch, void **ptr, int64_t *bytes);
/* Duration of the most recent launch of a kernel family, measured with HIP events on the
 * stream the kernels were launched on.  which: 1 = precompute, 2 = pointwise_mvc,
 * 3 = reverse sweep, 4 = forward sweep.  Synchronises on the stop event. */
int  batotp_hip_last_kernel_ms(batotp_batch *batch, int32_t which, float *ms);
/* bytes of HBM held by the batch */
int  batotp_hip_batch_bytes(batotp_batch *batch, int64_t *bytes);
/* tuning knob: lanes per path in the sweep kernel: 8 (a joint per lane), 16 (two halves of 8 lanes share the bounds of the
 * sddot interval), 32 (four bisection candidates per pass), 4 / 2 / 1 (two / four / all joints per lane), 64 (a wavefront
 * per path: the kernel written for the latency-bound regime; a parallel mechanism's torque limits without BATOTP_F_PAR2SER
 * and paths with uploaded sites fall back to 32) or 0 = automatic (default: 64 up to 6144 paths in the reverse and 3072 in the forward sweep -- 32 where 64 does not apply and
 * every path can have a wavefront to itself --, 8 beyond) */
int  batotp_hip_set_sweep_group(batotp_ctx *ctx, int32_t lanes);
/* tuning knob: paths per 64-lane wavefront in the sweep kernel, 1 .. 64/lanes (0 = automatic:
 * few paths are spread over more wavefronts, many paths fill every lane) */
int  batotp_hip_set_paths_per_wave(batotp_ctx *ctx, int32_t n);
/* loop form of the 8-lane sweep kernel, per direction, for problems with joint velocity / acceleration limits only on
 * uniform knot sites (everything else always runs the nested loops): -1 = the stage loop contains the bisection loop (a
 * wavefront stays in a stage's bisection as long as any of its paths does); 0..8 = one flat loop in which every path is
 * either waiting for its next stage or inside a constraint check, and the next stage is started as soon as hold/8 of the
 * wavefront's live paths wait for it (8 = all of them); -2 (default) = automatic: hold 4 for the reverse sweep (measured on
 * the bench batch: 855 ms against 1130 ms), nested loops for the forward sweep (which does not gain).  Same arithmetic per
 * path: results are bit-identical in every test (tests/test_gpu_parity.py, test_gpu_fuzz.py) and bench.py re-checks the
 * result rows of every run against the nested loops.  DESIGN.md 4 has the history. */
int  batotp_hip_set_sweep_hold(batotp_ctx *ctx, int32_t reverse, int32_t forward);
/* which code runs that flat loop in the 8-lane layout: 1 (default) = k_sweep8 (batotp_amd/csrc/sweep8.hip.h: the same loop and
 * arithmetic -- reference batotp/ba.cpp:1053-1123 with sdotLim :1204-1236 and the bisection :1248-1332 -- hand-structured for
 * the instruction count: one level of divergence, wavefront-uniform segment walks, curve points stored 64 bytes at a time),
 * 0 = the flat instantiation of the general kernel k_sweep (kept for A/B runs and as a second implementation the parity tests
 * compare).  Never changes a result. */
int  batotp_hip_set_flat_form(batotp_ctx *ctx, int32_t form);
/* The gate of that automatic choice.  The flat loop's torque instantiation gave wrong results with the toolchain named
 * below for a reason that is not understood and is not compiled; the instantiations that ship passed the whole
 * parity / fuzz suite with exactly that toolchain.  So the automatic choice takes the flat loop only if (a) the library was
 * built by the validated toolchain (batotp_hip_toolchain reports both strings) and (b) a canary on this device -- two small
 * batches with ordinary, crawling and always-failing paths, 8 per wavefront, nested against flat loop, result rows and
 * curves compared bit for bit; run once per context on first use, tens of milliseconds -- found no difference.
 * status: 1 = flat loop in use, -1 = library built by another toolchain (nested loops), -2 = the canary found a
 * difference (nested loops), -3 = the canary could not run (nested loops).  An explicit batotp_hip_set_sweep_hold(>= 0)
 * is honoured regardless: that is the developer's switch. */
int  batotp_hip_flat_loop_status(batotp_ctx *ctx, int32_t *status);
/* toolchain this library was built with / toolchain the flat loop was validated with (NUL-terminated, truncated to cap) */
int  batotp_hip_toolchain(char *built_with, char *validated_with, int32_t cap);
/* what the most recent sweep launch of this batch in direction dir used: lanes per path (64 = a wavefront per path), paths
 * per wavefront, and the hold of the flat loop (-1 = nested loops).  Any pointer may be NULL. */
int  batotp_hip_last_sweep_launch(batotp_batch *batch, int32_t dir, int32_t *lanes, int32_t *paths_per_wave, int32_t *hold);
/* tuning knob: software prefetch in the sweep kernel, per direction: bit 0 = touch the spline rows ahead of the cursor, bit 1 = touch
 * the reverse curve ahead of its cursor (forward sweep only); -1 (default) = automatic: rows in the reverse sweep always, rows and
 * curve in the forward sweep while every path has a wavefront to itself (latency-bound regime).  Never changes a result. */
int  batotp_hip_set_sweep_prefetch(batotp_ctx *ctx, int32_t reverse, int32_t forward);
/* K1 (the spline build) in tiles of knots: the Thomas recurrences run from 64 knots before / beyond each chunk of 16 knots
 * (they contract by 0.268 per knot: the warm-up arrives with the sequential value's bits), every warm-up value is compared
 * bit for bit with the neighbouring chunk's, and a series with a disagreement is redone by the sequential kernel, so the
 * result IS the sequential kernel's (batotp_amd/csrc/spline_tile.hip.h).  on: -1 (default) automatic -- tiles for batches of
 * up to 4096 series (paths x channels; 16 384 where coefficient rows are written), where the lane-per-series kernel is a dependent chain of N steps with most of the GPU
 * idle (one 6-joint trajectory of 1e5 knots: 0.05 ms instead of 9.5 ms), the lane-per-series kernels beyond --, 1 always,
 * 0 never (the parity tests run both).  Round 4: beyond that size, batches that keep their splines as pairs take the lane-per-series
 * kernel in its single-pass form (batotp_amd/csrc/spline_stream.hip.h: the forward elimination of reference batotp/spline.cpp:257-268
 * keeps its last 64 values in LDS, the back substitution of spline.cpp:269-274 runs in blocks that start 48 knots ahead and are
 * checked against their neighbours the same way; half the HBM traffic of the two-sweep kernel); 2 = that kernel whatever the size. */
int  batotp_hip_set_spline_tiles(batotp_ctx *ctx, int32_t on);
/* the bisection of BA::applyAccelConstraintsBisectionPt (reference batotp/ba.cpp:1248-1332) in the one-path-per-wavefront sweep
 * kernel (batotp_amd/csrc/sweep1.hip.h), problems with joint velocity / acceleration limits and / or serial torque limits (no
 * Cartesian acceleration limit): after a violated first check the kernel computes the speed at which the joints' sddot intervals
 * stop intersecting (closed form where the constraints are lines in sdot^2, a division-free approximate check per candidate
 * otherwise), and takes every iteration of the loop whose outcome is CERTAIN given the rounding-error bounds of the check without
 * running the check;
 * candidates within the error band, and the last one (whose sddot bounds are the result), get the real check.  Results are
 * identical with it on (default) and off (the parity tests run both).  on: 1 / 0.  Round 4: the batch kernel of the 8-lane layout
 * (batotp_amd/csrc/sweep8.hip.h) carries the same certificate in its forward sweep, where the eight paths of a wavefront run in
 * lockstep and wait for a path that bisects; round 6: and in its reverse sweep (batotp_hip_set_cert_hold below). */
int  batotp_hip_set_fast_forward(batotp_ctx *ctx, int32_t on);
/* the same certificate in the REVERSE sweep of the 8-lane batch kernel (k_sweep8), where 22 % of the stages bisect and the passes of
 * the flat loop are shared by the paths of a wavefront: run once per arriving path the certificate costs more than the checks it
 * removes (profiles/r04_j_*), so a path whose first check of a stage (reference batotp/ba.cpp:1270-1280) was violated waits in a
 * phase of its own and ONE certificate block serves the paths gathered there -- once hold/8 of the wavefront's live paths wait, or
 * when nothing else can run.  hold: 1..8; 0 = no certificate in the reverse sweep (every iteration of ba.cpp:1267-1321 gets its
 * check); -1 (default) = automatic (the measured optimum, profiles/r06_a_*).  batotp_hip_set_fast_forward(0) switches it off as
 * well.  Never changes a result (tests/test_gpu_parity.py, tests/test_gpu_fuzz.py, tests/test_gpu_zz_as_worded.py run both). */
int  batotp_hip_set_cert_hold(batotp_ctx *ctx, int32_t hold);
/* debug aid: with on != 0 every workspace a stage is about to use (the resampler's and the output stage's scratch and result
 * arrays, every array of a batch at its creation) is first filled with 0xFF bytes -- NaNs as doubles, -1 as integers -- so that a
 * kernel that reads memory nobody wrote gives a reproducible wrong (or crashing) result instead of one that depends on what the
 * memory happened to hold.  tests/ run the GPU suite once with it (BATOTP_TEST_POISON=1 python -m pytest -m gpu); results with and
 * without it are identical.  Costs a memset per call; off by default. */
int  batotp_hip_set_poison(batotp_ctx *ctx, int32_t on);
/* which kernel evaluates the per-knot values (batotp_hip_pointwise_mvc) of velocity / acceleration-only problems: 1 (default)
 * k_pointwise_va, written for that constraint family (shared reciprocals, select-form passes, the certified fast-forward), 0 the
 * general kernel that runs the loop of reference ba.cpp:1267-1321 literally.  Same results bit for bit; the switch exists for
 * A/B measurements and parity tests. */
int  batotp_hip_set_k3_form(batotp_ctx *ctx, int32_t form);
/* Scratch memory the resampler (batotp_hip_resample) and the output stage (batotp_hip_output) may take per chunk of paths, in
 * bytes; 0 (default) = a share of the device memory that is free when the call starts.  A batch that needs more is processed
 * in several chunks of paths through the same workspace; results do not depend on the chunking. */
int  batotp_hip_set_workspace_budget(batotp_ctx *ctx, int64_t resample_bytes, int64_t output_bytes);
/* Ragged batches (paths of different knot counts): 1 (default) the sweeps take the paths sorted by knot count, longest first --
 * longest-processing-time-first for the kernels with a wavefront per path (workgroups are handed out in launch order as SIMD
 * slots free up), paths of similar length in one wavefront for the kernels that carry several (a wavefront lasts as long as its
 * longest path); 0 the order given.  Results, result rows and curve slots are those of the path whatever slot runs it. */
int  batotp_hip_set_path_order(batotp_ctx *ctx, int32_t mode);
/* diagnostic: series of the most recent tiled spline build of this batch whose boundary comparison failed and that the
 * sequential kernel therefore recomputed (paths too short for tiles are not counted).  Expected: 0. */
int  batotp_hip_spline_tile_fallbacks(batotp_batch *batch, int32_t *series);
/* tuning knob: with overlap on, batotp_hip_pointwise_mvc returns at once and its kernel shares the GPU with the
 * sweeps that follow (second HIP stream; nothing in the sweeps reads its output); batotp_hip_get_results,
 * batotp_hip_download_mvc, batotp_hip_synchronize and the next batotp_hip_precompute wait for it.  Default off. */
int  batotp_hip_set_overlap(batotp_ctx *ctx, int32_t on);

/* ---- path resampling before the hot path -------------------------------- */
/* Replaces, for the path kinds listed below, everything BA::interpInputData does between loading
 * the taught points and the final spline build: remClosePts (batotp/util.cpp:452-524), the two
 * BA::adjust_s passes (batotp/ba.cpp:412-638) with BA::interpSpecial (ba.cpp:651-781) and the
 * resampling BA::evalSplineFullTraj (ba.cpp:790-863), Robot::invKinCSPR3DOF (robot.cpp:243-278).
 * Supported: path_type JOINT on a robot without kinematic model (GENJNT), on KUKA (7 joints) and RR (2 joints) with
 * their forward kinematics (Robot::fwdKinKuka / fwdKinRR, robot.cpp:105-202; flag BATOTP_F_HOST_TRIG: cos / sin of the
 * joint angles from the host libm -- bit parity with the reference -- else the device libm), path_type BOTH with
 * 6 pose rows (any robot type: nothing is recomputed from the joints), and path_type CART on the
 * cable robot (CSPR3DOF) with a joint constraint on, including the input decimation and smoothing of
 * ba.cpp:195-242 (smooth / decimate, util.cpp:263-290,347-356); timestamps only set sres_in (the caller
 * drops repeated ones first); no automatic integration resolution.  Anything else returns
 * BATOTP_ERR_ARG: the caller keeps using its host resampler for those. */
#define BATOTP_PATH_JOINT 1   /* reference batotp/ba.h pathType JOINT */
#define BATOTP_PATH_CART  2
#define BATOTP_PATH_BOTH  3   /* joint rows AND tool poses taught together (the UR5 example): n_cart = 6 rows (x, y, z, rx, ry, rz:
                                 axis-angle) go in, n_cart + 1 = 7 rows come out of the resampler -- BA::aa2qVect (ba.cpp:327-369) turns
                                 the orientations into hemisphere-aligned quaternions, which is what is splined; the output stage
                                 turns them back (BA::q2aaVect, ba.cpp:384-403) */

/* per-path status bits of the resampler (0 = resampled) */
#define BATOTP_RS_TOO_SHORT   1u  /* a stage has fewer than 4 points (reference switches to interpTrajLinear) */
#define BATOTP_RS_IDENTICAL   2u  /* "all points identical" exit, ba.cpp:484-488 */
#define BATOTP_RS_CAPACITY    4u  /* interpSpecial emitted more points than planned for */
#define BATOTP_RS_SMALL_STEP  8u  /* "s-resolution is too small between two points", ba.cpp:607-611 */
#define BATOTP_RS_SEG_ERROR  16u  /* findInterpSegs division by zero, spline.cpp:84-88 */

typedef struct batotp_resample_params {
    int32_t  n_joints, n_cart;
    int32_t  robot_type;             /* BATOTP_ROBOT_GENJNT or BATOTP_ROBOT_CSPR3DOF           */
    int32_t  path_type;              /* BATOTP_PATH_*                                          */
    int32_t  scale_type;             /* _scaleType 0 / 1 / 2                                   */
    uint32_t flags;                  /* BATOTP_F_CART_VEL_ON / CART_ACC_ON decide whether interpSpecial
                                        re-evaluates the Cartesian channels (ba.cpp:1362)      */
    double   s_weights[3];           /* _sWeights                                              */
    double   theta_norm_res, theta_norm_res2; /* _thetaNormRes, _thetaNormRes2                 */
    double   cart_norm_res, cart_norm_res2;   /* _cartNormRes,  _cartNormRes2                  */
    double   jnt_thresh, cart_thresh;         /* remClosePts thresholds                        */
    double   pmat[9];                /* cable exit points (CSPR3DOF), row-major 3x3            */
    int32_t  input_decim_fact;       /* _inputDecimFact (values < 2: no decimation)            */
    int32_t  smooth_window;          /* _smoothWindow                                          */
    /* Automatic integration resolution (BATOTP_RS_AUTO_INTEG_RES in `flags`; the class default of the reference,
     * ba.h:309): both adjust_s passes derive the integration step, the s weights, the scale type and the Cartesian
     * resolution of the pass from the path itself (ba.cpp:462-470, 493-556) -- PER PATH; what they leave is reported by
     * batotp_hip_resampled_auto and goes into the batch with batotp_hip_set_path_integ_res.  Inputs of the rule: */
    double   jnt_vel_max[BATOTP_MAX_JOINTS], jnt_acc_max[BATOTP_MAX_JOINTS]; /* _JntVelMax, _JntAccMax */
    double   cart_vel_max, cart_acc_max;  /* _CartVelMax, _CartAccMax                           */
    double   quad_rad_thresh;        /* _quadraticRadThresh (= cartThresh^2, ba.cpp:2048)       */
    int32_t  degrees;                /* _areJointAnglesDegrees                                  */
    int32_t  reserved;
} batotp_resample_params;
#define BATOTP_RS_AUTO_INTEG_RES (1u<<16)  /* batotp_resample_params.flags: run the automatic integration resolution */

typedef struct batotp_resampled batotp_resampled;

/* Resample n_paths taught paths.  x: the points of every path, path after path, each
 * channel-major [n_joints + n_cart][n_in[p]] (theta rows then cart rows; for JOINT paths the cart
 * rows are ignored and taken as zero, for CART paths the theta rows are overwritten by the inverse
 * kinematics); sres_in[p] = traj.sres of the taught points.  The result stays in HBM. */
int  batotp_hip_resample(batotp_ctx *ctx, const batotp_resample_params *prm, int32_t n_paths,
                         const int64_t *n_in, const double *x, const double *sres_in,
                         batotp_resampled **out);
int  batotp_hip_resampled_destroy(batotp_resampled *r);
/* knots per path, traj.sres per path, status bits per path (any pointer may be NULL) */
int  batotp_hip_resampled_info(batotp_resampled *r, int64_t *n_knots, double *sres, uint32_t *status);
/* automatic integration resolution (BATOTP_RS_AUTO_INTEG_RES): what the two adjust_s passes left per path -- _integRes,
 * _sWeights (3 per path) and _scaleType (reference ba.cpp:493-556).  Without the flag s_weights / scale_type come back as the
 * caller passed them and integ_res[] is 0.0 -- batotp_resample_params carries no integration step, and 0.0 is a value
 * batotp_hip_set_path_integ_res refuses, so it cannot be fed on by accident.  Any pointer may be NULL. */
int  batotp_hip_resampled_auto(batotp_resampled *r, double *integ_res, double *s_weights, int32_t *scale_type);
/* device pointer of the knots, laid out as batotp_hip_upload_knots_device expects them (paths
 * with a non-zero status hold 4 zero knots).  The knots live in a workspace the context keeps
 * between calls (allocating tens of GB per call costs more than the kernels): they stay valid until
 * the next batotp_hip_resample on the same context, batotp_hip_ctx_trim or batotp_hip_ctx_destroy;
 * after that this call and batotp_hip_resampled_download return BATOTP_ERR_STATE. */
int  batotp_hip_resampled_knots_device(batotp_resampled *r, const double **y_dev, int64_t *n_doubles);
/* knots of one path to the host: y[n_joints + n_cart][n_knots] */
int  batotp_hip_resampled_download(batotp_resampled *r, int32_t path, double *y);
/* milliseconds the resampling kernels of this object took (HIP events on the context's stream) */
/* an order-independent 64-bit checksum per path of the knots as they lie in HBM (0 for a path with a non-zero status): the wrap-around
 * sum over the values of mix64(bits XOR (index + 1) * 0x9E3779B97F4A7C15), mix64 = the splitmix64 finaliser.  What it is for: the host
 * library evaluates batotp_hip_resample TWICE for every call of BA::interpInputData / BA::optimizeBatch and uses the knots only when
 * counts, spacings, status words and these sums agree (one unexplained wrong result in ~11 000 one-path calls of
 * round 4) -- a many-path batch is compared in HBM, 8 bytes per path cross PCIe. */
int  batotp_hip_resampled_checksums(batotp_resampled *r, uint64_t *sums);
/* diagnostic of that protection: with batotp_hip_set_resample_trace(ctx, 1) a ONE-PATH call of batotp_hip_resample also keeps the same
 * kind of checksum of every intermediate stage -- [0] the taught points after close-point removal / filtering (ba.cpp:160-242), [1] their
 * sites s (adjust_s, first pass), [2] their second derivatives, [3] the points interpSpecial emitted (ba.cpp:651-781), [4] those points as
 * the second pass sees them, [5] their sites, [6] their second derivatives, [7] the knots -- so that two evaluations that disagree can
 * name the stage where they first differ.  batotp_hip_resampled_trace returns BATOTP_ERR_STATE for a call that was not traced.
 * BA::interpInputData switches it on for its evaluations (milliseconds per path). */
int  batotp_hip_set_resample_trace(batotp_ctx *ctx, int32_t on);
int  batotp_hip_resampled_trace(batotp_resampled *r, uint64_t *sums /* [8] */);
/* ... and, for the stages kept on the host as well (2: the second derivatives of the taught points, channel-major; 3: the points
 * interpSpecial emitted, point-major), the array itself: *count doubles (0 for a stage that is not kept); out may be NULL to ask for the
 * count.  BA::interpInputData writes both evaluations' arrays to files when two evaluations first differ in one of these stages. */
int  batotp_hip_resampled_trace_data(batotp_resampled *r, int32_t stage, double *out, int64_t cap, int64_t *count);
int  batotp_hip_resampled_ms(batotp_resampled *r, float *ms);

/* ---- output stage behind the hot path ------------------------------------ */
/* Replaces BA::interpOutputData (batotp/ba.cpp:1661-1931): the forward curve s(t) of every path is
 * re-sampled at constant time steps, the path splines are evaluated there, the result is smoothed and
 * down-sampled by _outSmoothFact (smooth, batotp/util.cpp:263-290; Spline::interp1linear,
 * spline.cpp:108-120) and, when out_res is finer than the integration step, re-interpolated
 * (ba.cpp:1873-1919).  Covered: JOINT paths of a robot without kinematic model and without torque
 * constraints (joint rows); JOINT paths of KUKA / RR (joint rows, three Cartesian rows by the forward kinematics at the
 * output points and, with torque constraints, the serial-robot torque recomputation of ba.cpp:1791-1827 with Robot::dynRR or
 * the batch's chain model); BOTH paths (joint rows and pose rows, quaternions back to axis-angle at the end); and CART paths of the 3-cable robot with torque constraints (Cartesian rows,
 * cable lengths by Robot::invKinCSPR3DOF, cable tensions recomputed as in ba.cpp:1744-1790 with
 * Robot::dynCSPR3DOF / setA / solveLinSys).  The batch must have completed the forward sweep.
 * Anything else returns BATOTP_ERR_ARG (the caller keeps its host code). */
typedef struct batotp_output_params {
    int32_t  n_joints;               /* joint channels to produce (the batch's n_joints)       */
    int32_t  path_type;              /* BATOTP_PATH_JOINT (also 0) or BATOTP_PATH_CART         */
    double   integ_res;              /* _integRes                                              */
    double   out_res;                /* _outRes                                                */
    double   out_smooth_fact;        /* _outSmoothFact                                         */
} batotp_output_params;

typedef struct batotp_output batotp_output;

/* constant-time trajectories of the paths [path0, path0 + n_paths) of the batch; they stay in HBM */
int  batotp_hip_output(batotp_batch *batch, const batotp_output_params *prm, int32_t path0, int32_t n_paths,
                       batotp_output **out);
int  batotp_hip_output_destroy(batotp_output *o);
/* known-answer test of one step of that stage, through the stage's own launch function: Spline::findInterpSegs' cursor never moves
 * back (reference batotp/spline.cpp:56-99), i.e. the segment of an output site is the running maximum of the raw segment indices of
 * the sites before it.  seg holds the raw indices of n_paths paths back to back (n1[k] sites each, >= 0) and returns their running
 * maxima, one wavefront per path as in batotp_hip_output. */
int  batotp_hip_out_segmax_kat(batotp_ctx *ctx, int32_t n_paths, const int32_t *n1, int32_t *seg);
/* points per path (0 for a path whose sweep ended with an error status) and traj.sres of the output */
int  batotp_hip_output_info(batotp_output *o, int64_t *n_pts /* [n_paths] */, double *sres /* [n_paths] */);
/* rows per point: n_theta joint rows, then n_cart Cartesian rows and n_trq torque rows (0 and 0 for JOINT paths) */
int  batotp_hip_output_channels(batotp_output *o, int32_t *n_theta, int32_t *n_cart, int32_t *n_trq);
/* trajectory of path path0 + k: rows[n_theta + n_cart + n_trq][n_pts[k]] */
int  batotp_hip_output_download(batotp_output *o, int32_t k, double *rows);
/* all trajectories of the range in one copy: path after path, each [n_theta + n_cart + n_trq][n_pts] */
int  batotp_hip_output_download_all(batotp_output *o, double *rows);
/* device pointer of all trajectories, path after path, each [n_theta + n_cart + n_trq][n_pts] */
int  batotp_hip_output_device(batotp_output *o, const double **theta_dev, int64_t *n_doubles);
/* milliseconds of the stage's kernels (HIP events on the context's stream) */
int  batotp_hip_output_ms(batotp_output *o, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* BATOTP_HIP_H */
