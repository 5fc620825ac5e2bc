/*
 * abi_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Implements the C-ABI of include/batotp_hip.h on top of the CPU oracle so that the host-side BA
 * library (batotp_amd/host) can be exercised end to end on a machine without a GPU, and compared
 * with the outputs of the reference itself (tests/test_reference_pin.py), and so that bench.py has
 * a CPU baseline.  It is built only by oracle/Makefile into oracle/_build/ and linked only by test
 * programs; the product links batotp_amd/csrc/libbatotp_hip.so instead and has no CPU path.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "batotp_hip.h"
#include "batotp_models.h" /* robot parameter tables: data, shared with the product */
#include "batotp_oracle.h"

struct batotp_ctx { int device; int rs_trace; };

struct batotp_batch {
    batotp_problem prob;
    int32_t n_paths;
    int64_t cap;
    bo_path **path;
    double **in_y;       /* uploaded knot values per path */
    double *in_sres;
    double **trig;
    double **rev_s, **rev_sd, **fwd_s, **fwd_sd;
    batotp_path_result *res;
    batotp_serial_model serial;
    int has_serial;
    double *integ_res;   /* per path (batotp_hip_set_path_integ_res); default prob.integ_res */
    int kin_done;
    int mvc_stale;       /* BATOTP_F_MVC_IN_CURVES: a sweep has run since the last pointwise evaluation (state rule of the product) */
    int rev_stale, fwd_stale; /* BATOTP_F_MVC_IN_CURVES: a pointwise evaluation has overwritten that curve since its sweep (state rule of the product) */
    int rev_gone;        /* BATOTP_F_CURVES_IN_PLACE: the forward sweep has consumed the reverse curve (state rules of the product) */
    float ms[5];
};

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

int batotp_hip_device_count(int *count) { if (count) *count = 1; return BATOTP_OK; }
int batotp_hip_ctx_create(int device, batotp_ctx **out)
{
    if (!out) return BATOTP_ERR_ARG;
    *out = (batotp_ctx *)calloc(1, sizeof(batotp_ctx));
    (*out)->device = device;
    return BATOTP_OK;
}
int batotp_hip_ctx_destroy(batotp_ctx *ctx) { free(ctx); return BATOTP_OK; }
const char *batotp_hip_last_error(void) { return "(oracle shim)"; }
int batotp_hip_sdiv_kat(batotp_ctx *ctx, int64_t n, const double *a, const double *b, double *q, int32_t *in_window)
{
    (void)ctx;
    for (int64_t i = 0; i < n; ++i) { q[i] = a[i] / b[i]; in_window[i] = 0; }
    return BATOTP_OK;
}
int batotp_hip_spline_lanes_kat(batotp_ctx *ctx, int64_t n, const double *y, double *sol, double *sol_seq, int32_t *redone)
{
    if (!ctx || n < 4 || !y || !sol || !sol_seq || !redone) return BATOTP_ERR_ARG;
    bo_spline_sol(y, n, sol);
    memcpy(sol_seq, sol, sizeof(double) * (size_t)n);
    *redone = 0;
    return BATOTP_OK;
}
int batotp_hip_div6_kat(batotp_ctx *ctx, int64_t n, const double *a, double *q)
{
    int64_t i;
    if (!ctx || n <= 0 || !a || !q) return BATOTP_ERR_ARG;
    for (i = 0; i < n; ++i) q[i] = a[i] / 6.0;
    return BATOTP_OK;
}
int batotp_hip_fp64_kat(batotp_ctx *ctx, int64_t n, const double *a, const double *b, double *q, double *r, double *p)
{
    (void)ctx; (void)n; (void)a; (void)b; (void)q; (void)r; (void)p;
    return BATOTP_ERR_NO_DEVICE;
}
int batotp_hip_synchronize(batotp_ctx *ctx) { (void)ctx; return BATOTP_OK; }
int batotp_hip_set_sweep_group(batotp_ctx *ctx, int32_t lanes) { (void)ctx; (void)lanes; return BATOTP_OK; }
int batotp_hip_set_paths_per_wave(batotp_ctx *ctx, int32_t n) { (void)ctx; (void)n; return BATOTP_OK; }
int batotp_hip_set_sweep_hold(batotp_ctx *ctx, int32_t reverse, int32_t forward) { (void)ctx; (void)reverse; (void)forward; return BATOTP_OK; }
int batotp_hip_set_sweep_prefetch(batotp_ctx *ctx, int32_t reverse, int32_t forward) { (void)ctx; (void)reverse; (void)forward; return BATOTP_OK; }
int batotp_hip_set_spline_tiles(batotp_ctx *ctx, int32_t on) { (void)ctx; (void)on; return BATOTP_OK; }
int batotp_hip_set_flat_form(batotp_ctx *ctx, int32_t form) { (void)ctx; (void)form; return BATOTP_OK; }
int batotp_hip_set_fast_forward(batotp_ctx *ctx, int32_t on) { (void)ctx; (void)on; return BATOTP_OK; }
int batotp_hip_set_poison(batotp_ctx *ctx, int32_t on) { (void)ctx; (void)on; return BATOTP_OK; }
/* the stage trace of one-path calls (batotp_hip_resampled_trace / _trace_data further down, beside the resampler): the checker observes
 * the stages of its own sequential resampler (bo_resample_set_stage_observer) */
int batotp_hip_set_resample_trace(batotp_ctx *ctx, int32_t on) { if (!ctx) return BATOTP_ERR_ARG; ctx->rs_trace = on != 0; return BATOTP_OK; }
int batotp_hip_set_cert_hold(batotp_ctx *ctx, int32_t hold) { (void)ctx; return (hold >= -1 && hold <= 8) ? BATOTP_OK : BATOTP_ERR_ARG; }
int batotp_hip_set_path_order(batotp_ctx *ctx, int32_t mode) { (void)ctx; return (mode == 0 || mode == 1) ? BATOTP_OK : BATOTP_ERR_ARG; }
int batotp_hip_set_workspace_budget(batotp_ctx *ctx, int64_t resample_bytes, int64_t output_bytes) { (void)ctx; return (resample_bytes < 0 || output_bytes < 0) ? BATOTP_ERR_ARG : BATOTP_OK; }
int batotp_hip_set_k3_form(batotp_ctx *ctx, int32_t form) { (void)ctx; return (form == 0 || form == 1) ? BATOTP_OK : BATOTP_ERR_ARG; }
int batotp_hip_spline_tile_fallbacks(batotp_batch *b, int32_t *series) { if (!b || !series) return BATOTP_ERR_ARG; *series = 0; return BATOTP_OK; }
/* the checker has one loop form (the reference's); the introspection calls of the product answer accordingly */
int batotp_hip_flat_loop_status(batotp_ctx *ctx, int32_t *status) { if (!ctx || !status) return BATOTP_ERR_ARG; *status = -1; return BATOTP_OK; }
int batotp_hip_toolchain(char *built_with, char *validated_with, int32_t cap)
{
    if (cap < 1) return BATOTP_ERR_ARG;
    if (built_with) { strncpy(built_with, "cpu checker", (size_t)cap - 1); built_with[cap - 1] = 0; }
    if (validated_with) validated_with[0] = 0;
    return BATOTP_OK;
}
int batotp_hip_last_sweep_launch(batotp_batch *b, int32_t dir, int32_t *lanes, int32_t *paths_per_wave, int32_t *hold)
{
    if (!b || (dir != 1 && dir != -1)) return BATOTP_ERR_ARG;
    if (lanes) *lanes = 1;
    if (paths_per_wave) *paths_per_wave = 1;
    if (hold) *hold = -1;
    return BATOTP_OK;
}

int batotp_hip_batch_create(batotp_ctx *ctx, const batotp_problem *prob, int32_t n_paths,
                            const int64_t *n_knots, int64_t max_steps, batotp_batch **out)
{
    batotp_batch *b;
    int32_t p;
    (void)ctx;
    if (!prob || !n_knots || !out || n_paths < 1) return BATOTP_ERR_ARG;
    b = (batotp_batch *)calloc(1, sizeof(*b));
    b->prob = *prob;
    b->n_paths = n_paths;
    b->cap = max_steps;
    b->path = (bo_path **)calloc((size_t)n_paths, sizeof(void *));
    b->in_y = (double **)calloc((size_t)n_paths, sizeof(void *));
    b->in_sres = (double *)calloc((size_t)n_paths, sizeof(double));
    b->trig = (double **)calloc((size_t)n_paths, sizeof(void *));
    b->rev_s = (double **)calloc((size_t)n_paths, sizeof(void *));
    b->rev_sd = (double **)calloc((size_t)n_paths, sizeof(void *));
    b->fwd_s = (double **)calloc((size_t)n_paths, sizeof(void *));
    b->fwd_sd = (double **)calloc((size_t)n_paths, sizeof(void *));
    b->res = (batotp_path_result *)calloc((size_t)n_paths, sizeof(batotp_path_result));
    b->integ_res = (double *)calloc((size_t)n_paths, sizeof(double));
    for (p = 0; p < n_paths; p++) b->integ_res[p] = prob->integ_res;
    for (p = 0; p < n_paths; p++) {
        if (n_knots[p] < 2) return BATOTP_ERR_ARG;
        b->path[p] = bo_path_new(prob, n_knots[p]);
        if (!b->path[p]) return BATOTP_ERR_ALLOC;
    }
    *out = b;
    return BATOTP_OK;
}

int batotp_hip_batch_destroy(batotp_batch *b)
{
    int32_t p;
    if (!b) return BATOTP_OK;
    for (p = 0; p < b->n_paths; p++) {
        bo_path_free(b->path[p]);
        free(b->in_y[p]); free(b->trig[p]);
        free(b->rev_s[p]); free(b->rev_sd[p]); free(b->fwd_s[p]); free(b->fwd_sd[p]);
    }
    free(b->path); free(b->in_y); free(b->in_sres); free(b->trig);
    free(b->rev_s); free(b->rev_sd); free(b->fwd_s); free(b->fwd_sd); free(b->res); free(b->integ_res);
    free(b);
    return BATOTP_OK;
}

/* the problem as path p sees it: its own integration step (batotp_hip_set_path_integ_res) */
static batotp_problem path_prob(const batotp_batch *b, int32_t p)
{
    batotp_problem q = b->prob;
    q.integ_res = b->integ_res[p];
    return q;
}

int batotp_hip_set_path_integ_res(batotp_batch *b, int32_t path0, int32_t n, const double *integ_res)
{
    int32_t k;
    if (!b || !integ_res || path0 < 0 || n < 0 || path0 + n > b->n_paths) return BATOTP_ERR_ARG;
    for (k = 0; k < n; k++) {
        if ((!(integ_res[k] > 0) && integ_res[k] == integ_res[k]) || isinf(integ_res[k])) return BATOTP_ERR_ARG; /* positive and finite, or NaN */
        b->integ_res[path0 + k] = integ_res[k];
    }
    return BATOTP_OK;
}

int batotp_hip_upload_knots(batotp_batch *b, int32_t path0, int32_t n, const double *y, const double *sres)
{
    int32_t k;
    const double *src = y;
    if (!b || path0 < 0 || path0 + n > b->n_paths) return BATOTP_ERR_ARG;
    for (k = 0; k < n; k++) {
        bo_path *p = b->path[path0 + k];
        size_t cnt = (size_t)(p->n_theta + p->n_cart) * (size_t)p->n;
        free(b->in_y[path0 + k]);
        b->in_y[path0 + k] = (double *)malloc(cnt * sizeof(double));
        memcpy(b->in_y[path0 + k], src, cnt * sizeof(double));
        b->in_sres[path0 + k] = sres[k];
        src += cnt;
    }
    return BATOTP_OK;
}
int batotp_hip_upload_knots_device(batotp_batch *b, int32_t path0, int32_t n, const double *y, const double *sres)
{
    return batotp_hip_upload_knots(b, path0, n, y, sres);
}
int batotp_hip_upload_knots_device_rows(batotp_batch *b, int32_t path0, int32_t n, const double *y, int32_t src_rows, const double *sres)
{
    /* the first n_theta + n_cart of src_rows rows of every path (the checker's "device" memory is host memory) */
    const double *src = y;
    int32_t k;
    if (!b || !y || !sres || path0 < 0 || n < 1 || path0 + n > b->n_paths) return BATOTP_ERR_ARG;
    for (k = 0; k < n; k++) {
        bo_path *p = b->path[path0 + k];
        const int rows = p->n_theta + p->n_cart;
        int rc;
        if (src_rows < rows) return BATOTP_ERR_ARG;
        rc = batotp_hip_upload_knots(b, path0 + k, 1, src, sres + k);
        if (rc) return rc;
        src += (size_t)src_rows * (size_t)p->n;
    }
    return BATOTP_OK;
}
int batotp_hip_upload_rr_trig(batotp_batch *b, int32_t path, const double *trig)
{
    size_t cnt;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    cnt = 4 * (size_t)b->path[path]->n;
    free(b->trig[path]);
    b->trig[path] = (double *)malloc(cnt * sizeof(double));
    memcpy(b->trig[path], trig, cnt * sizeof(double));
    return BATOTP_OK;
}
int batotp_hip_set_serial_model(batotp_batch *b, const batotp_serial_model *model)
{
    int32_t p;
    if (!b || !model || model->n_links != b->prob.n_joints || model->n_links > BATOTP_MAX_LINKS) return BATOTP_ERR_ARG;
    if (!(b->prob.flags & BATOTP_F_TRQ_ON) || (b->prob.flags & BATOTP_F_PARALLEL)) return BATOTP_ERR_ARG;
    b->serial = *model;
    b->has_serial = 1;
    for (p = 0; p < b->n_paths; p++) b->path[p]->serial = &b->serial;
    return BATOTP_OK;
}
int batotp_hip_builtin_serial_model(int32_t robot_type, batotp_serial_model *out)
{
    if (!out) return BATOTP_ERR_ARG;
    return batotp_builtin_serial_model(robot_type, out) == 0 ? BATOTP_OK : BATOTP_ERR_ARG;
}
int batotp_hip_upload_joint_trig(batotp_batch *b, int32_t path, const double *trig)
{
    size_t cnt;
    if (!b || !trig || path < 0 || path >= b->n_paths || !b->has_serial) return BATOTP_ERR_ARG;
    cnt = 2 * (size_t)b->prob.n_joints * (size_t)b->path[path]->n;
    free(b->trig[path]);
    b->trig[path] = (double *)malloc(cnt * sizeof(double));
    memcpy(b->trig[path], trig, cnt * sizeof(double));
    return BATOTP_OK;
}
int batotp_hip_upload_path_sites(batotp_batch *b, int32_t path, const double *sites, double vfact, double afact, int32_t parallel_now)
{
    bo_path *p;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    p = b->path[path];
    memcpy(p->sC, sites, sizeof(double) * (size_t)p->n);
    p->vfact = vfact; p->afact = afact; p->parallel_now = parallel_now;
    b->kin_done = 1;
    return BATOTP_OK;
}
int batotp_hip_upload_coeffs(batotp_batch *b, int32_t path, int32_t channel, const double *c)
{
    bo_path *p;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    p = b->path[path];
    if (channel < 0 || channel >= p->n_ch) return BATOTP_ERR_ARG;
    memcpy(p->coef + (size_t)channel * 4 * (size_t)p->n, c, sizeof(double) * 4 * (size_t)p->n);
    return BATOTP_OK;
}
static void set_curve(double **s, double **sd, int32_t path, const double *a, const double *c, int64_t n)
{
    free(s[path]); free(sd[path]);
    s[path] = (double *)malloc(sizeof(double) * (size_t)n);
    sd[path] = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(s[path], a, sizeof(double) * (size_t)n);
    memcpy(sd[path], c, sizeof(double) * (size_t)n);
}
int batotp_hip_upload_curve(batotp_batch *b, int32_t path, const double *s, const double *sdot, int64_t n)
{
    if (!b || path < 0 || path >= b->n_paths || n < 2) return BATOTP_ERR_ARG;
    set_curve(b->rev_s, b->rev_sd, path, s, sdot, n);
    b->res[path].n_rev = n;
    b->rev_gone = 0;
    b->rev_stale = 0;
    return BATOTP_OK;
}

int batotp_hip_upload_forward_curve(batotp_batch *b, int32_t path, const double *s, const double *sdot, int64_t n, double t_total)
{
    if (!b || path < 0 || path >= b->n_paths || n < 2 || !s || !sdot) return BATOTP_ERR_ARG;
    if (b->n_paths != 1) return BATOTP_ERR_STATE; /* state rule of the product: a batch of one */
    set_curve(b->fwd_s, b->fwd_sd, path, s, sdot, n);
    b->res[path].n_fwd = n;
    b->res[path].steps_fwd = n - 1;
    b->res[path].t_total = t_total;
    b->res[path].status_fwd = 0;
    b->fwd_stale = 0;
    return BATOTP_OK;
}

int batotp_hip_precompute(batotp_batch *b, int32_t stage)
{
    int32_t p;
    int bad = 0;
    double t0 = now_ms();
    if (!b) return BATOTP_ERR_ARG;
    for (p = 0; p < b->n_paths; p++)
        if ((stage == 0 || stage == 1) && !b->in_y[p]) return BATOTP_ERR_STATE;
    #pragma omp parallel for schedule(dynamic, 1)
    for (p = 0; p < b->n_paths; p++) {
        if (stage == 0 || stage == 1) {
            if (bo_precompute_kin(&b->prob, b->path[p], b->in_y[p], b->in_sres[p]) != 0)
                b->res[p].status_rev |= BATOTP_ST_SEG_ERROR;
        }
        if (stage == 0 || stage == 2) {
            if (bo_precompute_dyn(&b->prob, b->path[p], b->trig[p]) != 0) bad = 1;
        }
    }
    if (bad) return BATOTP_ERR_ARG;
    b->kin_done = 1;
    b->ms[1] = (float)(now_ms() - t0);
    return BATOTP_OK;
}

int batotp_hip_pointwise_mvc(batotp_batch *b)
{
    int32_t p;
    double t0 = now_ms();
    if (!b || !b->kin_done) return BATOTP_ERR_STATE;
    #pragma omp parallel for schedule(dynamic, 1)
    for (p = 0; p < b->n_paths; p++) { const batotp_problem q = path_prob(b, p); bo_pointwise_mvc(&q, b->path[p]); }
    b->ms[2] = (float)(now_ms() - t0);
    b->mvc_stale = 0;
    if (b->prob.flags & BATOTP_F_MVC_IN_CURVES) { b->rev_stale = 1; b->fwd_stale = 1; }
    return BATOTP_OK;
}

int batotp_hip_sweep(batotp_batch *b, int32_t dir)
{
    int32_t p;
    double t0 = now_ms();
    if (!b || !b->kin_done || (dir != 1 && dir != -1)) return BATOTP_ERR_STATE;
    if (dir == 1 && b->rev_gone) return BATOTP_ERR_STATE;
    b->mvc_stale = 1;
    for (p = 0; p < b->n_paths; p++)
        if (dir == 1 && !b->rev_s[p]) return BATOTP_ERR_STATE;
    #pragma omp parallel for schedule(dynamic, 1)
    for (p = 0; p < b->n_paths; p++) {
        double *s = (double *)malloc(sizeof(double) * (size_t)b->cap);
        double *sd = (double *)malloc(sizeof(double) * (size_t)b->cap);
        int64_t n = 0, steps = 0;
        double T = 0;
        uint32_t st = 0;
        int32_t nf = 0;
        batotp_path_result *r = &b->res[p];
        if (dir == 1 && r->n_rev < 2) {
            /* no reverse curve to follow: same convention as the HIP kernel */
            free(s); free(sd);
            r->n_fwd = 0; r->steps_fwd = 0; r->t_total = 0; r->status_fwd = r->status_rev | BATOTP_ST_CAPACITY; r->n_bisect_fail_fwd = 0;
            continue;
        }
        const batotp_problem q = path_prob(b, p);
        bo_sweep_ex(&q, b->path[p], dir, b->rev_s[p], b->rev_sd[p], r->n_rev, s, sd, b->cap, &n, &steps, &T, &st, &nf,
                    (b->prob.flags & BATOTP_F_CURVES_IN_PLACE) != 0);
        if (dir == -1) {
            free(b->rev_s[p]); free(b->rev_sd[p]);
            b->rev_s[p] = s; b->rev_sd[p] = sd;
            r->n_rev = n; r->t_rev = T; r->status_rev = (r->status_rev & BATOTP_ST_SEG_ERROR) | st; r->n_bisect_fail_rev = nf;
            r->steps_rev = steps;
        } else {
            free(b->fwd_s[p]); free(b->fwd_sd[p]);
            b->fwd_s[p] = s; b->fwd_sd[p] = sd;
            r->n_fwd = n; r->t_total = T; r->status_fwd = st; r->n_bisect_fail_fwd = nf;
            r->steps_fwd = steps;
        }
    }
    b->ms[dir == -1 ? 3 : 4] = (float)(now_ms() - t0);
    if (dir == -1) b->rev_stale = 0; else b->fwd_stale = 0;
    if (dir == -1) b->rev_gone = 0;
    else if (b->prob.flags & BATOTP_F_CURVES_IN_PLACE) b->rev_gone = 1; /* the checker keeps both curves; the state rule and the capacity margin (bo_sweep_ex) are mirrored */
    return BATOTP_OK;
}

int batotp_hip_optimize(batotp_batch *b)
{
    int rc = batotp_hip_precompute(b, 0);
    if (rc) return rc;
    rc = batotp_hip_sweep(b, -1);
    if (rc) return rc;
    return batotp_hip_sweep(b, 1);
}

int batotp_hip_get_results(batotp_batch *b, batotp_path_result *out)
{
    if (!b || !out) return BATOTP_ERR_ARG;
    memcpy(out, b->res, sizeof(batotp_path_result) * (size_t)b->n_paths);
    return BATOTP_OK;
}

int batotp_hip_download_curve(batotp_batch *b, int32_t path, int32_t which, double *s, double *sdot, int64_t cap, int64_t *n)
{
    int64_t m, avail;
    const double *ss, *sd;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    if (which == -1 && b->rev_gone) return BATOTP_ERR_STATE;
    if (which == -1 ? b->rev_stale : b->fwd_stale) return BATOTP_ERR_STATE;
    ss = which == 1 ? b->fwd_s[path] : b->rev_s[path];
    sd = which == 1 ? b->fwd_sd[path] : b->rev_sd[path];
    avail = which == 1 ? b->res[path].n_fwd : b->res[path].n_rev;
    if (avail <= 0) { if (n) *n = 0; return BATOTP_OK; }
    if (!ss) return BATOTP_ERR_STATE;
    m = avail < cap ? avail : cap;
    if (s) memcpy(s, ss, sizeof(double) * (size_t)m);
    if (sdot) memcpy(sdot, sd, sizeof(double) * (size_t)m);
    if (n) *n = avail;
    return BATOTP_OK;
}

int batotp_hip_download_coeffs(batotp_batch *b, int32_t path, int32_t channel, double *c)
{
    bo_path *p;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    p = b->path[path];
    if (channel < 0 || channel >= p->n_ch) return BATOTP_ERR_ARG;
    memcpy(c, p->coef + (size_t)channel * 4 * (size_t)p->n, sizeof(double) * 4 * (size_t)p->n);
    return BATOTP_OK;
}
int batotp_hip_download_samples(batotp_batch *b, int32_t path, int32_t channel, double *out)
{
    bo_path *p;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    p = b->path[path];
    if (channel < 0 || channel >= p->n_theta + p->n_cart) return BATOTP_ERR_ARG;
    memcpy(out, p->samp + (size_t)channel * 3 * (size_t)p->n, sizeof(double) * 3 * (size_t)p->n);
    return BATOTP_OK;
}
int batotp_hip_download_dyn(batotp_batch *b, int32_t path, int32_t k, int32_t row, double *out)
{
    bo_path *p;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    p = b->path[path];
    if (k < 1 || k > 4 || row < 0 || row >= p->dyn_dim) return BATOTP_ERR_ARG;
    memcpy(out, p->dyn + ((size_t)(k - 1) * (size_t)p->dyn_dim + (size_t)row) * (size_t)p->n, sizeof(double) * (size_t)p->n);
    return BATOTP_OK;
}
int batotp_hip_download_mvc(batotp_batch *b, int32_t path, double *sdot_max, double *sddot_l, double *sddot_h)
{
    bo_path *p;
    size_t n;
    if (!b || path < 0 || path >= b->n_paths) return BATOTP_ERR_ARG;
    if ((b->prob.flags & BATOTP_F_MVC_IN_CURVES) && b->mvc_stale) return BATOTP_ERR_STATE;
    p = b->path[path]; n = (size_t)p->n;
    if (sdot_max) memcpy(sdot_max, p->mvc, sizeof(double) * n);
    if (sddot_l) memcpy(sddot_l, p->mvc + n, sizeof(double) * n);
    if (sddot_h) memcpy(sddot_h, p->mvc + 2 * n, sizeof(double) * n);
    return BATOTP_OK;
}
int batotp_hip_results_device_ptr(batotp_batch *b, void **ptr, int64_t *bytes)
{
    if (!b) return BATOTP_ERR_ARG;
    if (ptr) *ptr = b->res;
    if (bytes) *bytes = (int64_t)sizeof(batotp_path_result) * b->n_paths;
    return BATOTP_OK;
}
int batotp_hip_pack_curves(batotp_batch *b, int32_t which, int32_t path0, int32_t n_paths, void *dst, int64_t dst_points, int64_t *total_points)
{
    int32_t k;
    int64_t total = 0, i;
    double *o = (double *)dst;
    if (!b || (which != 1 && which != -1) || path0 < 0 || n_paths < 0 || path0 + n_paths > b->n_paths || !total_points) return BATOTP_ERR_ARG;
    if (which == -1 && b->rev_gone) { *total_points = 0; return BATOTP_ERR_STATE; }
    if (which == -1 ? b->rev_stale : b->fwd_stale) { *total_points = 0; return BATOTP_ERR_STATE; }
    for (k = 0; k < n_paths; k++) total += which == 1 ? b->res[path0 + k].n_fwd : b->res[path0 + k].n_rev;
    *total_points = total;
    if (total == 0) return BATOTP_OK;
    if (!dst || dst_points < total) return BATOTP_ERR_ARG;
    for (k = 0; k < n_paths; k++) {
        const int64_t cnt = which == 1 ? b->res[path0 + k].n_fwd : b->res[path0 + k].n_rev;
        const double *s = which == 1 ? b->fwd_s[path0 + k] : b->rev_s[path0 + k];
        const double *sd = which == 1 ? b->fwd_sd[path0 + k] : b->rev_sd[path0 + k];
        for (i = 0; i < cnt; i++) { *o++ = s[i]; *o++ = sd[i]; }
    }
    return BATOTP_OK;
}
int batotp_hip_last_kernel_ms(batotp_batch *b, int32_t which, float *ms)
{
    if (!b || which < 1 || which > 4 || !ms) return BATOTP_ERR_ARG;
    *ms = b->ms[which];
    return BATOTP_OK;
}
int batotp_hip_batch_bytes(batotp_batch *b, int64_t *bytes) { (void)b; if (bytes) *bytes = 0; return BATOTP_OK; }

/* ---- path resampling over bo_resample --------------------------------------- */
struct batotp_resampled {
    int32_t n_paths;
    int C;
    int64_t *n, *off;
    double *sres;
    uint32_t *status;
    double *y; /* concatenated knots, the layout batotp_hip_upload_knots expects */
    double *au; /* per path: integ_res, s_weights[3], scale_type (automatic integration resolution) */
    float ms;
    /* stage trace of a one-path call (batotp_hip_set_resample_trace): the checksum of every stage, copies of stages 2 and 3 */
    int traced;
    uint64_t trace[8];
    double *tdata[2];
    int64_t tcnt[2];
};

static uint64_t shim_mix64(uint64_t x);
static uint64_t shim_checksum(const double *y, int64_t total)
{
    uint64_t h = 0;
    int64_t i;
    for (i = 0; i < total; i++) {
        uint64_t bits;
        memcpy(&bits, y + i, sizeof(bits));
        h += shim_mix64(bits ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull));
    }
    return h;
}
static void shim_observe_stage(void *user, int stage, const double *data, int64_t count)
{
    batotp_resampled *r = (batotp_resampled *)user;
    if (stage < 0 || stage > 7 || count < 0) return;
    r->trace[stage] = shim_checksum(data, count);
    if (stage == 2 || stage == 3) {
        const int k = stage - 2;
        free(r->tdata[k]);
        r->tdata[k] = (double *)malloc(sizeof(double) * (size_t)(count ? count : 1));
        memcpy(r->tdata[k], data, sizeof(double) * (size_t)count);
        r->tcnt[k] = count;
    }
}

int batotp_hip_set_overlap(batotp_ctx *ctx, int32_t on) { (void)on; return ctx ? BATOTP_OK : BATOTP_ERR_ARG; }
int batotp_hip_ctx_trim(batotp_ctx *ctx) { return ctx ? BATOTP_OK : BATOTP_ERR_ARG; }

int batotp_hip_resampled_destroy(batotp_resampled *r)
{
    if (!r) return BATOTP_OK;
    free(r->n); free(r->off); free(r->sres); free(r->status); free(r->y); free(r->au);
    free(r->tdata[0]); free(r->tdata[1]);
    free(r);
    return BATOTP_OK;
}

/* TEST INFRASTRUCTURE: fault injection for the host library's "evaluate twice" guard (BA::interpInputData, BA::optimizeBatch).
 * BATOTP_SHIM_RESAMPLE_FAULT = comma-separated call numbers (1-based, per process): those calls of batotp_hip_resample return knots with
 * one value of the last path moved by one ulp and status 0 -- what the unexplained event of round 4 looked like from outside. */
static void shim_inject_resample_fault(batotp_resampled *r, int64_t total)
{
    static int calls = 0;
    const char *env = getenv("BATOTP_SHIM_RESAMPLE_FAULT");
    int me;
#pragma omp atomic capture
    me = ++calls;
    if (!env || total < 1) return;
    while (*env) {
        char *end;
        const long k = strtol(env, &end, 10);
        if (end == env) break;
        if (k == me) {
            const int p = r->n_paths - 1;
            if (r->n[p] > 0 && !r->status[p]) {
                double *v = r->y + r->off[p] * r->C + (r->n[p] * r->C) / 2;
                *v = nextafter(*v, 1e300);
                if (r->traced) {
                    /* a traced call shows the fault the way the event of round 6 looked: stages 0 and 1 as they were, one value of
                     * stage 2 moved by an ulp, every later stage different */
                    int k;
                    if (r->tcnt[0] > 0) {
                        double *w = r->tdata[0] + r->tcnt[0] / 2;
                        *w = nextafter(*w, 1e300);
                        r->trace[2] = shim_checksum(r->tdata[0], r->tcnt[0]);
                    }
                    for (k = 3; k < 7; ++k) r->trace[k] ^= 1;
                    r->trace[7] = shim_checksum(r->y + r->off[p] * r->C, r->n[p] * r->C);
                }
            }
        }
        env = (*end == ',') ? end + 1 : end;
    }
}

int batotp_hip_resample(batotp_ctx *ctx, const batotp_resample_params *prm, int32_t n_paths, const int64_t *n_in, const double *x,
                        const double *sres_in, batotp_resampled **out)
{
    batotp_resampled *r;
    double **ys;
    int64_t *xoff, total = 0;
    int p, bad = 0, C;
    struct timespec t0, t1;
    if (!ctx || !prm || !n_in || !x || !sres_in || !out || n_paths < 1) return BATOTP_ERR_ARG;
    *out = NULL;
    const int Cin = prm->n_joints + prm->n_cart;
    C = Cin + ((prm->path_type == BATOTP_PATH_BOTH && prm->n_cart == 6) ? 1 : 0);   /* poses leave as position + quaternion */
    for (p = 0; p < n_paths; ++p)
        if (n_in[p] < 4) return BATOTP_ERR_ARG;
    r = (batotp_resampled *)calloc(1, sizeof(*r));
    r->n_paths = n_paths; r->C = C;
    r->n = (int64_t *)calloc((size_t)n_paths, sizeof(int64_t));
    r->off = (int64_t *)calloc((size_t)n_paths, sizeof(int64_t));
    r->sres = (double *)calloc((size_t)n_paths, sizeof(double));
    r->status = (uint32_t *)calloc((size_t)n_paths, sizeof(uint32_t));
    r->au = (double *)calloc((size_t)n_paths * 5, sizeof(double));
    ys = (double **)calloc((size_t)n_paths, sizeof(double *));
    xoff = (int64_t *)calloc((size_t)n_paths, sizeof(int64_t));
    for (p = 1; p < n_paths; ++p) xoff[p] = xoff[p - 1] + n_in[p - 1] * Cin;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    r->traced = ctx->rs_trace && n_paths == 1;
    if (r->traced) {
        /* one path, on the calling thread, its stages observed */
        bo_resample_set_stage_observer(shim_observe_stage, r);
        if (bo_resample_auto(prm, n_in[0], x, sres_in[0], &ys[0], &r->n[0], &r->sres[0], &r->status[0], r->au) != 0) bad = 1;
        bo_resample_set_stage_observer(NULL, NULL);
    } else
#pragma omp parallel for schedule(dynamic, 1)
    for (p = 0; p < n_paths; ++p)
        if (bo_resample_auto(prm, n_in[p], x + xoff[p], sres_in[p], &ys[p], &r->n[p], &r->sres[p], &r->status[p], r->au + 5 * (size_t)p) != 0) {
#pragma omp atomic write
            bad = 1;
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    r->ms = (float)((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
    if (bad) {
        for (p = 0; p < n_paths; ++p) free(ys[p]);
        free(ys); free(xoff);
        batotp_hip_resampled_destroy(r);
        return BATOTP_ERR_ARG;
    }
    for (p = 0; p < n_paths; ++p) { r->off[p] = total; total += r->n[p]; }
    r->y = (double *)malloc(sizeof(double) * (size_t)total * C);
    for (p = 0; p < n_paths; ++p) {
        memcpy(r->y + r->off[p] * C, ys[p], sizeof(double) * (size_t)r->n[p] * C);
        free(ys[p]);
    }
    free(ys); free(xoff);
    shim_inject_resample_fault(r, total);
    *out = r;
    return BATOTP_OK;
}

int batotp_hip_resampled_auto(batotp_resampled *r, double *integ_res, double *s_weights, int32_t *scale_type)
{
    int p, k;
    if (!r) return BATOTP_ERR_ARG;
    for (p = 0; p < r->n_paths; ++p) {
        if (integ_res) integ_res[p] = r->au[5 * p];
        if (s_weights) for (k = 0; k < 3; ++k) s_weights[3 * p + k] = r->au[5 * p + 1 + k];
        if (scale_type) scale_type[p] = (int32_t)r->au[5 * p + 4];
    }
    return BATOTP_OK;
}

int batotp_hip_resampled_info(batotp_resampled *r, int64_t *n_knots, double *sres, uint32_t *status)
{
    int p;
    if (!r) return BATOTP_ERR_ARG;
    for (p = 0; p < r->n_paths; ++p) {
        if (n_knots) n_knots[p] = r->n[p];
        if (sres) sres[p] = r->sres[p];
        if (status) status[p] = r->status[p];
    }
    return BATOTP_OK;
}

int batotp_hip_resampled_knots_device(batotp_resampled *r, const double **y_dev, int64_t *n_doubles)
{
    if (!r || !y_dev) return BATOTP_ERR_ARG;
    *y_dev = r->y;
    if (n_doubles) *n_doubles = (r->off[r->n_paths - 1] + r->n[r->n_paths - 1]) * r->C;
    return BATOTP_OK;
}

int batotp_hip_resampled_download(batotp_resampled *r, int32_t path, double *y)
{
    if (!r || !y || path < 0 || path >= r->n_paths) return BATOTP_ERR_ARG;
    memcpy(y, r->y + r->off[path] * r->C, sizeof(double) * (size_t)r->n[path] * r->C);
    return BATOTP_OK;
}

static uint64_t shim_mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
int batotp_hip_resampled_checksums(batotp_resampled *r, uint64_t *sums)
{
    int32_t p;
    if (!r || !sums) return BATOTP_ERR_ARG;
    for (p = 0; p < r->n_paths; p++) {
        const int64_t total = r->status[p] ? 0 : r->n[p] * r->C;
        const double *y = r->y + r->off[p] * r->C;
        uint64_t h = 0;
        int64_t i;
        for (i = 0; i < total; i++) {
            uint64_t bits;
            memcpy(&bits, y + i, sizeof(bits));
            h += shim_mix64(bits ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull));
        }
        sums[p] = h;
    }
    return BATOTP_OK;
}

int batotp_hip_resampled_trace(batotp_resampled *r, uint64_t *sums)
{
    int k;
    if (!r || !sums) return BATOTP_ERR_ARG;
    if (!r->traced) return BATOTP_ERR_STATE;
    for (k = 0; k < 8; ++k) sums[k] = r->trace[k];
    return BATOTP_OK;
}
int batotp_hip_resampled_trace_data(batotp_resampled *r, int32_t stage, double *out, int64_t cap, int64_t *count)
{
    if (!r || !count || stage < 0 || stage > 7) return BATOTP_ERR_ARG;
    if (!r->traced) return BATOTP_ERR_STATE;
    *count = (stage == 2 || stage == 3) ? r->tcnt[stage - 2] : 0;
    if (out) {
        if (cap < *count) return BATOTP_ERR_ARG;
        if (*count) memcpy(out, r->tdata[stage - 2], sizeof(double) * (size_t)*count);
    }
    return BATOTP_OK;
}

int batotp_hip_resampled_ms(batotp_resampled *r, float *ms)
{
    if (!r || !ms) return BATOTP_ERR_ARG;
    *ms = r->ms;
    return BATOTP_OK;
}

/* ---- output stage over bo_output ----------------------------------------------- */
struct batotp_output {
    int32_t n_paths;
    int nJ;      /* rows per point: theta + cart + trq */
    int nTheta, nCart, nTrq;
    int64_t *n, *off;
    double *sres;
    double *theta; /* path after path, [nJ][n] each */
    float ms;
};

/* findInterpSegs' cursor never moves back (reference batotp/spline.cpp:56-99): running maximum per path */
int batotp_hip_out_segmax_kat(batotp_ctx *ctx, int32_t n_paths, const int32_t *n1, int32_t *seg)
{
    if (!ctx || n_paths < 1 || !n1 || !seg) return BATOTP_ERR_ARG;
    for (int k = 0; k < n_paths; ++k)
    {
        if (n1[k] < 0) return BATOTP_ERR_ARG;
        for (int i = 1; i < n1[k]; ++i)
            if (seg[i] < seg[i - 1]) seg[i] = seg[i - 1];
        seg += n1[k];
    }
    return BATOTP_OK;
}

int batotp_hip_output_destroy(batotp_output *o)
{
    if (!o) return BATOTP_OK;
    free(o->n); free(o->off); free(o->sres); free(o->theta);
    free(o);
    return BATOTP_OK;
}

int batotp_hip_output(batotp_batch *b, const batotp_output_params *prm, int32_t path0, int32_t n_paths, batotp_output **out)
{
    batotp_output *o;
    double **th;
    int64_t total = 0;
    int k, bad = 0;
    struct timespec t0, t1;
    if (!b || !prm || !out || path0 < 0 || n_paths < 1 || path0 + n_paths > b->n_paths) return BATOTP_ERR_ARG;
    *out = NULL;
    if (prm->n_joints != b->prob.n_joints || !(prm->out_res > 0) || !(prm->integ_res > 0) || !(prm->out_smooth_fact >= 1))
        return BATOTP_ERR_ARG;
    for (k = 0; k < n_paths; ++k)
        if (b->integ_res[path0 + k] != prm->integ_res) return BATOTP_ERR_ARG; /* one call per integration step (batotp_hip_set_path_integ_res) */
    {
        const int cable = prm->path_type == BATOTP_PATH_CART && b->prob.robot_type == BATOTP_ROBOT_CSPR3DOF && prm->n_joints == 3 &&
                          b->prob.n_cart == 3 && (b->prob.flags & BATOTP_F_TRQ_ON) && (b->prob.flags & BATOTP_F_PARALLEL);
        const int jointPath = prm->path_type == BATOTP_PATH_JOINT || prm->path_type == 0;
        const int kin = jointPath && bo_fwdkin_trig_rows(b->prob.robot_type, prm->n_joints) != 0 && b->prob.n_cart >= 3;
        const int serialTrq = jointPath && (b->prob.flags & BATOTP_F_TRQ_ON) && !(b->prob.flags & BATOTP_F_PARALLEL) &&
                              (b->has_serial || b->prob.robot_type == BATOTP_ROBOT_RR);
        const int joint = jointPath && (!(b->prob.flags & BATOTP_F_TRQ_ON) || serialTrq);
        const int both = prm->path_type == BATOTP_PATH_BOTH && b->prob.n_cart >= 3 && !(b->prob.flags & BATOTP_F_TRQ_ON);
        const int pose = both && b->prob.n_cart == 7;
        if (!cable && !joint && !both) return BATOTP_ERR_ARG;
        o = (batotp_output *)calloc(1, sizeof(*o));
        o->nTheta = prm->n_joints; o->nCart = cable ? 3 : (pose ? 6 : (both ? b->prob.n_cart : (kin ? 3 : 0))); o->nTrq = cable ? 3 : (serialTrq ? prm->n_joints : 0);
    }
    o->n_paths = n_paths; o->nJ = o->nTheta + o->nCart + o->nTrq;
    o->n = (int64_t *)calloc((size_t)n_paths, sizeof(int64_t));
    o->off = (int64_t *)calloc((size_t)n_paths, sizeof(int64_t));
    o->sres = (double *)calloc((size_t)n_paths, sizeof(double));
    th = (double **)calloc((size_t)n_paths, sizeof(double *));
    clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(dynamic, 1)
    for (k = 0; k < n_paths; ++k) {
        const int p = path0 + k;
        const batotp_path_result *r = &b->res[p];
        const uint32_t st = r->status_rev | r->status_fwd;
        if (r->n_fwd < 4 || (st & (BATOTP_ST_MAX_INTEG_TIME | BATOTP_ST_CAPACITY | BATOTP_ST_NONFINITE))) continue;
        {
            const double t_step = (r->status_fwd & BATOTP_ST_SHORT) ? r->t_total / 3. : prm->integ_res;
            int32_t nc = 0, nt = 0;
            if (bo_output(&b->prob, prm, b->path[p], b->fwd_s[p], r->n_fwd, t_step, &th[k], &nc, &nt, &o->n[k], &o->sres[k]) != 0) {
#pragma omp atomic write
                bad = 1;
            }
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    o->ms = (float)((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
    for (k = 0; k < n_paths; ++k) { o->off[k] = total; total += o->n[k] * o->nJ; }
    o->theta = (double *)malloc(sizeof(double) * (size_t)(total ? total : 1));
    for (k = 0; k < n_paths; ++k) {
        if (th[k]) memcpy(o->theta + o->off[k], th[k], sizeof(double) * (size_t)(o->n[k] * o->nJ));
        free(th[k]);
    }
    free(th);
    if (bad) { batotp_hip_output_destroy(o); return BATOTP_ERR_ARG; }
    *out = o;
    return BATOTP_OK;
}

int batotp_hip_output_info(batotp_output *o, int64_t *n_pts, double *sres)
{
    int k;
    if (!o) return BATOTP_ERR_ARG;
    for (k = 0; k < o->n_paths; ++k) {
        if (n_pts) n_pts[k] = o->n[k];
        if (sres) sres[k] = o->sres[k];
    }
    return BATOTP_OK;
}

int batotp_hip_output_channels(batotp_output *o, int32_t *n_theta, int32_t *n_cart, int32_t *n_trq)
{
    if (!o) return BATOTP_ERR_ARG;
    if (n_theta) *n_theta = o->nTheta;
    if (n_cart) *n_cart = o->nCart;
    if (n_trq) *n_trq = o->nTrq;
    return BATOTP_OK;
}

int batotp_hip_output_download(batotp_output *o, int32_t k, double *theta)
{
    if (!o || !theta || k < 0 || k >= o->n_paths) return BATOTP_ERR_ARG;
    memcpy(theta, o->theta + o->off[k], sizeof(double) * (size_t)(o->n[k] * o->nJ));
    return BATOTP_OK;
}

int batotp_hip_output_download_all(batotp_output *o, double *rows)
{
    int64_t total;
    if (!o || !rows) return BATOTP_ERR_ARG;
    total = o->off[o->n_paths - 1] + o->n[o->n_paths - 1] * o->nJ;
    memcpy(rows, o->theta, sizeof(double) * (size_t)total);
    return BATOTP_OK;
}

int batotp_hip_output_device(batotp_output *o, const double **theta_dev, int64_t *n_doubles)
{
    if (!o || !theta_dev) return BATOTP_ERR_ARG;
    *theta_dev = o->theta;
    if (n_doubles) *n_doubles = o->off[o->n_paths - 1] + o->n[o->n_paths - 1] * o->nJ;
    return BATOTP_OK;
}

int batotp_hip_output_ms(batotp_output *o, float *ms)
{
    if (!o || !ms) return BATOTP_ERR_ARG;
    *ms = o->ms;
    return BATOTP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ORACLE-ONLY test hooks (not part of include/batotp_hip.h): one call of the reference's       */
/* per-point routines on a path of a batch whose precompute has run -- replays of the fp64       */
/* known-answer vectors read out of the reference binary (oracle/make_golden_f64.py)             */
/* ------------------------------------------------------------------------------------------ */
int batotp_oracle_kat_accel(batotp_batch *b, int32_t path, int32_t dir, double s_cur, double sdot_cur, int64_t cur_seg_c, double sddot_in,
                            double *out5 /* sdotCur, sddot, sddotL, sddotH, curSegC */, int32_t *n_iter, int32_t *rc)
{
    int64_t seg = 0;
    if (!b || path < 0 || path >= b->n_paths || !b->kin_done || !out5 || !n_iter || !rc) return BATOTP_ERR_ARG;
    bo_kat_accel(&b->prob, b->path[path], dir, s_cur, sdot_cur, cur_seg_c, sddot_in, &out5[0], &out5[1], &out5[2], &out5[3], n_iter, rc, &seg);
    out5[4] = (double)seg;
    return BATOTP_OK;
}

int batotp_oracle_kat_sdot_lim(batotp_batch *b, int32_t path, int32_t dir, double s_cur, double sdot_in, double sdot_min,
                               const double *theta_d_pt, double cart_coeff0, const double *mvc_s, const double *mvc_sdot, int64_t n_mvc,
                               int64_t cur_seg_mvc, double *out2 /* sdot, curSegMVC */)
{
    int64_t seg = 0;
    if (!b || path < 0 || path >= b->n_paths || !b->kin_done || !theta_d_pt || !out2) return BATOTP_ERR_ARG;
    bo_kat_sdot_lim(&b->prob, b->path[path], dir, s_cur, sdot_in, sdot_min, theta_d_pt, cart_coeff0, mvc_s, mvc_sdot, n_mvc, cur_seg_mvc, &out2[0], &seg);
    out2[1] = (double)seg;
    return BATOTP_OK;
}
