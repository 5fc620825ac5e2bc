// batotp_hip.hip -- implementation of the C-ABI declared in include/batotp_hip.h for MI355X
// (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see Makefile).
// Host side: device context, batch memory management, launches, HIP-event timing.  The kernels
// are in kernels.hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "batotp_hip.h"
#include "batotp_models.h"
#include "kernels.hip.h"
#include "sweep1.hip.h"
#include "sweep8.hip.h"
#include "pointwise_va.hip.h"
#include "spline_stream.hip.h"
#include "spline_tile.hip.h"
#include "spline_lanes.hip.h"
#include "resample.hip.h"
#include "output.hip.h"

// Lanes per workgroup of a lane-per-series Thomas solve outside the hot path (resampler, output stage).  Every lane walks its own
// stream, so a load or store instruction costs its compute unit's address path one cycle per line it touches on top of a fixed ~20
// (measured on the resampler's 10 240 series of 10^5 knots: 73 cycles per memory instruction of a 64-lane wavefront, 30 of an
// 8-lane one), and the wavefronts of a compute unit queue for it (eight-lane wavefronts, five per compute unit: 2x SLOWER than full
// ones).  So: one wavefront per compute unit, as narrow as that allows.
static inline unsigned seriesBlock(int64_t series)
{
   const int64_t lanes = (series + 255) / 256;
   return (unsigned)(lanes < 4 ? 4 : (lanes > 64 ? 64 : lanes));
}

// A wavefront per series (spline_lanes.hip.h) or a lane per series?  The wavefront walks its series with 64 uncoalesced streams, the
// cost of which the compute units share: 3 us per series of 1e5 values, whatever their number, against 14 ms for ANY number of
// series up to one per lane of the chip (measured on the resampler: 10 240 series 30.8 ms against 14.0 ms; output stage, 256 series
// of 2.2e5 values: 1.3 ms against 16.7 ms).
static inline bool seriesLanesPay(int64_t series) { return series <= 4096; }

using namespace bk;

// k_sweep8, reverse sweep: the certificate block runs once this many eighths of a wavefront's live paths wait for it (measured on the
// headline batch, profiles/r06_a_*; batotp_hip_set_cert_hold overrides)
#ifndef BATOTP_CERT_HOLD_DEFAULT
#define BATOTP_CERT_HOLD_DEFAULT 3
#endif

// ---------------------------------------------------------------------------------------------
// error bookkeeping
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int hipFail(hipError_t e, const char *what)
{
   snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
   return BATOTP_ERR_HIP;
}
#define HIP_TRY(call)                                              \
   do                                                              \
   {                                                               \
      hipError_t e_ = (call);                                      \
      if (e_ != hipSuccess) return hipFail(e_, #call);             \
   } while (0)

extern "C" const char *batotp_hip_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// toolchain gate of the flat sweep loop (its torque instantiation was miscompiled by the
// toolchain below for a reason that is not understood; the instantiations that ship passed the whole
// parity / fuzz suite with exactly this compiler, so another compiler gets the nested loops until
// somebody has run the suite with it and updated the string)
// ---------------------------------------------------------------------------------------------
#define BK_STR2(x) #x
#define BK_STR(x) BK_STR2(x)
static const char kBuiltWith[] = "clang " __clang_version__ " / HIP " BK_STR(HIP_VERSION_MAJOR) "." BK_STR(HIP_VERSION_MINOR) "." BK_STR(HIP_VERSION_PATCH);
static const char kFlatValidatedWith[] =
   "clang 22.0.0git (https://github.com/RadeonOpenCompute/llvm-project roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5) / HIP 7.2.26015";

extern "C" int batotp_hip_toolchain(char *built_with, char *validated_with, int32_t cap)
{
   if (cap < 1) return BATOTP_ERR_ARG;
   if (built_with) { strncpy(built_with, kBuiltWith, (size_t)cap - 1); built_with[cap - 1] = 0; }
   if (validated_with) { strncpy(validated_with, kFlatValidatedWith, (size_t)cap - 1); validated_with[cap - 1] = 0; }
   return BATOTP_OK;
}

// ---------------------------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------------------------
// grow-only device workspace cached in the context: hipMalloc of tens of GB costs ~40 ms/GB on this
// system, far more than the kernels that use the memory
struct Arena
{
   void *p = nullptr;
   size_t cap = 0;
};

struct batotp_ctx
{
   Arena ws[4]; // resampler: stage-0 arrays, per-chunk scratch, resampled knots, forward-kinematics scratch (packed joint rows + trig tables)
   std::vector<double> kinTheta, kinTrig; // host side of the trig tables (BATOTP_F_HOST_TRIG)
   Arena xfer;  // staging of curve uploads / downloads (packed double2 <-> separate s / sdot arrays)
   uint64_t rsEpoch = 0; // resample calls so far (a batotp_resampled is valid while its epoch is the current one)
   int device = 0;
   hipStream_t stream = nullptr;
   hipStream_t stream2 = nullptr; // the per-knot evaluation (K3) runs here when overlap is on: nothing in the sweeps depends on it
   hipEvent_t evJoin = nullptr;
   int overlap = 0;
   int sweepGroup = 0; // lanes per path in the sweep kernel; 0 = automatic
   int pathsPerWave = 0; // 0 = automatic
   int sweepTouch[2] = {-1, -1}; // reverse, forward: -1 = automatic, else bit 0 rows, bit 1 reverse curve (kernels.hip.h touch_*)
   int sweepHold[2] = {-2, -2}; // reverse, forward: -2 = automatic, -1 = nested stage / bisection loops, 0..8 = flat loop with this hold (kernels.hip.h)
   // May the AUTOMATIC choice use the flat stage / bisection loop?  0 = not decided yet, 1 = yes (this library was built by the
   // toolchain the loop was validated with and the canary of flatLoopStatus agreed with the nested loops on this device),
   // -1 = built by another toolchain, -2 = the canary disagreed, -3 = the canary could not run
   int splineTiles = -1;  // K1 in tiles of knots (spline_tile.hip.h): -1 automatic (small batches), 1 always, 0 never
   int flatForm = 1;      // flat loop of the 8-lane layout: 1 = k_sweep8 (sweep8.hip.h), 0 = k_sweep's own flat instantiation (A/B, parity)
   int fastForward = 1;   // certified fast-forward of the bisection in the sweep kernels that have it (bisect_fast_forward)
   int certHold = -1;     // k_sweep8, reverse sweep: hold of the certificate phase (-1 automatic, 0 = no certificate there, 1..8)
   int poison = 0;        // debug aid (batotp_hip_set_poison): every workspace / batch allocation is filled with 0xFF bytes before use
   int rsTrace = 0;       // diagnostic (batotp_hip_set_resample_trace): a one-path resample call keeps a checksum of every intermediate stage
   uint64_t rsTraceSums[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // ... of the call in progress (copied into its batotp_resampled)
   unsigned long long *dTrace = nullptr;               // ... device scratch of the checksum kernel (3 x 8 bytes)
   std::vector<double> rsTraceData[8];                 // ... and host copies of the stages a caller may want to look at (2, 3: see rsTraceSum)
   int64_t rsBudget = 0, outBudget = 0; // scratch bytes a chunk of the resampler / output stage may take; 0 = from the free memory
   int pathOrder = 1;     // ragged batches: 1 = the sweeps take the paths longest first (SweepArgs::order), 0 = in the order given
   int k3Form = 1;        // per-knot evaluation of velocity / acceleration-only problems: 1 = k_pointwise_va (pointwise_va.hip.h), 0 = the general kernel
   int flatStatus = 0;
   char builtWith[192] = "";    // toolchain the gate compares (the real one; only a -DBATOTP_TEST_HOOKS build lets a test override it)
};

struct batotp_batch
{
   batotp_ctx *ctx = nullptr;
   batotp_problem prob;
   DevProblem P;
   int32_t B = 0;
   int64_t cap = 0;
   int64_t totalKnots = 0;
   std::vector<PathInfo> pinfo; // host mirror
   bool needPar = false;        // some path may run the parallel-mechanism torque branch
   bool k3Pending = false;      // an overlapped per-knot evaluation may still be running on ctx->stream2
   bool lastForm8[2] = {false, false}; // the most recent sweep per direction ran k_sweep8
   bool compact = false;        // BATOTP_F_COMPACT_SPLINES: (value, second derivative) pairs in dKM, there is no dCoef
   bool pairsAll = false;       // ... of a problem with Cartesian / dynamics channels: ALL C channels are pairs (kmC = C), the kernels of the
                                // coefficient-row layouts (FEAT >= 0) form their rows from them, no sample and no dynamics array exists
   int kmC = 0;                 // channels per knot in dKM (Cin, or C with pairsAll)
   bool kinDone = false, dynDone = false, sitesSet = false, revDone = false, trigSet = false;
   bool inPlace = false;   // BATOTP_F_CURVES_IN_PLACE: dFwd aliases dRev
   bool revGone = false;   // ... and the forward sweep has overwritten the reverse curve
   bool mvcInCurves = false, mvcValid = false; // BATOTP_F_MVC_IN_CURVES: K3's values live in the curve slots until a sweep starts
   bool revStale = false, fwdStale = false;    // ... and a pointwise evaluation has overwritten that curve since its sweep

   // device memory
   DevProblem *dP = nullptr;
   PathInfo *dPinfo = nullptr;
   double *dY = nullptr, *dSC = nullptr, *dCoef = nullptr, *dSamp = nullptr, *dDyn = nullptr, *dTrig = nullptr, *dMvc = nullptr;
   double2 *dRev = nullptr, *dFwd = nullptr;
   batotp_path_result *dRes = nullptr;
   int *dOrder = nullptr;    // ragged batch: path indices sorted by knot count, longest first (nullptr: all paths equally long)
   double *dProf = nullptr;  // diagnostic build (-DS8_PROFILE): 16 doubles per wavefront of k_sweep8
   double *dStage = nullptr; // staging for marshalling (4*maxN doubles)
   int *dSink = nullptr;     // consumer of the sweep kernel's prefetch touches
   double *dElim = nullptr;  // Thomas elimination values of k_spline, [max(Cin,4d)][N] per path
   double *dKM = nullptr;    // compact splines: [N][Cin][2] (knot value, second derivative) per path; replaces dY, dElim and dCoef
   batotp_serial_model *dModel = nullptr; // serial-chain dynamics model (batotp_hip_set_serial_model)
   batotp_serial_model hModel;            // ... its host copy (the output stage needs `degrees` for its trig tables)
   double *dJTrig = nullptr; // [2*nJ][N] per path: host cosines / sines of the joint angles for the serial-chain dynamics
   bool hasSerial = false, jtrigSet = false;
   // K1 in tiles (spline_tile.hip.h): tiles per path (prefix sums), boundary values between tiles, series left to the
   // sequential kernel (short paths from the start; a series whose boundary comparison failed)
   int *dTileOff = nullptr, *dDirty = nullptr;
   double *dEdge = nullptr;
   int totalTiles = 0, nchMax = 0, lastTileNch = 0;
   double *dUp = nullptr;    // compact splines: staging of host knots on their way into dKM
   int64_t upDoubles = 0;
   int64_t maxN = 0;
   int64_t bytes = 0;

   hipEvent_t ev[5][2] = {};
   bool evValid[5] = {};
   int lastLanes[2] = {0, 0}, lastPpw[2] = {0, 0}, lastHold[2] = {-1, -1}; // reverse, forward: what the last launch used
};

// debug aid (batotp_hip_set_poison): memory a stage is about to use is filled with 0xFF bytes -- NaNs as doubles, -1 as integers -- so
// that a kernel that reads what nobody wrote gives a loud, reproducible wrong answer instead of one that depends on what the memory held
static int poisonFill(const batotp_ctx *ctx, void *p, size_t bytes, hipStream_t st)
{
   if (!ctx->poison || !p || !bytes) return BATOTP_OK;
   HIP_TRY(hipMemsetAsync(p, 0xFF, bytes, st));
   return BATOTP_OK;
}

static int devAlloc(batotp_batch *b, void **p, size_t bytes)
{
   if (bytes == 0) bytes = 8;
   hipError_t e = hipMalloc(p, bytes);
   if (e != hipSuccess)
   {
      hipFail(e, "hipMalloc");
      (void)hipGetLastError(); // reported; do not leave the sticky error for a later, unrelated hipGetLastError() check
      return BATOTP_ERR_ALLOC;
   }
   b->bytes += (int64_t)bytes;
   if (b->ctx && b->ctx->poison)
   {
      // on the context's stream, and complete before anything else touches the allocation (a fill on the null stream is not ordered
      // against the context's non-blocking stream: the first poison build "found" upload kernels racing with its own fill)
      e = hipMemsetAsync(*p, 0xFF, bytes, b->ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(b->ctx->stream);
      if (e != hipSuccess) return hipFail(e, "hipMemset (poison)");
   }
   return BATOTP_OK;
}

static int bind(batotp_ctx *ctx)
{
   HIP_TRY(hipSetDevice(ctx->device));
   return BATOTP_OK;
}

// ---------------------------------------------------------------------------------------------
// device context
// ---------------------------------------------------------------------------------------------
extern "C" int batotp_hip_device_count(int *count)
{
   if (!count) return BATOTP_ERR_ARG;
   int n = 0;
   hipError_t e = hipGetDeviceCount(&n);
   if (e != hipSuccess) { *count = 0; hipFail(e, "hipGetDeviceCount"); return BATOTP_ERR_NO_DEVICE; }
   *count = n;
   return BATOTP_OK;
}

static int uploadThomasTable()
{
   // super-diagonal of the eliminated (1,4,1) system, spline.cpp:256-266: c[1] = 1/4,
   // c[i] = 1/(4 - c[i-1]); it reaches a fixed point in double precision after ~25 steps
   double tab[64];
   tab[0] = 0.0;
   double c = 1.0;
   c /= 4.0;
   tab[1] = c;
   int conv = -1;
   for (int i = 2; i < 64; ++i)
   {
      double ci = 1.0;
      ci /= 4.0 - 1.0 * tab[i - 1];
      tab[i] = ci;
      if (conv < 0 && tab[i] == tab[i - 1]) conv = i;
   }
   if (conv < 0 || conv > 60)
   {
      snprintf(g_err, sizeof(g_err), "Thomas coefficient table did not converge");
      return BATOTP_ERR_STATE;
   }
   // verify it really is a fixed point (so that index 63 stands for every later row)
   double chk = 1.0;
   chk /= 4.0 - 1.0 * tab[63];
   if (chk != tab[63])
   {
      snprintf(g_err, sizeof(g_err), "Thomas coefficient table is not a fixed point");
      return BATOTP_ERR_STATE;
   }
   // slot 0 (unused by the recurrence) carries RN(1/(4 - c_inf)) for the reciprocal-based divide of
   // k_spline; its preconditions are checked here: the divisor is not of the form 1.11..1 * 2^k
   {
      const double den = 4.0 - 1.0 * tab[63];
      int e;
      const double m = frexp(den, &e); // in [0.5, 1)
      if (!(m < 1.0 - 1e-9))
      {
         snprintf(g_err, sizeof(g_err), "Thomas pivot unsuitable for the reciprocal divide");
         return BATOTP_ERR_STATE;
      }
      tab[0] = 1.0 / den;
   }
   HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_ctab), tab, sizeof(tab)));
   // the same for Spline::solveTriDiagClamped (spline.cpp:229-237): c[0] = 1/2, c[i] = 1/(4 - c[i-1])
   double tcl[64];
   {
      double c0 = 1.0;
      c0 /= 2.0;
      tcl[0] = c0;
   }
   for (int i = 1; i < 64; ++i)
   {
      double ci = 1.0;
      ci /= 4.0 - 1.0 * tcl[i - 1];
      tcl[i] = ci;
   }
   {
      double chk2 = 1.0;
      chk2 /= 4.0 - 1.0 * tcl[63];
      if (chk2 != tcl[63] || tcl[62] != tcl[63])
      {
         snprintf(g_err, sizeof(g_err), "clamped Thomas coefficient table is not a fixed point");
         return BATOTP_ERR_STATE;
      }
   }
   HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_ctab_cl), tcl, sizeof(tcl)));
   return BATOTP_OK;
}

extern "C" int batotp_hip_ctx_create(int device, batotp_ctx **out)
{
   if (!out) return BATOTP_ERR_ARG;
   *out = nullptr;
   int n = 0;
   hipError_t e = hipGetDeviceCount(&n);
   if (e != hipSuccess || n <= 0)
   {
      snprintf(g_err, sizeof(g_err), "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
      return BATOTP_ERR_NO_DEVICE;
   }
   if (device < 0 || device >= n) return BATOTP_ERR_ARG;
   batotp_ctx *c = new (std::nothrow) batotp_ctx;
   if (!c) return BATOTP_ERR_ALLOC;
   c->device = device;
   strncpy(c->builtWith, kBuiltWith, sizeof(c->builtWith) - 1);
#ifdef BATOTP_TEST_HOOKS
   {
      // TEST BUILD ONLY (csrc/libbatotp_hip_testhooks.so, never the shipped library): a test can pretend the library came from
      // ANOTHER compiler -- the automatic choice must then run the nested loops (tests/test_gpu_parity.py).  The variable can
      // only close the gate: a value equal to the validated string is ignored.
      const char *assume = getenv("BATOTP_ASSUME_TOOLCHAIN");
      if (assume && assume[0] && strcmp(assume, kFlatValidatedWith) != 0)
      {
         memset(c->builtWith, 0, sizeof(c->builtWith));
         strncpy(c->builtWith, assume, sizeof(c->builtWith) - 1);
      }
   }
#endif
   e = hipSetDevice(device);
   if (e != hipSuccess) { delete c; return hipFail(e, "hipSetDevice"); }
   e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
   if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
   if (e == hipSuccess) e = hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming);
   if (e != hipSuccess) { delete c; return hipFail(e, "hipStreamCreate"); }
   int rc = uploadThomasTable();
   if (rc != BATOTP_OK) { hipStreamDestroy(c->stream); delete c; return rc; }
   *out = c;
   return BATOTP_OK;
}

extern "C" int batotp_hip_ctx_destroy(batotp_ctx *ctx)
{
   if (!ctx) return BATOTP_OK;
   hipSetDevice(ctx->device);
   if (ctx->stream) hipStreamDestroy(ctx->stream);
   if (ctx->stream2) hipStreamDestroy(ctx->stream2);
   if (ctx->evJoin) hipEventDestroy(ctx->evJoin);
   for (Arena &a : ctx->ws)
      if (a.p) hipFree(a.p);
   if (ctx->xfer.p) hipFree(ctx->xfer.p);
   if (ctx->dTrace) hipFree(ctx->dTrace);
   delete ctx;
   return BATOTP_OK;
}

extern "C" int batotp_hip_ctx_trim(batotp_ctx *ctx)
{
   if (!ctx) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   for (Arena &a : ctx->ws)
   {
      if (a.p) hipFree(a.p);
      a.p = nullptr; a.cap = 0;
   }
   if (ctx->xfer.p) { hipFree(ctx->xfer.p); ctx->xfer.p = nullptr; ctx->xfer.cap = 0; }
   ++ctx->rsEpoch;
   return BATOTP_OK;
}

extern "C" int batotp_hip_synchronize(batotp_ctx *ctx)
{
   if (!ctx) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   HIP_TRY(hipStreamSynchronize(ctx->stream));
   HIP_TRY(hipStreamSynchronize(ctx->stream2));
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_sweep_group(batotp_ctx *ctx, int32_t lanes)
{
   if (!ctx || !(lanes == 0 || lanes == 1 || lanes == 2 || lanes == 4 || lanes == 8 || lanes == 16 || lanes == 32 || lanes == 64)) return BATOTP_ERR_ARG;
   ctx->sweepGroup = lanes;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_overlap(batotp_ctx *ctx, int32_t on)
{
   if (!ctx) return BATOTP_ERR_ARG;
   ctx->overlap = on ? 1 : 0;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_sweep_hold(batotp_ctx *ctx, int32_t reverse, int32_t forward)
{
   if (!ctx || reverse < -2 || reverse > 8 || forward < -2 || forward > 8) return BATOTP_ERR_ARG;
   ctx->sweepHold[0] = reverse;
   ctx->sweepHold[1] = forward;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_flat_form(batotp_ctx *ctx, int32_t form)
{
   if (!ctx || form < 0 || form > 1) return BATOTP_ERR_ARG;
   ctx->flatForm = form;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_sweep_prefetch(batotp_ctx *ctx, int32_t reverse, int32_t forward)
{
   if (!ctx || reverse < -1 || reverse > 3 || forward < -1 || forward > 3) return BATOTP_ERR_ARG;
   ctx->sweepTouch[0] = reverse;
   ctx->sweepTouch[1] = forward;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_spline_tiles(batotp_ctx *ctx, int32_t on)
{
   if (!ctx) return BATOTP_ERR_ARG;
   ctx->splineTiles = on < 0 ? -1 : (on > 2 ? 2 : on);   // (2: the single-pass kernel for pairs whatever the batch size)
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_fast_forward(batotp_ctx *ctx, int32_t on)
{
   if (!ctx) return BATOTP_ERR_ARG;
   ctx->fastForward = on != 0 ? 1 : 0;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_resample_trace(batotp_ctx *ctx, int32_t on)
{
   if (!ctx) return BATOTP_ERR_ARG;
   ctx->rsTrace = on ? 1 : 0;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_poison(batotp_ctx *ctx, int32_t on)
{
   if (!ctx) return BATOTP_ERR_ARG;
   ctx->poison = on ? 1 : 0;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_cert_hold(batotp_ctx *ctx, int32_t hold)
{
   if (!ctx || hold < -1 || hold > 8) return BATOTP_ERR_ARG;
   ctx->certHold = hold;
   return BATOTP_OK;
}
// the reverse sweep of k_sweep8 gathers the paths whose first check failed and certifies them in batches (sweep8.hip.h): automatic hold
static int certHoldOf(const batotp_ctx *ctx) { return ctx->certHold < 0 ? BATOTP_CERT_HOLD_DEFAULT : ctx->certHold; }
static int sweep8FastForward(const batotp_ctx *ctx) { return ctx->fastForward ? (1 | (certHoldOf(ctx) > 0 ? 2 : 0)) : 0; }

#ifdef S8_PROFILE
// diagnostic build only: the counters k_sweep8 left (16 doubles per wavefront, n_waves <= B)
extern "C" int batotp_hip_debug_sweep8_counters(batotp_batch *b, double *out, int64_t n_doubles)
{
   if (!b || !out || n_doubles > (int64_t)b->B * 16 + 16) return BATOTP_ERR_ARG;
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   HIP_TRY(hipMemcpy(out, b->dProf, sizeof(double) * (size_t)n_doubles, hipMemcpyDeviceToHost));
   return BATOTP_OK;
}
#endif

extern "C" int batotp_hip_set_path_order(batotp_ctx *ctx, int32_t mode)
{
   if (!ctx || (mode != 0 && mode != 1)) return BATOTP_ERR_ARG;
   ctx->pathOrder = mode;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_workspace_budget(batotp_ctx *ctx, int64_t resample_bytes, int64_t output_bytes)
{
   if (!ctx || resample_bytes < 0 || output_bytes < 0) return BATOTP_ERR_ARG;
   ctx->rsBudget = resample_bytes;
   ctx->outBudget = output_bytes;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_k3_form(batotp_ctx *ctx, int32_t form)
{
   if (!ctx || (form != 0 && form != 1)) return BATOTP_ERR_ARG;
   ctx->k3Form = form;
   return BATOTP_OK;
}

extern "C" int batotp_hip_spline_tile_fallbacks(batotp_batch *b, int32_t *series)
{
   if (!b || !series) return BATOTP_ERR_ARG;
   *series = 0;
   if (!b->totalTiles || b->lastTileNch == 0) return BATOTP_OK;
   int rc = bind(b->ctx);
   if (rc) return rc;
   std::vector<int> d((size_t)b->B * b->lastTileNch);
   HIP_TRY(hipMemcpy(d.data(), b->dDirty, sizeof(int) * d.size(), hipMemcpyDeviceToHost));
   int cnt = 0;
   for (int p = 0; p < b->B; ++p)
      if (b->pinfo[p].n >= ST_MIN_KNOTS)
         for (int c = 0; c < b->lastTileNch; ++c) cnt += d[(size_t)p * b->lastTileNch + c] ? 1 : 0;
   *series = cnt;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_paths_per_wave(batotp_ctx *ctx, int32_t n)
{
   if (!ctx || n < 0 || n > 64) return BATOTP_ERR_ARG;
   ctx->pathsPerWave = n;
   return BATOTP_OK;
}

extern "C" int batotp_hip_fp64_kat(batotp_ctx *ctx, int64_t n, const double *a, const double *b, double *q, double *r, double *p)
{
   if (!ctx || n <= 0 || !a || !b || !q || !r || !p) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   double *d = nullptr;
   const size_t sz = sizeof(double) * (size_t)n;
   HIP_TRY(hipMalloc((void **)&d, 5 * sz));
   hipMemcpyAsync(d, a, sz, hipMemcpyHostToDevice, ctx->stream);
   hipMemcpyAsync(d + n, b, sz, hipMemcpyHostToDevice, ctx->stream);
   const int bs = 256;
   hipLaunchKernelGGL(k_kat, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, ctx->stream, n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n);
   hipMemcpyAsync(q, d + 2 * n, sz, hipMemcpyDeviceToHost, ctx->stream);
   hipMemcpyAsync(r, d + 3 * n, sz, hipMemcpyDeviceToHost, ctx->stream);
   hipMemcpyAsync(p, d + 4 * n, sz, hipMemcpyDeviceToHost, ctx->stream);
   hipError_t e = hipStreamSynchronize(ctx->stream);
   hipFree(d);
   if (e != hipSuccess) return hipFail(e, "fp64_kat");
   return BATOTP_OK;
}

extern "C" int batotp_hip_div6_kat(batotp_ctx *ctx, int64_t n, const double *a, double *q)
{
   if (!ctx || n <= 0 || !a || !q) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   double *d = nullptr;
   const size_t sz = sizeof(double) * (size_t)n;
   HIP_TRY(hipMalloc((void **)&d, 2 * sz));
   hipMemcpyAsync(d, a, sz, hipMemcpyHostToDevice, ctx->stream);
   hipLaunchKernelGGL(k_kat_div6, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d, d + n);
   hipMemcpyAsync(q, d + n, sz, hipMemcpyDeviceToHost, ctx->stream);
   hipError_t e = hipStreamSynchronize(ctx->stream);
   hipFree(d);
   if (e != hipSuccess) return hipFail(e, "div6_kat");
   return BATOTP_OK;
}

extern "C" int batotp_hip_spline_lanes_kat(batotp_ctx *ctx, int64_t n, const double *y, double *sol, double *sol_seq, int32_t *redone)
{
   if (!ctx || n < 4 || n > (int64_t)1 << 30 || !y || !sol || !sol_seq || !redone) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   double *d = nullptr;
   const size_t sz = sizeof(double) * (size_t)n;
   HIP_TRY(hipMalloc((void **)&d, 3 * sz + 64));
   double *dy = d, *dA = d + n, *dB = d + 2 * n;
   int64_t *dOff = reinterpret_cast<int64_t *>(d + 3 * n); // one 0: both offsets
   int *dCnt = reinterpret_cast<int *>(dOff + 1), *dFlag = dCnt + 1;
   const int64_t zero = 0;
   const int cnt = (int)n;
   hipStream_t st = ctx->stream;
   // every call and launch is checked where it is made; the device block is released on every way out
   struct Guard { double *p; ~Guard() { if (p) (void)hipFree(p); } } guard{d};
   HIP_TRY(hipMemcpyAsync(dy, y, sz, hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemcpyAsync(dOff, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemcpyAsync(dCnt, &cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemsetAsync(dA, 0xff, sz, st)); // what the kernels do not write shows up as NaN
   HIP_TRY(hipMemsetAsync(dB, 0xff, sz, st));
   HIP_TRY(hipMemsetAsync(dFlag, 0, sizeof(int), st));
   hipLaunchKernelGGL(k_spline_series_lanes, dim3(1), dim3(64), 0, st, 1, dOff, dOff, dCnt, dy, 1, dA, dFlag);
   HIP_TRY(hipGetLastError());
   hipLaunchKernelGGL(k_spline_series, dim3(1), dim3(seriesBlock(1)), 0, st, 1, dOff, dOff, dCnt, dy, 1, dA, dFlag);
   HIP_TRY(hipGetLastError());
   hipLaunchKernelGGL(k_spline_series, dim3(1), dim3(seriesBlock(1)), 0, st, 1, dOff, dOff, dCnt, dy, 1, dB, (const int *)nullptr);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(sol, dA, sz, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipMemcpyAsync(sol_seq, dB, sz, hipMemcpyDeviceToHost, st));
   int flag = 0;
   HIP_TRY(hipMemcpyAsync(&flag, dFlag, sizeof(int), hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));
   *redone = flag;
   return BATOTP_OK;
}

extern "C" int batotp_hip_sdiv_kat(batotp_ctx *ctx, int64_t n, const double *a, const double *b, double *q, int32_t *in_window)
{
   if (!ctx || n <= 0 || !a || !b || !q || !in_window) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   double *d = nullptr;
   const size_t sz = sizeof(double) * (size_t)n;
   HIP_TRY(hipMalloc((void **)&d, 4 * sz));
   hipMemcpyAsync(d, a, sz, hipMemcpyHostToDevice, ctx->stream);
   hipMemcpyAsync(d + n, b, sz, hipMemcpyHostToDevice, ctx->stream);
   hipLaunchKernelGGL(k_kat_sdiv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d, d + n, d + 2 * n, reinterpret_cast<int *>(d + 3 * n));
   hipMemcpyAsync(q, d + 2 * n, sz, hipMemcpyDeviceToHost, ctx->stream);
   hipMemcpyAsync(in_window, d + 3 * n, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
   hipError_t e = hipStreamSynchronize(ctx->stream);
   hipFree(d);
   if (e != hipSuccess) return hipFail(e, "sdiv_kat");
   return BATOTP_OK;
}

// grow-only staging buffer of the context (the caller has bound the device)
static int xferReserve(batotp_ctx *ctx, size_t bytes, double **out)
{
   if (ctx->xfer.cap < bytes)
   {
      if (ctx->xfer.p) { hipFree(ctx->xfer.p); ctx->xfer.p = nullptr; ctx->xfer.cap = 0; }
      const size_t want = std::max(bytes, (size_t)1 << 20);
      hipError_t e = hipMalloc(&ctx->xfer.p, want);
      if (e != hipSuccess) { hipFail(e, "hipMalloc(staging)"); (void)hipGetLastError(); return BATOTP_ERR_ALLOC; }
      ctx->xfer.cap = want;
   }
   *out = static_cast<double *>(ctx->xfer.p);
   return BATOTP_OK;
}

// ---------------------------------------------------------------------------------------------
// batch lifetime
// ---------------------------------------------------------------------------------------------
static int dynDim(const batotp_problem *prob)
{
   if (!(prob->flags & BATOTP_F_TRQ_ON)) return 0;
   return (prob->flags & BATOTP_F_PARALLEL) ? prob->n_cart : prob->n_joints; // ba.cpp:876-888
}

extern "C" int batotp_hip_batch_destroy(batotp_batch *b)
{
   if (!b) return BATOTP_OK;
   if (b->ctx) hipSetDevice(b->ctx->device);
   if (b->ctx && b->k3Pending) hipStreamSynchronize(b->ctx->stream2);
   if (b->dFwd == b->dRev) b->dFwd = nullptr; // BATOTP_F_CURVES_IN_PLACE: one buffer
   void *ptrs[] = {b->dP, b->dPinfo, b->dY, b->dSC, b->dCoef, b->dSamp, b->dDyn, b->dTrig, b->dMvc, b->dRev, b->dFwd, b->dRes, b->dStage, b->dSink, b->dElim, b->dKM, b->dUp, b->dModel, b->dJTrig, b->dTileOff, b->dDirty, b->dEdge, b->dProf, b->dOrder};
   for (void *p : ptrs)
      if (p) hipFree(p);
   for (int k = 0; k < 5; ++k)
      for (int s = 0; s < 2; ++s)
         if (b->ev[k][s]) hipEventDestroy(b->ev[k][s]);
   delete b;
   return BATOTP_OK;
}

extern "C" int batotp_hip_batch_create(batotp_ctx *ctx, const batotp_problem *prob, int32_t n_paths, const int64_t *n_knots,
                                       int64_t max_steps, batotp_batch **out)
{
   if (!ctx || !prob || !n_knots || !out || n_paths < 1 || max_steps < 4) return BATOTP_ERR_ARG;
   *out = nullptr;
   if (prob->n_joints < 1 || prob->n_joints > BATOTP_MAX_JOINTS || prob->n_cart < 0 || prob->n_cart > BATOTP_MAX_CART) return BATOTP_ERR_ARG;
   const int d = dynDim(prob);
   if (d != 0 && d != prob->n_joints) return BATOTP_ERR_ARG; // ba.cpp:940-946 iterates dynamics rows over nJoints
   if ((prob->flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)) && prob->n_cart < 3) return BATOTP_ERR_ARG;
   if ((prob->flags & BATOTP_F_TRQ_ON) && (prob->flags & BATOTP_F_PARALLEL) && (prob->n_joints != 3 || prob->n_cart < 3)) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;

   batotp_batch *b = new (std::nothrow) batotp_batch;
   if (!b) return BATOTP_ERR_ALLOC;
   b->ctx = ctx;
   b->prob = *prob;
   b->B = n_paths;
   b->cap = max_steps;
   DevProblem &P = b->P;
   memset(&P, 0, sizeof(P));
   P.nJ = prob->n_joints; P.nC = prob->n_cart; P.d = d; P.robot = prob->robot_type;
   P.flags = prob->flags;
   P.Cin = P.nJ + P.nC;
   P.C = P.Cin + 4 * d;
   for (int k = 0; k < 8; ++k)
   {
      P.vmax[k] = prob->jnt_vel_max[k]; P.amax[k] = prob->jnt_acc_max[k];
      P.tmax[k] = prob->jnt_trq_max[k]; P.tmin[k] = prob->jnt_trq_min[k];
   }
   P.cart_vel_max = prob->cart_vel_max; P.cart_acc_max = prob->cart_acc_max;
   P.jnt_thresh = prob->jnt_thresh; P.quad_thresh = prob->quad_rad_thresh;
   P.integ_res = prob->integ_res; P.max_integ_time = prob->max_integ_time;
   for (int k = 0; k < 9; ++k) P.pmat[k] = prob->pmat[k];
   b->needPar = (prob->flags & BATOTP_F_TRQ_ON) && (prob->flags & BATOTP_F_PARALLEL) && !(prob->flags & BATOTP_F_PAR2SER);

   b->pinfo.resize(n_paths);
   int64_t off = 0;
   for (int p = 0; p < n_paths; ++p)
   {
      if (n_knots[p] < 2) { delete b; return BATOTP_ERR_ARG; }
      PathInfo &pi = b->pinfo[p];
      memset(&pi, 0, sizeof(pi));
      pi.koff = off;
      pi.n = n_knots[p];
      pi.parallel_now = (prob->flags & BATOTP_F_PARALLEL) ? 1 : 0;
      pi.integ_res = prob->integ_res;
      off += n_knots[p];
      if (n_knots[p] > b->maxN) b->maxN = n_knots[p];
   }
   b->totalKnots = off;
   if (off > ((int64_t)1 << 32) - 4096) // kernels with one lane per knot: HIP limits grid x block to 32 bits
   {
      snprintf(g_err, sizeof(g_err), "batch of %lld knots exceeds the 2^32 knots one batch may hold", (long long)off);
      delete b;
      return BATOTP_ERR_ARG;
   }

   b->compact = (prob->flags & BATOTP_F_COMPACT_SPLINES) != 0;
   if (b->compact && (!(prob->flags & BATOTP_F_NO_SAMPLES) || b->needPar))
   {
      snprintf(g_err, sizeof(g_err), "compact splines need BATOTP_F_NO_SAMPLES and constraints in serial form (no parallel-mechanism torque branch)");
      batotp_hip_batch_destroy(b);
      return BATOTP_ERR_ARG;
   }
   b->pairsAll = b->compact && (d != 0 || (prob->flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)));
   b->kmC = b->compact ? (b->pairsAll ? P.C : P.Cin) : 0;

#define ALLOC(ptr, count, type)                                                        \
   rc = devAlloc(b, (void **)&(ptr), sizeof(type) * (size_t)(count));                   \
   if (rc) { batotp_hip_batch_destroy(b); return rc; }
   ALLOC(b->dP, 1, DevProblem)
   ALLOC(b->dPinfo, n_paths, PathInfo)
   ALLOC(b->dY, b->compact ? 0 : off * P.Cin, double)
   // knot sites: kept for the row layouts (K2 reads them); compact batches compute uniform sites where they are needed and get
   // the array only when a path's sites are uploaded (batotp_hip_upload_path_sites)
   if (!b->compact) { ALLOC(b->dSC, off, double) }   // (devAlloc turns a zero-size request into 8 bytes: the pointer must stay null here)
   ALLOC(b->dCoef, b->compact ? 0 : off * P.C * 4, double)
   if ((prob->flags & BATOTP_F_NO_SAMPLES) && d != 0 && !b->pairsAll) { batotp_hip_batch_destroy(b); return BATOTP_ERR_ARG; } // K2 reads the samples
   ALLOC(b->dSamp, (prob->flags & BATOTP_F_NO_SAMPLES) ? 0 : off * P.Cin * 3, double)
   ALLOC(b->dDyn, b->pairsAll ? 0 : off * 4 * (d ? d : 0), double)
   ALLOC(b->dTrig, (prob->robot_type == BATOTP_ROBOT_RR && d) ? off * 4 : 0, double)
   b->mvcInCurves = (prob->flags & BATOTP_F_MVC_IN_CURVES) != 0;
   if (b->mvcInCurves && 2 * max_steps < 3 * b->maxN)
   {
      snprintf(g_err, sizeof(g_err), "BATOTP_F_MVC_IN_CURVES needs max_steps >= 1.5 x the knots of the longest path");
      batotp_hip_batch_destroy(b);
      return BATOTP_ERR_ARG;
   }
   if (!b->mvcInCurves) { ALLOC(b->dMvc, off * 3, double) }
   {
      // ragged batch: the launch order of the sweeps (SweepArgs::order)
      bool ragged = false;
      for (int p = 1; p < n_paths; ++p) ragged |= n_knots[p] != n_knots[0];
      if (ragged)
      {
         std::vector<int> order((size_t)n_paths);
         for (int p = 0; p < n_paths; ++p) order[(size_t)p] = p;
         std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return n_knots[x] > n_knots[y]; });
         ALLOC(b->dOrder, n_paths, int)
         if (hipMemcpy(b->dOrder, order.data(), sizeof(int) * (size_t)n_paths, hipMemcpyHostToDevice) != hipSuccess)
         {
            batotp_hip_batch_destroy(b);
            return BATOTP_ERR_HIP;
         }
      }
   }
#ifdef S8_PROFILE
   ALLOC(b->dProf, (int64_t)n_paths * 16 + 16, double)
#endif
   ALLOC(b->dRev, (int64_t)n_paths * max_steps, double2)
   b->inPlace = (prob->flags & BATOTP_F_CURVES_IN_PLACE) != 0;
   if (b->inPlace) b->dFwd = b->dRev; // one curve buffer: the forward sweep overwrites the reverse points behind its cursor
   else { ALLOC(b->dFwd, (int64_t)n_paths * max_steps, double2) }
   ALLOC(b->dRes, n_paths, batotp_path_result)
   ALLOC(b->dStage, 4 * b->maxN, double)
   ALLOC(b->dSink, n_paths, int)
   ALLOC(b->dElim, b->compact ? 0 : off * (P.Cin > 4 * d ? P.Cin : 4 * d), double)
   ALLOC(b->dKM, b->compact ? off * b->kmC * 2 : 0, double)
   std::vector<int> tileOff((size_t)n_paths + 1, 0);
   {
      b->nchMax = std::max(P.Cin, 4 * d);
      for (int p = 0; p < n_paths; ++p)
         tileOff[p + 1] = tileOff[p] + (n_knots[p] >= ST_MIN_KNOTS ? (int)((n_knots[p] - 1 + ST_T - 1) / ST_T) : 0);   // the last knot joins the last tile
      b->totalTiles = tileOff[n_paths];
   }
   ALLOC(b->dTileOff, n_paths + 1, int)
   ALLOC(b->dDirty, (size_t)n_paths * b->nchMax, int)
   ALLOC(b->dEdge, (size_t)b->totalTiles * b->nchMax * 4, double)
#undef ALLOC
   if (hipMemcpy(b->dTileOff, tileOff.data(), sizeof(int) * tileOff.size(), hipMemcpyHostToDevice) != hipSuccess)
   {
      batotp_hip_batch_destroy(b);
      return hipFail(hipGetLastError(), "batch_create (tile table)");
   }
   hipError_t e = hipMemcpyAsync(b->dP, &P, sizeof(P), hipMemcpyHostToDevice, ctx->stream);
   if (e == hipSuccess) e = hipMemcpyAsync(b->dPinfo, b->pinfo.data(), sizeof(PathInfo) * n_paths, hipMemcpyHostToDevice, ctx->stream);
   if (e == hipSuccess) e = hipMemsetAsync(b->dRes, 0, sizeof(batotp_path_result) * n_paths, ctx->stream);
   if (e == hipSuccess && !b->compact) e = hipMemsetAsync(b->dCoef, 0, sizeof(double) * (size_t)(off * P.C * 4), ctx->stream);
   for (int k = 0; k < 5 && e == hipSuccess; ++k)
      for (int s = 0; s < 2 && e == hipSuccess; ++s) e = hipEventCreate(&b->ev[k][s]);
   if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
   if (e != hipSuccess) { batotp_hip_batch_destroy(b); return hipFail(e, "batch_create"); }
   *out = b;
   return BATOTP_OK;
}

static int pushPinfo(batotp_batch *b)
{
   HIP_TRY(hipMemcpyAsync(b->dPinfo, b->pinfo.data(), sizeof(PathInfo) * b->B, hipMemcpyHostToDevice, b->ctx->stream));
   // the host vector may be modified again right after this call
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

static void setSres(batotp_batch *b, int p, double sres)
{
   PathInfo &pi = b->pinfo[p];
   const int64_t N = pi.n;
   pi.sres_c = sres;                                        // ba.cpp:815
   pi.vfact = 1 / pi.sres_c;                                // ba.cpp:816
   pi.afact = pi.vfact * pi.vfact;                          // ba.cpp:817
   pi.sres = sres * (double)(N - 1) / (double)(N - 1);      // ba.cpp:798,818 with nPtsNew == nPtsOld
   pi.uniform = 1;
}

static int uploadKnotsImpl(batotp_batch *b, int32_t path0, int32_t n, const double *y, const double *sres, hipMemcpyKind kind, int srcRows = 0)
{
   if (!b || !y || !sres || path0 < 0 || n < 1 || path0 + n > b->B) return BATOTP_ERR_ARG;
   if (srcRows == 0) srcRows = b->P.Cin;
   if (srcRows < b->P.Cin || (srcRows != b->P.Cin && kind != hipMemcpyDeviceToDevice)) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   const int64_t first = b->pinfo[path0].koff;
   const int64_t last = b->pinfo[path0 + n - 1].koff + b->pinfo[path0 + n - 1].n;
   for (int k = 0; k < n; ++k)
   {
      if (b->pinfo[path0 + k].n < 4) return BATOTP_ERR_ARG;
      setSres(b, path0 + k, sres[k]);
   }
   hipStream_t st = b->ctx->stream;
   const int Cin = b->P.Cin, bs = 256;
   if (!b->compact && srcRows == Cin)
      HIP_TRY(hipMemcpyAsync(b->dY + first * Cin, y, sizeof(double) * (size_t)((last - first) * Cin), kind, st));
   else if (!b->compact)
   {
      // the caller's paths carry more rows than the batch keeps (the resampler's Cartesian rows of a problem without Cartesian limits)
      const int64_t total = last - first;
      hipLaunchKernelGGL(k_rows_take, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st, b->dPinfo, path0, n, Cin, srcRows, y, b->dY, total);
      HIP_TRY(hipGetLastError());
   }
   else if (kind == hipMemcpyDeviceToDevice)
   {
      // compact splines: the values go straight from the caller's device buffer into the pair array
      const int64_t total = last - first;
      hipLaunchKernelGGL(k_pairs_from_rows, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st, b->dPinfo, path0, n, Cin, srcRows, b->kmC, y, b->dKM, total);
      HIP_TRY(hipGetLastError());
   }
   else
   {
      // host knots: through a bounded staging buffer, a run of whole paths at a time
      const int64_t want = std::min<int64_t>((last - first) * Cin, std::max<int64_t>((int64_t)1 << 25, b->maxN * Cin));
      if (b->upDoubles < want)
      {
         if (b->dUp) { hipFree(b->dUp); b->dUp = nullptr; b->upDoubles = 0; }
         rc = devAlloc(b, (void **)&b->dUp, sizeof(double) * (size_t)want);
         if (rc) return rc;
         b->upDoubles = want;
      }
      int p = path0;
      while (p < path0 + n)
      {
         int q = p;
         int64_t knots = 0;
         while (q < path0 + n && (knots + b->pinfo[q].n) * Cin <= b->upDoubles) { knots += b->pinfo[q].n; ++q; }
         if (q == p) return BATOTP_ERR_STATE;
         const int64_t off = b->pinfo[p].koff - first;
         HIP_TRY(hipMemcpyAsync(b->dUp, y + off * Cin, sizeof(double) * (size_t)(knots * Cin), hipMemcpyHostToDevice, st));
         hipLaunchKernelGGL(k_pairs_from_rows, dim3((unsigned)((knots + bs - 1) / bs)), dim3(bs), 0, st, b->dPinfo, p, q - p, Cin, Cin, b->kmC, b->dUp, b->dKM, knots);
         HIP_TRY(hipGetLastError());
         HIP_TRY(hipStreamSynchronize(st)); // the staging buffer is reused by the next run
         p = q;
      }
   }
   rc = pushPinfo(b);
   b->kinDone = false; b->dynDone = false;
   return rc;
}

extern "C" int batotp_hip_upload_knots(batotp_batch *b, int32_t path0, int32_t n, const double *y, const double *sres)
{
   return uploadKnotsImpl(b, path0, n, y, sres, hipMemcpyHostToDevice);
}
extern "C" int batotp_hip_upload_knots_device(batotp_batch *b, int32_t path0, int32_t n, const double *y_dev, const double *sres)
{
   return uploadKnotsImpl(b, path0, n, y_dev, sres, hipMemcpyDeviceToDevice);
}
extern "C" int batotp_hip_upload_knots_device_rows(batotp_batch *b, int32_t path0, int32_t n, const double *y_dev, int32_t src_rows, const double *sres)
{
   if (src_rows < 1) return BATOTP_ERR_ARG;
   return uploadKnotsImpl(b, path0, n, y_dev, sres, hipMemcpyDeviceToDevice, src_rows);
}

extern "C" int batotp_hip_upload_rr_trig(batotp_batch *b, int32_t path, const double *trig)
{
   if (!b || !trig || path < 0 || path >= b->B || !b->dTrig || b->P.d == 0) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   const PathInfo &pi = b->pinfo[path];
   HIP_TRY(hipMemcpyAsync(b->dTrig + pi.koff * 4, trig, sizeof(double) * 4 * (size_t)pi.n, hipMemcpyHostToDevice, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   b->trigSet = true;
   return BATOTP_OK;
}

extern "C" int batotp_hip_builtin_serial_model(int32_t robot_type, batotp_serial_model *out)
{
   if (!out) return BATOTP_ERR_ARG;
   return batotp_builtin_serial_model(robot_type, out) == 0 ? BATOTP_OK : BATOTP_ERR_ARG;
}

extern "C" int batotp_hip_set_serial_model(batotp_batch *b, const batotp_serial_model *model)
{
   if (!b || !model) return BATOTP_ERR_ARG;
   if (model->n_links != b->P.nJ || model->n_links < 1 || model->n_links > BATOTP_MAX_LINKS) return BATOTP_ERR_ARG;
   if (!(b->prob.flags & BATOTP_F_TRQ_ON) || (b->prob.flags & BATOTP_F_PARALLEL))
   {
      snprintf(g_err, sizeof(g_err), "a serial-chain model needs BATOTP_F_TRQ_ON on a serial robot");
      return BATOTP_ERR_ARG;
   }
   for (int i = 0; i < model->n_links; ++i)
   {
      // unit axis: the rotation of rot_axis is a rotation only then
      const double *a = model->link[i].axis;
      const double nn = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
      if (!(fabs(nn - 1.0) < 1e-9)) { snprintf(g_err, sizeof(g_err), "joint axis %d is not a unit vector", i); return BATOTP_ERR_ARG; }
   }
   int rc = bind(b->ctx);
   if (rc) return rc;
   if (!b->dModel)
   {
      rc = devAlloc(b, (void **)&b->dModel, sizeof(batotp_serial_model));
      if (rc) return rc;
   }
   HIP_TRY(hipMemcpyAsync(b->dModel, model, sizeof(*model), hipMemcpyHostToDevice, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   b->hModel = *model;
   b->hasSerial = true;
   b->dynDone = false;
   return BATOTP_OK;
}

extern "C" int batotp_hip_upload_joint_trig(batotp_batch *b, int32_t path, const double *trig)
{
   if (!b || !trig || path < 0 || path >= b->B || !b->hasSerial) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   if (!b->dJTrig)
   {
      rc = devAlloc(b, (void **)&b->dJTrig, sizeof(double) * (size_t)(b->totalKnots * 2 * b->P.nJ));
      if (rc) return rc;
   }
   const PathInfo &pi = b->pinfo[path];
   HIP_TRY(hipMemcpyAsync(b->dJTrig + pi.koff * 2 * b->P.nJ, trig, sizeof(double) * 2 * (size_t)b->P.nJ * (size_t)pi.n, hipMemcpyHostToDevice,
                          b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   b->jtrigSet = true;
   return BATOTP_OK;
}

// ABI channel -> device channel (device order interleaves a1..a4 per dynamics row)
static int devChannel(const batotp_batch *b, int ch)
{
   const int Cin = b->P.Cin, d = b->P.d;
   if (ch < 0 || ch >= b->P.C) return -1;
   if (ch < Cin) return ch;
   const int k = (ch - Cin) / d, r = (ch - Cin) % d;
   return Cin + r * 4 + k;
}

extern "C" int batotp_hip_upload_path_sites(batotp_batch *b, int32_t path, const double *sites, double vfact, double afact, int32_t parallel_now)
{
   if (!b || !sites || path < 0 || path >= b->B) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   PathInfo &pi = b->pinfo[path];
   if (!b->dSC)
   {
      // compact batch: the site array exists from the first uploaded path on; the other paths keep their computed sites
      rc = devAlloc(b, (void **)&b->dSC, sizeof(double) * (size_t)b->totalKnots);
      if (rc) return rc;
      const int bs = 256;
      hipLaunchKernelGGL(k_sites, dim3((unsigned)((b->totalKnots + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream, b->dPinfo, b->B, b->dSC, b->totalKnots);
   }
   HIP_TRY(hipMemcpyAsync(b->dSC + pi.koff, sites, sizeof(double) * (size_t)pi.n, hipMemcpyHostToDevice, b->ctx->stream));
   pi.vfact = vfact; pi.afact = afact;
   pi.parallel_now = parallel_now;
   pi.sres_c = sites[1];
   pi.sres = pi.sres_c;
   pi.uniform = 1; // sites[k] == sites[1]*k bit for bit (what ba.cpp:800-806 produces)?
   for (int64_t k = 0; k < pi.n; ++k)
      if (sites[k] != pi.sres_c * (double)k) { pi.uniform = 0; break; }
   if (parallel_now && (b->prob.flags & BATOTP_F_TRQ_ON))
   {
      if (b->P.nJ != 3 || b->P.nC < 3) return BATOTP_ERR_ARG;
      b->needPar = true;
   }
   rc = pushPinfo(b);
   b->sitesSet = true;
   return rc;
}

extern "C" int batotp_hip_upload_coeffs(batotp_batch *b, int32_t path, int32_t channel, const double *c)
{
   if (!b || !c || path < 0 || path >= b->B) return BATOTP_ERR_ARG;
   const int dc = devChannel(b, channel);
   if (dc < 0) return BATOTP_ERR_ARG;
   if (b->compact) { snprintf(g_err, sizeof(g_err), "a batch with compact splines holds no coefficient rows to overwrite"); return BATOTP_ERR_STATE; }
   int rc = bind(b->ctx);
   if (rc) return rc;
   const PathInfo &pi = b->pinfo[path];
   HIP_TRY(hipMemcpyAsync(b->dStage, c, sizeof(double) * 4 * (size_t)pi.n, hipMemcpyHostToDevice, b->ctx->stream));
   const int bs = 256;
   hipLaunchKernelGGL(k_coef_scatter, dim3((unsigned)((pi.n + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream,
                      b->dCoef + pi.koff * b->P.C * 4, b->P.C, dc, pi.n, b->dStage);
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

extern "C" int batotp_hip_upload_curve(batotp_batch *b, int32_t path, const double *s, const double *sdot, int64_t n)
{
   if (!b || !s || !sdot || path < 0 || path >= b->B || n < 2 || n > b->cap) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   double *tmp = nullptr;
   rc = xferReserve(b->ctx, sizeof(double) * 2 * (size_t)n, &tmp);
   if (rc) return rc;
   hipMemcpyAsync(tmp, s, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, b->ctx->stream);
   hipMemcpyAsync(tmp + n, sdot, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, b->ctx->stream);
   const int bs = 256;
   hipLaunchKernelGGL(k_curve_pack, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream,
                      b->dRev + (int64_t)path * b->cap + (b->cap - n), tmp, tmp + n, n);
   // n_rev of the result row tells the forward kernel where the curve starts
   batotp_path_result r;
   hipMemcpyAsync(&r, b->dRes + path, sizeof(r), hipMemcpyDeviceToHost, b->ctx->stream);
   hipError_t e = hipStreamSynchronize(b->ctx->stream);
   if (e == hipSuccess)
   {
      r.n_rev = n;
      e = hipMemcpy(b->dRes + path, &r, sizeof(r), hipMemcpyHostToDevice);
   }
   if (e != hipSuccess) return hipFail(e, "upload_curve");
   b->revDone = true;
   b->revGone = false;
   b->revStale = false;
   b->mvcValid = false;
   return BATOTP_OK;
}

// ---------------------------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------------------------
static int joinK3(batotp_batch *b)
{
   if (b->k3Pending)
   {
      HIP_TRY(hipStreamSynchronize(b->ctx->stream2));
      b->k3Pending = false;
   }
   return BATOTP_OK;
}

static void evStart(batotp_batch *b, int which) { hipEventRecord(b->ev[which][0], b->ctx->stream); }
static void evStop(batotp_batch *b, int which) { hipEventRecord(b->ev[which][1], b->ctx->stream); b->evValid[which] = true; }

// K1 of nch series per path: the tiled kernel (spline_tile.hip.h) for every path of at least ST_MIN_KNOTS knots, then the
// sequential kernel for what is left (short paths; a series whose boundary comparison failed -- never observed).
// pairs: the compact layout (in place in dKM); else channel-major rows `src` in, coefficient rows out.
// pairs: series c0 .. c0 + nch - 1 of the kmC channels a knot holds.
static int launchSpline(batotp_batch *b, int nch, int mode, const double *src, int64_t srcStridePerKnot, bool pairs = false, int c0 = 0)
{
   hipStream_t st = b->ctx->stream;
   const int threads = b->B * nch;
   const int bs = 64;
   // Automatic choice (splineTiles = -1): tiles while the batch is too small to fill the GPU with one lane per series -- there
   // the sequential kernel is a dependent chain of N steps whatever the batch (9.5 ms for ONE 6-joint path of 1e5 knots, 54 ms
   // with the 28 dynamics channels of a 7-joint arm; tiles: 0.05 and 0.22 ms) -- and the lane-per-series kernel beyond:
   // a chunk of 16 knots costs a tile 80 forward and 80 backward steps (the warm-ups), and the tiles of a CU are limited by
   // LDS, so at 2048 paths x 7 series of 1e5 knots the tiles take 34 ms against 22 ms (16 384 paths: 285 against 156 ms).
   // Coefficient rows switch later than pairs: the lane-per-series kernel writes them as 32-byte pieces with a stride of a whole
   // row, the tiles store them coalesced from LDS (1024 cable-robot paths of 2e5 knots, 6 + 12 series per path: 785 against
   // 978 ms of precompute per 4096 paths; 1024 GEN7DOF paths x 7 series as pairs: 12.3 against 7.4 ms).
   const int64_t tileLimit = pairs ? 4096 : 16384;
   const bool tiled = b->totalTiles > 0 && (b->ctx->splineTiles == 1 || (b->ctx->splineTiles < 0 && (int64_t)b->B * nch <= tileLimit));
   const int *only = nullptr;
   if (tiled)
   {
      TileArgs a;
      a.pinfo = b->dPinfo; a.tileOff = b->dTileOff; a.B = b->B; a.mode = mode; a.pairs = pairs ? 1 : 0; a.nch = nch;
      a.C = pairs ? b->kmC : b->P.C; a.Cin = b->P.Cin; a.d = b->P.d > 0 ? b->P.d : 1; a.c0 = pairs ? c0 : 0;
      a.src = src; a.srcStride = srcStridePerKnot; a.km = b->dKM; a.coef = b->dCoef; a.edge = b->dEdge; a.dirty = b->dDirty;
      hipLaunchKernelGGL(k_tile_dirty_init, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, b->dPinfo, b->B, nch, b->dDirty);
      hipLaunchKernelGGL(k_spline_tile, dim3((unsigned)b->totalTiles, (unsigned)((nch + ST_CH - 1) / ST_CH)), dim3(ST_BLOCK), 0, st, a);
      const int64_t slots = (int64_t)b->totalTiles * nch;
      hipLaunchKernelGGL(k_spline_tile_check, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, st, a, b->totalTiles);
      only = b->dDirty;
      b->lastTileNch = nch;
   }
   // Large batches of pairs (no tiles): the single-pass kernel (spline_stream.hip.h) for every series of at least ST_MIN_KNOTS knots,
   // then the sequential kernel for what is left (short paths; a series whose boundary comparison failed -- never observed).
   // splineTiles: 2 = this kernel whatever the batch size (tests), 0 = never
   const bool stream = pairs && !tiled && b->totalTiles > 0 &&
                       (b->ctx->splineTiles == 2 || (b->ctx->splineTiles < 0 && (int64_t)b->B * nch > tileLimit));
   if (stream)
   {
      hipLaunchKernelGGL(k_tile_dirty_init, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, b->dPinfo, b->B, nch, b->dDirty);
      hipLaunchKernelGGL(k_spline_pairs_stream, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, b->dPinfo, b->B, nch, c0, b->kmC, b->dKM, b->dDirty);
      only = b->dDirty;
      b->lastTileNch = nch;
   }
   if (pairs) hipLaunchKernelGGL(k_spline_pairs, dim3((unsigned)((threads + bs - 1) / bs)), dim3(bs), 0, st, b->dPinfo, b->B, nch, c0, b->kmC, b->dKM, only);
   else
      hipLaunchKernelGGL(k_spline, dim3((unsigned)((threads + bs - 1) / bs)), dim3(bs), 0, st, b->dPinfo, b->B, nch, mode,
                         b->P.C, b->P.Cin, b->P.d > 0 ? b->P.d : 1, src, srcStridePerKnot, b->dElim, b->dCoef, only);
   HIP_TRY(hipGetLastError());
   return BATOTP_OK;
}

extern "C" int batotp_hip_precompute(batotp_batch *b, int32_t stage)
{
   if (!b || stage < 0 || stage > 2) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   hipStream_t st = b->ctx->stream;
   const int bs = 256;
   const unsigned gridKnots = (unsigned)((b->totalKnots + bs - 1) / bs);
   for (int p = 0; p < b->B; ++p)
      if (b->pinfo[p].n < 4 || b->pinfo[p].sres_c == 0.0) return BATOTP_ERR_STATE; // knots not uploaded
   if ((rc = joinK3(b))) return rc; // an overlapped per-knot evaluation still reads the splines
   if (stage == 0 || stage == 1) evStart(b, 1);
   if (stage == 0 || stage == 1)
   {
      if (b->dSC) hipLaunchKernelGGL(k_sites, dim3(gridKnots), dim3(bs), 0, st, b->dPinfo, b->B, b->dSC, b->totalKnots);
      if (b->compact)
      {
         rc = launchSpline(b, b->P.Cin, 0, nullptr, 0, true);
         if (rc) return rc;
      }
      else
      {
         rc = launchSpline(b, b->P.Cin, 0, b->dY, b->P.Cin);
         if (rc) return rc;
      }
      if (!(b->prob.flags & BATOTP_F_NO_SAMPLES))
         hipLaunchKernelGGL(k_samples, dim3(gridKnots), dim3(bs), 0, st, b->dPinfo, b->B, b->P.C, b->P.Cin, b->dSC, b->dCoef, b->dSamp,
                            b->dRes, b->totalKnots);
      HIP_TRY(hipGetLastError());
      b->kinDone = true;
      b->sitesSet = true;
   }
   if ((stage == 0 || stage == 2) && b->P.d > 0)
   {
      if (!b->kinDone) return BATOTP_ERR_STATE;
      const bool serialModel = b->hasSerial && !(b->prob.flags & BATOTP_F_PARALLEL);
      if (!(b->prob.flags & BATOTP_F_PARALLEL) && !serialModel && b->prob.robot_type != BATOTP_ROBOT_RR)
      {
         // robot.cpp:349-360: "No dynamics model provided for serial robotType"
         snprintf(g_err, sizeof(g_err), "no dynamics model for this serial robot: call batotp_hip_set_serial_model");
         return BATOTP_ERR_ARG;
      }
      if ((b->prob.flags & BATOTP_F_PARALLEL) && b->prob.robot_type != BATOTP_ROBOT_CSPR3DOF) return BATOTP_ERR_ARG; // robot.cpp:452-463
      if (serialModel)
      {
         const bool hostTrig = (b->prob.flags & BATOTP_F_HOST_TRIG) != 0;
         if (hostTrig && !b->jtrigSet) return BATOTP_ERR_STATE;
         hipLaunchKernelGGL(k_dyn_serial, dim3((unsigned)((b->totalKnots + KDS_BLOCK - 1) / KDS_BLOCK)), dim3(KDS_BLOCK), 0, st, b->dModel,
                            b->P.Cin, b->dPinfo, b->B, b->dSamp, hostTrig ? b->dJTrig : (const double *)nullptr, b->dDyn, b->totalKnots,
                            b->pairsAll ? b->dKM : (double *)nullptr, b->kmC, b->dRes);
      }
      else
      {
         const bool hostTrig = (b->prob.flags & BATOTP_F_HOST_TRIG) && b->prob.robot_type == BATOTP_ROBOT_RR;
         if (hostTrig && !b->trigSet) return BATOTP_ERR_STATE;
         hipLaunchKernelGGL(k_dynamics, dim3(gridKnots), dim3(bs), 0, st, b->P, b->dP, b->dPinfo, b->B, b->dSamp,
                            hostTrig ? b->dTrig : (const double *)nullptr, b->dDyn, b->totalKnots, b->pairsAll ? b->dKM : (double *)nullptr, b->kmC,
                            b->dRes);
      }
      HIP_TRY(hipGetLastError());
      // (pairs for all channels: the values are already in their slots, device channel Cin + 4 r + k; solved in place)
      if (b->pairsAll) rc = launchSpline(b, 4 * b->P.d, 0, nullptr, 0, true, b->P.Cin); // (mode 0: series e IS channel c0 + e)
      else rc = launchSpline(b, 4 * b->P.d, 1, b->dDyn, 4 * b->P.d);
      if (rc) return rc;
      if ((b->prob.flags & BATOTP_F_PARALLEL) && (b->prob.flags & BATOTP_F_PAR2SER))
      {
         for (int p = 0; p < b->B; ++p) b->pinfo[p].parallel_now = 0; // ba.cpp:937
         rc = pushPinfo(b);
         if (rc) return rc;
      }
      b->dynDone = true;
   }
   if (stage == 0 || stage == 1 || stage == 2) evStop(b, 1);
   HIP_TRY(hipStreamSynchronize(st));
   return BATOTP_OK;
}

// which constraint families the kernels must carry (template parameter FEAT of kernels.hip.h)
static int featureLevel(const batotp_batch *b)
{
   if (b->needPar) return 3;
   if (b->prob.flags & BATOTP_F_TRQ_ON) return 2;
   if (b->prob.flags & (BATOTP_F_CART_VEL_ON | BATOTP_F_CART_ACC_ON)) return 1;
   return (b->compact && !b->pairsAll) ? -1 : 0;
}

static int readyForSweep(const batotp_batch *b)
{
   if (!b->sitesSet) return BATOTP_ERR_STATE;
   return BATOTP_OK;
}

extern "C" int batotp_hip_pointwise_mvc(batotp_batch *b)
{
   if (!b) return BATOTP_ERR_ARG;
   int rc = readyForSweep(b);
   if (rc) return rc;
   rc = bind(b->ctx);
   if (rc) return rc;
   // a lane per knot, or (when a lane grouping is selected explicitly) a lane group per knot as in the sweep.
   // Measured on UR6, 16384 paths: lane per knot 366 ms, lane group per knot 461 ms (2 of 8 lanes idle, the
   // scalar work of a knot replicated over its 8 lanes) -- the lane-per-knot form is the default.
   const bool grouped = b->ctx->sweepGroup > 1 && !b->pairsAll; // (pairs for all channels: the lane-per-knot kernel forms its row from them)
   const int bs = grouped ? K3G_BLOCK : K3_BLOCK;
   const int64_t knotsPerBlock = grouped ? K3G_BLOCK / 8 : K3_BLOCK;
   const int64_t sliceKnots = (int64_t)1 << 27; // x 8 lanes = 2^30 threads per launch
   const unsigned grid = (unsigned)((b->totalKnots + knotsPerBlock - 1) / knotsPerBlock);
   // (compact splines read their pair rows directly: no LDS tile, which would only cap the occupancy)
   // LDS staging of the coefficient rows (coalesced copy, then conflict-free reads) is available but off: measured, it
   // is the occupancy it costs that matters -- UR6 rows (C = 6) 90 ms with the tile, 60 ms reading the rows directly;
   // cable robot (C = 18) 125 vs 53 ms.  BATOTP_K3_TILE=1 switches it on for experiments.
   const bool useTile = false;
   // velocity / acceleration-only problems: the kernel written for them (same bits; batotp_hip_set_k3_form(ctx, 0) runs the general one)
   const bool formVA = !grouped && b->ctx->k3Form == 1 && featureLevel(b) <= 0;
   const size_t ldsBytes = useTile ? sizeof(double) * (size_t)bs * (size_t)(b->P.C * 4 + 2) : 0;
   // overlap: K3 reads what the precompute wrote and nothing reads K3's output before the caller downloads it, so it
   // can share the GPU with the sweeps (second stream, joined by get_results / synchronize / the next precompute)
   const bool async = b->ctx->overlap != 0 && !b->mvcInCurves; // in the curve slots the values must be complete before a sweep starts
   double *mvcOut = b->mvcInCurves ? reinterpret_cast<double *>(b->dRev) : b->dMvc;
   const int64_t mvcSlot = b->mvcInCurves ? 2 * b->cap : 0;
   hipStream_t k3s = async ? b->ctx->stream2 : b->ctx->stream;
   if ((rc = joinK3(b))) return rc;
   if (async)
   {
      HIP_TRY(hipEventRecord(b->ctx->evJoin, b->ctx->stream));
      HIP_TRY(hipStreamWaitEvent(k3s, b->ctx->evJoin, 0));
   }
   hipEventRecord(b->ev[2][0], k3s);
#define LAUNCH_K3(F)                                                                                                                     \
   do {                                                                                                                                  \
      if (grouped)                                                                                                                       \
         for (int64_t first = 0; first < b->totalKnots; first += sliceKnots)                                                            \
         {                                                                                                                               \
            const int64_t cnt = (b->totalKnots - first) < sliceKnots ? (b->totalKnots - first) : sliceKnots;                            \
            hipLaunchKernelGGL(k_pointwise_grp<F>, dim3((unsigned)((cnt + knotsPerBlock - 1) / knotsPerBlock)), dim3(bs), 0,           \
                               k3s, b->P, b->dPinfo, b->B, b->dP, b->dSC, b->dCoef, b->compact ? b->dKM : (double *)nullptr, mvcOut, first,         \
                               first + cnt, mvcSlot);                                                                                             \
         }                                                                                                                               \
      else if (formVA && F <= 0)                                                                                                        \
         hipLaunchKernelGGL(k_pointwise_va<(F <= 0 ? F : 0)>, dim3((unsigned)((b->totalKnots + K3V_BLOCK - 1) / K3V_BLOCK)), dim3(K3V_BLOCK), 0, k3s, b->P,  \
                            b->dPinfo, b->B, b->dP, b->dSC, b->dCoef, b->compact ? b->dKM : (double *)nullptr, mvcOut, b->totalKnots, mvcSlot,           \
                            b->ctx->fastForward);                                                                                         \
      else hipLaunchKernelGGL(k_pointwise<F>, dim3(grid), dim3(bs), ldsBytes, k3s, b->P, b->dPinfo, b->B, b->dP, b->dSC,      \
                              b->dCoef, b->compact ? b->dKM : (double *)nullptr, mvcOut, b->totalKnots, useTile ? 1 : 0, mvcSlot);                                      \
   } while (0)
   switch (featureLevel(b))
   {
   case -1: LAUNCH_K3(-1); break;
   case 0: LAUNCH_K3(0); break;
   case 1: LAUNCH_K3(1); break;
   case 2: LAUNCH_K3(2); break;
   default: LAUNCH_K3(3); break;
   }
#undef LAUNCH_K3
   hipEventRecord(b->ev[2][1], k3s);
   b->evValid[2] = true;
   HIP_TRY(hipGetLastError());
   if (async) b->k3Pending = true;
   else HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   if (b->mvcInCurves)
   {
      b->mvcValid = true;
      b->revDone = false; b->revGone = false; // whatever curves the slots held are overwritten
      b->revStale = true; b->fwdStale = true;
   }
   return BATOTP_OK;
}

extern "C" int batotp_hip_upload_forward_curve(batotp_batch *b, int32_t path, const double *s, const double *sdot, int64_t n, double t_total)
{
   if (!b || !s || !sdot || path < 0 || path >= b->B || n < 2 || n > b->cap) return BATOTP_ERR_ARG;
   if (b->inPlace || b->mvcInCurves) { snprintf(g_err, sizeof(g_err), "upload_forward_curve: not for batches that share curve slots"); return BATOTP_ERR_STATE; }
   // the batch-wide state flags below (revDone, fwdStale) describe EVERY path: the call is for the batch of one that
   // BA::interpOutputData builds around a caller's Traj, not for marking one path of many as swept
   if (b->B != 1) { snprintf(g_err, sizeof(g_err), "upload_forward_curve: only for a batch of one path"); return BATOTP_ERR_STATE; }
   int rc = bind(b->ctx);
   if (rc) return rc;
   if ((rc = joinK3(b))) return rc; // an overlapped pointwise evaluation may still be running on the second stream
   double *tmp = nullptr;
   rc = xferReserve(b->ctx, sizeof(double) * 2 * (size_t)n, &tmp);
   if (rc) return rc;
   HIP_TRY(hipMemcpyAsync(tmp, s, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, b->ctx->stream));
   HIP_TRY(hipMemcpyAsync(tmp + n, sdot, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, b->ctx->stream));
   const int bs = 256;
   hipLaunchKernelGGL(k_curve_pack, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream, b->dFwd + (int64_t)path * b->cap, tmp, tmp + n, n);
   HIP_TRY(hipGetLastError());
   batotp_path_result r;
   HIP_TRY(hipMemcpyAsync(&r, b->dRes + path, sizeof(r), hipMemcpyDeviceToHost, b->ctx->stream));
   hipError_t e = hipStreamSynchronize(b->ctx->stream);
   if (e == hipSuccess)
   {
      r.n_fwd = n; r.steps_fwd = n - 1; r.t_total = t_total; r.status_fwd = 0; r.status_rev &= BATOTP_ST_SEG_ERROR;
      if (r.n_rev < 2) r.n_rev = 2; // the sweeps of this path ran elsewhere (the output stage only asks for a forward curve)
      e = hipMemcpy(b->dRes + path, &r, sizeof(r), hipMemcpyHostToDevice);
   }
   if (e != hipSuccess) return hipFail(e, "upload_forward_curve");
   b->revDone = true;
   b->fwdStale = false;
   b->mvcValid = false;
   return BATOTP_OK;
}

extern "C" int batotp_hip_set_path_integ_res(batotp_batch *b, int32_t path0, int32_t n, const double *integ_res)
{
   if (!b || !integ_res || path0 < 0 || n < 0 || path0 + n > b->B) return BATOTP_ERR_ARG;
   for (int k = 0; k < n; ++k)
      // positive and finite, or NaN: what the automatic rule leaves for a robot without Cartesian limits (0/0 kept by std::min /
      // std::max, ba.cpp:519-533) -- such a path takes no integration step and ends with BATOTP_ST_MAX_INTEG_TIME; an infinite
      // step is never a result of the rule
      if ((!(integ_res[k] > 0) && integ_res[k] == integ_res[k]) || std::isinf(integ_res[k])) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   for (int k = 0; k < n; ++k) b->pinfo[path0 + k].integ_res = integ_res[k];
   return pushPinfo(b);
}

// ---------------------------------------------------------------------------------------------
// Gate of the flat stage / bisection loop.  The AUTOMATIC loop choice uses it only when (a) this library was compiled by the
// toolchain the loop was validated with and (b) a canary on this device agrees with the nested loops: two small batches
// (compact pairs and coefficient rows) of 24 velocity / acceleration-only paths, 8 per wavefront, of three kinds -- ordinary
// ones, paths that crawl under a tiny acceleration limit (long searches for a first feasible speed) and paths on which every
// bisection fails (a negative limit: 100 iterations per stage, stale sddot) until they run out of curve capacity -- i.e. the
// population on which the torque instantiation of the loop once went wrong.  Result rows and reverse curves
// must be identical bit for bit.  Runs once per context, the first time the automatic choice would take the flat loop
// (a few tens of milliseconds); an explicit batotp_hip_set_sweep_hold is honoured without it.
// ---------------------------------------------------------------------------------------------
static int flatCanaryOnce(batotp_ctx *ctx, bool compact, bool *same)
{
   *same = false;
   const int B = 24, nJ = 6;
   const int64_t cap = 40;
   batotp_problem prob;
   memset(&prob, 0, sizeof(prob));
   prob.n_joints = nJ; prob.n_cart = 0; prob.robot_type = BATOTP_ROBOT_GENJNT;
   prob.flags = BATOTP_F_JNT_ACC_ON | (compact ? (BATOTP_F_NO_SAMPLES | BATOTP_F_COMPACT_SPLINES) : 0u);
   for (int j = 0; j < nJ; ++j) { prob.jnt_vel_max[j] = 5.0; prob.jnt_acc_max[j] = 10.0; }
   prob.jnt_acc_max[4] = 1e-7; // paths on which joint 4 moves crawl
   prob.jnt_acc_max[5] = -1.0; // paths on which joint 5 moves fail every bisection
   prob.jnt_thresh = 1e-6; prob.quad_rad_thresh = 1e-12;
   prob.integ_res = 0.02; prob.max_integ_time = 1e6;
   std::vector<int64_t> nk(B);
   for (int p = 0; p < B; ++p) nk[p] = 12 + (p * 7) % 23;
   batotp_batch *b = nullptr;
   int rc = batotp_hip_batch_create(ctx, &prob, B, nk.data(), cap, &b);
   if (rc) return rc;
   std::vector<double> y, sres(B);
   uint64_t lcg = 0x9E3779B97F4A7C15ull;
   auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) * (1.0 / 9007199254740992.0); };
   for (int p = 0; p < B; ++p)
   {
      const int64_t N = nk[p];
      sres[p] = 0.05 + 0.01 * (double)(p % 5);
      for (int j = 0; j < nJ; ++j)
      {
         const bool still = (j == 4 && p % 3 != 1) || (j == 5 && p % 3 != 2); // kinds: 0 ordinary, 1 crawls, 2 fails
         const double c1 = 2.0 * rnd() - 1.0, c2 = 2.0 * rnd() - 1.0, c3 = 2.0 * rnd() - 1.0;
         for (int64_t i = 0; i < N; ++i)
         {
            const double x = (double)i / (double)(N - 1);
            y.push_back(still ? 0.25 : c1 * x + c2 * x * x * (1.0 - x) + c3 * x * x * x);
         }
      }
   }
   std::vector<batotp_path_result> rows[2];
   std::vector<double2> curves[2], curvesF[2];
   const int holdSaved[2] = {ctx->sweepHold[0], ctx->sweepHold[1]}, groupSaved = ctx->sweepGroup, ppwSaved = ctx->pathsPerWave;
   const int formSaved = ctx->flatForm, ffSaved = ctx->fastForward, certSaved = ctx->certHold;
   // the canary compares exactly what the automatic choice launches: 8 lanes per path, k_sweep8, the certified fast-forward on --
   // whatever the developer switches of this context say at the moment
   ctx->sweepGroup = 8; ctx->pathsPerWave = 8; ctx->flatForm = 1; ctx->fastForward = 1; ctx->certHold = -1;
   rc = batotp_hip_upload_knots(b, 0, B, y.data(), sres.data());
   if (!rc) rc = batotp_hip_precompute(b, 0);
   for (int form = 0; form < 2 && !rc; ++form)
   {
      ctx->sweepHold[0] = form == 0 ? -1 : 4; // nested, then the flat loop as the automatic choice would run it
      ctx->sweepHold[1] = form == 0 ? -1 : 8;
      rc = batotp_hip_sweep(b, -1);
      if (!rc) rc = batotp_hip_sweep(b, +1);
      rows[form].resize(B);
      if (!rc) rc = batotp_hip_get_results(b, rows[form].data());
      curves[form].resize((size_t)B * (size_t)cap);
      curvesF[form].resize((size_t)B * (size_t)cap);
      if (!rc && hipMemcpy(curves[form].data(), b->dRev, sizeof(double2) * curves[form].size(), hipMemcpyDeviceToHost) != hipSuccess) rc = BATOTP_ERR_HIP;
      if (!rc && hipMemcpy(curvesF[form].data(), b->dFwd, sizeof(double2) * curvesF[form].size(), hipMemcpyDeviceToHost) != hipSuccess) rc = BATOTP_ERR_HIP;
   }
   ctx->sweepHold[0] = holdSaved[0]; ctx->sweepHold[1] = holdSaved[1]; ctx->sweepGroup = groupSaved; ctx->pathsPerWave = ppwSaved;
   ctx->flatForm = formSaved; ctx->fastForward = ffSaved; ctx->certHold = certSaved;
   batotp_hip_batch_destroy(b);
   if (rc) return rc;
   bool eq = memcmp(rows[0].data(), rows[1].data(), sizeof(batotp_path_result) * (size_t)B) == 0;
   int finished = 0, stalled = 0;
   for (int p = 0; p < B && eq; ++p)
   {
      const int64_t n = rows[0][p].n_rev;
      if (n > 0) ++finished; else ++stalled;
      const size_t at = (size_t)p * (size_t)cap + (size_t)(cap - n);
      if (n > 0 && memcmp(&curves[0][at], &curves[1][at], sizeof(double2) * (size_t)n) != 0) eq = false;
      const int64_t nf = rows[0][p].n_fwd;
      const size_t atF = (size_t)p * (size_t)cap;
      if (nf > 0 && memcmp(&curvesF[0][atF], &curvesF[1][atF], sizeof(double2) * (size_t)nf) != 0) eq = false;
   }
   // the canary must contain both populations, otherwise it says nothing
   if (eq && (finished < 4 || stalled < 4))
   {
      snprintf(g_err, sizeof(g_err), "flat-loop canary degenerate: %d finished, %d stalled paths", finished, stalled);
      return BATOTP_ERR_STATE;
   }
   *same = eq;
   return BATOTP_OK;
}

static int flatLoopStatus(batotp_ctx *ctx)
{
   if (ctx->flatStatus != 0) return ctx->flatStatus;
   if (strcmp(ctx->builtWith, kFlatValidatedWith) != 0) return ctx->flatStatus = -1;
   ctx->flatStatus = -3; // (the canary's own sweeps set their loop form explicitly and never ask)
   bool sameA = false, sameB = false;
   if (flatCanaryOnce(ctx, true, &sameA) != BATOTP_OK || flatCanaryOnce(ctx, false, &sameB) != BATOTP_OK) return ctx->flatStatus = -3;
   return ctx->flatStatus = (sameA && sameB) ? 1 : -2;
}

extern "C" int batotp_hip_flat_loop_status(batotp_ctx *ctx, int32_t *status)
{
   if (!ctx || !status) return BATOTP_ERR_ARG;
   int rc = bind(ctx);
   if (rc) return rc;
   *status = flatLoopStatus(ctx);
   return BATOTP_OK;
}

extern "C" int batotp_hip_last_sweep_launch(batotp_batch *b, int32_t dir, int32_t *lanes, int32_t *paths_per_wave, int32_t *hold)
{
   if (!b || (dir != 1 && dir != -1)) return BATOTP_ERR_ARG;
   const int k = dir == -1 ? 0 : 1;
   if (b->lastLanes[k] == 0) return BATOTP_ERR_STATE;
   if (lanes) *lanes = b->lastLanes[k];
   if (paths_per_wave) *paths_per_wave = b->lastPpw[k];
   if (hold) *hold = b->lastHold[k];
   return BATOTP_OK;
}

template <int G>
static void launchSweep(batotp_batch *b, SweepArgs &a)
{
   const int maxPpw = 64 / G;
   int ppw = b->ctx->pathsPerWave;
   if (ppw <= 0)
   {
      // automatic: with few paths spread them over more wavefronts (latency-bound regime), with many
      // fill every lane (throughput-bound regime); aim at ~2 wavefronts per SIMD (1024 SIMDs, the
      // register budget of the kernel admits 2 per SIMD).  The reverse sweep runs more bisection
      // iterations, whose count differs between the paths of a wavefront, so it prefers fewer
      // paths per wavefront than the forward sweep (measured, B = 4096: rev best at 2, fwd at 4).
      // rounded up: one path too many per wavefront costs little, a second round of wavefronts on the SIMDs costs a lot
      ppw = (a.dir == -1) ? (b->B + 2047) / 2048 : (b->B + 1023) / 1024;
      // forward, more than 4096 paths: two wavefronts per SIMD as well, but never fewer than the 4 paths per wavefront that
      // were best at 4096 (11 264 GEN7DOF paths: 2600 ms with 6 paths per wavefront / 1878 wavefronts, 2762 ms with 8 / 1408 --
      // 1.4 wavefronts per SIMD leaves the SIMDs with one wavefront idle while those with two finish)
      if (a.dir == 1 && ppw > 4) ppw = std::max(4, (b->B + 2047) / 2048);
   }
   if (ppw < 1) ppw = 1;
   if (ppw > maxPpw) ppw = maxPpw;
   a.ppw = ppw;
   // software prefetch of the lines ahead of the cursors: the reverse sweep always (descending addresses: -17 % at 4096 paths);
   // the forward sweep only while every path has a wavefront to itself -- there the sweep waits a third of its time for the
   // dependent loads of segment changes (SQ_WAIT_ANY at B = 1, profiles/), with many paths per wavefront it cost +5 %
   a.touch = (a.dir == -1) ? 1 : (ppw == 1 ? 3 : 0);
   a.ff = sweep8FastForward(b->ctx);   // (the general kernel k_sweep reads no bit of it)
   a.holdc = std::max(1, certHoldOf(b->ctx));
   if (b->ctx->sweepTouch[a.dir == -1 ? 0 : 1] >= 0) a.touch = b->ctx->sweepTouch[a.dir == -1 ? 0 : 1];
   const unsigned waves = (unsigned)((b->B + ppw - 1) / ppw);
   const unsigned grid = (waves + (K4_BLOCK / 64) - 1) / (K4_BLOCK / 64);
   bool uni = true;
   for (int p = 0; p < b->B; ++p) uni = uni && b->pinfo[p].uniform;
   hipStream_t st = b->ctx->stream;
   int hold = b->ctx->sweepHold[a.dir == -1 ? 0 : 1];
   // automatic: the flat stage / bisection loop for the reverse sweep (measured on the bench batches: -25 % with hold 4,
   // bit-identical results) -- where it exists (below) and only behind the gate of flatLoopStatus: validated toolchain and
   // a canary on this device --, the nested loops for the forward sweep (which does not gain)
   if (hold == -2)
   {
      // reverse: hold 4 (round 2: 855 against 1130 ms on the UR6 bench batch).  Forward: k_sweep8 with hold 8 -- the nested
      // loops' schedule, every path of the wavefront starts its stage together -- because that kernel executes 9 % fewer
      // vector and 37 % fewer scalar instructions than the nested form of k_sweep (round 4: 444 against 560 ms at 16 384
      // paths, profiles/r04_a_*); the general kernel's own flat form does not gain in the forward sweep and keeps the nested loops
      // Only the form the canary of flatLoopStatus compares: 8 lanes per path in k_sweep8.  With 2 or 4 lanes per path, or with
      // k_sweep's own flat instantiation selected (batotp_hip_set_flat_form 0), the automatic choice keeps the nested loops --
      // an explicit batotp_hip_set_sweep_hold remains the developer's switch for those.
      const bool candidate = G == 8 && featureLevel(b) <= 0 && uni && b->ctx->flatForm == 1 && b->cap < ((int64_t)1 << 30);
      hold = -1;
      if (candidate && flatLoopStatus(b->ctx) == 1) hold = a.dir == -1 ? 4 : 8;
   }
   if (hold > 8) hold = 8;
   a.hold = hold;
   // The flat loop exists for the 8-lane layout of the velocity / acceleration-only problems (FEAT <= 0) on uniform knot
   // sites and nowhere else: the instantiation with the serial torque branch gave hold-dependent results on stalled paths
   // with this toolchain (ROCm 7.2.0 hipcc, clang 22; DESIGN.md 4, tools/experiments/), so problems with torque or
   // Cartesian limits always run the nested loops, and so do paths with uploaded (non-uniform) sites.
   const bool flat = (G == 8 || G == 4 || G == 2) && hold >= 0 && featureLevel(b) <= 0 && uni;
   b->lastLanes[a.dir == -1 ? 0 : 1] = G; b->lastPpw[a.dir == -1 ? 0 : 1] = ppw; b->lastHold[a.dir == -1 ? 0 : 1] = flat ? hold : -1;
   // the flat loop of the 8-lane layout written for the instruction count (sweep8.hip.h); 32-bit step counters
   const bool form8 = flat && (G == 8 || G == 4) && b->ctx->flatForm == 1 && b->cap < ((int64_t)1 << 30);
   b->lastForm8[a.dir == -1 ? 0 : 1] = form8;
#define LAUNCH_K4(F)                                                                           \
   do {                                                                                        \
      if (form8 && F <= 0)                                                                     \
      {                                                                                        \
         constexpr int F8 = F <= 0 ? F : 0, G8 = G == 4 ? 4 : 8;                               \
         if (a.dir == 1) hipLaunchKernelGGL((k_sweep8<G8, F8, 1>), dim3(grid), dim3(S8_BLOCK), 0, st, a);  \
         else hipLaunchKernelGGL((k_sweep8<G8, F8, -1>), dim3(grid), dim3(S8_BLOCK), 0, st, a);    \
      }                                                                                        \
      else if (flat) hipLaunchKernelGGL((k_sweep<G, F, true, ((G == 8 || G == 4 || G == 2) && F <= 0)>), dim3(grid), dim3(K4_BLOCK), 0, st, a);  \
      else if (uni) hipLaunchKernelGGL((k_sweep<G, F, true>), dim3(grid), dim3(K4_BLOCK), 0, st, a);      \
      else hipLaunchKernelGGL((k_sweep<G, F, false>), dim3(grid), dim3(K4_BLOCK), 0, st, a);        \
   } while (0)
   switch (featureLevel(b))
   {
   case -1: LAUNCH_K4(-1); break;
   case 0: LAUNCH_K4(0); break;
   case 1: LAUNCH_K4(1); break;
   case 2: LAUNCH_K4(2); break;
   default: LAUNCH_K4(3); break;
   }
#undef LAUNCH_K4
}

// the one-path-per-wavefront kernel of sweep1.hip.h: joint velocity / acceleration limits only, uniform knot sites
static bool sweep1Applies(const batotp_batch *b)
{
   // every constraint family except the torque limits of a parallel mechanism that is still parallel (the cable robot without
   // isPar2Ser: an LU solve per joint and limit); uniform knot sites
   if (featureLevel(b) > 2) return false;
   for (int p = 0; p < b->B; ++p)
      if (!b->pinfo[p].uniform) return false;
   return true;
}

static void launchSweep1(batotp_batch *b, SweepArgs &a)
{
   a.ppw = 1; a.hold = -1; a.touch = 0; a.ff = b->ctx->fastForward; a.holdc = 0;
   hipStream_t st = b->ctx->stream;
   const bool cableLines = featureLevel(b) == 2 && (b->P.flags & BATOTP_F_PARALLEL) != 0 && b->pairsAll;
   // Two paths per wavefront (k_sweep1's NP = 2; the cable robot in serial form with every channel as pairs): where one path per
   // wavefront would leave wavefronts queueing for the two slots per SIMD the kernel's registers allow, or where the caller asks
   // for it (batotp_hip_set_paths_per_wave(ctx, 2) with 64 lanes per path).  Measured on BASELINE config 5 (N = 2e5, reverse +
   // forward ms; profiles/r05_h_*): up to 2048 paths every path has a resident wavefront either way and the sweeps last as long as
   // the slowest path -- 1147 + 1326 with one path per wavefront, 1238 + 1474 with two (a wavefront that serves two paths is 8-11 %
   // slower); 3072 paths 1613 + 1496 against 1244 + 1479; 4096 paths 2159 + 2033 against 1228 + 1479.  The crossover is at ~2300.
   const int want = b->ctx->pathsPerWave;
   // ... and the compact velocity / acceleration-only layout (round 5; the batch sizes between the one-path kernel and k_sweep8)
   const bool compactVA = featureLevel(b) == -1;
   // (a two-path wavefront of this family is 20 % slower in the reverse sweep and as fast as a one-path wavefront in the forward
   //  sweep: 2048 paths 713 / 513 against 596 / 515 ms; one-path wavefronts queue beyond 2048 paths: 3072 paths 775 / 829 ms
   //  against 901 / 646)
   const bool two = (cableLines && (want == 2 || (want <= 0 && b->B > 2304))) ||
                    (compactVA && (want == 2 || (want <= 0 && b->B > (a.dir == -1 ? 3500 : 2304))));
   a.ppw = two ? 2 : 1;
   b->lastLanes[a.dir == -1 ? 0 : 1] = 64; b->lastPpw[a.dir == -1 ? 0 : 1] = a.ppw; b->lastHold[a.dir == -1 ? 0 : 1] = -1;
   const unsigned perBlock = (unsigned)(S1_BLOCK / 64) * (unsigned)a.ppw;
   const unsigned grid = ((unsigned)b->B + perBlock - 1) / perBlock;
   if (two && cableLines)
   {
      if (a.dir == 1) hipLaunchKernelGGL((k_sweep1<2, 1, 0, true, 2>), dim3(grid), dim3(S1_BLOCK), 0, st, a);
      else hipLaunchKernelGGL((k_sweep1<2, -1, 0, true, 2>), dim3(grid), dim3(S1_BLOCK), 0, st, a);
      return;
   }
   if (two)
   {
      if (a.dir == 1) hipLaunchKernelGGL((k_sweep1<-1, 1, 0, false, 2>), dim3(grid), dim3(S1_BLOCK), 0, st, a);
      else hipLaunchKernelGGL((k_sweep1<-1, -1, 0, false, 2>), dim3(grid), dim3(S1_BLOCK), 0, st, a);
      return;
   }
   // (the last template argument: the batch keeps all its channels as pairs)
#define LAUNCH_S1(F, FFORM)                                                                                              \
   do {                                                                                                                 \
      if (b->pairsAll)                                                                                                  \
      {                                                                                                                 \
         if (a.dir == 1) hipLaunchKernelGGL((k_sweep1<F, 1, FFORM, true>), dim3(grid), dim3(S1_BLOCK), 0, st, a);       \
         else hipLaunchKernelGGL((k_sweep1<F, -1, FFORM, true>), dim3(grid), dim3(S1_BLOCK), 0, st, a);                 \
      }                                                                                                                 \
      else                                                                                                              \
      {                                                                                                                 \
         if (a.dir == 1) hipLaunchKernelGGL((k_sweep1<F, 1, FFORM, false>), dim3(grid), dim3(S1_BLOCK), 0, st, a);      \
         else hipLaunchKernelGGL((k_sweep1<F, -1, FFORM, false>), dim3(grid), dim3(S1_BLOCK), 0, st, a);                \
      }                                                                                                                 \
   } while (0)
   if (featureLevel(b) == 2)
   {
      // the fast-forward of the bisection in the form that fits the mechanism: a parallel robot converted to serial form has
      // a3 = 0 (tension bounds are lines in sdot^2), a serial chain has friction (bounds quadratic in sdot)
      const bool lines = (b->P.flags & BATOTP_F_PARALLEL) != 0;
      if (lines) LAUNCH_S1(2, 0);
      else LAUNCH_S1(2, 1);
   }
   else if (featureLevel(b) == 1) LAUNCH_S1(1, 0);
   else if (b->compact && !b->pairsAll)
   {
      if (a.dir == 1) hipLaunchKernelGGL((k_sweep1<-1, 1>), dim3(grid), dim3(S1_BLOCK), 0, st, a);
      else hipLaunchKernelGGL((k_sweep1<-1, -1>), dim3(grid), dim3(S1_BLOCK), 0, st, a);
   }
   else LAUNCH_S1(0, 0);
#undef LAUNCH_S1
}

extern "C" int batotp_hip_sweep(batotp_batch *b, int32_t dir)
{
   if (!b || (dir != 1 && dir != -1)) return BATOTP_ERR_ARG;
   int rc = readyForSweep(b);
   if (rc) return rc;
   if (dir == 1 && (!b->revDone || b->revGone)) return BATOTP_ERR_STATE; // in-place curves: the reverse curve is consumed by one forward sweep
   rc = bind(b->ctx);
   if (rc) return rc;
   b->mvcValid = false; // BATOTP_F_MVC_IN_CURVES: the sweep writes over the pointwise values
   SweepArgs a;
   a.P = b->P; a.dP = b->dP; a.pinfo = b->dPinfo; a.sC = b->dSC; a.coef = b->dCoef; a.km = b->compact ? b->dKM : nullptr; // (a zero-size allocation is not a null pointer: the kernels of the row layouts take a non-null km for "all channels as pairs")
   a.rev = b->dRev; a.fwd = b->dFwd; a.res = b->dRes; a.sink = b->dSink; a.prof = b->dMvc; a.cap = b->cap; a.B = b->B; a.dir = dir; a.ppw = 1;
   a.order = b->ctx->pathOrder ? b->dOrder : nullptr;
#ifdef S8_PROFILE
   a.prof = b->dProf;
   hipMemsetAsync(b->dProf, 0, sizeof(double) * ((size_t)b->B * 16 + 16), b->ctx->stream);
#endif
   const int which = dir == -1 ? 3 : 4;
   int lanes = b->ctx->sweepGroup;
   if (lanes == 0)
   {
      // automatic: while the batch cannot fill the 8-lane layout anyway (<= 4 paths per SIMD), give a
      // path 16 lanes and split the interval bounds over the two halves (fewer instructions per check)
      // automatic (measured on UR6, N = 100k): up to ~2k paths the batch is latency-bound and the
      // 32-lane layout (four bisection candidates per pass) wins; beyond that the 8-lane layout, which
      // packs more paths per wavefront, has the higher throughput
      lanes = (b->B <= (a.dir == -1 ? 2048 : 1024)) ? 32 : 8;
      // ... and where every path has a wavefront to itself anyway, the kernel written for that case (sweep1.hip.h).  It also
      // wins for a while beyond that, with its wavefronts queueing for the two slots per SIMD its registers leave: the 8-lane
      // layout at 2-4 paths per wavefront is latency-bound and slower (GEN7DOF, N = 5e4, reverse: 3072 paths 944 vs 1882 ms,
      // 6144 paths 1784 vs 2024, 8192 paths 2328 vs 2137; forward: 3072 paths 923 vs 981, 4096 paths 1118 vs 1031; UR6 alike)
      // (round 4, against k_sweep8 instead of k_sweep -- N = 5e4, reverse / forward ms: 2048 paths 594 / 516 vs 1117 / 687, 4096 paths
      //  1051 / 1008 vs 1231 / 738: the reverse crossover moved from ~6000 to ~4900 paths, the forward one stays near 3000)
      if (sweep1Applies(b) && b->B <= (a.dir == -1 ? 4608 : 3072)) lanes = 64;
      // (round 5, compact velocity / acceleration-only batches: with TWO paths per wavefront the one-path kernel keeps its lead over
      //  the 8-lane layout up to ~7600 paths in the reverse and ~5000 in the forward sweep -- GEN7DOF, N = 5e4, reverse / forward ms,
      //  one path | two paths | 8 lanes: 4096 paths 1055 / 1004 | 902 / 646 | 1345 / 738; 6144 paths 1460 / 1480 | 1152 / 1050 |
      //  1460 / 868; 8192 paths 1905 / 1950 | 1588 / 1260 | 1505 / 868: profiles/r05_i_*; launchSweep1 picks one or two)
      if (sweep1Applies(b) && featureLevel(b) == -1 && b->B <= (a.dir == -1 ? 7600 : 5000)) lanes = 64;
   }
   if (b->pairsAll)
   {
      // rows exist in the LDS windows of the one-path-per-wavefront kernel only
      if (!sweep1Applies(b)) { snprintf(g_err, sizeof(g_err), "pairs for all channels need uniform knot sites and constraints in serial form"); return BATOTP_ERR_STATE; }
      lanes = 64;
   }
   if (lanes == 64 && !sweep1Applies(b)) lanes = 32; // a parallel mechanism's torque limits, uploaded sites: the general kernel
   // the gate of the flat loop (its canary launches sweeps of its own) is settled before this sweep's timed region starts
   if (b->ctx->sweepHold[dir == -1 ? 0 : 1] == -2 && (lanes == 8 || lanes == 4 || lanes == 2) && featureLevel(b) <= 0) (void)flatLoopStatus(b->ctx);
   evStart(b, which);
   switch (lanes)
   {
   case 64: launchSweep1(b, a); break;
   case 32: launchSweep<32>(b, a); break;
   case 1: launchSweep<1>(b, a); break;
   case 16: launchSweep<16>(b, a); break;
   case 4: launchSweep<4>(b, a); break;
   case 2: launchSweep<2>(b, a); break;
   default: launchSweep<8>(b, a); break;
   }
   evStop(b, which);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   if (dir == -1) { b->revDone = true; b->revGone = false; b->revStale = false; }
   else
   {
      b->fwdStale = false;
      if (b->inPlace) b->revGone = true;
   }
   return BATOTP_OK;
}

extern "C" int batotp_hip_optimize(batotp_batch *b)
{
   int rc = batotp_hip_precompute(b, 0);
   if (rc) return rc;
   rc = batotp_hip_sweep(b, -1);
   if (rc) return rc;
   return batotp_hip_sweep(b, 1);
}

// ---------------------------------------------------------------------------------------------
// results
// ---------------------------------------------------------------------------------------------
extern "C" int batotp_hip_get_results(batotp_batch *b, batotp_path_result *out)
{
   if (!b || !out) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   if ((rc = joinK3(b))) return rc;
   HIP_TRY(hipMemcpyAsync(out, b->dRes, sizeof(batotp_path_result) * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

extern "C" int batotp_hip_download_curve(batotp_batch *b, int32_t path, int32_t which, double *s, double *sdot, int64_t cap, int64_t *n)
{
   if (!b || path < 0 || path >= b->B || (which != 1 && which != -1)) return BATOTP_ERR_ARG;
   if (which == -1 && b->revGone) return BATOTP_ERR_STATE; // BATOTP_F_CURVES_IN_PLACE: overwritten by the forward sweep
   if (which == -1 ? b->revStale : b->fwdStale) return BATOTP_ERR_STATE; // BATOTP_F_MVC_IN_CURVES: overwritten by a pointwise evaluation
   int rc = bind(b->ctx);
   if (rc) return rc;
   batotp_path_result r;
   HIP_TRY(hipMemcpy(&r, b->dRes + path, sizeof(r), hipMemcpyDeviceToHost));
   const int64_t avail = which == 1 ? r.n_fwd : r.n_rev;
   if (n) *n = avail;
   const int64_t m = avail < cap ? avail : cap;
   if (m <= 0 || (!s && !sdot)) return BATOTP_OK;
   const double2 *src = which == 1 ? b->dFwd + (int64_t)path * b->cap : b->dRev + (int64_t)path * b->cap + (b->cap - avail);
   double *tmp = nullptr;
   rc = xferReserve(b->ctx, sizeof(double) * 2 * (size_t)m, &tmp);
   if (rc) return rc;
   const int bs = 256;
   hipLaunchKernelGGL(k_curve_unpack, dim3((unsigned)((m + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream, src, tmp, tmp + m, m);
   if (s) hipMemcpyAsync(s, tmp, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, b->ctx->stream);
   if (sdot) hipMemcpyAsync(sdot, tmp + m, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, b->ctx->stream);
   hipError_t e = hipStreamSynchronize(b->ctx->stream);
   if (e != hipSuccess) return hipFail(e, "download_curve");
   return BATOTP_OK;
}

extern "C" int batotp_hip_download_coeffs(batotp_batch *b, int32_t path, int32_t channel, double *c)
{
   if (!b || !c || path < 0 || path >= b->B) return BATOTP_ERR_ARG;
   const int dc = devChannel(b, channel);
   if (dc < 0) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   const PathInfo &pi = b->pinfo[path];
   const int bs = 256;
   if (b->compact)
      hipLaunchKernelGGL(k_coef_from_sol, dim3((unsigned)((pi.n + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream,
                         b->dKM + pi.koff * b->kmC * 2, b->kmC, dc, pi.n, b->dStage);
   else
      hipLaunchKernelGGL(k_coef_gather, dim3((unsigned)((pi.n + bs - 1) / bs)), dim3(bs), 0, b->ctx->stream,
                         b->dCoef + pi.koff * b->P.C * 4, b->P.C, dc, pi.n, b->dStage);
   HIP_TRY(hipMemcpyAsync(c, b->dStage, sizeof(double) * 4 * (size_t)pi.n, hipMemcpyDeviceToHost, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

extern "C" int batotp_hip_download_samples(batotp_batch *b, int32_t path, int32_t channel, double *out)
{
   if (!b || !out || path < 0 || path >= b->B || channel < 0 || channel >= b->P.Cin) return BATOTP_ERR_ARG;
   if (b->prob.flags & BATOTP_F_NO_SAMPLES) return BATOTP_ERR_STATE;
   int rc = bind(b->ctx);
   if (rc) return rc;
   const PathInfo &pi = b->pinfo[path];
   HIP_TRY(hipMemcpyAsync(out, b->dSamp + pi.koff * b->P.Cin * 3 + (int64_t)channel * 3 * pi.n, sizeof(double) * 3 * (size_t)pi.n,
                          hipMemcpyDeviceToHost, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

extern "C" int batotp_hip_download_dyn(batotp_batch *b, int32_t path, int32_t k, int32_t row, double *out)
{
   if (!b || !out || path < 0 || path >= b->B || k < 1 || k > 4 || row < 0 || row >= b->P.d) return BATOTP_ERR_ARG;
   if (b->pairsAll) return BATOTP_ERR_STATE; // no dynamics array: the values are c0 of the channel's rows (batotp_hip_download_coeffs)
   int rc = bind(b->ctx);
   if (rc) return rc;
   const PathInfo &pi = b->pinfo[path];
   HIP_TRY(hipMemcpyAsync(out, b->dDyn + pi.koff * 4 * b->P.d + ((int64_t)(k - 1) * b->P.d + row) * pi.n, sizeof(double) * (size_t)pi.n,
                          hipMemcpyDeviceToHost, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

extern "C" int batotp_hip_download_mvc(batotp_batch *b, int32_t path, double *sdot_max, double *sddot_l, double *sddot_h)
{
   if (!b || path < 0 || path >= b->B) return BATOTP_ERR_ARG;
   int rc = bind(b->ctx);
   if (rc) return rc;
   if ((rc = joinK3(b))) return rc;
   if (b->mvcInCurves && !b->mvcValid) return BATOTP_ERR_STATE; // a sweep has used the curve slots since the last pointwise evaluation
   const PathInfo &pi = b->pinfo[path];
   const double *base = b->mvcInCurves ? reinterpret_cast<const double *>(b->dRev + (int64_t)path * b->cap) : b->dMvc + pi.koff * 3;
   const size_t sz = sizeof(double) * (size_t)pi.n;
   if (sdot_max) HIP_TRY(hipMemcpyAsync(sdot_max, base, sz, hipMemcpyDeviceToHost, b->ctx->stream));
   if (sddot_l) HIP_TRY(hipMemcpyAsync(sddot_l, base + pi.n, sz, hipMemcpyDeviceToHost, b->ctx->stream));
   if (sddot_h) HIP_TRY(hipMemcpyAsync(sddot_h, base + 2 * pi.n, sz, hipMemcpyDeviceToHost, b->ctx->stream));
   HIP_TRY(hipStreamSynchronize(b->ctx->stream));
   return BATOTP_OK;
}

extern "C" int batotp_hip_results_device_ptr(batotp_batch *b, void **ptr, int64_t *bytes)
{
   if (!b) return BATOTP_ERR_ARG;
   if (ptr) *ptr = b->dRes;
   if (bytes) *bytes = (int64_t)sizeof(batotp_path_result) * b->B;
   return BATOTP_OK;
}

extern "C" int batotp_hip_pack_curves(batotp_batch *b, int32_t which, int32_t path0, int32_t n_paths, void *dst_dev, int64_t dst_points,
                                      int64_t *total_points)
{
   if (!b || (which != 1 && which != -1) || path0 < 0 || n_paths < 0 || path0 + n_paths > b->B || !total_points) return BATOTP_ERR_ARG;
   *total_points = 0;
   if (which == -1 && b->revGone) return BATOTP_ERR_STATE; // BATOTP_F_CURVES_IN_PLACE: overwritten by the forward sweep
   if (which == -1 ? b->revStale : b->fwdStale) return BATOTP_ERR_STATE; // BATOTP_F_MVC_IN_CURVES: overwritten by a pointwise evaluation
   if (n_paths == 0) return BATOTP_OK;
   int rc = bind(b->ctx);
   if (rc) return rc;
   std::vector<batotp_path_result> res((size_t)n_paths);
   HIP_TRY(hipMemcpy(res.data(), b->dRes + path0, sizeof(batotp_path_result) * (size_t)n_paths, hipMemcpyDeviceToHost));
   std::vector<int64_t> meta(2 * (size_t)n_paths + 1);   // off[n + 1], start[n]
   int64_t total = 0;
   for (int k = 0; k < n_paths; ++k)
   {
      const int64_t cnt = which == 1 ? res[k].n_fwd : res[k].n_rev;
      meta[k] = total;
      meta[(size_t)n_paths + 1 + k] = which == 1 ? 0 : b->cap - cnt;   // the reverse curve is stored at the end of its slot
      total += cnt;
   }
   meta[n_paths] = total;
   *total_points = total;
   if (total == 0) return BATOTP_OK;
   if (!dst_dev || dst_points < total) return BATOTP_ERR_ARG;
   double *stage = nullptr;
   rc = xferReserve(b->ctx, sizeof(int64_t) * meta.size(), &stage);
   if (rc) return rc;
   hipStream_t st = b->ctx->stream;
   HIP_TRY(hipMemcpyAsync(stage, meta.data(), sizeof(int64_t) * meta.size(), hipMemcpyHostToDevice, st));
   const int64_t *dOff = reinterpret_cast<const int64_t *>(stage), *dStart = dOff + n_paths + 1;
   const int bs = 256;
   hipLaunchKernelGGL(k_curves_pack, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, st, which == 1 ? b->dFwd : b->dRev, b->cap, path0, n_paths,
                      dOff, dStart, static_cast<double2 *>(dst_dev), total);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipStreamSynchronize(st));
   return BATOTP_OK;
}

extern "C" int batotp_hip_last_kernel_ms(batotp_batch *b, int32_t which, float *ms)
{
   if (!b || !ms || which < 1 || which > 4) return BATOTP_ERR_ARG;
   if (!b->evValid[which]) return BATOTP_ERR_STATE;
   int rc = bind(b->ctx);
   if (rc) return rc;
   HIP_TRY(hipEventSynchronize(b->ev[which][1]));
   HIP_TRY(hipEventElapsedTime(ms, b->ev[which][0], b->ev[which][1]));
   return BATOTP_OK;
}

extern "C" int batotp_hip_batch_bytes(batotp_batch *b, int64_t *bytes)
{
   if (!b || !bytes) return BATOTP_ERR_ARG;
   *bytes = b->bytes;
   return BATOTP_OK;
}

#include "resample_api.inc"
#include "output_api.inc"
