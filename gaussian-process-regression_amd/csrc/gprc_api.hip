// gprc_api.hip -- the C ABI declared in include/gprc_native.h, composed from the gfx950 launchers.
//
// Host side of the hot path (reference R/GPRclass.R:127-170, R/GPCclass.R:66-115): owns device
// memory through opaque handles, stages host arrays when the caller hands over host pointers (the
// `.Call` case) and uses device pointers in place (the resident-data case).  No CPU arithmetic on
// matrices happens here: without a gfx950 device every entry point fails with GPRC_ERR_NO_DEVICE /
// GPRC_ERR_HIP.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "gprc_internal.h"

namespace gprc {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  g_last_error = std::string("HIP error '") + hipGetErrorString(e) + "' in " + what + " at " + file + ":" + std::to_string(line);
  (void)hipGetLastError();
  return e == hipErrorOutOfMemory ? GPRC_ERR_NOMEM : GPRC_ERR_HIP;
}

// ---- event profiler ------------------------------------------------------------------------------
namespace {
struct ProfRec { int kind; double flops, bytes; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
std::vector<ProfRec> g_prof_open;  // begun, not ended (per kind nesting is not used)
hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
std::mutex g_prof_mu;  // gprc_mgpu_gpr_predict runs one host thread per rank
bool prof_enabled() { return g_prof_on; }
void prof_begin(hipStream_t s, int kind) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{kind, 0.0, 0.0, prof_event(), nullptr};
  (void)hipEventRecord(r.e0, s);
  g_prof_open.push_back(r);
}
void prof_end(hipStream_t s, int kind, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (size_t i = g_prof_open.size(); i-- > 0;) {
    if (g_prof_open[i].kind != kind) continue;
    ProfRec r = g_prof_open[i];
    g_prof_open.erase(g_prof_open.begin() + (long)i);
    r.flops = flops; r.bytes = bytes; r.e1 = prof_event();
    (void)hipEventRecord(r.e1, s);
    g_prof_recs.push_back(r);
    return;
  }
}

}  // namespace gprc

using namespace gprc;

constexpr int SVC_TRACE_PANELS = 48;

struct gprc_ctx {
  uint64_t id = 0;             // unique per context ever created: a model remembers (pointer, id), so a LATER context at the same address is not mistaken for its own
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int* info_dev = nullptr;     // LAPACK info written by the diagonal-block kernel
  void* sync_dev = nullptr;    // 64 bytes of flags for the fused panel kernel (zeroed before every launch, stream-ordered)
                               // + another 64 for launches on the look-ahead stream
  // the factor service's persistent launch runs on a high-priority side stream beside the caller's kernels (factor_group_service)
  hipStream_t side_stream = nullptr;
  hipStream_t side_stream2 = nullptr;  // the shared service's second launch (the 4-wave roles) runs beside the first
  void* svc_trace = nullptr;          // GPRC_SERVICE_TRACE: 16 stamps x SVC_TRACE_PANELS of the last factor-service sweep (measurement)
  hipEvent_t ev_pool[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned ev_next = 0;
  double* scal_dev = nullptr;  // 8 doubles of scalar results
  size_t chunk_bytes = (size_t)40 << 30;  // budget for one K_star^T chunk (n* = n = 65536 in one piece: fewer, fuller launches)
  // grow-only workspace slots (predict chunks): a multi-GiB hipMalloc/hipFree per call costs 100s of ms
  double* ws[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t ws_cap[4] = {0, 0, 0, 0};
  int64_t vt_pad = 0;                     // extra doubles in the chunk's leading dimension (keeps it off powers of two)
  // Free-list of device blocks released by calls on this context (exact-size reuse).  fit() evaluates the same n
  // dozens to hundreds of times; without it every dens(v) pays seven hipMalloc/hipFree pairs (each hipFree is a
  // device-wide sync).  Reuse is safe without events: everything on a context runs on its one stream, in order.
  struct Block { void* p; size_t bytes; };
  std::vector<Block> pool;
  size_t pool_bytes = 0;
  size_t pool_cap = (size_t)16 << 30;     // GPRC_POOL_BYTES; blocks larger than the cap are never kept
};

enum ModelType { MODEL_GPR = 1, MODEL_GPC = 2 };

struct gprc_model {
  gprc_ctx* ctx = nullptr;
  uint64_t ctx_id = 0;
  int type = 0;
  KernelSpec ks{};
  int64_t n = 0, d = 0, n_pad = 0;
  double* X = nullptr;       // d x n
  double* y = nullptr;       // n_pad (zero padded)
  double* packed = nullptr;  // factor, packed block columns
  double* winv = nullptr;    // inverses of the 128x128 diagonal blocks
  double* alpha = nullptr;   // n_pad (GPR) ; g = (y+1)/2 - P (GPC)
  double* f_hat = nullptr;   // GPC
  double* sw = nullptr;      // GPC sqrt(W)
  double* work = nullptr;    // trsv partials
  double logp = 0.0, noise = 0.0, logq = 0.0;
  bool borrowed = false;     // X, y, packed, winv, alpha belong to the caller
};

namespace {

thread_local gprc_ctx* g_cur_ctx = nullptr;  // set by use_device(): whose pool the scoped temporaries below use

// Contexts that exist.  A model may outlive its context (a host language's garbage collector finalising objects in arbitrary order
// at exit: Python does): gprc_model_free then must not touch the context's stream or block pool.
static std::mutex g_live_mu;
static std::vector<const gprc_ctx*> g_live_ctx;
static uint64_t g_next_ctx_id = 1;
static void ctx_register(gprc_ctx* c) { std::lock_guard<std::mutex> lk(g_live_mu); c->id = g_next_ctx_id++; g_live_ctx.push_back(c); }
static void ctx_unregister(const gprc_ctx* c) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  g_live_ctx.erase(std::remove(g_live_ctx.begin(), g_live_ctx.end(), c), g_live_ctx.end());
}
static bool ctx_alive(const gprc_ctx* c, uint64_t id) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return std::find(g_live_ctx.begin(), g_live_ctx.end(), c) != g_live_ctx.end() && c->id == id;
}

int pool_alloc(gprc_ctx* ctx, size_t bytes, void** out) {
  if (bytes == 0) bytes = 8;
  if (ctx) {
    for (size_t i = ctx->pool.size(); i-- > 0;)
      if (ctx->pool[i].bytes == bytes) {
        *out = ctx->pool[i].p;
        ctx->pool_bytes -= bytes;
        ctx->pool.erase(ctx->pool.begin() + (long)i);
        return 0;
      }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess && ctx && !ctx->pool.empty()) {  // give the cached blocks back and retry once
    (void)hipGetLastError();
    for (auto& b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
    e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  return 0;
}
void pool_release(gprc_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  if (bytes == 0) bytes = 8;
  if (!ctx || bytes > ctx->pool_cap) { (void)hipFree(p); return; }
  while (!ctx->pool.empty() && ctx->pool_bytes + bytes > ctx->pool_cap) {  // evict the oldest
    (void)hipFree(ctx->pool.front().p);
    ctx->pool_bytes -= ctx->pool.front().bytes;
    ctx->pool.erase(ctx->pool.begin());
  }
  ctx->pool.push_back({p, bytes});
  ctx->pool_bytes += bytes;
}
void pool_trim(gprc_ctx* ctx) {
  for (auto& b : ctx->pool) (void)hipFree(b.p);
  ctx->pool.clear();
  ctx->pool_bytes = 0;
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeDevice;
}

// Input staging: device pointers pass through, host arrays are copied to a temporary.
struct In {
  const double* dev = nullptr;
  double* tmp = nullptr;
  gprc_ctx* owner = nullptr;
  size_t bytes = 0;
  ~In() { if (tmp) pool_release(owner, tmp, bytes); }
  int set(hipStream_t s, const double* p, int64_t count) {
    if (count <= 0) { dev = nullptr; return 0; }
    if (!p) { set_error("null input pointer"); return GPRC_ERR_ARG; }
    if (is_device_ptr(p)) { dev = p; return 0; }
    owner = g_cur_ctx;
    bytes = sizeof(double) * (size_t)count;
    GPRC_TRY(pool_alloc(owner, bytes, (void**)&tmp));
    GPRC_HIP(hipMemcpyAsync(tmp, p, bytes, hipMemcpyHostToDevice, s));
    dev = tmp;
    return 0;
  }
};
// Output staging: device pointers are written in place; host arrays get a temporary + D2H at finish().
struct Out {
  double* dev = nullptr;
  double* tmp = nullptr;
  double* host = nullptr;
  int64_t count = 0;
  gprc_ctx* owner = nullptr;
  ~Out() { if (tmp) pool_release(owner, tmp, sizeof(double) * (size_t)count); }
  int set(double* p, int64_t cnt) {
    count = cnt;
    if (cnt <= 0) return 0;
    if (!p) { set_error("null output pointer"); return GPRC_ERR_ARG; }
    if (is_device_ptr(p)) { dev = p; return 0; }
    host = p;
    owner = g_cur_ctx;
    GPRC_TRY(pool_alloc(owner, sizeof(double) * (size_t)cnt, (void**)&tmp));
    dev = tmp;
    return 0;
  }
  int finish(hipStream_t s) {
    if (host && count > 0) GPRC_HIP(hipMemcpyAsync(host, tmp, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    return 0;
  }
};

// workspace slot `slot` of the context, at least `count` doubles (contents undefined)
int ws_get(gprc_ctx* ctx, int slot, int64_t count, double** out) {
  if (count <= 0) count = 1;
  if (ctx->ws_cap[slot] < count) {
    if (ctx->ws[slot]) {
      GPRC_HIP(hipStreamSynchronize(ctx->stream));
      GPRC_HIP(hipFree(ctx->ws[slot]));
      ctx->ws[slot] = nullptr;
      ctx->ws_cap[slot] = 0;
    }
    hipError_t e = hipMalloc(&ctx->ws[slot], sizeof(double) * (size_t)count);
    if (e != hipSuccess) { ctx->ws[slot] = nullptr; return hip_fail(e, "hipMalloc(workspace)", __FILE__, __LINE__); }
    ctx->ws_cap[slot] = count;
  }
  *out = ctx->ws[slot];
  return 0;
}

struct DevMem {  // scoped device allocation
  double* p = nullptr;
  gprc_ctx* owner = nullptr;
  size_t bytes = 0;
  ~DevMem() { if (p) pool_release(owner, p, bytes); }
  int alloc(int64_t count) {
    if (count <= 0) count = 1;
    owner = g_cur_ctx;
    bytes = sizeof(double) * (size_t)count;
    return pool_alloc(owner, bytes, (void**)&p);
  }
};

int make_spec(int kernel, const double* params, int n_params, int64_t d, KernelSpec* ks) {
  if (n_params > MAX_PARAMS) { set_error("at most 256 kernel parameters (a per-coordinate sigma of the linear kernel needs d <= 256; a scalar sigma has no limit)"); return GPRC_ERR_ARG; }
  if (n_params < 0 || (n_params > 0 && !params)) { set_error("bad kernel parameter vector"); return GPRC_ERR_ARG; }
  bool ok = false;
  switch (kernel) {
    case GPRC_CONSTANT: ok = n_params == 1; break;
    case GPRC_LINEAR: ok = n_params == 1 || n_params == d; break;
    case GPRC_POLYNOMIAL: case GPRC_GAMMAEXP: case GPRC_RATQUAD: ok = n_params == 2; break;
    case GPRC_SQREXP: ok = n_params == 1; break;
    default: set_error("unknown kernel id"); return GPRC_ERR_ARG;
  }
  if (!ok) { set_error("wrong number of kernel parameters for this kernel"); return GPRC_ERR_ARG; }
  ks->id = kernel;
  ks->n_params = n_params;
  for (int i = 0; i < MAX_PARAMS; ++i) ks->p[i] = i < n_params ? params[i] : 0.0;
  return 0;
}

int use_device(const gprc_ctx* ctx) {
  if (!ctx) { set_error("null context"); return GPRC_ERR_ARG; }
  GPRC_HIP(hipSetDevice(ctx->device));
  g_cur_ctx = const_cast<gprc_ctx*>(ctx);
  return 0;
}

// ---- factorisation of all panels of a packed matrix (single GPU) ------------------------------
// 128-column sub-step j of panel p: factor + invert the diagonal block, solve the rows below it, update the rest of the
// panel.  After it, columns [128 j, 128 (j+1)) of the panel are final (what the pipelined broadcast relies on).
// 128-column sub-step j of panel p, left-looking inside the panel: first the columns of block j receive the
// contributions of the blocks 0..j-1 of the same panel in one pass (K = 128 j, C tile in the accumulators: the same
// products in the same order as three K = 128 updates from the left, with half the C traffic and a third of the
// launches), then the diagonal block is factored + inverted and the rows below it are solved.  After it the columns
// [128 j, 128 (j+1)) of the panel are final (what the pipelined broadcast relies on).
// part: 1 = the whole sub-step, 2 = nothing (kept so that callers written for the right-looking form still work), 0 = 1.
int factor_subpanel(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, int j, int part, double* winv, int* info_dev) {
  if (part == 2) return 0;
  hipStream_t s = ctx->stream;
  const int64_t ld = panel_ld(n_pad, p);
  double* pan = packed + panel_offset(n_pad, p);
  const int64_t cj = (int64_t)j * NBI;
  double* wblk = winv + (p * (NB / NBI) + j) * NBI * NBI;
  if (cj > 0)  // rows cj.. of block j's columns -= (rows cj.. of the earlier columns) * (rows cj..cj+127 of the earlier columns)^T
    GPRC_TRY(launch_gemm_nt(s, pan + cj + cj * ld, ld, pan + cj, ld, pan + cj, ld, ld - cj, NBI, cj, 1, PK_GEMM_INNER));
  GPRC_TRY(launch_potf2_inv(s, pan + cj + cj * ld, ld, wblk, info_dev, (int)(p * NB + cj)));
  const int64_t below = ld - cj - NBI;
  if (below > 0) GPRC_TRY(launch_trsm_panel(s, pan + (cj + NBI) + cj * ld, ld, below, wblk));
  return 0;
}

// The whole panel: ONE launch (panel_fused_kernel: the four sub-steps overlap across row strips, dependencies carried by
// device-side flags), bit-identical to the twelve launches of the sub-step form.  GPRC_PANEL=steps selects the latter.
int factor_panel(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev) {
  static const bool steps = [] { const char* e = std::getenv("GPRC_PANEL"); return e && std::strcmp(e, "steps") == 0; }();
  if (!steps) return launch_panel_fused(ctx->stream, packed, n_pad, p, winv, info_dev, ctx->sync_dev);
  for (int j = 0; j < NB / NBI; ++j) GPRC_TRY(factor_subpanel(ctx, packed, n_pad, p, j, 0, winv, info_dev));
  return 0;
}

// All panels of a packed matrix on one GPU, asynchronously (info stays on the device).  Schedule: the panels are taken
// in GROUPS; before a group is factored its panels receive the contributions of every earlier panel in one
// left-looking pass (trailing_left_kernel, K = g0 NB, C tile held in the accumulators); inside the group the panels
// update each other right-looking.  Bit-identical to the plain right-looking sweep (same products, same order).  A
// group is the shortest run of panels whose lower tiles number >= 8192: a left-looking tile is long (K / 512 x 110 us),
// so a pass needs many generations of tiles per CU or the partially filled last one costs more than the saved
// prologues (measured at n = 32768 / 65536: 1024 tiles -6 %, 8192 tiles +3 % / +7 % on the fit against right-looking).
// GPRC_FACTOR=right: one group = right-looking; GPRC_FACTOR=<tiles> changes the threshold.
// `to` waits for everything enqueued on `from` so far (events recycled round-robin: a wait captures the record made here)
int stream_after(gprc_ctx* ctx, hipStream_t to, hipStream_t from) {
  hipEvent_t& ev = ctx->ev_pool[ctx->ev_next++ % 8];
  if (!ev) GPRC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  GPRC_HIP(hipEventRecord(ev, from));
  GPRC_HIP(hipStreamWaitEvent(to, ev, 0));
  return 0;
}

// Set when a factorisation under the factor service ended in a device-side wait timeout: the persistent launch and the caller's
// kernels did not run concurrently (a tool that serialises dispatches, e.g. rocprofv3 --pmc).  From then on this process factors with
// one fused launch per panel (what GPRC_SERVICE=0 selects).
static std::atomic<bool> g_service_off{false};

// Lower tiles a group of the left-looking schedule should have at least (see factor_all_async)
static int64_t want_for(int64_t n_pad) {
  // below n_pad = 20480 one group -- the plain right-looking sweep under the factor service -- is fastest (measured,
  // profiles/r02_factor_schedules.txt: n = 16384 30.0 ms against 29.2..32.8 with groups of 1000..6000 tiles); from there on 8192
  return n_pad < 20480 ? INT64_MAX : 8192;
}

static bool service_carries_inverse(int64_t n_pad) {
  static const int forced = [] { const char* e = std::getenv("GPRC_SERVICE_INV"); return e ? std::atoi(e) : -1; }();   // 0 / 1: experiment switch
  return forced >= 0 ? forced != 0 : n_pad < 20480;   // measured (profiles/r03_experiments.txt): no difference up to 20480 (+0.5 % there without)
}

// One GROUP of panels [g0, g1) with the FACTOR SERVICE (kernels_chol.hip): the group's columns have received every earlier panel
// (left-looking pass, or g0 = 0); inside the group the sweep is right-looking with the whole dependent chain -- diagonal blocks, the
// strips around them, the rows of the next diagonal block and that block's update -- in ONE persistent 21-workgroup launch on a side
// stream; the caller's stream carries the ordinary strips and the rest of the within-group update, one launch per panel, tied to the
// service by counters.  Same tiles in the same k order: bit-identical.  sync: panel_service_sync_bytes(P), zeroed once per
// factorisation; launches: service launches on it so far (this one included).
int factor_group_service(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_dev, double* inv, int64_t g0, int64_t g1, void* sync,
                         void* trace, int launches) {
  hipStream_t s = ctx->stream, side = ctx->side_stream;
  GPRC_TRY(stream_after(ctx, side, s));                       // everything the group's first panel needs precedes the service
  // The explicit inverses ride in the service (four more resident workgroups, off the chain) below n_pad = 20480, where the whole
  // matrix is one group; in the grouped schedule beyond, one launch after the sweep computes them (factor_all_async) and the four
  // CUs go to the update.
  const bool shared = service_shared(n_pad);
  if (shared) {
    hipStream_t side2 = ctx->side_stream2;
    GPRC_TRY(stream_after(ctx, side2, s));
    GPRC_TRY(launch_panel_service(side, packed, n_pad, winv, info_dev, sync, trace, service_carries_inverse(n_pad) ? inv : nullptr, g0, g1, 1));
    GPRC_TRY(launch_panel_service(side2, packed, n_pad, winv, info_dev, sync, trace, service_carries_inverse(n_pad) ? inv : nullptr, g0, g1, 2));
  } else
  GPRC_TRY(launch_panel_service(side, packed, n_pad, winv, info_dev, sync, trace, service_carries_inverse(n_pad) ? inv : nullptr, g0, g1));
  GPRC_TRY(launch_service_gate(s, n_pad, info_dev, sync, launches));   // nothing that waits on the service starts before the service is resident
  GPRC_TRY(launch_panel_strips(s, packed, n_pad, g0, winv, info_dev, sync, trace));        // the later panels' strips ride in the update kernels
  // (Round 3 measured a batched form of this loop -- panel s applied at once only to the next B + 1 panels, the batch's B panels
  //  to everything further in ONE K = 512 B pass, bit-identical -- and it was SLOWER at every size: n = 16384 29.9 -> 31.6 / 30.9 /
  //  30.6 ms for B = 2 / 4 / 8, because the chain idles behind the long pass and the near launches are small and ragged.
  //  profiles/r03_experiments.txt; removed.)
  // ... in ONE persistent launch for the whole group (trailing_sweep_kernel: no partly filled last generation of tiles and no drained
  // GPU at every panel boundary; n = 8192 / 16384: see profiles/r03_factor_schedules.txt).  GPRC_SWEEP=0: one launch per panel.
  static const bool per_panel = [] { const char* e = std::getenv("GPRC_SWEEP"); return e && std::atoi(e) == 0; }();
  if (per_panel) {
    for (int64_t p = g0; p + 1 < g1; ++p) GPRC_TRY(launch_trailing_service(s, packed, n_pad, p, winv, info_dev, sync, trace, g1));
  } else {
    GPRC_TRY(launch_trailing_sweep(s, packed, n_pad, g0, g1, winv, info_dev, sync, trace, service_workgroups(service_carries_inverse(n_pad) && inv, n_pad)));
  }
  GPRC_TRY(stream_after(ctx, s, side));
  if (shared) GPRC_TRY(stream_after(ctx, s, ctx->side_stream2));
  return 0;
}

// inv (may be null): n_pad x NB doubles that receive, per panel, the explicit inverse of its diagonal block (transposed) -- what
// launch_trsv works with; the factor service produces it on the side, the other schedules in one launch after the sweep
int factor_all_async(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_dev, double* inv, bool* used_service = nullptr) {
  hipStream_t s = ctx->stream;
  const int64_t P = n_pad / NB;
  // Groups: the shortest run of panels with >= `want` lower tiles (GPRC_FACTOR=<tiles>, =right: one group); a left-looking tile is
  // long, so a pass needs many generations of tiles per CU or its ragged last one costs more than the saved prologues.
  const char* mode = std::getenv("GPRC_FACTOR");
  int64_t want = want_for(n_pad);
  if (mode && std::strcmp(mode, "right") == 0) want = INT64_MAX;
  else if (mode && std::atoll(mode) > 0) want = std::atoll(mode);
  // GPRC_PANEL=steps: the launch-per-stage panel kernels; GPRC_SERVICE=0: one fused launch per panel inside the groups instead of
  // the factor service.  (A look-ahead sweep on two streams, with and without CU masks, was measured in round 2 and removed in
  // favour of the service: DESIGN.md section 3, profiles/r02_experiments.txt.)
  static const bool panel_steps = [] { const char* e = std::getenv("GPRC_PANEL"); return e && std::strcmp(e, "steps") == 0; }();
  static const int sv_env = [] { const char* e = std::getenv("GPRC_SERVICE"); return e ? std::atoi(e) : -1; }();
  const bool service = !panel_steps && sv_env != 0 && P >= 2 && !g_service_off.load();
  if (used_service) *used_service = service;
  DevMem sync;   // flags of every panel + the counters; goes back to the pool when every launch below has been ordered behind it
  void* trace = nullptr;
  if (service) {
    if (!ctx->side_stream) {
      int lo = 0, hi = 0;
      GPRC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      GPRC_HIP(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, hi));
      GPRC_HIP(hipStreamCreateWithPriority(&ctx->side_stream2, hipStreamNonBlocking, hi));
    }
    GPRC_TRY(sync.alloc((int64_t)(panel_service_sync_bytes(P) + 7) / 8));
    GPRC_HIP(hipMemsetAsync(sync.p, 0, panel_service_sync_bytes(P), s));
    static const bool want_trace = std::getenv("GPRC_SERVICE_TRACE") != nullptr;
    if (want_trace && P <= SVC_TRACE_PANELS) {
      if (!ctx->svc_trace) GPRC_HIP(hipMalloc(&ctx->svc_trace, SVC_TRACE_PANELS * 16 * sizeof(int64_t)));
      GPRC_HIP(hipMemsetAsync(ctx->svc_trace, 0, SVC_TRACE_PANELS * 16 * sizeof(int64_t), s));
      trace = ctx->svc_trace;
    }
  }
  int launches = 0;
  auto sweep = [&]() -> int {
    for (int64_t g0 = 0; g0 < P;) {
      int64_t g1 = g0, tiles = 0;
      while (g1 < P && tiles < want) { tiles += (int64_t)TPP * TPP * (P - g1) - TPP * (TPP - 1) / 2; ++g1; }
      GPRC_TRY(launch_trailing_left(s, packed, n_pad, g0, g1));
      if (service) {
        GPRC_TRY(factor_group_service(ctx, packed, n_pad, winv, info_dev, inv, g0, g1, sync.p, trace, ++launches));
      } else {
        for (int64_t p = g0; p < g1; ++p) {
          GPRC_TRY(factor_panel(ctx, packed, n_pad, p, winv, info_dev));
          if (p + 1 < g1) GPRC_TRY(launch_trailing_update(s, packed, n_pad, p, p + 1, g1, 1));
        }
      }
      g0 = g1;
    }
    return (inv && !(service && service_carries_inverse(n_pad))) ? launch_inv512(s, packed, n_pad, winv, inv, 0, P) : 0;
  };
  const int rc = sweep();
  if (rc != 0 && service) {
    // a launch failed half way: the persistent service kernel may still be running on the side stream and spinning on the flags in
    // `sync` (its waits are bounded).  The block must not go back to the pool -- to be handed to somebody else -- before it has left.
    (void)hipStreamSynchronize(ctx->side_stream);
    if (ctx->side_stream2) (void)hipStreamSynchronize(ctx->side_stream2);
    (void)hipStreamSynchronize(s);
  }
  return rc;
}

int factor_all(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_host, double* inv = nullptr, bool* used_service = nullptr);

// factor_all for callers that can rebuild the matrix.  If THIS call ran under the factor service and a device-side wait timed out
// (see g_service_off), the service is switched off for the process (with one line on stderr: it is a permanent change of schedule),
// the matrix is rebuilt (refill) and factored again with one fused launch per panel.  Decided per call, not once per process: two
// host threads that time out in the same window (each with its own context) both retry.
template <class Refill>
int factor_all_or_refill(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_host, double* inv, Refill refill) {
  bool used_service = false;
  int rc = factor_all(ctx, packed, n_pad, winv, info_host, inv, &used_service);
  if (rc == GPRC_ERR_HIP && *info_host == GPRC_INFO_WAIT_TIMEOUT && used_service) {
    if (!g_service_off.exchange(true))
      std::fprintf(stderr, "gprc: a device-side wait of the factor service timed out (kernels of two streams did not run concurrently, e.g. under "
                           "rocprofv3 --pmc); the service is now OFF for this process and the factorisation is repeated with one launch per panel\n");
    GPRC_TRY(refill());
    rc = factor_all(ctx, packed, n_pad, winv, info_host, inv);
  }
  return rc;
}

int factor_all(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_host, double* inv, bool* used_service) {
  hipStream_t s = ctx->stream;
  GPRC_HIP(hipMemsetAsync(ctx->info_dev, 0, sizeof(int), s));
  GPRC_TRY(factor_all_async(ctx, packed, n_pad, winv, ctx->info_dev, inv, used_service));
  GPRC_HIP(hipMemcpyAsync(info_host, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  GPRC_HIP(hipStreamSynchronize(s));
  if (*info_host < 0) {
    set_error("factorisation: a device-side dependency wait timed out (info = " + std::to_string(*info_host) + ")" + wait_timeout_report() +
              "; the factor is not valid.  The factor service needs its persistent launch and the caller's kernels to run CONCURRENTLY: "
              "under a tool that serialises dispatches (e.g. rocprofv3 --pmc) set GPRC_SERVICE=0");
    return GPRC_ERR_HIP;
  }
  return 0;
}

// vt (m_pad x n_pad) := vt * L^-T.  Schedules with bit-identical results (same products, same order):
//   right-looking: after panel p is solved, subtract its contribution from every column to the right (K = NB per pass);
//   left-looking:  before a GROUP of G panels is solved, subtract the contributions of ALL earlier panels from the
//                  group's columns in one pass (K = p NB), the C tile staying in the accumulators -- one C load/store
//                  and one tile prologue instead of p; inside the group the panels update each other right-looking.
// G is the smallest group that gives a pass (m_pad / 128) * 4 G >= 4096 tiles (eight generations of two workgroups on
// each of 256 CUs: the partially filled last generation of long tiles stays cheap); G >= P degenerates to plain right-looking.  GPRC_SOLVE=right forces that, =left forces G = 1, =<n> G = n.
// sspart != nullptr: the panel solve that finalises a 128-column block also leaves that block's per-row sum of squares in
// sspart[block * m_pad + row] (n_pad / 128 blocks): colSums(v * v) without another pass over the chunk.
// tri_row0 >= 0 (fit()'s gradient, diag(K^-1)): the m_pad rows of vt are rows tri_row0, tri_row0 + 1, ... of the IDENTITY.  Row i of
// the result, (L^-1 e_{tri_row0 + i})^T, is zero left of column tri_row0 + i, so (a) a panel touches only the rows that start at or
// left of its last column and (b) a row tile's left-looking pass starts at the tile's first column: n^3 / 3 flops for the whole
// inverse instead of n^3.  Everything skipped is a product with an exact zero: the same bits as the dense solve
// (test_fit_gradient_triangular_solve_is_bit_identical); sspart must have been zeroed (rows never reached keep their zeros).
int solve_rows(gprc_ctx* ctx, const double* packed, const double* winv, int64_t n_pad, double* vt, int64_t ldv, int64_t m_pad,
               double* sspart = nullptr, int64_t tri_row0 = -1) {
  hipStream_t s = ctx->stream;
  const int64_t P = n_pad / NB;
  const char* mode = std::getenv("GPRC_SOLVE");
  int64_t G = (4096 + (m_pad / 128) * (NB / NBI) - 1) / ((m_pad / 128) * (NB / NBI));
  if (mode && std::strcmp(mode, "left") == 0) G = 1;
  else if (mode && std::atoi(mode) > 0) G = std::atoi(mode);  // an explicit group size
  if ((mode && std::strcmp(mode, "right") == 0) || G > P) G = P;
  auto rows_of = [&](int64_t col_end) {   // rows of the chunk that are not identically zero left of column col_end
    if (tri_row0 < 0) return m_pad;
    return std::max<int64_t>(0, std::min<int64_t>(m_pad, pad_up(col_end - tri_row0, 128)));
  };
  for (int64_t g0 = 0; g0 < P; g0 += G) {
    const int64_t g1 = std::min(P, g0 + G);  // panels [g0, g1)
    const int64_t mg = rows_of(g1 * NB);
    if (mg == 0) continue;
    GPRC_TRY(launch_solve_left(s, vt, ldv, mg, packed, n_pad, g0, g1 - g0, tri_row0));
    static const bool panel_steps = [] { const char* e = std::getenv("GPRC_SOLVE_PANEL"); return e && std::strcmp(e, "steps") == 0; }();
    for (int64_t p = g0; p < g1; ++p) {
      const int64_t ld = panel_ld(n_pad, p);
      const double* pan = packed + panel_offset(n_pad, p);
      const int64_t mp = rows_of((p + 1) * NB);
      if (mp == 0) continue;
      if (!panel_steps) {                   // the four sub-steps of the panel in one launch (GPRC_SOLVE_PANEL=steps: seven launches, same bits)
        GPRC_TRY(launch_solve_panel_fused(s, vt, ldv, mp, packed, n_pad, p, winv, sspart, m_pad));
      } else
      for (int j = 0; j < NB / NBI; ++j) {  // inside the panel, left-looking by 128-column blocks (K = 128 j, as factor_subpanel)
        const int64_t cj = p * NB + (int64_t)j * NBI;  // global column
        const double* wblk = winv + (p * (NB / NBI) + j) * NBI * NBI;
        if (j > 0)
          GPRC_TRY(launch_gemm_nt(s, vt + cj * ldv, ldv, vt + p * NB * ldv, ldv, pan + (int64_t)j * NBI, ld, mp, NBI, (int64_t)j * NBI, 0,
                                  PK_GEMM_INNER));
        GPRC_TRY(launch_trsm_panel(s, vt + cj * ldv, ldv, mp, wblk, sspart ? sspart + (cj / NBI) * m_pad : nullptr));
      }
      const int64_t right = (g1 - (p + 1)) * NB;  // the rest of the group
      if (right > 0)
        GPRC_TRY(launch_gemm_nt(s, vt + (p + 1) * NB * ldv, ldv, vt + p * NB * ldv, ldv, pan + NB, ld, mp, right, NB, 0, PK_SOLVE_UPDATE));
    }
  }
  return 0;
}

void free_model(gprc_model* m) {
  if (!m) return;
  if (m->ctx) (void)hipSetDevice(m->ctx->device);
  if (m->borrowed) m->X = m->y = m->packed = m->winv = m->alpha = nullptr;
  const int64_t n_pad = m->n_pad;
  const std::pair<double*, int64_t> parts[] = {{m->X, m->d * m->n}, {m->y, n_pad}, {m->packed, gprc_packed_size(n_pad)},
                                               {m->winv, gprc_winv_size(n_pad)}, {m->alpha, n_pad}, {m->f_hat, n_pad},
                                               {m->sw, n_pad}, {m->work, gprc_trsv_work_size(n_pad)}};
  for (const auto& pr : parts)
    if (pr.first) pool_release(m->ctx, pr.first, sizeof(double) * (size_t)pr.second);
  delete m;
}

int alloc_model(gprc_ctx* ctx, int type, const KernelSpec& ks, int64_t n, int64_t d, gprc_model** out) {
  gprc_model* m = new (std::nothrow) gprc_model();
  if (!m) { set_error("out of host memory"); return GPRC_ERR_NOMEM; }
  m->ctx = ctx; m->ctx_id = ctx->id; m->type = type; m->ks = ks; m->n = n; m->d = d; m->n_pad = pad_up(n, NB);
  const int64_t n_pad = m->n_pad;
  int rc = 0;
  auto A = [&](double** p, int64_t cnt) {
    if (rc) return;
    rc = pool_alloc(ctx, sizeof(double) * (size_t)cnt, (void**)p);
  };
  A(&m->X, d * n);
  A(&m->y, n_pad);
  A(&m->packed, gprc_packed_size(n_pad));
  A(&m->winv, gprc_winv_size(n_pad));
  A(&m->alpha, n_pad);
  A(&m->work, gprc_trsv_work_size(n_pad));
  if (type == MODEL_GPC) { A(&m->f_hat, n_pad); A(&m->sw, n_pad); }
  if (rc) { free_model(m); return rc; }
  *out = m;
  return 0;
}

int check_fit_args(gprc_ctx* ctx, const double* X, int64_t d, int64_t n, const double* y, gprc_model** out) {
  if (!ctx || !X || !y || !out || d < 1 || n < 1) { set_error("fit: bad arguments"); return GPRC_ERR_ARG; }
  return 0;
}

// one attempt; model must already hold X and y.  *info_out = LAPACK info.
int gpr_attempt(gprc_model* m, double noise, int* info_out) {
  gprc_ctx* ctx = m->ctx;
  hipStream_t s = ctx->stream;
  const int64_t n = m->n, n_pad = m->n_pad, P = n_pad / NB;
  auto fill = [&]() -> int {
    for (int64_t p = 0; p < P; ++p)
      GPRC_TRY(launch_fill(s, m->ks, m->X, n, m->X, n, m->d, m->packed + panel_offset(n_pad, p), panel_ld(n_pad, p), p * NB,
                           n_pad - p * NB, p * NB, NB, PAD_IDENTITY, noise));
    return 0;
  };
  GPRC_TRY(fill());
  DevMem inv;   // explicit inverses of the diagonal blocks: needed by the two vector solves only
  GPRC_TRY(inv.alloc(gprc_solve_inv_size(n_pad)));
  GPRC_TRY(factor_all_or_refill(ctx, m->packed, n_pad, m->winv, info_out, inv.p, fill));
  if (*info_out != 0) return 0;
  GPRC_HIP(hipMemcpyAsync(m->alpha, m->y, sizeof(double) * n_pad, hipMemcpyDeviceToDevice, s));
  GPRC_TRY(launch_trsv(s, m->packed, inv.p, n_pad, m->alpha, 0, m->work));
  GPRC_TRY(launch_trsv(s, m->packed, inv.p, n_pad, m->alpha, 1, m->work));
  GPRC_TRY(launch_logp(s, m->packed, n_pad, n, m->y, m->alpha, ctx->scal_dev));
  GPRC_HIP(hipMemcpyAsync(&m->logp, ctx->scal_dev, sizeof(double), hipMemcpyDeviceToHost, s));
  GPRC_HIP(hipStreamSynchronize(s));
  m->noise = noise;
  return 0;
}

int gpr_prepare(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d, int64_t n,
                const double* y, double noise, gprc_model** mout) {
  GPRC_TRY(check_fit_args(ctx, X, d, n, y, mout));
  if (!(noise >= 0.0)) { set_error("noise must be >= 0"); return GPRC_ERR_ARG; }  // R/GPRclass.R:130
  GPRC_TRY(use_device(ctx));
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params, n_params, d, &ks));
  gprc_model* m = nullptr;
  GPRC_TRY(alloc_model(ctx, MODEL_GPR, ks, n, d, &m));
  hipStream_t s = ctx->stream;
  hipError_t e = hipMemcpyAsync(m->X, X, sizeof(double) * d * n, hipMemcpyDefault, s);
  if (e == hipSuccess) e = hipMemsetAsync(m->y, 0, sizeof(double) * m->n_pad, s);
  if (e == hipSuccess) e = hipMemcpyAsync(m->y, y, sizeof(double) * n, hipMemcpyDefault, s);
  if (e != hipSuccess) { free_model(m); return hip_fail(e, "copy X,y", __FILE__, __LINE__); }
  *mout = m;
  return 0;
}

// eigen(A, symmetric = TRUE) on the device: cyclic Jacobi (kernels_eig.hip).  A_dev: m x m, lower triangle read.
// On return V_dev (m x m) holds the eigenvectors in Jacobi order, `values` the matching eigenvalues and `perm` the
// column order that makes them decreasing (R's convention).
int sym_eigen_dev(gprc_ctx* ctx, const double* A_dev, int64_t lda, int64_t m, double* V_dev, std::vector<double>& values,
                  std::vector<int>& perm, int* sweeps_out) {
  if (m < 1 || m > 16384) { set_error("eigen: m must be in [1, 16384]"); return GPRC_ERR_ARG; }
  hipStream_t s = ctx->stream;
  DevMem W, cs, od;
  GPRC_TRY(W.alloc(m * m));
  GPRC_TRY(cs.alloc(m + 2));
  GPRC_TRY(od.alloc(2 * m));
  GPRC_TRY(launch_sym_copy(s, A_dev, lda, m, W.p, V_dev));
  std::vector<double> h(2 * m);
  int sweeps = 0;
  for (;; ++sweeps) {
    GPRC_TRY(launch_jacobi_offnorm(s, W.p, (int)m, od.p, od.p + m));
    GPRC_HIP(hipMemcpyAsync(h.data(), od.p, sizeof(double) * 2 * m, hipMemcpyDeviceToHost, s));
    GPRC_HIP(hipStreamSynchronize(s));
    long double off2 = 0.0L, dg2 = 0.0L;
    bool finite = true;
    for (int64_t j = 0; j < m; ++j) { off2 += h[j]; dg2 += (long double)h[m + j] * h[m + j]; finite = finite && std::isfinite(h[j]) && std::isfinite(h[m + j]); }
    if (!finite) { set_error("eigen: matrix has non-finite entries"); return GPRC_ERR_ARG; }
    // ||off||_F <= max(1e-15, m eps) ||A||_F: below m*eps the off-diagonal part is rounding noise of the rotations
    // themselves (a rank-deficient covariance keeps ~m^2 such entries alive in its null space) and never shrinks
    const long double rel = std::max(1e-15L, (long double)m * 2.220446049250313e-16L);
    if (off2 <= rel * rel * (off2 + dg2) || sweeps >= 40) break;
    if (m > 1) GPRC_TRY(launch_jacobi_sweep(s, W.p, V_dev, (int)m, cs.p));
  }
  values.assign(h.begin() + m, h.end());
  perm.resize(m);
  for (int64_t j = 0; j < m; ++j) perm[j] = (int)j;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return values[a] > values[b]; });
  if (sweeps_out) *sweeps_out = sweeps;
  return 0;
}

// t(chol(cov)) or, when that fails, eigen$vectors %*% diag(sqrt(pmax(eigen$values, 0)))  (R/GPRclass.R:362-368).
// L_dev: m x m (ld m).  *method: 1 Cholesky (L lower triangular), 2 eigen.
int mvn_factor_dev(gprc_ctx* ctx, const double* cov_dev, int64_t ld, int64_t m, double tol, double* L_dev, int* method) {
  hipStream_t s = ctx->stream;
  const int64_t n_pad = pad_up(m, NB);
  {
    DevMem packed, winv;
    GPRC_TRY(packed.alloc(gprc_packed_size(n_pad)));
    GPRC_TRY(winv.alloc(gprc_winv_size(n_pad)));
    auto pack = [&]() -> int { return launch_pack_dense(s, cov_dev, ld, m, n_pad, packed.p); };
    GPRC_TRY(pack());
    int info = 0;
    GPRC_TRY(factor_all_or_refill(ctx, packed.p, n_pad, winv.p, &info, nullptr, pack));
    if (info == 0) {
      GPRC_TRY(launch_unpack_L(s, packed.p, n_pad, m, L_dev, m));
      GPRC_HIP(hipStreamSynchronize(s));
      *method = 1;
      return 0;
    }
  }
  DevMem V, scale;
  struct IntMem { int* p = nullptr; ~IntMem() { if (p) (void)hipFree(p); } } permd;
  GPRC_TRY(V.alloc(m * m));
  GPRC_TRY(scale.alloc(m));
  GPRC_HIP(hipMalloc(&permd.p, sizeof(int) * (size_t)m));
  std::vector<double> values;
  std::vector<int> perm;
  GPRC_TRY(sym_eigen_dev(ctx, cov_dev, ld, m, V.p, values, perm, nullptr));
  std::vector<double> sc(m);
  const double lead = std::fabs(values[perm[0]]);
  for (int64_t k = 0; k < m; ++k) {
    const double ev = values[perm[k]];
    if (!(ev > -tol * lead)) {  // stopifnot(all(eigval > -tol * abs(eigval[1])))  :366
      set_error("multivariate_normal: covariance is not positive semi-definite (eigenvalue " + std::to_string(ev) + ")");
      return GPRC_ERR_NOT_PD;
    }
    sc[k] = std::sqrt(ev > 0.0 ? ev : 0.0);
  }
  GPRC_HIP(hipMemcpyAsync(scale.p, sc.data(), sizeof(double) * m, hipMemcpyHostToDevice, s));
  GPRC_HIP(hipMemcpyAsync(permd.p, perm.data(), sizeof(int) * m, hipMemcpyHostToDevice, s));
  GPRC_TRY(launch_gather_scale_cols(s, V.p, (int)m, permd.p, scale.p, L_dev, m));
  GPRC_HIP(hipStreamSynchronize(s));
  *method = 2;
  return 0;
}

// doubles of partial-sum workspace per chunk row: the fill's K*^T w partials (one per 64 columns) + the solve's
// sums of squares (one per 128 columns)
int64_t predict_partials(int64_t n_pad) { return fill_mean_tiles(n_pad) + n_pad / NBI; }

// One chunk of the pointwise predict, everything fused ("Predict epilogues"):
//   fill K*^T chunk   + per-tile partials of K*^T w            (w = alpha; GPC: g, with the stored columns scaled by sqrt(W))
//   vt := vt L^-T     + per-block sums of squares in the panel solves
//   tail              mean = sum of the fill partials; var = k(x*,x*) - sum of the block sums     (R/GPRclass.R:161,164)
// The chunk is written once by the fill and read/written only by the solve: the two row-reduction passes over it are gone.
int predict_chunk(gprc_model* m, const double* xc, int64_t mcur, double* vt, int64_t ldv, double* part, double* kss_c,
                  const double* colscale, double* mean_out, double* var_out) {
  gprc_ctx* ctx = m->ctx;
  hipStream_t s = ctx->stream;
  const int64_t n = m->n, n_pad = m->n_pad, d = m->d, m_pad = pad_up(mcur, 128);
  const int64_t mt = fill_mean_tiles(n_pad);
  double* mpart = part;
  double* sspart = part + mt * m_pad;
  GPRC_TRY(launch_fill_cross_fused(s, m->ks, xc, mcur, m->X, n, d, vt, ldv, m_pad, n_pad, m->alpha, mpart, colscale));  // :160-161
  GPRC_TRY(solve_rows(ctx, m->packed, m->winv, n_pad, vt, ldv, m_pad, var_out ? sspart : nullptr));                       // :162
  GPRC_TRY(launch_sum_partials(s, mpart, mt, m_pad, mcur, nullptr, mean_out));
  if (var_out) {
    GPRC_TRY(launch_colwise(s, m->ks, xc, xc, d, mcur, kss_c));                                                           // k(X*,X*)  :164
    GPRC_TRY(launch_sum_partials(s, sspart, n_pad / NBI, m_pad, mcur, kss_c, var_out));
  }
  return 0;
}

int chunk_rows(const gprc_ctx* ctx, int64_t n_pad, int64_t ns) {
  int64_t rows = ((int64_t)(ctx->chunk_bytes / (sizeof(double) * (size_t)n_pad)) - ctx->vt_pad) / 256 * 256;
  if (rows < 256) rows = 256;
  const int64_t need = pad_up(ns, 128);
  return (int)(rows < need ? rows : need);
}

// The workspaces of a chunked pass over test points (K*^T chunk, its partial sums, k(x*, x*)): `rows` rows per chunk.
// The budget (GPRC_CHUNK_BYTES, 40 GiB) is only a wish: the chunk is sized to what the device can actually give --
// hipMemGetInfo's free figure plus what the context's slots already hold, less a reserve -- and if an allocation still fails
// (another process took the memory in between; a fragmented heap) the chunk is HALVED and tried again, down to 256 rows, before
// the call fails with GPRC_ERR_NOMEM.  Results do not depend on the chunking, bit for bit (a row's arithmetic depends on columns
// only: test_chunked_predict_is_bitwise_chunk_invariant), so shrinking is free.  want_tmp: also slot 2 (`rows` doubles).
int chunk_workspace(gprc_ctx* ctx, int64_t n_pad, int64_t ns, bool want_tmp, int64_t* rows_out, double** vt, double** part, double** tmp) {
  int64_t rows = chunk_rows(ctx, n_pad, ns);
  const int64_t per_row = (n_pad + predict_partials(n_pad) + 1) * (int64_t)sizeof(double);
  static const bool ignore_meminfo = std::getenv("GPRC_IGNORE_MEMINFO") != nullptr;   // test hook: exercise the retry path itself
  size_t free_b = 0, total_b = 0;
  if (!ignore_meminfo && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    int64_t have = 0;
    for (int i = 0; i < 3; ++i) have += ctx->ws_cap[i] * (int64_t)sizeof(double);
    const int64_t avail = (int64_t)free_b + have - ((int64_t)256 << 20);   // 256 MiB stay free: the pool's small blocks, the runtime
    const int64_t fit = (avail / per_row - ctx->vt_pad) / 256 * 256;
    if (fit < rows) rows = fit < 256 ? 256 : fit;
  } else {
    (void)hipGetLastError();
  }
  for (;;) {
    const int64_t ldv = rows + ctx->vt_pad;
    int rc = ws_get(ctx, 0, ldv * n_pad, vt);
    if (rc == 0) rc = ws_get(ctx, 1, rows * predict_partials(n_pad), part);
    if (rc == 0 && want_tmp) rc = ws_get(ctx, 2, rows, tmp);
    if (rc == 0) break;
    if (rc != GPRC_ERR_NOMEM || rows <= 256) return rc;
    pool_trim(ctx);                                    // cached blocks of earlier calls go back first
    rows = std::max<int64_t>(256, rows / 2 / 256 * 256);
  }
  *rows_out = rows;
  return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int gprc_abi_version(void) { return GPRC_ABI_VERSION; }
const char* gprc_last_error(void) { return g_last_error.c_str(); }

int gprc_device_count(int* count_out) {
  if (!count_out) { set_error("null argument"); return GPRC_ERR_ARG; }
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
  *count_out = c;
  return 0;
}

int gprc_ctx_create(int device, void* stream, gprc_ctx** ctx_out) {
  if (!ctx_out) { set_error("null argument"); return GPRC_ERR_ARG; }
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible: the gprc native path needs an MI355X (gfx950); there is no CPU fallback");
    return GPRC_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= c) { set_error("device index out of range"); return GPRC_ERR_ARG; }
  GPRC_HIP(hipSetDevice(device));
  gprc_ctx* ctx = new (std::nothrow) gprc_ctx();
  if (!ctx) { set_error("out of host memory"); return GPRC_ERR_NOMEM; }
  ctx->device = device;
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    ctx->own_stream = true;
  }
  hipError_t e = hipMalloc(&ctx->info_dev, 64);
  if (e == hipSuccess) e = hipMalloc(&ctx->scal_dev, 64);
  if (e == hipSuccess) e = hipMalloc(&ctx->sync_dev, 512);
  if (e == hipSuccess) e = hipMemset(ctx->sync_dev, 0, 512);
  if (e != hipSuccess) { gprc_ctx_destroy(ctx); return hip_fail(e, "hipMalloc(ctx)", __FILE__, __LINE__); }
  if (const char* vp = std::getenv("GPRC_VT_PAD")) {
    const long long v = std::atoll(vp);
    if (v >= 0 && v % 2 == 0) ctx->vt_pad = v;
  }
  if (const char* pb = std::getenv("GPRC_POOL_BYTES")) {
    const long long v = std::atoll(pb);
    if (v >= 0) ctx->pool_cap = (size_t)v;
  }
  if (const char* cb = std::getenv("GPRC_CHUNK_BYTES")) {
    const long long v = std::atoll(cb);
    if (v > 0) ctx->chunk_bytes = (size_t)v;
  }
  ctx_register(ctx);
  *ctx_out = ctx;
  return 0;
}

int gprc_ctx_destroy(gprc_ctx* ctx) {
  if (!ctx) return 0;
  ctx_unregister(ctx);
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 4; ++i)
    if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
  pool_trim(ctx);
  if (g_cur_ctx == ctx) g_cur_ctx = nullptr;
  if (ctx->info_dev) (void)hipFree(ctx->info_dev);
  if (ctx->scal_dev) (void)hipFree(ctx->scal_dev);
  if (ctx->sync_dev) (void)hipFree(ctx->sync_dev);
  if (ctx->svc_trace) (void)hipFree(ctx->svc_trace);
  if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
  if (ctx->side_stream2) { (void)hipStreamSynchronize(ctx->side_stream2); (void)hipStreamDestroy(ctx->side_stream2); }
  for (hipEvent_t ev : ctx->ev_pool)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int gprc_ctx_trim(gprc_ctx* ctx) {
  GPRC_TRY(use_device(ctx));
  GPRC_HIP(hipStreamSynchronize(ctx->stream));
  pool_trim(ctx);
  for (int i = 0; i < 4; ++i)
    if (ctx->ws[i]) { (void)hipFree(ctx->ws[i]); ctx->ws[i] = nullptr; ctx->ws_cap[i] = 0; }
  return 0;
}
int gprc_ctx_synchronize(gprc_ctx* ctx) {
  GPRC_TRY(use_device(ctx));
  GPRC_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- L1 -----------------------------------------------------------------------------------------
int gprc_kernel_matrix(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* A, int64_t d,
                       int64_t nA, const double* B, int64_t nB, double* out, int64_t ld_out) {
  GPRC_TRY(use_device(ctx));
  if (d < 1 || nA < 0 || nB < 0 || ld_out < nA) { set_error("kernel_matrix: bad dimensions"); return GPRC_ERR_ARG; }
  if (nA == 0 || nB == 0) return 0;
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params, n_params, d, &ks));
  hipStream_t s = ctx->stream;
  In a, b;
  GPRC_TRY(a.set(s, A, d * nA));
  GPRC_TRY(b.set(s, B, d * nB));
  // device output: written in place with the caller's leading dimension; host output: a compact nA x nB staging
  // matrix copied back with the caller's pitch, so rows nA..ld_out-1 of the caller's array are never touched
  const bool dev_out = is_device_ptr(out);
  DevMem stage;
  double* dst = out;
  int64_t ld = ld_out;
  if (!dev_out) {
    GPRC_TRY(stage.alloc(nA * nB));
    dst = stage.p;
    ld = nA;
  }
  for (int64_t c0 = 0; c0 < nB; c0 += 1 << 20) {  // grid.y limit
    const int64_t nc = (nB - c0 < (1 << 20)) ? nB - c0 : (1 << 20);
    GPRC_TRY(launch_fill(s, ks, a.dev, nA, b.dev, nB, d, dst + c0 * ld, ld, 0, nA, c0, nc, PAD_NONE, 0.0));
  }
  if (!dev_out)
    GPRC_HIP(hipMemcpy2DAsync(out, sizeof(double) * ld_out, stage.p, sizeof(double) * nA, sizeof(double) * nA, nB,
                              hipMemcpyDeviceToHost, s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

int gprc_kernel_colwise(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* x,
                        const double* y, int64_t d, int64_t m, double* out) {
  GPRC_TRY(use_device(ctx));
  if (d < 1 || m < 0) { set_error("kernel_colwise: bad dimensions"); return GPRC_ERR_ARG; }
  if (m == 0) return 0;
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params, n_params, d, &ks));
  hipStream_t s = ctx->stream;
  In a, b;
  Out o;
  GPRC_TRY(a.set(s, x, d * m));
  GPRC_TRY(b.set(s, y, d * m));
  GPRC_TRY(o.set(out, m));
  GPRC_TRY(launch_colwise(s, ks, a.dev, b.dev, d, m, o.dev));
  GPRC_TRY(o.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

// ---- GPR ----------------------------------------------------------------------------------------
int gprc_gpr_fit(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                 int64_t n, const double* y, double noise, gprc_model** model_out) {
  gprc_model* m = nullptr;
  GPRC_TRY(gpr_prepare(ctx, kernel, params, n_params, X, d, n, y, noise, &m));
  int info = 0;
  int rc = gpr_attempt(m, noise, &info);
  if (rc != 0 || info != 0) {
    free_model(m);
    if (rc == 0) set_error("the leading minor of order " + std::to_string(info) + " is not positive definite");
    return rc != 0 ? rc : info;
  }
  *model_out = m;
  return 0;
}

int gprc_gpr_log_marginal(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                          int64_t n, const double* y, double noise, double* logp_out) {
  if (!logp_out) { set_error("log_marginal: null output"); return GPRC_ERR_ARG; }
  gprc_model* m = nullptr;
  GPRC_TRY(gpr_prepare(ctx, kernel, params, n_params, X, d, n, y, noise, &m));
  int info = 0;
  const int rc = gpr_attempt(m, noise, &info);
  if (rc == 0 && info == 0) *logp_out = m->logp;
  free_model(m);
  if (rc != 0) return rc;
  if (info != 0) { set_error("the leading minor of order " + std::to_string(info) + " is not positive definite"); return info; }
  return 0;
}

// dens_deriv(v) of R/fit.R:126-139, quirks included: K is the NOISE-FREE kernel matrix, alpha = K^-1 y, and
//   grad_i = 0.5 * sum( diag(alpha alpha^T - K^-1) %*% dK/dv_i )  =  0.5 * sum_r (alpha_r^2 - (K^-1)_rr) * rowsum_r(dK/dv_i)
// (a vector-matrix product where a trace is meant; `deriv` binds v positionally in its own argument order).
// The reference inverts K with solve() (LU); here K = L L^T (K must be numerically positive definite, else info > 0,
// which the host treats like solve()'s "computationally singular" error) and diag(K^-1)_r = sum_k (L^-1)_kr^2.
int gprc_fit_gradient(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d, int64_t n,
                      const double* y, double* grad_out) {
  if (!grad_out) { set_error("fit_gradient: null output"); return GPRC_ERR_ARG; }
  if (kernel != GPRC_SQREXP && kernel != GPRC_GAMMAEXP && kernel != GPRC_POLYNOMIAL && kernel != GPRC_RATQUAD) {
    set_error("fit_gradient: defined for sqrexp, gammaexp, polynomial, rationalquadratic (R/fit.R:125)");
    return GPRC_ERR_ARG;
  }
  gprc_model* m = nullptr;
  GPRC_TRY(gpr_prepare(ctx, kernel, params, n_params, X, d, n, y, 0.0, &m));
  struct Guard { gprc_model* m; ~Guard() { free_model(m); } } guard{m};
  int info = 0;
  GPRC_TRY(gpr_attempt(m, 0.0, &info));  // L, alpha = K^-1 y
  if (info != 0) { set_error("fit_gradient: K is not positive definite (leading minor " + std::to_string(info) + ")"); return info; }
  hipStream_t s = ctx->stream;
  const int64_t n_pad = m->n_pad;
  const int n_deriv = n_params;  // 1 (sqrexp) or 2
  DevMem kinv, S;
  GPRC_TRY(kinv.alloc(n_pad));
  GPRC_TRY(S.alloc(2 * n));
  // diag(K^-1): rows of L^-T, chunk by chunk
  int64_t rows = 0;
  double *vt = nullptr, *red = nullptr, *unused = nullptr;
  GPRC_TRY(chunk_workspace(ctx, n_pad, n, false, &rows, &vt, &red, &unused));
  for (int64_t s0 = 0; s0 < n; s0 += rows) {
    const int64_t mcur = std::min<int64_t>(rows, n - s0), m_pad = pad_up(mcur, 128);
    GPRC_TRY(launch_set_identity_rows(s, vt, m_pad, m_pad, n_pad, s0));
    // rows s0.. of the identity: the triangular form of the solve (n^3 / 3 over all chunks; GPRC_FITGRAD_DENSE=1: the dense n^3 form)
    static const bool dense = std::getenv("GPRC_FITGRAD_DENSE") != nullptr;
    if (!dense) GPRC_HIP(hipMemsetAsync(red, 0, sizeof(double) * (size_t)(m_pad * (n_pad / NBI)), s));
    GPRC_TRY(solve_rows(ctx, m->packed, m->winv, n_pad, vt, m_pad, m_pad, red, dense ? -1 : s0));
    GPRC_TRY(launch_sum_partials(s, red, n_pad / NBI, m_pad, m_pad, nullptr, kinv.p + s0));  // writes m_pad entries: kinv has n_pad
  }
  GPRC_TRY(launch_deriv_rowsum(s, kernel, params[0], n_params > 1 ? params[1] : 0.0, m->X, d, n, S.p));
  std::vector<double> ha(n), hk(n), hs(2 * n);
  GPRC_HIP(hipMemcpyAsync(ha.data(), m->alpha, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  GPRC_HIP(hipMemcpyAsync(hk.data(), kinv.p, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  GPRC_HIP(hipMemcpyAsync(hs.data(), S.p, sizeof(double) * n_deriv * n, hipMemcpyDeviceToHost, s));
  GPRC_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < n_deriv; ++i) {
    long double acc = 0.0L;
    for (int64_t r = 0; r < n; ++r) acc += (long double)(ha[r] * ha[r] - hk[r]) * (long double)hs[(int64_t)i * n + r];
    grad_out[i] = (double)(0.5L * acc);
  }
  return 0;
}

int gprc_gpr_fit_retry(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                       int64_t n, const double* y, double noise, gprc_model** model_out, double* noise_used,
                       int* attempts) {
  gprc_model* m = nullptr;
  GPRC_TRY(gpr_prepare(ctx, kernel, params, n_params, X, d, n, y, noise, &m));
  double new_noise = noise;
  for (int i = 1; i <= 10; ++i) {  // R/GPRclass.R:141-148
    int info = 0;
    int rc = gpr_attempt(m, new_noise, &info);
    if (rc != 0) { free_model(m); return rc; }
    if (info == 0) {
      if (noise_used) *noise_used = new_noise;
      if (attempts) *attempts = i;
      *model_out = m;
      return 0;
    }
    new_noise = 0.01 * i + noise;
  }
  free_model(m);
  if (attempts) *attempts = 10;
  set_error("Inputs lead to non positive definite covariance matrix. Try using a larger noise or a smaller lengthscale.");
  return GPRC_ERR_NOT_PD;
}

int gprc_gpr_predict(gprc_model* m, const double* X_star, int64_t ns, int pointwise, double* mean_out, double* var_out) {
  if (!m || m->type != MODEL_GPR) { set_error("predict: not a GPR model"); return GPRC_ERR_ARG; }
  if (ns < 0 || (ns > 0 && (!X_star || !mean_out || !var_out))) { set_error("predict: bad arguments"); return GPRC_ERR_ARG; }
  if (ns == 0) return 0;
  gprc_ctx* ctx = m->ctx;
  GPRC_TRY(use_device(ctx));
  hipStream_t s = ctx->stream;
  const int64_t n_pad = m->n_pad, d = m->d;
  In xs;
  Out mean, var;
  GPRC_TRY(xs.set(s, X_star, d * ns));
  GPRC_TRY(mean.set(mean_out, ns));
  GPRC_TRY(var.set(var_out, pointwise ? ns : ns * ns));

  int64_t rows = pad_up(ns, 128);
  struct { double* p; } vt, part, tmp;
  if (pointwise) {
    GPRC_TRY(chunk_workspace(ctx, n_pad, ns, true, &rows, &vt.p, &part.p, &tmp.p));
  } else {   // the full covariance needs all of v at once: no chunking to fall back on
    GPRC_TRY(ws_get(ctx, 0, (rows + ctx->vt_pad) * n_pad, &vt.p));
    GPRC_TRY(ws_get(ctx, 1, rows * predict_partials(n_pad), &part.p));
    GPRC_TRY(ws_get(ctx, 2, rows, &tmp.p));
  }
  const int64_t ldv = rows + ctx->vt_pad;  // one leading dimension for every chunk
  double* kss_c = tmp.p;

  if (pointwise) {
    for (int64_t s0 = 0; s0 < ns; s0 += rows) {
      const int64_t mcur = (ns - s0 < rows) ? ns - s0 : rows;
      GPRC_TRY(predict_chunk(m, xs.dev + s0 * d, mcur, vt.p, ldv, part.p, kss_c, nullptr, mean.dev + s0, var.dev + s0));
    }
  } else {
    const int64_t m_pad = rows;
    struct { double* p; } cov;
    GPRC_TRY(ws_get(ctx, 3, m_pad * m_pad, &cov.p));
    GPRC_TRY(predict_chunk(m, xs.dev, ns, vt.p, ldv, part.p, kss_c, nullptr, mean.dev, nullptr));
    GPRC_TRY(launch_fill(s, m->ks, xs.dev, ns, xs.dev, ns, d, cov.p, m_pad, 0, m_pad, 0, m_pad, PAD_ZERO, 0.0));  // :167
    GPRC_TRY(launch_gemm_nt(s, cov.p, m_pad, vt.p, ldv, vt.p, ldv, m_pad, m_pad, n_pad, 0, PK_COV_SYRK));                  // - t(v) %*% v
    GPRC_HIP(hipMemcpy2DAsync(var.dev, sizeof(double) * ns, cov.p, sizeof(double) * m_pad, sizeof(double) * ns, ns,
                              hipMemcpyDeviceToDevice, s));
    GPRC_HIP(hipStreamSynchronize(s));  // cov goes out of scope
  }
  GPRC_TRY(mean.finish(s));
  GPRC_TRY(var.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

int gprc_model_dims(const gprc_model* m, int64_t* n_out, int64_t* d_out) {
  if (!m) { set_error("null model"); return GPRC_ERR_ARG; }
  if (n_out) *n_out = m->n;
  if (d_out) *d_out = m->d;
  return 0;
}

int gprc_model_get_L(gprc_model* m, double* L_out, int64_t ld_out) {
  if (!m || !L_out || ld_out < m->n) { set_error("get_L: bad arguments"); return GPRC_ERR_ARG; }
  GPRC_TRY(use_device(m->ctx));
  hipStream_t s = m->ctx->stream;
  const int64_t n = m->n;
  if (is_device_ptr(L_out)) {
    GPRC_TRY(launch_unpack_L(s, m->packed, m->n_pad, n, L_out, ld_out));
  } else {  // compact staging + pitched copy: only the n x n block of the caller's array is written
    DevMem stage;
    GPRC_TRY(stage.alloc(n * n));
    GPRC_TRY(launch_unpack_L(s, m->packed, m->n_pad, n, stage.p, n));
    GPRC_HIP(hipMemcpy2DAsync(L_out, sizeof(double) * ld_out, stage.p, sizeof(double) * n, sizeof(double) * n, n,
                              hipMemcpyDeviceToHost, s));
    GPRC_HIP(hipStreamSynchronize(s));
    return 0;
  }
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

static int copy_out_vec(gprc_model* m, const double* src, double* dst, int64_t cnt) {
  if (!m || !dst || !src) { set_error("bad arguments"); return GPRC_ERR_ARG; }
  GPRC_TRY(use_device(m->ctx));
  GPRC_HIP(hipMemcpyAsync(dst, src, sizeof(double) * cnt, hipMemcpyDefault, m->ctx->stream));
  GPRC_HIP(hipStreamSynchronize(m->ctx->stream));
  return 0;
}

int gprc_gpr_get_alpha(gprc_model* m, double* alpha_out) {
  if (!m || m->type != MODEL_GPR) { set_error("not a GPR model"); return GPRC_ERR_ARG; }
  return copy_out_vec(m, m->alpha, alpha_out, m->n);
}
int gprc_gpr_get_logp(gprc_model* m, double* logp_out) {
  if (!m || m->type != MODEL_GPR || !logp_out) { set_error("not a GPR model"); return GPRC_ERR_ARG; }
  *logp_out = m->logp;
  return 0;
}
int gprc_gpr_get_noise(gprc_model* m, double* noise_out) {
  if (!m || m->type != MODEL_GPR || !noise_out) { set_error("not a GPR model"); return GPRC_ERR_ARG; }
  *noise_out = m->noise;
  return 0;
}
int gprc_model_free(gprc_model* m) {
  if (!m) return 0;
  if (m->ctx && !ctx_alive(m->ctx, m->ctx_id)) {   // the context went first: its stream is gone (and was synchronised), its pool too
    m->ctx = nullptr;                   // -> the buffers go straight back to the driver
    (void)hipDeviceSynchronize();
  }
  if (m->ctx && m->ctx->stream) (void)hipStreamSynchronize(m->ctx->stream);
  free_model(m);
  return 0;
}

// ---- GPC ----------------------------------------------------------------------------------------
int gprc_gpc_fit(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X, int64_t d,
                 int64_t n, const double* y, double epsilon, int max_iter, int flags, gprc_model** model_out,
                 int* iters_out) {
  GPRC_TRY(check_fit_args(ctx, X, d, n, y, model_out));
  if (!(epsilon > 0.0)) { set_error("epsilon must be > 0"); return GPRC_ERR_ARG; }  // R/GPCclass.R:68
  if (max_iter <= 0) max_iter = 1000;
  GPRC_TRY(use_device(ctx));
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params, n_params, d, &ks));
  gprc_model* m = nullptr;
  GPRC_TRY(alloc_model(ctx, MODEL_GPC, ks, n, d, &m));
  hipStream_t s = ctx->stream;
  const int64_t n_pad = m->n_pad;
  DevMem Kf, vec, red;
  int rc = 0;
#define GPC_TRY(call) do { rc = (call); if (rc != 0) { free_model(m); return rc; } } while (0)
#define GPC_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { free_model(m); return hip_fail(e__, #call, __FILE__, __LINE__); } } while (0)
  GPC_TRY(Kf.alloc(n_pad * n_pad));
  GPC_TRY(vec.alloc(4 * n_pad));
  GPC_TRY(red.alloc(n_pad * rowreduce_splits(n_pad)));
  double *b = vec.p, *t = vec.p + n_pad, *a = vec.p + 2 * n_pad, *f = m->f_hat;
  GPC_HIP(hipMemcpyAsync(m->X, X, sizeof(double) * d * n, hipMemcpyDefault, s));
  GPC_HIP(hipMemsetAsync(m->y, 0, sizeof(double) * n_pad, s));
  GPC_HIP(hipMemcpyAsync(m->y, y, sizeof(double) * n, hipMemcpyDefault, s));
  GPC_HIP(hipMemsetAsync(f, 0, sizeof(double) * n_pad, s));  // f <- rep(0, n)  R/GPCclass.R:74
  GPC_HIP(hipMemsetAsync(vec.p, 0, sizeof(double) * 4 * n_pad, s));
  for (int64_t c0 = 0; c0 < n_pad; c0 += 32768) {  // K <- covariance_matrix(X, X, k)  :73 (dense, zero padded)
    const int64_t nc = (n_pad - c0 < 32768) ? n_pad - c0 : 32768;
    GPC_TRY(launch_fill(s, ks, m->X, n, m->X, n, d, Kf.p + c0 * n_pad, n_pad, 0, n_pad, c0, nc, PAD_ZERO, 0.0));
  }
  DevMem inv;   // explicit inverses of B's diagonal blocks, for the two vector solves of an iteration
  GPC_TRY(inv.alloc(gprc_solve_inv_size(n_pad)));
  int it = 0;
  double objective = 0.0, last_objective = 0.0, least_objective = 0.0;
  int status = 0;
  for (;;) {
    ++it;
    GPC_TRY(launch_gpc_pre(s, f, m->y, n, m->sw, b));                       // :78-81
    auto build_B = [&]() -> int { return launch_gpc_build_B(s, Kf.p, n_pad, m->sw, m->packed); };
    GPC_TRY(build_B());                                                      // :80
    int info = 0;
    GPC_TRY(factor_all_or_refill(ctx, m->packed, n_pad, m->winv, &info, inv.p, build_B));
    if (info != 0) { free_model(m); set_error("GPC: I + sqrt(W) K sqrt(W) not positive definite"); return info; }
    GPC_TRY(launch_row_reduce(s, Kf.p, n_pad, n_pad, n_pad, b, t, red.p));  // K %*% b
    GPC_TRY(launch_gpc_scale(s, m->sw, t, t, n_pad));                        // sqrt(W) * .
    GPC_TRY(launch_trsv(s, m->packed, inv.p, n_pad, t, 0, m->work));         // :82
    GPC_TRY(launch_trsv(s, m->packed, inv.p, n_pad, t, 1, m->work));         // :83
    GPC_TRY(launch_gpc_a(s, b, m->sw, t, a, n_pad));                         // :84
    GPC_TRY(launch_row_reduce(s, Kf.p, n_pad, n_pad, n_pad, a, f, red.p));  // f <- K %*% a  :85
    GPC_TRY(launch_gpc_objective(s, a, f, m->y, n, ctx->scal_dev));          // :86
    GPC_HIP(hipMemcpyAsync(&objective, ctx->scal_dev, sizeof(double), hipMemcpyDeviceToHost, s));
    GPC_HIP(hipStreamSynchronize(s));
    if (it > 1) {
      if (std::fabs(objective - last_objective) < epsilon) break;                          // :88
      else if ((flags & GPRC_GPC_REFERENCE_STOP) && least_objective + 10.0 < objective) { status = GPRC_ERR_DIVERGED; break; }  // :90
    } else {
      least_objective = objective;
    }
    last_objective = objective;
    if (it >= max_iter) { status = GPRC_ERR_MAXITER; break; }
  }
  if (iters_out) *iters_out = it;
  if (status != 0) {
    free_model(m);
    set_error(status == GPRC_ERR_DIVERGED ? "Apparently does not converge." : "GPC: iteration cap reached");
    return status;
  }
  // final L from the converged f (:99-102), logq = objective - sum(diag(L)) (:103, sic)
  GPC_TRY(launch_gpc_pre(s, f, m->y, n, m->sw, b));
  auto build_B = [&]() -> int { return launch_gpc_build_B(s, Kf.p, n_pad, m->sw, m->packed); };
  GPC_TRY(build_B());
  int info = 0;
  GPC_TRY(factor_all_or_refill(ctx, m->packed, n_pad, m->winv, &info, nullptr, build_B));
  if (info != 0) { free_model(m); set_error("GPC: final factorisation failed"); return info; }
  double dsum = 0.0;
  GPC_TRY(launch_diag_sum(s, m->packed, n_pad, n, ctx->scal_dev));
  GPC_HIP(hipMemcpyAsync(&dsum, ctx->scal_dev, sizeof(double), hipMemcpyDeviceToHost, s));
  GPC_TRY(launch_gpc_grad(s, f, m->y, n, m->alpha, m->sw));  // g = (y+1)/2 - P, sw = sqrt(P(1-P)) for predict
  GPC_HIP(hipStreamSynchronize(s));
  m->logq = objective - dsum;
#undef GPC_TRY
#undef GPC_HIP
  *model_out = m;
  return 0;
}

int gprc_gpc_predict_latent(gprc_model* m, const double* X_star, int64_t ns, double* fs_bar_out, double* Vfs_out) {
  if (!m || m->type != MODEL_GPC) { set_error("predict_latent: not a GPC model"); return GPRC_ERR_ARG; }
  if (ns < 0 || (ns > 0 && (!X_star || !fs_bar_out || !Vfs_out))) { set_error("predict_latent: bad arguments"); return GPRC_ERR_ARG; }
  if (ns == 0) return 0;
  gprc_ctx* ctx = m->ctx;
  GPRC_TRY(use_device(ctx));
  hipStream_t s = ctx->stream;
  const int64_t n_pad = m->n_pad, d = m->d;
  In xs;
  Out fs, vf;
  GPRC_TRY(xs.set(s, X_star, d * ns));
  GPRC_TRY(fs.set(fs_bar_out, ns));
  GPRC_TRY(vf.set(Vfs_out, ns));
  int64_t rows = 0;
  struct { double* p; } vt, part, tmp;
  GPRC_TRY(chunk_workspace(ctx, n_pad, ns, true, &rows, &vt.p, &part.p, &tmp.p));
  const int64_t ldv = rows + ctx->vt_pad;
  for (int64_t s0 = 0; s0 < ns; s0 += rows) {   // R/GPCclass.R:112-115: m->alpha holds g = (y+1)/2 - P, the stored columns are sqrt(W) * K_star
    const int64_t mcur = (ns - s0 < rows) ? ns - s0 : rows;
    GPRC_TRY(predict_chunk(m, xs.dev + s0 * d, mcur, vt.p, ldv, part.p, tmp.p, m->sw, fs.dev + s0, vf.dev + s0));
  }
  GPRC_TRY(fs.finish(s));
  GPRC_TRY(vf.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

int gprc_class_probability(gprc_ctx* ctx, const double* fs_bar, const double* Vfs, int64_t n, double* prob_out) {
  GPRC_TRY(use_device(ctx));
  if (n < 0 || (n > 0 && (!fs_bar || !Vfs || !prob_out))) { set_error("class_probability: bad arguments"); return GPRC_ERR_ARG; }
  if (n == 0) return 0;
  hipStream_t s = ctx->stream;
  In a, b;
  Out o;
  GPRC_TRY(a.set(s, fs_bar, n));
  GPRC_TRY(b.set(s, Vfs, n));
  GPRC_TRY(o.set(prob_out, n));
  GPRC_TRY(launch_gpc_class_prob(s, a.dev, b.dev, o.dev, n));
  GPRC_TRY(o.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

int gprc_gpc_predict_class(gprc_model* m, const double* X_star, int64_t ns, double* prob_out) {
  if (!m || m->type != MODEL_GPC) { set_error("predict_class: not a GPC model"); return GPRC_ERR_ARG; }
  if (ns < 0 || (ns > 0 && (!X_star || !prob_out))) { set_error("predict_class: bad arguments"); return GPRC_ERR_ARG; }
  if (ns == 0) return 0;
  DevMem lat;
  GPRC_TRY(use_device(m->ctx));
  GPRC_TRY(lat.alloc(2 * ns));
  GPRC_TRY(gprc_gpc_predict_latent(m, X_star, ns, lat.p, lat.p + ns));   // device outputs: used in place
  return gprc_class_probability(m->ctx, lat.p, lat.p + ns, ns, prob_out);
}

int gprc_gpc_get_f_hat(gprc_model* m, double* f_hat_out) {
  if (!m || m->type != MODEL_GPC) { set_error("not a GPC model"); return GPRC_ERR_ARG; }
  return copy_out_vec(m, m->f_hat, f_hat_out, m->n);
}
int gprc_gpc_get_logq(gprc_model* m, double* logq_out) {
  if (!m || m->type != MODEL_GPC || !logq_out) { set_error("not a GPC model"); return GPRC_ERR_ARG; }
  *logq_out = m->logq;
  return 0;
}

// ---- layout helpers + device-level building blocks ------------------------------------------------
int64_t gprc_panel_width(void) { return NB; }
int64_t gprc_pad(int64_t n) { return pad_up(n, NB); }
int64_t gprc_panel_count(int64_t n_pad) { return n_pad / NB; }
int64_t gprc_panel_offset(int64_t n_pad, int64_t p) { return panel_offset(n_pad, p); }
int64_t gprc_panel_elems(int64_t n_pad, int64_t p) { return panel_ld(n_pad, p) * NB; }
int64_t gprc_packed_size(int64_t n_pad) { return panel_offset(n_pad, n_pad / NB); }
int64_t gprc_winv_size(int64_t n_pad) { return n_pad * NBI; }
int64_t gprc_trsv_work_size(int64_t n_pad) { return n_pad + n_pad / NB + 8; }   // x_p staging + one gate (8 bytes) per panel
int64_t gprc_solve_inv_size(int64_t n_pad) { return n_pad * NB; }
int64_t gprc_rowreduce_splits(int64_t cols) { return rowreduce_splits(cols); }

int gprc_dev_fill_panel(gprc_ctx* ctx, int kernel, const double* params_host, int n_params, const double* X,
                        int64_t d, int64_t n, int64_t n_pad, double noise, double* packed, int64_t p) {
  GPRC_TRY(use_device(ctx));
  if (n_pad != pad_up(n, NB) || p < 0 || p >= n_pad / NB) { set_error("fill_panel: bad layout arguments"); return GPRC_ERR_ARG; }
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params_host, n_params, d, &ks));
  return launch_fill(ctx->stream, ks, X, n, X, n, d, packed + panel_offset(n_pad, p), panel_ld(n_pad, p), p * NB, n_pad - p * NB,
                     p * NB, NB, PAD_IDENTITY, noise);
}

int gprc_dev_factor_panel(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, double* winv, int* info_dev) {
  GPRC_TRY(use_device(ctx));
  if (n_pad % NB || p < 0 || p >= n_pad / NB || !info_dev) { set_error("factor_panel: bad arguments"); return GPRC_ERR_ARG; }
  return factor_panel(ctx, packed, n_pad, p, winv, info_dev);
}

int gprc_dev_factor_subpanel(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, int j, int part, double* winv, int* info_dev) {
  GPRC_TRY(use_device(ctx));
  if (n_pad % NB || p < 0 || p >= n_pad / NB || j < 0 || j >= NB / NBI || part < 0 || part > 2 || !info_dev) {
    set_error("factor_subpanel: bad arguments");
    return GPRC_ERR_ARG;
  }
  return factor_subpanel(ctx, packed, n_pad, p, j, part, winv, info_dev);
}
int gprc_dev_factor_all(gprc_ctx* ctx, double* packed, int64_t n_pad, double* winv, int* info_dev, double* inv) {
  GPRC_TRY(use_device(ctx));
  if (!packed || !winv || !info_dev || n_pad <= 0 || n_pad % NB) { set_error("dev_factor_all: bad arguments"); return GPRC_ERR_ARG; }
  return factor_all_async(ctx, packed, n_pad, winv, info_dev, inv);
}
int gprc_factor_service(int on) {
  const int was = g_service_off.load() ? 0 : 1;
  if (on == 0) g_service_off.store(true);
  else if (on > 0) g_service_off.store(false);
  return was;
}
int gprc_dev_solve_prepare(gprc_ctx* ctx, const double* packed, const double* winv, int64_t n_pad, double* inv, int64_t p_begin, int64_t p_end) {
  GPRC_TRY(use_device(ctx));
  if (!packed || !winv || !inv || n_pad <= 0 || n_pad % NB || p_begin < 0 || p_end > n_pad / NB) { set_error("dev_solve_prepare: bad arguments"); return GPRC_ERR_ARG; }
  return launch_inv512(ctx->stream, packed, n_pad, winv, inv, p_begin, p_end);
}
int gprc_dev_update_range(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p_begin, int64_t p_end, int64_t q_begin, int64_t q_end,
                          int64_t q_stride) {
  GPRC_TRY(use_device(ctx));
  if (!packed || n_pad <= 0 || n_pad % NB || p_begin < 0 || p_end > n_pad / NB) { set_error("dev_update_range: bad arguments"); return GPRC_ERR_ARG; }
  if (p_end - p_begin == 1) return launch_trailing_update(ctx->stream, packed, n_pad, p_begin, q_begin, q_end, q_stride);  // the K = 512 kernel
  return launch_trailing_range(ctx->stream, packed, n_pad, p_begin, p_end, q_begin, q_end, q_stride);
}
int gprc_dev_update_trailing(gprc_ctx* ctx, double* packed, int64_t n_pad, int64_t p, int64_t q_begin,
                             int64_t q_end, int64_t q_stride) {
  GPRC_TRY(use_device(ctx));
  if (n_pad % NB) { set_error("update_trailing: bad n_pad"); return GPRC_ERR_ARG; }
  if (q_begin >= q_end) return 0;
  return launch_trailing_update(ctx->stream, packed, n_pad, p, q_begin, q_end, q_stride);
}

int gprc_dev_trsv(gprc_ctx* ctx, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose,
                  double* work) {
  GPRC_TRY(use_device(ctx));
  if (!packed || !inv || !b || !work || n_pad <= 0 || n_pad % NB) { set_error("dev_trsv: bad arguments"); return GPRC_ERR_ARG; }
  return launch_trsv(ctx->stream, packed, inv, n_pad, b, transpose, work);
}

int gprc_dev_trsv_step(gprc_ctx* ctx, const double* packed, const double* inv, int64_t n_pad, double* b, int transpose, int64_t p, double* work) {
  GPRC_TRY(use_device(ctx));
  if (!packed || !inv || !b || !work || n_pad <= 0 || n_pad % NB) { set_error("dev_trsv_step: bad arguments"); return GPRC_ERR_ARG; }
  return launch_trsv_step(ctx->stream, packed, inv, n_pad, b, transpose, (int)p, work);
}

int gprc_dev_fill_cross(gprc_ctx* ctx, int kernel, const double* params_host, int n_params, const double* X_star,
                        int64_t d, int64_t m, int64_t m_pad, const double* X, int64_t n, int64_t n_pad, double* vt,
                        int64_t ld) {
  GPRC_TRY(use_device(ctx));
  if (m_pad % 128 || m_pad < m || n_pad < n || ld < m_pad || ld % 2) { set_error("fill_cross: bad padding"); return GPRC_ERR_ARG; }
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params_host, n_params, d, &ks));
  return launch_fill(ctx->stream, ks, X_star, m, X, n, d, vt, ld, 0, m_pad, 0, n_pad, PAD_ZERO, 0.0);
}

int gprc_dev_row_reduce(gprc_ctx* ctx, const double* vt, int64_t ld, int64_t rows, int64_t cols, const double* w,
                        double* out, double* work) {
  GPRC_TRY(use_device(ctx));
  return launch_row_reduce(ctx->stream, vt, ld, rows, cols, w, out, work);
}

// ---- sampling -----------------------------------------------------------------------------
int gprc_sym_eigen(gprc_ctx* ctx, const double* A, int64_t lda, int64_t m, double* values_out, double* vectors_out, int* sweeps_out) {
  GPRC_TRY(use_device(ctx));
  if (!A || !values_out || m < 1 || lda < m) { set_error("sym_eigen: bad arguments"); return GPRC_ERR_ARG; }
  hipStream_t s = ctx->stream;
  In a;
  Out vals, vecs;
  GPRC_TRY(a.set(s, A, lda * m));
  GPRC_TRY(vals.set(values_out, m));
  if (vectors_out) GPRC_TRY(vecs.set(vectors_out, m * m));
  DevMem V;
  struct IntMem { int* p = nullptr; ~IntMem() { if (p) (void)hipFree(p); } } permd;
  GPRC_TRY(V.alloc(m * m));
  std::vector<double> values;
  std::vector<int> perm;
  GPRC_TRY(sym_eigen_dev(ctx, a.dev, lda, m, V.p, values, perm, sweeps_out));
  std::vector<double> sorted(m);
  for (int64_t k = 0; k < m; ++k) sorted[k] = values[perm[k]];
  GPRC_HIP(hipMemcpyAsync(vals.dev, sorted.data(), sizeof(double) * m, hipMemcpyHostToDevice, s));
  if (vectors_out) {
    GPRC_HIP(hipMalloc(&permd.p, sizeof(int) * (size_t)m));
    GPRC_HIP(hipMemcpyAsync(permd.p, perm.data(), sizeof(int) * m, hipMemcpyHostToDevice, s));
    GPRC_TRY(launch_gather_scale_cols(s, V.p, (int)m, permd.p, nullptr, vecs.dev, m));
    GPRC_TRY(vecs.finish(s));
  }
  GPRC_TRY(vals.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

int gprc_mvn_factor(gprc_ctx* ctx, const double* cov, int64_t ld, int64_t m, double tol, double* L_out, int* method_out) {
  GPRC_TRY(use_device(ctx));
  if (!cov || !L_out || m < 1 || ld < m) { set_error("mvn_factor: bad arguments"); return GPRC_ERR_ARG; }
  hipStream_t s = ctx->stream;
  In c;
  Out L;
  GPRC_TRY(c.set(s, cov, ld * m));
  GPRC_TRY(L.set(L_out, m * m));
  int method = 0;
  GPRC_TRY(mvn_factor_dev(ctx, c.dev, ld, m, tol, L.dev, &method));
  GPRC_TRY(L.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  if (method_out) *method_out = method;
  return 0;
}

int gprc_mvn_sample(gprc_ctx* ctx, const double* cov, int64_t ld, int64_t m, const double* mean, double tol, const double* Z,
                    int64_t n_draws, double* out, int* method_out) {
  GPRC_TRY(use_device(ctx));
  if (!cov || !mean || !Z || !out || m < 1 || ld < m || n_draws < 1) { set_error("mvn_sample: bad arguments"); return GPRC_ERR_ARG; }
  hipStream_t s = ctx->stream;
  In c, mu, z;
  Out o;
  GPRC_TRY(c.set(s, cov, ld * m));
  GPRC_TRY(mu.set(s, mean, m));
  GPRC_TRY(z.set(s, Z, m * n_draws));
  GPRC_TRY(o.set(out, m * n_draws));
  DevMem L;
  GPRC_TRY(L.alloc(m * m));
  int method = 0;
  GPRC_TRY(mvn_factor_dev(ctx, c.dev, ld, m, tol, L.p, &method));
  GPRC_TRY(launch_affine_lz(s, L.p, m, m, mu.dev, z.dev, m, n_draws, o.dev, m, method == 1));  // drop(mean) + L %*% Z  :369
  GPRC_TRY(o.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  if (method_out) *method_out = method;
  return 0;
}

// combine_all(lst)  --  R/simulation.R:338-349 (the test grid of the simulate_* harness, :101-102)
int gprc_combine_all(gprc_ctx* ctx, const double* axis_values, const int64_t* lengths, int d, double* out) {
  GPRC_TRY(use_device(ctx));
  if (!axis_values || !lengths || !out || d < 1 || d > 64) { set_error("combine_all: bad arguments"); return GPRC_ERR_ARG; }
  int64_t sum = 0, total = 1;
  for (int k = 0; k < d; ++k) {
    if (lengths[k] < 1) { set_error("combine_all: every axis needs at least one value"); return GPRC_ERR_ARG; }
    sum += lengths[k];
    if (total > ((int64_t)1 << 40) / lengths[k]) { set_error("combine_all: grid too large"); return GPRC_ERR_ARG; }
    total *= lengths[k];
  }
  hipStream_t s = ctx->stream;
  In v;
  Out o;
  GPRC_TRY(v.set(s, axis_values, sum));
  GPRC_TRY(o.set(out, total * d));
  GPRC_TRY(launch_combine_all(s, v.dev, lengths, d, o.dev));
  GPRC_TRY(o.finish(s));
  GPRC_HIP(hipStreamSynchronize(s));
  return 0;
}

int gprc_prof_panel_trace(gprc_ctx* ctx, int side, int64_t* ticks_out, int n) {
  GPRC_TRY(use_device(ctx));
  if (!ticks_out || n < 1 || n > 24) { set_error("prof_panel_trace: 1..24 stamps"); return GPRC_ERR_ARG; }
  GPRC_HIP(hipDeviceSynchronize());
  GPRC_HIP(hipMemcpy(ticks_out, static_cast<char*>(ctx->sync_dev) + (side ? 256 : 0) + 64, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost));
  return 0;
}

int gprc_prof_service_trace(gprc_ctx* ctx, int64_t* ticks_out, int panels) {
  GPRC_TRY(use_device(ctx));
  if (!ticks_out || panels < 1 || panels > SVC_TRACE_PANELS) { set_error("prof_service_trace: 1..48 panels"); return GPRC_ERR_ARG; }
  if (!ctx->svc_trace) { set_error("prof_service_trace: no traced sweep on this context (set GPRC_SERVICE_TRACE before the first call)"); return GPRC_ERR_ARG; }
  GPRC_HIP(hipDeviceSynchronize());
  GPRC_HIP(hipMemcpy(ticks_out, ctx->svc_trace, sizeof(int64_t) * 16 * (size_t)panels, hipMemcpyDeviceToHost));
  return 0;
}

int gprc_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}
int gprc_prof_reset(void) {
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1); }
  g_prof_recs.clear();
  return 0;
}
int gprc_prof_kinds(void) { return PK_COUNT; }
int gprc_prof_summary(int kind, int64_t* count_out, double* ms_out, double* flops_out, double* bytes_out) {
  if (kind < 0 || kind >= PK_COUNT) { set_error("prof_summary: bad kind"); return GPRC_ERR_ARG; }
  int64_t cnt = 0;
  double ms = 0.0, fl = 0.0, by = 0.0;
  for (auto& r : g_prof_recs) {
    if (r.kind != kind) continue;
    GPRC_HIP(hipEventSynchronize(r.e1));
    float t = 0.f;
    GPRC_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t; fl += r.flops; by += r.bytes; ++cnt;
  }
  if (count_out) *count_out = cnt;
  if (ms_out) *ms_out = ms;
  if (flops_out) *flops_out = fl;
  if (bytes_out) *bytes_out = by;
  return 0;
}

int gprc_dev_logp(gprc_ctx* ctx, const double* packed, int64_t n_pad, int64_t n, const double* y, const double* alpha,
                  double* out_dev) {
  GPRC_TRY(use_device(ctx));
  return launch_logp(ctx->stream, packed, n_pad, n, y, alpha, out_dev);
}

int gprc_gpr_model_from_device(gprc_ctx* ctx, int kernel, const double* params, int n_params, const double* X,
                               int64_t d, int64_t n, const double* y, double* packed, double* winv, double* alpha,
                               double noise, double logp, gprc_model** model_out) {
  GPRC_TRY(use_device(ctx));
  if (!X || !y || !packed || !winv || !alpha || !model_out || d < 1 || n < 1) { set_error("model_from_device: bad arguments"); return GPRC_ERR_ARG; }
  KernelSpec ks;
  GPRC_TRY(make_spec(kernel, params, n_params, d, &ks));
  gprc_model* m = new (std::nothrow) gprc_model();
  if (!m) { set_error("out of host memory"); return GPRC_ERR_NOMEM; }
  m->ctx = ctx; m->ctx_id = ctx->id; m->type = MODEL_GPR; m->ks = ks; m->n = n; m->d = d; m->n_pad = pad_up(n, NB);
  m->borrowed = true;
  m->X = const_cast<double*>(X); m->y = const_cast<double*>(y); m->packed = packed; m->winv = winv; m->alpha = alpha;
  m->noise = noise; m->logp = logp;
  *model_out = m;
  return 0;
}

int gprc_dev_solve_rows(gprc_ctx* ctx, const double* packed, const double* winv, int64_t n_pad, double* vt,
                        int64_t ld, int64_t m_pad) {
  GPRC_TRY(use_device(ctx));
  if (m_pad % 128 || n_pad % NB || ld < m_pad || ld % 2) { set_error("solve_rows: bad padding"); return GPRC_ERR_ARG; }
  return solve_rows(ctx, packed, winv, n_pad, vt, ld, m_pad);
}

}  // extern "C"
