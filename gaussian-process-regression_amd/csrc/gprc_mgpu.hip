// gprc_mgpu.hip -- the multi-GPU predict step driven by ONE host process through the C ABI (gprc_mgpu_*).
//
// Why it exists: the reference's host is a single R process (SURVEY 8b "Threading": `.Call` is entered on R's main thread;
// R cannot be forked per GPU), so the one-process-per-GPU driver of distributed.py cannot sit behind the R6 classes.
// Here one thread owns G "ranks" -- rank r = (device, three HIP streams, its own buffers) -- and runs the SAME sweep as
// distributed.py's DistributedGPR: 1-D block-cyclic 512-column panels, the owner factors a panel on a
// high-priority side stream while the trailing update of the previous panel is still running (look-ahead), every rank
// receives every panel (so L ends up replicated and the predict needs no exchange), alpha redundantly everywhere,
// test points sliced.  The arithmetic is the public device-level ABI of this library (gprc_dev_*): nothing here
// touches a matrix element on the host.
//
// The ONE exchange step -- "panel p from its owner to everybody":
//   * GPRC_MGPU_RCCL: ncclBroadcast on every rank's communication stream inside one ncclGroupStart/End (the
//     single-process form of the collective; communicators from ncclCommInitAll, the devices must be distinct).  RCCL is
//     resolved at run time (dlsym / dlopen of librccl.so.1): the library has no link-time dependency on it.
//   * otherwise (default; also the only choice when a device is listed more than once): the owner's panel is copied to
//     each rank by hipMemcpyPeerAsync / device-to-device hipMemcpyAsync on the RECEIVER's communication stream, ordered
//     behind the owner's "panel complete" event.  Listing one device several times gives VIRTUAL RANKS -- G ranks with
//     their own buffers and streams sharing one GPU -- which is how a one-GPU box tests the G = 2, 3 sweeps
//     (tests/c_abi_client.c), bit-identical to gprc_gpr_fit.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gprc_internal.h"

using namespace gprc;

namespace {

// ---- RCCL, resolved at run time (signatures: /opt/rocm/include/rccl/rccl.h:236,260,339,591 and the group calls) ----------
typedef void* nccl_comm_t;
struct Rccl {
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};
constexpr int NCCL_DOUBLE = 8;  // ncclFloat64 = ncclDouble = 8 (rccl.h:467)

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(RTLD_DEFAULT, name);  // a copy already in the process (PyTorch-ROCm brings its own) wins
      if (!p && h) p = dlsym(h, name);
      return p;
    };
    if (!dlsym(RTLD_DEFAULT, "ncclCommInitAll")) {
      for (const char* cand : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(cand, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    }
    x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
    x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
    x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
    x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.ok = x.CommInitAll && x.CommDestroy && x.GetErrorString && x.Broadcast && x.GroupStart && x.GroupEnd;
    return x;
  }();
  return r;
}

struct Rank {
  int device = 0;
  hipStream_t main = nullptr, side = nullptr, comm = nullptr;
  gprc_ctx *ctx_main = nullptr, *ctx_side = nullptr;
  nccl_comm_t nccl = nullptr;
  std::vector<hipEvent_t> events;  // recycled round-robin: an event is re-recorded long after its waiters were enqueued
  size_t next_event = 0;
  // per-model buffers (owned by the model, listed here while a fit is running)
};

}  // namespace

struct gprc_mgpu {
  std::vector<Rank> ranks;
  bool use_rccl = false;
  bool lookahead = true;
  int batch = 4;  // far panels receive the panels a rank has collected every `batch` steps, in one pass (gprc_dev_update_range)
};

struct gprc_mgpu_model {
  gprc_mgpu* mg = nullptr;
  int kernel = 0;
  std::vector<double> params;
  int64_t n = 0, d = 0, n_pad = 0;
  double noise = 0.0, logp = 0.0;
  struct PerRank { double *X = nullptr, *y = nullptr, *packed = nullptr, *winv = nullptr, *inv = nullptr, *alpha = nullptr, *work = nullptr, *scal = nullptr; int* info = nullptr; gprc_model* model = nullptr; };
  std::vector<PerRank> pr;
  std::vector<double> alpha_host;
};

namespace {

#define MG_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e__ = (call);                                                      \
    if (e__ != hipSuccess) return ::gprc::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

int nccl_fail(int rc, const char* what) {
  set_error(std::string("RCCL error '") + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "?") + "' in " + what);
  return GPRC_ERR_HIP;
}
#define MG_NCCL(call)                                  \
  do {                                                 \
    int r__ = (call);                                  \
    if (r__ != 0) return nccl_fail(r__, #call);        \
  } while (0)

// stream `waiter` (on rank w's device) waits for everything enqueued so far on `on` (rank o's device)
int wait_stream(Rank& w, hipStream_t waiter, Rank& o, hipStream_t on) {
  if (o.events.empty()) { set_error("mgpu: no events"); return GPRC_ERR_ARG; }
  hipEvent_t ev = o.events[o.next_event++ % o.events.size()];
  MG_HIP(hipSetDevice(o.device));
  MG_HIP(hipEventRecord(ev, on));
  MG_HIP(hipSetDevice(w.device));
  MG_HIP(hipStreamWaitEvent(waiter, ev, 0));
  return 0;
}

void free_model_buffers(gprc_mgpu_model* m) {
  if (!m) return;
  for (size_t r = 0; r < m->pr.size(); ++r) {
    auto& b = m->pr[r];
    (void)hipSetDevice(m->mg->ranks[r].device);
    if (b.model) gprc_model_free(b.model);
    for (void* p : {(void*)b.X, (void*)b.y, (void*)b.packed, (void*)b.winv, (void*)b.inv, (void*)b.alpha, (void*)b.work, (void*)b.scal, (void*)b.info})
      if (p) (void)hipFree(p);
  }
  delete m;
}

// the exchange step: panel p (+ the inverses of its four diagonal blocks) from its owner to every rank
int share_panel(gprc_mgpu* mg, gprc_mgpu_model* m, int64_t p) {
  const int G = (int)mg->ranks.size();
  if (G == 1 && !mg->use_rccl) return 0;
  const int src = (int)(p % G);
  const int64_t off = gprc_panel_offset(m->n_pad, p), cnt = gprc_panel_elems(m->n_pad, p);
  const int64_t woff = p * (NB / NBI) * NBI * NBI, wcnt = (int64_t)(NB / NBI) * NBI * NBI;
  Rank& o = mg->ranks[src];
  GPRC_TRY(wait_stream(o, o.comm, o, o.side));  // the panel is complete on the owner's side stream
  if (mg->use_rccl) {
    MG_NCCL(rccl().GroupStart());
    for (int r = 0; r < G; ++r) {
      MG_HIP(hipSetDevice(mg->ranks[r].device));
      MG_NCCL(rccl().Broadcast(m->pr[r].packed + off, m->pr[r].packed + off, (size_t)cnt, NCCL_DOUBLE, src, mg->ranks[r].nccl, mg->ranks[r].comm));
      MG_NCCL(rccl().Broadcast(m->pr[r].winv + woff, m->pr[r].winv + woff, (size_t)wcnt, NCCL_DOUBLE, src, mg->ranks[r].nccl, mg->ranks[r].comm));
    }
    MG_NCCL(rccl().GroupEnd());
    return 0;
  }
  for (int r = 0; r < G; ++r) {
    if (r == src) continue;
    Rank& w = mg->ranks[r];
    GPRC_TRY(wait_stream(w, w.comm, o, o.comm));
    MG_HIP(hipSetDevice(w.device));
    if (w.device == o.device) {
      MG_HIP(hipMemcpyAsync(m->pr[r].packed + off, m->pr[src].packed + off, sizeof(double) * cnt, hipMemcpyDeviceToDevice, w.comm));
      MG_HIP(hipMemcpyAsync(m->pr[r].winv + woff, m->pr[src].winv + woff, sizeof(double) * wcnt, hipMemcpyDeviceToDevice, w.comm));
    } else {
      MG_HIP(hipMemcpyPeerAsync(m->pr[r].packed + off, w.device, m->pr[src].packed + off, o.device, sizeof(double) * cnt, w.comm));
      MG_HIP(hipMemcpyPeerAsync(m->pr[r].winv + woff, w.device, m->pr[src].winv + woff, o.device, sizeof(double) * wcnt, w.comm));
    }
  }
  return 0;
}

int64_t owned_after(int64_t p, int rank, int G) {  // smallest panel q > p owned by `rank`
  const int64_t q = p + 1;
  return q + (((rank - q) % G) + G) % G;
}

// One Cholesky attempt on all ranks: F1 (own panels), F2 (sweep + exchange), F3 (alpha, logp).  *info_out = LAPACK info.
int mgpu_attempt(gprc_mgpu* mg, gprc_mgpu_model* m, double noise, int* info_out) {
  const int G = (int)mg->ranks.size();
  const int64_t n = m->n, d = m->d, n_pad = m->n_pad, P = n_pad / NB;
  const double* par = m->params.data();
  const int npar = (int)m->params.size();
  for (int r = 0; r < G; ++r) {
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipMemsetAsync(m->pr[r].info, 0, 16, k.main));
    for (int64_t p = r; p < P; p += G)   // F1: own panels, no exchange
      GPRC_TRY(gprc_dev_fill_panel(k.ctx_main, m->kernel, par, npar, m->pr[r].X, d, n, n_pad, noise, m->pr[r].packed, p));
  }
  if (G == 1 && !mg->use_rccl) {
    Rank& k = mg->ranks[0];
    GPRC_TRY(gprc_dev_factor_all(k.ctx_main, m->pr[0].packed, n_pad, m->pr[0].winv, m->pr[0].info, m->pr[0].inv));
  } else {
    auto factor_and_share = [&](int64_t p) -> int {
      const int src = (int)(p % G);
      Rank& o = mg->ranks[src];
      MG_HIP(hipSetDevice(o.device));
      GPRC_TRY(gprc_dev_factor_panel(o.ctx_side, m->pr[src].packed, n_pad, p, m->pr[src].winv, m->pr[src].info));
      return share_panel(mg, m, p);
    };
    for (int r = 0; r < G; ++r) GPRC_TRY(wait_stream(mg->ranks[r], mg->ranks[r].side, mg->ranks[r], mg->ranks[r].main));  // fork
    GPRC_TRY(factor_and_share(0));
    std::vector<int64_t> far_from(G, 0);  // rank r has applied panels [0, far_from[r]) to all its unfactored panels
    for (int64_t p = 0; p < P; ++p) {
      for (int r = 0; r < G; ++r) {  // panel p is factored (owner: side stream) / received (others: comm stream)
        Rank& k = mg->ranks[r];
        GPRC_TRY(wait_stream(k, k.main, k, k.side));
        GPRC_TRY(wait_stream(k, k.main, k, k.comm));
      }
      if (p + 1 >= P) break;
      const int nxt = (int)((p + 1) % G);
      std::vector<int64_t> q0(G, -1);
      for (int r = 0; r < G; ++r) {
        Rank& k = mg->ranks[r];
        MG_HIP(hipSetDevice(k.device));
        if (r == nxt && !mg->lookahead) {  // no look-ahead: bring all my panels up to date on the main stream, then factor
          GPRC_TRY(gprc_dev_update_range(k.ctx_main, m->pr[r].packed, n_pad, far_from[r], p + 1, p + 1, P, G));
          far_from[r] = p + 1;
          GPRC_TRY(wait_stream(k, k.side, k, k.main));
        } else if (r == nxt) {             // look-ahead: panel p + 1 first, on the side stream
          GPRC_TRY(wait_stream(k, k.side, k, k.main));
          GPRC_TRY(gprc_dev_update_range(k.ctx_side, m->pr[r].packed, n_pad, far_from[r], p + 1, p + 1, p + 2, 1));
          q0[r] = p + 1 + G;
        } else {
          q0[r] = owned_after(p, r, G);
        }
      }
      GPRC_TRY(factor_and_share(p + 1));  // owner: side stream; everybody: the exchange on the comm streams
      for (int r = 0; r < G; ++r) {       // the far panels, every `batch` steps in one pass: beside the factorisation and the exchange
        if (q0[r] < 0) continue;
        const bool flush = (p + 1 - far_from[r] >= mg->batch) || p + 2 >= P;
        if (!flush) continue;
        Rank& k = mg->ranks[r];
        MG_HIP(hipSetDevice(k.device));
        if (q0[r] < P) GPRC_TRY(gprc_dev_update_range(k.ctx_main, m->pr[r].packed, n_pad, far_from[r], p + 1, q0[r], P, G));
        far_from[r] = p + 1;
      }
    }
  }
  // info: the first failing column over all ranks (every rank holds its own panels' verdicts)
  int info = 0;
  for (int r = 0; r < G; ++r) {
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipStreamSynchronize(k.side));
    MG_HIP(hipStreamSynchronize(k.comm));
    int ir = 0;
    MG_HIP(hipMemcpyAsync(&ir, m->pr[r].info, sizeof(int), hipMemcpyDeviceToHost, k.main));
    MG_HIP(hipStreamSynchronize(k.main));
    if (ir < 0) { set_error("mgpu: a device-side dependency wait timed out on rank " + std::to_string(r) + " (info = " + std::to_string(ir) + ")"); return GPRC_ERR_HIP; }
    if (ir > 0 && (info == 0 || ir < info)) info = ir;
  }
  *info_out = info;
  if (info != 0) return 0;
  for (int r = 0; r < G; ++r) {  // F3, replicated: L is complete on every rank
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipMemcpyAsync(m->pr[r].alpha, m->pr[r].y, sizeof(double) * n_pad, hipMemcpyDeviceToDevice, k.main));
    // the explicit diagonal-block inverses the vector solves work with: every rank computes its own copy (never exchanged)
    if (!(G == 1 && !mg->use_rccl)) GPRC_TRY(gprc_dev_solve_prepare(k.ctx_main, m->pr[r].packed, m->pr[r].winv, n_pad, m->pr[r].inv, 0, P));
    GPRC_TRY(gprc_dev_trsv(k.ctx_main, m->pr[r].packed, m->pr[r].inv, n_pad, m->pr[r].alpha, 0, m->pr[r].work));
    GPRC_TRY(gprc_dev_trsv(k.ctx_main, m->pr[r].packed, m->pr[r].inv, n_pad, m->pr[r].alpha, 1, m->pr[r].work));
    GPRC_TRY(gprc_dev_logp(k.ctx_main, m->pr[r].packed, n_pad, n, m->pr[r].y, m->pr[r].alpha, m->pr[r].scal));
  }
  m->alpha_host.resize((size_t)n);
  for (int r = 0; r < G; ++r) {
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    if (r == 0) {
      MG_HIP(hipMemcpyAsync(m->alpha_host.data(), m->pr[0].alpha, sizeof(double) * n, hipMemcpyDeviceToHost, k.main));
      MG_HIP(hipMemcpyAsync(&m->logp, m->pr[0].scal, sizeof(double), hipMemcpyDeviceToHost, k.main));
    }
    MG_HIP(hipStreamSynchronize(k.main));
  }
  m->noise = noise;
  return 0;
}

int mgpu_prepare(gprc_mgpu* mg, int kernel, const double* params, int n_params, const double* X, int64_t d, int64_t n, const double* y,
                 double noise, gprc_mgpu_model** out) {
  if (!mg || !X || !y || !out || d < 1 || n < 1 || n_params < 0 || (n_params > 0 && !params)) { set_error("mgpu fit: bad arguments"); return GPRC_ERR_ARG; }
  if (!(noise >= 0.0)) { set_error("noise must be >= 0"); return GPRC_ERR_ARG; }
  gprc_mgpu_model* m = new (std::nothrow) gprc_mgpu_model();
  if (!m) { set_error("out of host memory"); return GPRC_ERR_NOMEM; }
  m->mg = mg; m->kernel = kernel; m->params.assign(params, params + n_params); m->n = n; m->d = d; m->n_pad = gprc_pad(n);
  const int G = (int)mg->ranks.size();
  m->pr.resize(G);
  const int64_t n_pad = m->n_pad;
  for (int r = 0; r < G; ++r) {
    Rank& k = mg->ranks[r];
    auto& b = m->pr[r];
    hipError_t e = hipSetDevice(k.device);
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 8); };
    A((void**)&b.X, sizeof(double) * d * n);
    A((void**)&b.y, sizeof(double) * n_pad);
    A((void**)&b.packed, sizeof(double) * gprc_packed_size(n_pad));
    A((void**)&b.winv, sizeof(double) * gprc_winv_size(n_pad));
    A((void**)&b.inv, sizeof(double) * gprc_solve_inv_size(n_pad));
    A((void**)&b.alpha, sizeof(double) * n_pad);
    A((void**)&b.work, sizeof(double) * gprc_trsv_work_size(n_pad));
    A((void**)&b.scal, 64);
    A((void**)&b.info, 64);
    if (e == hipSuccess) e = hipMemcpyAsync(b.X, X, sizeof(double) * d * n, hipMemcpyHostToDevice, k.main);
    if (e == hipSuccess) e = hipMemsetAsync(b.y, 0, sizeof(double) * n_pad, k.main);
    if (e == hipSuccess) e = hipMemcpyAsync(b.y, y, sizeof(double) * n, hipMemcpyHostToDevice, k.main);
    if (e != hipSuccess) { free_model_buffers(m); return hip_fail(e, "mgpu: allocate / stage model buffers", __FILE__, __LINE__); }
  }
  *out = m;
  return 0;
}

int wrap_rank_models(gprc_mgpu_model* m) {
  for (size_t r = 0; r < m->pr.size(); ++r) {
    auto& b = m->pr[r];
    Rank& k = m->mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    GPRC_TRY(gprc_gpr_model_from_device(k.ctx_main, m->kernel, m->params.data(), (int)m->params.size(), b.X, m->d, m->n, b.y, b.packed, b.winv,
                                        b.alpha, m->noise, m->logp, &b.model));
  }
  return 0;
}

}  // namespace

extern "C" {

int gprc_mgpu_create(const int* devices, int n_ranks, int flags, gprc_mgpu** out) {
  if (!devices || n_ranks < 1 || n_ranks > 64 || !out) { set_error("mgpu_create: bad arguments"); return GPRC_ERR_ARG; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device visible: the gprc native path needs an MI355X (gfx950); there is no CPU fallback");
    return GPRC_ERR_NO_DEVICE;
  }
  bool distinct = true;
  for (int i = 0; i < n_ranks; ++i) {
    if (devices[i] < 0 || devices[i] >= count) { set_error("mgpu_create: device index out of range"); return GPRC_ERR_ARG; }
    for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
  }
  const bool want_rccl = (flags & GPRC_MGPU_RCCL) != 0;
  if (want_rccl && !distinct) { set_error("mgpu_create: GPRC_MGPU_RCCL needs distinct devices (virtual ranks exchange by device copies)"); return GPRC_ERR_ARG; }
  if (want_rccl && !rccl().ok) { set_error("mgpu_create: RCCL (librccl.so.1) could not be resolved at run time"); return GPRC_ERR_HIP; }
  gprc_mgpu* mg = new (std::nothrow) gprc_mgpu();
  if (!mg) { set_error("out of host memory"); return GPRC_ERR_NOMEM; }
  mg->use_rccl = want_rccl;
  mg->lookahead = !(flags & GPRC_MGPU_NO_LOOKAHEAD);
  if (const char* e = std::getenv("GPRC_UPDATE_BATCH")) mg->batch = std::max(1, std::atoi(e));
  mg->ranks.resize(n_ranks);
  int rc = 0;
  for (int r = 0; r < n_ranks && rc == 0; ++r) {
    Rank& k = mg->ranks[r];
    k.device = devices[r];
    hipError_t e = hipSetDevice(k.device);
    int lo = 0, hi = 0;
    if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&k.main, hipStreamNonBlocking);
    // the panel work and the exchange must be dispatched AHEAD of the queued workgroups of the trailing update beside them
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&k.side, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&k.comm, hipStreamNonBlocking, hi);
    for (int i = 0; i < 64 && e == hipSuccess; ++i) {
      hipEvent_t ev = nullptr;
      e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e == hipSuccess) k.events.push_back(ev);
    }
    if (e != hipSuccess) { rc = hip_fail(e, "mgpu_create: streams / events", __FILE__, __LINE__); break; }
    rc = gprc_ctx_create(k.device, k.main, &k.ctx_main);
    if (rc == 0) rc = gprc_ctx_create(k.device, k.side, &k.ctx_side);
  }
  if (rc == 0) {
    for (int a = 0; a < n_ranks; ++a)   // direct peer copies where the fabric allows them (errors: already enabled / unsupported -> staged copies)
      for (int b = 0; b < n_ranks; ++b)
        if (mg->ranks[a].device != mg->ranks[b].device) {
          int can = 0;
          (void)hipSetDevice(mg->ranks[a].device);
          if (hipDeviceCanAccessPeer(&can, mg->ranks[a].device, mg->ranks[b].device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(mg->ranks[b].device, 0);
          (void)hipGetLastError();
        }
  }
  if (rc == 0 && mg->use_rccl) {
    std::vector<nccl_comm_t> comms(n_ranks, nullptr);
    const int r0 = rccl().CommInitAll(comms.data(), n_ranks, devices);
    if (r0 != 0) rc = nccl_fail(r0, "ncclCommInitAll");
    else for (int r = 0; r < n_ranks; ++r) mg->ranks[r].nccl = comms[r];
  }
  if (rc != 0) { gprc_mgpu_destroy(mg); return rc; }
  *out = mg;
  return 0;
}

int gprc_mgpu_destroy(gprc_mgpu* mg) {
  if (!mg) return 0;
  for (auto& k : mg->ranks) {
    (void)hipSetDevice(k.device);
    for (hipStream_t s : {k.main, k.side, k.comm})
      if (s) (void)hipStreamSynchronize(s);
    if (k.nccl && rccl().CommDestroy) (void)rccl().CommDestroy(k.nccl);
    if (k.ctx_main) gprc_ctx_destroy(k.ctx_main);
    if (k.ctx_side) gprc_ctx_destroy(k.ctx_side);
    for (hipEvent_t ev : k.events) (void)hipEventDestroy(ev);
    for (hipStream_t s : {k.main, k.side, k.comm})
      if (s) (void)hipStreamDestroy(s);
  }
  delete mg;
  return 0;
}

int gprc_mgpu_ranks(const gprc_mgpu* mg, int* n_ranks_out) {
  if (!mg || !n_ranks_out) { set_error("mgpu_ranks: bad arguments"); return GPRC_ERR_ARG; }
  *n_ranks_out = (int)mg->ranks.size();
  return 0;
}

int gprc_mgpu_gpr_fit(gprc_mgpu* mg, int kernel, const double* params, int n_params, const double* X, int64_t d, int64_t n, const double* y,
                      double noise, gprc_mgpu_model** model_out) {
  gprc_mgpu_model* m = nullptr;
  GPRC_TRY(mgpu_prepare(mg, kernel, params, n_params, X, d, n, y, noise, &m));
  int info = 0;
  int rc = mgpu_attempt(mg, m, noise, &info);
  if (rc == 0 && info == 0) rc = wrap_rank_models(m);
  if (rc != 0 || info != 0) {
    free_model_buffers(m);
    if (rc == 0) set_error("the leading minor of order " + std::to_string(info) + " is not positive definite");
    return rc != 0 ? rc : info;
  }
  *model_out = m;
  return 0;
}

int gprc_mgpu_gpr_fit_retry(gprc_mgpu* mg, int kernel, const double* params, int n_params, const double* X, int64_t d, int64_t n,
                            const double* y, double noise, gprc_mgpu_model** model_out, double* noise_used, int* attempts) {
  gprc_mgpu_model* m = nullptr;
  GPRC_TRY(mgpu_prepare(mg, kernel, params, n_params, X, d, n, y, noise, &m));
  double new_noise = noise;
  for (int i = 1; i <= 10; ++i) {  // R/GPRclass.R:141-148
    int info = 0;
    int rc = mgpu_attempt(mg, m, new_noise, &info);
    if (rc == 0 && info == 0) rc = wrap_rank_models(m);
    if (rc != 0) { free_model_buffers(m); return rc; }
    if (info == 0) {
      if (noise_used) *noise_used = new_noise;
      if (attempts) *attempts = i;
      *model_out = m;
      return 0;
    }
    new_noise = 0.01 * i + noise;
  }
  free_model_buffers(m);
  if (attempts) *attempts = 10;
  set_error("Inputs lead to non positive definite covariance matrix. Try using a larger noise or a smaller lengthscale.");
  return GPRC_ERR_NOT_PD;
}

int gprc_mgpu_gpr_predict(gprc_mgpu_model* m, const double* X_star, int64_t ns, double* mean_out, double* var_out) {
  if (!m || ns < 0 || (ns > 0 && (!X_star || !mean_out || !var_out))) { set_error("mgpu predict: bad arguments"); return GPRC_ERR_ARG; }
  if (ns == 0) return 0;
  const int G = (int)m->pr.size();
  const int64_t per = (ns + G - 1) / G;  // contiguous slices of the test points, no exchange
  std::vector<int> rcs(G, 0);
  std::vector<std::string> errs(G);
  auto work = [&](int r) {
    const int64_t lo = std::min<int64_t>(r * per, ns), hi = std::min<int64_t>((r + 1) * per, ns);
    if (hi <= lo) return;
    rcs[r] = gprc_gpr_predict(m->pr[r].model, X_star + lo * m->d, hi - lo, 1, mean_out + lo, var_out + lo);
    if (rcs[r] != 0) errs[r] = gprc_last_error();  // thread-local text: carry it to the caller's thread
  };
  // one host thread per rank for the predict only: each call stages its slice, runs on its own device and stream and
  // synchronises at the end; the threads never touch the caller's API (R's is single-threaded: SURVEY 8b)
  std::vector<std::thread> th;
  for (int r = 1; r < G; ++r) th.emplace_back(work, r);
  work(0);
  for (auto& t : th) t.join();
  for (int r = 0; r < G; ++r)
    if (rcs[r] != 0) { set_error(errs[r]); return rcs[r]; }
  return 0;
}

int gprc_mgpu_gpr_get_alpha(gprc_mgpu_model* m, double* alpha_out) {
  if (!m || !alpha_out) { set_error("mgpu get_alpha: bad arguments"); return GPRC_ERR_ARG; }
  std::memcpy(alpha_out, m->alpha_host.data(), sizeof(double) * (size_t)m->n);
  return 0;
}
int gprc_mgpu_gpr_get_logp(gprc_mgpu_model* m, double* logp_out) {
  if (!m || !logp_out) { set_error("mgpu get_logp: bad arguments"); return GPRC_ERR_ARG; }
  *logp_out = m->logp;
  return 0;
}
int gprc_mgpu_gpr_get_noise(gprc_mgpu_model* m, double* noise_out) {
  if (!m || !noise_out) { set_error("mgpu get_noise: bad arguments"); return GPRC_ERR_ARG; }
  *noise_out = m->noise;
  return 0;
}
int gprc_mgpu_model_rank(gprc_mgpu_model* m, int rank, gprc_model** model_out) {
  if (!m || !model_out || rank < 0 || rank >= (int)m->pr.size()) { set_error("mgpu model_rank: bad arguments"); return GPRC_ERR_ARG; }
  *model_out = m->pr[rank].model;
  return 0;
}
int gprc_mgpu_model_free(gprc_mgpu_model* m) {
  if (!m) return 0;
  for (auto& k : m->mg->ranks) {
    (void)hipSetDevice(k.device);
    if (k.main) (void)hipStreamSynchronize(k.main);
  }
  free_model_buffers(m);
  return 0;
}

}  // extern "C"
