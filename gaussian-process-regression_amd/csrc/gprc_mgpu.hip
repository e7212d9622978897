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
//     their own buffers and streams sharing one GPU -- which is how a one-GPU box tests the G = 2, 3, 8 sweeps
//     (tests/c_abi_client.c, tests/c_abi_mgpu8.c), bit-identical to gprc_gpr_fit.
//   * GPRC_MGPU_SCATTER_ALLGATHER (with either transport): the large-message form of the same step.  xGMI is point-to-point
//     (7 links per GPU): a rooted broadcast moves the panel at ONE link's rate per receiver out of the owner, G - 1 full
//     copies of it.  Here the owner hands piece r (1/G of the panel) to rank r -- all its links in parallel, each carrying
//     1/G -- and every rank then collects the other pieces from their holders, so every link carries 1/G twice.  RCCL:
//     grouped ncclSend / ncclRecv; copies: pulls ordered by events.  Pure data movement: results identical.
//     GPRC_MGPU_AUTO_EXCHANGE times both forms on a panel-sized buffer at creation (gprc_mgpu_calibrate) and keeps the faster.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
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
  int (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
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
    x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
    x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
    x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
    x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
    x.ok = x.CommInitAll && x.CommDestroy && x.GetErrorString && x.Broadcast && x.Send && x.Recv && x.GroupStart && x.GroupEnd;
    return x;
  }();
  return r;
}

struct Rank {
  int device = 0;
  hipStream_t main = nullptr, side = nullptr, comm = nullptr;
  gprc_ctx *ctx_main = nullptr, *ctx_side = nullptr;
  nccl_comm_t nccl = nullptr;
  // Recycled round-robin.  Safe by HIP's event semantics: hipStreamWaitEvent captures the record that is current when it is
  // CALLED, so re-recording an event later never disturbs waits enqueued earlier.  A C4 fit on 8 ranks records ~600 per rank.
  std::vector<hipEvent_t> events;
  size_t next_event = 0;
  hipEvent_t t_begin = nullptr, t_swept = nullptr, t_solved = nullptr;  // timing events of the last attempt (gprc_mgpu_stats)
};

// What the last fit / predict on a gprc_mgpu did: a schedule rehearsal record (tests, tools/mgpu_rehearsal.py), not a benchmark
struct Stats {
  int64_t panels = 0, exchange_ops = 0, bytes_in_per_rank = 0, event_pairs = 0, far_passes = 0, lookahead_updates = 0;
  double fit_ms = 0.0, predict_ms = 0.0;
  std::vector<double> sweep_ms, solve_ms, predict_rank_ms;
};

}  // namespace

enum ExchangeMode { EX_COPY_BCAST = 0, EX_RCCL_BCAST = 1, EX_COPY_SAG = 2, EX_RCCL_SAG = 3 };

struct gprc_mgpu {
  std::vector<Rank> ranks;
  bool use_rccl = false;
  bool scatter_allgather = false;   // the large-message form of the exchange step
  bool lookahead = true;
  int batch = 4;  // far panels receive the panels a rank has collected every `batch` steps, in one pass (gprc_dev_update_range)
  Stats st;
  uint64_t id = 0;  // models remember it: a model whose gprc_mgpu is gone must not touch it (gprc_mgpu_model_free, predict)
};

struct gprc_mgpu_model {
  gprc_mgpu* mg = nullptr;
  uint64_t mg_id = 0;
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

// The gprc_mgpu objects that exist (by id): a model may be freed, or asked to predict, after its gprc_mgpu was destroyed -- a host
// language's garbage collector finalises in any order (R: the shim's finalizers) -- and must then refuse instead of
// walking freed ranks.
std::mutex g_live_mu;
std::vector<uint64_t> g_live_ids;
uint64_t g_next_id = 1;
bool mgpu_alive(uint64_t id) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  return std::find(g_live_ids.begin(), g_live_ids.end(), id) != g_live_ids.end();
}

// an event recorded now on `on` (a stream of rank o's device)
int record_on(gprc_mgpu* mg, Rank& o, hipStream_t on, hipEvent_t* ev_out) {
  if (o.events.empty()) { set_error("mgpu: no events"); return GPRC_ERR_ARG; }
  hipEvent_t ev = o.events[o.next_event++ % o.events.size()];
  MG_HIP(hipSetDevice(o.device));
  MG_HIP(hipEventRecord(ev, on));
  *ev_out = ev;
  return 0;
}
int wait_on(gprc_mgpu* mg, Rank& w, hipStream_t waiter, hipEvent_t ev) {
  MG_HIP(hipSetDevice(w.device));
  MG_HIP(hipStreamWaitEvent(waiter, ev, 0));
  ++mg->st.event_pairs;
  return 0;
}
// stream `waiter` (on rank w's device) waits for everything enqueued so far on `on` (rank o's device)
int wait_stream(gprc_mgpu* mg, Rank& w, hipStream_t waiter, Rank& o, hipStream_t on) {
  hipEvent_t ev = nullptr;
  GPRC_TRY(record_on(mg, o, on, &ev));
  return wait_on(mg, w, waiter, ev);
}

void free_model_buffers(gprc_mgpu_model* m) {
  if (!m) return;
  const bool alive = mgpu_alive(m->mg_id);   // otherwise the ranks (devices, contexts) are gone: hand the memory back blindly
  for (size_t r = 0; r < m->pr.size(); ++r) {
    auto& b = m->pr[r];
    if (alive) (void)hipSetDevice(m->mg->ranks[r].device);
    if (b.model) gprc_model_free(b.model);   // borrowed buffers: frees the handle only (and checks its own context's liveness)
    for (void* p : {(void*)b.X, (void*)b.y, (void*)b.packed, (void*)b.winv, (void*)b.inv, (void*)b.alpha, (void*)b.work, (void*)b.scal, (void*)b.info})
      if (p) (void)hipFree(p);
  }
  delete m;
}

int copy_between(Rank& w, double* dst, Rank& o, const double* src, int64_t cnt) {  // on the RECEIVER's communication stream
  if (cnt <= 0) return 0;
  MG_HIP(hipSetDevice(w.device));
  if (w.device == o.device) MG_HIP(hipMemcpyAsync(dst, src, sizeof(double) * cnt, hipMemcpyDeviceToDevice, w.comm));
  else MG_HIP(hipMemcpyPeerAsync(dst, w.device, src, o.device, sizeof(double) * cnt, w.comm));
  return 0;
}

constexpr int64_t SAG_MIN = 1 << 17;  // doubles (1 MiB): below it two rounds cost more latency than they save (as distributed.py)

// The exchange step on one buffer: bufs[r] + off, cnt doubles, from rank src to every rank, on the communication streams.
// The owner's communication stream has already been ordered behind the data (share_panel / calibrate do that).
int exchange(gprc_mgpu* mg, const std::vector<double*>& bufs, int64_t off, int64_t cnt, int src, bool allow_split) {
  const int G = (int)mg->ranks.size();
  Rank& o = mg->ranks[src];
  const bool split = allow_split && mg->scatter_allgather && G > 1 && cnt >= SAG_MIN;
  // piece j = [j * piece, min(cnt, (j + 1) * piece)): 4 KiB granules, the last one may be short or empty
  const int64_t piece = split ? ((cnt + G - 1) / G + 511) / 512 * 512 : cnt;
  auto lo = [&](int j) { return std::min<int64_t>(cnt, (int64_t)j * piece); };
  auto len = [&](int j) { return std::min<int64_t>(cnt, (int64_t)(j + 1) * piece) - lo(j); };
  if (mg->use_rccl) {
    if (!split) {
      MG_NCCL(rccl().GroupStart());
      for (int r = 0; r < G; ++r) {
        MG_HIP(hipSetDevice(mg->ranks[r].device));
        MG_NCCL(rccl().Broadcast(bufs[r] + off, bufs[r] + off, (size_t)cnt, NCCL_DOUBLE, src, mg->ranks[r].nccl, mg->ranks[r].comm));
      }
      MG_NCCL(rccl().GroupEnd());
      ++mg->st.exchange_ops;
      return 0;
    }
    MG_NCCL(rccl().GroupStart());   // round 1: the owner hands piece r to rank r
    for (int r = 0; r < G; ++r) {
      MG_HIP(hipSetDevice(mg->ranks[r].device));
      if (r == src) {
        for (int j = 0; j < G; ++j)
          if (j != src && len(j) > 0) MG_NCCL(rccl().Send(bufs[src] + off + lo(j), (size_t)len(j), NCCL_DOUBLE, j, o.nccl, o.comm));
      } else if (len(r) > 0) {
        MG_NCCL(rccl().Recv(bufs[r] + off + lo(r), (size_t)len(r), NCCL_DOUBLE, src, mg->ranks[r].nccl, mg->ranks[r].comm));
      }
    }
    MG_NCCL(rccl().GroupEnd());
    MG_NCCL(rccl().GroupStart());   // round 2: every holder hands its piece to everybody who lacks it (the owner lacks nothing)
    for (int r = 0; r < G; ++r) {
      Rank& k = mg->ranks[r];
      MG_HIP(hipSetDevice(k.device));
      for (int j = 0; j < G; ++j) {
        if (j == r) continue;
        if (j != src && len(r) > 0) MG_NCCL(rccl().Send(bufs[r] + off + lo(r), (size_t)len(r), NCCL_DOUBLE, j, k.nccl, k.comm));   // my piece -> j
        if (r != src && len(j) > 0) MG_NCCL(rccl().Recv(bufs[r] + off + lo(j), (size_t)len(j), NCCL_DOUBLE, j, k.nccl, k.comm));   // j's piece -> me
      }
    }
    MG_NCCL(rccl().GroupEnd());
    mg->st.exchange_ops += 2;
    return 0;
  }
  if (G == 1) return 0;
  hipEvent_t ready = nullptr;   // the data is complete on the owner's communication stream: ONE record, every receiver waits on it
  GPRC_TRY(record_on(mg, o, o.comm, &ready));
  if (!split) {
    for (int r = 0; r < G; ++r) {
      if (r == src) continue;
      Rank& w = mg->ranks[r];
      GPRC_TRY(wait_on(mg, w, w.comm, ready));
      GPRC_TRY(copy_between(w, bufs[r] + off, o, bufs[src] + off, cnt));
      ++mg->st.exchange_ops;
    }
    return 0;
  }
  std::vector<hipEvent_t> have(G, nullptr);   // rank j holds piece j
  for (int r = 0; r < G; ++r) {               // round 1: rank r pulls piece r from the owner
    if (r == src) continue;
    Rank& w = mg->ranks[r];
    GPRC_TRY(wait_on(mg, w, w.comm, ready));
    GPRC_TRY(copy_between(w, bufs[r] + off + lo(r), o, bufs[src] + off + lo(r), len(r)));
    GPRC_TRY(record_on(mg, w, w.comm, &have[r]));
    ++mg->st.exchange_ops;
  }
  for (int r = 0; r < G; ++r) {               // round 2: rank r pulls every other piece from its holder
    if (r == src) continue;
    Rank& w = mg->ranks[r];
    for (int jj = 1; jj < G; ++jj) {
      const int j = (r + jj) % G;             // staggered: at any moment the G - 1 pullers address G - 1 different holders
      if (len(j) <= 0) continue;
      if (j != src) GPRC_TRY(wait_on(mg, w, w.comm, have[j]));
      GPRC_TRY(copy_between(w, bufs[r] + off + lo(j), mg->ranks[j], bufs[j] + off + lo(j), len(j)));
      ++mg->st.exchange_ops;
    }
  }
  return 0;
}

// the exchange step: panel p (+ the inverses of its four diagonal blocks) from its owner to every rank
int share_panel(gprc_mgpu* mg, gprc_mgpu_model* m, int64_t p) {
  const int G = (int)mg->ranks.size();
  if (G == 1 && !mg->use_rccl) return 0;
  const int src = (int)(p % G);
  const int64_t off = gprc_panel_offset(m->n_pad, p), cnt = gprc_panel_elems(m->n_pad, p);
  const int64_t woff = p * (NB / NBI) * NBI * NBI, wcnt = (int64_t)(NB / NBI) * NBI * NBI;
  Rank& o = mg->ranks[src];
  GPRC_TRY(wait_stream(mg, o, o.comm, o, o.side));  // the panel is complete on the owner's side stream
  std::vector<double*> pk(G), wi(G);
  for (int r = 0; r < G; ++r) { pk[r] = m->pr[r].packed; wi[r] = m->pr[r].winv; }
  GPRC_TRY(exchange(mg, pk, off, cnt, src, true));
  GPRC_TRY(exchange(mg, wi, woff, wcnt, src, false));
  mg->st.bytes_in_per_rank += (int64_t)sizeof(double) * (cnt + wcnt) * (G - 1) / G;   // the average rank owns 1/G of the panels
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
  const auto wall0 = std::chrono::steady_clock::now();
  mg->st = Stats();
  mg->st.panels = P;
  mg->st.sweep_ms.assign(G, 0.0);
  mg->st.solve_ms.assign(G, 0.0);
  auto fill_own = [&](int r) -> int {   // F1: own panels, no exchange
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipMemsetAsync(m->pr[r].info, 0, 16, k.main));
    for (int64_t p = r; p < P; p += G)
      GPRC_TRY(gprc_dev_fill_panel(k.ctx_main, m->kernel, par, npar, m->pr[r].X, d, n, n_pad, noise, m->pr[r].packed, p));
    return 0;
  };
  for (int r = 0; r < G; ++r) {
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipEventRecord(k.t_begin, k.main));
    GPRC_TRY(fill_own(r));
  }
  if (G == 1 && !mg->use_rccl) {
    Rank& k = mg->ranks[0];
    GPRC_TRY(gprc_dev_factor_all(k.ctx_main, m->pr[0].packed, n_pad, m->pr[0].winv, m->pr[0].info, m->pr[0].inv));
    // gprc_dev_factor_all is asynchronous and may run under the factor service, whose device-side waits run out when kernels of
    // two streams cannot run concurrently (a tool that serialises dispatches: rocprofv3 --pmc).  What gprc_gpr_fit does then
    // (include/gprc_native.h), this path does too: service off, matrix rebuilt, one fused launch per panel.
    int ir = 0;
    MG_HIP(hipMemcpyAsync(&ir, m->pr[0].info, sizeof(int), hipMemcpyDeviceToHost, k.main));
    MG_HIP(hipStreamSynchronize(k.main));
    if (ir == GPRC_INFO_WAIT_TIMEOUT && gprc_factor_service(-1) == 1) {
      gprc_factor_service(0);
      GPRC_TRY(fill_own(0));
      GPRC_TRY(gprc_dev_factor_all(k.ctx_main, m->pr[0].packed, n_pad, m->pr[0].winv, m->pr[0].info, m->pr[0].inv));
    }
  } else {
    auto factor_and_share = [&](int64_t p) -> int {
      const int src = (int)(p % G);
      Rank& o = mg->ranks[src];
      MG_HIP(hipSetDevice(o.device));
      GPRC_TRY(gprc_dev_factor_panel(o.ctx_side, m->pr[src].packed, n_pad, p, m->pr[src].winv, m->pr[src].info));
      return share_panel(mg, m, p);
    };
    for (int r = 0; r < G; ++r) GPRC_TRY(wait_stream(mg, mg->ranks[r], mg->ranks[r].side, mg->ranks[r], mg->ranks[r].main));  // fork
    GPRC_TRY(factor_and_share(0));
    std::vector<int64_t> far_from(G, 0);  // rank r has applied panels [0, far_from[r]) to all its unfactored panels
    for (int64_t p = 0; p < P; ++p) {
      for (int r = 0; r < G; ++r) {  // panel p is factored (owner: side stream) / received (others: comm stream)
        Rank& k = mg->ranks[r];
        GPRC_TRY(wait_stream(mg, k, k.main, k, k.side));
        GPRC_TRY(wait_stream(mg, k, k.main, k, k.comm));
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
          ++mg->st.far_passes;
          GPRC_TRY(wait_stream(mg, k, k.side, k, k.main));
        } else if (r == nxt) {             // look-ahead: panel p + 1 first, on the side stream
          GPRC_TRY(wait_stream(mg, k, k.side, k, k.main));
          GPRC_TRY(gprc_dev_update_range(k.ctx_side, m->pr[r].packed, n_pad, far_from[r], p + 1, p + 1, p + 2, 1));
          ++mg->st.lookahead_updates;
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
        if (q0[r] < P) { GPRC_TRY(gprc_dev_update_range(k.ctx_main, m->pr[r].packed, n_pad, far_from[r], p + 1, q0[r], P, G)); ++mg->st.far_passes; }
        far_from[r] = p + 1;
      }
    }
  }
  // info: the first failing column over all ranks (every rank holds its own panels' verdicts)
  int info = 0;
  for (int r = 0; r < G; ++r) {
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipEventRecord(k.t_swept, k.main));
    MG_HIP(hipStreamSynchronize(k.side));
    MG_HIP(hipStreamSynchronize(k.comm));
    int ir = 0;
    MG_HIP(hipMemcpyAsync(&ir, m->pr[r].info, sizeof(int), hipMemcpyDeviceToHost, k.main));
    MG_HIP(hipStreamSynchronize(k.main));
    if (ir < 0) { set_error("mgpu: a device-side dependency wait timed out on rank " + std::to_string(r) + " (info = " + std::to_string(ir) + ")"); return GPRC_ERR_HIP; }
    if (ir > 0 && (info == 0 || ir < info)) info = ir;
  }
  *info_out = info;
  auto close_stats = [&](bool solved) {
    for (int r = 0; r < G; ++r) {
      Rank& k = mg->ranks[r];
      (void)hipSetDevice(k.device);
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, k.t_begin, k.t_swept) == hipSuccess) mg->st.sweep_ms[r] = ms;
      if (solved && hipEventElapsedTime(&ms, k.t_swept, k.t_solved) == hipSuccess) mg->st.solve_ms[r] = ms;
      (void)hipGetLastError();
    }
    mg->st.fit_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  };
  if (info != 0) { close_stats(false); return 0; }
  for (int r = 0; r < G; ++r) {  // F3, replicated: L is complete on every rank
    Rank& k = mg->ranks[r];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipMemcpyAsync(m->pr[r].alpha, m->pr[r].y, sizeof(double) * n_pad, hipMemcpyDeviceToDevice, k.main));
    // the explicit diagonal-block inverses the vector solves work with: every rank computes its own copy (never exchanged)
    if (!(G == 1 && !mg->use_rccl)) GPRC_TRY(gprc_dev_solve_prepare(k.ctx_main, m->pr[r].packed, m->pr[r].winv, n_pad, m->pr[r].inv, 0, P));
    GPRC_TRY(gprc_dev_trsv(k.ctx_main, m->pr[r].packed, m->pr[r].inv, n_pad, m->pr[r].alpha, 0, m->pr[r].work));
    GPRC_TRY(gprc_dev_trsv(k.ctx_main, m->pr[r].packed, m->pr[r].inv, n_pad, m->pr[r].alpha, 1, m->pr[r].work));
    GPRC_TRY(gprc_dev_logp(k.ctx_main, m->pr[r].packed, n_pad, n, m->pr[r].y, m->pr[r].alpha, m->pr[r].scal));
    MG_HIP(hipEventRecord(k.t_solved, k.main));
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
  close_stats(true);
  return 0;
}

int mgpu_prepare(gprc_mgpu* mg, int kernel, const double* params, int n_params, const double* X, int64_t d, int64_t n, const double* y,
                 double noise, gprc_mgpu_model** out) {
  if (!mg || !X || !y || !out || d < 1 || n < 1 || n_params < 0 || (n_params > 0 && !params)) { set_error("mgpu fit: bad arguments"); return GPRC_ERR_ARG; }
  if (!(noise >= 0.0)) { set_error("noise must be >= 0"); return GPRC_ERR_ARG; }
  gprc_mgpu_model* m = new (std::nothrow) gprc_mgpu_model();
  if (!m) { set_error("out of host memory"); return GPRC_ERR_NOMEM; }
  m->mg = mg; m->mg_id = mg->id; m->kernel = kernel; m->params.assign(params, params + n_params); m->n = n; m->d = d; m->n_pad = gprc_pad(n);
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
  mg->scatter_allgather = (flags & GPRC_MGPU_SCATTER_ALLGATHER) != 0;
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
    for (hipEvent_t* tev : {&k.t_begin, &k.t_swept, &k.t_solved})
      if (e == hipSuccess) e = hipEventCreate(tev);
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
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    mg->id = g_next_id++;
    g_live_ids.push_back(mg->id);
  }
  if (flags & GPRC_MGPU_AUTO_EXCHANGE) {   // the slower form loses; a tie keeps the rooted broadcast
    rc = gprc_mgpu_calibrate(mg, (int64_t)1 << 24, 3, nullptr, nullptr);
    if (rc != 0) { gprc_mgpu_destroy(mg); return rc; }
  }
  *out = mg;
  return 0;
}

int gprc_mgpu_destroy(gprc_mgpu* mg) {
  if (!mg) return 0;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live_ids.erase(std::remove(g_live_ids.begin(), g_live_ids.end(), mg->id), g_live_ids.end());
  }
  for (auto& k : mg->ranks) {
    (void)hipSetDevice(k.device);
    for (hipStream_t s : {k.main, k.side, k.comm})
      if (s) (void)hipStreamSynchronize(s);
    if (k.nccl && rccl().CommDestroy) (void)rccl().CommDestroy(k.nccl);
    if (k.ctx_main) gprc_ctx_destroy(k.ctx_main);
    if (k.ctx_side) gprc_ctx_destroy(k.ctx_side);
    for (hipEvent_t ev : k.events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {k.t_begin, k.t_swept, k.t_solved})
      if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t s : {k.main, k.side, k.comm})
      if (s) (void)hipStreamDestroy(s);
  }
  delete mg;
  return 0;
}

// Times the two forms of the exchange step on a `doubles`-sized buffer per rank (the root rotating), checks that they deliver the
// same data, and keeps the faster one (scatter + all-gather has to win by 10 %).
int gprc_mgpu_calibrate(gprc_mgpu* mg, int64_t doubles, int reps, int* choice_out, double* ms_out) {
  if (!mg || doubles < 1 || reps < 1) { set_error("mgpu_calibrate: bad arguments"); return GPRC_ERR_ARG; }
  const int G = (int)mg->ranks.size();
  doubles = (doubles + 511) / 512 * 512;
  std::vector<double*> bufs(G, nullptr);
  struct Free { std::vector<double*>& b; gprc_mgpu* mg; ~Free() { for (size_t r = 0; r < b.size(); ++r) if (b[r]) { (void)hipSetDevice(mg->ranks[r].device); (void)hipFree(b[r]); } } } guard{bufs, mg};
  for (int r = 0; r < G; ++r) {
    MG_HIP(hipSetDevice(mg->ranks[r].device));
    MG_HIP(hipMalloc((void**)&bufs[r], sizeof(double) * (size_t)doubles));
  }
  const bool keep = mg->scatter_allgather;
  const Stats keep_st = mg->st;
  double ms[2] = {0.0, 0.0};
  bool agree = true;
  const int64_t probe = 512;   // doubles checked at the start and the end of every piece's range
  std::vector<uint32_t> host((size_t)probe * 2);
  int rc = 0;
  for (int mode = 0; mode < 2 && rc == 0; ++mode) {
    mg->scatter_allgather = mode == 1;
    std::chrono::steady_clock::time_point t0;
    for (int it = 0; it <= reps && rc == 0; ++it) {   // iteration 0 warms up
      const int src = it % G;
      {   // nobody may still be pulling from a buffer that is about to be overwritten: every communication stream waits for all others
        std::vector<hipEvent_t> done(G, nullptr);
        for (int r = 0; r < G && rc == 0; ++r) rc = record_on(mg, mg->ranks[r], mg->ranks[r].comm, &done[r]);
        for (int r = 0; r < G && rc == 0; ++r)
          for (int j = 0; j < G && rc == 0; ++j)
            if (j != r) rc = wait_on(mg, mg->ranks[r], mg->ranks[r].comm, done[j]);
        if (rc != 0) break;
      }
      for (int r = 0; r < G; ++r) {
        Rank& k = mg->ranks[r];
        MG_HIP(hipSetDevice(k.device));
        MG_HIP(hipMemsetD32Async((hipDeviceptr_t)bufs[r], r == src ? 0x3ff00000 + it : 0, (size_t)doubles * 2, k.comm));
      }
      if (it == 1) {
        for (int r = 0; r < G; ++r) { MG_HIP(hipSetDevice(mg->ranks[r].device)); MG_HIP(hipStreamSynchronize(mg->ranks[r].comm)); }
        t0 = std::chrono::steady_clock::now();
      }
      rc = exchange(mg, bufs, 0, doubles, src, true);
    }
    if (rc != 0) break;
    for (int r = 0; r < G; ++r) { MG_HIP(hipSetDevice(mg->ranks[r].device)); MG_HIP(hipStreamSynchronize(mg->ranks[r].comm)); }
    ms[mode] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    const uint32_t expect = 0x3ff00000 + (uint32_t)reps;
    for (int r = 0; r < G && agree; ++r) {
      MG_HIP(hipSetDevice(mg->ranks[r].device));
      for (int64_t at = 0; at < doubles && agree; at += std::max<int64_t>(probe, doubles / (4 * G) / 512 * 512)) {
        const int64_t cntp = std::min<int64_t>(probe, doubles - at);
        MG_HIP(hipMemcpy(host.data(), bufs[r] + at, sizeof(double) * (size_t)cntp, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < cntp * 2; ++i) agree = agree && host[(size_t)i] == expect;
      }
    }
  }
  mg->st = keep_st;
  mg->scatter_allgather = keep;
  if (rc != 0) return rc;
  if (!agree) { set_error("mgpu_calibrate: the two exchange forms delivered different data"); return GPRC_ERR_HIP; }
  mg->scatter_allgather = G > 1 && ms[1] < 0.9 * ms[0];
  if (choice_out) *choice_out = mg->scatter_allgather ? 1 : 0;
  if (ms_out) { ms_out[0] = ms[0]; ms_out[1] = ms[1]; }
  return 0;
}

int gprc_mgpu_exchange_mode(const gprc_mgpu* mg, int* mode_out) {
  if (!mg || !mode_out) { set_error("mgpu_exchange_mode: bad arguments"); return GPRC_ERR_ARG; }
  *mode_out = (mg->use_rccl ? EX_RCCL_BCAST : EX_COPY_BCAST) + (mg->scatter_allgather ? 2 : 0);
  return 0;
}

// Layout: [0] ranks G, [1] panels P, [2] exchange mode (gprc_mgpu_exchange_mode), [3] exchange operations issued (copies /
// collectives), [4] bytes received per rank, [5] event record + wait pairs, [6] far-update passes, [7] look-ahead updates,
// [8] fit wall ms, [9] predict wall ms, then per rank r: [10 + 3r] fill + sweep ms, [11 + 3r] alpha / logp ms (HIP events on the
// rank's main stream), [12 + 3r] predict ms of the rank's slice (host clock of its thread).
int gprc_mgpu_stats(const gprc_mgpu* mg, double* out, int n) {
  if (!mg || !out || n < 10) { set_error("mgpu_stats: bad arguments (n >= 10)"); return GPRC_ERR_ARG; }
  const Stats& st = mg->st;
  const int G = (int)mg->ranks.size();
  int mode = 0;
  (void)gprc_mgpu_exchange_mode(mg, &mode);
  const double head[10] = {(double)G, (double)st.panels, (double)mode, (double)st.exchange_ops, (double)st.bytes_in_per_rank,
                           (double)st.event_pairs, (double)st.far_passes, (double)st.lookahead_updates, st.fit_ms, st.predict_ms};
  for (int i = 0; i < 10; ++i) out[i] = head[i];
  for (int r = 0; r < G && 12 + 3 * r < n; ++r) {
    out[10 + 3 * r] = r < (int)st.sweep_ms.size() ? st.sweep_ms[r] : 0.0;
    out[11 + 3 * r] = r < (int)st.solve_ms.size() ? st.solve_ms[r] : 0.0;
    out[12 + 3 * r] = r < (int)st.predict_rank_ms.size() ? st.predict_rank_ms[r] : 0.0;
  }
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
  if (!mgpu_alive(m->mg_id)) { set_error("mgpu predict: the gprc_mgpu this model was fitted on has been destroyed"); return GPRC_ERR_ARG; }
  const int G = (int)m->pr.size();
  const int64_t per = (ns + G - 1) / G;  // contiguous slices of the test points, no exchange
  std::vector<int> rcs(G, 0);
  std::vector<std::string> errs(G);
  const auto wall0 = std::chrono::steady_clock::now();
  m->mg->st.predict_rank_ms.assign(G, 0.0);
  auto work = [&](int r) {
    const int64_t lo = std::min<int64_t>(r * per, ns), hi = std::min<int64_t>((r + 1) * per, ns);
    if (hi <= lo) return;
    rcs[r] = gprc_gpr_predict(m->pr[r].model, X_star + lo * m->d, hi - lo, 1, mean_out + lo, var_out + lo);
    if (rcs[r] != 0) errs[r] = gprc_last_error();  // thread-local text: carry it to the caller's thread
    m->mg->st.predict_rank_ms[r] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  };
  // one host thread per rank for the predict only: each call stages its slice, runs on its own device and stream and
  // synchronises at the end; the threads never touch the caller's API (R's is single-threaded: SURVEY 8b)
  std::vector<std::thread> th;
  for (int r = 1; r < G; ++r) th.emplace_back(work, r);
  work(0);
  for (auto& t : th) t.join();
  m->mg->st.predict_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
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
  if (mgpu_alive(m->mg_id)) {
    for (auto& k : m->mg->ranks) {
      (void)hipSetDevice(k.device);
      if (k.main) (void)hipStreamSynchronize(k.main);
    }
  } else {
    (void)hipDeviceSynchronize();   // the gprc_mgpu went first (its streams were synchronised and destroyed): only the memory is left
  }
  free_model_buffers(m);
  return 0;
}

}  // extern "C"
