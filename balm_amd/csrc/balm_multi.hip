// One context over several GPUs of one process (balm_create_multi), and the collective transport of the path.
//
// The path shards over features: Hess, JacT and the residual are sums over features, which the reference
// adds thread by thread (`Hess += hessians[i]`, src/benchmark/bavoxel.hpp:1049-1056).  Here every device holds a
// contiguous shard of the features and a replica of the small per-window state; one host thread per device enqueues
// that device's kernels on its own stream; the per-device payload [SYRK tiles | per-pose gradient + block diagonal |
// residual] is summed in place by ONE stream-ordered RCCL all-reduce over xGMI per Hessian evaluation (one 8-byte
// all-reduce per residual-only evaluation), and assemble + LDL^T solve + pose update run replicated on every device
// (identical inputs, identical arithmetic -> identical steps; the LM decision scalars are nevertheless taken from
// device 0 on all threads so that the collective sequences can never diverge).
//
// RCCL is loaded on first use (dlopen "librccl.so.1": a process that already carries RCCL -- PyTorch's process group --
// shares that copy), so a single-GPU user of libbalm_hip.so never needs it.  The same transport serves the
// one-process-per-GPU launch (torchrun): balm_comm_unique_id / balm_comm_init_rank.
//
// BALM_FLAG_LOOPBACK_SHARDS puts all shards on ONE physical device and replaces the RCCL call by an in-library sum over
// the shards' buffers (same place in the stream order): the test vehicle for the sharding / threading / replication
// logic on a one-GPU box, where a communicator cannot hold the same device twice.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "balm_internal.h"

namespace balm {

// ---- RCCL, loaded lazily ---------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl *rccl() {
  std::call_once(g_rccl_once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names)
      if ((g_rccl.h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!g_rccl.h) { g_rccl.err = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
    auto sym = [&](const char *s) { void *p = dlsym(g_rccl.h, s); if (!p) g_rccl.err = std::string("librccl: missing ") + s; return p; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(g_rccl.h, "ncclCommAbort");     // optional
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.h, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(g_rccl.h, "ncclCommUserRank");
  });
  return g_rccl.err.empty() ? &g_rccl : nullptr;
}
}  // namespace

const char *rccl_load_error() { rccl(); return g_rccl.err.c_str(); }

int comm_unique_id(void *out128) {
  const Rccl *r = rccl();
  if (!r) return BALM_ERR_STATE;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return BALM_ERR_HIP;
  std::memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return BALM_OK;
}

int comm_init_rank(balm_ctx *ctx, int nranks, int rank, const void *id128) {
  const Rccl *r = rccl();
  if (!r) { ctx->err = rccl_load_error(); return BALM_ERR_STATE; }
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t e = r->CommInitRank(&c, nranks, id, rank);
  if (e != ncclSuccess) { ctx->err = std::string("ncclCommInitRank: ") + r->GetErrorString(e); return BALM_ERR_HIP; }
  ctx->comm = (void *)c; ctx->rank = rank; ctx->nranks = nranks;
  return BALM_OK;
}

void comm_destroy(balm_ctx *ctx) {
  if (!ctx->comm) return;
  const Rccl *r = rccl();
  if (r && !ctx->comm_aborted) r->CommDestroy((ncclComm_t)ctx->comm);      // (an aborted communicator is already freed)
  ctx->comm = nullptr;
}

// ranks / own rank as the communicator itself reports them (-1: not available)
void comm_query(const balm_ctx *ctx, int *count, int *rank) {
  *count = -1; *rank = -1;
  const Rccl *r = rccl();
  if (!r || !ctx->comm) return;
  if (r->CommCount) r->CommCount((ncclComm_t)ctx->comm, count);
  if (r->CommUserRank) r->CommUserRank((ncclComm_t)ctx->comm, rank);
}

int comm_allreduce(balm_ctx *ctx, double *buf, long n) {
  const Rccl *r = rccl();
  ncclResult_t e;
  {
    // the enqueue and a peer's ncclCommAbort of this communicator (multi_abort) exclude each other: the handle is
    // either used before it is aborted or not at all
    std::lock_guard<std::timed_mutex> lk(ctx->comm_mu);
    void *comm = ctx->comm;
    if (!comm || ctx->comm_dead.load(std::memory_order_acquire)) {
      ctx->err = "the communicator was aborted after a peer device failed";
      return BALM_ERR_STATE;
    }
    e = r->AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)comm, ctx->stream);
  }
  if (e != ncclSuccess) { ctx->err = std::string("ncclAllReduce: ") + r->GetErrorString(e); return BALM_ERR_HIP; }
  return BALM_OK;
}

// ---- device threads ----------------------------------------------------------------------------------------------
struct Barrier {
  std::mutex mu; std::condition_variable cv; int n = 1, waiting = 0; uint64_t gen = 0; int rc_acc = 0, rc_out = 0;
  const std::atomic<int> *abort_rc = nullptr;      // the job's abort status (balm_multi::abort_rc)
  int arrive(int rc) {          // returns the first non-zero rc any thread brought to this round, or the job's abort status
    std::unique_lock<std::mutex> lk(mu);
    if (rc && !rc_acc) rc_acc = rc;
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; rc_out = rc_acc; rc_acc = 0; gen++; cv.notify_all(); return rc_out; }
    cv.wait(lk, [&] { return gen != g || (abort_rc && abort_rc->load(std::memory_order_acquire)); });
    if (gen == g) {             // a peer left the job with an error and will never arrive: this round is void
      waiting--;
      return abort_rc->load(std::memory_order_acquire);
    }
    return rc_out;
  }
  void reset() { std::lock_guard<std::mutex> lk(mu); waiting = 0; rc_acc = 0; }
  void wake() { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
};

// A device thread returned `rc` != 0 from the current job.  Its peers may be waiting for it on the host (the LM loop's
// scalar hand-over, the loopback barrier) or on the device (an all-reduce it never enqueued): the former are woken with the
// status, the latter are released by ncclCommAbort of their communicators -- the one RCCL call that is meant to come from
// another thread while a collective hangs.  This thread never writes a peer's `comm` pointer: it marks the communicator dead
// under the peer's comm_mu (so the peer either enqueued before the abort or refuses to enqueue after it), and multi_run
// clears the pointers once every device thread of the job has returned.  The context answers BALM_ERR_STATE from then on.
void multi_abort(balm_multi *m, int rc) {
  int expected = 0;
  if (!m->abort_rc.compare_exchange_strong(expected, rc, std::memory_order_acq_rel)) return;
  m->bar->wake();
  if (!m->loopback && m->n > 1) {
    const Rccl *r = rccl();
    if (r && r->CommAbort) {
      m->dead.store(true, std::memory_order_release);
      for (auto *c : m->sub) {
        c->comm_dead.store(true, std::memory_order_release);       // first: no new enqueue starts
        std::unique_lock<std::timed_mutex> lk(c->comm_mu, std::chrono::seconds(2));   // an enqueue in flight returns first (bounded:
        if (c->comm && !c->comm_aborted) {                                             //  a wedged one must not wedge the abort too)
          c->comm_aborted = true;
          r->CommAbort((ncclComm_t)c->comm);
        }
      }
    }
  }
}

static void worker_main(balm_multi *m, int k) {
  hipSetDevice(m->sub[(size_t)k]->device);
  uint64_t seen = 0;
  for (;;) {
    std::unique_lock<std::mutex> lk(m->mu);
    m->cv_go.wait(lk, [&] { return m->quit || m->gen != seen; });
    if (m->quit) return;
    seen = m->gen;
    lk.unlock();
    const int rc = (*m->job)(k);
    if (rc) multi_abort(m, rc);
    lk.lock();
    m->rc[(size_t)k] = rc;
    if (--m->pending == 0) m->cv_done.notify_all();
  }
}

// f(k) on the thread of device k (k = 0: the calling thread); returns the first non-zero result
int multi_run(balm_multi *m, const std::function<int(int)> &f) {
  if (m->n == 1) return f(0);
  if (m->dead.load(std::memory_order_acquire)) { m->sub[0]->err = "the context lost its communicators when a device failed (ncclCommAbort): destroy it"; return BALM_ERR_STATE; }
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->abort_rc.store(0, std::memory_order_release);
    m->bar->reset();
    m->job = &f; m->pending = m->n - 1; m->gen++; m->lm_epoch++;
  }
  m->cv_go.notify_all();
  hipSetDevice(m->sub[0]->device);
  m->rc[0] = f(0);
  if (m->rc[0]) multi_abort(m, m->rc[0]);
  {
    std::unique_lock<std::mutex> lk(m->mu);
    m->cv_done.wait(lk, [&] { return m->pending == 0; });
    m->job = nullptr;
  }
  if (m->dead.load(std::memory_order_acquire))       // every device thread is back: the aborted handles can go (ncclCommAbort freed them)
    for (auto *c : m->sub) { std::lock_guard<std::timed_mutex> lk(c->comm_mu); c->comm = nullptr; }
  for (int rc : m->rc) if (rc) return rc;
  return BALM_OK;
}

balm_multi *multi_new(const std::vector<balm_ctx *> &subs, bool loopback, std::string *err) {
  balm_multi *m = new balm_multi();
  m->n = (int)subs.size(); m->sub = subs; m->loopback = loopback;
  m->fbeg.assign((size_t)m->n + 1, 0); m->rc.assign((size_t)m->n, 0);
  m->bar = new Barrier(); m->bar->n = m->n; m->bar->abort_rc = &m->abort_rc;
  if (!loopback) {
    const Rccl *r = rccl();
    if (!r) { *err = rccl_load_error(); delete m->bar; delete m; return nullptr; }
    std::vector<int> devs;
    for (auto *c : subs) devs.push_back(c->device);
    std::vector<ncclComm_t> comms((size_t)m->n, nullptr);
    const ncclResult_t e = r->CommInitAll(comms.data(), m->n, devs.data());
    if (e != ncclSuccess) { *err = std::string("ncclCommInitAll: ") + r->GetErrorString(e); delete m->bar; delete m; return nullptr; }
    for (int k = 0; k < m->n; k++) subs[(size_t)k]->comm = (void *)comms[(size_t)k];
  } else {
    m->ev1.assign((size_t)m->n, nullptr); m->ev2.assign((size_t)m->n, nullptr);
    m->lb_tmp.assign((size_t)m->n, nullptr); m->lb_cap.assign((size_t)m->n, 0); m->lb_buf.assign((size_t)m->n, nullptr);
    for (int k = 0; k < m->n; k++) {
      hipSetDevice(subs[(size_t)k]->device);
      if (hipEventCreateWithFlags(&m->ev1[(size_t)k], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&m->ev2[(size_t)k], hipEventDisableTiming) != hipSuccess) {
        *err = "loopback transport: event / table allocation failed"; delete m->bar; delete m; return nullptr;
      }
    }
  }
  for (int k = 0; k < m->n; k++) { subs[(size_t)k]->multi = m; subs[(size_t)k]->rank = k; subs[(size_t)k]->nranks = m->n; }
  for (int k = 1; k < m->n; k++) m->th.emplace_back(worker_main, m, k);
  return m;
}

void multi_delete(balm_multi *m) {
  {
    std::lock_guard<std::mutex> lk(m->mu);
    m->quit = true;
  }
  m->cv_go.notify_all();
  for (auto &t : m->th) t.join();
  for (auto *c : m->sub) comm_destroy(c);
  for (auto e : m->ev1) if (e) hipEventDestroy(e);
  for (auto e : m->ev2) if (e) hipEventDestroy(e);
  for (auto p : m->lb_tmp) if (p) hipFree(p);
  delete m->bar;
  delete m;
}

bool multi_is_loopback(const balm_ctx *ctx) { return ctx->multi && ctx->multi->loopback; }

int multi_host_barrier_rc(balm_ctx *ctx, int rc) {
  if (!ctx->multi || ctx->multi->n == 1) return rc;
  return ctx->multi->bar->arrive(rc);
}

// the LM decision scalars of device 0 reach every device thread (identical by construction; this makes it a guarantee)
int multi_share_scalars(balm_ctx *ctx, int it, double *vals, int count) {
  balm_multi *m = ctx->multi;
  if (!m || m->n == 1) return BALM_OK;
  const uint64_t want = (m->lm_epoch << 24) + (uint64_t)it + 1;
  if (ctx->rank == 0) {
    for (int k = 0; k < count; k++) m->lm_vals[it & 1][k] = vals[k];
    m->lm_seq.store(want, std::memory_order_release);
  } else {
    while (m->lm_seq.load(std::memory_order_acquire) < want) {
      // device 0 left the loop with an error (it will never publish this iteration), or another peer failed
      if (const int a = m->abort_rc.load(std::memory_order_acquire)) { ctx->err = "a peer device of the sharded context failed"; return a; }
      std::this_thread::yield();
    }
    for (int k = 0; k < count; k++) vals[k] = m->lm_vals[it & 1][k];
  }
  return m->abort_rc.load(std::memory_order_acquire);
}

// ---- loopback transport: sum over the shards' buffers on one physical device ------------------------------------
struct BufList { const double *p[MAX_SHARDS]; };
__global__ __launch_bounds__(256) void k_sum_buffers(BufList bufs, int nb, long n, double *__restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    double s = bufs.p[0][i];
    for (int b = 1; b < nb; b++) s += bufs.p[b][i];     // fixed order: every shard gets the same bits
    out[i] = s;
  }
}

int loopback_allreduce(balm_ctx *ctx, double *buf, long n) {
  balm_multi *m = ctx->multi;
  const int k = ctx->rank;
  if (m->n == 1) return BALM_OK;
  int rc = BALM_OK;
  if (m->lb_cap[(size_t)k] < (size_t)n) {
    if (m->lb_tmp[(size_t)k]) hipFree(m->lb_tmp[(size_t)k]);
    m->lb_tmp[(size_t)k] = nullptr; m->lb_cap[(size_t)k] = 0;
    if (hipMalloc((void **)&m->lb_tmp[(size_t)k], (size_t)n * sizeof(double)) != hipSuccess) rc = BALM_ERR_HIP;
    else m->lb_cap[(size_t)k] = (size_t)n;
  }
  m->lb_buf[(size_t)k] = buf;
  if (hipEventRecord(m->ev1[(size_t)k], ctx->stream) != hipSuccess) rc = BALM_ERR_HIP;
  if ((rc = m->bar->arrive(rc))) { ctx->err = "loopback all-reduce failed"; return rc; }     // every payload is enqueued
  for (int j = 0; j < m->n; j++) hipStreamWaitEvent(ctx->stream, m->ev1[(size_t)j], 0);
  BufList bl;
  for (int j = 0; j < m->n; j++) bl.p[j] = m->lb_buf[(size_t)j];
  long grid = (n + 255) / 256; if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_sum_buffers, dim3((unsigned)grid), dim3(256), 0, ctx->stream, bl, m->n, n, m->lb_tmp[(size_t)k]);
  hipEventRecord(m->ev2[(size_t)k], ctx->stream);
  if ((rc = m->bar->arrive(0))) { ctx->err = "loopback all-reduce: a peer shard failed"; return rc; }   // every sum is enqueued
  for (int j = 0; j < m->n; j++) hipStreamWaitEvent(ctx->stream, m->ev2[(size_t)j], 0);      // ... and done reading buf
  hipMemcpyAsync(buf, m->lb_tmp[(size_t)k], (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
  return BALM_OK;
}

}  // namespace balm
