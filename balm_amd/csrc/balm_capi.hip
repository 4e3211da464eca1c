// Host side of libbalm_hip.so: the C ABI of include/balm_hip.h.  Owns the HBM-resident problem
// (per-feature SoA clusters, poses, Hessian), sequences the HIP kernels on one stream, and runs the
// Levenberg-Marquardt loop of BALM2::damping_iter (src/benchmark/bavoxel.hpp:1069-1166) with every
// matrix device-resident; only four scalars cross PCIe per iteration.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "balm_internal.h"

using namespace balm;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                \
      return BALM_ERR_HIP;                                                                         \
    }                                                                                              \
  } while (0)

namespace {

// A/B builds only (tools/build_ab.sh cold balm_capi.hip -DBALM_COLD_TRACE): where the first call of a process spends its time, on stderr
#ifdef BALM_COLD_TRACE
static void cold_mark(const char *what) {
  static const auto t0 = std::chrono::steady_clock::now();
  static double last = 0;
  const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  fprintf(stderr, "[cold] %9.3f ms (+%8.3f)  %s\n", now, now - last, what);
  last = now;
}
#else
#define cold_mark(what) ((void)0)
#endif

// ---- device warm-up in the background ----------------------------------------------------------------------------------
// What the FIRST use of a device costs a process beyond the runtime's own start: 13-16 ms for the pinned ring (hipHostMalloc of 3 x 32
// MB), 35-45 ms for the code objects of the association / cluster-build / solve translation units, which the runtime loads at the first
// launch of one of their kernels (profiles/r06_cold_call.txt).  The reference's drivers are one-shot programs (benchmark_realworld.cpp:179
// runs its loop once): that first use IS their run.  balm_prewarm(device) -- called by balm_create, and by the shim's constructor
// before anything else is known -- does both on a background thread; the first upload adopts the ring, the first launches find their
// code loaded.  A caller that declares its BALM2_HIP first thing in main() overlaps all of it with reading its scans.
struct DeviceWarm {
  std::mutex mu;
  std::thread th;
  bool started = false;
  PinnedRing ring;
  std::atomic<bool> ring_ready{false};
  ~DeviceWarm() { if (th.joinable()) th.join(); }       // (a ring nobody adopted is left to the process teardown)
};
static DeviceWarm g_warm[64];

static void warm_body(int device) {
  DeviceWarm &w = g_warm[device];
  if (hipSetDevice(device) != hipSuccess) { hipGetLastError(); return; }
  std::thread ring_th([&w, device] {
    if (hipSetDevice(device) == hipSuccess && w.ring.init(device) == hipSuccess) w.ring_ready.store(true, std::memory_order_release);
    else hipGetLastError();
  });
  (void)preload_voxel(); (void)preload_build(); (void)prepare_device_accum(); (void)preload_solve(); (void)prepare_device_cov();
  hipGetLastError();
  ring_th.join();
}

static int warm_start(int device) {
  if (device < 0 || device >= 64) return BALM_ERR_ARG;
  DeviceWarm &w = g_warm[device];
  std::lock_guard<std::mutex> lk(w.mu);
  if (!w.started) { w.started = true; w.th = std::thread(warm_body, device); }
  return BALM_OK;
}

// before the context's first big upload: the ring the warm-up allocated becomes this context's (once per device; later contexts
// allocate their own on first use, as before)
static void adopt_warm_ring(balm_ctx *ctx) {
  if (ctx->ring.buf[0] || ctx->device < 0 || ctx->device >= 64) return;
  DeviceWarm &w = g_warm[ctx->device];
  std::lock_guard<std::mutex> lk(w.mu);
  if (!w.started) return;
  if (w.th.joinable()) w.th.join();
  if (w.ring_ready.load(std::memory_order_acquire)) {
    const int key = ctx->ring.pool_key;
    ctx->ring = w.ring; ctx->ring.pool_key = key;
    w.ring = PinnedRing(); w.ring_ready.store(false);
  }
}

// ---- timing ------------------------------------------------------------------------------------
struct Span {
  balm_ctx *c; int slot; hipEvent_t a = nullptr, b = nullptr;
  Span(balm_ctx *ctx, int s) : c(ctx), slot(s) {
    if (!c->timer.on) return;
    a = take(); b = take();
    hipEventRecord(a, c->stream);
  }
  ~Span() {
    if (!c->timer.on) return;
    hipEventRecord(b, c->stream);
    c->timer.pending.push_back({a, b, slot});
  }
  hipEvent_t take() {
    if (!c->timer.pool.empty()) { hipEvent_t e = c->timer.pool.back(); c->timer.pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
};

void collect_timing(balm_ctx *c) {   // call only after the stream is synchronised
  for (auto &sp : c->timer.pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { c->timer.ms[sp.slot] += ms; c->timer.cnt[sp.slot]++; }
    c->timer.pool.push_back(sp.a); c->timer.pool.push_back(sp.b);
  }
  c->timer.pending.clear();
}

template <class T>
int dalloc(balm_ctx *ctx, T **p, size_t count) {
  if (*p) { hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
  return BALM_OK;
}

// grow-only buffers of the feature table: a caller that installs a table per window (balm_associate, balm_window_features,
// balm_set_features in a loop) pays for hipMalloc / hipFree only while the tables grow
template <class T>
int keep(balm_ctx *ctx, T **p, size_t *cap, size_t count) {
  if (*p && *cap >= count) return BALM_OK;
  for (int k = 0; k < balm_ctx::LM_GRAPHS; k++)           // captured LM graphs hold the old pointer (as in ensure())
    if (ctx->lm_graph[k]) { hipGraphExecDestroy(ctx->lm_graph[k]); ctx->lm_graph[k] = nullptr; ctx->lm_graph_form[k] = -1; }
  int rc = dalloc(ctx, p, count + count / 4);
  *cap = rc ? 0 : count + count / 4;
  return rc;
}

template <class T>
int ensure(balm_ctx *ctx, T **p, size_t *cap, size_t count) {
  if (*p && *cap >= count) return BALM_OK;
  for (int k = 0; k < balm_ctx::LM_GRAPHS; k++)           // captured LM graphs hold the old pointer
    if (ctx->lm_graph[k]) { hipGraphExecDestroy(ctx->lm_graph[k]); ctx->lm_graph[k] = nullptr; ctx->lm_graph_form[k] = -1; }
  int rc = dalloc(ctx, p, count);
  if (rc) return rc;
  *cap = count;
  return BALM_OK;
}

// staging arena: stage_begin(total) once per call (may reallocate: nothing staged may be live), then stage_take pieces
int stage_begin(balm_ctx *ctx, size_t total_bytes) {
  total_bytes += 16 * 256;                                 // alignment slack for up to 16 pieces
  ctx->stage_off = 0;
  if (ctx->stage_cap >= total_bytes) return BALM_OK;
  if (ctx->d_stage) { hipStreamSynchronize(ctx->stream); hipFree(ctx->d_stage); ctx->d_stage = nullptr; ctx->stage_cap = 0; }
  const size_t want = total_bytes + total_bytes / 8;
  HIP_TRY(hipMalloc((void **)&ctx->d_stage, want));
  ctx->stage_cap = want;
  return BALM_OK;
}

template <class T>
T *stage_take(balm_ctx *ctx, size_t count) {
  const size_t off = (ctx->stage_off + 255) & ~(size_t)255;
  ctx->stage_off = off + count * sizeof(T);
  return reinterpret_cast<T *>(ctx->d_stage + off);
}

int sync_stream(balm_ctx *ctx) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipGetLastError());
  collect_timing(ctx);
  return BALM_OK;
}

long red_dacc_off(const balm_ctx *c) { return (long)c->ntiles * TILE_ELEMS; }
long red_r_off(const balm_ctx *c) { return red_dacc_off(c) + (long)DACC_MAX * c->W; }

// Sum `n` doubles at `buf` over all ranks, in place, ordered after everything enqueued on ctx->stream so far and
// before everything enqueued later.  Transports: RCCL inside the library (a device of a balm_create_multi context, or
// a rank of balm_comm_init_rank) -- stream-ordered, no host synchronisation; the loopback sum of shards that share one
// physical device; or the caller's hook (balm_set_allreduce), which gets a synchronised stream.
bool has_transport(const balm_ctx *ctx) { return ctx->comm || multi_is_loopback(ctx) || ctx->allreduce; }

int hook_allreduce(balm_ctx *ctx, double *buf, long n) {
  if (ctx->comm) { Span sp(ctx, BALM_T_COMM); return comm_allreduce(ctx, buf, n); }
  if (multi_is_loopback(ctx)) { Span sp(ctx, BALM_T_COMM); return loopback_allreduce(ctx, buf, n); }
  if (!ctx->allreduce) return BALM_OK;
  int rc = sync_stream(ctx);
  if (rc) return rc;
  if (ctx->allreduce((void *)buf, n, ctx->allreduce_user) != 0) {
    ctx->err = "all-reduce hook failed";
    return BALM_ERR_STATE;
  }
  return BALM_OK;
}

// the damping u of the next solve: a kernel argument of the plain launches (launch_solve) ...
int set_damping(balm_ctx *ctx, double u) {
  ctx->u_value = u;
  ctx->u_on_device = false;
  return BALM_OK;
}
// ... or, for a captured / replayed LM graph, a value in device memory that travels through a pinned ring (the copy is
// asynchronous; 64 solves can be in flight)
int push_damping(balm_ctx *ctx) {
  double *slot = ctx->h_scal + 16 + (ctx->u_ring++ & 63);
  *slot = ctx->u_value;
  HIP_TRY(hipMemcpyAsync(ctx->d_scal + SCAL_U, slot, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ctx->u_on_device = true;
  return BALM_OK;
}

// (kernel timing: the spans' events are read once the stream is known to be idle -- the stamp says it is, the synchronise is then cheap)
static void collect_timing_if_idle(balm_ctx *ctx) {
  if (!ctx->timer.on || ctx->timer.pending.empty()) return;
  if (hipStreamSynchronize(ctx->stream) == hipSuccess) collect_timing(ctx);
}

// The host's end of k_scalars_mail: poll the stamp in the pinned mirror (a copy command + hipStreamSynchronize cost 15-25 us per
// LM iteration on this stack -- 5 % of an iteration on the shipped window); after ~300 us of polling the wait becomes a
// stream synchronise (long iterations: nothing to gain from spinning).  Errors of the stream surface there or at the
// next synchronising call.
int wait_scalars(balm_ctx *ctx) {
  volatile double *stamp = ctx->h_scal + SCAL_STAMP;
  const double want = (double)ctx->mail_seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (int spin = 0;; spin++) {
    if (*stamp == want) { std::atomic_thread_fence(std::memory_order_acquire); collect_timing_if_idle(ctx); return BALM_OK; }
    if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
  }
  int rc = sync_stream(ctx);
  if (rc) return rc;
  if (*stamp != want) { ctx->err = "balm_damping_iter: the iteration's scalars did not arrive"; return BALM_ERR_HIP; }
  return BALM_OK;
}

void drop_lm_graphs(balm_ctx *ctx) {
  for (int k = 0; k < balm_ctx::LM_GRAPHS; k++) {
    if (ctx->lm_graph[k]) hipGraphExecDestroy(ctx->lm_graph[k]);
    ctx->lm_graph[k] = nullptr; ctx->lm_graph_form[k] = -1;
  }
}

// residual-only evaluation of features [f0,f1) at the TRIAL poses -> d_scal[slot] (summed over
// ranks).  The per-feature eigen records it produces are kept (d_feat_tmp): if the step is accepted
// they are exactly what the next Hessian evaluation needs at the same poses (the reference recomputes
// them, bavoxel.hpp:331-351 after :443-457).
// send_mail (contexts without a collective transport, the LM loop, at most 256 features): the eigen kernel's one workgroup puts its sum
// into d_scal[slot] and sends the iteration's scalars and the stamp to the host mirror itself (-> wait_scalars) -- no launch behind it.
int residual_device(balm_ctx *ctx, const double *d_poses, int f0, int f1, int slot, bool defer_sum = false, bool send_mail = false) {
  if (f1 <= f0) {          // a shard the requested feature range does not reach: contributes zero
    HIP_TRY(hipMemsetAsync(ctx->d_scal + slot, 0, sizeof(double), ctx->stream));
    ctx->nr_tmp = 0;
  } else {
    Span sp(ctx, BALM_T_MOMENTS);
    launch_world_moments(ctx->stream, ctx->d_cl, d_poses, ctx->W, f0, f1, ctx->d_C);
    const EigenMail mail{ctx->d_scal, ctx->d_hscal, send_mail ? (double)++ctx->mail_seq : 0.0, slot};
    ctx->nr_tmp = launch_feature_eigen(ctx->stream, ctx->d_C, ctx->has_fix ? ctx->d_fix : nullptr, ctx->d_coe, f0, f1, ctx->d_feat_tmp,
                                       ctx->d_rpart_tmp, send_mail ? &mail : nullptr);
    if (!defer_sum) launch_sum_scalar(ctx->stream, ctx->d_rpart_tmp, ctx->nr_tmp, ctx->d_scal + slot);
  }
  HIP_TRY(hipGetLastError());          // k_world_moments: dynamic LDS above the 64 KiB default
  if (defer_sum) return BALM_OK;       // (no transport: the caller's k_scalars_mail adds the partials up on its way out)
  return hook_allreduce(ctx, ctx->d_scal + slot, 1);
}

// The LM loop's trial evaluation as ONE pass over the clusters (k_moments_factors): residual AND the factors of the trial
// poses, which an accepted step's Hessian evaluation then starts from.  OPT-IN (BALM_FUSE_TRIAL=1): measured at config 2
// (profiles/r03g_one_pass_trial.txt) the one-pass kernel takes 1.55 ms where K1 + K1b + K2 take 0.74 -- a workgroup's features
// are serialised behind a single lane's 3x3 Jacobi (~10 us; K1b runs 64 of them per wavefront), and the 62 KB of per-pose
// accumulators allow only two workgroups per CU to hide it.  Kept because it is parity-tested and halves the HBM reads of the
// clusters; see DESIGN.md 4.2 for what a version that wins would need.  One pose per lane: W <= 256.
bool fuse_trial(const balm_ctx *ctx) {
  const char *e = getenv("BALM_FUSE_TRIAL");
  return ctx->W <= 256 && ctx->F > 0 && e && e[0] == '1';
}

// BALM_FLAG_SYRK_INT8 / BALM_SYRK=int8: the dense SYRK on the INT8 matrix cores (kernels_syrk_i8.hip; opt-in, the default stays FP64 MFMA)
static bool syrk_int8_mode(const balm_ctx *ctx) {
  if (ctx->flags & BALM_FLAG_SYRK_INT8) return true;
  const char *m = getenv("BALM_SYRK");
  return m && !strcmp(m, "int8");
}
// ... where it pays: from 12 288 columns (4 096 features) and 96 poses on -- below, the FP64 product is as fast or faster (0.07 ms either way at
// 9 000 columns; a 64-pose window is 6 tiles x 8 k-slices = 48 workgroups for 256 CUs: 0.085 against 0.053 ms at 15 000 columns, measured) and a
// short sum does not average the digits' truncation (1e-10 of the largest entry at 180 columns, 1e-12 at 150 000).
// BALM_SYRK_INT8_MIN_COLS replaces both thresholds by its column count (tests: 0).
static bool syrk_int8_for(const balm_ctx *ctx, long K) {
  if (!syrk_int8_mode(ctx)) return false;
  if (const char *m = getenv("BALM_SYRK_INT8_MIN_COLS")) return K >= atol(m);
  return K >= 12288L && ctx->W >= 96;
}

// scratch of one Hessian evaluation over nf features (grown, never shrunk): every allocation an evaluation can need
// happens here, so that a rank of a sharded run can fail BEFORE the collectives start (see one_damping_iter)
int prepare_evaluate(balm_ctx *ctx, int form, int nf) {
  if (nf <= 0) return BALM_OK;
  const SyrkPlan plan = plan_syrk(ctx->ntiles, 3L * nf);
  int rc;
  size_t gcols = (size_t)plan.Kpad + 64, parts = (size_t)plan.SG * ctx->ntiles;
  if (ctx->sparse && nf == ctx->F) {
    gcols = (size_t)ctx->sp_nchunks * ctx->sp_nsteps * 4 + 64;
    parts = (size_t)ctx->sp_nitems;
  }
  {
    const double *before = ctx->d_Gt;
    if ((rc = ensure(ctx, &ctx->d_Gt, &ctx->cap_Gt, gcols * ctx->npad))) return rc;
    if (ctx->d_Gt != before) ctx->gt_dirty_cols = ~(size_t)0;        // fresh memory: nothing is known to be zero
  }
  if ((rc = ensure(ctx, &ctx->d_part, &ctx->cap_part, parts * TILE_ELEMS))) return rc;
  if (syrk_int8_for(ctx, 3L * nf) && !(ctx->sparse && nf == ctx->F)) {
    if ((rc = ensure(ctx, &ctx->d_i8, &ctx->cap_i8, syrk_i8_scratch_bytes(ctx->n, 3L * nf, nullptr)))) return rc;
    // the rows' largest |entries|, left by the factor kernels beside Gt / Gt2 (where their pose-per-lane variants run)
    if (!ctx->d_rowmax) ctx->rowmax_cur_valid = false;
    if (!ctx->d_rowmax2) ctx->rowmax_trial_valid = false;
    if ((rc = ensure(ctx, &ctx->d_rowmax, &ctx->cap_rowmax, (size_t)ctx->npad + 128))) return rc;
    if ((rc = ensure(ctx, &ctx->d_rowmax2, &ctx->cap_rowmax2, (size_t)ctx->npad + 128))) return rc;
    if ((rc = ensure(ctx, &ctx->d_rowmax_part, &ctx->cap_rowmax_part, (size_t)factors_grid(ctx->W, nf, form) * 6 * ctx->W))) return rc;
    HIP_TRY(prepare_device_syrk_i8());
  }
  if ((rc = ensure(ctx, &ctx->d_dpart, &ctx->cap_dpart, (size_t)factors_grid(ctx->W, nf, form) * DACC_MAX * ctx->W))) return rc;
  if (nf == ctx->F && fuse_trial(ctx)) {       // the trial poses' factors (same sizes; reallocation invalidates what they held)
    if (ctx->cap_Gt2 < gcols * ctx->npad || ctx->cap_dpart2 < (size_t)factors_grid(ctx->W, nf, form) * DACC_MAX * ctx->W) ctx->gt_trial_valid = false;
    if ((rc = ensure(ctx, &ctx->d_Gt2, &ctx->cap_Gt2, gcols * ctx->npad))) return rc;
    if ((rc = ensure(ctx, &ctx->d_dpart2, &ctx->cap_dpart2, (size_t)factors_grid(ctx->W, nf, form) * DACC_MAX * ctx->W))) return rc;
  }
  return BALM_OK;
}

// Hessian + gradient + residual of features [f0,f1) at device poses -> d_H, d_g, d_scal[slot]
int evaluate_device(balm_ctx *ctx, int form, const double *d_poses, int f0, int f1, int slot) {
  const int W = ctx->W, nf = f1 - f0;
  const int dacc = form == 0 ? DACC_LEFT : DACC_RIGHT;
  int rc;
  if (nf <= 0) {           // a shard the requested feature range does not reach: an all-zero payload
    HIP_TRY(hipMemsetAsync(ctx->d_red, 0, ctx->red_len * sizeof(double), ctx->stream));
    if ((rc = hook_allreduce(ctx, ctx->d_red, (long)ctx->red_len))) return rc;
    Span sp(ctx, BALM_T_ASSEMBLE);
    launch_assemble(ctx->stream, form, ctx->d_red, red_dacc_off(ctx), ctx->d_sub, ctx->ntiles, W, ctx->d_H, ctx->d_g,
                    ctx->d_red + red_r_off(ctx), ctx->d_scal + slot);
    return BALM_OK;
  }
  SyrkPlan plan = plan_syrk(ctx->ntiles, 3L * nf);
  if ((rc = prepare_evaluate(ctx, form, nf))) return rc;
  const int nblk = factors_grid(W, nf, form);
  hipStream_t s = ctx->stream;
  if (!(ctx->feat_cur_valid && f0 == 0 && f1 == ctx->F)) {
    Span sp(ctx, BALM_T_MOMENTS);
    launch_world_moments(s, ctx->d_cl, d_poses, W, f0, f1, ctx->d_C);
    ctx->nr_cur = launch_feature_eigen(s, ctx->d_C, ctx->has_fix ? ctx->d_fix : nullptr, ctx->d_coe, f0, f1, ctx->d_feat, ctx->d_rpart);
    ctx->feat_cur_valid = (f0 == 0 && f1 == ctx->F);
  }
  const int nr = ctx->nr_cur;
  const bool sparse = ctx->sparse && f0 == 0 && f1 == ctx->F;       // sub-ranges keep the dense plan and column order
  const size_t kpad = sparse ? (size_t)ctx->sp_nchunks * ctx->sp_nsteps * 4 : (size_t)plan.Kpad;
  if (!(ctx->gt_cur_valid && ctx->feat_cur_valid && f0 == 0 && f1 == ctx->F)) {
    Span sp(ctx, BALM_T_FACTORS);
    // The Gt columns the factor kernel does not write -- [3 nf, Kpad + 64: the k-ring's prefetch overrun) -- and the row padding
    // must be zero.  The kernel never writes either, so they are zeroed when they can be dirty: once per (re)allocation, and
    // when a narrower evaluation (a feature sub-range) follows a wider one -- not by two memsets per evaluation.
    const size_t k0 = (size_t)3 * nf, k1 = kpad + 64;
    if (ctx->gt_dirty_cols == ~(size_t)0) {
      HIP_TRY(hipMemsetAsync(ctx->d_Gt, 0, ctx->cap_Gt * sizeof(double), s));
      ctx->gt_dirty_cols = 0;
    } else if (ctx->gt_dirty_cols > k0) {
      HIP_TRY(hipMemsetAsync(ctx->d_Gt + k0 * ctx->npad, 0, (std::max(ctx->gt_dirty_cols, k1) - k0) * ctx->npad * sizeof(double), s));
    }
    ctx->gt_dirty_cols = k0;
    ctx->rowmax_cur_valid = launch_factors(s, form, ctx->d_cl, d_poses, ctx->d_feat, W, ctx->npad, f0, f1, ctx->d_Gt, ctx->d_dpart, nblk,
                                           sparse ? ctx->d_slot : nullptr, (!sparse && syrk_int8_for(ctx, 3L * nf)) ? ctx->d_rowmax : nullptr, ctx->d_rowmax_part);
    ctx->gt_cur_valid = false;                // (set by the LM loop only, when an accepted trial's factors become current)
  }
  // the moments / factor kernels ask for up to 150 KB of dynamic LDS (above the 64 KiB default: granted per device by
  // prepare_device_accum); a refused launch must surface here, not as stale results at the next synchronisation
  HIP_TRY(hipGetLastError());
  // BALM_SYRK=int8 (opt-in, round 6): the dense product on the INT8 matrix cores by error-free slicing (kernels_syrk_i8.hip); it leaves ONE
  // split-K slice in d_part
  const bool int8 = !sparse && syrk_int8_for(ctx, 3L * nf);      // (its scratch: prepare_evaluate)
  {
    Span sp(ctx, BALM_T_SYRK);
    if (sparse) launch_syrk_sparse(s, ctx->d_Gt, ctx->npad, ctx->d_jobs, ctx->d_items, ctx->d_chunk_ids, ctx->sp_nsteps, ctx->sp_nitems, ctx->d_part);
    else if (int8) {
      if (launch_syrk_i8(s, ctx->d_Gt, ctx->npad, ctx->n, 3L * nf, ctx->d_sub, ctx->ntiles, ctx->d_i8, ctx->d_part,
                         ctx->rowmax_cur_valid ? ctx->d_rowmax : nullptr)) { ctx->err = "INT8 SYRK: unsupported size"; return BALM_ERR_ARG; }
      plan.SG = 1;
    }
    else launch_syrk(s, ctx->d_Gt, ctx->npad, ctx->ntiles, ctx->d_jobs, plan, ctx->d_part);
  }
  {
    Span sp(ctx, BALM_T_ASSEMBLE);
    launch_reduce(s, ctx->d_part, plan.SG, (long)ctx->ntiles * TILE_ELEMS, ctx->d_dpart, nblk, dacc * W,
                  ctx->d_rpart, nr, ctx->d_red, red_dacc_off(ctx), red_r_off(ctx), sparse ? ctx->d_csr : nullptr);
  }
  if ((rc = hook_allreduce(ctx, ctx->d_red, (long)ctx->red_len))) return rc;
  {
    Span sp(ctx, BALM_T_ASSEMBLE);
    launch_assemble(s, form, ctx->d_red, red_dacc_off(ctx), ctx->d_sub, ctx->ntiles, W, ctx->d_H, ctx->d_g, ctx->d_red + red_r_off(ctx),
                    ctx->d_scal + slot);
  }
  return BALM_OK;
}

// The trial evaluation of the LM loop, fused: residual at the trial poses -> d_scal[slot] (summed over ranks), eigen records ->
// d_feat_tmp, AND the factors of the trial poses -> d_Gt2 / d_dpart2.  All features of this context (shard).
int trial_device(balm_ctx *ctx, int form, const double *d_poses, int slot) {
  const int W = ctx->W, F = ctx->F;
  hipStream_t s = ctx->stream;
  const SyrkPlan plan = plan_syrk(ctx->ntiles, 3L * F);
  const bool sparse = ctx->sparse;
  const size_t kpad = sparse ? (size_t)ctx->sp_nchunks * ctx->sp_nsteps * 4 : (size_t)plan.Kpad;
  const int nblk = factors_grid(W, F, form);
  {
    Span sp(ctx, BALM_T_FACTORS);
    const size_t k0 = (size_t)3 * F, k1 = kpad + 64;
    HIP_TRY(hipMemsetAsync(ctx->d_Gt2 + k0 * ctx->npad, 0, (k1 - k0) * ctx->npad * sizeof(double), s));
    if (ctx->npad > ctx->n)
      HIP_TRY(hipMemset2DAsync(ctx->d_Gt2 + ctx->n, (size_t)ctx->npad * sizeof(double), 0,
                               (size_t)(ctx->npad - ctx->n) * sizeof(double), k0, s));
    ctx->nr_tmp = launch_moments_factors(s, form, ctx->d_cl, d_poses, ctx->has_fix ? ctx->d_fix : nullptr, ctx->d_coe, W, ctx->npad, F,
                                         ctx->d_Gt2, ctx->d_dpart2, nblk, sparse ? ctx->d_slot : nullptr, ctx->d_feat_tmp, ctx->d_rpart_tmp,
                                         (!sparse && syrk_int8_for(ctx, 3L * F) && ctx->d_rowmax2) ? ctx->d_rowmax2 : nullptr, ctx->d_rowmax_part);
    ctx->rowmax_trial_valid = !sparse && syrk_int8_for(ctx, 3L * F) && ctx->d_rowmax2;
  }
  HIP_TRY(hipGetLastError());
  {
    Span sp(ctx, BALM_T_MOMENTS);
    launch_sum_scalar(s, ctx->d_rpart_tmp, ctx->nr_tmp, ctx->d_scal + slot);
  }
  ctx->gt_trial_valid = true;
  return hook_allreduce(ctx, ctx->d_scal + slot, 1);
}

int read_scalars(balm_ctx *ctx) {
  HIP_TRY(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  return sync_stream(ctx);
}

// host bookkeeping after a feature table is known on the host: planes per pose (precheck at bavoxel.hpp:1071-1085),
// the work model, and the degenerate inputs the evaluators cannot digest: a feature nobody observes and without a
// fix cluster has NN = 0 (1/NN poisons H, g and the residual of the whole window), a negative weight has no
// real square-root scaling (the Hessian's rank-3 factors carry sqrt(2 coe)).
// obs[a * W + i] != 0: pose i observes feature a (obs_of: from a host table; launch_obs_mask: from a table built on the device)
std::vector<unsigned char> obs_of(const double *clusters, int F, int W) {
  std::vector<unsigned char> obs((size_t)F * W);
  parallel_ranges(obs.size(), (size_t)1 << 16, [&](size_t lo, size_t hi) {
    for (size_t t = lo; t < hi; t++) obs[t] = clusters[t * 10 + 9] != 0 ? 1 : 0;
  });
  return obs;
}

int feature_bookkeeping(balm_ctx *ctx, int F, const unsigned char *obs, const double *fix, const double *coeffs) {
  const int W = ctx->W;
  ctx->planes_per_pose.assign(W, 0);
  double S = 0, B = 0;
  int bad_a = F, bad_rc = BALM_OK;          // the first offending feature, whichever thread meets it
  std::mutex mu;
  parallel_ranges((size_t)F, (size_t)(65536 / W + 1), [&](size_t lo, size_t hi) {
    std::vector<int> ppp((size_t)W, 0);
    double s = 0, b = 0;
    int my_bad = F, my_rc = BALM_OK;
    for (size_t a = lo; a < hi; a++) {
      int na = 0;
      const unsigned char *oa = obs + a * W;
      for (int i = 0; i < W; i++)
        if (oa[i]) { ppp[(size_t)i]++; na++; }
      if (my_rc == BALM_OK) {
        if (na == 0 && !(fix && fix[a * 10 + 9] != 0)) { my_bad = (int)a; my_rc = BALM_ERR_NUMERIC; }
        else if (!(coeffs[a] >= 0) || !std::isfinite(coeffs[a])) { my_bad = (int)a; my_rc = BALM_ERR_ARG; }
      }
      s += na; b += 0.5 * na * (na + 1.0);
    }
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < W; i++) ctx->planes_per_pose[(size_t)i] += ppp[(size_t)i];
    S += s; B += b;
    if (my_rc != BALM_OK && my_bad < bad_a) { bad_a = my_bad; bad_rc = my_rc; }
  });
  if (bad_rc == BALM_ERR_NUMERIC) {
    ctx->err = "feature " + std::to_string(bad_a) + " has no observation and no fix cluster (zero point count)";
    return BALM_ERR_NUMERIC;
  }
  if (bad_rc == BALM_ERR_ARG) {
    ctx->err = "feature " + std::to_string(bad_a) + " has a negative or non-finite weight";
    return BALM_ERR_ARG;
  }
  ctx->work_S = S; ctx->work_B = B;       // (integers below 2^53: the sum does not depend on the split)
  return BALM_OK;
}

}  // namespace

extern "C" {

const char *balm_version(void) { return "balm_hip 0.6.0 (gfx950)"; }
int balm_abi_version(void) { return BALM_ABI_VERSION; }

const char *balm_last_error(balm_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static void one_destroy(balm_ctx *ctx);

balm_ctx *balm_create(int win_size, int device, int flags) {
  if (win_size < 1 || win_size > MAX_W) return nullptr;
  cold_mark("balm_create: enter");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  cold_mark("balm_create: hipGetDeviceCount + hipSetDevice");
  warm_start(device);               // pinned ring + the other translation units' code objects, beside the rest of this function
  if (prepare_device_accum() != hipSuccess || prepare_device_cov() != hipSuccess) return nullptr;
  cold_mark("balm_create: prepare_device (function attributes -> code objects)");
  balm_ctx *ctx = new balm_ctx();
  ctx->W = win_size;
  ctx->n = 6 * win_size;
  ctx->npad = (ctx->n + TILE - 1) / TILE * TILE;
  ctx->T = ctx->npad / TILE;
  ctx->nA = (ctx->n + NB - 1) / NB * NB;
  ctx->device = device;
  ctx->flags = flags;
  ctx->timer.on = (flags & BALM_FLAG_TIMING) != 0;
  auto fail = [&]() -> balm_ctx * { one_destroy(ctx); return nullptr; };
  // The SYRK's jobs per k-slice, each 25 accumulator sub-tiles (16x16) = one wavefront:
  //   the off-diagonal 80x80 tiles (I < J), in shells of growing J: a prefix of the list touches few row blocks.
  //     An XCD has 128 wave slots for the jobs of a k-slice, so every "generation" of waves holds the tail of one
  //     slice and the head of the next; the head's rows are fetched for it alone, and a compact head needs fewer
  //     of them (2.29 GB vs 2.63 GB fetched per launch with an I-major order);
  //   the diagonal blocks: five consecutive ones have 5 x 15 = 75 upper sub-tiles = three "mixed" jobs of 25
  //     (csrc/gen/gen_syrk_asm.py), placed after the shell of their last block; blocks left over (T mod 5) run as
  //     full tiles whose lower half is ignored.
  // jobs[4 j] = {type 0 | 1..3, I or first block of the group, J, 0}; sub[25 j + t] = (R << 16) | C, the global
  // 16-row sub-tile coordinates of accumulator tile t (what the assemble kernels need).
  std::vector<int> jobs, sub;
  {
    const int T = ctx->T, ngroups = T / 5;
    std::vector<std::array<int, 3>> mixed[4];           // (block offset, r, c) lists, as the generator builds them
    std::vector<std::array<int, 2>> pairs;
    for (int r = 0; r < TM; r++) for (int c = r; c < TM; c++) pairs.push_back({r, c});
    for (int k = 0; k < 15; k++) mixed[1].push_back({0, pairs[k][0], pairs[k][1]});
    for (int k = 0; k < 10; k++) mixed[1].push_back({1, pairs[k][0], pairs[k][1]});
    for (int k = 10; k < 15; k++) mixed[2].push_back({1, pairs[k][0], pairs[k][1]});
    for (int k = 0; k < 15; k++) mixed[2].push_back({2, pairs[k][0], pairs[k][1]});
    for (int k = 0; k < 5; k++) mixed[2].push_back({3, pairs[k][0], pairs[k][1]});
    for (int k = 5; k < 15; k++) mixed[3].push_back({3, pairs[k][0], pairs[k][1]});
    for (int k = 0; k < 15; k++) mixed[3].push_back({4, pairs[k][0], pairs[k][1]});
    auto regular = [&](int I, int J) {
      jobs.insert(jobs.end(), {0, I, J, 0});
      for (int t = 0; t < 25; t++) sub.push_back(((TM * I + t / TM) << 16) | (TM * J + t % TM));
    };
    for (int m = 0; m < T; m++) {
      for (int i = 0; i < m; i++) regular(i, m);
      if (m >= TM * ngroups) regular(m, m);
      else if (m % TM == TM - 1) {
        const int b = m - (TM - 1);
        for (int v = 1; v <= 3; v++) {
          jobs.insert(jobs.end(), {v, b, b, 0});
          for (const auto &e : mixed[v]) sub.push_back(((TM * (b + e[0]) + e[1]) << 16) | (TM * (b + e[0]) + e[2]));
        }
      }
    }
  }
  ctx->ntiles = (int)(jobs.size() / 4);
  ctx->h_jobs = jobs;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail();
  cold_mark("balm_create: job tables + stream");
  const int W = ctx->W, n = ctx->n, nA = ctx->nA;
  ctx->red_len = (size_t)ctx->ntiles * TILE_ELEMS + (size_t)DACC_MAX * W + 2;
  if (dalloc(ctx, &ctx->d_poses, (size_t)12 * W) || dalloc(ctx, &ctx->d_poses_tmp, (size_t)12 * W) ||
      dalloc(ctx, &ctx->d_red, ctx->red_len) || dalloc(ctx, &ctx->d_jobs, jobs.size()) || dalloc(ctx, &ctx->d_sub, sub.size()) ||
      dalloc(ctx, &ctx->d_H, (size_t)n * n) || dalloc(ctx, &ctx->d_g, (size_t)n) ||
      dalloc(ctx, &ctx->d_A, (size_t)(2 * nA + NB) * nA) || dalloc(ctx, &ctx->d_Wp, (size_t)2 * NB * (2 * nA + NB)) ||
      dalloc(ctx, &ctx->d_dvec, (size_t)nA) || dalloc(ctx, &ctx->d_z, (size_t)nA) || dalloc(ctx, &ctx->d_x, (size_t)16 * nA) ||
      dalloc(ctx, &ctx->d_perm, (size_t)nA) || dalloc(ctx, &ctx->d_flags, (size_t)2 * (2 * (nA / NB) + 1) * (nA / NB) + (nA / NB) + 8) || dalloc(ctx, &ctx->d_minv, (size_t)2 * (nA / NB) * NB * NB) || dalloc(ctx, &ctx->d_dx, (size_t)n) ||
      dalloc(ctx, &ctx->d_scal, (size_t)16) || dalloc(ctx, &ctx->d_pre, (size_t)W + 2))
    return fail();
  if (hipHostMalloc((void **)&ctx->h_scal, (16 + 64 + 8) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return fail();
  if (hipHostGetDevicePointer((void **)&ctx->d_hscal, ctx->h_scal, 0) != hipSuccess) return fail();
  ctx->h_scal[SCAL_STAMP] = 0.0;
  if (hipMemcpy(ctx->d_jobs, jobs.data(), jobs.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ctx->d_sub, sub.data(), sub.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return fail();
  if (getenv("BALM_SOLVE_TRACE")) {
    const size_t cnt = (size_t)(2 * (nA / NB) + 1) * (nA / NB) * 6 + (size_t)(nA / NB) * 6;
    if (hipMalloc((void **)&ctx->d_trace, cnt * sizeof(long long)) != hipSuccess) return fail();
    hipMemset(ctx->d_trace, 0, cnt * sizeof(long long));
  }
  {   // a valid permutation from the start (a NaN diagonal must not leave slots unwritten, see k_rank_diag)
    std::vector<int> ident(nA);
    for (int i = 0; i < nA; i++) ident[i] = i;
    if (hipMemcpy(ctx->d_perm, ident.data(), (size_t)nA * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return fail();
  }
  if (hipMemset(ctx->d_scal, 0, 16 * sizeof(double)) != hipSuccess) return fail();
  if (hipMemset(ctx->d_red, 0, ctx->red_len * sizeof(double)) != hipSuccess) return fail();
  if (hipMemset(ctx->d_minv, 0, (size_t)2 * (nA / NB) * NB * NB * sizeof(double)) != hipSuccess) return fail();
  cold_mark("balm_create: device buffers + small copies");
  context_born(device); ctx->counted_live = true;
  return ctx;
}

static void one_destroy(balm_ctx *ctx) {
  if (!ctx) return;
  if (ctx->counted_live) { context_gone(ctx->device); ctx->counted_live = false; }
  comm_destroy(ctx);
  drop_lm_graphs(ctx);
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (ctx->window) { window_close(ctx->window); ctx->window = nullptr; }
  void *ptrs[] = {ctx->d_cl, ctx->d_fix, ctx->d_coe, ctx->d_poses, ctx->d_poses_tmp, ctx->d_C, ctx->d_feat,
                  ctx->d_Gt, ctx->d_Gt2, ctx->d_dpart2, ctx->d_part, ctx->d_dpart, ctx->d_rpart, ctx->d_feat_tmp, ctx->d_rpart_tmp, ctx->d_red, ctx->d_jobs, ctx->d_sub, ctx->d_H,
                  ctx->d_g, ctx->d_A, ctx->d_Wp, ctx->d_dvec, ctx->d_z, ctx->d_x, ctx->d_perm, ctx->d_dx, ctx->d_scal, ctx->d_arena, ctx->d_pre, ctx->d_flags, ctx->d_minv, ctx->d_macro_tab, ctx->d_trace, ctx->d_slot, ctx->d_items, ctx->d_csr, ctx->d_chunk_ids, ctx->d_stage, ctx->d_i8, ctx->d_rowmax, ctx->d_rowmax2, ctx->d_rowmax_part};
  for (void *p : ptrs) if (p) hipFree(p);
  if (ctx->h_scal) hipHostFree(ctx->h_scal);
  ctx->ring.release();
  if (ctx->amail.host) hipHostFree((void *)ctx->amail.host);
  if (ctx->amail.scan_state) hipFree(ctx->amail.scan_state);
  for (auto &sp : ctx->timer.pending) { hipEventDestroy(sp.a); hipEventDestroy(sp.b); }
  for (auto e : ctx->timer.pool) hipEventDestroy(e);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

// Block-sparse plan of hessian_syrk for the installed features (real co-visibility; dense scenes keep the dense plan).
// The reference's pair loop only visits observed poses (bavoxel.hpp:365,404-418); the dense SYRK multiplies the zero
// rows of unobserved ones: 6.5x the algorithmic flops on the shipped window (15 % block fill).  Here the columns of Gt
// (features) are ordered by which 80-row blocks they touch -- features that see the same stretch of the trajectory
// become neighbours -- and cut into chunks of C features; an item (job, chunk) exists only where the chunk touches
// the job's row blocks.  Chosen when it issues < 80 % of the dense plan's MFMAs (BALM_SYRK=dense|sparse forces).
static int build_sparse_plan(balm_ctx *ctx, int F, const unsigned char *obs) {
  ctx->sparse = false;
  const char *mode = getenv("BALM_SYRK");
  if (mode && !strcmp(mode, "dense")) return BALM_OK;
  if (syrk_int8_for(ctx, 3L * F)) return BALM_OK;      // (the INT8 product is a dense plan; below its threshold the 80 %-rule decides as ever)
  const int W = ctx->W, T = ctx->T, ntiles = ctx->ntiles;
  if (T > 128 || T <= 2 || F < 64) return BALM_OK;      // (two row blocks: three tile jobs, nothing a plan could skip -- and a 20-pose
                                                         //  sliding window installs a table per slide: the plan's host time is not free)
  struct Key { uint64_t hi, lo; int a; };
  std::vector<Key> keys((size_t)F);
  parallel_ranges((size_t)F, (size_t)(65536 / W + 1), [&](size_t a0, size_t a1) {       // (pieces of 64 KB of flags; the shipped window's 400 KB on ONE thread: 0.3 ms, measured round 6)
    for (size_t a = a0; a < a1; a++) {
      uint64_t hi = 0, lo = 0;
      const unsigned char *oa = obs + a * W;
      for (int i = 0; i < W; i++)
        if (oa[i])
          for (int b = (6 * i) / TILE; b <= (6 * i + 5) / TILE; b++) {      // a pose's six rows may straddle two blocks
            if (b < 64) hi |= 1ull << (63 - b); else lo |= 1ull << (127 - b);   // block 0 = most significant bit
          }
      keys[a] = {hi, lo, (int)a};
    }
  });
  // order: first touched block, then last touched block, then the pattern itself (measured on the shipped window against a
  // plain lexicographic order and a centre/span order: the tightest chunk unions)
  auto first_of = [&](const Key &k) { for (int b = 0; b < T; b++) if (b < 64 ? (k.hi >> (63 - b)) & 1 : (k.lo >> (127 - b)) & 1) return b; return T; };
  auto last_of = [&](const Key &k) { for (int b = T - 1; b >= 0; b--) if (b < 64 ? (k.hi >> (63 - b)) & 1 : (k.lo >> (127 - b)) & 1) return b; return -1; };
  std::vector<int> fb((size_t)F), lb((size_t)F);
  for (int a = 0; a < F; a++) { fb[(size_t)a] = first_of(keys[(size_t)a]); lb[(size_t)a] = last_of(keys[(size_t)a]); }
  std::sort(keys.begin(), keys.end(), [&](const Key &x, const Key &y) {
    if (fb[(size_t)x.a] != fb[(size_t)y.a]) return fb[(size_t)x.a] < fb[(size_t)y.a];
    if (lb[(size_t)x.a] != lb[(size_t)y.a]) return lb[(size_t)x.a] < lb[(size_t)y.a];
    return x.hi != y.hi ? x.hi > y.hi : (x.lo != y.lo ? x.lo > y.lo : x.a < y.a);
  });
  const int C = 16, nsteps = 3 * C / 4, nchunks = (F + C - 1) / C;       // 12 k-steps: three turns of the operand ring
  auto touched = [](const Key &k, int b) { return b < 64 ? (k.hi >> (63 - b)) & 1 : (k.lo >> (127 - b)) & 1; };
  std::vector<int> slot((size_t)F);
  std::vector<std::vector<int>> chunks_of((size_t)ntiles);
  long total = 0;
  for (int c = 0; c < nchunks; c++) {
    Key cm{0, 0, 0};
    for (int k = c * C; k < F && k < (c + 1) * C; k++) { cm.hi |= keys[(size_t)k].hi; cm.lo |= keys[(size_t)k].lo; slot[(size_t)keys[(size_t)k].a] = k; }
    for (int j = 0; j < ntiles; j++) {
      const int type = ctx->h_jobs[(size_t)4 * j], I = ctx->h_jobs[(size_t)4 * j + 1], J = ctx->h_jobs[(size_t)4 * j + 2];
      bool need;
      if (type == 0) need = touched(cm, I) && touched(cm, J);
      else if (type == 1) need = touched(cm, I) || touched(cm, I + 1);
      else if (type == 2) need = touched(cm, I + 1) || touched(cm, I + 2) || touched(cm, I + 3);
      else need = touched(cm, I + 3) || touched(cm, I + 4);
      if (need) { chunks_of[(size_t)j].push_back(c); total++; }
    }
  }
  const double dense_steps = (double)ntiles * ((3.0 * F + 3) / 4), sparse_steps = (double)total * nsteps;
  const bool forced = mode && !strcmp(mode, "sparse");
  if (total == 0 || (!forced && !(sparse_steps < 0.8 * dense_steps))) return BALM_OK;
  // items: every job's chunk list cut into runs of about total / 2048 chunks (two rounds of the 1024 wave slots), at least 8
  long per = total / 2048;
  if (per < 8) per = 8;
  std::vector<int> items, chunk_ids, csr((size_t)ntiles + 1, 0);
  struct Order { int first_chunk, item; };
  std::vector<Order> order;
  int run = 0;
  for (int j = 0; j < ntiles; j++) {
    csr[(size_t)j] = run;
    const std::vector<int> &L = chunks_of[(size_t)j];
    const long pieces = ((long)L.size() + per - 1) / per;
    for (long q = 0; q < pieces; q++) {
      const size_t b = (size_t)((long)L.size() * q / pieces), e = (size_t)((long)L.size() * (q + 1) / pieces);
      order.push_back({L[b], (int)(items.size() / 4)});
      items.insert(items.end(), {j, (int)chunk_ids.size(), (int)(e - b), run++});
      chunk_ids.insert(chunk_ids.end(), L.begin() + (long)b, L.begin() + (long)e);
    }
  }
  csr[(size_t)ntiles] = run;
  // launch order: by first chunk, so that waves running side by side walk the same stretch of Gt
  std::stable_sort(order.begin(), order.end(), [](const Order &x, const Order &y) { return x.first_chunk < y.first_chunk; });
  std::vector<int> sorted_items;
  sorted_items.reserve(items.size());
  for (const Order &o : order) sorted_items.insert(sorted_items.end(), items.begin() + 4L * o.item, items.begin() + 4L * o.item + 4);
  if (chunk_ids.empty()) chunk_ids.push_back(0);
  int rc;
  if ((rc = keep(ctx, &ctx->d_slot, &ctx->cap_slot, (size_t)F)) || (rc = keep(ctx, &ctx->d_items, &ctx->cap_items, sorted_items.size())) ||
      (rc = keep(ctx, &ctx->d_chunk_ids, &ctx->cap_chunk_ids, chunk_ids.size())) || (rc = keep(ctx, &ctx->d_csr, &ctx->cap_csr, csr.size())))
    return rc;
  // (blocking copies of a few KB each.  Stream-ordered copies out of the pageable vectors + one synchronise were measured in their place in
  //  round 6: 1.10 instead of 0.49 ms for this function on the shipped window -- the runtime's asynchronous pageable path is the slow one)
  HIP_TRY(hipMemcpy(ctx->d_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ctx->d_items, sorted_items.data(), sorted_items.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ctx->d_chunk_ids, chunk_ids.data(), chunk_ids.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ctx->d_csr, csr.data(), csr.size() * sizeof(int), hipMemcpyHostToDevice));
  ctx->sparse = true; ctx->sp_nsteps = nsteps; ctx->sp_nchunks = nchunks; ctx->sp_nitems = run; ctx->sp_steps = sparse_steps;
  return BALM_OK;
}

static int install_feature_buffers(balm_ctx *ctx, int F, const double *fix, const double *coeffs,
                                   hipMemcpyKind kind = hipMemcpyHostToDevice) {       // (DeviceToDevice: a table built on the device)
  int rc;
  drop_lm_graphs(ctx);
  ctx->has_fix = fix != nullptr;
  if (fix) {
    if ((rc = keep(ctx, &ctx->d_fix, &ctx->cap_fix, (size_t)F * 10))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->d_fix, fix, (size_t)F * 10 * sizeof(double), kind, ctx->stream));
  }
  if ((rc = keep(ctx, &ctx->d_coe, &ctx->cap_coe, (size_t)F))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->d_coe, coeffs, (size_t)F * sizeof(double), kind, ctx->stream));
  if ((rc = keep(ctx, &ctx->d_C, &ctx->cap_C, (size_t)F * 10))) return rc;
  if ((rc = keep(ctx, &ctx->d_feat, &ctx->cap_feat, (size_t)F * FEAT_STRIDE))) return rc;
  if ((rc = keep(ctx, &ctx->d_feat_tmp, &ctx->cap_feat_tmp, (size_t)F * FEAT_STRIDE))) return rc;
  if ((rc = keep(ctx, &ctx->d_rpart, &ctx->cap_rpart, (size_t)(F + 255) / 256 + 1024 + 1))) return rc;       // + one per workgroup of the fused trial evaluation
  if ((rc = keep(ctx, &ctx->d_rpart_tmp, &ctx->cap_rpart_tmp, (size_t)(F + 255) / 256 + 1024 + 1))) return rc;
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  ctx->F = F;
  return BALM_OK;
}

static int assoc_clusters_host(balm_ctx *ctx);

// fill(f0, f1, dst): the clusters of features [f0, f1) in the ABI's layout ((f1 - f0) * W * 10 doubles) -> dst, a pinned staging
// chunk; called from several host threads at once on disjoint ranges (host_stage.h).  The observation mask is read off the chunk
// while it is hot in the filling thread's cache.
using FillClusters = std::function<void(int, int, double *, unsigned char *)>;      // (f0, f1, dst, obs of those features: W bytes each)

// shard_books: a device of a multi context keeps ITS share of the bookkeeping (planes per pose, work model) -- the shards were cut
// before any table existed on the host (balm_set_features_cb, balm_build_clusters*); the LM loop's precheck sums the shares
static int one_set_features_fn(balm_ctx *ctx, int F, const FillClusters &fill, const double *fix, const double *coeffs, bool shard_books = false) {
  if (!ctx) return BALM_ERR_ARG;
  if (ctx->multi && F == 0) { ctx->F = 0; ctx->feat_cur_valid = false; ctx->gt_cur_valid = false; ctx->planes_per_pose.clear(); ctx->work_S = ctx->work_B = 0; return BALM_OK; }      // a shard without features
  if (F < 1 || !fill || !coeffs) { ctx->err = "balm_set_features: bad argument"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  const int W = ctx->W;
  const size_t count = (size_t)F * W * 10;
  int rc;
  if ((rc = assoc_clusters_host(ctx))) return rc;          // (an association's table not fetched yet: d_cl is about to be overwritten)
  ctx->F = 0;
  adopt_warm_ring(ctx);
  if ((rc = keep(ctx, &ctx->d_cl, &ctx->cap_cl, count))) return rc;
  if ((rc = stage_begin(ctx, count * sizeof(double)))) return rc;
  double *d_aos = stage_take<double>(ctx, count);
  std::vector<unsigned char> obs((size_t)F * W);
  const size_t unit = (size_t)W * 10 * sizeof(double);
  hipError_t e;
  {
    Span sp(ctx, BALM_T_UPLOAD);
    e = staged_upload(ctx->ring, ctx->device, ctx->stream, d_aos, count * sizeof(double), unit, [&](char *dst, size_t off, size_t len) {
      const int f0 = (int)(off / unit), f1 = f0 + (int)(len / unit);
      fill(f0, f1, reinterpret_cast<double *>(dst), obs.data() + (size_t)f0 * W);
    });
  }
  if (e == hipSuccess) launch_transpose_clusters(ctx->stream, d_aos, ctx->d_cl, F, W);
  HIP_TRY(e);
  // (the host's bookkeeping runs beside the last DMAs and the transpose)
  if ((!ctx->multi || shard_books) && (rc = feature_bookkeeping(ctx, F, obs.data(), fix, coeffs))) { hipStreamSynchronize(ctx->stream); return rc; }    // (balm_set_features on a sharded context: done once on the whole table)
  if ((rc = build_sparse_plan(ctx, F, obs.data()))) { hipStreamSynchronize(ctx->stream); return rc; }
  if ((rc = install_feature_buffers(ctx, F, fix, coeffs))) { hipStreamSynchronize(ctx->stream); return rc; }
  return sync_stream(ctx);
}

static int one_set_features(balm_ctx *ctx, int F, const double *clusters, const double *fix, const double *coeffs) {
  if (ctx && !(ctx->multi && F == 0) && !clusters) { ctx->err = "balm_set_features: bad argument"; return BALM_ERR_ARG; }
  const size_t row = ctx ? (size_t)ctx->W * 10 : 0;
  return one_set_features_fn(ctx, F, [clusters, row](int f0, int f1, double *dst, unsigned char *o) {
    const double *src = clusters + (size_t)f0 * row;
    stream_copy(dst, src, (size_t)(f1 - f0) * row * sizeof(double));      // (streaming stores: the mask is read off the SOURCE)
    const size_t cnt = (size_t)(f1 - f0) * (row / 10);
    for (size_t t = 0; t < cnt; t++) o[t] = src[t * 10 + 9] != 0 ? 1 : 0;
  }, fix, coeffs);
}

// The caller's points of balm_build_clusters: flat arrays (xyz, feat_id, pose_id), or `planes` = its own per-plane containers
// (balm_build_clusters_planes: element stride + the byte offset of the float that holds the observing pose), which the pool
// packs into 16-byte records and the device expands.
// feat_base: the caller's feature index of this context's feature 0 (a shard of a multi context builds ITS features from its stretch
// of the caller's points: the ids are rebased on the device).
static int one_build_clusters(balm_ctx *ctx, int F, const float *xyz, const int *feat_id, const int *pose_id, long n_pts,
                        const double *fix, const double *coeffs, double *clusters_out, const StridedPoints *planes = nullptr, int feat_base = 0) {
  if (!ctx) return BALM_ERR_ARG;
  if (planes) n_pts = planes->total();
  if (F < 1 || n_pts < 0 || (!planes && (!xyz || !feat_id || !pose_id)) || !coeffs || (planes && planes->n != F)) {
    ctx->err = "balm_build_clusters: bad argument"; return BALM_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  const int W = ctx->W;
  const size_t count = (size_t)F * W * 10;
  int rc;
  if ((rc = assoc_clusters_host(ctx))) return rc;
  ctx->F = 0;
  adopt_warm_ring(ctx);
  if ((rc = keep(ctx, &ctx->d_cl, &ctx->cap_cl, count))) return rc;
  const size_t np1 = (size_t)(n_pts ? n_pts : 1);
  if ((rc = stage_begin(ctx, np1 * 20 + sizeof(double) * std::max(count, planes ? np1 * 2 + (size_t)F + 33 : (size_t)0)))) return rc;
  float *d_xyz = stage_take<float>(ctx, np1 * 3);
  int *d_f = stage_take<int>(ctx, np1), *d_p = stage_take<int>(ctx, np1);
  double *d_aos = stage_take<double>(ctx, std::max(count, planes ? np1 * 2 + (size_t)F + 33 : (size_t)0));
  hipError_t e = hipSuccess;
  if (planes) {
    // 16-byte records land where the table's host copy is staged later (d_aos is free until then): xyz / pose unpacked, plane index
    // expanded from the counts
    float *d_rec = reinterpret_cast<float *>(d_aos);
    long *d_first = reinterpret_cast<long *>(reinterpret_cast<char *>(d_aos) + ((np1 * 16 + 255) & ~(size_t)255));
    {
      Span sp(ctx, BALM_T_UPLOAD);
      e = staged_points(ctx->ring, ctx->device, ctx->stream, d_rec, *planes);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_first, planes->first.data(), ((size_t)F + 1) * sizeof(long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      launch_unpack_xyzw(ctx->stream, d_rec, n_pts, d_xyz, d_p);
      launch_expand_ids(ctx->stream, d_first, F, n_pts, d_f);      // (d_aos is reused further down the same stream)
    }
  } else {
    Span sp(ctx, BALM_T_UPLOAD);
    if (e == hipSuccess) e = staged_copy(ctx->ring, ctx->device, ctx->stream, d_xyz, xyz, (size_t)n_pts * 3 * sizeof(float));
    if (e == hipSuccess) e = staged_copy(ctx->ring, ctx->device, ctx->stream, d_f, feat_id, (size_t)n_pts * sizeof(int));
    if (e == hipSuccess) e = staged_copy(ctx->ring, ctx->device, ctx->stream, d_p, pose_id, (size_t)n_pts * sizeof(int));
    if (e == hipSuccess && feat_base != 0) launch_rebase_ids(ctx->stream, d_f, n_pts, feat_base);
  }
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_cl, 0, count * sizeof(double), ctx->stream);
  int *d_flag = reinterpret_cast<int *>(ctx->d_scal + 8);           // a spare device scalar slot
  if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream);
  if (e == hipSuccess) {
    {
      Span sp(ctx, BALM_T_BUILD);
      launch_build_clusters(ctx->stream, d_xyz, d_f, d_p, n_pts, F, W, ctx->d_cl, d_flag);
    }
    int unsorted = 0;
    e = hipMemcpyAsync(&unsorted, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && unsorted) {       // points not grouped by (feature, pose): the order-free build
      e = hipMemsetAsync(ctx->d_cl, 0, count * sizeof(double), ctx->stream);
      if (e == hipSuccess) {
        Span sp(ctx, BALM_T_BUILD);
        launch_build_clusters_any(ctx->stream, d_xyz, d_f, d_p, n_pts, F, W, ctx->d_cl);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
  }
  HIP_TRY(e);
  // the host's bookkeeping (planes per pose, work model, sparse plan) needs one byte per (feature, pose); the table itself
  // crosses PCIe only if the caller asked for a copy
  std::vector<unsigned char> obs((size_t)F * W);
  {
    unsigned char *d_obs = reinterpret_cast<unsigned char *>(d_f);       // (the ids are consumed; 4 n_pts bytes -- but F * W may exceed them)
    if (obs.size() > np1 * sizeof(int)) d_obs = reinterpret_cast<unsigned char *>(d_aos);
    launch_obs_mask(ctx->stream, ctx->d_cl, F, W, d_obs);
    e = hipMemcpyAsync(obs.data(), d_obs, obs.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    HIP_TRY(e);
  }
  if (clusters_out) {
    launch_soa_to_aos(ctx->stream, ctx->d_cl, d_aos, F, W);
    e = hipMemcpyAsync(clusters_out, d_aos, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    HIP_TRY(e);
  }
  if ((rc = feature_bookkeeping(ctx, F, obs.data(), fix, coeffs))) return rc;
  if ((rc = build_sparse_plan(ctx, F, obs.data()))) return rc;
  if ((rc = install_feature_buffers(ctx, F, fix, coeffs))) return rc;
  return sync_stream(ctx);
}

void balm_voxel_defaults(balm_voxel_opts *o) {
  if (!o) return;
  o->voxel_size = 1.0;                                        // bavoxel.hpp:15
  o->eigen_thr[0] = 1.0f / 16; o->eigen_thr[1] = 1.0f / 16; o->eigen_thr[2] = 1.0f / 9;   // benchmark_realworld.cpp:183-185
  o->min_ps = 15; o->layer_limit = 2; o->min_observers = 2; o->fix_frames = 0;
  o->max_plane_dist = 0; o->max_lambda21 = 0; o->max_lambda0 = 0;
  o->want_point_features = 0;
  o->fix_point_limit = 50; o->defer_recut = 0;              // bavoxel.hpp:793
}

// the feature table an association left on the device ([F][W][10] clusters, weights, fix clusters, layers, optionally the
// feature of every point) becomes the context's feature table; the device arrays are freed if they are the caller's to free
static int install_associated(balm_ctx *ctx, int F, double *d_out, double *d_coe, double *d_fix, int *d_lay, int *d_pf, long n_pts,
                              bool has_fix, bool owned = true) {
  const int W = ctx->W;
  const size_t count = (size_t)F * W * 10;
  // The table never leaves the device on this path (round 3): d_out -> d_cl by a transpose, weights and fix clusters device to
  // device.  The host gets what its bookkeeping needs -- one byte per (feature, pose), the weights, fix clusters and layers --
  // and the clusters themselves only if somebody asks (balm_get_features fetches them from d_cl): 32 MB of pageable D2H per
  // association of the shipped window, 3 of balm_associate's 10.5 ms through the C ABI.
  ctx->assoc_clusters.clear(); ctx->assoc_cl_on_device = false;
  ctx->assoc_coeffs.resize(F); ctx->assoc_layer.resize(F); ctx->assoc_fix.resize((size_t)F * 10);
  if (d_pf) ctx->assoc_point_feat.resize((size_t)n_pts);
  std::vector<unsigned char> obs((size_t)F * W);
  hipError_t e = hipSuccess;
  int rc = keep(ctx, &ctx->d_cl, &ctx->cap_cl, count);
  if (!rc) rc = stage_begin(ctx, obs.size());
  if (!rc) {
    unsigned char *d_obs = stage_take<unsigned char>(ctx, obs.size());
    launch_transpose_clusters(ctx->stream, d_out, ctx->d_cl, F, W);
    launch_obs_mask(ctx->stream, ctx->d_cl, F, W, d_obs);
    e = hipMemcpyAsync(obs.data(), d_obs, obs.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->assoc_coeffs.data(), d_coe, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->assoc_fix.data(), d_fix, (size_t)F * 10 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->assoc_layer.data(), d_lay, (size_t)F * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && d_pf)
      e = hipMemcpyAsync(ctx->assoc_point_feat.data(), d_pf, (size_t)n_pts * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = BALM_ERR_HIP;
  }
  cold_mark("install: table transposed, obs / weights / fix / layers on the host");
  if (!rc) rc = feature_bookkeeping(ctx, F, obs.data(), has_fix ? ctx->assoc_fix.data() : nullptr, ctx->assoc_coeffs.data());
  cold_mark("install: bookkeeping");
  if (!rc) rc = build_sparse_plan(ctx, F, obs.data());
  cold_mark("install: sparse plan");
  if (!rc) rc = install_feature_buffers(ctx, F, has_fix ? d_fix : nullptr, d_coe, hipMemcpyDeviceToDevice);
  if (!rc) rc = sync_stream(ctx);
  cold_mark("install: buffers + sync");
  if (owned) {
    hipFree(d_out); hipFree(d_coe); hipFree(d_fix); hipFree(d_lay);
    if (d_pf) hipFree(d_pf);
  }
  if (rc == BALM_ERR_HIP && e != hipSuccess) { ctx->err = hipGetErrorString(e); return rc; }
  if (rc) return rc;
  ctx->assoc_cl_on_device = true;
  return BALM_OK;
}

// the association's clusters on the host ([F][W][10], the caller's layout), fetched from d_cl the first time somebody wants them
static int assoc_clusters_host(balm_ctx *ctx) {
  if (!ctx->assoc_clusters.empty() || !ctx->assoc_cl_on_device) return BALM_OK;
  const int F = (int)ctx->assoc_coeffs.size(), W = ctx->W;
  const size_t count = (size_t)F * W * 10;
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = stage_begin(ctx, count * sizeof(double));
  if (rc) return rc;
  double *d_aos = stage_take<double>(ctx, count);
  ctx->assoc_clusters.resize(count);
  launch_soa_to_aos(ctx->stream, ctx->d_cl, d_aos, F, W);
  HIP_TRY(hipMemcpyAsync(ctx->assoc_clusters.data(), d_aos, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->assoc_cl_on_device = false;
  return BALM_OK;
}

// `scans` (balm_associate_scans): the caller's own per-scan containers instead of flat xyz / frame_id arrays -- packed by the pool
// straight into the pinned ring, the scan index of every point expanded on the device from the 177-odd counts.
static int one_associate(balm_ctx *ctx, const balm_voxel_opts *opts, const float *xyz, const int *frame_id, long n_pts,
                   const double *poses, int *F_out, long *n_root_voxels, const StridedPoints *scans = nullptr) {
  if (!ctx) return BALM_ERR_ARG;
  if (scans) n_pts = scans->total();
  if (!opts || (!scans && (!xyz || !frame_id)) || !poses || !F_out || n_pts < 1 || !(opts->voxel_size > 0) || opts->fix_frames < 0 ||
      opts->layer_limit < 0 || opts->layer_limit > 2 || opts->min_observers < 0) {
    ctx->err = "balm_associate: bad argument"; return BALM_ERR_ARG;
  }
  const int W = ctx->W, WT = W + opts->fix_frames;
  if (scans && scans->n != WT) { ctx->err = "balm_associate_scans: n_scans must be win_size + fix_frames"; return BALM_ERR_ARG; }
  if (WT > 512) { ctx->err = "balm_associate: more than 512 scans not supported"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  *F_out = 0;
  ctx->F = 0;
  ctx->assoc_clusters.clear(); ctx->assoc_coeffs.clear(); ctx->assoc_layer.clear(); ctx->assoc_fix.clear(); ctx->assoc_cl_on_device = false;
  ctx->assoc_point_feat.clear();
  float *d_xyz = nullptr; int *d_f = nullptr;
  cold_mark("associate: enter");
  double *d_out = nullptr, *d_coe = nullptr, *d_fix = nullptr, *d_pos = nullptr; int *d_lay = nullptr, *d_pf = nullptr;
  int F = 0; long nroots = 0; int arc = 0;
  bool owned = true;
  size_t need = 0;
  if (!ctx->d_arena) {                  // first call: 160 B per point covers the per-point arrays, records and sort scratch (the shipped
    const size_t want = (size_t)n_pts * 160 + (64u << 20);      // window asks for 145.6; HBM is 288 GB) -- no overflow pieces, no regrow
    if (hipMalloc(&ctx->d_arena, want) == hipSuccess) ctx->arena_cap = want; else { ctx->d_arena = nullptr; hipGetLastError(); }
    cold_mark("associate: arena hipMalloc");
  }
  {
    int rcs = stage_begin(ctx, (size_t)n_pts * 16 + (size_t)12 * WT * sizeof(double) + ((size_t)WT + 1) * sizeof(long));
    if (rcs) return rcs;
  }
  d_xyz = stage_take<float>(ctx, (size_t)n_pts * 3);
  d_f = stage_take<int>(ctx, (size_t)n_pts);
  d_pos = stage_take<double>(ctx, (size_t)12 * WT);
  long *d_first = stage_take<long>(ctx, (size_t)WT + 1);
  adopt_warm_ring(ctx);
  hipError_t e = hipSuccess;
  cold_mark("associate: stage arena");
#ifdef BALM_STAGE_TRACE
  const auto tw0 = std::chrono::steady_clock::now();
#endif
  if (scans) {
    e = hipMemcpyAsync(d_first, scans->first.data(), ((size_t)WT + 1) * sizeof(long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      Span sp(ctx, BALM_T_UPLOAD);
      e = staged_points(ctx->ring, ctx->device, ctx->stream, d_xyz, *scans);
    }
  } else {
    Span sp(ctx, BALM_T_UPLOAD);
    if (e == hipSuccess) e = staged_copy(ctx->ring, ctx->device, ctx->stream, d_xyz, xyz, (size_t)n_pts * 3 * sizeof(float));
#ifdef BALM_STAGE_TRACE
    fprintf(stderr, "[stage] xyz copy returned after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
#endif
    if (e == hipSuccess) e = staged_copy(ctx->ring, ctx->device, ctx->stream, d_f, frame_id, (size_t)n_pts * sizeof(int));
  }
#ifdef BALM_STAGE_TRACE
  fprintf(stderr, "[stage] both copies returned after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
  hipStreamSynchronize(ctx->stream);
  fprintf(stderr, "[stage] stream drained after %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
#endif
  cold_mark("associate: uploads enqueued (host side done)");
  if (e == hipSuccess) e = hipMemcpyAsync(d_pos, poses, (size_t)12 * WT * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    Span sp(ctx, BALM_T_VOXEL);
    AssocOpts ao{WT, opts->voxel_size, {opts->eigen_thr[0], opts->eigen_thr[1], opts->eigen_thr[2]}, opts->min_ps,
                 opts->layer_limit, opts->min_observers, opts->fix_frames, opts->max_plane_dist, opts->max_lambda21,
                 opts->max_lambda0, opts->fix_point_limit > 0 ? opts->fix_point_limit : 50, 0};
    if (!ctx->amail.host) {
      unsigned int *h = nullptr, *d = nullptr;
      if (hipHostMalloc((void **)&h, 16 * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
          hipHostGetDevicePointer((void **)&d, h, 0) == hipSuccess) {
        h[0] = h[1] = 0; ctx->amail.host = h; ctx->amail.dev = d;
      } else { if (h) hipHostFree(h); hipGetLastError(); }
    }
    if (!ctx->amail.scan_state) {                        // (without it the scans clear a scratch buffer per call)
      void *st = nullptr;
      if (hipMalloc(&st, SCAN_TILES_CAP * sizeof(unsigned long long)) == hipSuccess &&
          hipMemsetAsync(st, 0, SCAN_TILES_CAP * sizeof(unsigned long long), ctx->stream) == hipSuccess) {
        ctx->amail.scan_state = (unsigned long long *)st; ctx->amail.scan_gen = 0;
      } else { if (st) hipFree(st); hipGetLastError(); }
    }
    arc = associate_device(ctx->stream, d_xyz, scans ? nullptr : d_f, scans ? d_first : nullptr, d_pos, n_pts, ao, ctx->d_arena, ctx->arena_cap, &need, &F, &d_out, &d_coe,
                           &d_fix, &d_lay, opts->want_point_features ? &d_pf : nullptr, &nroots, &ctx->amail, &owned);
  }
  cold_mark("associate: associate_device returned");
  int rc = BALM_OK;
  if (e == hipSuccess && arc == 0 && F > 0)       // (the table may live in the arena: installed before the arena can be replaced)
    rc = install_associated(ctx, F, d_out, d_coe, d_fix, d_lay, d_pf, n_pts, opts->fix_frames > 0, owned);
  cold_mark("associate: feature table installed");
  if (need > ctx->arena_cap) {          // grow for the next call of this size
#ifdef BALM_COLD_TRACE
    fprintf(stderr, "[cold] arena: wanted %zu bytes, had %zu (%.1f B / point)\n", need, ctx->arena_cap, (double)need / (double)n_pts);
#endif
    hipStreamSynchronize(ctx->stream);
    cold_mark("associate: arena regrow: stream synchronised");
    if (ctx->d_arena) hipFree(ctx->d_arena);
    cold_mark("associate: arena regrow: hipFree");
    ctx->arena_cap = need + need / 8;
    if (hipMalloc(&ctx->d_arena, ctx->arena_cap) != hipSuccess) { ctx->d_arena = nullptr; ctx->arena_cap = 0; hipGetLastError(); }
    cold_mark("associate: arena regrow: hipMalloc");
  }
  HIP_TRY(e);
  if (arc == -3) { ctx->err = "balm_associate: frame_id out of range or non-finite point"; return BALM_ERR_ARG; }
  if (arc) { ctx->err = arc == -2 ? "balm_associate: unsupported size" : "balm_associate: device failure"; return BALM_ERR_HIP; }
  if (n_root_voxels) *n_root_voxels = nroots;
  if (F == 0) return BALM_OK;
  if (rc) return rc;
  *F_out = F;
  return BALM_OK;
}

// ---- sliding-window map (kernels_window.inc) ------------------------------------------------------------------------
static int window_rc(balm_ctx *ctx, const char *who, int rc) {
  if (rc == 0) return BALM_OK;
  ctx->err = std::string(who) + (rc == -2 ? ": window full / bad size / scans that no recut has seen" : rc == -3 ? ": non-finite point or beyond 2^20 voxels" : ": device failure");
  // a device failure (allocation, copy, launch) can strike after the call has already advanced the map's bookkeeping (slot
  // poses, node counts, the root table) but before its points and clusters are in: a retry would build on a corrupted
  // map.  The session answers BALM_ERR_STATE from here on; balm_window_open starts a fresh one.
  if (rc == -1) ctx->window_dead = true;
  return rc == -1 ? BALM_ERR_HIP : BALM_ERR_ARG;
}

static int window_alive(balm_ctx *ctx, const char *who) {
  if (!ctx->window) { ctx->err = std::string(who) + ": no open window"; return BALM_ERR_STATE; }
  if (ctx->window_dead) { ctx->err = std::string(who) + ": an earlier window call failed on the device; re-open the window (balm_window_open)"; return BALM_ERR_STATE; }
  return BALM_OK;
}

static int one_window_open(balm_ctx *ctx, const balm_voxel_opts *opts) {
  if (!ctx) return BALM_ERR_ARG;
  if (!opts || !(opts->voxel_size > 0) || opts->layer_limit < 0 || opts->layer_limit > 2 || opts->min_observers < 0 || opts->fix_frames < 0 ||
      ctx->W + opts->fix_frames > 512 || opts->max_plane_dist < 0 || opts->max_lambda21 < 0 || opts->max_lambda0 < 0 || opts->fix_point_limit < 0) {
    ctx->err = "balm_window_open: bad argument";
    return BALM_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->window) { window_close(ctx->window); ctx->window = nullptr; }
  AssocOpts ao{ctx->W, opts->voxel_size, {opts->eigen_thr[0], opts->eigen_thr[1], opts->eigen_thr[2]}, opts->min_ps, opts->layer_limit,
               opts->min_observers, opts->fix_frames, opts->max_plane_dist, opts->max_lambda21, opts->max_lambda0,
               opts->fix_point_limit > 0 ? opts->fix_point_limit : 50, opts->defer_recut != 0};
  ctx->window = window_open(ctx->stream, ao);
  ctx->window_dead = false;
  if (!ctx->window) { ctx->err = "balm_window_open: allocation failed"; return BALM_ERR_HIP; }
  return BALM_OK;
}

static int one_window_add_scan(balm_ctx *ctx, const float *xyz, long n_pts, const double *pose12) {
  if (!ctx) return BALM_ERR_ARG;
  if (int rcw = window_alive(ctx, "balm_window_add_scan")) return rcw;
  if (!xyz || !pose12 || n_pts < 1) { ctx->err = "balm_window_add_scan: bad argument"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  Span sp(ctx, BALM_T_VOXEL);
  return window_rc(ctx, "balm_window_add_scan", window_add_scan(ctx->window, xyz, n_pts, pose12, /*xyz_on_host=*/true));
}

// the same for a scan held in the caller's own container (elements `stride` bytes apart, xyz at offset 0): packed into a pinned
// chunk of the context's ring -- by the pool when the scan is big enough to pay for the wake-up -- and copied from there
static int one_window_add_scan_strided(balm_ctx *ctx, const void *points, long n_pts, size_t stride, const double *pose12) {
  if (!ctx) return BALM_ERR_ARG;
  if (int rcw = window_alive(ctx, "balm_window_add_scan")) return rcw;
  StridedPoints sp;
  const void *base[1] = {points};
  const long cnt[1] = {n_pts};
  if (!points || !pose12 || n_pts < 1 || !sp.set(1, base, cnt, stride)) { ctx->err = "balm_window_add_scan_strided: bad argument"; return BALM_ERR_ARG; }
  if (stride == 12) return one_window_add_scan(ctx, static_cast<const float *>(points), n_pts, pose12);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)n_pts * 12;
  if (bytes > PinnedRing::CHUNK) {                     // (2.7 M points: not a lidar scan; flattened the plain way)
    std::vector<float> flat((size_t)n_pts * 3);
    parallel_ranges((size_t)n_pts, (size_t)1 << 15, [&](size_t lo, size_t hi) { sp.gather(reinterpret_cast<char *>(flat.data() + 3 * lo), (long)lo, (long)(hi - lo)); });
    return one_window_add_scan(ctx, flat.data(), n_pts, pose12);
  }
  adopt_warm_ring(ctx);
  HIP_TRY(ctx->ring.init(ctx->device));
  PinnedRing &ring = ctx->ring;
  const int b = ring.pos;
  if (ring.busy[b]) { HIP_TRY(hipEventSynchronize(ring.ev[b])); ring.busy[b] = false; }
  char *dst = ring.buf[b];
  parallel_ranges((size_t)n_pts, (size_t)1 << 15, [&](size_t lo, size_t hi) {        // (pieces start on multiples of 4 points = 48 bytes only by luck: gather re-aligns)
    sp.gather(dst + 12 * lo, (long)lo, (long)(hi - lo));
  });
  Span span(ctx, BALM_T_VOXEL);
  // (window_add_scan returns after the map's mail round trips, which follow the copy on the stream: the chunk is free again)
  return window_rc(ctx, "balm_window_add_scan", window_add_scan(ctx->window, reinterpret_cast<const float *>(dst), n_pts, pose12, /*xyz_on_host=*/true));
}

static int one_window_recut(balm_ctx *ctx) {
  if (!ctx) return BALM_ERR_ARG;
  if (int rcw = window_alive(ctx, "balm_window_recut")) return rcw;
  HIP_TRY(hipSetDevice(ctx->device));
  Span sp(ctx, BALM_T_VOXEL);
  return window_rc(ctx, "balm_window_recut", window_recut(ctx->window));
}

static int one_window_marginalize(balm_ctx *ctx, int mg, const double *poses) {
  if (!ctx) return BALM_ERR_ARG;
  if (int rcw = window_alive(ctx, "balm_window_marginalize")) return rcw;
  HIP_TRY(hipSetDevice(ctx->device));
  Span sp(ctx, BALM_T_VOXEL);
  return window_rc(ctx, "balm_window_marginalize", window_marginalize(ctx->window, mg, poses));
}

static int one_window_features(balm_ctx *ctx, int *F_out) {
  if (!ctx) return BALM_ERR_ARG;
  if (int rcw = window_alive(ctx, "balm_window_features")) return rcw;
  if (!F_out) { ctx->err = "balm_window_features: bad argument"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  *F_out = 0;
  ctx->F = 0;
  ctx->assoc_clusters.clear(); ctx->assoc_coeffs.clear(); ctx->assoc_layer.clear(); ctx->assoc_fix.clear(); ctx->assoc_point_feat.clear(); ctx->assoc_cl_on_device = false;    // (with the vectors: a failed or empty result must not leave "the table is on the device" behind)
  int F = 0, *d_lay = nullptr;
  double *d_out = nullptr, *d_coe = nullptr, *d_fix = nullptr;
  int rc;
  {
    Span sp(ctx, BALM_T_VOXEL);
    rc = window_features(ctx->window, window_min_observers(ctx->window), &F, &d_out, &d_coe, &d_fix, &d_lay);
  }
  if (rc) return window_rc(ctx, "balm_window_features", rc);
  if (F == 0) return BALM_OK;
  if ((rc = install_associated(ctx, F, d_out, d_coe, d_fix, d_lay, nullptr, 0, true, /*owned=*/false))) return rc;
  *F_out = F;
  return BALM_OK;
}

int balm_get_features(balm_ctx *ctx, double *clusters, double *coeffs, int *layer) {
  if (!ctx) return BALM_ERR_ARG;
  if (ctx->assoc_coeffs.empty()) { ctx->err = "balm_get_features: no balm_associate result"; return BALM_ERR_STATE; }
  if (clusters) {
    if (int rc = assoc_clusters_host(ctx)) return rc;
    if (ctx->assoc_clusters.empty()) { ctx->err = "balm_get_features: the association's table is gone"; return BALM_ERR_STATE; }
    std::memcpy(clusters, ctx->assoc_clusters.data(), ctx->assoc_clusters.size() * sizeof(double));
  }
  if (coeffs) std::memcpy(coeffs, ctx->assoc_coeffs.data(), ctx->assoc_coeffs.size() * sizeof(double));
  if (layer) std::memcpy(layer, ctx->assoc_layer.data(), ctx->assoc_layer.size() * sizeof(int));
  return BALM_OK;
}

int balm_get_association(balm_ctx *ctx, double *fix, int *point_feature) {
  if (!ctx) return BALM_ERR_ARG;
  if (ctx->assoc_coeffs.empty()) { ctx->err = "balm_get_association: no balm_associate result"; return BALM_ERR_STATE; }
  if (point_feature && ctx->assoc_point_feat.empty()) {
    ctx->err = "balm_get_association: want_point_features was not set"; return BALM_ERR_STATE;
  }
  if (fix) std::memcpy(fix, ctx->assoc_fix.data(), ctx->assoc_fix.size() * sizeof(double));
  if (point_feature) std::memcpy(point_feature, ctx->assoc_point_feat.data(), ctx->assoc_point_feat.size() * sizeof(int));
  return BALM_OK;
}

static int one_pose_covariance(balm_ctx *ctx, const double *poses, const double *cluster_cov, double point_sigma, double *Rcov,
                         double *Rcov_raw) {
  if (!ctx) return BALM_ERR_ARG;
  if (!ctx->multi && ctx->F < 1) { ctx->err = "balm_pose_covariance: no features installed"; return BALM_ERR_STATE; }
  if (!poses || (!cluster_cov && !(point_sigma > 0))) { ctx->err = "balm_pose_covariance: bad argument"; return BALM_ERR_ARG; }
  if (ctx->W > MAX_W_LDS) { ctx->err = "balm_pose_covariance: windows above 480 poses are not supported"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  const int W = ctx->W, n = ctx->n, nA = ctx->nA, F = ctx->F;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->d_poses, poses, (size_t)12 * W * sizeof(double), hipMemcpyHostToDevice, s));
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  int rc = evaluate_device(ctx, BALM_FORM_LEFT, ctx->d_poses, 0, F, 0);     // d_H, and the eigen records in d_feat
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  if (rc) return rc;
  const SyrkPlan plan = plan_syrk(ctx->ntiles, 3L * (F > 0 ? F : 1));
  const size_t gcols = (size_t)plan.Kpad + 64, tiles = (size_t)ctx->ntiles * TILE_ELEMS;
  if ((rc = ensure(ctx, &ctx->d_Gt, &ctx->cap_Gt, 2 * gcols * ctx->npad))) return rc;
  ctx->gt_dirty_cols = ~(size_t)0;                // X and Y are about to be written all over it
  if ((rc = ensure(ctx, &ctx->d_part, &ctx->cap_part, (size_t)plan.SG * tiles))) return rc;
  const int nblk = cov_factors_grid(W, F > 0 ? F : 1);
  if ((rc = ensure(ctx, &ctx->d_dpart, &ctx->cap_dpart, (size_t)nblk * DACC_MAX * W))) return rc;
  // BALM_SYRK=int8: X X^T and Y Y^T -- 79 % of the stage -- on the INT8 matrix cores as well (the scratch is the Hessian evaluation's: same n, same K)
  const bool int8 = F > 0 && syrk_int8_for(ctx, 3L * F);
  if (int8) {
    if ((rc = ensure(ctx, &ctx->d_i8, &ctx->cap_i8, syrk_i8_scratch_bytes(ctx->n, 3L * F, nullptr)))) return rc;
    HIP_TRY(prepare_device_syrk_i8());
  }
  // scratch: [redX | redY | S (21 W)] (one all-reduce payload) | Rraw | Rcov | T0, T1 (nA x nA each)
  const size_t pay = 2 * tiles + (size_t)21 * W, nn = (size_t)n * n;
  const size_t ncc = (cluster_cov && F > 0) ? (size_t)F * W * 81 : 0;
  if ((rc = stage_begin(ctx, (pay + 2 * nn + 2 * (size_t)nA * nA + ncc) * sizeof(double)))) return rc;
  double *buf = stage_take<double>(ctx, pay + 2 * nn + 2 * (size_t)nA * nA), *d_cc = nullptr;
  double *redx = buf, *redy = buf + tiles, *sdiag = buf + 2 * tiles, *Rraw = buf + pay, *Rc = Rraw + nn,
         *T0 = Rc + nn, *T1 = T0 + (size_t)nA * nA;
  hipError_t e = hipSuccess;
  if (ncc) {
    d_cc = stage_take<double>(ctx, ncc);
    adopt_warm_ring(ctx);
    if (e == hipSuccess) { Span sp(ctx, BALM_T_UPLOAD); e = staged_copy(ctx->ring, ctx->device, s, d_cc, cluster_cov, (size_t)F * W * 81 * sizeof(double)); }
  }
  double *Gx = ctx->d_Gt, *Gy = ctx->d_Gt + gcols * ctx->npad;
  if (e == hipSuccess && F == 0) e = hipMemsetAsync(buf, 0, pay * sizeof(double), s);      // a shard without features
  if (e == hipSuccess && F > 0) {
    Span sp(ctx, BALM_T_COV);
    const size_t k0 = (size_t)3 * F;
    hipMemsetAsync(Gx + k0 * ctx->npad, 0, (gcols - k0) * ctx->npad * sizeof(double), s);
    hipMemsetAsync(Gy + k0 * ctx->npad, 0, (gcols - k0) * ctx->npad * sizeof(double), s);
    launch_cov_factors(s, ctx->d_cl, d_cc, point_sigma * point_sigma, ctx->d_poses, ctx->d_feat, W, ctx->npad, F, Gx, Gy,
                       ctx->d_dpart, nblk);
    e = hipGetLastError();             // dynamic LDS above the 64 KiB default
    if (int8) {
      if (launch_syrk_i8(s, Gx, ctx->npad, ctx->n, 3L * F, ctx->d_sub, ctx->ntiles, ctx->d_i8, ctx->d_part)) e = hipErrorInvalidValue;
      launch_cov_reduce_tiles(s, ctx->d_part, 1, (long)tiles, redx);
      if (launch_syrk_i8(s, Gy, ctx->npad, ctx->n, 3L * F, ctx->d_sub, ctx->ntiles, ctx->d_i8, ctx->d_part)) e = hipErrorInvalidValue;
      launch_cov_reduce_tiles(s, ctx->d_part, 1, (long)tiles, redy);
    } else {
      launch_syrk(s, Gx, ctx->npad, ctx->ntiles, ctx->d_jobs, plan, ctx->d_part);
      launch_cov_reduce_tiles(s, ctx->d_part, plan.SG, (long)tiles, redx);
      launch_syrk(s, Gy, ctx->npad, ctx->ntiles, ctx->d_jobs, plan, ctx->d_part);
      launch_cov_reduce_tiles(s, ctx->d_part, plan.SG, (long)tiles, redy);
    }
    launch_cov_reduce_dacc(s, ctx->d_dpart, nblk, W, sdiag);
  }
  if (e == hipSuccess) rc = hook_allreduce(ctx, buf, (long)pay);
  if (e == hipSuccess && !rc) {
    Span sp(ctx, BALM_T_COV);
    launch_cov_assemble(s, redx, redy, sdiag, ctx->d_sub, ctx->ntiles, W, Rraw);
    hipMemsetAsync(ctx->d_g, 0, (size_t)n * sizeof(double), s);
    set_damping(ctx, 0.0);
    ctx->need_minv = true;                               // M = L^-T D^+ is what the congruence below multiplies with
    launch_solve(ctx, true);                             // P H P^T = L D L^T stays in d_A / d_dvec / d_perm
    ctx->need_minv = false;
    launch_congruence_inverse(ctx, Rraw, T0, T1, Rc);
    if (Rcov) e = hipMemcpyAsync(Rcov, Rc, nn * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && Rcov_raw) e = hipMemcpyAsync(Rcov_raw, Rraw, nn * sizeof(double), hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (rc) return rc;
  HIP_TRY(e);
  collect_timing(ctx);
  return BALM_OK;
}

int balm_comm_info(balm_ctx *ctx, long *out4) {
  if (!ctx || !out4) return BALM_ERR_ARG;
  out4[0] = 1; out4[1] = 0; out4[2] = (long)ctx->red_len; out4[3] = 0;
  if (ctx->comm) {
    int cnt = -1, rk = -1;
    comm_query(ctx, &cnt, &rk);
    out4[0] = cnt; out4[1] = rk; out4[3] = 1;
  } else if (multi_is_loopback(ctx)) {
    out4[0] = ctx->multi->n; out4[1] = ctx->rank; out4[3] = 2;
  } else if (ctx->allreduce) {
    out4[0] = -1; out4[3] = 3;                      // the caller's transport: its size is the caller's to know
  }
  return BALM_OK;
}

int balm_set_allreduce(balm_ctx *ctx, balm_allreduce_fn fn, void *user) {
  if (!ctx) return BALM_ERR_ARG;
  if (fn && (ctx->multi || ctx->comm)) { ctx->err = "balm_set_allreduce: the context already has a collective transport"; return BALM_ERR_STATE; }
  ctx->allreduce = fn;
  ctx->allreduce_user = user;
  return BALM_OK;
}

static int one_evaluate(balm_ctx *ctx, int form, const double *poses, int head, int end, double *Hess, double *JacT,
                  double *residual) {
  if (!ctx) return BALM_ERR_ARG;
  if (!ctx->multi) {
    if (ctx->F < 1) { ctx->err = "balm_evaluate: no features installed"; return BALM_ERR_STATE; }
    if ((form != 0 && form != 1) || !poses || head < 0 || end > ctx->F || head >= end) {
      ctx->err = "balm_evaluate: bad argument"; return BALM_ERR_ARG;
    }
  }
  HIP_TRY(hipSetDevice(ctx->device));
  const int n = ctx->n;
  HIP_TRY(hipMemcpyAsync(ctx->d_poses, poses, (size_t)12 * ctx->W * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  int rc = evaluate_device(ctx, form, ctx->d_poses, head, end, 0);
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  if (rc) return rc;
  if (Hess) HIP_TRY(hipMemcpyAsync(Hess, ctx->d_H, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (JacT) HIP_TRY(hipMemcpyAsync(JacT, ctx->d_g, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if ((rc = read_scalars(ctx))) return rc;
  if (residual) *residual = ctx->h_scal[0];
  if (!std::isfinite(ctx->h_scal[0])) { ctx->err = "balm_evaluate: non-finite residual"; return BALM_ERR_NUMERIC; }
  return BALM_OK;
}

static int one_only_residual(balm_ctx *ctx, const double *poses, double *residual) {
  if (!ctx) return BALM_ERR_ARG;
  if (!ctx->multi && ctx->F < 1) { ctx->err = "balm_only_residual: no features installed"; return BALM_ERR_STATE; }
  if (!poses || !residual) { ctx->err = "balm_only_residual: bad argument"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(ctx->d_poses_tmp, poses, (size_t)12 * ctx->W * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int rc = residual_device(ctx, ctx->d_poses_tmp, 0, ctx->F, 1);
  if (rc) return rc;
  if ((rc = read_scalars(ctx))) return rc;
  *residual = ctx->h_scal[1];
  if (!std::isfinite(ctx->h_scal[1])) { ctx->err = "balm_only_residual: non-finite residual"; return BALM_ERR_NUMERIC; }
  return BALM_OK;
}

int balm_solve_damped(balm_ctx *ctx, const double *Hess, const double *JacT, double u, double *dxi, double *q1) {
  if (!ctx) return BALM_ERR_ARG;
  if (!Hess || !JacT || !dxi) { ctx->err = "balm_solve_damped: bad argument"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  const int n = ctx->n;
  HIP_TRY(hipMemcpyAsync(ctx->d_H, Hess, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->d_g, JacT, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    Span sp(ctx, BALM_T_SOLVE);
    int rcu = set_damping(ctx, u);
    if (rcu) return rcu;
    launch_solve(ctx, true);
  }
  HIP_TRY(hipMemcpyAsync(dxi, ctx->d_dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  int rc = read_scalars(ctx);
  if (rc) return rc;
  if (!std::isfinite(ctx->h_scal[2]) && !ctx->persistent_off && solve_timed_out(ctx)) {      // (see one_damping_iter: once, on the launch path)
    ctx->persistent_off = true;
    {
      Span sp(ctx, BALM_T_SOLVE);
      launch_solve(ctx, true);
    }
    HIP_TRY(hipMemcpyAsync(dxi, ctx->d_dx, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = read_scalars(ctx))) return rc;
  }
  if (q1) *q1 = ctx->h_scal[2];
  return BALM_OK;
}

// One LM iteration on the stream: [Hessian evaluation] -> damped solve -> trial poses -> residual at the trial poses ->
// the scalars to the pinned mirror; returns after the stream has drained.  From the second iteration on the launch
// sequence depends only on (evaluated, which pose buffer is current): it is captured once per combination and replayed
// as a hipGraph (a window of 20 poses is ~26 launches for ~0.1 ms of device work).  Not with a
// collective transport (the all-reduce is not ours to capture), not with kernel timing (events), not if a capture ever
// failed on this context (e.g. a runtime that will not capture the cooperative launch of k_ldl_fused).
static int lm_enqueue(balm_ctx *ctx, int form, bool evaluated, bool mail) {
  int rc;
  if (evaluated && (rc = evaluate_device(ctx, form, ctx->d_poses, 0, ctx->F, 0))) return rc;
  {
    Span sp(ctx, BALM_T_SOLVE);       // ... and the trial poses: k_ldl_finish applies the step it has assembled (bavoxel.hpp:1116-1126)
    launch_solve(ctx, evaluated, form, ctx->d_poses, ctx->d_poses_tmp);
  }
  ctx->gt_trial_valid = false;
  bool sum_in_mail = false;
  if (fuse_trial(ctx) && ctx->d_Gt2 && ctx->d_dpart2) {
    if ((rc = trial_device(ctx, form, ctx->d_poses_tmp, 1))) return rc;
  } else {
    sum_in_mail = mail && !has_transport(ctx) && ctx->F > 0;
    const bool eigen_mails = sum_in_mail && ctx->F <= 256;      // one workgroup of k_feature_eigen: it sends the mail itself
    if ((rc = residual_device(ctx, ctx->d_poses_tmp, 0, ctx->F, 1, sum_in_mail, eigen_mails))) return rc;
    if (eigen_mails) return BALM_OK;
  }
  if (mail) launch_scalars_mail(ctx->stream, ctx->d_scal, ctx->d_hscal, (double)++ctx->mail_seq, sum_in_mail ? ctx->d_rpart_tmp : nullptr,
                                ctx->nr_tmp, 1);      // -> wait_scalars
  else HIP_TRY(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));      // (a replayed graph cannot carry a new stamp)
  return BALM_OK;
}

static int lm_iteration(balm_ctx *ctx, int form, bool evaluated, int it) {
  // Opt-in (BALM_GRAPH=1, or "debug" for a line per capture).  Measured on the box (tools/bench_small.py,
  // profiles/r02i_small_windows_graph.txt): 0.130 -> 0.125 ms per iteration at W=20/F=20, 0.302 -> 0.301 at W=64/F=5000 --
  // the iteration is bound by the dependency latency between its ~26 short kernels on the device, not by the host's launch
  // calls, so a replayed graph buys 0-4 % and is not worth being the default.
  const char *genv = getenv("BALM_GRAPH");
  const bool disabled = !(genv && (!strcmp(genv, "1") || !strcmp(genv, "debug")));
  // windows whose factorisation is the persistent cooperative kernel are left alone: their iterations are not
  // launch-bound, and a cooperative launch inside a capture is not something every runtime takes
  const bool graphable = !disabled && ctx->graphs_ok && it >= 1 && !ctx->timer.on && !has_transport(ctx) && !ctx->multi &&
                         ctx->feat_cur_valid && ctx->F > 0 && !solve_is_persistent(ctx);
  if (graphable) {
    const bool fused = fuse_trial(ctx) && ctx->d_Gt2 && ctx->d_dpart2;       // what lm_enqueue would issue depends on all four
    const int slot = (ctx->gt_parity ? 16 : 0) | (fused ? 8 : 0) | (ctx->gt_cur_valid ? 4 : 0) | (evaluated ? 2 : 0) | ctx->parity;
    if (ctx->lm_graph[slot] && ctx->lm_graph_form[slot] != form) {
      hipGraphExecDestroy(ctx->lm_graph[slot]); ctx->lm_graph[slot] = nullptr;
    }
    if (!ctx->lm_graph[slot]) {
      hipGraph_t g = nullptr;
      ctx->u_on_device = true;               // the captured kernels read the damping from d_scal[SCAL_U]
      bool ok = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        const int rc = lm_enqueue(ctx, form, evaluated, false);
        const hipError_t e = hipStreamEndCapture(ctx->stream, &g);
        ok = rc == BALM_OK && e == hipSuccess && g != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&ctx->lm_graph[slot], g, nullptr, nullptr, 0) == hipSuccess;
      if (g) hipGraphDestroy(g);
      if (genv && !strcmp(genv, "debug")) fprintf(stderr, "balm_hip: LM graph slot %d capture %s\n", slot, ok ? "ok" : "FAILED");
      if (!ok) {
        hipGetLastError();
        ctx->lm_graph[slot] = nullptr;
        ctx->graphs_ok = false;
      } else {
        ctx->lm_graph_form[slot] = form;
      }
    }
    if (ctx->lm_graph[slot]) {
      int rcu = push_damping(ctx);
      if (rcu) return rcu;
      HIP_TRY(hipGraphLaunch(ctx->lm_graph[slot], ctx->stream));
      return sync_stream(ctx);
    }
    ctx->u_on_device = false;                // (capture failed: plain launches with the damping as an argument)
  }
  int rc = lm_enqueue(ctx, form, evaluated, true);
  if (rc) return rc;
  return wait_scalars(ctx);
}

static int one_damping_iter(balm_ctx *ctx, const balm_lm_opts *o, double *poses, balm_iter_log *log, int *n_iters) {
  if (!ctx) return BALM_ERR_ARG;
  if (!ctx->multi && ctx->F < 1) { ctx->err = "balm_damping_iter: no features installed"; return BALM_ERR_STATE; }
  if (!o || !poses || (o->form != 0 && o->form != 1) || o->max_iter < 1) {
    ctx->err = "balm_damping_iter: bad argument"; return BALM_ERR_ARG;
  }
  if (n_iters) *n_iters = 0;
  HIP_TRY(hipSetDevice(ctx->device));
  const int W = ctx->W, F = ctx->F;
  hipStream_t s = ctx->stream;
  cold_mark("damping_iter: enter");
  int rc = prepare_evaluate(ctx, o->form, F);
  cold_mark("damping_iter: prepare_evaluate (scratch allocations)");
  // bavoxel.hpp:1071-1085: every pose must see >= 20 planes (printf + exit(0) in the reference).  Sharded runs hold
  // per-shard counts: they are summed once, before the loop, in a collective that also carries an error flag, so that
  // a rank that could not allocate its scratch takes every rank out together instead of leaving them in an all-reduce.
  std::vector<double> pre((size_t)W + 2, 0.0);
  for (size_t i = 0; i < ctx->planes_per_pose.size() && i < (size_t)W; i++) pre[i] = ctx->planes_per_pose[i];
  pre[(size_t)W] = rc ? 1.0 : 0.0;
  if (has_transport(ctx)) {
    int rc2;
    if (hipMemcpyAsync(ctx->d_pre, pre.data(), pre.size() * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return BALM_ERR_HIP;
    if ((rc2 = hook_allreduce(ctx, ctx->d_pre, (long)pre.size()))) return rc2;
    if (hipMemcpyAsync(pre.data(), ctx->d_pre, pre.size() * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return BALM_ERR_HIP;
    if ((rc2 = sync_stream(ctx))) return rc2;
    if (pre[(size_t)W] != 0.0 && !rc) { ctx->err = "balm_damping_iter: another rank failed before the loop"; rc = BALM_ERR_STATE; }
  }
  if (rc) return rc;
  if (o->min_planes_per_pose > 0) {
    double mn = pre[0];
    for (int i = 1; i < W; i++) mn = pre[(size_t)i] < mn ? pre[(size_t)i] : mn;
    if (mn < o->min_planes_per_pose) {
      ctx->err = "Initial error too large. Please loose plane determination criteria for more planes. "
                 "The optimization is terminated.";
      return BALM_ERR_TOO_FEW_PLANES;
    }
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_poses, poses, (size_t)12 * W * sizeof(double), hipMemcpyHostToDevice, s));
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  double u = o->u0, v = 2, r1 = 0, r2 = 0;
  bool calc = true;
  int it = 0;
  // fault injection for tests/test_gpu_multi.py (a device thread that leaves the loop must take its peers out with it, not
  // leave them waiting): BALM_FAULT_INJECT="<rank>,<iteration>" makes that device of a sharded context fail there
  // ... and BALM_FAULT_INJECT="timeout,<iteration>" (any context): that iteration's solve finds its abort flag raised, as a wait
  // inside a persistent solve kernel that hit its poll limit leaves it (tests/test_gpu_solve.py: the retry on the launch path)
  int fault_rank = -1, fault_it = -1, timeout_it = -1;
  if (const char *fi = getenv("BALM_FAULT_INJECT")) {
    if (!strncmp(fi, "timeout,", 8)) timeout_it = atoi(fi + 8);
    else if (ctx->multi) sscanf(fi, "%d,%d", &fault_rank, &fault_it);
  }
  while (it < o->max_iter) {
    const bool evaluated = calc || o->force_hess;
    if (ctx->rank == fault_rank && it == fault_it) { ctx->err = "balm_damping_iter: injected fault"; return BALM_ERR_HIP; }
    if (it == timeout_it) { ctx->inject_solve_timeout = true; timeout_it = -1; }
    if ((rc = set_damping(ctx, u))) return rc;
    if ((rc = lm_iteration(ctx, o->form, evaluated, it))) return rc;
    double sc[3] = {ctx->h_scal[0], ctx->h_scal[1], ctx->h_scal[2]};
    if ((rc = multi_share_scalars(ctx, it, sc, 3))) return rc;       // device 0's scalars decide on every device thread
    r1 = sc[0]; r2 = sc[1];
    const double q1 = sc[2];
    double q = r1 - r2;
    if (!ctx->multi && !ctx->persistent_off && !(std::isfinite(r2) && std::isfinite(q1)) && std::isfinite(r1) && solve_timed_out(ctx)) {
      // not arithmetic: a wait inside the persistent solve kernel ran into its poll limit -- its workgroups were not all resident
      // (another process or another stream's kernel held CUs).  The Hessian of this iteration is intact; the solve is repeated on
      // the launch path (no co-residency needed), where this context then stays.
      ctx->persistent_off = true;
      drop_lm_graphs(ctx);
      ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;      // (the iteration is re-issued from its evaluation)
      calc = true;
      continue;
    }
    if (!(std::isfinite(r1) && std::isfinite(r2))) {
      ctx->err = "balm_damping_iter: non-finite residual";
      return BALM_ERR_NUMERIC;
    }
    if (log) { log[it].r1 = r1; log[it].r2 = r2; log[it].u = u; log[it].v = v; log[it].q = q; log[it].q1 = q1;
               log[it].accepted = q > 0; log[it].hess_evaluated = evaluated; }
    if (o->verbose)   // the reference's progress line, bavoxel.hpp:1132
    {
      printf("iter%d: (%lf %lf) u: %lf v: %.1lf q: %.3lf %lf %lf\n", it, r1, r2, u, v, q / q1, q1, q);
      fflush(stdout);       // progress lines interleave correctly with a host language's own output
    }
    if (q > 0) {      // bavoxel.hpp:1134-1143
      double *t = ctx->d_poses; ctx->d_poses = ctx->d_poses_tmp; ctx->d_poses_tmp = t;
      // the trial poses become current: so do their eigen records and residual partials
      t = ctx->d_feat; ctx->d_feat = ctx->d_feat_tmp; ctx->d_feat_tmp = t;
      t = ctx->d_rpart; ctx->d_rpart = ctx->d_rpart_tmp; ctx->d_rpart_tmp = t;
      ctx->nr_cur = ctx->nr_tmp;
      ctx->feat_cur_valid = true;
      ctx->gt_cur_valid = ctx->gt_trial_valid;        // the trial's factors (fused evaluation) become the current poses'
      if (ctx->gt_trial_valid) {
        std::swap(ctx->d_Gt, ctx->d_Gt2); std::swap(ctx->cap_Gt, ctx->cap_Gt2);
        std::swap(ctx->d_rowmax, ctx->d_rowmax2); std::swap(ctx->cap_rowmax, ctx->cap_rowmax2);        // (BALM_SYRK=int8: the row maxima go with their Gt)
        ctx->rowmax_cur_valid = ctx->rowmax_trial_valid;
        ctx->gt_dirty_cols = ~(size_t)0;       // (the other buffer's zero state is the trial kernel's business, not tracked)
        std::swap(ctx->d_dpart, ctx->d_dpart2); std::swap(ctx->cap_dpart, ctx->cap_dpart2);
        ctx->gt_parity ^= 1;
      }
      ctx->parity ^= 1;
      q = q / q1; v = 2; q = 1 - std::pow(2 * q - 1, 3);
      u *= (q < 1.0 / 3.0 ? 1.0 / 3.0 : q);
      calc = true;
    } else {          // :1144-1149
      u = u * v; v = 2 * v; calc = false;
    }
    it++;
    if (!o->no_stop && std::fabs(r1 - r2) / r1 < o->rel_tol) break;   // :1155
    if (!o->no_stop && o->abs_tol > 0 && std::fabs(r1 - r2) < o->abs_tol) break;   // BAs_left.hpp:1083
  }
  ctx->feat_cur_valid = false; ctx->gt_cur_valid = false;
  if (o->reanchor) launch_reanchor(s, W, ctx->d_poses);
  HIP_TRY(hipMemcpyAsync(poses, ctx->d_poses, (size_t)12 * W * sizeof(double), hipMemcpyDeviceToHost, s));
  if ((rc = sync_stream(ctx))) return rc;
  if (n_iters) *n_iters = it;
  cold_mark("damping_iter: done");
  return BALM_OK;
}

// ---- public entry points: one device, or the devices of a balm_create_multi context ---------------------------------
static balm_multi *leader_of(balm_ctx *ctx) { return (ctx && ctx->multi && ctx->rank == 0) ? ctx->multi : nullptr; }

int balm_prewarm(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { hipGetLastError(); return BALM_ERR_ARG; }
  return warm_start(device);
}

balm_ctx *balm_create_multi(int win_size, int first_device, int n_devices, int flags) {
  if (n_devices < 1 || n_devices > MAX_SHARDS) return nullptr;
  const bool loopback = (flags & BALM_FLAG_LOOPBACK_SHARDS) != 0;
  HostPool::shard_pool_threads() = n_devices > 4 ? 8 : 0;          // (pools that exist already keep their size)
  std::vector<balm_ctx *> subs;
  for (int k = 0; k < n_devices; k++) {
    balm_ctx *c = balm_create(win_size, loopback ? first_device : first_device + k, flags);
    if (!c) { for (auto *q : subs) one_destroy(q); return nullptr; }
    c->ring.pool_key = k;            // the device threads' uploads run side by side, each with its own host pool (0 = the process's)
    subs.push_back(c);
  }
  std::string err;
  balm_multi *m = multi_new(subs, loopback, &err);
  if (!m) {
    fprintf(stderr, "balm_create_multi: %s\n", err.c_str());
    for (auto *q : subs) one_destroy(q);
    return nullptr;
  }
  return subs[0];
}

void balm_destroy(balm_ctx *ctx) {
  if (!ctx) return;
  if (balm_multi *m = ctx->multi) {
    std::vector<balm_ctx *> subs = m->sub;
    for (auto *q : subs) { hipSetDevice(q->device); if (q->stream) hipStreamSynchronize(q->stream); }
    multi_delete(m);
    for (auto *q : subs) { q->multi = nullptr; one_destroy(q); }
    return;
  }
  one_destroy(ctx);
}

int balm_comm_unique_id(void *id128) {
  if (!id128) return BALM_ERR_ARG;
  return comm_unique_id(id128);
}

int balm_comm_init_rank(balm_ctx *ctx, int n_ranks, int rank, const void *id128) {
  if (!ctx) return BALM_ERR_ARG;
  if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) { ctx->err = "balm_comm_init_rank: bad argument"; return BALM_ERR_ARG; }
  if (ctx->multi || ctx->comm) { ctx->err = "balm_comm_init_rank: the context already has a collective transport"; return BALM_ERR_STATE; }
  HIP_TRY(hipSetDevice(ctx->device));
  return comm_init_rank(ctx, n_ranks, rank, id128);
}

// the whole table is checked and counted once; shard k gets the contiguous range [fbeg[k], fbeg[k+1])
static int multi_set_features(balm_ctx *ctx, balm_multi *m, int F, const double *clusters, const double *fix, const double *coeffs) {
  if (F < 1 || !clusters || !coeffs) { ctx->err = "balm_set_features: bad argument"; return BALM_ERR_ARG; }
  m->F = 0;
  m->books_per_shard = false;
  for (size_t k = 1; k < m->sub.size(); k++) { m->sub[k]->planes_per_pose.clear(); m->sub[k]->work_S = m->sub[k]->work_B = 0; }     // (shares of an earlier sharded install)
  const std::vector<unsigned char> obs = obs_of(clusters, F, ctx->W);
  int rc = feature_bookkeeping(ctx, F, obs.data(), fix, coeffs);
  if (rc) return rc;
  // contiguous shards of equal COST: a feature costs its observed pose pairs n_a (n_a + 1) / 2 (the block-sparse SYRK
  // plan skips what it does not observe; with dense co-visibility every feature costs the same and this is an equal split)
  // plus a constant for its moments / factor passes
  {
    const int W = ctx->W;
    std::vector<double> cum((size_t)F + 1, 0.0);
    parallel_ranges((size_t)F, (size_t)(65536 / W + 1), [&](size_t lo, size_t hi) {
      for (size_t a = lo; a < hi; a++) {
        int na = 0;
        const unsigned char *oa = obs.data() + a * W;
        for (int i = 0; i < W; i++) na += oa[i];
        cum[a + 1] = 0.5 * na * (na + 1.0) + 4.0 * na + 1.0;
      }
    });
    for (int a = 0; a < F; a++) cum[(size_t)a + 1] += cum[(size_t)a];
    m->fbeg[0] = 0;
    for (int k = 1; k < m->n; k++) {
      const double target = cum[(size_t)F] * k / m->n;
      int f = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
      if (f < m->fbeg[(size_t)k - 1]) f = m->fbeg[(size_t)k - 1];
      if (f > F) f = F;
      m->fbeg[(size_t)k] = f;
    }
    m->fbeg[(size_t)m->n] = F;
  }
  const size_t W = (size_t)ctx->W;
  rc = multi_run(m, [&](int k) {
    const int f0 = m->fbeg[(size_t)k], nf = m->fbeg[(size_t)k + 1] - f0;
    return one_set_features(m->sub[(size_t)k], nf, clusters + (size_t)f0 * W * 10, fix ? fix + (size_t)f0 * 10 : nullptr, coeffs + f0);
  });
  if (rc) { if (ctx->err.empty()) ctx->err = "balm_set_features: a device failed"; return rc; }
  m->F = F;
  return BALM_OK;
}

int balm_set_features(balm_ctx *ctx, int F, const double *clusters, const double *fix, const double *coeffs) {
  if (balm_multi *m = leader_of(ctx)) return multi_set_features(ctx, m, F, clusters, fix, coeffs);
  return one_set_features(ctx, F, clusters, fix, coeffs);
}

// contiguous shards of (about) equal cost: fbeg[k] = the first feature whose cumulated cost reaches k / n of the total
static void multi_cut(balm_multi *m, int F, const std::vector<double> &cum /* [F + 1], cum[0] = 0 */) {
  m->fbeg[0] = 0;
  for (int k = 1; k < m->n; k++) {
    const double target = cum[(size_t)F] * k / m->n;
    int f = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
    if (f < m->fbeg[(size_t)k - 1]) f = m->fbeg[(size_t)k - 1];
    if (f > F) f = F;
    m->fbeg[(size_t)k] = f;
  }
  m->fbeg[(size_t)m->n] = F;
}

static int multi_done(balm_ctx *ctx, balm_multi *m, int F, int rc, const char *who) {
  if (rc) { if (ctx->err.empty()) ctx->err = std::string(who) + ": a device failed"; for (auto *q : m->sub) if (!q->err.empty()) { ctx->err = q->err; break; } return rc; }
  m->F = F;
  m->books_per_shard = true;
  return BALM_OK;
}

int balm_set_features_cb(balm_ctx *ctx, int F, balm_fill_clusters_fn fill, void *user, const double *fix, const double *coeffs) {
  if (!ctx) return BALM_ERR_ARG;
  if (F < 1 || !fill || !coeffs) { ctx->err = "balm_set_features_cb: bad argument"; return BALM_ERR_ARG; }
  if (balm_multi *m = leader_of(ctx)) {
    // Sharded, in two passes over the CALLER's table and none over a copy of it (round 5 flattened all F x W x 80 bytes into a host
    // vector first: 3.2 GB at BASELINE configs[3]): (1) every feature's observation count, read through `fill` into a scratch row per
    // pool thread -- the cost-balanced cut needs them before any shard can leave; (2) every device thread pulls ITS features straight
    // into its own pinned ring, the shards side by side.
    const size_t Wc = (size_t)ctx->W, row = Wc * 10;
    m->F = 0;
    std::vector<double> cum((size_t)F + 1, 0.0);
    parallel_ranges((size_t)F, (size_t)(32768 / row + 1), [&](size_t lo, size_t hi) {
      const size_t blk = std::max<size_t>(1, (size_t)65536 / (row * sizeof(double)));      // features per call of fill: a 64 KB scratch
      std::vector<double> scratch(blk * row);
      for (size_t a0 = lo; a0 < hi; a0 += blk) {
        const size_t a1 = std::min(hi, a0 + blk);
        fill(user, (int)a0, (int)a1, scratch.data());
        for (size_t a = a0; a < a1; a++) {
          int na = 0;
          const double *r = scratch.data() + (a - a0) * row;
          for (size_t i = 0; i < Wc; i++) na += r[i * 10 + 9] != 0 ? 1 : 0;
          cum[a + 1] = 0.5 * na * (na + 1.0) + 4.0 * na + 1.0;
        }
      }
    });
    for (int a = 0; a < F; a++) cum[(size_t)a + 1] += cum[(size_t)a];
    multi_cut(m, F, cum);
    int rc = multi_run(m, [&](int k) {
      const int f0 = m->fbeg[(size_t)k], nf = m->fbeg[(size_t)k + 1] - f0;
      return one_set_features_fn(m->sub[(size_t)k], nf, [fill, user, Wc, f0](int a0, int a1, double *dst, unsigned char *o) {
        fill(user, f0 + a0, f0 + a1, dst);
        const size_t cnt = (size_t)(a1 - a0) * Wc;
        for (size_t t = 0; t < cnt; t++) o[t] = dst[t * 10 + 9] != 0 ? 1 : 0;
      }, fix ? fix + (size_t)f0 * 10 : nullptr, coeffs + f0, /*shard_books=*/true);
    });
    return multi_done(ctx, m, F, rc, "balm_set_features_cb");
  }
  const size_t Wc = (size_t)ctx->W;
  return one_set_features_fn(ctx, F, [fill, user, Wc](int f0, int f1, double *dst, unsigned char *o) {
    fill(user, f0, f1, dst);                                             // (the caller's ordinary stores: the chunk is in this thread's cache)
    const size_t cnt = (size_t)(f1 - f0) * Wc;
    for (size_t t = 0; t < cnt; t++) o[t] = dst[t * 10 + 9] != 0 ? 1 : 0;
  }, fix, coeffs);
}

int balm_build_clusters(balm_ctx *ctx, int F, const float *xyz, const int *feat_id, const int *pose_id, long n_pts,
                        const double *fix, const double *coeffs, double *clusters_out) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_build_clusters(ctx, F, xyz, feat_id, pose_id, n_pts, fix, coeffs, clusters_out);
  if (F < 1 || n_pts < 0 || !xyz || !feat_id || !pose_id || !coeffs) { ctx->err = "balm_build_clusters: bad argument"; return BALM_ERR_ARG; }
  // Sharded: points that arrive feature by feature (feat_id ascending -- every driver of the reference pushes them that way,
  // benchmark_virtual.cpp:392-403) are cut into stretches of equal length at feature boundaries and every device builds ITS features
  // from its stretch: no whole table on the first device, no download and re-upload of it (3.2 GB each way at BASELINE configs[3]).
  {
    std::atomic<int> bad{0};
    parallel_ranges((size_t)n_pts, (size_t)1 << 20, [&](size_t lo, size_t hi) {
      int prev = lo > 0 ? feat_id[lo - 1] : 0, b = 0;
      for (size_t p = lo; p < hi; p++) { const int f = feat_id[p]; b |= (f < prev || f < 0 || f >= F) ? 1 : 0; prev = f; }
      if (b) bad.store(1, std::memory_order_relaxed);
    });
    if (!bad.load() && n_pts > 0) {
      m->F = 0;
      std::vector<long> pbeg((size_t)m->n + 1, 0);
      m->fbeg[0] = 0; pbeg[0] = 0;
      for (int k = 1; k < m->n; k++) {
        const long target = n_pts * k / m->n;
        int f = feat_id[target];                                                  // the feature the cut falls into starts the next shard
        if (f < m->fbeg[(size_t)k - 1]) f = m->fbeg[(size_t)k - 1];
        m->fbeg[(size_t)k] = f;
        pbeg[(size_t)k] = (long)(std::lower_bound(feat_id, feat_id + n_pts, f) - feat_id);
      }
      m->fbeg[(size_t)m->n] = F; pbeg[(size_t)m->n] = n_pts;
      const size_t Wc = (size_t)ctx->W;
      int rc = multi_run(m, [&](int k) {
        const int f0 = m->fbeg[(size_t)k], nf = m->fbeg[(size_t)k + 1] - f0;
        const long p0 = pbeg[(size_t)k], np = pbeg[(size_t)k + 1] - p0;
        if (nf == 0) return one_set_features(m->sub[(size_t)k], 0, nullptr, nullptr, nullptr);
        return one_build_clusters(m->sub[(size_t)k], nf, xyz + 3 * p0, feat_id + p0, pose_id + p0, np, fix ? fix + (size_t)f0 * 10 : nullptr, coeffs + f0,
                                  clusters_out ? clusters_out + (size_t)f0 * Wc * 10 : nullptr, nullptr, f0);
      });
      return multi_done(ctx, m, F, rc, "balm_build_clusters");
    }
  }
  // points in no feature order: the first device builds the whole table, then it is sharded like any other
  std::vector<double> host((size_t)F * ctx->W * 10);
  ctx->multi = nullptr;                               // as a plain context for this one call
  int rc = one_build_clusters(ctx, F, xyz, feat_id, pose_id, n_pts, fix, coeffs, host.data());
  ctx->multi = m;
  if (rc) return rc;
  if (clusters_out) std::memcpy(clusters_out, host.data(), host.size() * sizeof(double));
  return multi_set_features(ctx, m, F, host.data(), fix, coeffs);
}

int balm_associate(balm_ctx *ctx, const balm_voxel_opts *opts, const float *xyz, const int *frame_id, long n_pts,
                   const double *poses, int *F_out, long *n_root_voxels) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_associate(ctx, opts, xyz, frame_id, n_pts, poses, F_out, n_root_voxels);
  ctx->multi = nullptr;
  int rc = one_associate(ctx, opts, xyz, frame_id, n_pts, poses, F_out, n_root_voxels);
  ctx->multi = m;
  m->F = 0;
  if (rc || *F_out == 0) return rc;
  if ((rc = assoc_clusters_host(ctx))) return rc;          // the shards are cut from the host copy
  return multi_set_features(ctx, m, *F_out, ctx->assoc_clusters.data(), opts->fix_frames > 0 ? ctx->assoc_fix.data() : nullptr,
                            ctx->assoc_coeffs.data());
}

int balm_associate_scans(balm_ctx *ctx, const balm_voxel_opts *opts, int n_scans, const void *const *scan_points, const long *scan_count,
                         size_t stride_bytes, const double *poses, int *F_out, long *n_root_voxels) {
  if (!ctx) return BALM_ERR_ARG;
  StridedPoints sp;
  if (n_scans < 1 || !sp.set(n_scans, scan_points, scan_count, stride_bytes)) { ctx->err = "balm_associate_scans: bad argument"; return BALM_ERR_ARG; }
  balm_multi *m = leader_of(ctx);
  if (!m) return one_associate(ctx, opts, nullptr, nullptr, 0, poses, F_out, n_root_voxels, &sp);
  ctx->multi = nullptr;
  int rc = one_associate(ctx, opts, nullptr, nullptr, 0, poses, F_out, n_root_voxels, &sp);
  ctx->multi = m;
  m->F = 0;
  if (rc || *F_out == 0) return rc;
  if ((rc = assoc_clusters_host(ctx))) return rc;
  return multi_set_features(ctx, m, *F_out, ctx->assoc_clusters.data(), opts->fix_frames > 0 ? ctx->assoc_fix.data() : nullptr,
                            ctx->assoc_coeffs.data());
}

int balm_build_clusters_planes(balm_ctx *ctx, int F, const void *const *plane_points, const long *plane_count, size_t stride_bytes,
                               size_t pose_offset_bytes, const double *fix, const double *coeffs, double *clusters_out) {
  if (!ctx) return BALM_ERR_ARG;
  StridedPoints sp;
  if (F < 1 || !sp.set(F, plane_points, plane_count, stride_bytes, pose_offset_bytes) || pose_offset_bytes == StridedPoints::NO_AUX) {
    ctx->err = "balm_build_clusters_planes: bad argument"; return BALM_ERR_ARG;
  }
  balm_multi *m = leader_of(ctx);
  if (!m) return one_build_clusters(ctx, F, nullptr, nullptr, nullptr, 0, fix, coeffs, clusters_out, &sp);
  if (!coeffs) { ctx->err = "balm_build_clusters_planes: bad argument"; return BALM_ERR_ARG; }
  // sharded: the planes are cut into runs of (about) equal point count and every device packs, uploads and builds ITS planes
  m->F = 0;
  std::vector<double> cum((size_t)F + 1, 0.0);
  for (int a = 0; a < F; a++) cum[(size_t)a + 1] = cum[(size_t)a] + (double)plane_count[a] + 1.0;
  multi_cut(m, F, cum);
  const size_t Wc = (size_t)ctx->W;
  int rc = multi_run(m, [&](int k) {
    const int f0 = m->fbeg[(size_t)k], nf = m->fbeg[(size_t)k + 1] - f0;
    if (nf == 0) return one_set_features(m->sub[(size_t)k], 0, nullptr, nullptr, nullptr);
    StridedPoints part;
    if (!part.set(nf, plane_points + f0, plane_count + f0, stride_bytes, pose_offset_bytes)) return (int)BALM_ERR_ARG;
    return one_build_clusters(m->sub[(size_t)k], nf, nullptr, nullptr, nullptr, 0, fix ? fix + (size_t)f0 * 10 : nullptr, coeffs + f0,
                              clusters_out ? clusters_out + (size_t)f0 * Wc * 10 : nullptr, &part);
  });
  return multi_done(ctx, m, F, rc, "balm_build_clusters_planes");
}

int balm_window_add_scan_strided(balm_ctx *ctx, const void *points, long n_pts, size_t stride_bytes, const double *pose12) {
  return one_window_add_scan_strided(ctx, points, n_pts, stride_bytes, pose12);
}

int balm_window_open(balm_ctx *ctx, const balm_voxel_opts *opts) { return one_window_open(ctx, opts); }
int balm_window_add_scan(balm_ctx *ctx, const float *xyz, long n_pts, const double *pose12) {
  return one_window_add_scan(ctx, xyz, n_pts, pose12);
}
int balm_window_recut(balm_ctx *ctx) { return one_window_recut(ctx); }
int balm_window_marginalize(balm_ctx *ctx, int mg_size, const double *poses) { return one_window_marginalize(ctx, mg_size, poses); }
int balm_window_features(balm_ctx *ctx, int *F_out) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_window_features(ctx, F_out);
  ctx->multi = nullptr;
  int rc = one_window_features(ctx, F_out);
  ctx->multi = m;
  m->F = 0;
  if (rc || *F_out == 0) return rc;
  if ((rc = assoc_clusters_host(ctx))) return rc;
  return multi_set_features(ctx, m, *F_out, ctx->assoc_clusters.data(), ctx->assoc_fix.data(), ctx->assoc_coeffs.data());
}
int balm_window_info(balm_ctx *ctx, int *scans, long *points, long *nodes) {
  if (!ctx) return BALM_ERR_ARG;
  if (!ctx->window) { ctx->err = "balm_window_info: no open window"; return BALM_ERR_STATE; }
  if (scans) *scans = window_count(ctx->window);
  if (points) *points = window_points(ctx->window);
  if (nodes) *nodes = window_nodes(ctx->window);
  return BALM_OK;
}
int balm_window_get_points(balm_ctx *ctx, float *xyz, int *slot, int *feature, long capacity, long *n_out) {
  if (!ctx) return BALM_ERR_ARG;
  if (int rcw = window_alive(ctx, "balm_window_get_points")) return rcw;
  if (!n_out || capacity < 0) { ctx->err = "balm_window_get_points: bad argument"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  const long n = window_get_points(ctx->window, xyz, slot, feature, capacity);
  if (n == -2) { ctx->err = "balm_window_get_points: capacity too small"; return BALM_ERR_ARG; }
  if (n < 0) return window_rc(ctx, "balm_window_get_points", -1);
  *n_out = n;
  return BALM_OK;
}
int balm_window_close(balm_ctx *ctx) {
  if (!ctx) return BALM_ERR_ARG;
  if (ctx->window) { hipSetDevice(ctx->device); window_close(ctx->window); ctx->window = nullptr; }
  return BALM_OK;
}

int balm_evaluate(balm_ctx *ctx, int form, const double *poses, int head, int end, double *Hess, double *JacT, double *residual) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_evaluate(ctx, form, poses, head, end, Hess, JacT, residual);
  if (m->F < 1) { ctx->err = "balm_evaluate: no features installed"; return BALM_ERR_STATE; }
  if ((form != 0 && form != 1) || !poses || head < 0 || end > m->F || head >= end) { ctx->err = "balm_evaluate: bad argument"; return BALM_ERR_ARG; }
  return multi_run(m, [&](int k) {
    const int f0 = m->fbeg[(size_t)k], f1 = m->fbeg[(size_t)k + 1];
    const int lh = (head > f0 ? head : f0) - f0, le = (end < f1 ? end : f1) - f0;
    double r = 0;
    return one_evaluate(m->sub[(size_t)k], form, poses, lh, le > lh ? le : lh, k ? nullptr : Hess, k ? nullptr : JacT, k ? &r : residual);
  });
}

int balm_only_residual(balm_ctx *ctx, const double *poses, double *residual) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_only_residual(ctx, poses, residual);
  if (m->F < 1) { ctx->err = "balm_only_residual: no features installed"; return BALM_ERR_STATE; }
  if (!poses || !residual) { ctx->err = "balm_only_residual: bad argument"; return BALM_ERR_ARG; }
  return multi_run(m, [&](int k) { double r = 0; return one_only_residual(m->sub[(size_t)k], poses, k ? &r : residual); });
}

int balm_damping_iter(balm_ctx *ctx, const balm_lm_opts *o, double *poses, balm_iter_log *log, int *n_iters) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_damping_iter(ctx, o, poses, log, n_iters);
  if (m->F < 1) { ctx->err = "balm_damping_iter: no features installed"; return BALM_ERR_STATE; }
  if (!o || !poses || o->max_iter < 1) { ctx->err = "balm_damping_iter: bad argument"; return BALM_ERR_ARG; }
  const size_t np = (size_t)12 * ctx->W;
  std::vector<std::vector<double>> pk((size_t)m->n, std::vector<double>(poses, poses + np));     // every device thread: its own copy
  std::vector<int> its((size_t)m->n, 0);
  balm_lm_opts ok = *o;
  int rc = multi_run(m, [&](int k) {
    balm_lm_opts oo = ok;
    if (k) oo.verbose = 0;
    return one_damping_iter(m->sub[(size_t)k], &oo, pk[(size_t)k].data(), k ? nullptr : log, &its[(size_t)k]);
  });
  if (rc) { if (ctx->err.empty()) for (auto *q : m->sub) if (!q->err.empty()) { ctx->err = q->err; break; } return rc; }
  std::memcpy(poses, pk[0].data(), np * sizeof(double));
  if (n_iters) *n_iters = its[0];
  return BALM_OK;
}

int balm_pose_covariance(balm_ctx *ctx, const double *poses, const double *cluster_cov, double point_sigma, double *Rcov,
                         double *Rcov_raw) {
  balm_multi *m = leader_of(ctx);
  if (!m) return one_pose_covariance(ctx, poses, cluster_cov, point_sigma, Rcov, Rcov_raw);
  if (m->F < 1) { ctx->err = "balm_pose_covariance: no features installed"; return BALM_ERR_STATE; }
  const size_t W = (size_t)ctx->W;
  return multi_run(m, [&](int k) {
    const size_t f0 = (size_t)m->fbeg[(size_t)k];
    return one_pose_covariance(m->sub[(size_t)k], poses, cluster_cov ? cluster_cov + f0 * W * 81 : nullptr, point_sigma,
                               k ? nullptr : Rcov, k ? nullptr : Rcov_raw);
  });
}

int balm_get_solve_trace(balm_ctx *ctx, long long *ticks, long capacity, int *dims3) {
  if (!ctx || !dims3) return BALM_ERR_ARG;
  const int P = ctx->nA / NB, RB = 2 * P + 1;
  dims3[0] = RB; dims3[1] = P; dims3[2] = 6;          // + P x 6 step-phase sums of the right-hand-side block's workgroup
  if (!ctx->d_trace) { ctx->err = "balm_get_solve_trace: create the context with BALM_SOLVE_TRACE=1 in the environment"; return BALM_ERR_STATE; }
  if (!ticks || capacity < (long)RB * P * 6 + 6L * P) { ctx->err = "balm_get_solve_trace: buffer too small"; return BALM_ERR_ARG; }
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(ticks, ctx->d_trace, ((size_t)RB * P * 6 + 6 * (size_t)P) * sizeof(long long), hipMemcpyDeviceToHost));
  return BALM_OK;
}

int balm_get_timing(balm_ctx *ctx, double *ms, long *count) {
  if (!ctx) return BALM_ERR_ARG;
  for (int k = 0; k < BALM_T_COUNT; k++) {
    if (ms) ms[k] = ctx->timer.ms[k];
    if (count) count[k] = ctx->timer.cnt[k];
  }
  return BALM_OK;
}

int balm_get_shard_timing(balm_ctx *ctx, int shard, double *ms, long *count) {
  if (!ctx) return BALM_ERR_ARG;
  balm_ctx *q = ctx;
  if (ctx->multi) {
    if (shard < 0 || shard >= ctx->multi->n) { ctx->err = "balm_get_shard_timing: no such shard"; return BALM_ERR_ARG; }
    q = ctx->multi->sub[(size_t)shard];
  } else if (shard != 0) { ctx->err = "balm_get_shard_timing: no such shard"; return BALM_ERR_ARG; }
  return balm_get_timing(q, ms, count);
}

int balm_chain_macro_plan(int panels, int helpers, int *table, long capacity) {
  if (!table || panels < 3 || helpers < 1 || capacity < (long)helpers * 64) return BALM_ERR_ARG;
  std::vector<int> tab;
  if (!balm::chain_macro_plan(panels, helpers, tab)) return BALM_ERR_ARG;
  memcpy(table, tab.data(), tab.size() * sizeof(int));
  return BALM_OK;
}

int balm_reset_timing(balm_ctx *ctx) {
  if (!ctx) return BALM_ERR_ARG;
  for (int k = 0; k < BALM_T_COUNT; k++) { ctx->timer.ms[k] = 0; ctx->timer.cnt[k] = 0; }
  if (balm_multi *m = leader_of(ctx))                      // every device's timers (balm_get_shard_timing reads them)
    for (size_t q = 1; q < m->sub.size(); q++)
      for (int k = 0; k < BALM_T_COUNT; k++) { m->sub[q]->timer.ms[k] = 0; m->sub[q]->timer.cnt[k] = 0; }
  return BALM_OK;
}

int balm_work_model(balm_ctx *ctx, double *out4) {
  if (!ctx || !out4) return BALM_ERR_ARG;
  const double W = ctx->W, F = ctx->multi ? ctx->multi->F : ctx->F;
  out4[0] = ctx->work_S;
  out4[1] = ctx->work_B;
  if (ctx->multi && ctx->multi->books_per_shard) {        // the shards were cut before a table existed on the host: every device keeps its share
    out4[0] = out4[1] = 0;
    for (const balm_ctx *q : ctx->multi->sub) { out4[0] += q->work_S; out4[1] += q->work_B; }
  }
  (void)W;
  out4[2] = 216.0 * out4[1];                      // 108 FMA = 216 flop per unordered OBSERVED pose pair incl. the diagonal
                                                  // (dense scenes: 108 F W (W+1))
  SyrkPlan p = plan_syrk(ctx->ntiles, 3L * (long)F);
  out4[3] = (double)ctx->ntiles * 25.0 * 2048.0 * ((double)p.Kpad / 4.0);   // 25 MFMAs per k-step of every job
  if (ctx->sparse && !ctx->multi) out4[3] = ctx->sp_steps * 25.0 * 2048.0;
  if (ctx->multi) {                        // what the shards issue, each with its own plan (dense or block-sparse)
    double issued = 0;
    for (size_t k = 0; k < ctx->multi->sub.size(); k++) {
      const balm_ctx *q = ctx->multi->sub[k];
      if (q->F < 1) continue;
      if (q->sparse) issued += q->sp_steps * 25.0 * 2048.0;
      else { const SyrkPlan pk = plan_syrk(q->ntiles, 3L * q->F); issued += (double)q->ntiles * 25.0 * 2048.0 * ((double)pk.Kpad / 4.0); }
    }
    out4[3] = issued;
  }
  return BALM_OK;
}

}  // extern "C"
