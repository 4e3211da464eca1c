// Internal declarations shared by the HIP translation units of libbalm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/balm_hip.h"
#include "host_stage.h"

namespace balm {

// ---- geometry constants ----------------------------------------------------------------------
constexpr int TILE = 80;            // hessian_syrk macro tile (5x5 f64 MFMA 16x16 tiles per wave)
constexpr int TM = 5;
constexpr int TILE_ELEMS = TILE * TILE;   // 6400 doubles, stored [mfma_tile(25)][reg(4)][lane(64)]
constexpr int FEAT_STRIDE = 24;     // per-feature eigen record (doubles)
constexpr int DACC_LEFT = 27;       // per-pose accumulators: 6 gradient + 21 (symmetric 6x6)
constexpr int DACC_RIGHT = 30;      // 6 gradient + 9 (TL) + 9 (TR) + 6 (BR symmetric)
constexpr int DACC_MAX = 30;
constexpr int NB = 48;              // LDL^T panel width
constexpr int MAX_W_LDS = 480;      // feature_factors keeps (12 + DACC) * W doubles in LDS up to here, pose chunks beyond
constexpr int MAX_W = 1024;         // window limit of a context (n = 6144 unknowns)
constexpr int MAX_SHARDS = 16;      // devices of one balm_create_multi context

// feat record layout
enum { FT_NN = 0, FT_VBAR = 1, FT_LAM = 4, FT_U0 = 7, FT_U1 = 10, FT_U2 = 13, FT_C0 = 16, FT_C1 = 17,
       FT_C2 = 18, FT_COE = 19 };

struct Timer {
  bool on = false;
  struct Span { hipEvent_t a, b; int slot; };
  std::vector<Span> pending;
  std::vector<hipEvent_t> pool;
  double ms[BALM_T_COUNT] = {0};
  long cnt[BALM_T_COUNT] = {0};
};

}  // namespace balm

namespace balm {
struct WindowSession;
struct AssocMail {                                      // a context's persistent small state for balm_associate
  volatile unsigned int *host = nullptr; unsigned int *dev = nullptr; unsigned int seq = 0;   // pinned mailbox for the counts the host reads between kernels
  // k_scan_heads' tile words: SCAN_TILES_CAP 64-bit words that persist between scans, each tagged with the scan's generation -- a word of
  // an older scan reads "not there yet", so no scan has to clear them first (a fill launch in front of each of the nine scans of an
  // association is ~4 us of launch and gap)
  unsigned long long *scan_state = nullptr; unsigned int scan_gen = 0;
};
constexpr long SCAN_TILES_CAP = 8192;
}

struct balm_ctx {
  int W = 0, n = 0, npad = 0, T = 0, ntiles = 0, device = 0, flags = 0;
  int nA = 0;                       // solver dimension (n rounded up to NB)
  int F = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // features (HBM resident)
  double *d_cl = nullptr;           // [F][10][W]  per-feature SoA
  double *d_fix = nullptr;          // [F][10] or null
  double *d_coe = nullptr;          // [F]
  // poses
  double *d_poses = nullptr;        // [W][12] current
  double *d_poses_tmp = nullptr;    // [W][12] trial
  // per-evaluation scratch
  double *d_C = nullptr;            // [F][10] world moments
  double *d_feat = nullptr;         // [F][FEAT_STRIDE]  per-feature eigen records at the CURRENT poses
  double *d_feat_tmp = nullptr;     // ... at the TRIAL poses (swapped in when a step is accepted)
  double *d_rpart_tmp = nullptr;    // residual partials at the trial poses
  int nr_cur = 0, nr_tmp = 0;       // number of valid residual partials in d_rpart / d_rpart_tmp
  bool feat_cur_valid = false;      // d_feat / d_rpart describe d_poses
  double *d_Gt = nullptr;           // [Kcols][npad]  factored Hessian columns (k-major)
  // the fused trial evaluation (k_moments_factors) leaves the TRIAL poses' factors here; swapped in when the step is accepted
  double *d_Gt2 = nullptr, *d_dpart2 = nullptr;
  size_t cap_Gt2 = 0, cap_dpart2 = 0;
  size_t gt_dirty_cols = ~(size_t)0; // columns [gt_dirty_cols, cap) and the row padding of d_Gt are known to be zero (~0: nothing is)
  bool gt_cur_valid = false;        // d_Gt / d_dpart (and d_feat / d_rpart) describe d_poses: the next evaluation starts at the SYRK
  int gt_parity = 0;                // flips with every (d_Gt, d_Gt2) swap: captured LM graphs hold the pointers
  bool gt_trial_valid = false;      // d_Gt2 / d_dpart2 describe d_poses_tmp (this iteration's trial)
  size_t cap_cl = 0, cap_fix = 0, cap_coe = 0, cap_C = 0, cap_feat = 0, cap_feat_tmp = 0, cap_rpart = 0, cap_rpart_tmp = 0;   // elements
  size_t cap_slot = 0, cap_items = 0, cap_chunk_ids = 0, cap_csr = 0;
  bool has_fix = false;             // the installed table carries fix clusters (d_fix may be a larger, older allocation)
  size_t cap_Gt = 0;                // doubles
  double *d_part = nullptr;         // [SG][ntiles][6400] split-K partial tiles
  size_t cap_part = 0;
  double *d_dpart = nullptr;        // [nblk_factors][DACC][W]
  size_t cap_dpart = 0;
  double *d_rpart = nullptr;        // residual partials at the current poses
  double *d_red = nullptr;          // [ntiles*6400 | DACC_MAX*W | r | pad]   all-reduce payload
  size_t red_len = 0;
  // block-sparse SYRK plan of the installed features (empty = dense): column order, (job, chunk) items, per-job slots
  std::vector<int> h_jobs;          // host copy of d_jobs
  bool sparse = false;
  int sp_nsteps = 0, sp_nchunks = 0, sp_nitems = 0;
  int *d_slot = nullptr;            // [F] feature -> column slot of Gt
  int *d_items = nullptr;           // [nitems][4] job, first entry of d_chunk_ids, entry count, output slot
  int *d_chunk_ids = nullptr;       // the jobs' chunk lists, back to back
  double sp_steps = 0;              // k-steps the plan issues over all items
  int *d_csr = nullptr;             // [ntiles + 1]
  int *d_jobs = nullptr;            // [ntiles][4]  SYRK jobs per k-slice: type, block I (or first block of a group), block J
  int *d_sub = nullptr;             // [ntiles][25] (R << 16) | C: global 16-row sub-tile coordinates of each accumulator tile
  double *d_H = nullptr;            // [n][n] column-major
  double *d_g = nullptr;            // [n]
  // solver
  double *d_A = nullptr;            // [nA cols][2 nA + NB rows] permuted damped matrix + RHS row tile + identity -> L, z, L^-T D^+
  double *d_Wp = nullptr;           // [2][NB][2 nA + NB]  W21 = L21 * D11 of the current panel (two alive with lookahead)
  double *d_dvec = nullptr;         // [nA] pivots D
  double *d_z = nullptr;            // [nA] z = D^+ L^-1 P b, then scratch of the backward sweep
  double *d_x = nullptr;            // [16][nA] column-chunk partials of the solution in permuted order
  int *d_perm = nullptr;            // [nA] position -> original index
  int *d_flags = nullptr;           // [2][2P+1][P] done / far flags of k_ldl_fused / k_ldl_chain, + [P] minv flags + abort (zeroed by k_build_A)
  long long *d_trace = nullptr;     // BALM_SOLVE_TRACE=1: [2P+1][P][6] phase timestamps of the last k_ldl_fused (diagnostics)
  int fused_cap = -1;               // co-resident workgroups of k_ldl_fused on this device (-1 = not asked yet, 0 = unavailable)
  int chain_cap = -1;               // ... of k_ldl_chain
  double *d_minv = nullptr;         // [P][48][48] Minv_p = L11^-T D11^-1 of every panel (k_ldl_chain), then [P][48][48] its exchange copies of L[p+2, p]
  int *d_macro_tab = nullptr;       // [NH][64] the macro-tiles every helper of k_ldl_chain owns (chain_macro_table), built at the first such launch
  int macro_tab_P = 0, macro_tab_NH = 0;
  double *d_dx = nullptr;           // [n]
  double *d_scal = nullptr;         // [16] device scalars: 0 r1, 1 r2, 2 q1, 3 flags
  double *h_scal = nullptr;         // pinned mirror (16) + a ring of damping values on their way to d_scal[SCAL_U] (64) + a stamp
  double *d_hscal = nullptr;        // its device alias: k_scalars_mail writes the mirror and the stamp straight into host memory
  unsigned long long mail_seq = 0;  // stamp of the last k_scalars_mail launch
  bool need_minv = false;           // the caller wants M = L^-T D^+ in the identity rows of d_A (balm_pose_covariance): no back-substitution path
  bool solve_tiled = false;         // d_A holds [A ; rhs] tile by tile (k_build_A -> k_ldl_chain without identity rows -> k_ldl_backsolve)
  int chain_refused_P[2] = {-1, -1};   // panels for which launch_factor_chain refused [without / with identity rows] (too few co-resident helpers, ...)
  bool solve_backsub = false;       // the last factorisation ran without identity rows: k_ldl_backsolve instead of k_ldl_apply
  bool counted_live = false;        // this context is in the per-device count of live contexts (context_born / context_gone)
  bool persistent_off = false;      // a wait inside k_ldl_chain / k_ldl_fused / k_ldl_backsolve timed out on this context (its workgroups were not all
                                    // resident: another process or stream held CUs): that solve was retried on the launch path, and so is every later one
  bool inject_solve_timeout = false;   // tests: the next solve finds its abort flag raised (BALM_FAULT_INJECT="timeout,<iteration>")
  bool small_ready = false, small_refused = false;   // k_solve_small (windows of <= 24 poses): its LDS attribute is set on this device / the device refused it
  double u_value = 0.0;             // damping of the next solve (set_damping)
  bool u_on_device = false;         // ... read by the solve's kernels from d_scal[SCAL_U] (graph capture / replay) instead of their arguments
  int u_ring = 0;
  // one LM iteration as a replayable hipGraph, per (Hessian evaluated?, which pose buffer is current): [4]
  static constexpr int LM_GRAPHS = 32;       // (which factor buffer is current, fused trial evaluation, factors current, evaluated, pose-buffer parity)
  hipGraphExec_t lm_graph[LM_GRAPHS] = {};
  int lm_graph_form[LM_GRAPHS] = {};
  bool graphs_ok = true;            // false after a failed capture: never tried again on this context
  int parity = 0;                   // toggles with every accepted step (pointer swap of current / trial buffers)
  // host bookkeeping
  std::vector<int> planes_per_pose;
  double work_S = 0, work_B = 0;
  std::vector<double> assoc_clusters, assoc_coeffs;   // host copies of the last balm_associate (the clusters: fetched from d_cl on first use)
  bool assoc_cl_on_device = false;  // d_cl still holds that association's table and assoc_clusters has not been fetched yet
  std::vector<int> assoc_layer, assoc_point_feat;
  std::vector<double> assoc_fix;
  balm::WindowSession *window = nullptr;   // balm_window_*: the sliding-window map (kernels_window.inc)
  bool window_dead = false;         // a window call failed on the device half-way: the map is in an undefined state until re-opened
  balm::AssocMail amail;            // pinned mailbox of balm_associate's host-read counts (allocated on first use)
  void *d_arena = nullptr;          // balm_associate scratch, grown to what the last call needed
  size_t arena_cap = 0;
  char *d_stage = nullptr;          // per-call staging (uploads, layout changes, covariance work matrices): grown, never
  size_t stage_cap = 0, stage_off = 0;   // shrunk, carved by a bump pointer -- no hipMalloc / hipFree per call
  balm::PinnedRing ring;            // pinned chunks the caller's big host arrays travel through (host_stage.h), allocated on first use
  balm_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  balm::Timer timer;
  // collective transport (balm_multi.hip): stream-ordered RCCL inside the library, either as one of the devices of a
  // balm_create_multi context or as one rank of a multi-process job (balm_comm_init_rank)
  void *comm = nullptr;             // ncclComm_t; written only by the thread that owns the context's calls (never by a peer)
  std::atomic<bool> comm_dead{false};   // a peer device thread aborted this communicator (multi_abort): no further collective
  bool comm_aborted = false;        // ncclCommAbort was called on `comm` (guarded by comm_mu)
  std::timed_mutex comm_mu;         // [check comm_dead + enqueue ncclAllReduce] vs [set comm_dead + ncclCommAbort]
  int rank = 0, nranks = 1;
  unsigned long long *d_rowmax = nullptr, *d_rowmax2 = nullptr; size_t cap_rowmax = 0, cap_rowmax2 = 0;      // ... the rows' maxima of Gt / Gt2
  double *d_rowmax_part = nullptr; size_t cap_rowmax_part = 0;      // ... per workgroup of the factor kernel, before their reduction
  bool rowmax_cur_valid = false, rowmax_trial_valid = false;
  unsigned char *d_i8 = nullptr; size_t cap_i8 = 0;     // scratch of the INT8 SYRK (BALM_SYRK=int8): digits, row scales, partial tiles
  struct balm_multi *multi = nullptr;   // set on every device context of a balm_create_multi context
  double *d_pre = nullptr;          // [W + 2] pre-loop all-reduce: planes per pose, error flag
};

namespace balm {

// launchers (kernels_accum.hip)
hipError_t prepare_device_accum();     // per-device kernel attributes (dynamic LDS limits); once per context
hipError_t prepare_device_cov();
hipError_t preload_solve();            // load a translation unit's code object on the current device (balm_prewarm)
hipError_t preload_build();
hipError_t preload_voxel();
void launch_transpose_clusters(hipStream_t s, const double *aos, double *soa, int F, int W);
void launch_world_moments(hipStream_t s, const double *cl, const double *poses, int W, int f0, int f1, double *C);
struct EigenMail { double *scal; double *host; double stamp; int slot; };      // a ONE-workgroup k_feature_eigen sends the LM iteration's scalars itself (host == NULL: no)
int launch_feature_eigen(hipStream_t s, const double *C, const double *fix, const double *coe, int f0, int f1,
                         double *feat, double *rpart, const EigenMail *mail = nullptr);   // returns #partials
int factors_grid(int W, int nfeat, int form);
int factors_chunk(int W);     // poses per workgroup of the factor kernel (the whole window up to MAX_W_LDS)
bool launch_factors(hipStream_t s, int form, const double *cl, const double *poses, const double *feat, int W,
                    int npad, int f0, int f1, double *Gt, double *dpart, int nblk, const int *slot = nullptr,
                    unsigned long long *rowmax = nullptr, double *rowmax_part = nullptr);      // rowmax: the INT8 product's row maxima ride along (true if they did)
int launch_moments_factors(hipStream_t s, int form, const double *cl, const double *poses, const double *fix, const double *coe, int W,
                           int npad, int F, double *Gt, double *dpart, int nblk, const int *slot, double *feat, double *rpart,
                           unsigned long long *rowmax = nullptr, double *rowmax_part = nullptr);
struct SyrkPlan { int SG; int nsteps; int Kpad; long nblocks; };   // k-slices, MFMA k-steps per wave, padded K, workgroups
SyrkPlan plan_syrk(int ntiles, long K);
void launch_syrk(hipStream_t s, const double *Gt, int npad, int ntiles, const int *tileIJ, const SyrkPlan &p,
                 double *part);
void launch_syrk_sparse(hipStream_t s, const double *Gt, int npad, const int *jobs, const int *items, const int *chunk_ids,
                        int nsteps, long nitems, double *part);
void launch_reduce(hipStream_t s, const double *part, int SG, long tile_elems_total, const double *dpart, int nblk,
                   int dacc_len, const double *rpart, int nr, double *red, long red_dacc_off, long red_r_off,
                   const int *csr_ptr = nullptr);
void launch_assemble(hipStream_t s, int form, const double *red, long red_dacc_off, const int *tileIJ, int ntiles,
                     int W, double *H, double *g, const double *r_in = nullptr, double *r_out = nullptr);
void launch_sum_scalar(hipStream_t s, const double *rpart, int nr, double *out);
// kernels_syrk_i8.hip: Gt Gt^T on the INT8 matrix cores by error-free slicing (BALM_SYRK=int8, opt-in)
struct I8Layout { int T, rows_p, NT, M; long Kp; size_t off_digits, off_rowmax, off_scale, off_part; };
size_t syrk_i8_scratch_bytes(int n, long K, I8Layout *lay);
hipError_t prepare_device_syrk_i8();
int launch_syrk_i8(hipStream_t s, const double *Gt, int npad, int n, long K, const int *tileIJ, int ntiles, unsigned char *scratch, double *part,
                   const unsigned long long *rowmax_known = nullptr);

// balm_multi.hip: one context over several devices of this process, RCCL loaded on first use
struct Barrier;
}  // namespace balm

struct balm_multi {
  int n = 0;                                // device contexts; sub[0] is the public handle
  bool loopback = false;                    // all shards on one physical device, in-library sum instead of RCCL
  std::vector<balm_ctx *> sub;
  std::vector<int> fbeg;                    // shard k holds features [fbeg[k], fbeg[k+1]) of the caller's table
  int F = 0;
  bool books_per_shard = false;             // planes per pose / work model: every device holds its share (else: device 0 holds the whole table's)
  // device threads (device 0 is driven by the calling thread)
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t gen = 0;
  int pending = 0;
  bool quit = false;
  const std::function<int(int)> *job = nullptr;
  std::vector<int> rc;
  balm::Barrier *bar = nullptr;
  // first non-zero status any device thread returned from the current job: the other threads leave their host-side waits
  // (the scalar hand-over of the LM loop, the loopback barrier) with it instead of waiting for a peer that is gone.  An
  // RCCL communicator whose peer never enqueued its collective is aborted (ncclCommAbort) and the context is dead.
  std::atomic<int> abort_rc{0};
  std::atomic<bool> dead{false};
  // LM decision scalars of device 0, per iteration parity
  std::atomic<uint64_t> lm_seq{0};
  uint64_t lm_epoch = 0;
  double lm_vals[2][8] = {{0}};
  // loopback transport
  std::vector<hipEvent_t> ev1, ev2;
  std::vector<double *> lb_tmp, lb_buf;
  std::vector<size_t> lb_cap;
};

namespace balm {
balm_multi *multi_new(const std::vector<balm_ctx *> &subs, bool loopback, std::string *err);
void multi_delete(balm_multi *m);
int multi_run(balm_multi *m, const std::function<int(int)> &f);
int comm_unique_id(void *out128);
int comm_init_rank(balm_ctx *ctx, int nranks, int rank, const void *id128);
void comm_destroy(balm_ctx *ctx);
const char *rccl_load_error();
int comm_allreduce(balm_ctx *ctx, double *buf, long n);      // stream-ordered sum over the ranks of ctx->comm
void comm_query(const balm_ctx *ctx, int *count, int *rank);  // what RCCL reports for ctx->comm
int loopback_allreduce(balm_ctx *ctx, double *buf, long n);   // shards of one physical device (test transport)
bool multi_is_loopback(const balm_ctx *ctx);
int multi_share_scalars(balm_ctx *ctx, int it, double *vals, int count);    // rank 0's LM scalars -> all device threads; != 0: a peer failed
void multi_abort(balm_multi *m, int rc);                      // a device thread failed: wake every host-side wait of the job
int multi_host_barrier_rc(balm_ctx *ctx, int rc);             // all device threads meet; returns the first non-zero rc

// launchers (kernels_solve.hip)
constexpr int SCAL_U = 5;            // d_scal slot of the damping u
constexpr int SCAL_STAMP = 16 + 64;  // h_scal slot of k_scalars_mail's stamp (behind the mirror and the damping ring)
bool chain_macro_plan(int P, int NH, std::vector<int> &tab);      // who owns which 2 x 2 macro-tile in k_ldl_chain (kernels_chain.inc; host only)
void context_born(int device);   // contexts of this process alive per device: the persistent solve kernels size their grids for their share of the slots
void context_gone(int device);
bool solve_timed_out(balm_ctx *c);     // host-synchronous: the last solve's abort flag (a poll limit was hit)
bool solve_is_persistent(const balm_ctx *c);      // the factorisation of this window runs as k_ldl_fused
void launch_solve(balm_ctx *c, bool new_hessian, int upd_form = 0, const double *upd_poses = nullptr, double *upd_out = nullptr);      // (H + u diag H) dx = -g, u = d_scal[SCAL_U]; q1 -> d_scal[2]
void launch_update_poses(hipStream_t s, int form, int W, const double *poses, const double *dx, double *out);
void launch_scalars_mail(hipStream_t s, double *d_scal, double *d_hscal, double stamp, const double *rpart = nullptr, int nr = 0,
                         int slot = 0);      // [d_scal[slot] = sum rpart;] d_scal[0..15] + stamp -> pinned host mirror
void launch_reanchor(hipStream_t s, int W, double *poses);

// kernels_voxel.hip
struct AssocOpts {
  int W;                    // scans, the marginalised ones included
  double voxel_size;
  float thr[3];
  int min_ps, layer_limit, min_observers, fix_frames;
  double max_dis, ratio21_max, lam0_max;
  int fix_limit = 50;       // to_margi: fix_point.N < fix_limit (bavoxel.hpp:793: 50; BAs_left.hpp:756: 30)
  int defer_recut = 0;      // window map: add_scan is cut_voxel only
};
// *outputs_owned: the output arrays were hipMalloc'ed for the caller (true) or live in the arena until the next call (false)
int associate_device(hipStream_t s, const float *d_xyz, const int *d_frame, const long *d_first, const double *d_poses, long n, const AssocOpts &o,
                     void *arena, size_t arena_cap, size_t *arena_need, int *F_out, double **d_out, double **d_coe,
                     double **d_fix, int **d_layer, int **d_point_feat, long *n_roots, AssocMail *mail, bool *outputs_owned);

// sliding-window map: the incremental use of the reference's octree (kernels_window.inc, part of kernels_voxel.hip).
// Return codes as associate_device.
WindowSession *window_open(hipStream_t s, const AssocOpts &o);          // o.W = window size
void window_close(WindowSession *w);
int window_add_scan(WindowSession *w, const float *xyz_new, long n_new, const double *pose12, bool xyz_on_host);
int window_recut(WindowSession *w);
int window_marginalize(WindowSession *w, int mg, const double *poses);
int window_features(WindowSession *w, int min_observers, int *F_out, double **d_out, double **d_coe, double **d_fix, int **d_layer);
long window_get_points(WindowSession *w, float *xyz, int *slot, int *feature, long cap);
int window_count(const WindowSession *w);
int window_min_observers(const WindowSession *w);
long window_points(const WindowSession *w);
long window_nodes(const WindowSession *w);

// launchers (kernels_cov.hip)
int cov_factors_grid(int W, int F);
void launch_cov_factors(hipStream_t s, const double *cl, const double *ccov, double sigma2, const double *poses,
                        const double *feat, int W, int npad, int F, double *Gx, double *Gy, double *dpart, int nblk);
void launch_cov_reduce_tiles(hipStream_t s, const double *part, int SG, long tile_total, double *red);
void launch_cov_reduce_dacc(hipStream_t s, const double *dpart, int nblk, int W, double *out);
void launch_cov_assemble(hipStream_t s, const double *redx, const double *redy, const double *sdiag, const int *tileIJ,
                         int ntiles, int W, double *Rout);
void launch_congruence_inverse(balm_ctx *c, const double *Rraw, double *T0, double *T1, double *Rcov);

// launchers (kernels_build.hip)
void launch_build_clusters(hipStream_t s, const float *xyz, const int *feat_id, const int *pose_id, long n_pts,
                           int F, int W, double *soa, int *unsorted_flag);      // grouped points: no atomics, bit-exact pushes
void launch_build_clusters_any(hipStream_t s, const float *xyz, const int *feat_id, const int *pose_id, long n_pts,
                               int F, int W, double *soa);                       // any order: atomics
void launch_soa_to_aos(hipStream_t s, const double *soa, double *aos, int F, int W);
void launch_rebase_ids(hipStream_t s, int *d_id, long n, int base);                              // id[p] -= base (a shard's feature ids)
void launch_expand_ids(hipStream_t s, const long *d_first, int m, long n, int *d_id);            // container index of every point from the containers' prefix counts
void launch_unpack_xyzw(hipStream_t s, const float *d_rec, long n, float *d_xyz, int *d_aux);    // (x, y, z, w) records -> xyz + (int)w
void launch_obs_mask(hipStream_t s, const double *soa, int F, int W, unsigned char *mask);      // N != 0 per (feature, pose)

}  // namespace balm
