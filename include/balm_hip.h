/* balm_hip.h -- C ABI of libbalm_hip.so: the MI355X (gfx950, HIP) implementation of BALM 2.0's
 * second-order bundle-adjustment hot path.  Plain C types, caller-owned host buffers, int status
 * returns, no exceptions, no exit().  One context per caller thread; every call is synchronous
 * (device work for the call has finished when it returns).
 *
 * The reference has no FFI for this path; its "API" is the method surface of two header classes
 * (VOX_HESS, BALM2) plus the mutable global `int win_size` (src/benchmark/bavoxel.hpp:17).  Each
 * entry point below names the reference interface it replaces (paths relative to the reference
 * root).  include/balm_shim.hpp re-declares VOX_HESS / BALM2 with the reference's signatures on
 * top of this ABI; INTEGRATION.md shows the binding.
 *
 * Layouts (all FP64):
 *   pose     12 doubles: R column-major (R(r,c) = q[3*c+r]) then p          include/tools.hpp:144-145
 *   cluster  10 doubles: Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N                 include/tools.hpp:290-295
 *   clusters F*W*10, feature-major: clusters[(a*W + i)*10 + c]; an all-zero cluster (N == 0)
 *            means "pose i does not observe feature a"
 *   Hess     (6W)x(6W) column-major, fully symmetric-filled; pose block i = rows 6i..6i+5 =
 *            [dtheta(3); dt(3)]  (DVEL = 6, include/tools.hpp:20)
 */
#ifndef BALM_HIP_H
#define BALM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct balm_ctx balm_ctx;

enum { BALM_FORM_LEFT = 0, BALM_FORM_RIGHT = 1 };

enum {
  BALM_OK = 0,
  BALM_ERR_ARG = 1,            /* bad argument                                              */
  BALM_ERR_HIP = 2,            /* HIP runtime error; see balm_last_error()                  */
  BALM_ERR_STATE = 3,          /* call order (e.g. evaluate before set_features)            */
  BALM_ERR_TOO_FEW_PLANES = 4, /* replaces printf+exit(0) at bavoxel.hpp:1079-1085          */
  BALM_ERR_NUMERIC = 5         /* non-finite residual / zero point count                    */
};

/* balm_create flags */
enum {
  BALM_FLAG_TIMING = 1,          /* record HIP events around every kernel class (balm_get_timing) */
  BALM_FLAG_LOOPBACK_SHARDS = 2, /* balm_create_multi: all n_devices shards on the ONE device `first_device`, summed by
                                    an in-library kernel instead of RCCL -- exercises the sharded path on a one-GPU box */
  BALM_FLAG_SYRK_INT8 = 4        /* OPT-IN: the Hessian's (and the covariance's) dense Gt Gt^T products of 12 288 columns or more (windows of 96 poses or more) on
                                    the INT8 matrix cores by error-free digit slicing instead of FP64 MFMA: twice as
                                    fast; g and the residual bit-identical, H within ~1e-12 of the largest entry of the FP64 path's at
                                    the benchmark size (an entry's error is ~2^-32 of the product of its rows' largest |entries| per
                                    column, unsigned).  Same as BALM_SYRK=int8 in the environment.  Default: off, FP64 */
};

/* One row per LM iteration; the quantities the reference prints at bavoxel.hpp:1132. */
typedef struct balm_iter_log {
  double r1, r2, u, v, q, q1;   /* q = r1 - r2 (before the gain-ratio rescale)              */
  int accepted;                 /* q > 0                                                     */
  int hess_evaluated;           /* Hessian/gradient were recomputed at this iteration        */
} balm_iter_log;

/* LM loop constants.  bavoxel.hpp:1087,1104,1155 use u0=0.01, max_iter=10, rel_tol=1e-6,
 * min_planes_per_pose=20; benchmark_virtual.cpp:380,408,453 use u0=0.1, max_iter=20, no plane
 * precheck. */
typedef struct balm_lm_opts {
  int form;                 /* BALM_FORM_LEFT (active in the reference) or BALM_FORM_RIGHT   */
  double u0;
  int max_iter;
  double rel_tol;
  int min_planes_per_pose;  /* 0 disables the precheck                                       */
  int force_hess;           /* benchmarking: re-evaluate the Hessian on every iteration      */
  int no_stop;              /* benchmarking: ignore the rel_tol stop, run max_iter           */
  int verbose;              /* print the reference's per-iteration line                      */
  int reanchor;             /* express poses relative to pose 0 at the end (:1159-1164)      */
  double abs_tol;           /* > 0: also stop when |r1-r2| < abs_tol (the consistency driver's
                               criterion, src/simulation/BAs_left.hpp:1083, 1e-9 with max_iter 1000) */
} balm_lm_opts;

/* Replaces the global `int win_size` (bavoxel.hpp:17) + object construction.  `device` is the
 * HIP device ordinal this context owns (one process per GPU sets it to LOCAL_RANK).  1 <= win_size <= 1024
 * (balm_pose_covariance: <= 480); NULL on failure. */
balm_ctx *balm_create(int win_size, int device, int flags);
void balm_destroy(balm_ctx *ctx);

/* Starts, on a background thread of the library, what the first use of `device` costs a process and needs no window size: the
 * runtime's own start, the pinned staging ring of the uploads and the code objects of the association / cluster-build / solve
 * kernels (60-100 ms together, profiles/r06_cold_call.txt).  Returns at once; idempotent; balm_create calls it too.  The reference's
 * drivers are one-shot programs (benchmark_realworld.cpp:179): a caller that declares its optimizer object FIRST -- the shim's
 * constructor calls this -- has the device ready by the time its scans are read.  BALM_ERR_ARG: no such device. */
int balm_prewarm(int device);

/* SURVEY 8(b)'s `n_devices`: one context that shards its features over the n_devices GPUs
 * first_device .. first_device+n_devices-1 of this process (one host thread, one stream and one
 * replica of the small per-window state per device) and sums the per-device Hessian/gradient/residual
 * payload with a stream-ordered RCCL all-reduce over xGMI inside every evaluation; replaces the serial
 * `Hess += hessians[i]` over threads at bavoxel.hpp:1049-1056.  Every other entry point takes the
 * returned context exactly like a balm_create one.  The reference's drivers are single-process C++
 * (benchmark_realworld.cpp:144-238, benchmark_virtual.cpp:505-524): this is how they reach 8 GPUs
 * (`BALM2_HIP::n_devices`).  n_devices == 1 is balm_create plus the collective path.  NULL on failure. */
balm_ctx *balm_create_multi(int win_size, int first_device, int n_devices, int flags);

/* The same stream-ordered RCCL transport for the one-process-per-GPU launch (torchrun): rank 0 draws an id
 * (128 bytes, ncclUniqueId), the launcher hands it to every rank, every rank calls balm_comm_init_rank on its own
 * balm_create context and installs ITS shard with balm_set_features.  From then on balm_evaluate /
 * balm_only_residual / balm_damping_iter / balm_pose_covariance sum over the ranks inside the library. */
int balm_comm_unique_id(void *id128);
int balm_comm_init_rank(balm_ctx *ctx, int n_ranks, int rank, const void *id128);
/* What the transport itself reports, for a run that has to prove it really was N ranks: out[0] = ranks in the
 * communicator as RCCL counts them (ncclCommCount; the shard count of a loopback context; 1 without a transport),
 * out[1] = this context's rank (ncclCommUserRank), out[2] = doubles in the all-reduce payload of one Hessian
 * evaluation ([SYRK tiles | per-pose gradient + block diagonal | residual]), out[3] = transport kind: 0 none,
 * 1 RCCL in the library, 2 loopback shards, 3 caller's hook. */
int balm_comm_info(balm_ctx *ctx, long *out4);

/* Replaces F calls of VOX_HESS::push_voxel (bavoxel.hpp:30-51): the shim flattens the borrowed
 * `const vector<PointCluster>*` / `const PointCluster* fix` pointers into these arrays.  Copies to
 * HBM.  `fix` (F*10) may be NULL (= all-empty fix clusters); `coeffs` are the per-feature weights
 * (bavoxel.hpp:42-44; benchmark_virtual.cpp:391), >= 0.  Features seen by fewer than 2 poses are the
 * caller's to drop (push_voxel :32-37); one observer (or none, with a fix cluster) is legal here and
 * contributes like any other; a feature with no observer and no fix cluster has a zero point count and
 * is rejected with BALM_ERR_NUMERIC (the reference would divide by it, bavoxel.hpp:341). */
int balm_set_features(balm_ctx *ctx, int F, const double *clusters, const double *fix,
                      const double *coeffs);

/* The same for a caller whose clusters are NOT one flat array -- VOX_HESS keeps F borrowed
 * `const vector<PointCluster>*` (bavoxel.hpp:24-26, filled by push_voxel :30-51): `fill(user, f0, f1, dst)`
 * writes the clusters of features [f0, f1), (f1 - f0) * W * 10 doubles in the layout above, to dst.  dst is
 * a pinned staging chunk of the library that leaves for the device as soon as it is full, so the table is
 * never flattened into a second host copy.  fill is called from several host threads of the library at
 * once, on disjoint feature ranges, every feature exactly once, before the call returns; it must only read
 * the caller's data and must NOT call back into this library (the library's host pool runs one job at a time: an
 * upload or any other entry point called from inside fill deadlocks).  balm_set_features is this call with a memcpy as fill. */
typedef void (*balm_fill_clusters_fn)(void *user, int f0, int f1, double *dst);
int balm_set_features_cb(balm_ctx *ctx, int F, balm_fill_clusters_fn fill, void *user, const double *fix,
                         const double *coeffs);

/* Replaces VOX_HESS::left_evaluate_acc2 (bavoxel.hpp:304-426; form 0) and
 * VOX_HESS::acc_evaluate2 (bavoxel.hpp:53-158; form 1) over features [head,end), and -- with
 * head=0,end=F -- BALM2::divide_thread_left/right (bavoxel.hpp:1025-1059,989-1023).  Outputs are
 * overwritten; Hess (may be NULL) is returned fully symmetric.  With an all-reduce hook installed
 * the outputs are the sums over all ranks. */
int balm_evaluate(balm_ctx *ctx, int form, const double *poses, int head, int end, double *Hess,
                  double *JacT, double *residual);

/* Replaces VOX_HESS::evaluate_only_residual / BALM2::only_residual (bavoxel.hpp:428-470,
 * 1061-1067). */
int balm_only_residual(balm_ctx *ctx, const double *poses, double *residual);

/* Replaces `D.diagonal() = Hess.diagonal(); dxi = (Hess + u*D).ldlt().solve(-JacT);`
 * (bavoxel.hpp:1113-1114) and `q1 = 0.5*dxi.dot(u*D*dxi-JacT)` (:1127).  Same elimination order
 * as Eigen's LDLT (diagonal pivots by decreasing |diag|), D may be indefinite.  n = 6*win_size.
 * q1 may be NULL. */
int balm_solve_damped(balm_ctx *ctx, const double *Hess, const double *JacT, double u, double *dxi,
                      double *q1);

/* Replaces BALM2::damping_iter (bavoxel.hpp:1069-1166) and the LM loop of BALM2::dampingIter
 * (benchmark_virtual.cpp:380-479): device-resident loop, poses updated in place.  `log` (may be
 * NULL) must hold opts->max_iter rows; *n_iters receives the iterations executed. */
int balm_damping_iter(balm_ctx *ctx, const balm_lm_opts *opts, double *poses_inout,
                      balm_iter_log *log, int *n_iters);

/* Replaces the per-point PointCluster::push loops (tools.hpp:311-316; benchmark_virtual.cpp:
 * 392-403; bavoxel.hpp:1194-1198): builds the F*W clusters on the device from body-frame points
 * keyed by (feature, pose) and installs them like balm_set_features.  xyz: n_pts*3 floats;
 * feat_id / pose_id: n_pts ints.  clusters_out (F*W*10, may be NULL) receives a host copy. */
int balm_build_clusters(balm_ctx *ctx, int F, const float *xyz, const int *feat_id,
                        const int *pose_id, long n_pts, const double *fix, const double *coeffs,
                        double *clusters_out);

/* The same for a caller that holds its points the way the reference does -- one pcl::PointCloud<PointType> per plane with the
 * observing pose in `intensity` (benchmark_virtual.cpp:375,392-403,586; PointType = pcl::PointXYZINormal, 48-byte elements,
 * include/tools.hpp:22): plane_points[a] -> plane_count[a] elements `stride_bytes` apart, float x, y, z at byte offset 0 of each
 * element and the pose index as a FLOAT at byte `pose_offset_bytes` (converted like the reference's `(int)ap.intensity`, :396).
 * The library's host threads read the containers where they lie and pack 16 bytes per point straight into the pinned upload
 * chunks; the plane index of every point is expanded on the device from the counts.  Nothing is flattened on the caller's thread.
 * stride_bytes >= 12, a multiple of 4; pose_offset_bytes + 4 <= stride_bytes; empty planes are legal (NULL pointer allowed). */
int balm_build_clusters_planes(balm_ctx *ctx, int F, const void *const *plane_points, const long *plane_count,
                               size_t stride_bytes, size_t pose_offset_bytes, const double *fix, const double *coeffs,
                               double *clusters_out);

/* Replaces the caller's association stage: cut_voxel over every scan, OCTO_TREE_NODE::recut and
 * ::tras_opt over every root voxel (bavoxel.hpp:1170-1223, 654-776, 908-929;
 * benchmark_realworld.cpp:183-200).  Points are body-frame floats with the index of the scan they
 * belong to, in scan order; `poses` are the initial poses of all scans.  Every voxel-membership and
 * plane decision uses the reference's float/double types and operation order, so the feature set
 * (and every N) equals the reference's.  The features (body-frame clusters, weight = sum_i N_i) are
 * installed like balm_set_features; their order is (layer, voxel key), not the reference's hash-map
 * order -- the optimizer does not depend on it.  *F_out = 0 (and BALM_OK) when no plane was found.
 *
 * The globals of the reference's association are fields here.  balm_voxel_defaults() fills the
 * benchmark drivers' values (bavoxel.hpp:8-15, launch/benchmark_realworld.launch).  The consistency
 * driver's copy of the state machine (src/simulation/BAs_left.hpp:18-23, 647-815; consistency.cpp:
 * 96-150) is the same code with layer_limit 0, thresholds 1/64, min_ps 10, min_observers 0, the
 * stricter plane test (max_plane_dist 1e-3, max_lambda21 25, max_lambda0 1e-10) and fix_frames 1:
 * the window's first scan(s) are marginalised into world-frame fix clusters (to_margi,
 * bavoxel.hpp:778-816, batch form).  With fix_frames = m the context's win_size counts the scans
 * that stay: frame_id runs over 0..win_size+m-1 and `poses` holds win_size+m poses. */
typedef struct balm_voxel_opts {
  double voxel_size;        /* benchmark_realworld.cpp:150 -> launch file                        */
  float eigen_thr[3];       /* per layer; bavoxel.hpp:11 / launch file: 1/16, 1/16, 1/9          */
  int min_ps;               /* bavoxel.hpp:12: 15                                                */
  int layer_limit;          /* bavoxel.hpp:8: 2 (0..2 supported)                                 */
  int min_observers;        /* VOX_HESS::push_voxel, bavoxel.hpp:32-37: 2                        */
  int fix_frames;           /* 0 for the benchmark drivers                                       */
  double max_plane_dist;    /* 0 = off; BAs_left.hpp:674                                         */
  double max_lambda21;      /* 0 = off                                                           */
  double max_lambda0;       /* 0 = off                                                           */
  int want_point_features;  /* keep the point -> feature map for balm_get_association            */
  int fix_point_limit;      /* to_margi folds marginalised scans into a fix cluster while it holds fewer points than
                               this: 50 (bavoxel.hpp:793), 30 in the consistency driver (BAs_left.hpp:756); 0 = 50  */
  int defer_recut;          /* balm_window_*: balm_window_add_scan is cut_voxel only; the caller runs
                               balm_window_recut when it wants OCTO_TREE_ROOT::recut (consistency.cpp:127-136 adds the
                               whole window first, then recuts and marginalises ONCE)                              */
} balm_voxel_opts;
void balm_voxel_defaults(balm_voxel_opts *opts);
int balm_associate(balm_ctx *ctx, const balm_voxel_opts *opts, const float *xyz, const int *frame_id,
                   long n_pts, const double *poses, int *F_out, long *n_root_voxels);

/* The same for the reference's own scan containers -- `vector<pcl::PointCloud<PointType>::Ptr> pl_fulls`
 * (benchmark_realworld.cpp:155,183-184): scan_points[i] -> scan_count[i] elements `stride_bytes` apart (48 for PointXYZINormal),
 * float x, y, z at byte offset 0 of each.  n_scans = win_size + opts->fix_frames, scans in window order.  The library's host
 * threads pack 12 bytes per point from the caller's clouds straight into the pinned upload chunks (no flattened copy, nothing on
 * the caller's thread) and the scan index of every point is expanded on the device from the n_scans counts: the per-point frame_id
 * array of balm_associate -- a quarter of its upload -- does not exist on this route.  stride_bytes >= 12, a multiple of 4; empty
 * scans are legal.  Same results as balm_associate on the flattened arrays, bit for bit. */
int balm_associate_scans(balm_ctx *ctx, const balm_voxel_opts *opts, int n_scans, const void *const *scan_points,
                         const long *scan_count, size_t stride_bytes, const double *poses, int *F_out,
                         long *n_root_voxels);

/* Sliding-window map: the INCREMENTAL use of the reference's adaptive voxel map, kept on the device between calls
 * (OCTO_TREE_ROOT / OCTO_TREE_NODE, src/benchmark/bavoxel.hpp:625-963; balm_associate above is the batch form).
 *   balm_window_open         an empty unordered_map<VOXEL_LOC, OCTO_TREE_ROOT*>; window size = the context's `win`, plus
 *                            opts->fix_frames extra scans the map can hold until they are marginalised (the
 *                            `win_size + fix_size` slots of src/simulation/BAs_left.hpp:640-641).  opts as for balm_associate:
 *                            the consistency driver's plane test (max_plane_dist / max_lambda21 / max_lambda0,
 *                            BAs_left.hpp:647-674) and its fix_point_limit (:756) apply to recut / marginalize here too
 *   balm_window_add_scan     cut_voxel(map, scan, pose, fnum = scans in the window)  (:1170-1223)  followed by
 *                            recut(win_count) of every root (:737-776): nodes that were cut stay cut and forward the
 *                            new scan, the others are judged again over fix cluster + all scans.  xyz: n_pts*3 body-frame
 *                            floats in scan order, pose12: column-major R then p.
 *   balm_window_features     tras_opt of every root (:908-929) + VOX_HESS::push_voxel (:30-51): installs the feature table
 *                            (with fix clusters) like balm_associate; read it with balm_get_features / balm_get_association
 *   balm_window_marginalize  OCTO_TREE_ROOT::marginalize(mg_size, x_poses, win_count) of every root (:948-963, to_margi
 *                            :778-816): with poses (scans_in_window*12) every cluster and point is re-transformed first;
 *                            the first mg_size scans of plane voxels join the fix clusters (fix_point.N < 50), the window
 *                            moves down by mg_size.  poses == NULL is the reference's empty x_poses (no re-transform).
 *   balm_window_recut        recut(win_count) of every root over the scans no recut has seen yet -- with opts->defer_recut the
 *                            calling sequence of consistency.cpp:108-136: cut_voxel for the whole window, ONE recut, ONE
 *                            marginalize(fix_size, {}, win_count).
 * One recut per scan is bavoxel.hpp's own calling convention (a cut node only forwards the newest scan).
 * balm_window_features / _marginalize want every scan recut, and at most `win` scans in the window for _features.  */
int balm_window_open(balm_ctx *ctx, const balm_voxel_opts *opts);
int balm_window_recut(balm_ctx *ctx);
int balm_window_add_scan(balm_ctx *ctx, const float *xyz, long n_pts, const double *pose12);
/* balm_window_add_scan for a scan in the caller's own container (`pcl::PointCloud<PointType> &pl`, bavoxel.hpp:1170): n_pts
 * elements `stride_bytes` apart, float x, y, z at byte offset 0 of each. */
int balm_window_add_scan_strided(balm_ctx *ctx, const void *points, long n_pts, size_t stride_bytes, const double *pose12);
int balm_window_features(balm_ctx *ctx, int *F_out);
int balm_window_marginalize(balm_ctx *ctx, int mg_size, const double *poses);
int balm_window_info(balm_ctx *ctx, int *scans_in_window, long *points, long *nodes);
/* The points the map holds, in scan order: body-frame xyz (n*3), window slot, and the feature of the last balm_window_features
 * each belongs to (-1: none) -- what OCTO_TREE_NODE::corrupt (src/simulation/BAs_left.hpp:886-906) walks, and what
 * balm_build_clusters needs to rebuild the clusters from perturbed points.  Arrays may be NULL; capacity in points
 * (balm_window_info gives the count); *n_out = points written. */
int balm_window_get_points(balm_ctx *ctx, float *xyz, int *slot, int *feature, long capacity, long *n_out);
int balm_window_close(balm_ctx *ctx);

/* Host copies of the feature table installed by the last balm_associate: clusters F*W*10, coeffs F,
 * layer F (octree depth of the feature's voxel).  Any pointer may be NULL. */
int balm_get_features(balm_ctx *ctx, double *clusters, double *coeffs, int *layer);

/* ... and its fix clusters (F*10, zero without fix_frames) and, if want_point_features was set, the
 * feature index of every input point (n_pts ints, -1 = the point belongs to no feature): what
 * OCTO_TREE_NODE::corrupt / a re-build of the clusters (balm_build_clusters) needs. */
int balm_get_association(balm_ctx *ctx, double *fix, int *point_feature);

/* Replaces the covariance tail of the consistency experiment's BALM2::damping_iter
 * (src/simulation/BAs_left.hpp:1089-1096): Hess at `poses` (left form), VOX_HESS::left_jacobian_point summed
 * over all features (:342-473; BALM2::multi_second :995-1023) and `Rcov = Hess^-1 Rcov Hess^-T`.
 * `cluster_cov` (F*W*81, row-major 9x9 per cluster, coordinates Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz) is the
 * PointCluster::c_cov of src/simulation/toolss.hpp:289,345; NULL means the isotropic point noise that
 * PointCluster::push accumulates (toolss.hpp:321-345, p_cov = point_sigma^2 I), rebuilt on the device from the
 * clusters themselves.  The reference's feature weight there is 1 (BAs_left.hpp:44); installed weights != 1
 * scale a feature's contribution like its gradient.  Rcov and Rcov_raw (either may be NULL) are (6W)x(6W),
 * symmetric.  Hess must be non-singular (a fix cluster or another gauge anchor), as in the reference.
 * NEES (consistency.cpp:159-170) = 2*q1 of balm_solve_damped(ctx, Rcov, -err, 0, x, &q1). */
int balm_pose_covariance(balm_ctx *ctx, const double *poses, const double *cluster_cov, double point_sigma,
                         double *Rcov, double *Rcov_raw);

/* A caller-supplied transport instead (any communication library): each rank installs its shard with
 * balm_set_features and a hook that sums a device buffer of n doubles across ranks in place.  The hook is
 * called with the library's stream already synchronised and must return after the reduced data is visible
 * on the device.  Return 0 on success.  (balm_create_multi / balm_comm_init_rank need no hook.) */
typedef int (*balm_allreduce_fn)(void *dev_buf, long n_doubles, void *user);
int balm_set_allreduce(balm_ctx *ctx, balm_allreduce_fn fn, void *user);

/* Timing (BALM_FLAG_TIMING): accumulated HIP-event milliseconds and launch counts per kernel
 * class since the last reset.  Slots: see BALM_T_* below.  ms/count arrays of BALM_T_COUNT. */
enum {
  BALM_T_MOMENTS = 0,   /* world_moments + feature_eigen + residual reduce (= only_residual)  */
  BALM_T_FACTORS = 1,   /* feature_factors (G-tilde, gradient, block-diagonal partials)       */
  BALM_T_SYRK = 2,      /* hessian_syrk (f64 MFMA) -- the dominant kernel                     */
  BALM_T_ASSEMBLE = 3,  /* split-K reduce + assemble H, g                                      */
  BALM_T_SOLVE = 4,     /* permute + blocked LDL^T + triangular solves                        */
  BALM_T_UPDATE = 5,    /* pose update + gain-ratio scalars                                   */
  BALM_T_BUILD = 6,     /* cluster build from points (balm_build_clusters kernel only)        */
  BALM_T_VOXEL = 7,     /* adaptive-voxel association (balm_associate, device part only)      */
  BALM_T_COV = 8,       /* balm_pose_covariance: covariance factors, its two SYRKs, H^-1 R H^-T */
  BALM_T_COMM = 9,      /* the all-reduces of the sharded path (stream time: includes waiting for the slowest rank) */
  BALM_T_UPLOAD = 10,   /* the caller's big host arrays on their way to HBM (clusters, points, ids): first DMA start to last DMA
                           end on the library's stream, i.e. including the host threads' fills of the pinned chunks            */
  BALM_T_COUNT = 11
};
int balm_get_timing(balm_ctx *ctx, double *ms, long *count);
/* The same for ONE device of a balm_create_multi context (shard = 0 .. n_devices - 1; balm_get_timing answers for device 0): the
 * per-device launch times a sharded run's roofline fractions are computed from.  A plain context has shard 0 only. */
int balm_get_shard_timing(balm_ctx *ctx, int shard, double *ms, long *count);

/* Diagnostics of the persistent factorisation kernel: with BALM_SOLVE_TRACE=1 in the environment at balm_create, the
 * panel workgroups of the last solve leave wall-clock ticks (100 MHz) per (row block, column block, phase):
 * 0 arrive, 1 inputs ready, 2 tiles loaded, 3 near update done, 4 factorised, 5 published.  dims3 = {2P+1, P, 6}. */
int balm_get_solve_trace(balm_ctx *ctx, long long *ticks, long capacity, int *dims3);
int balm_reset_timing(balm_ctx *ctx);

/* Host-only diagnostic (no device needed): the ownership table of k_ldl_chain's macro-tile helpers for a window of `panels`
 * 48-column panels and `helpers` helper workgroups -- table[h * 64 + i] = r0 | j0 << 16 of helper h's i-th 2 x 2 macro-tile
 * (rows r0, r0+1 x columns j0, j0+1 of the tall matrix [A ; rhs]; r0 = panels is the right-hand side row), -1 = none;
 * a helper's entries ascend by column.  capacity >= helpers * 64 ints.  BALM_ERR_ARG: fewer than 3 panels, no helper, a
 * helper with more than 64 macro-tiles, or a short buffer.  (Here so that the CPU tests cover the scheduling logic.) */
int balm_chain_macro_plan(int panels, int helpers, int *table, long capacity);

/* Work model of the last balm_set_features: out[0] = S = sum_a n_a, out[1] = sum_a n_a(n_a+1)/2,
 * out[2] = algorithmic FLOPs of one hessian_syrk launch = 216 * out[1] (three 6x6 rank-1 updates per observed
 * unordered pose pair incl. the diagonal; 108*F*W*(W+1) when every pose sees every feature), out[3] = FLOPs the
 * launch actually issues (dense plan: tile padding incl.; block-sparse plan: its (job, chunk) items). */
int balm_work_model(balm_ctx *ctx, double *out4);

const char *balm_last_error(balm_ctx *ctx);
const char *balm_version(void);

/* ABI revision of this header.  It changes whenever a struct above grows, an enum gains a member that sizes a caller's array
 * (BALM_T_COUNT) or an entry point changes its meaning: 3 = round 3 (balm_voxel_opts gained fix_point_limit / defer_recut,
 * BALM_T_COUNT went from 9 to 10), 4 = round 4 (balm_abi_version itself), 5 = round 5 (balm_set_features_cb, BALM_T_UPLOAD: BALM_T_COUNT 11), 6 = this header (the strided point-container entries balm_associate_scans / balm_build_clusters_planes / balm_window_add_scan_strided, balm_prewarm, balm_get_shard_timing).  A caller built against another revision must not call anything else:
 *     if (balm_abi_version() != BALM_ABI_VERSION) { refuse }                                                              */
#define BALM_ABI_VERSION 6
int balm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BALM_HIP_H */
