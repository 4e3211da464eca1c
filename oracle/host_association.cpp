// ORACLE -- TEST INFRASTRUCTURE ONLY: the comparator of the device association (balm_associate, csrc/kernels_voxel.hip).
// A host restatement of the reference's adaptive-voxel association state machine, decision for decision (same
// float/double types where they decide voxel membership), as SURVEY.md Appendix D sanctions for this off-GPU code:
//   src/benchmark/bavoxel.hpp:1170-1223           cut_voxel      (world point -> root voxel key)
//   src/benchmark/bavoxel.hpp:654-699             judge_eigen    (lambda0/lambda1 < threshold[layer])
//   src/benchmark/bavoxel.hpp:701-776             cut_func / recut (<= 2 subdivisions into octants)
//   src/benchmark/bavoxel.hpp:908-929, 30-51      tras_opt / VOX_HESS::push_voxel (filters, weight = sum N)
// The consistency driver's copy of the same state machine (src/simulation/BAs_left.hpp:647-815, consistency.cpp:
// 96-150) differs in three rules, selectable through balm_assoc_set_rules: a stricter plane test
// (max point-to-plane distance, lambda2/lambda1 and lambda0 bounds, :674), the marginalisation of the window's
// first scan(s) into world-frame fix clusters (to_margi, bavoxel.hpp:778-816 == BAs_left.hpp:754-792; batch form:
// the tree is built once, so `fix_point` starts empty), and no minimum number of observers (:38).
// It is itself pinned, bit for bit, to both compiled copies of the reference's state machine (tests/test_association.py).
// Built into oracle/libassoc_host.so (oracle/Makefile); loaded by oracle/assoc_host.py.  The product package never uses it:
// python -m balm_amd.realworld / .consistency associate on the device.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct V3 { double x, y, z; };

struct Cluster {                      // include/tools.hpp:290-349
  double P[6] = {0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz
  double v[3] = {0, 0, 0};
  int N = 0;
  void push(const V3 &p) {
    N++;
    P[0] += p.x * p.x; P[1] += p.x * p.y; P[2] += p.x * p.z; P[3] += p.y * p.y; P[4] += p.y * p.z; P[5] += p.z * p.z;
    v[0] += p.x; v[1] += p.y; v[2] += p.z;
  }
  void add(const Cluster &o) {
    for (int k = 0; k < 6; k++) P[k] += o.P[k];
    for (int k = 0; k < 3; k++) v[k] += o.v[k];
    N += o.N;
  }
};

// eigenvalues of the 3x3 covariance, ascending (cyclic Jacobi; stands in for SelfAdjointEigenSolver)
void eigvals3(double a00, double a01, double a02, double a11, double a12, double a22, double lam[3]) {
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = a01 * a01 + a02 * a02 + a12 * a12, dia = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    auto rot = [](double &app, double &aqq, double &apq, double &arp, double &arq) {
      if (apq == 0.0) return;
      const double theta = (aqq - app) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      app -= t * apq; aqq += t * apq; apq = 0.0;
      const double rp = c * arp - s * arq, rq = s * arp + c * arq;
      arp = rp; arq = rq;
    };
    rot(a00, a11, a01, a02, a12);
    rot(a00, a22, a02, a01, a12);
    rot(a11, a22, a12, a01, a02);
  }
  lam[0] = a00; lam[1] = a11; lam[2] = a22;
  std::sort(lam, lam + 3);
}

// eigen-decomposition of the 3x3 covariance with eigenvectors (same Jacobi sweeps, rotations accumulated):
// only the strict plane test needs the normal
void eig3_vec(double a00, double a01, double a02, double a11, double a12, double a22, double lam[3], double V[3][3]) {
  double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) V[r][c] = r == c;
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    const int pq[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int t = 0; t < 3; t++) {
      const int p = pq[t][0], q = pq[t][1], r = 3 - p - q;
      if (A[p][q] == 0.0) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * c;
      A[p][p] -= tt * A[p][q]; A[q][q] += tt * A[p][q]; A[p][q] = A[q][p] = 0.0;
      const double rp = c * A[r][p] - sn * A[r][q], rq = sn * A[r][p] + c * A[r][q];
      A[r][p] = A[p][r] = rp; A[r][q] = A[q][r] = rq;
      for (int k = 0; k < 3; k++) {
        const double vp = c * V[k][p] - sn * V[k][q], vq = sn * V[k][p] + c * V[k][q];
        V[k][p] = vp; V[k][q] = vq;
      }
    }
  }
  int o[3] = {0, 1, 2};
  std::sort(o, o + 3, [&](int x, int y) { return A[x][x] < A[y][y]; });
  double Vs[3][3];
  for (int k = 0; k < 3; k++) { lam[k] = A[o[k]][o[k]]; for (int r = 0; r < 3; r++) Vs[r][k] = V[r][o[k]]; }
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) V[r][c] = Vs[r][c];
}

struct Params {
  int win = 0;
  double voxel_size = 1.0;
  float eigen_thr[4] = {1.0f / 16, 1.0f / 16, 1.0f / 16, 1.0f / 16};   // bavoxel.hpp:11 (floats)
  int layer_limit = 2, min_ps = 15;
  int layer_size[4] = {30, 30, 30, 30};
  // the consistency driver's rules (BAs_left.hpp:674, :754-792, :38); 0 = the benchmark drivers' behaviour
  double max_dis = 0, ratio21_max = 0, lam0_max = 0;
  int fix_frames = 0;
  int min_observers = 2;
};

struct Node {                         // OCTO_TREE_NODE, bavoxel.hpp:626-931
  int octo_state = 0, push_state = 0, layer = 0;
  std::vector<std::vector<V3>> vec_orig, vec_tran;
  std::vector<Cluster> sig_orig, sig_tran;
  Node *leaves[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float voxel_center[3] = {0, 0, 0};
  float quater_length = 0;
  double decision = 0;
  Cluster fix_point;
  explicit Node(int win) : vec_orig(win), vec_tran(win), sig_orig(win), sig_tran(win) {}
  ~Node() { for (Node *l : leaves) delete l; }

  bool judge_eigen(const Params &pr, int win_count) {        // :654-699 (fix cluster empty in batch mode)
    Cluster c = fix_point;
    for (int i = 0; i < win_count; i++) c.add(sig_tran[i]);
    const double n = c.N, cx = c.v[0] / n, cy = c.v[1] / n, cz = c.v[2] / n;
    double lam[3];
    if (pr.max_dis <= 0 && pr.ratio21_max <= 0 && pr.lam0_max <= 0) {
      eigvals3(c.P[0] / n - cx * cx, c.P[1] / n - cx * cy, c.P[2] / n - cx * cz, c.P[3] / n - cy * cy,
               c.P[4] / n - cy * cz, c.P[5] / n - cz * cz, lam);
      decision = lam[0] / lam[1];
      return decision < pr.eigen_thr[layer];
    }
    // src/simulation/BAs_left.hpp:647-675: the simulator's planes are exact, so the test is too
    double V[3][3];
    eig3_vec(c.P[0] / n - cx * cx, c.P[1] / n - cx * cy, c.P[2] / n - cx * cz, c.P[3] / n - cy * cy,
             c.P[4] / n - cy * cz, c.P[5] / n - cz * cz, lam, V);
    double max_dis = 0;
    for (int i = 0; i < win_count; i++)
      for (const V3 &p : vec_tran[i]) {
        const double d = std::fabs(V[0][0] * (p.x - cx) + V[1][0] * (p.y - cy) + V[2][0] * (p.z - cz));
        if (d > max_dis) max_dis = d;
      }
    decision = lam[0] / lam[1];
    return decision < pr.eigen_thr[layer] && (pr.max_dis <= 0 || max_dis < pr.max_dis) &&
           (pr.ratio21_max <= 0 || lam[2] / lam[1] < pr.ratio21_max) && (pr.lam0_max <= 0 || lam[0] < pr.lam0_max);
  }

  void cut_func(const Params &pr, int ci) {                   // :701-735
    std::vector<V3> &po = vec_orig[ci], &pt = vec_tran[ci];
    for (size_t j = 0; j < pt.size(); j++) {
      const double q[3] = {pt[j].x, pt[j].y, pt[j].z};
      int xyz[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) if (q[k] > voxel_center[k]) xyz[k] = 1;
      const int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
      if (!leaves[leafnum]) {
        Node *l = leaves[leafnum] = new Node(pr.win);
        for (int k = 0; k < 3; k++) l->voxel_center[k] = voxel_center[k] + (2 * xyz[k] - 1) * quater_length;
        l->quater_length = quater_length / 2;
        l->layer = layer + 1;
      }
      Node *l = leaves[leafnum];
      l->vec_orig[ci].push_back(po[j]);
      l->vec_tran[ci].push_back(pt[j]);
      if (l->octo_state != 1) { l->sig_orig[ci].push(po[j]); l->sig_tran[ci].push(pt[j]); }
    }
    std::vector<V3>().swap(po);
    std::vector<V3>().swap(pt);
  }

  void recut(const Params &pr, int win_count) {               // :737-776
    if (octo_state != 1) {
      int point_size = fix_point.N;                           // BAs_left.hpp:717 (0 in batch mode)
      for (int i = 0; i < win_count; i++) point_size += sig_orig[i].N;
      push_state = 0;
      if (point_size <= pr.min_ps) return;
      if (judge_eigen(pr, win_count)) {
        if (octo_state == 0 && point_size > pr.layer_size[layer]) octo_state = 2;
        point_size -= fix_point.N;
        if (point_size > pr.min_ps) push_state = 1;
        return;
      } else if (layer == pr.layer_limit) {
        octo_state = 2;
        return;
      }
      octo_state = 1;
      std::vector<Cluster>().swap(sig_orig);
      std::vector<Cluster>().swap(sig_tran);
      for (int i = 0; i < win_count; i++) cut_func(pr, i);
    } else {
      cut_func(pr, win_count - 1);
    }
    for (Node *l : leaves) if (l) l->recut(pr, win_count);
  }

  // to_margi (bavoxel.hpp:778-816): the first mg scans of a plane leaf become its world-frame fix cluster
  void marginalize(int mg, int win_count) {
    if (octo_state != 1) {
      if (fix_point.N < 50 && push_state == 1)
        for (int i = 0; i < mg; i++) fix_point.add(sig_tran[i]);
      for (int i = mg; i < win_count; i++) {
        sig_orig[i - mg] = sig_orig[i]; sig_tran[i - mg] = sig_tran[i];
        vec_orig[i - mg].swap(vec_orig[i]); vec_tran[i - mg].swap(vec_tran[i]);
      }
      for (int i = win_count - mg; i < win_count; i++) {
        sig_orig[i] = Cluster(); sig_tran[i] = Cluster();
        vec_orig[i].clear(); vec_tran[i].clear();
      }
    } else {
      for (Node *l : leaves) if (l) l->marginalize(mg, win_count);
    }
  }

  // tras_opt (:908-929) + VOX_HESS::push_voxel (:30-51)
  void collect(const Params &pr, int win_count, std::vector<const Node *> &out) const {
    if (octo_state != 1) {
      int points_size = 0;
      for (int i = 0; i < win_count; i++) points_size += sig_orig[i].N;
      if (points_size < pr.min_ps) return;
      if (push_state != 1) return;
      int process_size = 0;
      for (int i = 0; i < win_count; i++) if (sig_orig[i].N != 0) process_size++;
      if (process_size < pr.min_observers) return;
      out.push_back(this);
    } else {
      for (const Node *l : leaves) if (l) l->collect(pr, win_count, out);
    }
  }
};

struct Key {
  int64_t x, y, z;
  bool operator==(const Key &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct KeyHash {
  size_t operator()(const Key &k) const {
    return (size_t)(((uint64_t)k.z * 116101ull + (uint64_t)k.y) * 116101ull + (uint64_t)k.x);
  }
};

struct Assoc {
  Params pr;
  std::unordered_map<Key, Node *, KeyHash> map;
  std::vector<const Node *> feats;
  long n_points = 0;
  ~Assoc() { for (auto &kv : map) delete kv.second; }
};

}  // namespace

extern "C" {

void *balm_assoc_create(int win_size, double voxel_size, const float *eigen_thresholds3, int layer_limit, int min_ps) {
  Assoc *a = new Assoc();
  a->pr.win = win_size;
  a->pr.voxel_size = voxel_size;
  if (eigen_thresholds3) for (int k = 0; k < 3; k++) a->pr.eigen_thr[k] = eigen_thresholds3[k];
  a->pr.layer_limit = layer_limit;
  a->pr.min_ps = min_ps;
  return a;
}

void balm_assoc_destroy(void *h) { delete (Assoc *)h; }

// the consistency driver's rules (src/simulation/BAs_left.hpp:674 plane test; consistency.cpp:125-131 marginalisation
// of the first `fix_frames` scans into fix clusters; BAs_left.hpp:38 no observer minimum).  Call before add_frame.
void balm_assoc_set_rules(void *h, double max_dis, double ratio21_max, double lam0_max, int fix_frames, int min_observers) {
  Assoc *a = (Assoc *)h;
  a->pr.max_dis = max_dis; a->pr.ratio21_max = ratio21_max; a->pr.lam0_max = lam0_max;
  a->pr.fix_frames = fix_frames; a->pr.min_observers = min_observers;
}

// cut_voxel (bavoxel.hpp:1170-1223): xyz = n body-frame points (float x,y,z), pose = 12 doubles.
int balm_assoc_add_frame(void *h, int frame, const float *xyz, long n, const double *pose) {
  Assoc *a = (Assoc *)h;
  if (frame < 0 || frame >= a->pr.win) return 1;
  const double *R = pose, *t = pose + 9;    // R column-major
  for (long k = 0; k < n; k++) {
    const V3 po{(double)xyz[3 * k], (double)xyz[3 * k + 1], (double)xyz[3 * k + 2]};
    const V3 pt{R[0] * po.x + R[3] * po.y + R[6] * po.z + t[0], R[1] * po.x + R[4] * po.y + R[7] * po.z + t[1],
                R[2] * po.x + R[5] * po.y + R[8] * po.z + t[2]};
    const double q[3] = {pt.x, pt.y, pt.z};
    float loc[3];
    for (int j = 0; j < 3; j++) {
      loc[j] = q[j] / a->pr.voxel_size;
      if (loc[j] < 0) loc[j] -= 1.0;
    }
    const Key key{(int64_t)loc[0], (int64_t)loc[1], (int64_t)loc[2]};
    auto it = a->map.find(key);
    Node *nd;
    if (it != a->map.end()) {
      nd = it->second;
      if (nd->octo_state != 2) { nd->vec_orig[frame].push_back(po); nd->vec_tran[frame].push_back(pt); }
      if (nd->octo_state != 1) { nd->sig_orig[frame].push(po); nd->sig_tran[frame].push(pt); }
    } else {
      nd = new Node(a->pr.win);
      nd->vec_orig[frame].push_back(po); nd->vec_tran[frame].push_back(pt);
      nd->sig_orig[frame].push(po); nd->sig_tran[frame].push(pt);
      nd->voxel_center[0] = (0.5 + key.x) * a->pr.voxel_size;
      nd->voxel_center[1] = (0.5 + key.y) * a->pr.voxel_size;
      nd->voxel_center[2] = (0.5 + key.z) * a->pr.voxel_size;
      nd->quater_length = a->pr.voxel_size / 4.0;
      nd->layer = 0;
      a->map[key] = nd;
    }
  }
  a->n_points += n;
  return 0;
}

// recut + tras_opt over every root voxel (benchmark_realworld.cpp:195-200).  Returns the number of features.
int balm_assoc_finish(void *h) {
  Assoc *a = (Assoc *)h;
  a->feats.clear();
  const int mg = a->pr.fix_frames;
  for (auto &kv : a->map) {
    kv.second->recut(a->pr, a->pr.win);
    if (mg > 0) kv.second->marginalize(mg, a->pr.win);
    kv.second->collect(a->pr, a->pr.win - mg, a->feats);
  }
  return (int)a->feats.size();
}

// the features' fix clusters (F*10, all zero without marginalisation) and their points: body-frame xyz with the
// feature and (shifted) scan index of each, in push order.  Pass xyz = NULL to query the count.
void balm_assoc_export_fix(void *h, double *fix) {
  Assoc *a = (Assoc *)h;
  for (size_t f = 0; f < a->feats.size(); f++) {
    const Cluster &c = a->feats[f]->fix_point;
    double *q = fix + f * 10;
    for (int k = 0; k < 6; k++) q[k] = c.P[k];
    for (int k = 0; k < 3; k++) q[6 + k] = c.v[k];
    q[9] = c.N;
  }
}

long balm_assoc_export_points(void *h, float *xyz, int *feat, int *frame) {
  Assoc *a = (Assoc *)h;
  const int W = a->pr.win - a->pr.fix_frames;
  long n = 0;
  for (size_t f = 0; f < a->feats.size(); f++)
    for (int i = 0; i < W; i++)
      for (const V3 &p : a->feats[f]->vec_orig[i]) {
        if (xyz) {
          xyz[3 * n] = (float)p.x; xyz[3 * n + 1] = (float)p.y; xyz[3 * n + 2] = (float)p.z;
          feat[n] = (int)f; frame[n] = i;
        }
        n++;
      }
  return n;
}

// clusters F*W*10 (layout of include/balm_hip.h; W = win - fix_frames), coeffs F (= sum_i N_i, bavoxel.hpp:42-44),
// layer F (optional)
void balm_assoc_export(void *h, double *clusters, double *coeffs, int *layer) {
  Assoc *a = (Assoc *)h;
  const int W = a->pr.win - a->pr.fix_frames;
  for (size_t f = 0; f < a->feats.size(); f++) {
    const Node *nd = a->feats[f];
    double coe = 0;
    for (int i = 0; i < W; i++) {
      const Cluster &c = nd->sig_orig[i];
      double *q = clusters + (f * W + i) * 10;
      for (int k = 0; k < 6; k++) q[k] = c.P[k];
      for (int k = 0; k < 3; k++) q[6 + k] = c.v[k];
      q[9] = c.N;
      coe += c.N;
    }
    coeffs[f] = coe;
    if (layer) layer[f] = nd->layer;
  }
}

}  // extern "C"
