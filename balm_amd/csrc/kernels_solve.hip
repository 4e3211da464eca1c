// Damped LM solve on gfx950:  (H + u diag H) dx = -g ,  q1 = 0.5 dx.(u D dx - g)
//   reference: src/benchmark/bavoxel.hpp:1113-1114,1127  (Eigen `.ldlt().solve()`)
//
// Eigen's LDLT (in-place lower, unblocked) searches its pivot on the NOT-yet-updated trailing
// diagonal (its update is left-looking), i.e. the elimination order is simply "decreasing
// |diagonal| of the input matrix".  We therefore apply that static symmetric permutation once and
// run an un-pivoted *blocked* LDL^T (D diagonal, possibly negative: the exact second-order Hessian
// is indefinite away from the optimum, SURVEY.md finding 4):
//   per panel of NB=48 columns:  ldl_panel (rank-4 steps on f64 MFMA) L11, D11 (redundantly per workgroup),
//                                                                     W21 = L21 D11, L21
//                                ldl_trail (f64 MFMA, 48x16 tiles)    A22 -= L21 W21^T
//   Two things ride along as extra rows below the matrix (ldA = 2 nA + NB rows of storage):
//     row nA          the right-hand side: the panel / trail kernels produce z = D^+ L^-1 P b for free;
//     rows nA+NB ..   the identity: its rows come out as M = L^-T D^+ (upper triangular; tile t is only touched
//                     from panel t on, so the factorisation carries ~25 % more trail tiles and no extra
//                     launches), which turns the backward solve  L^T x = z  -- a chain of one launch per panel
//                     -- into ONE matrix-vector product  x = M (D z)  (k_ldl_apply).
// Pose update kernels (bavoxel.hpp:1116-1126, 1159-1164) live here too.
#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>

#include "balm_internal.h"

namespace balm {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// permutation by decreasing |diag H| (ties by index); padded positions (>= n) go last.
// 16 ranks per workgroup, the j-range split sixteen ways (the kernel is on the critical path of every solve and
// has no other parallelism to offer: nA / 16 workgroups instead of nA / 64).  The diagonal's loads are unrolled over a
// compile-time bound (MAXI: nA <= 256 MAXI) so that all of them are in flight at once -- as a rolled loop every
// workgroup walked nA / 256 dependent memory round trips: 23 us at n = 3000 (round 3), 10.5 -> 9 us at n = 1200.
// ------------------------------------------------------------------------------------------------
template <int MAXI>
__global__ __launch_bounds__(256) void k_rank_diag(const double *__restrict__ H, int n, int nA, int *__restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) double dabs[];   // [nA] then int part[256]
  int *part = reinterpret_cast<int *>(dabs + nA);
  double dv[MAXI];
#pragma unroll
  for (int it = 0; it < MAXI; it++) {
    const int i = threadIdx.x + 256 * it;
    dv[it] = i < n ? fabs(H[(size_t)i * n + i]) : -1.0;
  }
  // a NaN diagonal ranks after every real entry and before the padding, so that perm stays a permutation
#pragma unroll
  for (int it = 0; it < MAXI; it++) {
    const int i = threadIdx.x + 256 * it;
    if (i < nA) dabs[i] = dv[it] == dv[it] ? dv[it] : -0.5;
  }
  __syncthreads();
  const int il = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il;
  int rank = 0;
  if (i < nA) {
    const double di = dabs[i];
    const int chunk = (nA + 15) / 16;
    const int j0 = q * chunk, j1 = min(nA, j0 + chunk);
    for (int j = j0; j < j1; j++) {
      const double dj = dabs[j];
      rank += (dj > di) || (dj == di && j < i);
    }
  }
  part[threadIdx.x] = rank;
  __syncthreads();
  if (q == 0 && i < nA) {
    rank = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) rank += part[16 * k + il];
    perm[rank] = i;
  }
}

// ------------------------------------------------------------------------------------------------
// The damped, symmetrically permuted matrix with its right-hand side (and, column-major, the identity rows that become
// L^-T D^+):  A[r][c] = H[perm[r]][perm[c]] (+ u H[..] on the diagonal: D = diag(H), bavoxel.hpp:1113), row block P = -g[perm[c]].
// ONE workgroup builds ONE column c: row perm[c] of H is contiguous (H is symmetric), goes into LDS with coalesced loads -- all in
// flight at once: the loop is unrolled over a compile-time bound, MAXI: nA <= 256 MAXI -- and the column is the LDS gather
// hrow[perm[r]], written in 384-byte segments.  (Round 3's form gathered H[perm[c] * n + perm[r]] from memory, 8 bytes out of every
// 64-byte sector: 114 us at n = 3000 = 145 MB at 1.3 TB/s; this one, measured in round 4: n = 3000 solve 0.992 -> 0.905 ms, n = 4800
// 3.01 -> 2.68, n = 1200 0.277 -> 0.273, profiles/r04a_solve_switches.txt.)  It also zeroes the flags of the persistent kernels and
// fills the exchange buffers whose payload is its own flag (k_ldl_backsolve's x, k_ldl_chain's xtile) with "not there yet" = all ones.
// tiled: 0 = column-major [A ; rhs ; identity] (ldA = 2 nA + 48), what k_ldl_chain with identity rows, k_ldl_fused, the launch path,
// k_ldl_apply and the covariance read; 1 = tile-major [A ; rhs] (kernels_chain.inc: ch_tile) for k_ldl_chain + k_ldl_backsolve, without
// the tiles above the diagonal -- nothing on that path reads them (-54 us of the 114 at n = 3000 by itself).
// ------------------------------------------------------------------------------------------------
template <int MAXI>
__global__ __launch_bounds__(256) void k_build_A(const double *__restrict__ H, const double *__restrict__ g, int n, int nA,
                                                 const int *__restrict__ perm, const double *__restrict__ pu, double u_arg,
                                                 double *__restrict__ A, int *__restrict__ flags, int nflags, double *__restrict__ xs,
                                                 double *__restrict__ xtile, int tiled) {
  extern __shared__ __attribute__((aligned(16))) double hrow[];       // [n] row perm[c] of H
  const double u = pu ? *pu : u_arg;    // replayed hipGraphs read the damping from device memory (the launch sequence of an LM
                                        // iteration is then the same for every iteration); plain launches carry it as an argument
  const int tid = threadIdx.x, c = blockIdx.x;
  for (long t = (long)c * 256 + tid; t < nflags; t += (long)gridDim.x * 256) flags[t] = 0;
  for (long t = (long)c * 256 + tid; t < nA; t += (long)gridDim.x * 256) xs[t] = __longlong_as_double(-1ll);
  if (tid < NB) {      // k_ldl_chain's exchange tiles [P][48][48] = nA x 48 doubles each: xtile all ones; Minv_p (in front of it) all ones in
                       // the upper 16 x 16 sub-tiles -- what the riders will write and the row workgroups poll -- and zero below
    xtile[(size_t)c * NB + tid] = __longlong_as_double(-1ll);
    const int k = c % NB;                                                        // element (k, tid) of panel c / 48
    (xtile - (size_t)nA * NB)[(size_t)c * NB + tid] = (k / 16 <= tid / 16) ? __longlong_as_double(-1ll) : 0.0;
  }
  const int P = nA / NB, cb = c / NB, cl = c - NB * cb;
  const int ldA = 2 * nA + NB;
  const int pc = perm[c];
  const int r_begin = tiled ? NB * cb : 0;
  double hv[MAXI];
  int prv[MAXI];
  const double *row = H + (size_t)(pc < n ? pc : 0) * n;
#pragma unroll
  for (int it = 0; it < MAXI; it++) {
    const int i = tid + 256 * it, r = r_begin + i;
    hv[it] = (i < n && pc < n) ? row[i] : 0.0;
    prv[it] = r < nA ? perm[r] : n;
  }
  const double gv = pc < n ? -g[pc] : 0.0;
#pragma unroll
  for (int it = 0; it < MAXI; it++) {
    const int i = tid + 256 * it;
    if (i < n) hrow[i] = hv[it];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MAXI; it++) {               // the rows of A proper
    const int r = r_begin + tid + 256 * it;
    if (r < nA) {
      const int pr = prv[it];
      double v;
      if (pr < n && pc < n) {
        v = hrow[pr];
        if (r == c) v += u * v;                     // D = diag(H)  (bavoxel.hpp:1113)
      } else {
        v = (r == c) ? 1.0 : 0.0;
      }
      const int rb = r / NB, rl = r - NB * rb;
      A[tiled ? ((size_t)cb * (P + 1) + rb) * (NB * NB) + (size_t)cl * NB + rl : (size_t)c * ldA + r] = v;
    }
  }
  const int rows = tiled ? nA + NB : ldA;           // the right-hand side rows and (column-major) the identity rows: no loads
  for (int r = nA + tid; r < rows; r += 256) {
    const double v = r < nA + NB ? (r == nA ? gv : 0.0) : ((r - (nA + NB) == c) ? 1.0 : 0.0);
    const int rb = P + (r - nA) / NB, rl = (r - nA) % NB;         // row block P = right-hand side
    A[tiled ? ((size_t)cb * (P + 1) + rb) * (NB * NB) + (size_t)cl * NB + rl : (size_t)c * ldA + r] = v;
  }
}

// wave-uniform broadcast of lane `src`'s value (src is a compile-time constant after unrolling)
__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------------
// ldl_panel: LDL^T of the NB x NB diagonal block AND of this workgroup's 64 rows below it, by twelve
// rank-4 steps whose trailing updates are single v_mfma_f64_16x16x4_f64 instructions (K = 4 is the
// MFMA's k extent).  A workgroup (4 waves) keeps 7 row tiles of 16 rows -- tiles 0..2 = the diagonal
// block (factored redundantly by every workgroup), tiles 3..6 = its own rows -- times 3 column tiles
// in MFMA accumulators, transposed: accumulator element (lane, reg) of tile (rt, ct) is matrix
// element (row 16 rt + (lane & 15), column 16 ct + (lane >> 4) + 4 reg), so that global loads and
// stores touch 128 contiguous bytes per 16 lanes.  Per step q (columns 4q..4q+3):
//   A  every lane publishes its one element of the four current columns to LDS;
//   B  every lane factors the 4x4 pivot block redundantly in registers (no cross-lane traffic),
//      lanes 0..111 turn their own row into W = row M4^T (= L D) and L = W D^-1, write the MFMA
//      operands to LDS and stream L21 / W21 / L11 / D to global memory;
//   C  acc(rt, ct) += L_op(ct) x (-W_op(rt))  -- one MFMA per live tile.
// ------------------------------------------------------------------------------------------------
constexpr int PANEL_ROWS = 64;                     // rows below the diagonal block per workgroup
constexpr int PANEL_LR = NB + PANEL_ROWS;          // local rows: 48 diagonal + 64 own = 7 tiles of 16
static_assert(PANEL_LR == 112 && NB == 48, "k_ldl_panel is written for 3 + 4 row tiles of 16");

__device__ __forceinline__ double rcp_nr(double d) {        // 1/d: v_rcp_f64 + two Newton steps (<= 1 ulp)
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-d, x, 1.0), x, x);
  return (fabs(d) > DBL_MIN) ? x : 0.0;                      // Eigen's D^+ rule for a vanished pivot
}

// LDL^T of the 4x4 pivot block of a rank-4 step (D^+ rule per pivot).  Measured and rejected: the four pivots from
// division-free leading minors with four independent reciprocals (dependency depth 14 instead of 33): 100 ns SLOWER per
// step -- the phase is bound by the number of f64 VALU operations a lone wave issues (~3.3 ns each), not by their
// dependencies, and the minor form has ~30 more of them.
struct Pivot4 { double d0, d1, d2, d3, i0, i1, i2, i3, l10, l20, l30, l21, l31, l32; };

__device__ __forceinline__ void pivot4(double a00, double a10, double a20, double a30, double a11, double a21, double a31,
                                       double a22, double a32, double a33, Pivot4 &o) {
  o.d0 = a00; o.i0 = rcp_nr(o.d0);
  o.l10 = a10 * o.i0; o.l20 = a20 * o.i0; o.l30 = a30 * o.i0;
  o.d1 = __builtin_fma(-o.l10, a10, a11); o.i1 = rcp_nr(o.d1);
  const double w21 = __builtin_fma(-o.l20, a10, a21), w31 = __builtin_fma(-o.l30, a10, a31);
  o.l21 = w21 * o.i1; o.l31 = w31 * o.i1;
  o.d2 = __builtin_fma(-o.l21, w21, __builtin_fma(-o.l20, a20, a22)); o.i2 = rcp_nr(o.d2);
  const double w32 = __builtin_fma(-o.l31, w21, __builtin_fma(-o.l30, a20, a32));
  o.l32 = w32 * o.i2;
  o.d3 = __builtin_fma(-o.l32, w32, __builtin_fma(-o.l31, w31, __builtin_fma(-o.l30, a30, a33)));
  o.i3 = rcp_nr(o.d3);
}

// a row's rank-4 pieces: w = L4^-1 r (forward substitution), e = w D4^-1
__device__ __forceinline__ void row4(const Pivot4 &v, double r0, double r1, double r2, double r3, double &w0, double &w1,
                                     double &w2, double &w3, double &e0, double &e1, double &e2, double &e3) {
  w0 = r0;
  w1 = __builtin_fma(-v.l10, r0, r1);
  w2 = __builtin_fma(-v.l21, w1, __builtin_fma(-v.l20, r0, r2));
  w3 = __builtin_fma(-v.l32, w2, __builtin_fma(-v.l31, w1, __builtin_fma(-v.l30, r0, r3)));
  e0 = w0 * v.i0; e1 = w1 * v.i1; e2 = w2 * v.i2; e3 = w3 * v.i3;
}

__device__ __forceinline__ void ldl_panel_body(double *__restrict__ A, int nA, int c0, int nR, double *__restrict__ dvec,
                                               double *__restrict__ Wp, double *__restrict__ zvec, const int bx) {
  __shared__ double cur[4][PANEL_LR];     // the four current columns, all local rows
  __shared__ double Wop[4][PANEL_LR];     // -W (B operand), k-major
  __shared__ double Lop[4][PANEL_LR];     //  L (A operand; only local rows < NB are read)
  const int ldA = 2 * nA + NB;           // nR = live rows: matrix, right-hand side tile, identity tiles <= this panel
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rbase = c0 + NB + bx * PANEL_ROWS;                // first own row
  auto grow = [&](int lr) { return lr < NB ? c0 + lr : rbase + (lr - NB); };   // local -> global row

  // ---- load the 7 x 3 tiles into accumulators (wave w owns row tiles w and w + 4) ----
  d4 acc[2][3];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int rt = wv + 4 * s;
    const int gr = grow(rt * 16 + l15);
    const bool ok = rt < 7 && gr < nR;
#pragma unroll
    for (int ct = 0; ct < 3; ct++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        acc[s][ct][e] = ok ? A[(size_t)(c0 + ct * 16 + l4 + 4 * e) * ldA + (ok ? gr : 0)] : 0.0;
  }

#pragma unroll
  for (int q = 0; q < NB / 4; q++) {
    const int jt = q >> 2, qq = q & 3;
    // ---- A: publish columns 4q..4q+3 (lane holds row rt*16 + l15, column 4q + l4) ----
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int rt = wv + 4 * s;
      if (rt < 7) cur[l4][rt * 16 + l15] = acc[s][jt][qq];
    }
    __syncthreads();
    // ---- B: 4x4 pivot block (rows 4q..4q+3 of the diagonal block), redundantly per lane ----
    const int p0 = 4 * q;
    const double a00 = cur[0][p0], a10 = cur[0][p0 + 1], a20 = cur[0][p0 + 2], a30 = cur[0][p0 + 3];
    const double a11 = cur[1][p0 + 1], a21 = cur[1][p0 + 2], a31 = cur[1][p0 + 3];
    const double a22 = cur[2][p0 + 2], a32 = cur[2][p0 + 3], a33 = cur[3][p0 + 3];
    const int lrq = tid < PANEL_LR ? tid : PANEL_LR - 1;     // this thread's row, requested together with the pivot block
    double r0 = cur[0][lrq], r1 = cur[1][lrq], r2 = cur[2][lrq], r3 = cur[3][lrq];
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    Pivot4 pv;
    pivot4(a00, a10, a20, a30, a11, a21, a31, a22, a32, a33, pv);
    const double d0 = pv.d0, d1 = pv.d1, d2 = pv.d2, d3 = pv.d3;
    if (tid < PANEL_LR) {
      const int lr = tid;
      double w0 = 0, w1 = 0, w2 = 0, w3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
      const bool below = lr > p0 + 3;                      // rows still to be eliminated
      if (below) row4(pv, r0, r1, r2, r3, w0, w1, w2, w3, e0, e1, e2, e3);
      Wop[0][lr] = -w0; Wop[1][lr] = -w1; Wop[2][lr] = -w2; Wop[3][lr] = -w3;
      Lop[0][lr] = e0; Lop[1][lr] = e1; Lop[2][lr] = e2; Lop[3][lr] = e3;
      // ---- stream the results out ----
      const int gr = grow(lr);
      const size_t cA = (size_t)(c0 + p0) * ldA + gr;
      if (lr >= NB) {                                       // own rows: L21 in place, W21 for ldl_trail
        if (gr < nR) {
          A[cA] = e0; A[cA + ldA] = e1; A[cA + 2 * (size_t)ldA] = e2; A[cA + 3 * (size_t)ldA] = e3;
          const size_t cW = (size_t)p0 * ldA + gr;
          Wp[cW] = w0; Wp[cW + ldA] = w1; Wp[cW + 2 * (size_t)ldA] = w2; Wp[cW + 3 * (size_t)ldA] = w3;
          if (gr == nA) {                                   // right-hand side row: z = D^+ L^-1 P b
            zvec[c0 + p0] = e0; zvec[c0 + p0 + 1] = e1; zvec[c0 + p0 + 2] = e2; zvec[c0 + p0 + 3] = e3;
          }
        }
      } else if (bx == 0 && lr >= p0 && lr <= p0 + 3) {
        // the diagonal block's D, once.  L11 is NOT written back: nothing downstream reads it, and the other
        // workgroups of this launch may not have loaded the un-factored diagonal block yet (they start late when
        // other streams share the device: a sharded context's replicated solves) -- an in-place L11 would race
        // with their loads.
        const int e = lr - p0;
        dvec[c0 + lr] = e == 0 ? d0 : (e == 1 ? d1 : (e == 2 ? d2 : d3));
      }
    }
    __syncthreads();
    // ---- C: rank-4 update of every live tile: one MFMA each ----
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int rt = wv + 4 * s;
      if (rt < 7) {
        const double bop = Wop[l4][rt * 16 + l15];
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
          if (ct >= jt) {
            const double aop = Lop[l4][ct * 16 + l15];
            acc[s][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[s][ct], 0, 0, 0);
          }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_ldl_panel(double *__restrict__ A, int nA, int c0, int nR,
                                                   double *__restrict__ dvec, double *__restrict__ Wp,
                                                   double *__restrict__ zvec) {
  ldl_panel_body(A, nA, c0, nR, dvec, Wp, zvec, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// ldl_trail: A22 -= W21 L21^T on the lower triangle; one wavefront per 48 (rows i) x 16 (cols j)
// tile, f64 MFMA, every operand prefetched before the first MFMA (the kernel is latency-bound).
//   D[m][nn] = sum_k L21[j0+m][k] * W21[i0+nn][k]  -> element (i0+nn, j0+m); lanes of a row group
//   touch 128 contiguous bytes of a column.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ldl_trail_body(double *__restrict__ A, int nA, int c0, const double *__restrict__ Wp, int mt,
                                               int tj_begin, int tj_end, const int bx, const int by) {
  const int ldA = 2 * nA + NB;
  const int lane = threadIdx.x & 63;
  const int ti = by;                                          // 48-row tile; ti == mt: right-hand side tile
  const int tj = tj_begin + bx * 4 + (threadIdx.x >> 6);      // 16-col tile, [tj_begin, tj_end) of the trailing square
  if (tj >= tj_end) return;
  if (ti < mt ? (tj > 3 * ti + 2) : (tj >= 3 * mt)) return;
  const int base = c0 + NB;
  const int i0 = base + ti * NB, j0 = base + tj * 16;
  const double *pl = A + (size_t)(c0 + (lane >> 4)) * ldA + j0 + (lane & 15);    // L21 rows j
  const double *pw = Wp + (size_t)(lane >> 4) * ldA + i0 + (lane & 15);          // W21 rows i
  double a[NB / 4], b[NB / 4][3];
#pragma unroll
  for (int ks = 0; ks < NB / 4; ks++) {
    a[ks] = pl[(size_t)ks * 4 * ldA];
#pragma unroll
    for (int q = 0; q < 3; q++) b[ks][q] = pw[(size_t)ks * 4 * ldA + 16 * q];
  }
  // the tile itself (read-modify-write), also in flight before the MFMAs
  double old[3][4];
#pragma unroll
  for (int y = 0; y < 3; y++)
#pragma unroll
    for (int e = 0; e < 4; e++)
      old[y][e] = A[(size_t)(j0 + (lane >> 4) + 4 * e) * ldA + i0 + 16 * y + (lane & 15)];
  d4 acc[3];
#pragma unroll
  for (int y = 0; y < 3; y++) acc[y] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < NB / 4; ks++)
#pragma unroll
    for (int y = 0; y < 3; y++) acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks][y], acc[y], 0, 0, 0);
#pragma unroll
  for (int y = 0; y < 3; y++)
#pragma unroll
    for (int e = 0; e < 4; e++)
      A[(size_t)(j0 + (lane >> 4) + 4 * e) * ldA + i0 + 16 * y + (lane & 15)] = old[y][e] - acc[y][e];
}

__global__ __launch_bounds__(256) void k_ldl_trail(double *__restrict__ A, int nA, int c0,
                                                   const double *__restrict__ Wp, int mt, int tj_begin, int tj_end) {
  ldl_trail_body(A, nA, c0, Wp, mt, tj_begin, tj_end, (int)blockIdx.x, (int)blockIdx.y);
}

// Lookahead in one launch: the first `npb` workgroups factor panel p (its 48 columns already carry panel p-1's update: the
// small launch A(p-1) before this one), the others apply panel p-1 to everything to the RIGHT of those columns (16-col tiles
// >= 3 of its trailing square).  The twelve latency-bound steps of panel p run beside the bulk of panel p-1's trailing update
// instead of behind it.  The two parts touch disjoint columns; W21 of both panels is alive: two buffers.
__global__ __launch_bounds__(256) void k_ldl_panel_trail(double *__restrict__ A, int nA, int c0, int nR, double *__restrict__ dvec,
                                                         double *__restrict__ Wp, double *__restrict__ zvec, int npb, int c0_prev,
                                                         const double *__restrict__ Wp_prev, int mt_prev, int gx_prev) {
  const int b = (int)blockIdx.x;
  if (b < npb) {
    __builtin_amdgcn_s_setprio(3);          // the panel's waves are the critical path: issue before the trailing tiles' waves
    ldl_panel_body(A, nA, c0, nR, dvec, Wp, zvec, b);
  } else {
    const int t = b - npb;
    ldl_trail_body(A, nA, c0_prev, Wp_prev, mt_prev, 3, 3 * mt_prev, t % gx_prev, t / gx_prev);
  }
}

// ------------------------------------------------------------------------------------------------
// k_ldl_fused: the whole blocked factorisation (every panel, every trailing update) in ONE persistent launch.
//
// The per-panel launch pair above is a chain of 2 P dependent kernels whose cost is launch + first-load latency
// (~5 us each), not work.  Here the workgroups of one cooperative launch talk through flags in HBM instead
// (0.7-0.8 us per hop on gfx950, profiles/r02a_ubench_sync.txt):
//
//   panel workgroup rb (one per 48-row block of the tall matrix [A ; rhs tile ; identity]) walks the column blocks p in
//     which its tile (rb, p) is live.  Per column: it loads its tile and the diagonal tile (p, p) -- both carry the "far"
//     updates of panels <= p-2, applied by the helpers -- subtracts the "near" update of panel p-1 itself (its own
//     L[rb, p-1] is still in LDS, L[p, p-1] comes from workgroup p), then runs the twelve rank-4 steps of k_ldl_panel:
//     the diagonal block redundantly, its own 48 rows riding along (TRSM for free).  No tile travels through memory
//     between the near update and the factorisation, and nobody waits for a trailing-update kernel: the critical path
//     per panel is   factor -> flag -> 18 KB of L[p+1, p] -> 180 MFMAs -> factor.
//   helper workgroups own the tiles to the right of the next column: tile (rb, j) -= L[rb, q] D_q L[j, q]^T for every
//     panel q <= j-2, in order, each as soon as the two L blocks are published.  One helper owns one tile for the whole
//     factorisation (fixed owner = program order between its updates, no flags among helpers).
//   done[rb][p]  set by panel workgroup rb when L[rb, p] (and, for rb == p, L11 and D) are in memory
//   far[rb][j]   set by the owner of tile (rb, j) after its last far update (q == j-2)
// Every wait is on a strictly earlier column, all workgroups are co-resident (cooperative launch): no deadlock.
// Same elimination order, pivot-block code and D^+ rule as the launch pair; a tile's updates arrive in panel order.  The
// one difference: W = L D is re-formed from L (one rounding) instead of being kept from the step that produced it, so
// the two paths agree to ~1e-15, not bit for bit (tests/test_gpu_solve.py compares them and both with LAPACK).
//
// Measured (profiles/r02*_solve*.txt, n = 1200): 14 us per panel = 8.2 us for the twelve rank-4 steps (a step is
// ~0.7 us: ~110 VALU instructions of ONE wavefront -- pivot recurrence, the rows' forward substitution, operand
// staging -- issue-bound at ~3 ns each, plus two LDS round trips of ~100 ns; not flops, not barriers), 3.5 us near
// update (18 KB of L[p, p-1] from the neighbour + 180 MFMAs on one CU), ~1.5 us flag hop / tile loads / write-out.
// The launch pair spends ~18.6 us per panel (since its panel kernel lost the in-place L11 store and got the row reads
// hoisted: 0.47 ms at n = 1200, 0.51 before; 0.455 with the lookahead of launch_factor).  Fused wins for 18 <= P <= 40
// panels (0.43 vs 0.455 ms at n = 1200, 0.39 vs 0.41 at the shipped window's n = 1062, 0.57 vs 0.64 at n = 1536); below that
// its fixed cost (cooperative launch, idle helpers) shows, above it the helpers (one CU per tile update, operands re-read
// per update) fall behind the trailing-update kernel (1.07 vs 0.94 ms at n = 2100).
// ------------------------------------------------------------------------------------------------
constexpr int FT = 576;                         // 9 wavefronts: one per 16x16 sub-tile of a 48x48 tile
constexpr int FLR = 2 * NB;                     // local rows of a panel workgroup: 48 diagonal + 48 own

__device__ __forceinline__ int flag_peek(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_wait(const int *p) {
  while (flag_peek(p) == 0) __builtin_amdgcn_s_sleep(1);
}
// Publishing without an L2 write-back: what other workgroups will read is stored write-through (agent-scope stores
// carry sc1 on gfx950), so "all my stores have been acknowledged" (vmcnt 0) + workgroup barrier is a release, and the
// flag itself is another write-through store.  (buffer_wbl2 per publish cost 3-5 us: it writes back every dirty line of
// the XCD's L2, the helpers' tiles included.)
__device__ __forceinline__ void st_wt(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish(int *flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FusedArgs {
  double *A; double *dvec; double *zvec; int *done; int *far;
  int nA, P, RB, NH;
  long long *trace;       // diagnostics (BALM_SOLVE_TRACE): [RB][P][6] wall-clock ticks of a panel workgroup's phases, or null
};
#define FUSED_TRACE(slot) do { if (a.trace && tid == 0) a.trace[((size_t)rb * P + p) * 6 + (slot)] = wall_clock64(); } while (0)

// sub-tiles of a panel workgroup's 96 x 48 window (row tile rt: 0..2 diagonal block, 3..5 own rows; column tile ct),
// the strictly upper ones of the diagonal block left out.  Slots 0..8 (one per wave) are the tiles with ct >= 1, which
// stay live while the factorisation moves right; slots 9..14 (second slot of waves 0..5) are the ct == 0 tiles.
__device__ __forceinline__ void fused_slot(int s, int &rt, int &ct) {
  const int RT[15] = {1, 2, 3, 4, 5, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5};
  const int CT[15] = {1, 1, 1, 1, 1, 2, 2, 2, 2, 0, 0, 0, 0, 0, 0};
  rt = RT[s]; ct = CT[s];
}

__device__ void fused_panel_role(const FusedArgs &a, int rb, double *lds) {
  const int nA = a.nA, P = a.P, ldA = 2 * nA + NB;
  double *__restrict__ A = a.A;
  double (*cur)[FLR] = reinterpret_cast<double (*)[FLR]>(lds);             // [4][96] the four current columns
  double (*Wop)[FLR] = reinterpret_cast<double (*)[FLR]>(lds + 4 * FLR);   // [4][96] -W
  double (*Lop)[FLR] = reinterpret_cast<double (*)[FLR]>(lds + 8 * FLR);   // [4][96]  L
  double (*Lsave)[NB] = reinterpret_cast<double (*)[NB]>(lds + 12 * FLR);              // [48 k][48 own rows]  L[rb, p]
  double (*Lnext)[NB] = reinterpret_cast<double (*)[NB]>(lds + 12 * FLR + NB * NB);     // [48 k][48 rows]      L[p, p-1]
  double *dsave = lds + 12 * FLR + 2 * NB * NB;                                          // [48] D of the last panel
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tI = rb - P - 1;                                 // identity block index (rb > P)
  const int c_first = rb <= P ? 0 : tI, c_last = rb < P ? rb : P - 1;
  const int rowbase = rb < P ? NB * rb : (rb == P ? nA : nA + NB + NB * tI);
  int srt[2], sct[2];
  fused_slot(wv, srt[0], sct[0]);
  const bool two = wv < 6;
  if (two) fused_slot(9 + wv, srt[1], sct[1]); else { srt[1] = 5; sct[1] = 0; }

  for (int p = c_first; p <= c_last; p++) {
    const int c0 = NB * p;
    const bool own = rb != p;                                // rb == p: the diagonal block's owner, no rows below
    const bool had_prev = p > c_first;                       // L[rb, p-1] exists (still in Lsave)
    FUSED_TRACE(0);
    // ---- tiles with their far updates ----
    if (tid == 0) {
      // the (up to four) flags of a column are polled together: a flag read is a round trip to memory
      const int *f0 = (own && p - c_first >= 2) ? a.far + (size_t)rb * P + p : nullptr;
      const int *f1 = p >= 2 ? a.far + (size_t)p * P + p : nullptr;
      const int *f2 = (p >= 1 && rb != p) ? a.done + (size_t)p * P + (p - 1) : nullptr;             // L[p, p-1]
      const int *f3 = (p >= 1 && !had_prev) ? a.done + (size_t)(p - 1) * P + (p - 1) : nullptr;     // D of panel p-1 from its owner
      for (;;) {
        const int v0 = f0 ? flag_peek(f0) : 1, v1 = f1 ? flag_peek(f1) : 1, v2 = f2 ? flag_peek(f2) : 1, v3 = f3 ? flag_peek(f3) : 1;
        if (v0 & v1 & v2 & v3) break;
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    FUSED_TRACE(1);
    d4 acc[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int rt = srt[s], ct = sct[s];
      const bool live = (s == 0 || two) && (rt < 3 || own);
      const int gr = rt < 3 ? c0 + rt * 16 + l15 : rowbase + (rt - 3) * 16 + l15;
#pragma unroll
      for (int e = 0; e < 4; e++)
        acc[s][e] = live ? A[(size_t)(c0 + ct * 16 + l4 + 4 * e) * ldA + gr] : 0.0;
    }
    FUSED_TRACE(2);
    // ---- near update: panel p-1 on the diagonal tile (always) and on the own tile (if it was live there) ----
    if (p >= 1) {
      const int cp = c0 - NB;
      if (rb == p) {                                         // own rows of the last column ARE block p
        for (int i = tid; i < NB * NB; i += FT) (&Lnext[0][0])[i] = (&Lsave[0][0])[i];
      } else {
        for (int i = tid; i < NB * NB; i += FT) {
          const int k = i / NB, r = i - k * NB;
          Lnext[k][r] = A[(size_t)(cp + k) * ldA + c0 + r];
        }
      }
      if (!had_prev && tid < NB) dsave[tid] = a.dvec[cp + tid];
      __syncthreads();
      // a dependent MFMA chain issues every ~81 ns, independent ones every ~60: the k range of every sub-tile goes to
      // two accumulators, and the two sub-tiles of a wave alternate
      d4 part[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
      bool live[2];
#pragma unroll
      for (int s = 0; s < 2; s++) live[s] = (s == 0 || two) && (srt[s] < 3 || (own && had_prev));
#pragma unroll
      for (int kc = 0; kc < NB / 4; kc++) {
        const int k = 4 * kc + l4;
        const double dk = dsave[k];
#pragma unroll
        for (int s = 0; s < 2; s++) {
          if (live[s]) {
            const int rt = srt[s], ct = sct[s];
            const double aop = Lnext[k][ct * 16 + l15];
            const double lrow = rt < 3 ? Lnext[k][rt * 16 + l15] : Lsave[k][(rt - 3) * 16 + l15];
            if (kc & 1) part[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, -(lrow * dk), part[s], 0, 0, 0);
            else acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, -(lrow * dk), acc[s], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 2; s++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[s][e] += part[s][e];
      __syncthreads();                                       // Lsave / dsave are rewritten by the steps below
    }
    FUSED_TRACE(3);
    long long tph[6] = {0, 0, 0, 0, 0, 0};
    const bool tracer = a.trace && rb == P && tid == 0;
    // ---- twelve rank-4 steps (k_ldl_panel's arithmetic) ----
#pragma unroll
    for (int q = 0; q < NB / 4; q++) {
      const int jt = q >> 2, qq = q & 3;
      long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
      if (tracer) ts0 = wall_clock64();
#pragma unroll
      for (int s = 0; s < 2; s++)
        if ((s == 0 || two) && sct[s] == jt) cur[l4][srt[s] * 16 + l15] = acc[s][qq];
      __syncthreads();
      if (tracer) ts1 = wall_clock64();
      const int p0 = 4 * q;
      const double a00 = cur[0][p0], a10 = cur[0][p0 + 1], a20 = cur[0][p0 + 2], a30 = cur[0][p0 + 3];
      const double a11 = cur[1][p0 + 1], a21 = cur[1][p0 + 2], a31 = cur[1][p0 + 3];
      const double a22 = cur[2][p0 + 2], a32 = cur[2][p0 + 3], a33 = cur[3][p0 + 3];
      // this thread's row, requested together with the pivot block (an LDS round trip is ~100 ns in this kernel: it must
      // overlap with the pivot chain instead of following it)
      const int lrq = tid < FLR ? tid : FLR - 1;
      double r0 = cur[0][lrq], r1 = cur[1][lrq], r2 = cur[2][lrq], r3 = cur[3][lrq];
      asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));     // keep the loads up here
      Pivot4 pv;
      pivot4(a00, a10, a20, a30, a11, a21, a31, a22, a32, a33, pv);
      const double d0 = pv.d0, d1 = pv.d1, d2 = pv.d2, d3 = pv.d3;
      if (tid < FLR) {
        const int lr = tid;
        double w0 = 0, w1 = 0, w2 = 0, w3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
        const bool below = lr > p0 + 3;
        if (below) row4(pv, r0, r1, r2, r3, w0, w1, w2, w3, e0, e1, e2, e3);
        Wop[0][lr] = -w0; Wop[1][lr] = -w1; Wop[2][lr] = -w2; Wop[3][lr] = -w3;
        Lop[0][lr] = e0; Lop[1][lr] = e1; Lop[2][lr] = e2; Lop[3][lr] = e3;
        if (lr >= NB) {                                      // own rows: L kept in LDS (next near update, write-out below)
          const int orow = lr - NB;
          Lsave[p0][orow] = e0; Lsave[p0 + 1][orow] = e1; Lsave[p0 + 2][orow] = e2; Lsave[p0 + 3][orow] = e3;
        } else if (lr >= p0 && lr <= p0 + 3) {
          dsave[lr] = lr == p0 ? d0 : (lr == p0 + 1 ? d1 : (lr == p0 + 2 ? d2 : d3));
        }
      }
      if (tracer) ts2 = wall_clock64();
      __syncthreads();
      if (tracer) ts3 = wall_clock64();
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int rt = srt[s], ct = sct[s];
        if ((s == 0 || two) && ct >= jt) {
          const double bop = Wop[l4][rt * 16 + l15];
          const double aop = Lop[l4][ct * 16 + l15];
          acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[s], 0, 0, 0);
        }
      }
      if (tracer) {
        const long long ts4 = wall_clock64() + (long long)(acc[0][0] * 0.0);      // after the MFMA result is readable
        tph[0] += ts1 - ts0; tph[1] += ts2 - ts1; tph[2] += ts3 - ts2; tph[3] += ts4 - ts3;

      }
    }
    if (tracer) for (int k = 0; k < 6; k++) a.trace[(size_t)a.RB * P * 6 + (size_t)p * 6 + k] = tph[k];
    FUSED_TRACE(4);
    // ---- write-out and publish: L[rb, p] in place (write-through: other workgroups read it), z from the right-hand
    // side row, D from the diagonal block's owner.  (L11 stays on chip: nothing downstream reads it, and the other
    // workgroups may still be loading the un-factored diagonal tile for their own copy of the factorisation.)
    __syncthreads();
    if (own) {
      for (int i = tid; i < NB * NB; i += FT) {
        const int k = i / NB, r = i - k * NB;
        st_wt(A + (size_t)(c0 + k) * ldA + rowbase + r, Lsave[k][r]);
      }
      if (rb == P && tid < NB) a.zvec[c0 + tid] = Lsave[tid][0];
    } else if (tid < NB) {
      st_wt(a.dvec + c0 + tid, dsave[tid]);
    }
    publish(a.done + (size_t)rb * P + p);
    FUSED_TRACE(5);
  }
}

// Measured and rejected (round 2): a software-pipelined helper (next job's tile and operands prefetched into registers
// while the current MFMAs run out of a second LDS buffer, flags peeked without blocking): 0.448 vs 0.422 ms at n = 1200,
// 0.754 vs 0.564 at n = 1536, 2.47 vs 2.68 at n = 2880 -- the twelve MFMAs of a job (0.5 us) cannot cover a 1.5 us
// load, and the extra flag round trip and barriers per job make the helpers the bottleneck where they were not.  What a
// helper needs is more jobs in flight per CU, not a deeper pipeline inside one workgroup.
__device__ void fused_helper_role(const FusedArgs &a, int h, double *lds) {
  const int nA = a.nA, P = a.P, ldA = 2 * nA + NB;
  double *__restrict__ A = a.A;
  double (*La)[NB] = reinterpret_cast<double (*)[NB]>(lds);                 // [48 k][48 rows]  L[rb, q]
  double (*Lb)[NB] = reinterpret_cast<double (*)[NB]>(lds + NB * NB);       //                  L[j, q]
  double *dq = lds + 2 * NB * NB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rt = wv / 3, ct = wv % 3;
  // Column j >= 2 has exactly P tiles with far updates: matrix blocks j..P-1, the right-hand side tile, identity blocks
  // 0..j-2.  Tile index = (j-2) P + idx; this workgroup owns the indices congruent to h modulo NH.
  for (int q = 0; q + 2 < P; q++) {
    const int cq = NB * q;
    for (int j = q + 2; j < P; j++) {
      const int first = (j - 2) * P;
      int idx = ((h - first) % a.NH + a.NH) % a.NH;          // smallest idx >= 0 with (first + idx) % NH == h
      for (; idx < P; idx += a.NH) {
        int rb;
        if (idx < P - j) rb = j + idx;                       // matrix block
        else if (idx == P - j) rb = P;                       // right-hand side tile
        else { rb = P + 1 + (idx - (P - j) - 1); if (rb - P - 1 > q) continue; }     // identity block t: live from panel t on
        const int rowbase = rb < P ? NB * rb : (rb == P ? nA : nA + NB + NB * (rb - P - 1));
        const int cj = NB * j;
        if (tid == 0) {      // a flag read is a ~1 us round trip to memory: the three of a job are polled together
          for (;;) {
            const int f0 = flag_peek(a.done + (size_t)rb * P + q), f1 = flag_peek(a.done + (size_t)j * P + q),
                      f2 = flag_peek(a.done + (size_t)q * P + q);
            if (f0 & f1 & f2) break;
            __builtin_amdgcn_s_sleep(1);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        d4 acc;
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] = A[(size_t)(cj + ct * 16 + l4 + 4 * e) * ldA + rowbase + rt * 16 + l15];
        for (int i = tid; i < NB * NB; i += FT) {
          const int k = i / NB, r = i - k * NB;
          La[k][r] = A[(size_t)(cq + k) * ldA + rowbase + r];
          Lb[k][r] = A[(size_t)(cq + k) * ldA + cj + r];
        }
        if (tid < NB) dq[tid] = a.dvec[cq + tid];
        __syncthreads();
#pragma unroll
        for (int kc = 0; kc < NB / 4; kc++) {
          const int k = 4 * kc + l4;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lb[k][ct * 16 + l15], -(La[k][rt * 16 + l15] * dq[k]), acc, 0, 0, 0);
        }
        if (q == j - 2) {                                    // the tile's last far update: hand it to the panel workgroups
#pragma unroll
          for (int e = 0; e < 4; e++) st_wt(A + (size_t)(cj + ct * 16 + l4 + 4 * e) * ldA + rowbase + rt * 16 + l15, acc[e]);
          publish(a.far + (size_t)rb * P + j);
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) A[(size_t)(cj + ct * 16 + l4 + 4 * e) * ldA + rowbase + rt * 16 + l15] = acc[e];
          __syncthreads();                                   // La / Lb are reloaded for the next tile
        }
      }
    }
  }
}

__global__ __launch_bounds__(FT) void k_ldl_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double fused_lds[];
  if ((int)blockIdx.x < a.RB) fused_panel_role(a, blockIdx.x, fused_lds);
  else fused_helper_role(a, blockIdx.x - a.RB, fused_lds);
}

// ------------------------------------------------------------------------------------------------
// backward solve  L^T x = z  as one product with the factor's own by-product: the identity rows appended to the
// matrix came out of the factorisation as M = L^-T D^+ (row r, columns c >= r), so x = M (D z).
// 64 rows per workgroup, the column range split over its four waves; lanes of a wave read 512 contiguous bytes.
// ------------------------------------------------------------------------------------------------
constexpr int APPLY_CHUNKS = 16;        // column chunks of the product; k_ldl_finish adds the partials in order
__global__ __launch_bounds__(256) void k_ldl_apply(const double *__restrict__ A, int nA, const double *__restrict__ dvec,
                                                   const double *__restrict__ z, double *__restrict__ xpart) {
  __shared__ double sq[256];
  const int ldA = 2 * nA + NB;
  const int rl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64, r = r0 + rl;
  // this workgroup: rows r0..r0+63 x column chunk blockIdx.y (multiples of 16 columns; only columns >= r0 matter)
  const int span = ((nA - r0) / 16 + APPLY_CHUNKS - 1) / APPLY_CHUNKS * 16;
  const int cbeg = r0 + blockIdx.y * span, cend = min(nA, cbeg + span);
  double acc0 = 0.0, acc1 = 0.0;
  const double *Mr = A + (size_t)(nA + NB) + min(r, nA - 1);
  for (int cb = cbeg + 16 * q; cb < cend; cb += 64) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const int c = cb + k;
      acc0 = __builtin_fma(Mr[(size_t)c * ldA], dvec[c] * z[c], acc0);
      acc1 = __builtin_fma(Mr[(size_t)(c + 1) * ldA], dvec[c + 1] * z[c + 1], acc1);
    }
  }
  sq[threadIdx.x] = acc0 + acc1;
  __syncthreads();
  if (q == 0 && r < nA) xpart[(size_t)blockIdx.y * nA + r] = (sq[rl] + sq[64 + rl]) + (sq[128 + rl] + sq[192 + rl]);
}

// un-permute, q1 = 0.5 dx.(u D dx - g)    (bavoxel.hpp:1127)
// The last solve of this context gave up on a wait (poll limit: its workgroups were not all resident -- somebody else held CUs):
// k_ldl_finish has poisoned dx.  Host-synchronous; called only after a non-finite step was seen.
bool solve_timed_out(balm_ctx *c) {
  const int P = c->nA / NB;
  int flag = 0;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
  if (hipMemcpy(&flag, c->d_flags + (size_t)2 * (2 * P + 1) * P + P, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
  return flag != 0;
}

// ------------------------------------------------------------------------------------------------
// pose update: left  R <- Exp(dth) R, p <- Exp(dth) p + dt   (bavoxel.hpp:1123-1125)
//              right R <- R Exp(dth), p <- p + dt            (bavoxel.hpp:1119-1120)
// Exp = Rodrigues with the reference's 1e-11 threshold (include/tools.hpp:56-71)
// (defined ahead of k_ldl_finish, which applies it to the step it has just assembled: one launch less per LM iteration)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void exp_so3(const double w[3], double E[3][3]) {
  const double nn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  E[0][0] = E[1][1] = E[2][2] = 1.0;
  E[0][1] = E[0][2] = E[1][0] = E[1][2] = E[2][0] = E[2][1] = 0.0;
  if (nn >= 1e-11) {
    const double x = w[0] / nn, y = w[1] / nn, z = w[2] / nn;
    const double K[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
    const double s = sin(nn), c1 = 1.0 - cos(nn);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        // evaluation order of the reference's `I33 + sin*K + (1-cos)*K*K`: ((1-cos)*K)*K
        const double kk = (c1 * K[r][0]) * K[0][c] + (c1 * K[r][1]) * K[1][c] + (c1 * K[r][2]) * K[2][c];
        E[r][c] = (E[r][c] + s * K[r][c]) + kk;
      }
  }
}

__device__ __forceinline__ void update_pose(int form, int j, const double *__restrict__ poses, const double *__restrict__ dx,
                                            double *__restrict__ out) {
  const double *q = poses + 12 * j;
  const double w[3] = {dx[6 * j], dx[6 * j + 1], dx[6 * j + 2]};
  const double dt[3] = {dx[6 * j + 3], dx[6 * j + 4], dx[6 * j + 5]};
  double E[3][3];
  exp_so3(w, E);
  double R[3][3], p[3] = {q[9], q[10], q[11]};
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R[r][c] = q[3 * c + r];
  double Rn[3][3], pn[3];
  if (form == 0) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) Rn[r][c] = E[r][0] * R[0][c] + E[r][1] * R[1][c] + E[r][2] * R[2][c];
      pn[r] = E[r][0] * p[0] + E[r][1] * p[1] + E[r][2] * p[2] + dt[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++) Rn[r][c] = R[r][0] * E[0][c] + R[r][1] * E[1][c] + R[r][2] * E[2][c];
      pn[r] = p[r] + dt[r];
    }
  }
  double *o = out + 12 * j;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) o[3 * c + r] = Rn[r][c];
  o[9] = pn[0]; o[10] = pn[1]; o[11] = pn[2];
}


__global__ __launch_bounds__(1024) void k_ldl_finish(const double *__restrict__ x, int nA, int n,
                                                     const int *__restrict__ perm, const double *__restrict__ H,
                                                     const double *__restrict__ g, const double *__restrict__ pu, double u_arg,
                                                     double *__restrict__ dx, double *__restrict__ scal, const int *__restrict__ abort_flag,
                                                     int upd_form, int W, const double *__restrict__ poses, double *__restrict__ poses_out,
                                                     int nchunks) {
  const double u = pu ? *pu : u_arg;
  // a persistent factorisation that gave up on a flag (k_ldl_chain's bounded waits) must not pass for a solution
  const double poison = *abort_flag ? __longlong_as_double(0x7ff8000000000000ll) : 0.0;
  __shared__ double red[1024];
  const int tid = threadIdx.x;
  double q = 0.0;
  for (int r = tid; r < nA; r += 1024) {
    const int p = perm[r];
    if (p < n) {
      double xv = 0.0;
      if (nchunks == APPLY_CHUNKS) {                                             // partial products of k_ldl_apply: all loads in flight
#pragma unroll
        for (int k = 0; k < APPLY_CHUNKS; k++) xv += x[(size_t)k * nA + r];
      } else {
        xv = x[r];                                                               // k_ldl_backsolve's x
      }
      xv += poison;
      dx[p] = xv;
      q += xv * (u * H[(size_t)p * n + p] * xv - g[p]);
    }
  }
  red[tid] = q;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) scal[2] = 0.5 * red[0];
  if (poses_out) {                       // the trial poses of the LM loop (every dx[] of this workgroup is written: barriers above)
    __threadfence_block();
    for (int j = tid; j < W; j += 1024) update_pose(upd_form, j, poses, dx, poses_out);
  }
}

// How the persistent kernels (k_ldl_fused, k_ldl_chain) are launched: PLAINLY.  Neither uses a grid-wide barrier: they only need every
// workgroup RESIDENT, which the grid size guarantees by construction (<= hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs, on a
// stream whose earlier kernels have drained), and every wait in them is bounded (-> abort flag -> BALM_ERR_NUMERIC, never a hang).
// hipLaunchCooperativeKernel adds the runtime's own guarantee at a price that round 4 measured (profiles/r04b_solve_coop.txt,
// r04b_dist_overhead.txt): the launch goes through the device's cooperative queue with barrier packets on both sides --
//   * 22-24 us per solve at every size (n = 240: 0.082 -> 0.060 ms, n = 1200: 0.277 -> 0.253, n = 3000: 0.893 -> 0.866);
//   * in a process that also holds an RCCL communicator on the same stream (one rank per GPU) the cross-queue barriers multiply: EVERY
//     kernel of the step runs late (k_rank_diag 51 instead of 9 us, assemble 0.19 instead of 0.06 ms, the SYRK +6 %): 0.7 ms per LM
//     step, round 3's unexplained "communicator tax".  With plain launches it is gone (4.29 vs 4.26 ms/step at config 2);
//   * issued from a thread other than the process's first it leaves ROCm 7.2 in a state that segfaults at exit,
//     which is why the device threads of balm_create_multi had no persistent solve in round 3.
// Round 5: the cooperative form is gone from the library (its A/B: profiles/r04b_solve_coop.txt).  What it used to guarantee --
// co-residency when somebody else holds CUs -- is covered in two ways: (1) the grid is sized for the device's slots DIVIDED by the
// contexts of this process alive on that device (live_contexts_on: two ordinary contexts driven from two threads each get half, as
// loopback shards always did), and (2) a wait that does time out (another PROCESS on the GPU, a foreign stream's long kernel) raises
// the abort flag, and the host retries that solve ONCE on the launch path and stays there (balm_capi.hip: solve_timed_out /
// persistent_off) instead of reporting BALM_ERR_NUMERIC.
static std::atomic<int> g_live_contexts[64];
void context_born(int device) { if (device >= 0 && device < 64) g_live_contexts[device].fetch_add(1); }
void context_gone(int device) { if (device >= 0 && device < 64) g_live_contexts[device].fetch_sub(1); }
static int live_contexts_on(int device) {
  const int v = (device >= 0 && device < 64) ? g_live_contexts[device].load() : 1;
  return v < 1 ? 1 : v;
}
// (The shards of a loopback multi-device context are contexts of one device like any others: n replicated solves at the same time.)
static int persistent_slots(const balm_ctx *c, int cap) {
  return cap > 0 ? cap / live_contexts_on(c->device) : cap;
}
template <class Args>
static hipError_t persistent_launch(const balm_ctx *c, void (*kernel)(Args), dim3 grid, dim3 block, Args &a, size_t lds, hipStream_t s) {
  (void)c;
  hipLaunchKernelGGL(kernel, grid, block, lds, s, a);
  return hipGetLastError();
}

// BALM_SOLVE (A/B runs, tests/test_gpu_solve.py): "launches" / "fused" / "chain" / "chainb" / "small" force one path; read per call
static const char *solve_mode() { return getenv("BALM_SOLVE"); }

#include "kernels_chain.inc"

// Co-resident workgroups of k_ldl_fused on the current device (0 = the cooperative launch is not available).
static int fused_capacity(size_t lds_bytes) {
  int dev = 0, coop = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipFuncSetAttribute((const void *)k_ldl_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ldl_fused, FT, lds_bytes) != hipSuccess) return 0;
  return per_cu * cus;
}

// The factorisation proper: one persistent cooperative launch (k_ldl_fused) where it is the faster one (18..40 panels,
// i.e. windows of ~140..320 poses: profiles/r02k_solve_paths_by_window.txt), launches per panel with lookahead otherwise
// (and on a device that refuses the cooperative launch).  BALM_SOLVE=launches / fused forces one of them (A/B runs, tests).
constexpr int FUSED_MAX_P = 40;              // the persistent kernel wins for 18 <= P <= 40 panels (profiles/r02k_solve_paths_by_window.txt)

// Which persistent kernel: BALM_SOLVE=chain forces k_ldl_chain (round 3), =fused k_ldl_fused (round 2).  Default: k_ldl_chain
// wherever a persistent kernel is the choice at all -- it beats k_ldl_fused at every size (profiles/r03d_solve_paths_by_window.txt:
// n = 1200: 0.283 vs 0.428 ms) and the launch path from 5 to 40 panels (n = 240: 0.086 vs 0.089, n = 1920: 0.73 vs 0.84).
constexpr int CHAIN_MIN_P = 5, CHAIN_MAX_P = 40;      // (from 31 panels on the default is its form without identity rows: below)
// ... and above that, up to 100 panels (n = 4800), k_ldl_chain on [A ; rhs] alone followed by the block back-substitution
// k_ldl_backsolve (kernels_chain.inc): the identity rows that yield L^-T D^+ are 70 % of the far updates at P = 63.  Not when the
// caller needs that inverse (balm_pose_covariance: c->need_minv).  BALM_SOLVE=chainb forces it from CHAIN_MIN_P panels on.
constexpr int CHAINB_MIN_P = 31, CHAINB_MAX_P = 100;      // n = 1488 .. 4800 (profiles/r03z_solve_paths_by_window.txt: n = 3600 1.44 vs 2.11 ms on the launch path, n = 4800 3.00 vs 4.10)
static bool solve_wants_backsub(const balm_ctx *c) {
  const int P = c->nA / NB;
  const char *mode = solve_mode();
  if (c->need_minv || c->chain_cap == 0 || c->chain_refused_P[0] == P || c->persistent_off) return false;
  if (P * live_contexts_on(c->device) > (c->chain_cap > 0 ? c->chain_cap : 256)) return false;   // k_ldl_backsolve's P workgroups per context of the device, all resident
  if (mode && !strcmp(mode, "chainb")) return P >= CHAIN_MIN_P && P <= 100;
  if (mode) return false;                              // launches / fused / chain: the other paths, as asked
  return P >= CHAINB_MIN_P && P <= CHAINB_MAX_P;
}
static bool solve_wants_chain(const balm_ctx *c, const char *mode) {
  if (mode && !strcmp(mode, "chain")) return true;
  if (mode && !strcmp(mode, "fused")) return false;
  return c->chain_cap != 0;
}

bool solve_is_persistent(const balm_ctx *c) {
  const int P = c->nA / NB;
  const char *mode = solve_mode();
  const bool forced = mode && (!strcmp(mode, "fused") || !strcmp(mode, "chain"));
  if (c->persistent_off) return false;                  // a wait of a persistent kernel timed out on this context once: launch path from then on
  if (mode && !strcmp(mode, "launches")) return false;
  if (solve_wants_backsub(c)) return true;
  if (forced) return P >= 2 && (c->fused_cap != 0 || c->chain_cap != 0);
  return (P >= CHAIN_MIN_P && P <= CHAIN_MAX_P && c->chain_cap != 0 && c->chain_refused_P[1] != P) || (P >= 18 && P <= FUSED_MAX_P && c->fused_cap != 0);
}

static void launch_build_A(balm_ctx *c) {
  const int n = c->n, nA = c->nA, P = nA / NB;
  const double *pu = c->u_on_device ? c->d_scal + SCAL_U : nullptr;
  const int tiled = c->solve_tiled ? 1 : 0;
  const int nflags = 2 * (2 * P + 1) * P + P + 8;
  const size_t lds = (size_t)n * sizeof(double);
#define BALM_BUILD_A(M) hipLaunchKernelGGL(k_build_A<M>, dim3(nA), dim3(256), lds, c->stream, c->d_H, c->d_g, n, nA, c->d_perm, pu, c->u_value, \
                                           c->d_A, c->d_flags, nflags, c->d_x + nA, c->d_minv + (size_t)P * NB * NB, tiled)
  if (nA <= 256 * 5) BALM_BUILD_A(5);             // n <= 1280: the bench window (n = 1200), the shipped one (1062)
  else if (nA <= 256 * 10) BALM_BUILD_A(10);
  else if (nA <= 256 * 16) BALM_BUILD_A(16);
  else BALM_BUILD_A(25);                          // nA <= 6400: every window balm_create takes (W <= 1024: nA = 6144)
#undef BALM_BUILD_A
}

static void launch_factor(balm_ctx *c) {
  hipStream_t s = c->stream;
  const int nA = c->nA, P = nA / NB;
  // The layout of d_A was decided ONCE for this solve, by launch_solve (c->solve_tiled = solve_wants_backsub at that moment).  The
  // answer depends on the process-wide count of live contexts on the device: a context created or destroyed by another thread between
  // the two evaluations must not send a tile-major matrix down a path that reads it column-major.
  const bool backsub = c->solve_tiled;
  bool want_fused = backsub || solve_is_persistent(c);
  c->solve_backsub = false;
  static const bool dbg = getenv("BALM_SOLVE_DEBUG") != nullptr;      // one line per factorisation: which path, and why not another
  if (dbg) fprintf(stderr, "balm_hip: solve P=%d persistent=%d backsub=%d chain_cap=%d fused_cap=%d multi=%d\n", P, (int)want_fused,
                   (int)backsub, c->chain_cap, c->fused_cap, c->multi ? c->multi->n : 0);
  if (backsub) {
    if (launch_factor_chain(c, /*ident=*/false, c->solve_tiled)) { c->solve_backsub = true; return; }
    if (P > FUSED_MAX_P) want_fused = false;             // (refused: such a window is the launch path's, not k_ldl_fused's)
    if (c->solve_tiled) { c->solve_tiled = false; launch_build_A(c); }      // ... and the other paths read the column-major matrix
  } else {
    if (want_fused && solve_wants_chain(c, solve_mode()) && launch_factor_chain(c, true, false)) return;
  }
  if (want_fused) {
    const size_t lds = (size_t)(12 * FLR + 2 * NB * NB + NB) * sizeof(double);
    if (c->fused_cap < 0) c->fused_cap = fused_capacity(lds);
    const int RB = 2 * P + 1;
    int NH = persistent_slots(c, c->fused_cap) - RB;
    const int most = (P - 2) * P;                        // tiles with far updates: more helpers than tiles idle
    if (NH > most) NH = most;
    if (P == 2) NH = 0;
    if (NH >= (P > 2 ? 1 : 0)) {
      FusedArgs fa{c->d_A, c->d_dvec, c->d_z, c->d_flags, c->d_flags + (size_t)RB * P, nA, P, RB, NH > 0 ? NH : 1, c->d_trace};
      if (persistent_launch(c, k_ldl_fused, dim3(RB + (P > 2 ? NH : 0)), dim3(FT), fa, lds, s) == hipSuccess)
        return;
      hipGetLastError();
      c->fused_cap = 0;                                  // not again on this context
    }
  }
  // Launch path (windows outside the persistent kernel's range), with LOOKAHEAD: the trailing update of panel p is split by
  // column -- A(p) = the next panel's own 48 columns (a small launch right behind the panel), B(p) = everything to the right
  // of them, which rides in the SAME launch as panel p+1 (k_ldl_panel_trail).  Two launches per panel as before, but the
  // bulk of the trailing update no longer stands in front of the next panel's twelve latency-bound steps.
  // (Also measured: A(p) folded into panel p+1's workgroups as an in-register update, one launch per panel: 1.61 instead
  // of 1.39 ms at n = 2880 -- every panel workgroup then repeats the diagonal block's update, 72 MFMAs on the critical path.)
  // What the shared launch costs: the panel's workgroups run 17 us instead of 11.6 at n = 2880 when trailing-tile waves
  // sit on their CUs.  Keeping them apart was tried three ways: the whole CU's LDS requested
  // per workgroup (one workgroup per CU: 0.89 instead of 0.94 ms at n = 2100, but 1.47 instead of 1.39 at n = 2880 -- the
  // trailing tiles starve); trailing tiles as workers that leave CUs marked busy by a panel workgroup (__smid) and pull
  // tiles from a counter (one counter: 12 ns per fetch serialised, 5.9 ms; one per XCD with eight tiles per fetch: 2.6 ms).
  // (Measured and rejected first: the same split on two streams with events -- bit-identical, but every cross-stream
  // dependency costs ~4 us on this runtime: 1.94 instead of 1.74 ms at n = 2880.)
  const char *la = getenv("BALM_LOOKAHEAD");              // A/B: 0 / 1 force
  const bool lookahead = la ? !strcmp(la, "1") : true;
  const size_t wp_stride = (size_t)NB * (2 * nA + NB);
  for (int p = 0; p < P; p++) {
    const int c0 = p * NB;
    const int m = nA - c0 - NB;                 // square part still to factor
    const int nR = nA + NB + NB * (p + 1);      // live rows: + right-hand side tile + identity tiles 0..p
    const int rows = nR - (c0 + NB);
    const int npb = (rows + PANEL_ROWS - 1) / PANEL_ROWS;
    const int mt = m / NB;
    double *Wp = c->d_Wp + (lookahead ? (size_t)(p & 1) * wp_stride : 0);
    const int mt_prev = mt + 1;                 // panel p-1's trailing square in 48-blocks
    if (lookahead && p > 0 && mt_prev > 1) {
      const int gx = (3 * (mt_prev - 1) + 3) / 4, gy = mt_prev + 1 + p;
      hipLaunchKernelGGL(k_ldl_panel_trail, dim3(npb + gx * gy), dim3(256), 0, s, c->d_A, nA, c0, nR, c->d_dvec, Wp, c->d_z, npb,
                         c0 - NB, c->d_Wp + (size_t)((p - 1) & 1) * wp_stride, mt_prev, gx);
    } else {
      hipLaunchKernelGGL(k_ldl_panel, dim3(npb), dim3(256), 0, s, c->d_A, nA, c0, nR, c->d_dvec, Wp, c->d_z);
    }
    if (m <= 0) continue;
    const int gy = mt + 1 + (p + 1);
    if (lookahead) hipLaunchKernelGGL(k_ldl_trail, dim3(1, gy), dim3(256), 0, s, c->d_A, nA, c0, Wp, mt, 0, 3);
    else hipLaunchKernelGGL(k_ldl_trail, dim3((3 * mt + 3) / 4, gy), dim3(256), 0, s, c->d_A, nA, c0, Wp, mt, 0, 3 * mt);
  }
}

#include "kernels_small.inc"

void launch_solve(balm_ctx *c, bool new_hessian, int upd_form, const double *upd_poses, double *upd_out) {      // damping u: c->u_value as a kernel argument, or (c->u_on_device: graph
  hipStream_t s = c->stream;                            // capture / replay) c->d_scal[SCAL_U], put there on the stream by push_damping
  const int n = c->n, nA = c->nA;
  if (solve_wants_small(c)) {                           // windows of up to 24 poses: the whole solve is one launch of one workgroup
    c->solve_backsub = false; c->solve_tiled = false;
    if (launch_solve_small(c, upd_form, upd_poses, upd_out)) return;
  }
  const double *pu = c->u_on_device ? c->d_scal + SCAL_U : nullptr;
  if (new_hessian) {
    const size_t lds = (size_t)nA * sizeof(double) + 256 * sizeof(int);
    const dim3 grid((nA + 15) / 16);
    if (nA <= 256 * 5) hipLaunchKernelGGL(k_rank_diag<5>, grid, dim3(256), lds, s, c->d_H, n, nA, c->d_perm);
    else if (nA <= 256 * 10) hipLaunchKernelGGL(k_rank_diag<10>, grid, dim3(256), lds, s, c->d_H, n, nA, c->d_perm);
    else if (nA <= 256 * 16) hipLaunchKernelGGL(k_rank_diag<16>, grid, dim3(256), lds, s, c->d_H, n, nA, c->d_perm);
    else hipLaunchKernelGGL(k_rank_diag<25>, grid, dim3(256), lds, s, c->d_H, n, nA, c->d_perm);
  }
  // [A ; rhs] tile by tile for k_ldl_chain + k_ldl_backsolve (no identity rows, nobody else reads the matrix)
  c->solve_tiled = solve_wants_backsub(c);
  launch_build_A(c);
  if (c->inject_solve_timeout) {       // tests (BALM_FAULT_INJECT="timeout,<iteration>"): the abort flag as a timed-out wait leaves it
    const int P = nA / NB;
    hipMemsetAsync(c->d_flags + (size_t)2 * (2 * P + 1) * P + P, 1, sizeof(int), s);
    c->inject_solve_timeout = false;
  }
  launch_factor(c);
  if (c->solve_backsub) {           // x (permuted order) -> chunk 0 of d_x; chunk 1 is the workgroups' exchange buffer
    const int P = nA / NB;
    hipLaunchKernelGGL(k_ldl_backsolve, dim3(P), dim3(256), 0, s, c->d_A, nA, P, c->solve_tiled ? 1 : 0, c->d_minv, c->d_dvec, c->d_z, c->d_x + nA, c->d_x,
                       c->d_flags + (size_t)2 * (2 * P + 1) * P + P);
  } else {
    hipLaunchKernelGGL(k_ldl_apply, dim3((nA + 63) / 64, APPLY_CHUNKS), dim3(256), 0, s, c->d_A, nA, c->d_dvec, c->d_z, c->d_x);
  }
  {
    const int P = nA / NB;
    int *abortf = c->d_flags + (size_t)2 * (2 * P + 1) * P + P;
    const int nch = c->solve_backsub ? 1 : APPLY_CHUNKS;
    hipLaunchKernelGGL(k_ldl_finish, dim3(1), dim3(1024), 0, s, c->d_x, nA, n, c->d_perm, c->d_H, c->d_g, pu, c->u_value, c->d_dx,
                       c->d_scal, abortf, upd_form, c->W, upd_poses, upd_out, nch);
  }
}

// ------------------------------------------------------------------------------------------------
// pose update: left  R <- Exp(dth) R, p <- Exp(dth) p + dt   (bavoxel.hpp:1123-1125)
//              right R <- R Exp(dth), p <- p + dt            (bavoxel.hpp:1119-1120)
// Exp = Rodrigues with the reference's 1e-11 threshold (include/tools.hpp:56-71)
// ------------------------------------------------------------------------------------------------
__global__ void k_update_poses(int form, int W, const double *__restrict__ poses, const double *__restrict__ dx,
                               double *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < W) update_pose(form, j, poses, dx, out);
}

void launch_update_poses(hipStream_t s, int form, int W, const double *poses, const double *dx, double *out) {
  hipLaunchKernelGGL(k_update_poses, dim3((W + 127) / 128), dim3(128), 0, s, form, W, poses, dx, out);
}

// bavoxel.hpp:1159-1164: p_j <- R_0^T (p_j - p_0), R_j <- R_0^T R_j  (pose 0 included)
__global__ __launch_bounds__(256) void k_reanchor(int W, double *__restrict__ poses) {
  __shared__ double e0[12];
  if (threadIdx.x < 12) e0[threadIdx.x] = poses[threadIdx.x];
  __syncthreads();
  for (int j = threadIdx.x; j < W; j += blockDim.x) {
    double *q = poses + 12 * j;
    double R[3][3], p[3] = {q[9] - e0[9], q[10] - e0[10], q[11] - e0[11]};
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 3; r++) R[r][c] = q[3 * c + r];
    // R0^T(r,k) = R0(k,r) = e0[3*r + k]
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
      for (int c = 0; c < 3; c++)
        q[3 * c + r] = e0[3 * r] * R[0][c] + e0[3 * r + 1] * R[1][c] + e0[3 * r + 2] * R[2][c];
      q[9 + r] = e0[3 * r] * p[0] + e0[3 * r + 1] * p[1] + e0[3 * r + 2] * p[2];
    }
  }
}

void launch_reanchor(hipStream_t s, int W, double *poses) {
  hipLaunchKernelGGL(k_reanchor, dim3(1), dim3(256), 0, s, W, poses);
}

// The 16 scalars of an LM iteration (residuals, q1, ...) straight into the pinned host mirror, the stamp behind them: the host
// polls the stamp instead of paying a copy command and a hipStreamSynchronize per iteration (balm_capi.hip: wait_scalars)
// rpart != NULL: scal[slot] = sum of the nr residual partials first (k_sum_scalar's fixed order), in the same launch
__global__ __launch_bounds__(256) void k_scalars_mail(double *__restrict__ scal, volatile double *__restrict__ host, double stamp,
                                                      const double *__restrict__ rpart, int nr, int slot) {
  __shared__ double sred[256];
  if (rpart) {
    double s = 0.0;
    for (int t = threadIdx.x; t < nr; t += 256) s += rpart[t];
    sred[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if (threadIdx.x < k) sred[threadIdx.x] += sred[threadIdx.x + k];
      __syncthreads();
    }
    if (threadIdx.x == 0) scal[slot] = sred[0];
    __syncthreads();
  }
  if (threadIdx.x < 16) host[threadIdx.x] = (rpart && (int)threadIdx.x == slot) ? sred[0] : scal[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) { host[SCAL_STAMP] = stamp; __threadfence_system(); }
}

void launch_scalars_mail(hipStream_t s, double *d_scal, double *d_hscal, double stamp, const double *rpart, int nr, int slot) {
  hipLaunchKernelGGL(k_scalars_mail, dim3(1), dim3(256), 0, s, d_scal, d_hscal, stamp, rpart, nr, slot);
}

// the code object of this translation unit, loaded on the current device now (the runtime loads it on the first use of any of its
// kernels otherwise: balm_prewarm does it on a background thread while the caller is still busy elsewhere)
hipError_t preload_solve() {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, (const void *)k_ldl_apply);
}

}  // namespace balm
