// K3 on the INT8 matrix cores, FP64-exact to the parity tests' tolerance (round 6; OPT-IN: BALM_SYRK=int8, default off --
// the shipped default, the bench line's dtype and every tolerance stay FP64).
//
// H -= Gt Gt^T  (bavoxel.hpp:404-418 as the rank-3 SYRK of DESIGN.md 2) by error-free slicing: every row i of Gt gets ONE exponent e_i
// (|Gt[i][k]| < 2^(e_i - 1) for all k) and every entry four signed digits of radix 254,
//     Gt[i][k] = 2^e_i * sum_{a < 4} d_a[i][k] 254^-(a+1)  (+ a remainder below 2^(e_i - 32)),   |d_a| <= 127,
// so that a digit-by-digit product  sum_k d_a[i][k] d_b[j][k]  is EXACT in int32 (127^2 * 4 pairs * 32 704 columns per k-slice < 2^31) on
// v_mfma_i32_16x16x64_i8.  Which products a 1e-10 Hessian needs was settled on the CPU (tools/study_int8_syrk.py, profiles/
// r06_int8_syrk_study.txt): the pairs a + b <= 3 AND (2, 2) -- a dropped diagonal pair is a sum of squares and adds up coherently over
// the 150 000 columns -- eleven ordered pairs; measured on the GPU: 1.4e-12 of the full-size test's scale.  Pairs with equal a + b share an
// accumulator (same weight 254^-(a+b+2)): five int32 accumulator sets.
// (Tried: the (2, 2) pair only on the diagonal -- per-row sums of squares of the third digits from the slicing kernel, ten MFMA pairs and four
// sets: k_syrk_i8 -7 %, but rows of neighbouring poses are nearly equal and their third digits correlate: off-diagonal error 1.4e-12 -> 6.9e-12.)
//
//   (row maxima)  left by the factor kernels where a lane owns a pose (kernels_accum.hip MAXR), else k_i8_rowmax: a pass over Gt
//   k_i8_slice    digits, transposed from Gt's [k][row] to [digit][16 rows][64 columns] pieces of 1 KB laid out as the MFMA operand of a wavefront
//   k_syrk_i8     one workgroup = one 128 x 128 output tile (I <= J) on one XCD's share of the columns; eight wavefronts, 32 x 64 each,
//                 all eleven digit pairs on one pass over the operands: per 64-column step 64 KB go from memory straight into LDS
//                 (global_load_lds, the 1 KB pieces as they lie), 24 fragment reads and 88 MFMAs per wavefront
//   k_i8_pack     sum over the k-slices (each slice's five sets leave k_syrk_i8 weighted and added: one double) x the rows' scales -> ONE split-K
//                 slice in k_hessian_syrk's tile layout: k_reduce_all / k_assemble behind it are the FP64 path's
#include <hip/hip_runtime.h>

#include "balm_internal.h"

namespace balm {

namespace {

constexpr int I8_DIGITS = 4, I8_SETS = 5, I8_TILE = 128, I8_KS = 64, I8_XCDS = 8;
// The digits' radix: 254, not 256 -- round-to-nearest digits of radix 256 reach +128, which an int8 does not hold (and a digit set [-128, 127]
// leaves the greedy expansion exactly one admissible rounding offset, 127/255: one ulp decides), 254 keeps |digit| <= 127 by construction.  A
// radix that is not a power of two costs the slicing one rounding (2^-53 of the entry) per digit: the products stay exact integers, the
// weights radix^-(a+b+2) are applied once, in FP64.  (Radix 128, |digit| <= 64, was the first build: 2.4e-11 of the full-size test's scale
// instead of 1.1e-12.)
constexpr double I8_RADIX = 254.0;
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_i8_rowmax(const double *__restrict__ Gt, int npad, int rows, long K, int kchunk,
                                                   unsigned long long *__restrict__ rowmax) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const long k0 = (long)blockIdx.y * kchunk, k1 = min(K, k0 + kchunk);
  double m = 0.0, bad = 0.0;                       // (bad: NaN from the first non-finite entry on -- fmax drops a NaN; k_feature_factors' track_rowmax)
  long k = k0;
  for (; k + 8 <= k1; k += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = Gt[(size_t)(k + u) * npad + row];
#pragma unroll
    for (int u = 0; u < 8; u++) { m = fmax(m, fabs(v[u])); bad = fma(v[u], 0.0, bad); }
  }
  for (; k < k1; k++) { const double v = Gt[(size_t)k * npad + row]; m = fmax(m, fabs(v)); bad = fma(v, 0.0, bad); }
  if (bad != bad) m = __longlong_as_double(0x7ff8000000000000ll);
  if (m > 0.0 || m != m) atomicMax(rowmax + row, (unsigned long long)__double_as_longlong(m));      // (non-negative doubles order like their bits, NaN above all)
}

// 64 rows x 64 columns per workgroup: coalesced loads along the rows, digits packed four columns to a word, out as 16-byte row pieces
__global__ __launch_bounds__(256) void k_i8_slice(const double *__restrict__ Gt, int npad, long K, long Kp, int rows_p,
                                                  const unsigned long long *__restrict__ rowmax, signed char *__restrict__ D,
                                                  double *__restrict__ rowscale) {
  __shared__ unsigned int tile[I8_DIGITS][64][17];          // [digit][row][16 words of four columns + 1 pad]
  const int r0 = blockIdx.x * 64;
  const long k0 = (long)blockIdx.y * 64;
  const int rl = threadIdx.x & 63, kg = threadIdx.x >> 6;  // the thread's row, its sixteen columns kg * 16 ..
  const int row = r0 + rl;
  int e = 0;
  bool live = false, nonfinite = false;
  if (row < npad && row < rows_p) {
    const double m = __longlong_as_double((long long)rowmax[row]);
    if (m > 0.0 && isfinite(m)) { e = ilogb(m) + 2; live = true; }       // |x| / 2^e < 0.5: the first digit stays within +-127
    else if (!(m == 0.0)) nonfinite = true;                              // a row with an infinite or NaN entry: NaN scale -> its row and column of H
  }
  if (blockIdx.y == 0 && kg == 0 && row < rows_p) rowscale[row] = live ? ldexp(1.0, e) : (nonfinite ? __longlong_as_double(0x7ff8000000000000ll) : 0.0);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long k = k0 + kg * 16 + 4 * j + u;
      x[u] = (live && k < K) ? Gt[(size_t)k * npad + row] : 0.0;
    }
    unsigned int w[I8_DIGITS] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double r = isfinite(x[u]) ? ldexp(x[u], -e) : 0.0;
#pragma unroll
      for (int a = 0; a < I8_DIGITS; a++) {
        r *= I8_RADIX;
        const double d = rint(r);            // |r| <= 127: |x / 2^e| < 0.5 for the first digit, a remainder |r - d| <= 0.5 for the next
        r -= d;
        w[a] |= ((unsigned int)(int)d & 0xffu) << (8 * u);
      }
    }
#pragma unroll
    for (int a = 0; a < I8_DIGITS; a++) tile[a][rl][kg * 4 + j] = w[a];
  }
  __syncthreads();
  {
    // a wavefront writes one 1 KB operand piece per digit: 16 rows x 64 columns as [column quarter][row][16 bytes], the order in which
    // k_syrk_i8's global_load_lds reads it (64 lanes x 16 consecutive bytes) and the MFMA takes it (lane = quarter * 16 + row)
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, orow = w * 16 + (l & 15);
    const size_t RG = rows_p / 16, KS = Kp / 64;
    if (r0 + orow < rows_p) {
#pragma unroll
      for (int a = 0; a < I8_DIGITS; a++) {
        const uint4 v = make_uint4(tile[a][orow][4 * q], tile[a][orow][4 * q + 1], tile[a][orow][4 * q + 2], tile[a][orow][4 * q + 3]);
        *reinterpret_cast<uint4 *>(D + (((size_t)a * RG + (r0 >> 4) + w) * KS + blockIdx.y) * 1024 + l * 16) = v;
      }
    }
  }
}

// 16 bytes per lane from global memory straight into LDS: the wavefront's 64 lanes fill 1 KB at `lds_wave` in lane order
__device__ __forceinline__ void i8_glds16(const signed char *src, void *lds_wave) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)lds_wave, 16, 0, 0);
}

// Eight wavefronts, two per SIMD, 32 x 64 of the 128 x 128 tile each: 5 sets x 2 x 4 accumulator tiles = 160 registers (all AGPRs -- with
// four wavefronts of 64 x 64 the 320 accumulator registers do not fit the 256 AGPRs and the compiler shuttles the rest through
// v_accvgpr moves, 768 per step: 2.05 ms, measured), the second wavefront of a SIMD covers the other's LDS waits.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_syrk_i8(const signed char *__restrict__ D, long Kp, int rows_p, int T,
                                                                                           int NT, int M, double *__restrict__ P) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];      // two stages of [side 2][digit 4][row group 8][1 KB]: 128 KB
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wr = wv >> 1, wc = wv & 1;      // rows wr * 32 .., columns wc * 64 .. of the tile
  const int xcd = blockIdx.x & 7;                 // (the hardware places workgroup b on XCD b mod 8: an XCD's workgroups share its eighth of the columns)
  const int ksl = xcd * M + (int)((blockIdx.x >> 3) % M);      // the k-slice: M per XCD (1 up to 262 144 columns: the int32 sums' bound)
  int tile = (int)((blockIdx.x >> 3) / M), I = 0;
  { int left = tile; while (left >= T - I) { left -= T - I; I++; } tile = left; }
  const int J = I + tile;
  const int tix = (int)((blockIdx.x >> 3) / M);
  const long Kx = Kp / (I8_XCDS * M);
  const int nsteps = (int)(Kx / I8_KS);
  const bool diag = I == J;
  // this wavefront's eight pieces of a stage: piece p = wv * 8 + i -> (side, digit, row group); the lane's source offset inside a stage
  // (byte offsets from the k-slice's base: the four digit matrices of one window are below 4 GB, checked by the launcher)
  unsigned int src[8];
  const unsigned int RG = rows_p / 16, KS = (unsigned int)(Kp / I8_KS);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int p = wv * 8 + i, side = p >> 5, dg = (p >> 3) & 3, rg = p & 7;
    src[i] = ((dg * RG + (side ? J : I) * 8 + rg) * KS + ksl * nsteps) * 1024u + lane * 16;
  }
  v4i acc[I8_SETS][2][4];
#pragma unroll
  for (int s = 0; s < I8_SETS; s++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[s][i][j] = (v4i){0, 0, 0, 0};
  // two 64 KB stage buffers: stage s + 1 flies from memory into one while the MFMAs read stage s out of the other -- one barrier per step
  auto issue = [&](int step) {
    unsigned char *buf = lds + (size_t)(step & 1) * 65536;
    if (diag && wv >= 4) return;                 // a diagonal tile's B rows are its A rows
#pragma unroll
    for (int i = 0; i < 8; i++) i8_glds16(D + ((size_t)step * 1024 + src[i]), buf + (size_t)(wv * 8 + i) * 1024);
  };
  if (nsteps > 0) issue(0);
  for (int step = 0; step < nsteps; step++) {
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this wavefront's pieces of the stage are in LDS
    __syncthreads();                             // ... and everybody else's; and everybody is done reading the other buffer
    // (the stage in two halves, the second between the MFMA groups: 1.40 ms against 1.10 -- the compiler fences the fragment reads behind it)
    if (step + 1 < nsteps) issue(step + 1);
    const unsigned char *buf = lds + (size_t)(step & 1) * 65536;
    const unsigned char *bufA = buf + (size_t)(wr * 2) * 1024 + lane * 16;
    const unsigned char *bufB = buf + (size_t)((diag ? 0 : 32) + wc * 4) * 1024 + lane * 16;
    v4i A[I8_DIGITS][2];
#pragma unroll
    for (int d = 0; d < I8_DIGITS; d++)
#pragma unroll
      for (int i = 0; i < 2; i++) A[d][i] = *reinterpret_cast<const v4i *>(bufA + (size_t)(d * 8 + i) * 1024);
    // by B digit: (a, b) with a + b <= 3, and (2, 2); the accumulator set is a + b except (2, 2)'s, which is 4.  (The next digit's B fragments read
    // before this digit's MFMAs, held there by a sched_barrier, 248 registers: 1.16 against 1.19 ms on the same box -- not kept.  The whole step
    // pinned by sched_group_barriers -- A's first digit and B's first, then every read one MFMA group ahead of its use: 1.16-1.17, no change: the
    // SIMD's other wavefront already covers these waits.)
#pragma unroll
    for (int b = 0; b < I8_DIGITS; b++) {
      v4i B[4];
#pragma unroll
      for (int j = 0; j < 4; j++) B[j] = *reinterpret_cast<const v4i *>(bufB + (size_t)(b * 8 + j) * 1024);
#pragma unroll
      for (int a = 0; a < I8_DIGITS; a++) {
        if (!(a + b <= 3 || (a == 2 && b == 2))) continue;
        const int set = (a == 2 && b == 2) ? 4 : a + b;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[set][i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a][i], B[j], acc[set][i][j], 0, 0, 0);
      }
    }
  }
  // C/D layout of the 16 x 16 forms: col = lane & 15, row = (lane >> 4) * 4 + reg.  The five exact int32 sums of an element leave as ONE double,
  // sum_s radix^-(s+2) sum_s, smallest weight first (2^-53 of the slice's value: FP64's own rounding) -- 8 bytes per element and slice instead
  // of 20, 16 lanes x 8 bytes = a full cache line per store
  double *out = P + ((size_t)ksl * NT + tix) * (I8_TILE * I8_TILE);
  constexpr double W0 = 1.0 / (I8_RADIX * I8_RADIX), W1 = W0 / I8_RADIX, W2 = W1 / I8_RADIX, W3 = W2 / I8_RADIX, W4 = W3 / I8_RADIX;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ri = wr * 32 + i * 16 + (lane >> 4) * 4 + r, cj = wc * 64 + j * 16 + (lane & 15);
        out[ri * I8_TILE + cj] = (((W4 * (double)acc[4][i][j][r] + W3 * (double)acc[3][i][j][r]) + W2 * (double)acc[2][i][j][r]) + W1 * (double)acc[1][i][j][r]) +
                                 W0 * (double)acc[0][i][j][r];
      }
}

// sum over k-slices -> FP64 in k_hessian_syrk's tile layout (one split-K slice): part[tile * 6400 + (mt * 4 + reg) * 64 + lane]
__global__ __launch_bounds__(256) void k_i8_pack(const double *__restrict__ P, int NT, int T, const double *__restrict__ rowscale, const int *__restrict__ tileIJ,
                                                 int ntiles, int n, int nslices, double *__restrict__ part) {
  const long total = (long)ntiles * TILE_ELEMS;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int tl = (int)(t / TILE_ELEMS);
    const int e = (int)(t - (long)tl * TILE_ELEMS);
    const int lane = e & 63, slot = e >> 6, reg = slot & 3, mt = slot >> 2;
    const int rc = tileIJ[tl * 25 + mt];
    int row = (rc >> 16) * 16 + (lane >> 4) + 4 * reg;          // MFMA f64 16x16x4 C/D layout (k_assemble)
    int col = (rc & 0xffff) * 16 + (lane & 15);
    double val = 0.0;
    if (row < n && col < n) {
      if (row > col) { const int q = row; row = col; col = q; }
      const int I = row / I8_TILE, J = col / I8_TILE;
      const int tix = I * T - I * (I - 1) / 2 + (J - I);
      const int ri = row - I * I8_TILE, cj = col - J * I8_TILE;
      for (int x = 0; x < nslices; x++) val += P[((size_t)x * NT + tix) * (I8_TILE * I8_TILE) + ri * I8_TILE + cj];
      val *= rowscale[row] * rowscale[col];
    }
    part[t] = val;
  }
}

}  // namespace

// scratch of the INT8 product for n rows and K columns (bytes): digits | row maxima | row scales | the k-slices' partial tiles (one double per element)
size_t syrk_i8_scratch_bytes(int n, long K, I8Layout *lay) {
  I8Layout L;
  L.T = (n + I8_TILE - 1) / I8_TILE;
  L.rows_p = L.T * I8_TILE;
  L.NT = L.T * (L.T + 1) / 2;
  // k-slices: eight (one per XCD) times M, M the smallest count that keeps a slice's int32 sums exact: 4 pairs x 128^2 x columns < 2^31
  L.M = 1;
  while ((K + (long)I8_XCDS * L.M * I8_KS - 1) / ((long)I8_XCDS * L.M * I8_KS) * I8_KS > 32704) L.M++;
  // ... and, for windows of few tiles, the count that fills the chip's 256 CUs best: 8 M NT workgroups run in ceil(8 M NT / 256) rounds of 1 / M
  // of the columns each (a 100-pose window: 15 tiles x 8 slices = 120 workgroups leave half of the CUs idle; M = 2: 240), against the extra
  // partial tiles written and read back (8 NT x 128 KB per unit of M).  Priced with the kernel's measured 1.4 us per 64-column step and
  // 5 TB/s for the partials; a slice keeps sixteen steps.  (From 41 tiles on -- 177 poses -- M stays what the int32 sums ask for: at 200 poses, 55 tiles,
  // M = 2 / 4 / 5 measured 1.23 / 1.20 / 2.69 ms against 1.19 for the product and 0.09 / 0.18 / 0.23 against 0.05 for the packing.)
  if (L.NT <= 40) {
    int best = L.M;
    double best_cost = 1e30;
    for (int m = L.M; m <= 8; m++) {
      if (K / ((long)I8_XCDS * m) < 16 * I8_KS) break;
      const double rounds = (double)((I8_XCDS * m * L.NT + 255) / 256), steps = (double)K / (I8_XCDS * m * I8_KS);
      const double cost = rounds * steps * 1.4e-6 + (double)m * L.NT * I8_XCDS * I8_TILE * I8_TILE * 8.0 * 2.0 / 5e12;
      if (cost < best_cost * 0.97) { best_cost = cost; best = m; }      // (3 %: a tie keeps the smaller count)
    }
    L.M = best;
  }
  L.Kp = (K + (long)I8_XCDS * L.M * I8_KS - 1) / ((long)I8_XCDS * L.M * I8_KS) * ((long)I8_XCDS * L.M * I8_KS);
  L.off_digits = 0;
  size_t off = (size_t)I8_DIGITS * L.rows_p * L.Kp;
  off = (off + 255) & ~(size_t)255; L.off_rowmax = off; off += (size_t)L.rows_p * 8;
  off = (off + 255) & ~(size_t)255; L.off_scale = off; off += (size_t)L.rows_p * 8;
  off = (off + 255) & ~(size_t)255; L.off_part = off; off += (size_t)I8_XCDS * L.M * L.NT * I8_TILE * I8_TILE * sizeof(double);
  if (lay) *lay = L;
  return off;
}

hipError_t prepare_device_syrk_i8() {
  return hipFuncSetAttribute((const void *)k_syrk_i8, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

// Gt [K][npad] (column k of the factor matrix = npad contiguous rows, as k_feature_factors writes it) -> part: one split-K slice of Gt Gt^T
// in k_hessian_syrk's tile layout.  The int32 accumulators bound a k-slice at 32 704 columns: 8 x M slices (syrk_i8_scratch_bytes).
// rowmax_known: the rows' largest |entries| where the factor kernel left them (k_feature_factors MAXR), else a pass over Gt finds them.
int launch_syrk_i8(hipStream_t s, const double *Gt, int npad, int n, long K, const int *tileIJ, int ntiles, unsigned char *scratch, double *part,
                   const unsigned long long *rowmax_known) {
  I8Layout L;
  syrk_i8_scratch_bytes(n, K, &L);
  if (K < 1 || (size_t)I8_DIGITS * L.rows_p * L.Kp >= ((size_t)1 << 32)) return -1;      // (k_syrk_i8's 32-bit operand offsets)
  signed char *D = reinterpret_cast<signed char *>(scratch + L.off_digits);
  auto *rowmax = reinterpret_cast<unsigned long long *>(scratch + L.off_rowmax);
  double *rowscale = reinterpret_cast<double *>(scratch + L.off_scale);
  double *P = reinterpret_cast<double *>(scratch + L.off_part);
  const int kchunk = 512;
  if (rowmax_known) rowmax = const_cast<unsigned long long *>(rowmax_known);
  else if (hipMemsetAsync(rowmax, 0, (size_t)L.rows_p * 8, s) != hipSuccess) return -1;
  if (!rowmax_known) hipLaunchKernelGGL(k_i8_rowmax, dim3((npad + 255) / 256, (unsigned int)((K + kchunk - 1) / kchunk)), dim3(256), 0, s, Gt, npad, npad < L.rows_p ? npad : L.rows_p, K, kchunk, rowmax);
  hipLaunchKernelGGL(k_i8_slice, dim3(L.rows_p / 64, (unsigned int)(L.Kp / 64)), dim3(256), 0, s, Gt, npad, K, L.Kp, L.rows_p, rowmax, D, rowscale);
  hipLaunchKernelGGL(k_syrk_i8, dim3(I8_XCDS * L.M * L.NT), dim3(512), 128 * 1024, s, D, L.Kp, L.rows_p, L.T, L.NT, L.M, P);
  long total = (long)ntiles * TILE_ELEMS;
  int grid = (int)((total + 255) / 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_i8_pack, dim3(grid), dim3(256), 0, s, P, L.NT, L.T, rowscale, tileIJ, ntiles, n, I8_XCDS * L.M, part);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace balm
