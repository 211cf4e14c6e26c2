// libmvba.so -- thin SVD of a tall-skinny measurement matrix for the factorization
// initialisation (ref lib/factorization.py:5-15; inline SVD call sites
// lib/affine_camera_calibration.py:19,71,152 with the centring of :224-240).
//
// The reference calls np.linalg.svd(W) with full_matrices=True on W = (2m|3m) x N and
// keeps U[:, :r], diag(sigma[:r]) Vt[:r]; the N x N factor it discards cannot exist at
// N = 5M.  Here W^T ("Wt", N x n, row-major: exactly the array the callers hold) is
// streamed twice:
//   K7a k_gram     G = Wt^T Wt (+ column sums), fp64 accumulation whatever the input dtype
//   K7b k_jacobi   eigen-decomposition of the n x n Gram matrix, one workgroup,
//                  round-robin parallel Jacobi (n/2 independent rotations per step)
//   K7c k_project  S = M^T W, i.e. S[i][row] = sum_c M[c][i] Wt[row][c]
// sigma = sqrt(eig), M = eigenvectors of the r largest.  With fp64 accumulation the Gram
// route loses nothing for fp32 data (eps32 >> eps64 * cond^2 for the leading triplets).
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "mvba_common.h"

using namespace mvba;

namespace {

constexpr int GT = 16;             // Gram tile = one v_mfma_f64_16x16x4_f64 accumulator
constexpr int ROWS_PER_STEP = 32;  // rows consumed per unrolled step (8 MFMA k-groups of 4 rows)
typedef double svd_d4 __attribute__((ext_vector_type(4)));

// G += Wt^T Wt on the f64 matrix cores.  For 4 consecutive rows r0..r3 and column tiles (ti,tj):
//   A[i][k] = Wt[r_k][16 ti + i]   (lane l: i = l & 15, k = l >> 4)
//   B[k][j] = Wt[r_k][16 tj + j]   (lane l: k = l >> 4, j = l & 15)
// so both operands are "this lane's row (l >> 4), this lane's column (l & 15) of a tile": one
// value per lane per tile, converted to fp64 on load.  C/D: col = l & 15, row = (l >> 4) + 4 reg.
// Partial tiles go to partial[chunk][pair][256] (no atomics: millions of f64 adds onto a
// 24 x 24 matrix serialise at the memory side); k_gram_reduce sums the chunks in fixed order.
__device__ __forceinline__ void gram_store_partial(const svd_d4 *acc, int npairs, int pair0, int pairs_total,
                                                   double *__restrict__ partial) {
  __shared__ double red[4][3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < npairs; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][q][r][lane] = acc[q][r];
  __syncthreads();
  double *out = partial + ((size_t)blockIdx.y * pairs_total + pair0) * 256;
  for (int e = threadIdx.x; e < npairs * 256; e += 256) {
    const int q = e >> 8, r = (e >> 6) & 3, l = e & 63;
    out[e] = red[0][q][r][l] + red[1][q][r][l] + red[2][q][r][l] + red[3][q][r][l];
  }
}

// n <= 16 * NT (NT = 1 or 2): one pass over the rows forms every tile pair (1 or 3 MFMAs per 4 rows).
// A workgroup takes GRAM_ROWS consecutive rows per step: the rows are one contiguous byte range of
// Wt, streamed with 16-byte loads by all 256 threads into LDS (a lane-per-element gather of the
// MFMA operands straight from HBM ran at 1.8 TB/s), then every wave picks the operands of its
// 32 rows out of LDS.  Rows past the end are zero-filled in LDS.
constexpr int GRAM_ROWS = 4 * ROWS_PER_STEP;  // 128 rows per workgroup step
template <typename T, int NT>
__global__ __launch_bounds__(256) void k_gram_fused(const T *__restrict__ Wt, long long n_rows, int n,
                                                    double *__restrict__ partial) {
  extern __shared__ double gram_lds_raw[];
  T *stage = reinterpret_cast<T *>(gram_lds_raw);  // GRAM_ROWS x n, row-major like Wt
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  constexpr int NP = NT * (NT + 1) / 2;
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  svd_d4 acc[3] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  int col[NT];
  bool cok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { cok[t] = GT * t + li < n; col[t] = min(GT * t + li, n - 1); }
  const long long total = n_rows * n;
  const int nvec = GRAM_ROWS * n / VEC;  // GRAM_ROWS * n is a multiple of 4
  for (long long r0 = (long long)blockIdx.y * GRAM_ROWS; r0 < n_rows; r0 += (long long)gridDim.y * GRAM_ROWS) {
    const long long e0 = r0 * n;  // first element of the step: 16-byte aligned (r0 is a multiple of 128)
    constexpr int MAXV = GRAM_ROWS * 32 / VEC / 256;  // n <= 32: at most this many vectors per thread
    vec_t xs[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {  // every load of the step is issued before the first LDS store
      const int v = threadIdx.x + 256 * u;
      const long long idx = e0 + (long long)v * VEC;
      if (v < nvec && idx + VEC <= total) {
        xs[u] = *reinterpret_cast<const vec_t *>(Wt + idx);
      } else {
#pragma unroll
        for (int w = 0; w < VEC; ++w) xs[u][w] = (v < nvec && idx + w < total) ? Wt[idx + w] : (T)0;
      }
    }
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int v = threadIdx.x + 256 * u;
      if (v < nvec) *reinterpret_cast<vec_t *>(stage + (size_t)v * VEC) = xs[u];
    }
    __syncthreads();
    const T *rows = stage + (size_t)(wave * ROWS_PER_STEP + lk) * n;
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) {
      double v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double x = (double)rows[(size_t)(4 * g) * n + col[t]];
        v[t] = cok[t] ? x : 0.0;
      }
      int q = 0;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = t; u < NT; ++u, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[t], v[u], acc[q], 0, 0, 0);
    }
    __syncthreads();
  }
  gram_store_partial(acc, NP, 0, NP, partial);
}

// larger n: blockIdx.x = one upper tile pair (the rows are re-read once per pair, from L2 / MALL)
template <typename T>
__global__ __launch_bounds__(256) void k_gram_pair(const T *__restrict__ Wt, long long n_rows, int n, int n_tiles,
                                                   int n_pairs, double *__restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  int ti = 0, pr = blockIdx.x;
  while (pr >= n_tiles - ti) { pr -= n_tiles - ti; ++ti; }
  const int tj = ti + pr;
  const bool aok = GT * ti + li < n, bok = GT * tj + li < n;
  const int ca = min(GT * ti + li, n - 1), cb = min(GT * tj + li, n - 1);
  svd_d4 acc[3] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  const long long stride = (long long)gridDim.y * 4 * ROWS_PER_STEP;
  for (long long r0 = ((long long)blockIdx.y * 4 + wave) * ROWS_PER_STEP; r0 < n_rows; r0 += stride) {
    double va[ROWS_PER_STEP / 4], vb[ROWS_PER_STEP / 4];
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) {
      const long long row = r0 + 4 * g + lk;
      const T *wr = Wt + min(row, n_rows - 1) * n;
      const double xa = (double)wr[ca], xb = (double)wr[cb];
      va[g] = (row < n_rows && aok) ? xa : 0.0;
      vb[g] = (row < n_rows && bok) ? xb : 0.0;
    }
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[g], vb[g], acc[0], 0, 0, 0);
  }
  gram_store_partial(acc, 1, blockIdx.x, n_pairs, partial);
}

// G[gi][gj] = sum over row chunks of the partial tiles (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_gram_reduce(const double *__restrict__ partial, int chunks, int n, int n_tiles,
                                                     int n_pairs, double *__restrict__ G) {
  int ti = 0, pr = blockIdx.x;
  while (pr >= n_tiles - ti) { pr -= n_tiles - ti; ++ti; }
  const int tj = ti + pr;
  const int e = threadIdx.x, r = (e >> 6) & 3, l = e & 63;
  double sacc = 0.0;  // gridDim.y slices of the chunk range, 8 independent loads in flight
  const int c0 = (int)((long long)chunks * blockIdx.y / gridDim.y), c1 = (int)((long long)chunks * (blockIdx.y + 1) / gridDim.y);
#pragma unroll 8
  for (int c = c0; c < c1; ++c) sacc += partial[((size_t)c * n_pairs + blockIdx.x) * 256 + e];
  const int gi = GT * ti + (l >> 4) + 4 * r, gj = GT * tj + (l & 15);
  if (gi < n && gj < n) atomicAdd(&G[(size_t)gi * n + gj], sacc);
}

// column sums (only when centring is requested): one block row per column, grid-stride over rows
template <typename T>
__global__ __launch_bounds__(256) void k_colsum(const T *__restrict__ Wt, long long n_rows, int n,
                                                double *__restrict__ colsum) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (long long r = (long long)blockIdx.y * 256 + threadIdx.x; r < n_rows; r += (long long)gridDim.y * 256)
    s += (double)Wt[r * n + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&colsum[c], red[0]);
}

// G (upper tiles) -> full symmetric, optionally centred: G - s s^T / N
__global__ void k_gram_finish(double *__restrict__ G, const double *__restrict__ colsum, int n, long long n_rows,
                              int center) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n || j < i) return;
  double v = G[(size_t)i * n + j];
  if (i / GT == j / GT && j > i) v = 0.5 * (v + G[(size_t)j * n + i]);  // diagonal tiles hold both halves
  if (center) v -= colsum[i] * colsum[j] / (double)n_rows;
  G[(size_t)i * n + j] = v;
  G[(size_t)j * n + i] = v;
}

// Round-robin parallel Jacobi on the symmetric n x n matrix A (global memory), eigenvectors in V.
// np = n rounded up to even (a phantom index np-1 == n is skipped).
template <bool IN_LDS>  // IN_LDS: both n x n matrices live in LDS (n <= 64): latency ~100 ns instead of ~1.5 us
__global__ __launch_bounds__(1024) void k_jacobi(double *__restrict__ Ag, double *__restrict__ Vg, int n, int max_sweeps,
                                                 double tol, int *__restrict__ sweeps_done) {
  extern __shared__ double sm[];  // cs[np/2][2], pairs as ints, then (IN_LDS) A and V
  const int np = (n + 1) & ~1, half = np / 2;
  double *rc = sm, *rs = sm + half;
  int *pp = reinterpret_cast<int *>(sm + 2 * half), *qq = pp + half;
  double *A = IN_LDS ? sm + 3 * half + 2 : Ag;
  double *V = IN_LDS ? A + (size_t)n * n : Vg;
  __shared__ double s_off;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (IN_LDS)
    for (int q = tid; q < n * n; q += nt) A[q] = Ag[q];
  for (int q = tid; q < n * n; q += nt) V[q] = (q / n == q % n) ? 1.0 : 0.0;
  __syncthreads();
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) s_off = 0.0;
    __syncthreads();
    for (int step = 0; step < np - 1; ++step) {
      // pairs of this step + rotation angles from the current A
      for (int i = tid; i < half; i += nt) {
        int a, b;
        if (i == 0) { a = np - 1; b = step; }
        else { a = (step + i) % (np - 1); b = (step - i + np - 1) % (np - 1); }
        const int p = min(a, b), q = max(a, b);
        double c = 1.0, s = 0.0;
        if (q < n) {
          const double apq = A[(size_t)p * n + q], app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
          if (fabs(apq) > tol * sqrt(fabs(app * aqq)) && apq != 0.0) {
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s = t * c;
            atomicAdd(&s_off, 1.0);
          }
        }
        pp[i] = p; qq[i] = q; rc[i] = c; rs[i] = s;
      }
      __syncthreads();
      // rows: A <- J^T A
      for (int w = tid; w < half * n; w += nt) {
        const int i = w / n, col = w % n, p = pp[i], q = qq[i];
        if (q >= n || rs[i] == 0.0) continue;
        const double c = rc[i], s = rs[i];
        const double ap = A[(size_t)p * n + col], aq = A[(size_t)q * n + col];
        A[(size_t)p * n + col] = c * ap - s * aq;
        A[(size_t)q * n + col] = s * ap + c * aq;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int w = tid; w < half * n; w += nt) {
        const int i = w / n, row = w % n, p = pp[i], q = qq[i];
        if (q >= n || rs[i] == 0.0) continue;
        const double c = rc[i], s = rs[i];
        const double ap = A[(size_t)row * n + p], aq = A[(size_t)row * n + q];
        A[(size_t)row * n + p] = c * ap - s * aq;
        A[(size_t)row * n + q] = s * ap + c * aq;
        const double vp = V[(size_t)row * n + p], vq = V[(size_t)row * n + q];
        V[(size_t)row * n + p] = c * vp - s * vq;
        V[(size_t)row * n + q] = s * vp + c * vq;
      }
      __syncthreads();
    }
    if (s_off == 0.0) { ++sweep; break; }  // a whole sweep without a rotation
    __syncthreads();
  }
  if (tid == 0) *sweeps_done = sweep;
  if (IN_LDS) {
    __syncthreads();
    for (int q = tid; q < n * n; q += nt) { Ag[q] = A[q]; Vg[q] = V[q]; }
  }
}

// n <= 32: both matrices in LDS, 256 threads, no divides or square roots on the serial path
// (v_rcp_f64 / v_rsq_f64 seeds + Newton steps to full precision), and the two-sided update
// A <- J^T A J in a single pass from the old A:
//   A'[i][j] = c_i c_j A[i][j] + c_i t_j A[i][pj] + t_i c_j A[pi][j] + t_i t_j A[pi][pj]
// with pi = partner of index i in this step's pairing and (c_i, t_i) its rotation coefficients
// (t = -s for the smaller index of a pair, +s for the larger; c = 1, t = 0, pi = i if unpaired).
// Thread = (row group i0 = tid >> 5, column j = tid & 31), rows i = i0 + 8u.  Measured at n = 24:
// 4570 cycles per step on one wave (LDS-instruction bound), ~1200 with the work on 4 waves.
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  return y * (2.0 - x * y);
}
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  return y * (1.5 - 0.5 * x * y * y);
}

constexpr int JW = 32, JWS = JW + 1;  // max order and padded LDS stride of the small solver
__global__ __launch_bounds__(256) void k_jacobi_small(double *__restrict__ Ag, double *__restrict__ Vg, int n, int max_sweeps,
                                                     double tol, int *__restrict__ sweeps_done) {
  __shared__ double A[JW * JWS], V[JW * JWS], cc[JW], tt[JW];
  __shared__ int pn[JW];
  __shared__ int s_rot;
  const int tid = threadIdx.x;
  const int np = (n + 1) & ~1, half = np / 2;
  for (int e = tid; e < JW * JW; e += 256) {
    const int i = e / JW, j = e % JW;
    A[i * JWS + j] = (i < n && j < n) ? Ag[(size_t)i * n + j] : 0.0;
    V[i * JWS + j] = (i == j) ? 1.0 : 0.0;
  }
  if (tid == 0) s_rot = 0;
  __syncthreads();
  const int j = tid & 31, i0 = tid >> 5;
  const double tol2 = tol * tol;
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    for (int step = 0; step < np - 1; ++step) {
      if (tid < JW) { cc[tid] = 1.0; tt[tid] = 0.0; pn[tid] = tid; }
      __syncthreads();
      if (tid < half) {  // round-robin pairing (same schedule as k_jacobi)
        int a, b;
        if (tid == 0) { a = np - 1; b = step; }
        else { a = (step + tid) % (np - 1); b = (step - tid + np - 1) % (np - 1); }
        const int p = min(a, b), q = max(a, b);
        if (q < n) {
          const double apq = A[p * JWS + q], app = A[p * JWS + p], aqq = A[q * JWS + q];
          if (apq * apq > tol2 * fabs(app * aqq) && apq != 0.0) {
            const double tau = (aqq - app) * rcp_nr(2.0 * apq);
            const double w = 1.0 + tau * tau;
            const double t = copysign(1.0, tau) * rcp_nr(fabs(tau) + w * rsqrt_nr(w));
            const double c = rsqrt_nr(1.0 + t * t), sn = t * c;
            cc[p] = c; tt[p] = -sn; pn[p] = q;
            cc[q] = c; tt[q] = sn; pn[q] = p;
            s_rot = 1;  // benign race: every writer stores 1
          }
        }
      }
      __syncthreads();
      const int pj = pn[j];
      const double cj = cc[j], tj = tt[j];
      double an[4], vn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 8 * u;  // rows >= n hold zeros / identity and are left alone
        const int pi = pn[i];
        const double ci = cc[i], ti = tt[i];
        an[u] = ci * (cj * A[i * JWS + j] + tj * A[i * JWS + pj]) + ti * (cj * A[pi * JWS + j] + tj * A[pi * JWS + pj]);
        vn[u] = cj * V[i * JWS + j] + tj * V[i * JWS + pj];
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 8 * u;
        if (i < n) { A[i * JWS + j] = an[u]; V[i * JWS + j] = vn[u]; }
      }
      __syncthreads();
    }
    const int any = s_rot;
    __syncthreads();
    if (tid == 0) s_rot = 0;
    if (any == 0) { ++sweep; break; }  // a whole sweep without a rotation (uniform)
  }
  if (tid == 0) *sweeps_done = sweep;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, jj = e % n;
    Ag[e] = A[i * JWS + jj];
    Vg[e] = V[i * JWS + jj];
  }
}

// S[i][row] = sum_c Mr[c][i] (Wt[row][c] - mu[c]).  A wave takes 64 consecutive rows: the 64 x n
// tile is contiguous in memory, so it is loaded with fully coalesced accesses into a padded LDS
// tile (odd stride -> conflict-free column reads), then each lane reduces its own row.
constexpr int PC = 64;  // columns per LDS chunk
template <typename T>
__global__ __launch_bounds__(256) void k_project(const T *__restrict__ Wt, long long n_rows, int n, int r,
                                                 const double *__restrict__ Mr, const double *__restrict__ mu,
                                                 T *__restrict__ S) {
  extern __shared__ double sm[];  // Mr [n][4] (zero padded), mu [n], then per-wave tiles of T
  double *sM = sm, *smu = sm + 4 * (size_t)n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = min(n, PC), ldt = nc | 1;  // odd stride
  T *tile = reinterpret_cast<T *>(smu + n) + (size_t)wave * 64 * ldt;
  for (int q = threadIdx.x; q < 4 * n; q += blockDim.x) sM[q] = ((q & 3) < r) ? Mr[(size_t)(q >> 2) * r + (q & 3)] : 0.0;
  for (int q = threadIdx.x; q < n; q += blockDim.x) smu[q] = mu ? mu[q] : 0.0;
  __syncthreads();
  const long long stride = (long long)gridDim.x * 256;
  for (long long base = (long long)blockIdx.x * 256 + 64 * wave; base < n_rows; base += stride) {
    const int rows_here = (int)min<long long>(64, n_rows - base);
    double acc[4] = {0, 0, 0, 0};
    for (int c0 = 0; c0 < n; c0 += PC) {
      const int w = min(PC, n - c0);
      if (w == n) {  // whole rows: the tile is one contiguous block of rows_here * n elements
        const T *src = Wt + base * n;
        if (sizeof(T) == 4 && (n & 3) == 0) {  // 16 B per lane; a float4 never straddles a row
          const int n4 = n >> 2, total4 = rows_here * n4;
          int rr = lane / n4, cc = lane - rr * n4;
          const int sr = 64 / n4, sc = 64 - sr * n4;
          const float4 *src4 = reinterpret_cast<const float4 *>(src);
          for (int q = lane; q < total4; q += 64) {
            const float4 v = src4[q];
            T *d = tile + rr * ldt + 4 * cc;
            d[0] = (T)v.x; d[1] = (T)v.y; d[2] = (T)v.z; d[3] = (T)v.w;
            cc += sc; rr += sr;
            if (cc >= n4) { cc -= n4; ++rr; }
          }
        } else {
          int rr = lane / n, cc = lane - rr * n;
          const int sr = 64 / n, sc = 64 - sr * n;
          for (int q = lane; q < rows_here * n; q += 64) {
            tile[rr * ldt + cc] = src[q];
            cc += sc; rr += sr;
            if (cc >= n) { cc -= n; ++rr; }
          }
        }
      } else {
        for (int q = lane; q < rows_here * w; q += 64) tile[(q / w) * ldt + (q % w)] = Wt[(base + q / w) * n + c0 + (q % w)];
      }
      wave_sync();
      if (lane < rows_here) {
        const T *tr = tile + lane * ldt;
        for (int c = 0; c < w; ++c) {
          const double x = (double)tr[c] - smu[c0 + c];
          const double *mc = sM + 4 * (size_t)(c0 + c);
          acc[0] += x * mc[0]; acc[1] += x * mc[1]; acc[2] += x * mc[2]; acc[3] += x * mc[3];
        }
      }
      wave_sync();
    }
    if (lane < rows_here)
      for (int i = 0; i < r; ++i) S[(size_t)i * n_rows + base + lane] = (T)acc[i];
  }
}

template <typename T>
int factorize(const T *Wt, long long n_rows, int n, int n_rank, int center, T *M, T *sigma, T *S, T *means,
              double *timings) {
  hipStream_t st = nullptr;
  MVBA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  T *dW = nullptr, *dS = nullptr;
  double *dG = nullptr, *dV = nullptr, *dsum = nullptr, *dMr = nullptr, *dmu = nullptr, *dpart = nullptr;
  int *dsw = nullptr;
  hipEvent_t ev[6];
  for (auto &e : ev) hipEventCreate(&e);
  int rc = MVBA_OK;
  auto cleanup = [&]() {
    for (void *p : {(void *)dW, (void *)dS, (void *)dG, (void *)dV, (void *)dsum, (void *)dMr, (void *)dmu, (void *)dsw, (void *)dpart})
      if (p) hipFree(p);
    for (auto &e : ev) hipEventDestroy(e);
    hipStreamDestroy(st);
  };
#define SVD_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(MVBA_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
  const size_t nn = (size_t)n * n;
  SVD_HIP(hipMalloc((void **)&dW, sizeof(T) * (size_t)n_rows * n));
  SVD_HIP(hipMalloc((void **)&dS, sizeof(T) * (size_t)n_rows * n_rank));
  SVD_HIP(hipMalloc((void **)&dG, sizeof(double) * nn));
  SVD_HIP(hipMalloc((void **)&dV, sizeof(double) * nn));
  SVD_HIP(hipMalloc((void **)&dsum, sizeof(double) * n));
  SVD_HIP(hipMalloc((void **)&dMr, sizeof(double) * (size_t)n * n_rank));
  SVD_HIP(hipMalloc((void **)&dmu, sizeof(double) * n));
  SVD_HIP(hipMalloc((void **)&dsw, sizeof(int)));
  const int n_tiles = (n + GT - 1) / GT, n_pairs = n_tiles * (n_tiles + 1) / 2;
  const bool fused = n_tiles <= 2;
  const int chunks = (int)std::max<long long>(1, std::min<long long>((n_rows + 4 * ROWS_PER_STEP - 1) / (4 * ROWS_PER_STEP),
                                                                    fused ? 2048 : std::max(8, 4096 / n_pairs)));
  SVD_HIP(hipMalloc((void **)&dpart, sizeof(double) * (size_t)chunks * n_pairs * 256));
  hipEventRecord(ev[0], st);
  SVD_HIP(hipMemcpyAsync(dW, Wt, sizeof(T) * (size_t)n_rows * n, hipMemcpyHostToDevice, st));
  hipEventRecord(ev[1], st);
  SVD_HIP(hipMemsetAsync(dG, 0, sizeof(double) * nn, st));
  SVD_HIP(hipMemsetAsync(dsum, 0, sizeof(double) * n, st));
  if (n_tiles == 1)
    hipLaunchKernelGGL((k_gram_fused<T, 1>), dim3(1, chunks), dim3(256), sizeof(T) * GRAM_ROWS * n, st, dW, n_rows, n, dpart);
  else if (n_tiles == 2)
    hipLaunchKernelGGL((k_gram_fused<T, 2>), dim3(1, chunks), dim3(256), sizeof(T) * GRAM_ROWS * n, st, dW, n_rows, n, dpart);
  else
    hipLaunchKernelGGL(k_gram_pair<T>, dim3(n_pairs, chunks), dim3(256), 0, st, dW, n_rows, n, n_tiles, n_pairs, dpart);
  hipLaunchKernelGGL(k_gram_reduce, dim3(n_pairs, std::min(chunks, 32)), dim3(256), 0, st, dpart, chunks, n, n_tiles, n_pairs, dG);
  if (center) {
    const int cy = (int)std::max<long long>(1, std::min<long long>(256, n_rows / 4096));
    hipLaunchKernelGGL(k_colsum<T>, dim3(n, cy), dim3(256), 0, st, dW, n_rows, n, dsum);
  }
  hipLaunchKernelGGL(k_gram_finish, dim3((n + 127) / 128, n), dim3(128), 0, st, dG, dsum, n, n_rows, center);
  hipEventRecord(ev[2], st);
  const int np = (n + 1) & ~1;
  const size_t jl = sizeof(double) * (3 * (np / 2) + 2) + 16;
  if (n <= JW)
    hipLaunchKernelGGL(k_jacobi_small, dim3(1), dim3(256), 0, st, dG, dV, n, 60, 1e-15, dsw);
  else if (n <= 64)
    hipLaunchKernelGGL(k_jacobi<true>, dim3(1), dim3(256), jl + sizeof(double) * 2 * nn, st, dG, dV, n, 60, 1e-15, dsw);
  else
    hipLaunchKernelGGL(k_jacobi<false>, dim3(1), dim3(1024), jl, st, dG, dV, n, 60, 1e-15, dsw);
  hipEventRecord(ev[3], st);
  // eigenvalues -> host, sort, build the rank-r basis with a deterministic sign
  std::vector<double> hG(nn), hV(nn), hsum(n);
  SVD_HIP(hipMemcpyAsync(hG.data(), dG, sizeof(double) * nn, hipMemcpyDeviceToHost, st));
  SVD_HIP(hipMemcpyAsync(hV.data(), dV, sizeof(double) * nn, hipMemcpyDeviceToHost, st));
  SVD_HIP(hipMemcpyAsync(hsum.data(), dsum, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  SVD_HIP(hipStreamSynchronize(st));
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return hG[(size_t)a * n + a] > hG[(size_t)b * n + b]; });
  std::vector<double> Mr((size_t)n * n_rank), mu(n, 0.0);
  for (int i = 0; i < n; ++i) sigma[i] = (T)std::sqrt(std::max(0.0, hG[(size_t)order[i] * n + order[i]]));
  for (int i = 0; i < n_rank; ++i) {
    const int col = order[i];
    int big = 0;
    for (int c = 1; c < n; ++c)
      if (std::fabs(hV[(size_t)c * n + col]) > std::fabs(hV[(size_t)big * n + col])) big = c;
    const double sg = hV[(size_t)big * n + col] < 0.0 ? -1.0 : 1.0;  // largest component positive
    for (int c = 0; c < n; ++c) {
      Mr[(size_t)c * n_rank + i] = sg * hV[(size_t)c * n + col];
      M[(size_t)c * n_rank + i] = (T)Mr[(size_t)c * n_rank + i];
    }
  }
  if (center)
    for (int c = 0; c < n; ++c) mu[c] = hsum[c] / (double)n_rows;
  if (means)
    for (int c = 0; c < n; ++c) means[c] = (T)mu[c];
  SVD_HIP(hipMemcpyAsync(dMr, Mr.data(), sizeof(double) * Mr.size(), hipMemcpyHostToDevice, st));
  SVD_HIP(hipMemcpyAsync(dmu, mu.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
  hipEventRecord(ev[4], st);
  const int pgrid = (int)std::max<long long>(1, std::min<long long>(4096, (n_rows + 255) / 256));
  const int ldt = std::min(n, PC) | 1;
  const size_t plds = sizeof(double) * (5 * (size_t)n) + sizeof(T) * 4 * 64 * (size_t)ldt + 16;
  hipLaunchKernelGGL(k_project<T>, dim3(pgrid), dim3(256), plds, st, dW, n_rows, n, n_rank, dMr,
                     center ? dmu : nullptr, dS);
  hipEventRecord(ev[5], st);
  SVD_HIP(hipMemcpyAsync(S, dS, sizeof(T) * (size_t)n_rows * n_rank, hipMemcpyDeviceToHost, st));
  SVD_HIP(hipStreamSynchronize(st));
  SVD_HIP(hipGetLastError());
  if (timings) {
    float ms;
    hipEventElapsedTime(&ms, ev[0], ev[1]); timings[0] = ms;  // H2D
    hipEventElapsedTime(&ms, ev[1], ev[2]); timings[1] = ms;  // Gram (+sums, finish)
    hipEventElapsedTime(&ms, ev[2], ev[3]); timings[2] = ms;  // Jacobi
    hipEventElapsedTime(&ms, ev[4], ev[5]); timings[3] = ms;  // projection
    int sw = 0;
    hipMemcpy(&sw, dsw, sizeof(int), hipMemcpyDeviceToHost);
    timings[4] = sw;  // Jacobi sweeps
  }
#undef SVD_HIP
  cleanup();
  return rc;
}

}  // namespace

extern "C" int mvsvd_factorize(const void *Wt, int64_t n_rows, int32_t n_cols, int32_t dtype, int32_t n_rank,
                               int32_t center, void *M, void *sigma, void *S, void *means, double *timings_ms,
                               int32_t device) {
  if (!Wt || !M || !sigma || !S) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_rows < 1 || n_cols < 1 || n_rank < 1 || n_rank > 4 || n_rank > n_cols || n_cols > 2048)
    return fail(MVBA_ERR_BADARG, "need n_rows >= 1, 1 <= n_rank <= min(4, n_cols), n_cols <= 2048");
  if (dtype != 0 && dtype != 1) return fail(MVBA_ERR_BADARG, "dtype must be 0 (float32) or 1 (float64)");
  if (device >= 0) MVBA_HIP(hipSetDevice(device));
  MVBA_HIP(hipFuncSetAttribute((const void *)k_jacobi<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  MVBA_HIP(hipFuncSetAttribute((const void *)k_project<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  MVBA_HIP(hipFuncSetAttribute((const void *)k_project<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  if (dtype == 0)
    return factorize<float>((const float *)Wt, n_rows, n_cols, n_rank, center, (float *)M, (float *)sigma, (float *)S,
                            (float *)means, timings_ms);
  return factorize<double>((const double *)Wt, n_rows, n_cols, n_rank, center, (double *)M, (double *)sigma,
                           (double *)S, (double *)means, timings_ms);
}
