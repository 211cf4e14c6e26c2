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
// For fp64 DATA one Gram pass squares the condition number (a singular value 1e-7 below the
// largest would only be good to 1e-2), so a second, preconditioned pass follows:
//   K7d k_rotate   B = (Wt - 1 mu^T) V1            (V1 = eigenvectors of the first pass)
//   K7a again      G2 = B^T B: nearly diagonal and GRADED -- its entries are formed from B's own
//                  columns, i.e. accurate relative to the product of the two column norms, not to
//                  sigma_1^2 -- and Jacobi with the relative threshold keeps that accuracy
//   V = V1 V2, sigma = sqrt(eig G2): small singular values good to ~eps * sigma_1, like gesdd.
// Centring subtracts the column means from the rows BEFORE they enter the Gram product
// (G - s s^T / N cancels when |mean| >> spread).
// A workspace handle (mvsvd_create / load / run / destroy) keeps the matrix, every buffer, the
// stream and the events resident across calls; mvsvd_factorize is the one-shot form.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "mvba_common.h"

using namespace mvba;

namespace {

constexpr int GT = 16;             // Gram tile = one v_mfma_f64_16x16x4_f64 accumulator
constexpr int GRAM_SLICES = 32;    // second level of the fixed-order Gram reduction
constexpr int COLSUM_SLICES = 256;
constexpr int ROWS_PER_STEP = 32;  // rows consumed per unrolled step (8 MFMA k-groups of 4 rows)
typedef double svd_d4 __attribute__((ext_vector_type(4)));

// G += Wt^T Wt on the f64 matrix cores.  For 4 consecutive rows r0..r3 and column tiles (ti,tj):
//   A[i][k] = Wt[r_k][16 ti + i]   (lane l: i = l & 15, k = l >> 4)
//   B[k][j] = Wt[r_k][16 tj + j]   (lane l: k = l >> 4, j = l & 15)
// so both operands are "this lane's row (l >> 4), this lane's column (l & 15) of a tile": one
// value per lane per tile, converted to fp64 on load.  C/D: col = l & 15, row = (l >> 4) + 4 reg.
// Partial tiles go to partial[chunk][pair][256] (no atomics: millions of f64 adds onto a
// 24 x 24 matrix serialise at the memory side); k_gram_reduce sums the chunks in fixed order.
// `red`: npairs x 8 KiB of LDS the caller no longer needs (24 KiB of STATIC shared memory here once held a block of the
// fused kernel to 36 KiB: four blocks per CU instead of eight, one wave per SIMD too few to keep the matrix core fed)
__device__ __forceinline__ void gram_store_partial(const svd_d4 *acc, int npairs, int pair0, int pairs_total,
                                                   double *__restrict__ partial, double *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < npairs; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * npairs + q) * 4 + r) * 64 + lane] = acc[q][r];
  __syncthreads();
  double *out = partial + ((size_t)blockIdx.y * pairs_total + pair0) * 256;
  for (int e = threadIdx.x; e < npairs * 256; e += 256) {
    const int q = e >> 8, rl = e & 255;  // rl = r * 64 + lane
    out[e] = (red[((0 * npairs + q) << 8) + rl] + red[((1 * npairs + q) << 8) + rl]) + (red[((2 * npairs + q) << 8) + rl] + red[((3 * npairs + q) << 8) + rl]);
  }
}

// n <= 32: one pass over the rows forms every tile pair.  MODE 1: n <= 16, one MFMA per 4 rows.  MODE 3: 24 < n <= 32,
// the three tile pairs (0,0), (0,1), (1,1).  MODE 2: 16 < n <= 24 (config 5: 24 columns), TWO MFMAs: tile (0,0) and a
// PACKED tile whose rows are columns 8..23 and whose columns are [16..23 | 0..7] -- it holds G[8..23][16..23] and
// G[16..23][0..7], i.e. everything tile (0,0) lacks; the half-empty (0,1) and quarter-full (1,1) tiles of MODE 3 cost a
// third more matrix-core time for the same 300 entries (the kernel is bound by the f64 MFMA issue rate).
// A workgroup takes GRAM_ROWS consecutive rows per step: the rows are one contiguous byte range of
// Wt, streamed with 16-byte loads by all 256 threads into LDS (a lane-per-element gather of the
// MFMA operands straight from HBM ran at 1.8 TB/s), then every wave picks the operands of its
// 32 rows out of LDS.  Rows past the end are zero-filled in LDS.
constexpr int GRAM_ROWS = 4 * ROWS_PER_STEP;  // 128 rows per workgroup step
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_gram_fused(const T *__restrict__ Wt, long long n_rows, int n,
                                                    const double *__restrict__ mu, double *__restrict__ partial) {
  extern __shared__ double gram_lds_raw[];
  T *stage = reinterpret_cast<T *>(gram_lds_raw);  // GRAM_ROWS x n, row-major like Wt
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  constexpr int NP = MODE == 1 ? 1 : (MODE == 2 ? 2 : 3);  // MFMAs per 4 rows = partial tiles
  constexpr int NC = MODE == 1 ? 1 : (MODE == 2 ? 3 : 2);  // operand columns a lane reads per row
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  svd_d4 acc[3] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  int col[NC];
  bool cok[NC];
  double muc[NC];  // column mean to subtract (0 without centring)
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    // MODE 1/3: column 16 t + li.  MODE 2: li | 8 + li | (li < 8 ? 16 + li : li - 8)
    const int c = MODE == 2 ? (t == 0 ? li : (t == 1 ? 8 + li : (li < 8 ? 16 + li : li - 8))) : GT * t + li;
    cok[t] = c < n; col[t] = min(c, n - 1); muc[t] = mu ? mu[col[t]] : 0.0;
  }
  // (PMC: the general operand path -- subtract the mean, mask absent columns and rows past the end -- was 12 vector
  // instructions per MFMA, 122 M issue cycles next to 160 M of matrix-core time, and the two did not overlap)
  const bool lean = mu == nullptr && n == (MODE == 1 ? 16 : (MODE == 2 ? 24 : 32));
  const long long total = n_rows * n;
  const int nvec = GRAM_ROWS * n / VEC;  // GRAM_ROWS * n is a multiple of 4
  constexpr int MAXV = GRAM_ROWS * 32 / VEC / 256;  // n <= 32: at most this many vectors per thread
  vec_t xs[MAXV];
  auto load_step = [&](long long r0) {  // every load of a step is issued before anything waits for one
    const long long e0 = r0 * n;        // first element of the step: 16-byte aligned (r0 is a multiple of 128)
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int v = threadIdx.x + 256 * u;
      const long long idx = e0 + (long long)v * VEC;
      if (v < nvec && idx + VEC <= total) {
        xs[u] = *reinterpret_cast<const vec_t *>(Wt + idx);
      } else {
#pragma unroll
        for (int w = 0; w < VEC; ++w) xs[u][w] = (v < nvec && idx + w < total) ? Wt[idx + w] : (T)0;
      }
    }
  };
  const long long r_first = (long long)blockIdx.y * GRAM_ROWS, r_stride = (long long)gridDim.y * GRAM_ROWS;
  if (r_first < n_rows) load_step(r_first);
  for (long long r0 = r_first; r0 < n_rows; r0 += r_stride) {
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int v = threadIdx.x + 256 * u;
      if (v < nvec) *reinterpret_cast<vec_t *>(stage + (size_t)v * VEC) = xs[u];
    }
    __syncthreads();
    if (r0 + r_stride < n_rows) load_step(r0 + r_stride);  // the next step's rows travel while this step's are multiplied
    const T *rows = stage + (size_t)(wave * ROWS_PER_STEP + lk) * n;
    const long long row0 = r0 + wave * ROWS_PER_STEP + lk;
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) {
      double v[NC];
      if (lean) {  // (uniform) no centring, every operand column exists: a row past the end is zeros in LDS and stays zero
#pragma unroll
        for (int t = 0; t < NC; ++t) v[t] = (double)rows[(4 * g) * n + col[t]];
      } else {
        const bool rok = row0 + 4 * g < n_rows;  // rows past the end are zeros in LDS: keep them zero when centring
#pragma unroll
        for (int t = 0; t < NC; ++t) {
          const double x = (double)rows[(size_t)(4 * g) * n + col[t]] - muc[t];
          v[t] = (cok[t] && rok) ? x : 0.0;
        }
      }
      if (MODE == 2) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[0], v[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[1], v[2], acc[1], 0, 0, 0);
      } else {
        int q = 0;
#pragma unroll
        for (int t = 0; t < NC; ++t)
#pragma unroll
          for (int u = t; u < NC; ++u, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[t], v[u], acc[q], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  gram_store_partial(acc, NP, 0, NP, partial, gram_lds_raw);  // (the staging buffer: the last step ended with a barrier)
}

// larger n: blockIdx.x = one upper tile pair (the rows are re-read once per pair, from L2 / MALL)
template <typename T>
__global__ __launch_bounds__(256) void k_gram_pair(const T *__restrict__ Wt, long long n_rows, int n, int n_tiles,
                                                   int n_pairs, const double *__restrict__ mu, double *__restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  int ti = 0, pr = blockIdx.x;
  while (pr >= n_tiles - ti) { pr -= n_tiles - ti; ++ti; }
  const int tj = ti + pr;
  const bool aok = GT * ti + li < n, bok = GT * tj + li < n;
  const int ca = min(GT * ti + li, n - 1), cb = min(GT * tj + li, n - 1);
  const double mua = mu ? mu[ca] : 0.0, mub = mu ? mu[cb] : 0.0;
  svd_d4 acc[3] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  const long long stride = (long long)gridDim.y * 4 * ROWS_PER_STEP;
  for (long long r0 = ((long long)blockIdx.y * 4 + wave) * ROWS_PER_STEP; r0 < n_rows; r0 += stride) {
    double va[ROWS_PER_STEP / 4], vb[ROWS_PER_STEP / 4];
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) {
      const long long row = r0 + 4 * g + lk;
      const T *wr = Wt + min(row, n_rows - 1) * n;
      const double xa = (double)wr[ca] - mua, xb = (double)wr[cb] - mub;
      va[g] = (row < n_rows && aok) ? xa : 0.0;
      vb[g] = (row < n_rows && bok) ? xb : 0.0;
    }
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[g], vb[g], acc[0], 0, 0, 0);
  }
  __shared__ double red1[4 * 4 * 64];
  gram_store_partial(acc, 1, blockIdx.x, n_pairs, partial, red1);
}

// Two fixed-order levels instead of atomics: k_gram_reduce sums a slice of the row chunks per block
// (slice s of tile pair p -> part2[s][p][256]), k_gram_finish adds the slices in order and writes the full symmetric G.
__global__ __launch_bounds__(256) void k_gram_reduce(const double *__restrict__ partial, int chunks, int n_pairs,
                                                     double *__restrict__ part2) {
  const int e = threadIdx.x;
  double sacc = 0.0;  // gridDim.y slices of the chunk range, 8 independent loads in flight
  const int c0 = (int)((long long)chunks * blockIdx.y / gridDim.y), c1 = (int)((long long)chunks * (blockIdx.y + 1) / gridDim.y);
#pragma unroll 8
  for (int c = c0; c < c1; ++c) sacc += partial[((size_t)c * n_pairs + blockIdx.x) * 256 + e];
  part2[((size_t)blockIdx.y * n_pairs + blockIdx.x) * 256 + e] = sacc;
}

// element (row, col) of a 16 x 16 MFMA C/D tile as gram_store_partial lays it out: col = l & 15, row = (l >> 4) + 4 reg
__device__ __forceinline__ int tile_elem(int row, int col) { return ((row >> 2) << 6) | ((row & 3) << 4) | col; }

// G[i][j] = G[j][i] = sum over the slices (in order) of the tile entry that holds it.  `packed`: the two-tile layout
// of k_gram_fused MODE 2.  Entries a layout holds twice (both halves of a diagonal tile) are averaged.
__global__ void k_gram_finish(const double *__restrict__ part2, int slices, int n_pairs, int n_tiles, int packed,
                              double *__restrict__ G, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= n || j < i) return;
  auto fetch = [&](int pair, int elem) {  // (every load issued before the first add: one memory latency, not `slices`)
    double t[GRAM_SLICES];
#pragma unroll
    for (int s = 0; s < GRAM_SLICES; ++s) t[s] = part2[((size_t)min(s, slices - 1) * n_pairs + pair) * 256 + elem];
    double v = 0.0;
#pragma unroll
    for (int s = 0; s < GRAM_SLICES; ++s) v += s < slices ? t[s] : 0.0;
    return v;
  };
  double v;
  if (packed) {
    if (j < 16) v = (i == j) ? fetch(0, tile_elem(i, j)) : 0.5 * (fetch(0, tile_elem(i, j)) + fetch(0, tile_elem(j, i)));
    else if (i < 8) v = fetch(1, tile_elem(j - 8, 8 + i));                      // G[j][i], j >= 16 > 8 > i
    else if (i < 16) v = fetch(1, tile_elem(i - 8, j - 16));                    // G[i][j], 8 <= i < 16 <= j
    else v = (i == j) ? fetch(1, tile_elem(i - 8, j - 16)) : 0.5 * (fetch(1, tile_elem(i - 8, j - 16)) + fetch(1, tile_elem(j - 8, i - 16)));
  } else {
    const int ti = i / GT, tj = j / GT;
    const int pair = ti * n_tiles - ti * (ti - 1) / 2 + (tj - ti);
    v = fetch(pair, tile_elem(i - GT * ti, j - GT * tj));
    if (ti == tj && j > i) v = 0.5 * (v + fetch(pair, tile_elem(j - GT * tj, i - GT * ti)));  // diagonal tiles hold both halves
  }
  G[(size_t)i * n + j] = v;
  G[(size_t)j * n + i] = v;
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
  if (threadIdx.x == 0) colsum[(size_t)blockIdx.y * n + c] = red[0];  // one partial per (slice, column): no atomics
}

__global__ void k_mean_from_sum(const double *__restrict__ colsum, int slices, int n, long long n_rows, double *__restrict__ mu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0.0;
  for (int q = 0; q < slices; ++q) s += colsum[(size_t)q * n + c];  // fixed order
  mu[c] = s / (double)n_rows;
}

// B[row][j] = sum_c (Wt[row][c] - mu[c]) V[c][j]   (fp64 out): the preconditioning rotation of the
// second pass, and V1 V2 at the end (rows = n).  64 x 64 output tile per block, k in steps of 16
// through LDS, 4 x 4 outputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void k_rotate(const T *__restrict__ Wt, long long n_rows, int n, const double *__restrict__ mu,
                                                const double *__restrict__ V, double *__restrict__ B) {
  __shared__ double sW[64][17], sV[16][65];
  const long long r0 = (long long)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;  // thread -> rows tr + 16 u, columns tc + 16 v
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  for (int k0 = 0; k0 < n; k0 += 16) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int r = e >> 4, k = e & 15;
      const long long row = r0 + r;
      sW[r][k] = (row < n_rows && k0 + k < n) ? (double)Wt[row * n + k0 + k] - (mu ? mu[k0 + k] : 0.0) : 0.0;
    }
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int k = e >> 6, j = e & 63;
      sV[k][j] = (k0 + k < n && j0 + j < n) ? V[(size_t)(k0 + k) * n + j0 + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = sW[tr + 16 * u][k];
#pragma unroll
      for (int v = 0; v < 4; ++v) b[v] = sV[k][tc + 16 * v];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long row = r0 + tr + 16 * u;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = j0 + tc + 16 * v;
      if (row < n_rows && j < n) B[row * n + j] = acc[u][v];
    }
  }
}

// The same product for n <= 32 columns and many rows (the preconditioning rotation of a tall matrix: 1.06 ms for 5 M x 24 with
// the generic kernel above -- 64 x 64 output tiles of which 24 columns exist, element-wise loads -- against 0.35 ms of HBM
// time for its 1.9 GB).  Built like k_gram_fused: a workgroup takes 128 consecutive rows per step, one contiguous byte range
// streamed with 16-byte loads by all threads into LDS (the next step's loads in flight during this step's products); every
// wave rotates its 32 rows on the f64 matrix cores -- A[i][k] = W[16 t + i][4 g + k] - mu, B[k][j] = V[4 g + k][16 c + j]
// (kept in registers) -- and stores the result from the MFMA output layout.
template <typename T>
__global__ __launch_bounds__(256) void k_rotate_rows(const T *__restrict__ Wt, long long n_rows, int n, const double *__restrict__ mu,
                                                     const double *__restrict__ V, double *__restrict__ B) {
  extern __shared__ double rot_lds[];
  T *stage = reinterpret_cast<T *>(rot_lds);                       // GRAM_ROWS x n, row-major like Wt
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const int nct = (n + 15) / 16, ng = (n + 3) / 4;                 // column tiles (<= 2), k-steps (<= 8)
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  double vb[2][8], mk[8];                                          // this lane's B operands and the means of its k columns
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int k = 4 * g + lk;
    mk[g] = (mu && k < n) ? mu[k] : 0.0;
#pragma unroll
    for (int c = 0; c < 2; ++c) vb[c][g] = (k < n && 16 * c + li < n) ? V[(size_t)k * n + 16 * c + li] : 0.0;
  }
  const long long total = n_rows * n;
  const int nvec = GRAM_ROWS * n / VEC;
  constexpr int MAXV = GRAM_ROWS * 32 / VEC / 256;
  vec_t xs[MAXV];
  auto load_step = [&](long long r0) {
    const long long e0 = r0 * n;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int v = threadIdx.x + 256 * u;
      const long long idx = e0 + (long long)v * VEC;
      if (v < nvec && idx + VEC <= total) {
        xs[u] = *reinterpret_cast<const vec_t *>(Wt + idx);
      } else {
#pragma unroll
        for (int w = 0; w < VEC; ++w) xs[u][w] = (v < nvec && idx + w < total) ? Wt[idx + w] : (T)0;
      }
    }
  };
  const long long r_first = (long long)blockIdx.x * GRAM_ROWS, r_stride = (long long)gridDim.x * GRAM_ROWS;
  if (r_first < n_rows) load_step(r_first);
  for (long long r0 = r_first; r0 < n_rows; r0 += r_stride) {
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
      const int v = threadIdx.x + 256 * u;
      if (v < nvec) *reinterpret_cast<vec_t *>(stage + (size_t)v * VEC) = xs[u];
    }
    __syncthreads();
    if (r0 + r_stride < n_rows) load_step(r0 + r_stride);
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // the wave's two 16-row tiles
      const T *rows = stage + (size_t)(wave * ROWS_PER_STEP + 16 * t + li) * n;
      svd_d4 acc[2] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
#pragma unroll
      for (int g = 0; g < 8; ++g) {  // (static indices into the operand registers; ng and nct are uniform)
        if (g >= ng) break;
        const int k = 4 * g + lk;
        const double a = (k < n) ? (double)rows[k] - mk[g] : 0.0;
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vb[0][g], acc[0], 0, 0, 0);
        if (nct > 1) acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vb[1][g], acc[1], 0, 0, 0);
      }
      // C/D layout: col = li, row = lk + 4 q: four 128-byte row segments per store instruction, straight to HBM (through a
      // second LDS buffer and contiguous 16-byte stores the kernel held two workgroups per CU and ran at 2.4 TB/s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long row = r0 + wave * ROWS_PER_STEP + 16 * t + lk + 4 * q;
        if (row < n_rows) {
          double *orow = B + row * n;
          if (li < n) orow[li] = acc[0][q];
          if (16 + li < n) orow[16 + li] = acc[1][q];
        }
      }
    }
    __syncthreads();  // every wave has read its rows: the staging buffer may take the next step
  }
}

// ------------------------------------------------------------------ wide matrices (n > WIDE_MIN columns)
// The n x n eigenproblem of the Gram route runs in ONE workgroup: 0.4 s at n = 256, 3.3 s at 512, minutes beyond -- and its
// projection staged 5 n doubles in LDS, so that n >= ~900 did not launch at all (round 5, tools/time_svd_wide.py).  The
// callers only ever keep r <= 4 triplets (ref lib/factorization.py:10-15), so beyond WIDE_MIN columns (and up to 16 triplets) the leading subspace is
// found by block power iteration with Rayleigh-Ritz on W^T W applied IMPLICITLY, WB = 32 vectors wide, two passes over W per
// iteration and no n x n matrix anywhere:
//   B = (W - mu) Q           k_wq    (N x WB, f64 matrix cores, W staged through LDS in 64 x 64 tiles)
//   H = B^T B = Q^T W^T W Q  the fused small Gram kernel on B; Jacobi (k_jacobi_small) -> theta, Y
//   Z = (W - mu)^T (B Y)     k_rotate_rows on B, then k_wtb (n x WB, partial sums per row chunk, fixed-order reduction)
//   Q <- Q Y                 k_ritz  (Ritz vectors; residuals ||z_j - theta_j q_j||)
//   Q <- orth(Z)             Gram of Z (n x WB), Cholesky-QR in Ritz order (k_chol_orth + k_rotate_rows), twice
// Convergence factor per iteration (sigma_33 / sigma_r)^2; a measurement matrix (rank 3-4 + noise) converges in 3-6 iterations.
// The products are formed from W itself, never from an accumulated Gram matrix, and the basis is kept GRADED (Ritz order,
// triangular orthogonalisation), so a small sigma_r keeps the accuracy of the two-pass form (rounding ~ eps sigma_1 sigma_r, not
// eps sigma_1^2).  At the end B = (W - mu) Q once more: its Gram matrix is nearly diagonal and graded, a last Jacobi gives
// sigma = sqrt(theta) and the final rotation, S is B's leading columns (S = M^T W by construction: no projection pass).
// sigma[WB..] is not computed (NaN).
constexpr int WIDE_MIN = 64;     // columns above which the implicit form takes over (n <= 64: the eigenproblem sits in LDS, 6 ms)
constexpr int JACOBI_MAX = 256;  // columns up to which the Gram + Jacobi route stays available (n_rank > 16: 0.4 s of Jacobi there)
constexpr int MVSVD_MAX_COLS = 3 * 4096;  // three rows of W per image at MAX_CAMERAS of the bundle-adjustment engine
constexpr int WB = 32;       // block width (= the small solver's order)
constexpr int WT = 64;       // tile edge of the two products

// The two products share their staging: a 64 x 64 tile of W (kept in its own dtype; converted and centred as the MFMA operand is
// read) and a 64 x WB tile of the other factor, fetched with 16-byte loads into REGISTERS while the previous tile is multiplied
// and put into LDS behind a barrier (the first version loaded element by element, stored doubles and waited for every tile's
// loads before its products: 0.9 / 1.6 TB/s on fp32 / fp64 input).  VEC: the rows of W are 16-byte aligned (n a multiple of
// 16 / sizeof(T)); otherwise the same tile comes in element by element.
template <typename T>
struct WideTile {
  static constexpr int VEC = 16 / (int)sizeof(T);          // elements per 16-byte load
  static constexpr int LD = WT + VEC;                       // padded row of the W tile (keeps 16-byte alignment)
  static constexpr int NV = WT * WT / VEC / 256;            // 16-byte loads per thread and W tile (4 / 8)
  typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
  vec_t w[NV];
  double2 o[WT * WB / 2 / 256];                             // the other factor's tile: 4 loads of two doubles per thread
  // W tile rows r0 .. r0 + 63 (below r_end), columns c0 .. c0 + 63 (below n)
  __device__ __forceinline__ void load_w(const T *__restrict__ Wt, long long r0, long long r_end, int c0, int n, bool vec) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int v = threadIdx.x + 256 * u, r = v / (WT / VEC), c = (v % (WT / VEC)) * VEC;
      const long long row = r0 + r;
      if (vec && row < r_end && c0 + c < n) {
        w[u] = *reinterpret_cast<const vec_t *>(Wt + row * n + c0 + c);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[u][e] = (row < r_end && c0 + c + e < n) ? Wt[row * n + c0 + c + e] : (T)0;
      }
    }
  }
  // 64 rows x WB doubles from a row-major [.][WB] matrix, rows k0 .. k0 + 63 (below k_end)
  __device__ __forceinline__ void load_o(const double *__restrict__ O, long long k0, long long k_end) {
#pragma unroll
    for (int u = 0; u < WT * WB / 2 / 256; ++u) {
      const int v = threadIdx.x + 256 * u, r = v / (WB / 2), c = (v % (WB / 2)) * 2;
      o[u] = (k0 + r < k_end) ? *reinterpret_cast<const double2 *>(O + (k0 + r) * WB + c) : double2{0.0, 0.0};
    }
  }
  __device__ __forceinline__ void store(T (*sW)[LD], double (*sO)[WB + 2]) const {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int v = threadIdx.x + 256 * u, r = v / (WT / VEC), c = (v % (WT / VEC)) * VEC;
      *reinterpret_cast<vec_t *>(&sW[r][c]) = w[u];
    }
#pragma unroll
    for (int u = 0; u < WT * WB / 2 / 256; ++u) {
      const int v = threadIdx.x + 256 * u, r = v / (WB / 2), c = (v % (WB / 2)) * 2;
      *reinterpret_cast<double2 *>(&sO[r][c]) = o[u];
    }
  }
};

// B[row][j] = sum_c (Wt[row][c] - mu[c]) Q[c][j], j < WB.  One workgroup per 64 rows, wave w its rows 16 w .. 16 w + 15.
template <typename T>
__global__ __launch_bounds__(256) void k_wq(const T *__restrict__ Wt, long long n_rows, int n, const double *__restrict__ mu,
                                            const double *__restrict__ Q, double *__restrict__ B) {
  typedef WideTile<T> Tile;
  __shared__ __attribute__((aligned(16))) T sW[WT][Tile::LD];
  __shared__ __attribute__((aligned(16))) double sQ[WT][WB + 2];
  __shared__ double smu[WT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const long long r0 = (long long)blockIdx.x * WT;
  const bool vec = n % Tile::VEC == 0;
  svd_d4 acc[2] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  Tile t;
  t.load_w(Wt, r0, n_rows, 0, n, vec);
  t.load_o(Q, 0, n);
  for (int c0 = 0; c0 < n; c0 += WT) {
    __syncthreads();  // the previous tile's products are done
    t.store(sW, sQ);
    if (threadIdx.x < WT) smu[threadIdx.x] = (mu && c0 + (int)threadIdx.x < n) ? mu[c0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (c0 + WT < n) {  // the next tile's loads fly under this tile's products
      t.load_w(Wt, r0, n_rows, c0 + WT, n, vec);
      t.load_o(Q, c0 + WT, n);
    }
#pragma unroll
    for (int kk = 0; kk < WT / 4; ++kk) {
      const int k = 4 * kk + lk;
      const double a = (double)sW[16 * wave + li][k] - smu[k];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sQ[k][li], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sQ[k][16 + li], acc[1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long long row = r0 + 16 * wave + lk + 4 * q;
    if (row < n_rows) {
      B[row * WB + li] = acc[0][q];
      B[row * WB + 16 + li] = acc[1][q];
    }
  }
}

// zpart[chunk][c][j] = sum over the chunk's rows of (Wt[row][c] - mu[c]) B[row][j].  blockIdx.x = 64 columns of W (wave w its
// columns 16 w ..), blockIdx.y = row chunk (rows_per_chunk, a multiple of 64).
template <typename T>
__global__ __launch_bounds__(256) void k_wtb(const T *__restrict__ Wt, long long n_rows, int n, const double *__restrict__ mu,
                                             const double *__restrict__ B, long long rows_per_chunk, double *__restrict__ zpart) {
  typedef WideTile<T> Tile;
  __shared__ __attribute__((aligned(16))) T sW[WT][Tile::LD];
  __shared__ __attribute__((aligned(16))) double sB[WT][WB + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const int c0 = blockIdx.x * WT;
  const long long rb = (long long)blockIdx.y * rows_per_chunk, re = min(n_rows, rb + rows_per_chunk);
  const bool vec = n % Tile::VEC == 0;
  const int ca = min(c0 + 16 * wave + li, n - 1);
  const double mua = mu ? mu[ca] : 0.0;  // this lane's column of W
  svd_d4 acc[2] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  Tile t;
  if (rb < re) {
    t.load_w(Wt, rb, re, c0, n, vec);
    t.load_o(B, rb, re);
  }
  for (long long r0 = rb; r0 < re; r0 += WT) {
    __syncthreads();
    t.store(sW, sB);
    __syncthreads();
    if (r0 + WT < re) {
      t.load_w(Wt, r0 + WT, re, c0, n, vec);
      t.load_o(B, r0 + WT, re);
    }
    const int rows_here = (int)min<long long>(WT, re - r0);  // (a zero row of W is -mu after centring: the rows past the end must not count)
#pragma unroll
    for (int kk = 0; kk < WT / 4; ++kk) {
      const int k = 4 * kk + lk;
      const double a = k < rows_here ? (double)sW[k][16 * wave + li] - mua : 0.0;  // A[i][k] = W[row k][column i]
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[k][li], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sB[k][16 + li], acc[1], 0, 0, 0);
    }
  }
  double *out = zpart + (size_t)blockIdx.y * n * WB;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + 16 * wave + lk + 4 * q;
    if (c < n) {
      out[(size_t)c * WB + li] = acc[0][q];
      out[(size_t)c * WB + 16 + li] = acc[1][q];
    }
  }
}

// out[e] = sum over the chunks, in order, of part[chunk][e]
__global__ void k_sum_chunks(const double *__restrict__ part, int chunks, long long count, double *__restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  double s = 0.0;
  int c = 0;
  for (; c + 8 <= chunks; c += 8) {  // eight loads in flight, added in order (one load per trip waited a memory latency per chunk: 0.14 ms at 400 chunks)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * count + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; c < chunks; ++c) s += part[(size_t)c * count + e];
  out[e] = s;
}

// Q[n][WB]: a fixed pseudo-random start (the same for every call: the result must not depend on a seed the caller cannot see)
__device__ __forceinline__ double wide_hash(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (double)(long long)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;  // [-1, 1)
}
__global__ void k_wide_init(double *__restrict__ Q, int n, unsigned long long salt) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (long long)n * WB) Q[e] = wide_hash((unsigned long long)e * 0x9e3779b97f4a7c15ull + salt);
}

// Ritz rotation of the basis, one thread per row: q <- q Y (Y = eigenvectors of H; `Hr` = Y^T H Y as the small solver leaves it:
// theta on the diagonal, whatever it did not rotate away beside it), and the residual partials
// respart[block][j] = sum_rows (z_j - sum_i q_i Hr_ij)^2 against Z = W^T (B Y), which the caller formed from the ROTATED product.
//  * Rotating Z itself -- z <- z Y -- forms a vector of length sigma_4^2 as a combination of vectors of length sigma_1^2: a
//    sigma_4 = 1e-7 sigma_1 then comes out with 1e-16 / 1e-14 of relative noise, most of it outside the row space of W, and the
//    block loses that direction to 2e-3 (found with a NumPy restatement of the iteration).  B Y loses 1e-16 / 1e-7 and W^T maps
//    whatever it is given into the row space.
//  * The residual is taken against the WHOLE row of Hr, not theta_j alone: the solver leaves |Hr_1j| up to 4.5e-16 theta_1 (its
//    absolute floor), i.e. q_j keeps 1e-16 of q_1 -- harmless for q_j, but theta_1 times it is 1e-2 of a theta_j = 1e-14 theta_1
//    and would sit in the plain residual for ever.  Projected (the Galerkin condition, numerically), what is left is the part of
//    W^T W q_j OUTSIDE the block: the quantity the next iteration can still improve.
// Z == nullptr: the rotation alone.
__global__ __launch_bounds__(256) void k_ritz(double *__restrict__ Q, const double *__restrict__ Z, int n, const double *__restrict__ Y,
                                              const double *__restrict__ Hr, double *__restrict__ respart) {
  __shared__ double sY[WB][WB], sH[WB][WB], sred[4][WB];
  for (int e = threadIdx.x; e < WB * WB; e += 256) { sY[e >> 5][e & 31] = Y[e]; sH[e >> 5][e & 31] = Hr[e]; }
  __syncthreads();
  const int row = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool ok = row < n;
  double q[WB];
#pragma unroll
  for (int k = 0; k < WB; ++k) q[k] = ok ? Q[(size_t)row * WB + k] : 0.0;
#pragma unroll 1
  for (int j = 0; j < WB; ++j) {
    double qj = 0.0;
#pragma unroll
    for (int k = 0; k < WB; ++k) qj = fma(q[k], sY[k][j], qj);
    if (ok) Q[(size_t)row * WB + j] = qj;
  }
  if (!Z || !respart) return;  // (uniform)
#pragma unroll
  for (int k = 0; k < WB; ++k) q[k] = ok ? Q[(size_t)row * WB + k] : 0.0;  // the rotated row (this thread's own stores)
#pragma unroll 1
  for (int j = 0; j < WB; ++j) {
    double zp = 0.0;
#pragma unroll
    for (int k = 0; k < WB; ++k) zp = fma(q[k], sH[k][j], zp);
    double d = (ok ? Z[(size_t)row * WB + j] : 0.0) - zp;
    d *= d;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
    if (lane == 0) sred[wave][j] = d;
  }
  __syncthreads();
  if (threadIdx.x < WB) respart[(size_t)blockIdx.x * WB + threadIdx.x] = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

// Orthonormalisation step: Cholesky-QR in Ritz order.  The columns of Z differ in length by (sigma_1 / sigma_32)^2, so C = Z^T Z
// is brought to unit diagonal (d_j = C_jj^-1/2: its entries are then accurate to eps whatever the columns' lengths), its rows and
// columns are put in the order of DESCENDING Ritz value (`Hd`: the diagonal the small solver left in H; nullptr: as they are), and
// P^T D C D P = L L^T.  Q = Z D P L^-T then has orthonormal columns, sorted by Ritz value, and column r only mixes the columns of
// rank <= r -- a small triplet's vector is cleaned of the large ones and never the other way round, so the basis stays graded
// and the next Rayleigh-Ritz step keeps the small triplets' relative accuracy.  (A symmetric orthogonalisation -- eigenvectors of C --
// mixes every column with the 28 noise columns at the 1 / sqrt(n) level: B^T B was then no longer graded, a sigma_4 = 1e-7 sigma_1
// came out of the Ritz step to 1e-2 only, and the residual test stalled there.)  A pivot below 1e-12 -- the column depends on
// its predecessors to 1e-6: the block has lost rank, W has fewer than WB independent rows / columns, or a column of Z is exactly
// zero -- gives a zero column and flag[r] = 1; k_wide_refill then puts a pseudo-random column in its place before the second pass.
// One wave; lane i owns row i of the factor.
__global__ __launch_bounds__(64) void k_chol_orth(const double *__restrict__ C, const double *__restrict__ Hd, double *__restrict__ Tm,
                                                  int *__restrict__ flag) {
  __shared__ double A[WB][WB + 1], Li[WB][WB + 1], sd[WB], sth[WB];
  __shared__ int perm[WB], dead[WB];
  const int lane = threadIdx.x;
  if (lane < WB) {
    const double c = C[lane * WB + lane];
    sd[lane] = c > 0.0 ? 1.0 / sqrt(c) : 0.0;
    sth[lane] = Hd ? Hd[lane * WB + lane] : 0.0;
  }
  __syncthreads();
  if (lane < WB) {  // rank of this column by Ritz value (ties: by index)
    int r = 0;
    for (int k = 0; k < WB; ++k) r += (sth[k] > sth[lane]) || (sth[k] == sth[lane] && k < lane);
    perm[Hd ? r : lane] = lane;
  }
  __syncthreads();
  for (int e = lane; e < WB * WB; e += 64) {
    const int i = e >> 5, j = e & 31, pi = perm[i], pj = perm[j];
    A[i][j] = i == j ? (sd[pi] > 0.0 ? 1.0 : 0.0) : C[pi * WB + pj] * sd[pi] * sd[pj];
    Li[i][j] = 0.0;
  }
  __syncthreads();
  // right-looking Cholesky, the factor overwrites the lower triangle of A
  for (int j = 0; j < WB; ++j) {
    if (lane == j) {
      const double piv = A[j][j];
      dead[j] = !(piv > 1e-12);
      A[j][j] = dead[j] ? 1.0 : sqrt(piv);
    }
    __syncthreads();
    const bool dj = dead[j] != 0;
    if (lane > j && lane < WB) A[lane][j] = dj ? 0.0 : A[lane][j] / A[j][j];
    __syncthreads();
    if (lane > j && lane < WB && !dj) {
      const double lij = A[lane][j];
      for (int k = j + 1; k <= lane; ++k) A[lane][k] -= lij * A[k][j];
    }
    __syncthreads();
  }
  // L^-1 by forward substitution, lane c its column c
  if (lane < WB) {
    const int c = lane;
    for (int i = c; i < WB; ++i) {
      double v = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) v -= A[i][k] * Li[k][c];
      Li[i][c] = v / A[i][i];
    }
  }
  __syncthreads();
  // T[p_i][r] = d_{p_i} (L^-1)[r][i], i <= r
  for (int e = lane; e < WB * WB; e += 64) {
    const int i = e >> 5, r = e & 31;
    Tm[perm[i] * WB + r] = (i <= r && !dead[r]) ? sd[perm[i]] * Li[r][i] : 0.0;
  }
  if (lane < WB) flag[lane] = dead[lane];
}
__global__ void k_wide_refill(double *__restrict__ Q, int n, const int *__restrict__ flag, unsigned long long salt) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)n * WB) return;
  const int j = (int)(e & (WB - 1));
  if (flag[j]) Q[e] = wide_hash((unsigned long long)e * 0x9e3779b97f4a7c15ull + salt) * rsqrt((double)n / 3.0);  // ~unit length
}

// S[i][row] = sgn[i] B[row][col[i]] (the workspace's dtype): the leading columns of the rotated B ARE M^T (W - mu)
template <typename T>
__global__ void k_take_cols(const double *__restrict__ B, long long n_rows, int r, const int *__restrict__ col, const double *__restrict__ sgn, T *__restrict__ S) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  for (int i = 0; i < r; ++i) S[(size_t)i * n_rows + row] = (T)(sgn[i] * B[row * WB + col[i]]);
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
            s_off = 1.0;  // (a flag: every writer stores the same value)
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

// n <= 32: the small eigen-solver (one workgroup, both matrices in LDS, cyclic Jacobi in the round-robin order).
// A step took 0.8-1 us however it was arranged in round 2 (two or four barriers, pipelined or not) and in the first
// attempts of round 3 (bank-conflict-free layouts, one barrier, every thread computing its own rotations): what a step
// costs is the number of INSTRUCTIONS a wave executes in it -- ~300 with the roles derived inside the step loop (two
// integer divisions by the runtime order, the pairing looked up or computed, address arithmetic) -- times the waves per
// SIMD.  So (0.79 -> 0.45 us per step, 0.17 -> 0.10 ms at 24 columns):
//  * the pairing never changes -- pairs are always the indices (2I, 2I + 1); what changes is where the data sits: a step
//    writes its results to the positions the tournament's rotation assigns them (player 0 stays, the others move one
//    seat on), into the OTHER of two buffers.  A thread's role and all its addresses are computed once, before the loops;
//  * a thread owns what it updates: the pairs cut A into 2 x 2 blocks (rows of pair I, columns of pair J) and V into
//    1 x 2 pieces, and a piece's new values depend on its old ones and the two rotations alone (before: every thread
//    read four old values per element it wrote, and a barrier separated reading from writing);
//  * two phases per step: `half` threads turn the pivot blocks into (c, s) -- out of two reciprocal square roots: with
//    d = a_qq - a_pp, b = 2 a_pq, 1/r = rsq(d^2 + b^2): cos 2θ = |d| / r, cos^2 θ = (1 + cos 2θ) / 2, 1/cos θ = rsq(cos^2 θ),
//    sin θ = |b| / (2 r cos θ) with the sign of b d (the inner rotation, |θ| <= π/4); one Newton step per rsq (v_rsq_f64
//    alone is good to 5e-8, one step to 4e-15: tools/microbench/rsq_accuracy.hip) and one step of re-normalisation
//    (c, s) *= 1.5 - 0.5 (c^2 + s^2), which squares what is left: orthogonal to 1e-28, the angle good to 1e-14 (an angle
//    error only slows convergence, quadratically little) -- barrier, everybody applies them, barrier.  (Every thread
//    computing its rotations itself saves a barrier and costs 30 instructions per rotation on twelve waves: 0.50 us.)
//  * convergence is tested directly after each sweep (every block thread looks at its four elements) instead of by a
//    whole sweep that finds nothing to rotate.
// The phantom index of an odd order is a zero row / column: its pivot a_pq is 0, so it is never rotated, wherever it sits.
// After a sweep (np - 1 steps) every index is back in its seat.
constexpr int JW = 32, JWS = JW + 1;  // max order and padded LDS stride of the small solver
constexpr int JT = 768;               // its threads

// Is a_pq worth a rotation?  Above the relative threshold tol sqrt(a_pp a_qq) (small eigenvalues keep their RELATIVE
// accuracy) -- and above the rounding noise of the update itself, 2 eps max(a_pp, a_qq): between a large and a small
// eigenvalue (1e9 and 0.2 in a rank-4 measurement matrix with noise) the relative threshold 1e-11 sits AT that noise
// (1.4e-7 against ~1e-7), such pairs flicker above it for ever, and a value of that size moves the eigenvalues by
// a_pq^2 / (λ_p - λ_q) ~ 1e-23 and the vectors by 1e-16.
__device__ __forceinline__ bool jacobi_needs(double app, double apq, double aqq, double tol2) {
  const double big = fmax(fabs(app), fabs(aqq)) * 4.5e-16;
  return apq * apq > fmax(tol2 * fabs(app * aqq), big * big);
}

// (c, s) that annihilate a_pq; false = the pair is left alone
__device__ __forceinline__ bool jacobi_rotation(double app, double apq, double aqq, double tol2, double &c, double &sn) {
  c = 1.0; sn = 0.0;
  if (!jacobi_needs(app, apq, aqq, tol2)) return false;
  const double d = aqq - app, b = 2.0 * apq;
  const double u = fma(d, d, b * b);
  double ir = __builtin_amdgcn_rsq(u);
  ir = ir * fma(-0.5 * u * ir, ir, 1.5);
  const double c2 = fma(0.5 * fabs(d), ir, 0.5);
  double ic = __builtin_amdgcn_rsq(c2);
  ic = ic * fma(-0.5 * c2 * ic, ic, 1.5);
  c = c2 * ic;
  sn = copysign(0.5 * fabs(b) * ir * ic, d == 0.0 ? b : b * d);  // (τ = 0: θ = π/4 with the sign of a_pq)
  const double f = fma(-0.5, fma(c, c, sn * sn), 1.5);
  c *= f; sn *= f;
  return true;
}

// LDS layout: A BLOCK-major, [row pair I][column pair J][2 x 2] with 16 blocks per block row, so a thread's block is 32
// contiguous bytes (two 16-byte loads, consecutive threads consecutive blocks); V row-major with stride 33, its threads
// running over the ROWS of one column pair (banks 2 i: conflict-free).
constexpr int JHB = JW / 2;  // block rows / columns of the block-major A
__device__ __forceinline__ int jblk(int r, int c) { return (((r >> 1) * JHB + (c >> 1)) << 2) + ((r & 1) << 1) + (c & 1); }

// Threads 0..255 own an A block (I = t >> 4, J = t & 15), threads 256.. a V piece (J = v >> 5, row i = v & 31); the launch
// brings 256 + 32 half threads.
__global__ __launch_bounds__(JT) void k_jacobi_small(double *__restrict__ Ag, double *__restrict__ Vg, int n, int max_sweeps,
                                                    double tol, int *__restrict__ sweeps_done) {
  __shared__ __attribute__((aligned(16))) double Ab[2][JHB * JHB * 4];
  __shared__ double Vb[2][JW * JWS];
  __shared__ double2 rot[JHB];  // (c, s) of this step's pairs
  __shared__ int s_rot[2];  // "a pair is still above the threshold" after the even / odd sweeps
  const int tid = threadIdx.x;
  const int np = (n + 1) & ~1, half = np / 2;
  // (a batch: block b solves problem b -- the dual depth iteration's 12 x 12 problems, one per image)
  Ag += (size_t)blockIdx.x * n * n; Vg += (size_t)blockIdx.x * n * n; sweeps_done += blockIdx.x;
  for (int e = tid; e < JW * JW; e += (int)blockDim.x) {
    const int i = e / JW, j = e % JW;
    Ab[0][jblk(i, j)] = (i < n && j < n) ? Ag[(size_t)i * n + j] : 0.0;  // (the phantom row / column of an odd order: zeros)
    Vb[0][i * JWS + j] = (i == j) ? 1.0 : 0.0;
  }
  if (tid == 0) s_rot[0] = s_rot[1] = 0;
  __syncthreads();
  // where the data at index x sits after a step: seats T[k] = 2k (top row), B[k] = 2k + 1 (bottom row), pairs (T[k], B[k]);
  // T[0] stays, B[0] -> T[1], T[k] -> T[k+1], T[half-1] -> B[half-1], B[k] -> B[k-1]
  auto seat = [&](int x) {
    if (x == 0 || half == 1) return x;
    if (x & 1) return x == 1 ? 2 : x - 2;
    return x == np - 2 ? np - 1 : x + 2;
  };
  const bool a_role = tid < JHB * JHB;
  // (A: the thread of a block BELOW the diagonal computes its mirror image above it -- the same inputs through the
  // same operations -- and writes the transposed result, so that A stays EXACTLY symmetric.  Computed independently,
  // the two copies of an element pick up different rounding (1e-10 absolute next to eigenvalues of 1e6), the seats
  // swap which copy a pivot reads, and a pair whose one copy is below the threshold and the other above it is then
  // never rotated and never accepted.)
  const int tI = tid >> 4, tJ = tid & (JHB - 1);
  const bool lower = a_role && tI > tJ, diag = a_role && tI == tJ;
  const int I = a_role ? min(tI, tJ) : 0, J = a_role ? max(tI, tJ) : (tid - JHB * JHB) >> 5, row = (tid - JHB * JHB) & (JW - 1);
  const bool active = a_role ? (tI < half && tJ < half) : (J < half && row < n);
  const int in0 = a_role ? (I * JHB + J) << 2 : row * JWS + 2 * J;  // the 2 x 2 block of A / the pair of V entries (adjacent)
  const int Pi = seat(2 * I), Qi = seat(2 * I + 1), Pj = seat(2 * J), Qj = seat(2 * J + 1);
  // where the four new values go: (Pi,Pj) (Pi,Qj) (Qi,Pj) (Qi,Qj), transposed for a thread below the diagonal
  const int o0 = a_role ? (lower ? jblk(Pj, Pi) : jblk(Pi, Pj)) : row * JWS + Pj;
  const int o1 = a_role ? (lower ? jblk(Qj, Pi) : jblk(Pi, Qj)) : row * JWS + Qj;
  const int o2 = lower ? jblk(Pj, Qi) : jblk(Qi, Pj), o3 = lower ? jblk(Qj, Qi) : jblk(Qi, Qj);
  const double tol2 = tol * tol;
  int sweep = 0, cur = 0;
  for (; sweep < max_sweeps; ++sweep) {
    for (int step = 0; step < np - 1; ++step, cur ^= 1) {
      const double *A = Ab[cur], *V = Vb[cur];
      double *An = Ab[cur ^ 1], *Vn = Vb[cur ^ 1];
      if (tid < half) {  // the rotation of pair `tid` from its pivot block
        const int pv = (tid * JHB + tid) << 2;
        const double2 p0 = *reinterpret_cast<const double2 *>(A + pv), p1 = *reinterpret_cast<const double2 *>(A + pv + 2);
        double c, sn;
        jacobi_rotation(p0.x, p0.y, p1.y, tol2, c, sn);
        rot[tid] = double2{c, sn};
      }
      __syncthreads();
      if (active) {
        const double2 rj = rot[J];
        if (a_role) {  // A' = J^T A J on the block (rows of pair I) x (columns of pair J)
          const double2 ri = rot[I];
          const double2 b0 = *reinterpret_cast<const double2 *>(A + in0), b1 = *reinterpret_cast<const double2 *>(A + in0 + 2);
          // rows first: row p <- c p - s q, row q <- s p + c q; then the same on the columns
          const double rpp = ri.x * b0.x - ri.y * b1.x, rpq = ri.x * b0.y - ri.y * b1.y, rqp = ri.y * b0.x + ri.x * b1.x, rqq = ri.y * b0.y + ri.x * b1.y;
          const double n1 = rj.y * rpp + rj.x * rpq;
          An[o0] = rj.x * rpp - rj.y * rpq;
          An[o1] = n1;
          An[o2] = diag ? n1 : rj.x * rqp - rj.y * rqq;  // (a diagonal block's lower element is the upper one's copy)
          An[o3] = rj.y * rqp + rj.x * rqq;
        } else {  // V' = V J on (row i) x (columns of pair J)
          const double vp = V[in0], vq = V[in0 + 1];
          Vn[o0] = rj.x * vp - rj.y * vq;
          Vn[o1] = rj.y * vp + rj.x * vq;
        }
      }
      __syncthreads();
    }
    // converged?  (what an empty sweep would find out in np - 1 steps: no pair left that the threshold would rotate;
    // after a sweep every index is back in its seat, so block (I, J) holds a[2I..2I+1][2J..2J+1])
    if (a_role && active && !lower) {
      const double *A = Ab[cur];
      const double2 b0 = *reinterpret_cast<const double2 *>(A + in0), b1 = *reinterpret_cast<const double2 *>(A + in0 + 2);
      const double dI0 = A[(I * JHB + I) << 2], dI1 = A[((I * JHB + I) << 2) + 3], dJ0 = A[(J * JHB + J) << 2], dJ1 = A[((J * JHB + J) << 2) + 3];
      auto big = [&](double apq, double app, double aqq) { return jacobi_needs(app, apq, aqq, tol2); };
      const bool any = I == J ? big(b0.y, dI0, dI1) : (big(b0.x, dI0, dJ0) || big(b0.y, dI0, dJ1) || big(b1.x, dI1, dJ0) || big(b1.y, dI1, dJ1));
      if (any) s_rot[sweep & 1] = 1;  // benign race: every writer stores 1
    }
    __syncthreads();
    const int more = s_rot[sweep & 1];
    if (tid == 0) s_rot[(sweep + 1) & 1] = 0;  // (the other flag: read last a sweep ago, written next a sweep from now, barriers between)
    if (more == 0) { ++sweep; break; }  // (uniform)
  }
  if (tid == 0) *sweeps_done = sweep;
  for (int e = tid; e < n * n; e += (int)blockDim.x) {  // (a whole number of sweeps: every index is back in its seat)
    const int i = e / n, jj = e % n;
    Ag[e] = Ab[cur][jblk(i, jj)];
    Vg[e] = Vb[cur][i * JWS + jj];
  }
}

// S[i][row] = sum_c Mr[c][i] (Wt[row][c] - mu[c]).  A wave takes 64 consecutive rows: the 64 x n
// tile is contiguous in memory, so it is loaded with fully coalesced accesses into a padded LDS
// tile (odd stride -> conflict-free column reads), then each lane reduces its own row.
constexpr int PC = 64;  // columns per LDS chunk
template <typename T>
__global__ __launch_bounds__(256) void k_project(const T *__restrict__ Wt, long long n_rows, int n, int r,
                                                 const double *__restrict__ Mr4 /*[n][4], zero padded beyond r*/, const double *__restrict__ mu,
                                                 T *__restrict__ S) {
  extern __shared__ double sm[];  // (4 n doubles unused since the basis is read by scalar loads), mu [n], then per-wave tiles of T
  double *smu = sm + 4 * (size_t)n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = min(n, PC), ldt = nc | 1;  // odd stride
  T *tile = reinterpret_cast<T *>(smu + n) + (size_t)wave * 64 * ldt;
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
        // (batches of eight independent loads, then their LDS stores: one load and one store per trip waited for every load
        // before issuing the next: 0.43 -> 0.26 ms at 5 M x 24 fp64.  A streaming MFMA form like k_rotate_rows, its 32 x r result
        // transposed through LDS, was slower than this kernel: 0.39 ms, fp32 0.18 against 0.16)
        constexpr int U = 8;
        if (sizeof(T) == 4 && (n & 3) == 0) {  // 16 B per lane; a float4 never straddles a row
          const int n4 = n >> 2, total4 = rows_here * n4;
          const float4 *src4 = reinterpret_cast<const float4 *>(src);
          for (int q0 = lane; q0 < total4; q0 += 64 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src4[min(q0 + 64 * u, total4 - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int q = q0 + 64 * u, rr = q / n4;
              if (q < total4) {
                T *d = tile + rr * ldt + 4 * (q - rr * n4);
                d[0] = (T)v[u].x; d[1] = (T)v[u].y; d[2] = (T)v[u].z; d[3] = (T)v[u].w;
              }
            }
          }
        } else {
          const int total = rows_here * n;
          for (int q0 = lane; q0 < total; q0 += 64 * U) {
            T v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[min(q0 + 64 * u, total - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int q = q0 + 64 * u, rr = q / n;
              if (q < total) tile[rr * ldt + (q - rr * n)] = v[u];
            }
          }
        }
      } else {
        for (int q = lane; q < rows_here * w; q += 64) tile[(q / w) * ldt + (q % w)] = Wt[(base + q / w) * n + c0 + (q % w)];
      }
      wave_sync();
      if (lane < rows_here) {
        // the four basis components of a column are the same for every lane: SCALAR loads of the padded [n][4] table (a
        // lane walking its row read them as 4 LDS broadcasts per column -- 96 of the 120 LDS reads per row at 24 columns)
        const T *tr = tile + lane * ldt;
        typedef const double __attribute__((address_space(4))) *cptr_t;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        cptr_t mcs = (cptr_t)(Mr4 + 4 * (size_t)c0);
#pragma clang diagnostic pop
        int c = 0;
        for (; c + 4 <= w; c += 4) {
          double mv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) mv[e] = mcs[4 * c + e];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double x = (double)tr[c + u] - smu[c0 + c + u];
            acc[0] += x * mv[4 * u]; acc[1] += x * mv[4 * u + 1]; acc[2] += x * mv[4 * u + 2]; acc[3] += x * mv[4 * u + 3];
          }
        }
        for (; c < w; ++c) {
          const double x = (double)tr[c] - smu[c0 + c];
          acc[0] += x * mcs[4 * c]; acc[1] += x * mcs[4 * c + 1]; acc[2] += x * mcs[4 * c + 2]; acc[3] += x * mcs[4 * c + 3];
        }
      }
      wave_sync();
    }
    if (lane < rows_here)
      for (int i = 0; i < r; ++i) S[(size_t)i * n_rows + base + lane] = (T)acc[i];
  }
}

// ---- 64 rows per wave through LDS: the rows [base, base + rows_here) x n of a row-major matrix are one contiguous byte
// range -- moved with coalesced accesses to / from a padded LDS tile (odd stride: a lane walking its own row is
// conflict-free), so that a thread-per-row kernel neither reads nor writes memory at a row stride per lane
// (k_scale_rows as one lane per row in global memory: ~4 ms of a 10 ms depth iteration at 5 M x 24 fp64).
// (batches of eight independent loads, then their eight stores: written as one load and one store per trip the loop waited
// for every load before it issued the next -- 24 round trips for a 64 x 24 tile, and the "tiled" kernels ran at 1-2.5 TB/s)
template <typename T>
__device__ __forceinline__ void tile_load(const T *__restrict__ src, long long base, int rows_here, int n, T *tile, int ldt, int lane) {
  const T *s = src + base * n;
  const int total = rows_here * n;
  constexpr int U = 8;
  for (int q0 = lane; q0 < total; q0 += 64 * U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[min(q0 + 64 * u, total - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + 64 * u, rr = q / n;
      if (q < total) tile[rr * ldt + (q - rr * n)] = v[u];
    }
  }
}
template <typename T>
__device__ __forceinline__ void tile_store(T *__restrict__ dst, long long base, int rows_here, int n, const T *tile, int ldt, int lane) {
  T *d = dst + base * n;
  const int total = rows_here * n;
  constexpr int U = 8;
  for (int q0 = lane; q0 < total; q0 += 64 * U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = min(q0 + 64 * u, total - 1), rr = q / n;
      v[u] = tile[rr * ldt + (q - rr * n)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (q0 + 64 * u < total) d[q0 + 64 * u] = v[u];
  }
}
constexpr size_t TILE_MAX_BYTES = 120 * 1024;  // LDS the four tiles of a block may take (plus the small operand tables: < 160 KiB)
template <typename T>
inline bool tile_fits(int cols) { return sizeof(T) * 4 * 64 * (size_t)(cols | 1) <= TILE_MAX_BYTES; }

// ---- depth-weighted matrix from a resident base (mvsvd_run_scaled): W[a][j] = X[a][j] z[a][j / group] s,
// s = 1 / |row a of X o z| (norm 1: every row to unit length, ref perspective_camera_calibration.py:86-87) or
// s = 1 / sum over rows and the group's columns of (X o z)^2 (norm 2: every column group -- image -- divided by its
// SQUARED Frobenius norm, ref :170-172).  One thread per row; the group sums of norm 2 are per-block partials added in
// fixed order (k_group_scale).
template <typename T>
__global__ __launch_bounds__(256) void k_group_sumsq(const T *__restrict__ X, const T *__restrict__ z, long long n_rows, int n,
                                                     int group, double *__restrict__ part /*[gridDim.x][n / group]*/) {
  const int ng = n / group;
  // thread t of the block owns group t % ng of the rows t / ng, t / ng + rows_per_pass, ...: every group's partial is
  // accumulated by a fixed set of threads in a fixed order, then summed over those threads in thread order
  if (ng > (int)blockDim.x) {  // more groups than threads: a thread owns the groups t, t + 256, ... of the block's rows (blockIdx.x, + gridDim.x, ...)
    for (int g = threadIdx.x; g < ng; g += blockDim.x) {
      double acc = 0.0;
      for (long long a = blockIdx.x; a < n_rows; a += gridDim.x) {
        const double zz = (double)z[a * ng + g];
        for (int c = 0; c < group; ++c) {
          const double w = (double)X[a * n + g * group + c] * zz;
          acc += w * w;
        }
      }
      part[(size_t)blockIdx.x * ng + g] = acc;
    }
    return;
  }
  const int rows_per_pass = blockDim.x / ng;
  const int g = threadIdx.x % ng, rloc = threadIdx.x / ng;
  double acc = 0.0;
  if (rloc < rows_per_pass)
    for (long long a = (long long)blockIdx.x * rows_per_pass + rloc; a < n_rows; a += (long long)gridDim.x * rows_per_pass) {
      const double zz = (double)z[a * ng + g];
      for (int c = 0; c < group; ++c) {
        const double w = (double)X[a * n + g * group + c] * zz;
        acc += w * w;
      }
    }
  __shared__ double s_acc[256];
  s_acc[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < ng) {
    double t = 0.0;
    for (int r = 0; r < rows_per_pass; ++r) t += s_acc[r * ng + threadIdx.x];
    part[(size_t)blockIdx.x * ng + threadIdx.x] = t;
  }
}

// fixed-order sum of `count` values spaced `stride` apart by ONE WAVE: lane l adds the values l, l + 64, ... in order, then a
// fixed tree over the lanes (the same result whatever the launch geometry; a single thread walking 2048 partials took 0.1-0.5 ms)
__device__ __forceinline__ double wave_strided_sum(const double *__restrict__ p, int count, size_t stride, int lane) {
  double t = 0.0;
  for (int b = lane; b < count; b += 64) t += p[(size_t)b * stride];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  return t;  // (valid in lane 0)
}

__global__ void k_group_scale(const double *__restrict__ part, int blocks, int ng, double *__restrict__ cs) {
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= ng) return;
  const double t = wave_strided_sum(part + g, blocks, (size_t)ng, lane);
  if (lane == 0) cs[g] = 1.0 / t;
}

// mvsvd_load_images: one image's coordinates [n_rows][2] (float or double, as the caller holds them) into its two columns of the
// measurement matrix W^T [n_rows][2 m]   (ref lib/affine_camera_calibration.py:224-240: np.hstack(data_list))
template <typename S, typename T>
__global__ __launch_bounds__(256) void k_image_cols(const S *__restrict__ xy, long long n_rows, int n, int col, T *__restrict__ W) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  T *o = W + i * n + col;
  o[0] = (T)xy[2 * i];
  o[1] = (T)xy[2 * i + 1];
}

// mvsvd_load_base_images: one image's pixel coordinates [n_rows][2] (doubles, as the caller holds them) into its three columns
// of the base, homogeneous: (x / f0, y / f0, 1)   (ref lib/perspective_camera_calibration.py:34-40)
template <typename T>
__global__ __launch_bounds__(256) void k_base_image(const double2 *__restrict__ xy, long long n_rows, int n, int col, double f0, T *__restrict__ X) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  const double2 p = xy[i];
  T *o = X + i * n + col;
  o[0] = (T)(p.x / f0);
  o[1] = (T)(p.y / f0);
  o[2] = (T)1;
}

// Rows too long for the LDS tiles (more than ~57 fp64 / 117 fp32 columns + depths): the lanes of a wave run ACROSS the column
// groups of a row -- P = ng rounded up to a power of two lanes per row, 64 / P rows per pass (beyond 64 groups: one row, the lanes
// striding over its groups) -- so that a pass reads and writes one contiguous range; the row's sum of squares (norm 1) is a
// fixed tree over its lanes.  (Rounds 3-5 had a lane per row walking the 720 bytes of its own row here: 3.9 ms at 1 M x 90 fp64,
// more than the factorisation it feeds; this form 0.6.)
template <typename T>
__global__ __launch_bounds__(256) void k_scale_rows_wide(const T *__restrict__ X, const T *__restrict__ z, long long n_rows, int n,
                                                         int group, int norm, const double *__restrict__ cs, T *__restrict__ W) {
  const int ng = n / group, lane = threadIdx.x & 63;
  int P = 1;
  while (P < ng && P < 64) P <<= 1;
  const int R = 64 / P, rr = lane / P, g0 = lane - rr * P;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long a0 = wave * R; a0 < n_rows; a0 += n_waves * R) {
    const long long a = a0 + rr;
    const bool row_ok = a < n_rows;
    double ss = 0.0;
    if (norm == 1) {
      for (int g = g0; g < ng; g += P) {
        if (row_ok) {
          const double zz = (double)z[a * ng + g];
          for (int c = 0; c < group; ++c) {
            const double w = (double)X[a * n + g * group + c] * zz;
            ss += w * w;
          }
        }
      }
      for (int off = P >> 1; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);  // (within the row's P lanes; every lane ends with the row's sum)
    }
    const double rs = norm == 1 ? 1.0 / sqrt(ss) : 1.0;
    if (row_ok)
      for (int g = g0; g < ng; g += P) {
        const double f = (double)z[a * ng + g] * (norm == 2 ? cs[g] : rs);
        for (int c = 0; c < group; ++c) W[a * n + g * group + c] = (T)((double)X[a * n + g * group + c] * f);
      }
  }
}


// the same through LDS tiles (one wave = 64 rows; columns [X row | z row] side by side in the tile)
template <typename T>
__global__ __launch_bounds__(256) void k_scale_rows_tiled(const T *__restrict__ X, const T *__restrict__ z, long long n_rows, int n,
                                                          int group, int norm, const double *__restrict__ cs, T *__restrict__ W) {
  extern __shared__ double sm_raw[];
  const int ng = n / group, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ldt = (n + ng) | 1;
  T *tile = reinterpret_cast<T *>(sm_raw) + (size_t)wave * 64 * ldt;
  for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < n_rows; base += (long long)gridDim.x * 256) {
    const int rows_here = (int)min<long long>(64, n_rows - base);
    tile_load(X, base, rows_here, n, tile, ldt, lane);
    tile_load(z, base, rows_here, ng, tile + n, ldt, lane);
    wave_sync();
    if (lane < rows_here) {
      T *xr = tile + lane * ldt;
      const T *zr = xr + n;
      double rs = 1.0;
      if (norm == 1) {
        double ss = 0.0;
        for (int g = 0; g < ng; ++g) {
          const double zz = (double)zr[g];
          for (int c = 0; c < group; ++c) {
            const double w = (double)xr[g * group + c] * zz;
            ss += w * w;
          }
        }
        rs = 1.0 / sqrt(ss);
      }
      for (int g = 0; g < ng; ++g) {
        const double f = (double)zr[g] * (norm == 2 ? cs[g] : rs);
        for (int c = 0; c < group; ++c) xr[g * group + c] = (T)((double)xr[g * group + c] * f);
      }
    }
    wave_sync();
    tile_store(W, base, rows_here, n, tile, ldt, lane);
    wave_sync();
  }
}

// ---- projective-depth iteration on the device (mvsvd_depth_step; ref lib/perspective_camera_calibration.py:93-129
// primary, :182-224 dual).  After the factorisation of the re-weighted matrix the workspace holds M = U[:, :4]
// (dMr, [3m][4]) and S = diag(sigma) Vt[:4] (dS, [4][rows]); the depth update reads them, the resident observations
// X ([rows][3m]) and writes the new depths z ([rows][m]) -- nothing crosses PCIe but the reprojection error.
// The eigenproblems in their low-rank form (oracle/depth_oracle.py restates them in NumPy):
//   primary  per point a: C[k][i] = (x_ak . u_ki) / |x_ak| (m x 4); dominant eigenvector v of the 4 x 4 companion
//            C^T C by cyclic Jacobi in registers; xi = C v / |C v|  (= the dominant eigenvector of the reference's
//            m x m matrix C C^T, :99-118)
//   dual     per image k: Z[a] = V4[a] (x) x_ak / |x_ak| (rows x 12), V4 = right singular vectors; the 12 x 12
//            companion Z^T Z = sum_a (v v^T) (x) (x^ x^^T) has 10 x 6 distinct entries, summed over the rows per block
//            in row order and over the blocks in block order (no atomics); batched Jacobi; xi = Z w / sqrt(lambda)
//            (= the dominant eigenvector of the reference's N x N matrix, :188-213, in O(N) memory)
// Signs: xi of a point is flipped when its sum is negative (ref :121 / :217); in the dual form every image's vector is
// first oriented to a non-negative sum (the reference inherits LAPACK's eigenvector sign there: projectively equivalent).
// Reprojection error (:43-58) from the same M, S, per-thread sums in row order, fixed tree, fixed block order.

// 1 / sqrt(u) and 1 / u from the hardware seeds and two Newton steps each (full double precision to an ulp or two): the
// per-point kernels below take a root and up to four quotients per observation, and the library sqrt / divide expansions
// (~35 instructions each) were most of their 1.5 ms at 5 M points x 8 images
__device__ __forceinline__ double fast_rsqrt(double u) {
  double r = __builtin_amdgcn_rsq(u);
  r = r * fma(-0.5 * u * r, r, 1.5);
  return r * fma(-0.5 * u * r, r, 1.5);
}
__device__ __forceinline__ double fast_rcp(double u) {
  double r = __builtin_amdgcn_rcp(u);
  r = fma(fma(-u, r, 1.0), r, r);
  return fma(fma(-u, r, 1.0), r, r);
}

// dominant eigenvector of the symmetric 4 x 4 matrix g (upper triangle, row-major 10 values): cyclic Jacobi with
// the rotations of the small solver above (same (c, s) convention), everything in registers
__device__ __forceinline__ void dominant_eigvec4(const double (&g)[10], double (&v)[4]) {
  double A[4][4], V[4][4];
  {
    int e = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = i; j < 4; ++j, ++e) A[i][j] = A[j][i] = g[e];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  const double tol2 = 1e-30;
  for (int sweep = 0; sweep < 24; ++sweep) {
    bool any = false;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double c, sn;
        if (!jacobi_rotation(A[p][p], A[p][q], A[q][q], tol2, c, sn)) continue;
        any = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // rows p, q
          const double ap = A[p][j], aq = A[q][j];
          A[p][j] = c * ap - sn * aq;
          A[q][j] = sn * ap + c * aq;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // columns p, q (and the eigenvectors)
          const double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - sn * aq;
          A[i][q] = sn * ap + c * aq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - sn * vq;
          V[i][q] = sn * vp + c * vq;
        }
        const double sym = 0.5 * (A[p][q] + A[q][p]);  // (annihilated up to rounding; keep the two copies equal)
        A[p][q] = A[q][p] = sym;
      }
    if (!any) break;
  }
  // (the largest diagonal entry tracked in a scalar: `A[best][best]` with a run-time `best` put the whole of A into scratch
  // memory -- 144 bytes per lane, stored and reloaded around every rotation of the loop above)
  int best = 0;
  double bestv = A[0][0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (A[i][i] > bestv) { best = i; bestv = A[i][i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = best == 0 ? V[i][0] : (best == 1 ? V[i][1] : (best == 2 ? V[i][2] : V[i][3]));
}

// sum of a value over the block's threads in a fixed tree -> part[blockIdx.x]
__device__ __forceinline__ void block_sum_to(double v, double *__restrict__ part) {
  __shared__ double s_red[256];
  s_red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = s_red[0];
}

template <typename T>
__global__ __launch_bounds__(256) void k_fill(T *__restrict__ p, long long n, T v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// squared reprojection error of observation x (3) under P = U4_k (3 x 4) and the point's S column
__device__ __forceinline__ double reproj_err2(const double *u /*[3][4]*/, const double (&s)[4], double x0, double x1, double x2) {
  const double p0 = u[0] * s[0] + u[1] * s[1] + u[2] * s[2] + u[3] * s[3];
  const double p1 = u[4] * s[0] + u[5] * s[1] + u[6] * s[2] + u[7] * s[3];
  const double p2 = u[8] * s[0] + u[9] * s[1] + u[10] * s[2] + u[11] * s[3];
  const double rp = fast_rcp(p2);
  const double d0 = x0 - p0 * rp, d1 = x1 - p1 * rp, d2 = x2 - p2 * rp;
  return d0 * d0 + d1 * d1 + d2 * d2;
}

template <typename T, bool TILED>
__global__ __launch_bounds__(256) void k_depth_primary(const T *__restrict__ X, const double *__restrict__ Mr, const T *__restrict__ S,
                                                       long long n_rows, int m, T *__restrict__ z, double *__restrict__ Epart) {
  extern __shared__ double sU[];  // [3m][4], then (TILED) one tile of 64 x [3m values | m depths] per wave
  for (int q = threadIdx.x; q < 12 * m; q += blockDim.x) sU[q] = Mr[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = 3 * m, ldt = (n + m) | 1;
  T *tile = reinterpret_cast<T *>(sU + 12 * (size_t)m) + (size_t)wave * 64 * ldt;
  double esum = 0.0;
  for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < n_rows; base += (long long)gridDim.x * 256) {
    const int rows_here = (int)min<long long>(64, n_rows - base);
    if (TILED) {
      tile_load(X, base, rows_here, n, tile, ldt, lane);
      wave_sync();
    }
    if (lane < rows_here) {
      const long long a = base + lane;
      const T *xr = TILED ? tile + lane * ldt : X + a * n;
      T *zr = TILED ? tile + lane * ldt + n : z + a * m;
      double s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] = (double)S[(size_t)i * n_rows + a];
      double g[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < m; ++k) {
        const double x0 = (double)xr[3 * k], x1 = (double)xr[3 * k + 1], x2 = (double)xr[3 * k + 2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double *u = sU + 12 * k;
        double c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = (x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i]) * inv;
        int e = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = i; j < 4; ++j, ++e) g[e] = fma(c[i], c[j], g[e]);
        esum += reproj_err2(u, s, x0, x1, x2);
      }
      double v[4];
      dominant_eigvec4(g, v);
      double nrm2 = 0.0, sum = 0.0;
      for (int k = 0; k < m; ++k) {  // xi_k = C[k] . v (unnormalised); z <- xi_k / |x_ak| for now
        const double x0 = (double)xr[3 * k], x1 = (double)xr[3 * k + 1], x2 = (double)xr[3 * k + 2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double *u = sU + 12 * k;
        double xi = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) xi = fma((x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i]) * inv, v[i], xi);
        nrm2 = fma(xi, xi, nrm2);
        sum += xi;
        zr[k] = (T)(xi * inv);
      }
      const double sc = (sum < 0.0 ? -1.0 : 1.0) * fast_rsqrt(nrm2);  // unit length, non-negative sum (ref :118, :121)
      for (int k = 0; k < m; ++k) zr[k] = (T)((double)zr[k] * sc);
    }
    if (TILED) {
      wave_sync();
      tile_store(z, base, rows_here, m, tile + n, ldt, lane);
      wave_sync();
    }
  }
  block_sum_to(esum, Epart);
}

// Rows too long for the LDS tiles (fp64 from ~15 images on): the lanes of a wave run ACROSS the images of a point -- P = m rounded
// up to a power of two (at most 64) lanes per point, 64 / P points per pass -- so that a pass reads one contiguous range of X
// and writes one of z; the sums over a point's images (the 4 x 4 companion, the norm and the sign of the new depths) are fixed
// trees over its lanes, and every lane of a point finds the companion's eigenvector itself.  (The lane-per-point form walks
// 24 m bytes of its own row per lane: 1.6 ms at 1 M points x 30 images against 0.4 for this one.)
template <typename T>
__global__ __launch_bounds__(256) void k_depth_primary_wide(const T *__restrict__ X, const double *__restrict__ Mr, const T *__restrict__ S,
                                                            long long n_rows, int m, T *__restrict__ z, double *__restrict__ Epart) {
  extern __shared__ double sU[];  // [m][3][4], then per wave [64 points][10 companion sums | 4 eigenvector components]
  for (int q = threadIdx.x; q < 12 * m; q += blockDim.x) sU[q] = Mr[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = 3 * m;
  double *sg = sU + 12 * (size_t)m + (size_t)wv * 64 * 14;
  int P = 1;
  while (P < m && P < 64) P <<= 1;
  const int R = 64 / P, rr = lane / P, g0 = lane - rr * P;
  const long long wave = (long long)blockIdx.x * 4 + wv, n_waves = (long long)gridDim.x * 4;
  double esum = 0.0;
  // a batch = 64 points: (1) their companions, R points per pass, the lanes across the images; (2) the 64 eigenvectors, a lane per
  // point (the 4 x 4 Jacobi is ~1000 instructions: run once per pass of R points it was most of the kernel); (3) the new depths
  for (long long b0 = wave * 64; b0 < n_rows; b0 += n_waves * 64) {
    for (int sub = 0; sub < 64; sub += R) {
      const long long a = min(b0 + sub + rr, n_rows - 1);  // (a clamped row repeats the last one: computed, never stored)
      const bool row_ok = b0 + sub + rr < n_rows;
      const T *xr = X + a * n;
      double s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] = (double)S[(size_t)i * n_rows + a];
      double g[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = g0; k < m; k += P) {
        const double x0 = (double)xr[3 * k], x1 = (double)xr[3 * k + 1], x2 = (double)xr[3 * k + 2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double *u = sU + 12 * k;
        double c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = (x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i]) * inv;
        int e = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = i; j < 4; ++j, ++e) g[e] = fma(c[i], c[j], g[e]);
        if (row_ok) esum += reproj_err2(u, s, x0, x1, x2);
      }
#pragma unroll
      for (int e = 0; e < 10; ++e) {
        for (int off = P >> 1; off > 0; off >>= 1) g[e] += __shfl_xor(g[e], off, 64);
        if (g0 == 0) sg[(sub + rr) * 14 + e] = g[e];
      }
    }
    wave_sync();
    {
      double g[10], v[4];
#pragma unroll
      for (int e = 0; e < 10; ++e) g[e] = sg[lane * 14 + e];
      dominant_eigvec4(g, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) sg[lane * 14 + 10 + i] = v[i];
    }
    wave_sync();
    for (int sub = 0; sub < 64; sub += R) {
      const long long a = min(b0 + sub + rr, n_rows - 1);
      const bool row_ok = b0 + sub + rr < n_rows;
      const T *xr = X + a * n;
      double v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = sg[(sub + rr) * 14 + 10 + i];
      double nrm2 = 0.0, sum = 0.0;
      for (int k = g0; k < m; k += P) {  // xi_k = C[k] . v (unnormalised)
        const double x0 = (double)xr[3 * k], x1 = (double)xr[3 * k + 1], x2 = (double)xr[3 * k + 2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double *u = sU + 12 * k;
        double xi = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) xi = fma((x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i]) * inv, v[i], xi);
        nrm2 = fma(xi, xi, nrm2);
        sum += xi;
        if (row_ok) z[a * m + k] = (T)(xi * inv);
      }
      for (int off = P >> 1; off > 0; off >>= 1) { nrm2 += __shfl_xor(nrm2, off, 64); sum += __shfl_xor(sum, off, 64); }
      const double sc = (sum < 0.0 ? -1.0 : 1.0) * fast_rsqrt(nrm2);  // unit length, non-negative sum (ref :118, :121)
      if (row_ok)
        for (int k = g0; k < m; k += P) z[a * m + k] = (T)((double)z[a * m + k] * sc);  // (this lane's own stores)
    }
    wave_sync();
  }
  block_sum_to(esum, Epart);
}

// the dual update's last pass in the same geometry: xi[a][k] = Z_k[a] . w_k, the point's sign rule, z = xi / |x|, reprojection error
template <typename T>
__global__ __launch_bounds__(256) void k_dual_apply_wide(const T *__restrict__ X, const T *__restrict__ S, double is0, double is1, double is2,
                                                         double is3, const double *__restrict__ Mr, const double *__restrict__ w12,
                                                         long long n_rows, int m, T *__restrict__ z, double *__restrict__ Epart) {
  extern __shared__ double sm[];  // U4 [m][3][4], w [m][12]
  double *sU = sm, *sW = sm + 12 * (size_t)m;
  for (int q = threadIdx.x; q < 12 * m; q += blockDim.x) { sU[q] = Mr[q]; sW[q] = w12[q]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, n = 3 * m;
  int P = 1;
  while (P < m && P < 64) P <<= 1;
  const int R = 64 / P, rr = lane / P, g0 = lane - rr * P;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  double esum = 0.0;
  for (long long a0 = wave * R; a0 < n_rows; a0 += n_waves * R) {
    const long long a = min(a0 + rr, n_rows - 1);
    const bool row_ok = a0 + rr < n_rows;
    const T *xr = X + a * n;
    double s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = (double)S[(size_t)i * n_rows + a];
    const double v4[4] = {s[0] * is0, s[1] * is1, s[2] * is2, s[3] * is3};
    double sum = 0.0;
    for (int k = g0; k < m; k += P) {
      const double x0 = (double)xr[3 * k], x1 = (double)xr[3 * k + 1], x2 = (double)xr[3 * k + 2];
      const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
      const double h0 = x0 * inv, h1 = x1 * inv, h2 = x2 * inv;
      const double *w = sW + 12 * k;
      double xi = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) xi = fma(v4[i], h0 * w[3 * i] + h1 * w[3 * i + 1] + h2 * w[3 * i + 2], xi);
      sum += xi;
      if (row_ok) {
        z[a * m + k] = (T)(xi * inv);
        esum += reproj_err2(sU + 12 * k, s, x0, x1, x2);
      }
    }
    for (int off = P >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (sum < 0.0 && row_ok)
      for (int k = g0; k < m; k += P) z[a * m + k] = (T)(-(double)z[a * m + k]);  // ref :217
  }
  block_sum_to(esum, Epart);
}

// dual, pass 1: per block and image the 60 distinct entries of sum_a (v v^T) (x) (x^ x^^T) and the 12 column sums of Z, laid out
// as 14 "tasks" of 6 values per image for k_dual_reduce: task t < 10 m = image t / 10, pair (i <= j) of V4 components t % 10 -> the 6
// sums over (c <= d); task 10 m <= t < 14 m = image, component i -> 3 column sums.
// A thread owns ONE image (all 72 sums of it in registers) and the rows rr = sub, sub + nsplit, ... of the staged batch, with
// nsplit = 256 / min(m, 256) thread groups per block and a partial per group (summed in group order by k_dual_reduce); more than
// 256 images go in chunks of 256, the rows staged again per chunk.  Per row a thread reads its image's normalised observation
// (3 values) and the row's v4 (4, broadcast) for 82 multiply-adds -- rounds 4-5 had a thread per TASK reading five values for
// seven multiply-adds, and the kernel ran at the LDS read rate: 1.76 ms at 1 M points x 30 images, this form 0.5.
__constant__ int c_pair_i[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3};
__constant__ int c_pair_j[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
constexpr int DG_ROWS_MAX = 128;  // rows staged per pass of k_dual_gram (fewer when 3 m columns of them do not fit the LDS: dual_gram_rows)
inline int dual_gram_split(int m) { return std::min(16, 256 / std::min(m, 256)); }  // thread groups (= partials) per block
template <typename T>
__global__ __launch_bounds__(256) void k_dual_gram(const T *__restrict__ X, const T *__restrict__ S, double is0, double is1, double is2,
                                                   double is3, long long n_rows, int m, long long rows_per_block, int DG_ROWS,
                                                   double *__restrict__ part /*[blocks][nsplit][14 m][6]*/) {
  // staged per pass: x^ [DG_ROWS][3m] (stride 3m | 1) and v4 [DG_ROWS][4]: the normalisation happens once per (row, image)
  extern __shared__ double sg[];
  const int n = 3 * m, ldx = n | 1;
  double *sx = sg, *sv = sg + (size_t)DG_ROWS * ldx;
  const long long a0 = (long long)blockIdx.x * rows_per_block, a1 = min(n_rows, a0 + rows_per_block);
  const double isg[4] = {is0, is1, is2, is3};
  const int tpi = min(m, 256), nsplit = min(16, 256 / tpi), sub = (int)threadIdx.x / tpi, kl = (int)threadIdx.x - sub * tpi;
  for (int k0 = 0; k0 < m; k0 += tpi) {
    const int k = k0 + kl;
    const bool mine = sub < nsplit && k < m;
    double acc[10][6], col[4][3];
#pragma unroll
    for (int p = 0; p < 10; ++p)
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[p][q] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) col[i][c] = 0.0;
    for (long long r0 = a0; r0 < a1; r0 += DG_ROWS) {
      const int nr = (int)min<long long>(DG_ROWS, a1 - r0);
      __syncthreads();
      for (int e = threadIdx.x; e < nr * m; e += 256) {  // (row, image): normalise
        const int rr = e / m, kk = e - rr * m;
        const T *xr = X + (r0 + rr) * n + 3 * kk;
        const double x0 = (double)xr[0], x1 = (double)xr[1], x2 = (double)xr[2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        double *d = sx + (size_t)rr * ldx + 3 * kk;
        d[0] = x0 * inv; d[1] = x1 * inv; d[2] = x2 * inv;
      }
      for (int e = threadIdx.x; e < nr * 4; e += 256) sv[e] = (double)S[(size_t)(e & 3) * n_rows + r0 + (e >> 2)] * isg[e & 3];
      __syncthreads();
      if (mine)
        for (int rr = sub; rr < nr; rr += nsplit) {
          const double *hp = sx + (size_t)rr * ldx + 3 * k;
          const double h[3] = {hp[0], hp[1], hp[2]};
          const double v[4] = {sv[4 * rr], sv[4 * rr + 1], sv[4 * rr + 2], sv[4 * rr + 3]};
          const double hh[6] = {h[0] * h[0], h[0] * h[1], h[0] * h[2], h[1] * h[1], h[1] * h[2], h[2] * h[2]};
          int p = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) col[i][c] = fma(v[i], h[c], col[i][c]);
#pragma unroll
            for (int j = i; j < 4; ++j, ++p) {
              const double pq = v[i] * v[j];
#pragma unroll
              for (int q = 0; q < 6; ++q) acc[p][q] = fma(pq, hh[q], acc[p][q]);
            }
          }
        }
    }
    if (mine) {
      double *o = part + (size_t)(blockIdx.x * nsplit + sub) * 14 * m * 6;
#pragma unroll
      for (int p = 0; p < 10; ++p)
#pragma unroll
        for (int q = 0; q < 6; ++q) o[((size_t)10 * k + p) * 6 + q] = acc[p][q];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[((size_t)10 * m + 4 * k + i) * 6 + c] = col[i][c];
#pragma unroll
        for (int c = 3; c < 6; ++c) o[((size_t)10 * m + 4 * k + i) * 6 + c] = 0.0;
      }
    }
  }
}

// dual, pass 2: block partials summed in block order -> G12[k][12][12] (symmetric, index (i, c) = 3 i + c) and colsum[k][12]
__global__ void k_dual_reduce(const double *__restrict__ part, int blocks, int m, double *__restrict__ G12, double *__restrict__ colsum) {
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // one wave per sum
  if (t >= 14 * m * 6) return;
  const int task = t / 6, q = t % 6;
  const double v = wave_strided_sum(part + (size_t)task * 6 + q, blocks, (size_t)14 * m * 6, lane);
  if (lane) return;
  if (task < 10 * m) {
    const int k = task / 10, i = c_pair_i[task % 10], j = c_pair_j[task % 10];
    const int c = q < 3 ? 0 : (q < 5 ? 1 : 2), d = q < 3 ? q : (q < 5 ? q - 2 : 2);
    double *G = G12 + (size_t)k * 144;
    G[(3 * i + c) * 12 + 3 * j + d] = v; G[(3 * i + d) * 12 + 3 * j + c] = v;
    G[(3 * j + c) * 12 + 3 * i + d] = v; G[(3 * j + d) * 12 + 3 * i + c] = v;
  } else if (q < 3) {
    const int k = (task - 10 * m) / 4, i = (task - 10 * m) % 4;
    colsum[(size_t)k * 12 + 3 * i + q] = v;
  }
}

// dual, pass 3 (after the batched Jacobi: eigenvalues on the diagonal of G12[k], eigenvectors in the columns of V12[k]):
// w_k = +- v_max / sqrt(lambda_max), oriented so that the image's depth vector Z w has a non-negative sum
__global__ void k_dual_vec(const double *__restrict__ G12, const double *__restrict__ V12, const double *__restrict__ colsum, int m,
                           double *__restrict__ w12, int *__restrict__ flag) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const double *G = G12 + (size_t)k * 144, *V = V12 + (size_t)k * 144;
  int best = 0;
  for (int i = 1; i < 12; ++i)
    if (G[i * 13] > G[best * 13]) best = i;
  const double lam = G[best * 13];
  if (!(lam > 0.0)) { atomicOr(flag, 1); return; }
  double dot = 0.0;
  for (int i = 0; i < 12; ++i) dot += colsum[(size_t)k * 12 + i] * V[i * 12 + best];
  const double sc = (dot < 0.0 ? -1.0 : 1.0) / sqrt(lam);
  for (int i = 0; i < 12; ++i) w12[(size_t)k * 12 + i] = sc * V[i * 12 + best];
}

// dual, pass 4: xi[a][k] = Z_k[a] . w_k, the row's sign rule, z = xi / |x|, reprojection error
template <typename T, bool TILED>
__global__ __launch_bounds__(256) void k_dual_apply(const T *__restrict__ X, const T *__restrict__ S, double is0, double is1, double is2,
                                                    double is3, const double *__restrict__ Mr, const double *__restrict__ w12,
                                                    long long n_rows, int m, T *__restrict__ z, double *__restrict__ Epart) {
  extern __shared__ double sm[];  // U4 [3m][4], w [m][12], then (TILED) one tile of 64 x [3m values | m depths] per wave
  double *sU = sm, *sW = sm + 12 * (size_t)m;
  for (int q = threadIdx.x; q < 12 * m; q += blockDim.x) { sU[q] = Mr[q]; sW[q] = w12[q]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = 3 * m, ldt = (n + m) | 1;
  T *tile = reinterpret_cast<T *>(sm + 24 * (size_t)m) + (size_t)wave * 64 * ldt;
  double esum = 0.0;
  for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < n_rows; base += (long long)gridDim.x * 256) {
    const int rows_here = (int)min<long long>(64, n_rows - base);
    if (TILED) {
      tile_load(X, base, rows_here, n, tile, ldt, lane);
      wave_sync();
    }
    if (lane < rows_here) {
      const long long a = base + lane;
      const T *xr = TILED ? tile + lane * ldt : X + a * n;
      T *zr = TILED ? tile + lane * ldt + n : z + a * m;
      double s[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] = (double)S[(size_t)i * n_rows + a];
      const double v4[4] = {s[0] * is0, s[1] * is1, s[2] * is2, s[3] * is3};
      double sum = 0.0;
      for (int k = 0; k < m; ++k) {
        const double x0 = (double)xr[3 * k], x1 = (double)xr[3 * k + 1], x2 = (double)xr[3 * k + 2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double h0 = x0 * inv, h1 = x1 * inv, h2 = x2 * inv;
        const double *w = sW + 12 * k;
        double xi = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) xi = fma(v4[i], h0 * w[3 * i] + h1 * w[3 * i + 1] + h2 * w[3 * i + 2], xi);
        sum += xi;
        zr[k] = (T)(xi * inv);
        esum += reproj_err2(sU + 12 * k, s, x0, x1, x2);
      }
      if (sum < 0.0)
        for (int k = 0; k < m; ++k) zr[k] = (T)(-(double)zr[k]);  // ref :217
    }
    if (TILED) {
      wave_sync();
      tile_store(z, base, rows_here, m, tile + n, ldt, lane);
      wave_sync();
    }
  }
  block_sum_to(esum, Epart);
}

// ---------------------------------------------------------------- the depth iteration without the re-weighted matrix (round 5)
// Round 4's iteration made ~7 passes over a 0.96 GB matrix at 5 M points x 8 images (write W = X o z normalised, Gram, rotate W V1
// into B, Gram of B, project S = M^T W, update) for ~1.5 GB of unavoidable traffic.  For few images (n = 3 m <= 32 columns, n even:
// the domain of the LDS-staged Gram kernel) nothing but X and z is streamed any more:
//   k_gram_xz     Gram of W formed on the fly: a workgroup stages 128 rows of X and of z, scales the rows IN LDS (norm 1: every row
//                 to unit length; norm 2: every image by its global scale cs), then the MFMA loop of k_gram_fused
//   k_rotgram_xz  the refinement pass in one kernel: the same staging, B = W V1 on the matrix cores into a second LDS tile, Gram of B
//                 from there -- B never exists in memory
//   k_primary_xz / k_dual_gram_mfma / k_dual_apply_xz   the depth updates with S = M^T W formed per row from X, z and M (24 x 4
//                 products) instead of read from a projection pass; the dual update also leaves the per-image sums of (x z)^2 of
//                 the NEW depths behind (the next iteration's norm-2 scales), so k_group_sumsq runs once per loop, not once per step
// Same arithmetic per entry as before up to the order of a few sums; oracle parity per step stays at 1e-9 (tests).
constexpr int FZ_MAXM = 10;  // images on this path (n = 3 m <= 32)

// Stage GRAM_ROWS rows of X ([rows][n]) and z ([rows][m]) and scale them in place: sx[r][c] <- x z[r][c / 3] s.
// NORM 1: s = 1 / |row of X o z|; NORM 2: s = cs[c / 3].  Rows past the end become zeros.  sp: [GRAM_ROWS][m] scratch.
struct FzRegs {
  double2 xs[GRAM_ROWS * 32 / 2 / 256];
  double2 zs[(GRAM_ROWS * FZ_MAXM / 2 + 255) / 256];
};
__device__ __forceinline__ void fz_load(const double *__restrict__ X, const double *__restrict__ z, long long r0, long long n_rows, int n, int m, FzRegs &R) {
  const long long tx = n_rows * n, tz = n_rows * m, ex = r0 * n, ez = r0 * m;  // (r0 is a multiple of 128: both offsets 16-byte aligned)
  const int nvx = GRAM_ROWS * n / 2, nvz = GRAM_ROWS * m / 2;
#pragma unroll
  for (int u = 0; u < (int)(sizeof(R.xs) / sizeof(double2)); ++u) {
    const int v = threadIdx.x + 256 * u;
    const long long i = ex + 2LL * v;
    if (v < nvx && i + 2 <= tx) R.xs[u] = *reinterpret_cast<const double2 *>(X + i);
    else R.xs[u] = double2{(v < nvx && i < tx) ? X[i] : 0.0, 0.0};
  }
#pragma unroll
  for (int u = 0; u < (int)(sizeof(R.zs) / sizeof(double2)); ++u) {
    const int v = threadIdx.x + 256 * u;
    const long long i = ez + 2LL * v;
    if (v < nvz && i + 2 <= tz) R.zs[u] = *reinterpret_cast<const double2 *>(z + i);
    else R.zs[u] = double2{(v < nvz && i < tz) ? z[i] : 0.0, 0.0};
  }
}
template <int NORM>
__device__ __forceinline__ void fz_stage(const FzRegs &R, long long r0, long long n_rows, int n, int m, const double *__restrict__ cs,
                                         double *sx, double *sz, double *sp) {
  const int nvx = GRAM_ROWS * n / 2, nvz = GRAM_ROWS * m / 2;
#pragma unroll
  for (int u = 0; u < (int)(sizeof(R.xs) / sizeof(double2)); ++u) {
    const int v = threadIdx.x + 256 * u;
    if (v < nvx) *reinterpret_cast<double2 *>(sx + 2 * v) = R.xs[u];
  }
#pragma unroll
  for (int u = 0; u < (int)(sizeof(R.zs) / sizeof(double2)); ++u) {
    const int v = threadIdx.x + 256 * u;
    if (v < nvz) *reinterpret_cast<double2 *>(sz + 2 * v) = R.zs[u];
  }
  __syncthreads();
  if (NORM == 1) {  // per (row, image): z^2 |x|^2; then ONE reciprocal root per row (the full-precision divide and root per
    // (row, image) of the first build cost 0.1 ms per pass at 5 M x 8); then the scaling
    for (int e = threadIdx.x; e < GRAM_ROWS * m; e += 256) {
      const int r = e / m, g = e - r * m;
      const double *x = sx + r * n + 3 * g;
      const double zz = sz[e];
      sp[e] = zz * zz * (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    }
    __syncthreads();
    if (threadIdx.x < GRAM_ROWS) {
      const int r = threadIdx.x;
      double ss = 0.0;
      for (int q = 0; q < m; ++q) ss += sp[r * m + q];
      sp[r * m] = (r0 + r < n_rows) ? fast_rsqrt(ss) : 0.0;  // (the row's slot 0: its partials have been read by this thread alone)
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < GRAM_ROWS * m; e += 256) {
    const int r = e / m, g = e - r * m;
    const double f = NORM == 1 ? sz[e] * sp[r * m] : sz[e] * cs[g];
    double *x = sx + r * n + 3 * g;
    x[0] *= f; x[1] *= f; x[2] *= f;
  }
  __syncthreads();
}

// the MFMA loop of k_gram_fused over the wave's 32 staged rows (`rows` = this lane's first row in a row-major [.][n] tile)
template <int MODE>
__device__ __forceinline__ void fz_gram_rows(const double *rows, int n, const int *col, const bool *cok, svd_d4 *acc) {
  constexpr int NC = MODE == 1 ? 1 : (MODE == 2 ? 3 : 2);
#pragma unroll
  for (int g = 0; g < ROWS_PER_STEP / 4; ++g) {
    double v[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) v[t] = cok[t] ? rows[(4 * g) * n + col[t]] : 0.0;
    if (MODE == 2) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[0], v[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[1], v[2], acc[1], 0, 0, 0);
    } else {
      int q = 0;
#pragma unroll
      for (int t = 0; t < NC; ++t)
#pragma unroll
        for (int u = t; u < NC; ++u, ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[t], v[u], acc[q], 0, 0, 0);
    }
  }
}

// ROT: the refinement pass (B = W V1 on the matrix cores into a second tile, Gram of B); else the first pass (Gram of W)
template <int M, int NORM, bool ROT>  // M images at compile time (the staging's divisions by m become shifts and multiplies)
__global__ __launch_bounds__(256) void k_gram_xz(const double *__restrict__ X, const double *__restrict__ z, long long n_rows, int n_rt, int m_rt,
                                                 const double *__restrict__ cs, const double *__restrict__ V1, double *__restrict__ partial) {
  constexpr int m = M, n = 3 * M, MODE = n <= 16 ? 1 : (n <= 24 ? 2 : 3);
  (void)n_rt; (void)m_rt;
  extern __shared__ double fz_lds[];
  double *sx = fz_lds, *sz = sx + GRAM_ROWS * n, *sp = sz + GRAM_ROWS * m, *sb = sp + GRAM_ROWS * m;  // sb only when ROT
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  constexpr int NP = MODE == 1 ? 1 : (MODE == 2 ? 2 : 3);
  constexpr int NC = MODE == 1 ? 1 : (MODE == 2 ? 3 : 2);
  svd_d4 acc[3] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
  int col[NC];
  bool cok[NC];
#pragma unroll
  for (int t = 0; t < NC; ++t) {
    const int c = MODE == 2 ? (t == 0 ? li : (t == 1 ? 8 + li : (li < 8 ? 16 + li : li - 8))) : GT * t + li;
    cok[t] = c < n; col[t] = min(c, n - 1);
  }
  const int nct = (n + 15) / 16, ng = (n + 3) / 4;
  double vb[2][8];
  if (ROT) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int k = 4 * g + lk;
#pragma unroll
      for (int c = 0; c < 2; ++c) vb[c][g] = (k < n && 16 * c + li < n) ? V1[(size_t)k * n + 16 * c + li] : 0.0;
    }
  }
  FzRegs R;
  const long long r_first = (long long)blockIdx.y * GRAM_ROWS, r_stride = (long long)gridDim.y * GRAM_ROWS;
  if (r_first < n_rows) fz_load(X, z, r_first, n_rows, n, m, R);
  for (long long r0 = r_first; r0 < n_rows; r0 += r_stride) {
    fz_stage<NORM>(R, r0, n_rows, n, m, cs, sx, sz, sp);
    if (r0 + r_stride < n_rows) fz_load(X, z, r0 + r_stride, n_rows, n, m, R);  // the next step's rows travel under this step's products
    if (ROT) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {  // the wave's two 16-row tiles: B = W V1 (as k_rotate_rows), into the wave's own rows of sb
        const double *rows = sx + (size_t)(wave * ROWS_PER_STEP + 16 * t + li) * n;
        svd_d4 ra[2] = {svd_d4{0, 0, 0, 0}, svd_d4{0, 0, 0, 0}};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          if (g >= ng) break;
          const int k = 4 * g + lk;
          const double a = (k < n) ? rows[k] : 0.0;
          ra[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vb[0][g], ra[0], 0, 0, 0);
          if (nct > 1) ra[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vb[1][g], ra[1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // C/D layout: col = li, row = lk + 4 q
          double *orow = sb + (size_t)(wave * ROWS_PER_STEP + 16 * t + lk + 4 * q) * n;
          if (li < n) orow[li] = ra[0][q];
          if (16 + li < n) orow[16 + li] = ra[1][q];
        }
      }
      wave_sync();  // (a wave multiplies the rows of sb it wrote itself)
      fz_gram_rows<MODE>(sb + (size_t)(wave * ROWS_PER_STEP + lk) * n, n, col, cok, acc);
    } else {
      fz_gram_rows<MODE>(sx + (size_t)(wave * ROWS_PER_STEP + lk) * n, n, col, cok, acc);
    }
    __syncthreads();
  }
  gram_store_partial(acc, NP, 0, NP, partial, fz_lds);  // (the staging tile: the last step ended with a barrier; NP x 8 KiB <= 128 n x 8 bytes for n >= 6 ...)
}

// per-point update of the primary scheme (k_depth_primary) with S formed from X, z and M on the fly; z in place
__global__ __launch_bounds__(256, 2) void k_primary_xz(const double *__restrict__ X, const double *__restrict__ Mr, long long n_rows, int m,
                                                    double *__restrict__ z, double *__restrict__ Epart) {
  extern __shared__ double sU[];  // [3m][4], then one tile of 64 x [3m values | m depths] per wave
  for (int q = threadIdx.x; q < 12 * m; q += blockDim.x) sU[q] = Mr[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = 3 * m, ldt = (n + m) | 1;
  double *tile = sU + 12 * (size_t)m + (size_t)wave * 64 * ldt;
  double esum = 0.0;
  for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < n_rows; base += (long long)gridDim.x * 256) {
    const int rows_here = (int)min<long long>(64, n_rows - base);
    tile_load(X, base, rows_here, n, tile, ldt, lane);
    tile_load(z, base, rows_here, m, tile + n, ldt, lane);
    wave_sync();
    if (lane < rows_here) {
      const double *xr = tile + lane * ldt;
      double *zr = tile + lane * ldt + n;
      double g[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s[4] = {0, 0, 0, 0}, ss = 0.0;
      for (int k = 0; k < m; ++k) {
        const double x0 = xr[3 * k], x1 = xr[3 * k + 1], x2 = xr[3 * k + 2], zk = zr[k];
        const double n2 = x0 * x0 + x1 * x1 + x2 * x2, inv = fast_rsqrt(n2);
        const double *u = sU + 12 * k;
        double c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double d = x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i];
          c[i] = d * inv;
          s[i] = fma(zk, d, s[i]);  // S = M^T w, w = x z / |row|: the row scale follows below
        }
        ss = fma(zk * zk, n2, ss);
        int e = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = i; j < 4; ++j, ++e) g[e] = fma(c[i], c[j], g[e]);
      }
      const double rs = 1.0 / sqrt(ss);
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] *= rs;
      double v[4];
      dominant_eigvec4(g, v);
      double nrm2 = 0.0, sum = 0.0;
      for (int k = 0; k < m; ++k) {
        const double x0 = xr[3 * k], x1 = xr[3 * k + 1], x2 = xr[3 * k + 2];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        const double *u = sU + 12 * k;
        double xi = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) xi = fma((x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i]) * inv, v[i], xi);
        esum += reproj_err2(u, s, x0, x1, x2);
        nrm2 = fma(xi, xi, nrm2);
        sum += xi;
        zr[k] = xi * inv;
      }
      const double sc = (sum < 0.0 ? -1.0 : 1.0) * fast_rsqrt(nrm2);
      for (int k = 0; k < m; ++k) zr[k] *= sc;
    }
    wave_sync();
    tile_store(z, base, rows_here, m, tile + n, ldt, lane);
    wave_sync();
  }
  block_sum_to(esum, Epart);
}

// dual scheme, pass 1 with V4 = S diag(1 / sigma) formed from X, z, cs and M while the rows are staged -- and the sums themselves
// on the matrix cores.  (A first build kept k_dual_gram's form, one thread per (image, entry) walking the staged rows: 1.08-1.16 ms at
// 5 M points x 8 images against 0.67 before S moved into the kernel -- the shares of S beside the observations doubled its LDS
// and left two workgroups per CU; 32 rows per pass instead of 128: 0.35 against 0.44 ms at 1 M points.)
// Per image k the 12 x 12 companion is the Gram matrix of Z_k[a] = V4[a] (x) x^_ak
// (rows x 12): one v_mfma_f64_16x16x4_f64 per 4 rows and image with the SAME operand on both sides -- Z_k[row][li], formed per lane
// from the staged V4 and x^ (two LDS reads, one product), column 12 of the 16-wide tile a constant 1 so that row 12 of the result
// is the column sum of Z_k that k_dual_vec needs.  A wave keeps the m tiles of its 32 rows in registers and writes them once:
// part[block x 4 + wave][m][256] in the C/D layout, summed in that order by k_dual_reduce_mfma (no atomics).  One thread per
// (image, entry) walking rows through LDS took 1.08 ms at 5 M points x 8 images (0.67 before S moved into the kernel).
template <int M>  // the image count at compile time (the fused path has m in {2, 4, 6, 8, 10}): no division per staging task, no branch between MFMAs
__global__ __launch_bounds__(256, 2) void k_dual_gram_mfma(const double *__restrict__ X, const double *__restrict__ z, const double *__restrict__ cs,
                                                            const double *__restrict__ Mr, double is0, double is1, double is2, double is3,
                                                            long long n_rows, int m_rt, double *__restrict__ part /*[gridDim.x * 4][m][256]*/) {
  constexpr int m = M;
  (void)m_rt;
  // Every WAVE stages its own 32 rows of a 128-row step (wave-private tiles, no workgroup barrier in the loop): the two waves of a
  // SIMD drift apart and one's staging runs under the other's MFMAs.  (With the workgroup staging 128 rows together behind
  // barriers the waves spent half their time waiting and the matrix cores were 37 % busy: 0.74 ms at 5 M x 8.)
  extern __shared__ double sg[];
  const int n = 3 * m, mp = m | 1, PL = ROWS_PER_STEP * mp + 8;  // shares of S: four planes [row][m | 1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const int wtile = ROWS_PER_STEP * (n + m + 4) + 4 * PL;        // doubles per wave: x (normalised in place) | z | V4 | shares
  double *sU = sg, *sx = sg + 12 * m + (size_t)wave * wtile, *sz = sx + ROWS_PER_STEP * n, *sv = sz + ROWS_PER_STEP * m, *ssh = sv + 4 * ROWS_PER_STEP;
  for (int q = threadIdx.x; q < 12 * m; q += 256) sU[q] = Mr[q];
  __syncthreads();
  const double isg[4] = {is0, is1, is2, is3};
  const int i3 = min(li / 3, 3), c3 = li - 3 * (li / 3);
  const double one12 = li == 12 ? 1.0 : 0.0, use = li < 12 ? 1.0 : 0.0;
  svd_d4 acc[M];
#pragma unroll
  for (int k = 0; k < M; ++k) acc[k] = svd_d4{0, 0, 0, 0};
  constexpr int MAXX = (ROWS_PER_STEP * 3 * M / 2 + 63) / 64, MAXZ = (ROWS_PER_STEP * M / 2 + 63) / 64;
  double2 xs[MAXX], zs[MAXZ];
  const int nvx = ROWS_PER_STEP * n / 2, nvz = ROWS_PER_STEP * m / 2;
  const long long tx = n_rows * n, tz = n_rows * m;
  auto wload = [&](long long rw) {  // this wave's 32 rows from row rw on (a multiple of 32: both offsets 16-byte aligned)
#pragma unroll
    for (int u = 0; u < MAXX; ++u) {
      const int v = lane + 64 * u;
      const long long i = rw * n + 2LL * v;
      if (v < nvx && i + 2 <= tx) xs[u] = *reinterpret_cast<const double2 *>(X + i);
      else xs[u] = double2{(v < nvx && i < tx) ? X[i] : 0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < MAXZ; ++u) {
      const int v = lane + 64 * u;
      const long long i = rw * m + 2LL * v;
      if (v < nvz && i + 2 <= tz) zs[u] = *reinterpret_cast<const double2 *>(z + i);
      else zs[u] = double2{(v < nvz && i < tz) ? z[i] : 0.0, 0.0};
    }
  };
  const long long w_first = ((long long)blockIdx.x * 4 + wave) * ROWS_PER_STEP, w_stride = (long long)gridDim.x * 4 * ROWS_PER_STEP;
  if (w_first < n_rows) wload(w_first);
  for (long long rw = w_first; rw < n_rows; rw += w_stride) {
    const int nr = (int)min<long long>(ROWS_PER_STEP, n_rows - rw);
#pragma unroll
    for (int u = 0; u < MAXX; ++u) {
      const int v = lane + 64 * u;
      if (v < nvx) *reinterpret_cast<double2 *>(sx + 2 * v) = xs[u];
    }
#pragma unroll
    for (int u = 0; u < MAXZ; ++u) {
      const int v = lane + 64 * u;
      if (v < nvz) *reinterpret_cast<double2 *>(sz + 2 * v) = zs[u];
    }
    wave_sync();
    if (rw + w_stride < n_rows) wload(rw + w_stride);  // the next step's rows travel while this step is staged and multiplied
    for (int e = lane; e < ROWS_PER_STEP * m; e += 64) {  // (row, image): normalise in place; the image's share of S = M^T w
      const int rr = e / m, k = e - rr * m;
      double *d = sx + rr * n + 3 * k, *sh = ssh + rr * mp + k;
      if (rr < nr) {
        const double x0 = d[0], x1 = d[1], x2 = d[2], f = sz[e] * cs[k];
        const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);
        d[0] = x0 * inv; d[1] = x1 * inv; d[2] = x2 * inv;
        const double *u = sU + 12 * k;
#pragma unroll
        for (int i = 0; i < 4; ++i) sh[i * PL] = f * (x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i]);
      } else {  // rows past the end contribute zeros
        d[0] = d[1] = d[2] = 0.0;
        sh[0] = sh[PL] = sh[2 * PL] = sh[3 * PL] = 0.0;
      }
    }
    wave_sync();
    for (int e = lane; e < ROWS_PER_STEP * 4; e += 64) {  // V4[row][i] = (sum of the images' shares) / sigma_i
      const int i = e / ROWS_PER_STEP, rr = e - i * ROWS_PER_STEP;
      const double *sh = ssh + i * PL + rr * mp;
      double t = 0.0;
      for (int k = 0; k < m; ++k) t += sh[k];
      sv[4 * rr + i] = t * isg[i];
    }
    wave_sync();
#pragma unroll
    for (int g = 0; g < ROWS_PER_STEP / 4; ++g) {
      const int row = 4 * g + lk;
      const double v4r = sv[4 * row + i3] * use, live = row < nr ? one12 : 0.0;
      const double *xh = sx + row * n + c3;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const double val = fma(v4r, xh[3 * k], live);
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(val, val, acc[k], 0, 0, 0);
      }
    }
    wave_sync();  // (the operands have been read before the next step's rows overwrite them)
  }
  double *o = part + ((size_t)(blockIdx.x * 4 + wave) * m) * 256;
#pragma unroll
  for (int k = 0; k < M; ++k) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(size_t)k * 256 + r * 64 + lane] = acc[k][r];
  }
}

// ... -> G12[k][12][12] (symmetric, index (i, c) = 3 i + c) and colsum[k][12]: one wave per entry, the partials in order
__global__ void k_dual_reduce_mfma(const double *__restrict__ part, int slots, int m, double *__restrict__ G12, double *__restrict__ colsum) {
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= m * 90) return;
  const int k = t / 90, e = t - 90 * k;  // e < 78: entry (i <= j) of the 12 x 12 matrix; else column sum e - 78
  int i = 0, j = 0;
  if (e < 78) {
    int rem = e;
    while (rem >= 12 - i) { rem -= 12 - i; ++i; }
    j = i + rem;
  } else {
    i = 12; j = e - 78;  // row 12 of the tile = sum over the rows of 1 x Z[.][j]
  }
  const double v = wave_strided_sum(part + (size_t)k * 256 + tile_elem(i, j), slots, (size_t)m * 256, lane);
  if (lane) return;
  if (e < 78) {
    G12[(size_t)k * 144 + i * 12 + j] = v;
    G12[(size_t)k * 144 + j * 12 + i] = v;
  } else {
    colsum[(size_t)k * 12 + j] = v;
  }
}

// dual scheme, pass 4 (k_dual_apply) with S formed on the fly; z in place; per block the per-image sums of (x z')^2 of the NEW depths
// (the image count stays a run-time value HERE: with it at compile time hipcc hoists the LDS reads of all M unrolled iterations --
// the w and U rows, 24 doubles per image -- in front of the loop: 256 registers + 400-600 bytes of scratch, 2.0 ms instead of 0.5)
__global__ __launch_bounds__(256, 2) void k_dual_apply_xz(const double *__restrict__ X, const double *__restrict__ cs, double is0, double is1, double is2,
                                                       double is3, const double *__restrict__ Mr, const double *__restrict__ w12,
                                                       long long n_rows, int m, double *__restrict__ z, double *__restrict__ Epart,
                                                       double *__restrict__ gpart /*[blocks][m]*/) {
  extern __shared__ double sm[];  // U4 [3m][4], w [m][12], then one tile of 64 x [3m values | m depths] per wave
  double *sU = sm, *sW = sm + 12 * (size_t)m;
  for (int q = threadIdx.x; q < 12 * m; q += blockDim.x) { sU[q] = Mr[q]; sW[q] = w12[q]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = 3 * m, ldt = (n + m) | 1;
  double *tile = sm + 24 * (size_t)m + (size_t)wave * 64 * ldt;
  double esum = 0.0, gs[FZ_MAXM];
#pragma unroll
  for (int k = 0; k < FZ_MAXM; ++k) gs[k] = 0.0;
  for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < n_rows; base += (long long)gridDim.x * 256) {
    const int rows_here = (int)min<long long>(64, n_rows - base);
    tile_load(X, base, rows_here, n, tile, ldt, lane);
    tile_load(z, base, rows_here, m, tile + n, ldt, lane);
    wave_sync();
    if (lane < rows_here) {
      const double *xr = tile + lane * ldt;
      double *zr = tile + lane * ldt + n;
      double s[4] = {0, 0, 0, 0};
      for (int k = 0; k < m; ++k) {
        const double x0 = xr[3 * k], x1 = xr[3 * k + 1], x2 = xr[3 * k + 2], f = zr[k] * cs[k];
        const double *u = sU + 12 * k;
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = fma(f, x0 * u[i] + x1 * u[4 + i] + x2 * u[8 + i], s[i]);
      }
      const double v4[4] = {s[0] * is0, s[1] * is1, s[2] * is2, s[3] * is3};
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < FZ_MAXM; ++k) {  // (fully unrolled: gs[] stays in registers)
        if (k < m) {
        const double x0 = xr[3 * k], x1 = xr[3 * k + 1], x2 = xr[3 * k + 2];
        const double n2 = x0 * x0 + x1 * x1 + x2 * x2, inv = fast_rsqrt(n2);
        const double h0 = x0 * inv, h1 = x1 * inv, h2 = x2 * inv;
        const double *w = sW + 12 * k;
        double xi = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) xi = fma(v4[i], h0 * w[3 * i] + h1 * w[3 * i + 1] + h2 * w[3 * i + 2], xi);
        sum += xi;
        const double zn = xi * inv;
        zr[k] = zn;
        gs[k] = fma(zn * zn, n2, gs[k]);  // (the row's sign flip below does not change it)
        esum += reproj_err2(sU + 12 * k, s, x0, x1, x2);
        }
      }
      if (sum < 0.0)
        for (int k = 0; k < m; ++k) zr[k] = -zr[k];  // ref :217
    }
    wave_sync();
    tile_store(z, base, rows_here, m, tile + n, ldt, lane);
    wave_sync();
  }
  block_sum_to(esum, Epart);
  // the m per-image sums: a fixed shuffle tree per wave, the four waves added in order by one thread per image -- two barriers
  // (m block-wide trees of eight barriers each were a tenth of this kernel: a block lives for ten rows per thread)
  __shared__ double s_gs[4][FZ_MAXM];
#pragma unroll
  for (int k = 0; k < FZ_MAXM; ++k) {
    if (k < m) {  // (m is uniform)
      double v = gs[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) s_gs[wave][k] = v;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < m) gpart[threadIdx.x * (size_t)gridDim.x + blockIdx.x] = (s_gs[0][threadIdx.x] + s_gs[1][threadIdx.x]) + (s_gs[2][threadIdx.x] + s_gs[3][threadIdx.x]);
}

// cs[g] = 1 / (sum over the blocks, in order, of gpart[g][block])
__global__ void k_group_scale_t(const double *__restrict__ gpart, int blocks, int ng, double *__restrict__ cs) {
  const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= ng) return;
  const double t = wave_strided_sum(gpart + (size_t)g * blocks, blocks, 1, lane);
  if (lane == 0) cs[g] = 1.0 / t;
}

__global__ void k_depth_error(const double *__restrict__ Epart, int blocks, double count, double f0, double *__restrict__ out) {
  const double t = wave_strided_sum(Epart, blocks, 1, (int)threadIdx.x);  // one wave
  if (threadIdx.x == 0) out[0] = f0 * sqrt(t / count);  // ref :56
}

}  // namespace

// Workspace: the resident matrix and every buffer a factorisation needs, allocated once.
struct mvsvd_handle {
  int device = 0, dtype = 0, n = 0;
  long long max_rows = 0, n_rows = 0, base_rows = 0;  // rows of the loaded matrix (dW) / of the resident base (dX)
  hipStream_t st = nullptr;
  hipEvent_t ev[8] = {};
  void *dW = nullptr, *dS = nullptr;
  double *dG = nullptr, *dV = nullptr, *dV1 = nullptr, *dsum = nullptr, *dMr = nullptr, *dmu = nullptr, *dpart = nullptr, *dpart2 = nullptr, *dB = nullptr;
  int *dsw = nullptr;
  void *dstage = nullptr;             // mvsvd_load_images: one image's [max_rows][2] array on its way into its two columns
  void *dX = nullptr, *dz = nullptr;  // mvsvd_load_base / mvsvd_run_scaled: resident base matrix, the depths of one call
  double *dgs = nullptr;              // column-group partial sums and scales
  bool base_loaded = false;
  int depth_group = 0;                // mvsvd_depth_begin: columns per image (3); 0 = no depth loop started
  bool cs_valid = false;              // fused depth iteration: the per-image scales of norm 2 belong to the depths in dz
  double *ddep = nullptr;             // depth iteration: error partials, 12 x 12 problems, vectors (one allocation)
  int *ddflag = nullptr;
  int chunks = 1, rank_cap = 0;
  // wide path (n > WIDE_MIN): bases Q, Z [n][WB], products B, B2 [max_rows][WB] (dBw / dB2), row-chunk partials of Z, small problems
  double *dQ = nullptr, *dZ = nullptr, *dQ2 = nullptr, *dBw = nullptr, *dB2 = nullptr, *dzpart = nullptr, *dsmall = nullptr;
  int *dwflag = nullptr;
  int zchunks = 1, wide_iters = 0;
  bool wide = false;
  bool wide_warm = false, wide_have_q = false, wide_q_center = false;  // warm start of the block iteration inside a depth loop
  long long zrows_per_chunk = 0;
  double h2d_ms = 0.0;
  bool loaded = false;
};

namespace {

// G = (W - mu)^T (W - mu) of an n_rows x n matrix through the workspace's partial buffers (the workspace's own matrix, or -- the
// wide path -- one of its n x 32 / N x 32 blocks)
template <typename T>
void launch_gram_n(mvsvd_handle *h, const T *W, long long n_rows, int n, const double *mu, int chunks, double *G) {
  const int n_tiles = (n + GT - 1) / GT;
  const bool packed = n > 16 && n <= 24;
  const int n_pairs = packed ? 2 : n_tiles * (n_tiles + 1) / 2;
  if (n <= 16)
    hipLaunchKernelGGL((k_gram_fused<T, 1>), dim3(1, chunks), dim3(256), std::max<size_t>(sizeof(T) * GRAM_ROWS * n, 1 * 8192), h->st, W, n_rows, n, mu, h->dpart);
  else if (packed)
    hipLaunchKernelGGL((k_gram_fused<T, 2>), dim3(1, chunks), dim3(256), std::max<size_t>(sizeof(T) * GRAM_ROWS * n, 2 * 8192), h->st, W, n_rows, n, mu, h->dpart);
  else if (n <= 32)
    hipLaunchKernelGGL((k_gram_fused<T, 3>), dim3(1, chunks), dim3(256), std::max<size_t>(sizeof(T) * GRAM_ROWS * n, 3 * 8192), h->st, W, n_rows, n, mu, h->dpart);
  else
    hipLaunchKernelGGL(k_gram_pair<T>, dim3(n_pairs, chunks), dim3(256), 0, h->st, W, n_rows, n, n_tiles, n_pairs, mu, h->dpart);
  const int slices = std::min(chunks, GRAM_SLICES);
  hipLaunchKernelGGL(k_gram_reduce, dim3(n_pairs, slices), dim3(256), 0, h->st, h->dpart, chunks, n_pairs, h->dpart2);
  hipLaunchKernelGGL(k_gram_finish, dim3((n + 127) / 128, n), dim3(128), 0, h->st, h->dpart2, slices, n_pairs, n_tiles, packed ? 1 : 0, G, n);
}
template <typename T>
void launch_gram(mvsvd_handle *h, const T *W, const double *mu, int chunks) { launch_gram_n<T>(h, W, h->n_rows, h->n, mu, chunks, h->dG); }

// tol: rotate while |a_pq| > tol sqrt(|a_pp a_qq|).  1e-15 for float64 input; float32 input carries
// 6e-8 of relative noise per entry, so its Gram matrix is diagonalised to 1e-11 (one or two sweeps fewer).
void launch_jacobi_n(mvsvd_handle *h, double *dA, int n, double *dVout, double tol) {
  const int np = (n + 1) & ~1;
  const size_t nn = (size_t)n * n, jl = sizeof(double) * (3 * (np / 2) + 2) + 16;
  if (n <= JW)
    hipLaunchKernelGGL(k_jacobi_small, dim3(1), dim3(JHB * JHB + JW * (np / 2)), 0, h->st, dA, dVout, n, 60, tol, h->dsw);  // (A blocks + V pieces)
  else if (n <= 64)
    hipLaunchKernelGGL(k_jacobi<true>, dim3(1), dim3(256), jl + sizeof(double) * 2 * nn, h->st, dA, dVout, n, 60, tol, h->dsw);
  else
    hipLaunchKernelGGL(k_jacobi<false>, dim3(1), dim3(1024), jl, h->st, dA, dVout, n, 60, tol, h->dsw);
}
void launch_jacobi(mvsvd_handle *h, double *dVout, double tol) { launch_jacobi_n(h, h->dG, h->n, dVout, tol); }

int chunks_for(long long n_rows, int n) {
  const int n_tiles = (n + GT - 1) / GT, n_pairs = n_tiles * (n_tiles + 1) / 2;
  return (int)std::max<long long>(1, std::min<long long>((n_rows + 4 * ROWS_PER_STEP - 1) / (4 * ROWS_PER_STEP),
                                                         n_tiles <= 2 ? 2048 : std::max(8, 4096 / n_pairs)));
}

template <typename T>
int run_wide(mvsvd_handle *h, int n_rank, int center, T *M, T *sigma, T *S, T *means, double *timings);

template <typename T>
int run(mvsvd_handle *h, int n_rank, int center, T *M, T *sigma, T *S, T *means, double *timings) {
  if (h->wide && (n_rank <= WB / 2 || h->n > JACOBI_MAX)) return run_wide<T>(h, n_rank, center, M, sigma, S, means, timings);
  const int n = h->n;
  const long long n_rows = h->n_rows;
  const size_t nn = (size_t)n * n;
  hipStream_t st = h->st;
  const T *dW = (const T *)h->dW;
  const int chunks = chunks_for(n_rows, n);
  const bool refine = sizeof(T) == 8;  // fp64 data: second, preconditioned pass (see the file header)
  hipEventRecord(h->ev[1], st);
  const double *mu = nullptr;
  if (center) {  // column means first: the rows are centred as they enter the products
    const int cy = (int)std::max<long long>(1, std::min<long long>(COLSUM_SLICES, n_rows / 4096));
    hipLaunchKernelGGL(k_colsum<T>, dim3(n, cy), dim3(256), 0, st, dW, n_rows, n, h->dsum);
    hipLaunchKernelGGL(k_mean_from_sum, dim3((n + 255) / 256), dim3(256), 0, st, h->dsum, cy, n, n_rows, h->dmu);
    mu = h->dmu;
  }
  launch_gram<T>(h, dW, mu, chunks);
  hipEventRecord(h->ev[2], st);
  launch_jacobi(h, refine ? h->dV1 : h->dV, sizeof(T) == 8 ? 1e-15 : 1e-11);
  hipEventRecord(h->ev[3], st);
  if (refine) {
    if (!h->dB) MVBA_HIP(hipMalloc((void **)&h->dB, sizeof(double) * (size_t)h->max_rows * n));
    if (n <= 32 && n % 2 == 0 && (n * sizeof(T)) % 16 == 0 && n_rows >= 4096) {  // tall and narrow: the streaming form
      const int rgrid = (int)std::max<long long>(1, std::min<long long>(6 * 256, (n_rows + GRAM_ROWS - 1) / GRAM_ROWS));
      const size_t rlds = (size_t)GRAM_ROWS * n * sizeof(T);
      hipLaunchKernelGGL(k_rotate_rows<T>, dim3(rgrid), dim3(256), rlds, st, dW, n_rows, n, mu, h->dV1, h->dB);
    } else
    hipLaunchKernelGGL(k_rotate<T>, dim3((unsigned)((n_rows + 63) / 64), (n + 63) / 64), dim3(256), 0, st, dW, n_rows, n, mu, h->dV1, h->dB);
    launch_gram<double>(h, h->dB, nullptr, chunks);
    launch_jacobi(h, h->dMr /* V2, n x n: dMr is sized for it */, 1e-15);
    // V = V1 V2
    hipLaunchKernelGGL(k_rotate<double>, dim3((n + 63) / 64, (n + 63) / 64), dim3(256), 0, st, h->dV1, (long long)n, n, (const double *)nullptr,
                       h->dMr, h->dV);
  }
  // eigenvalues -> host, sort, build the rank-r basis with a deterministic sign
  std::vector<double> hG(nn), hV(nn), hmu(n, 0.0);
  MVBA_HIP(hipMemcpyAsync(hG.data(), h->dG, sizeof(double) * nn, hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipMemcpyAsync(hV.data(), h->dV, sizeof(double) * nn, hipMemcpyDeviceToHost, st));
  if (center) MVBA_HIP(hipMemcpyAsync(hmu.data(), h->dmu, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i)  // (np.linalg.svd raises LinAlgError("SVD did not converge") on such input; max() and sort() below would swallow the NaN)
    if (!std::isfinite(hG[(size_t)i * n + i])) return fail(MVBA_ERR_SINGULAR, "SVD did not converge (non-finite values in the measurement matrix)");
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return hG[(size_t)a * n + a] > hG[(size_t)b * n + b]; });
  std::vector<double> Mr((size_t)n * n_rank);
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
  if (means)
    for (int c = 0; c < n; ++c) means[c] = (T)hmu[c];
  // S = M^T W in groups of (up to) 4 basis vectors per pass over W
  if (n_rank > h->rank_cap) {
    if (h->dS) MVBA_HIP(hipFree(h->dS));
    h->dS = nullptr;
    MVBA_HIP(hipMalloc(&h->dS, sizeof(T) * (size_t)h->max_rows * n_rank));
    h->rank_cap = n_rank;
  }
  hipEventRecord(h->ev[4], st);
  const int pgrid = (int)std::max<long long>(1, std::min<long long>(4096, (n_rows + 255) / 256));
  const int ldt = std::min(n, PC) | 1;
  const size_t plds = sizeof(double) * (5 * (size_t)n) + sizeof(T) * 4 * 64 * (size_t)ldt + 16;
  std::vector<double> Mg((size_t)n * 4);
  for (int g0 = 0; g0 < n_rank; g0 += 4) {
    const int rg = std::min(4, n_rank - g0);
    for (int c = 0; c < n; ++c)
      for (int i = 0; i < 4; ++i) Mg[(size_t)c * 4 + i] = i < rg ? Mr[(size_t)c * n_rank + g0 + i] : 0.0;  // [n][4], zero padded
    MVBA_HIP(hipMemcpyAsync(h->dMr, Mg.data(), sizeof(double) * (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_project<T>, dim3(pgrid), dim3(256), plds, st, dW, n_rows, n, rg, h->dMr, mu, (T *)h->dS + (size_t)g0 * n_rows);
    MVBA_HIP(hipStreamSynchronize(st));  // Mg is reused by the next group
  }
  hipEventRecord(h->ev[5], st);
  if (S) MVBA_HIP(hipMemcpyAsync(S, h->dS, sizeof(T) * (size_t)n_rows * n_rank, hipMemcpyDeviceToHost, st));  // (null: S stays on the device)
  MVBA_HIP(hipStreamSynchronize(st));
  MVBA_HIP(hipGetLastError());
  if (timings) {
    float ms;
    timings[0] = h->h2d_ms;                                                   // H2D of the last load
    hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); timings[1] = ms;            // means + Gram
    hipEventElapsedTime(&ms, h->ev[2], h->ev[3]); timings[2] = ms;            // Jacobi (first pass)
    hipEventElapsedTime(&ms, h->ev[4], h->ev[5]); timings[3] = ms;            // projection
    int sw = 0;
    hipMemcpy(&sw, h->dsw, sizeof(int), hipMemcpyDeviceToHost);
    timings[4] = sw;                                                          // Jacobi sweeps (last pass)
    if (refine) { hipEventElapsedTime(&ms, h->ev[3], h->ev[4]); timings[5] = ms; }  // refinement pass (rotate, Gram, Jacobi, V1 V2)
    else timings[5] = 0.0;
  }
  return MVBA_OK;
}

}  // namespace

namespace {

// The factorisation of a wide matrix (n > WIDE_MIN columns): block power iteration with Rayleigh-Ritz, see the kernels' header.
template <typename T>
int run_wide(mvsvd_handle *h, int n_rank, int center, T *M, T *sigma, T *S, T *means, double *timings) {
  const int n = h->n;
  const long long N = h->n_rows;
  hipStream_t st = h->st;
  const T *dW = (const T *)h->dW;
  if (n_rank > WB / 2)
    return fail(MVBA_ERR_BADARG, "n_cols > " + std::to_string(JACOBI_MAX) + ": the block iteration carries " + std::to_string(WB) + " vectors, n_rank <= " + std::to_string(WB / 2));
  hipEventRecord(h->ev[1], st);
  const double *mu = nullptr;
  if (center) {
    const int cy = (int)std::max<long long>(1, std::min<long long>(COLSUM_SLICES, N / 4096));
    hipLaunchKernelGGL(k_colsum<T>, dim3(n, cy), dim3(256), 0, st, dW, N, n, h->dsum);
    hipLaunchKernelGGL(k_mean_from_sum, dim3((n + 255) / 256), dim3(256), 0, st, h->dsum, cy, n, N, h->dmu);
    mu = h->dmu;
  }
  hipEventRecord(h->ev[2], st);
  double *H = h->dsmall, *Y = H + WB * WB, *C = Y + WB * WB, *V = C + WB * WB, *Tm = V + WB * WB, *dres = Tm + WB * WB, *respart = dres + WB;
  int *dcol = h->dwflag + WB;                      // [WB] columns to take at the end
  double *dsgn = respart + (size_t)WB * ((n + 255) / 256);  // [WB] their signs
  const int chunksB = chunks_for(N, WB), chunksZ = chunks_for(n, WB);
  const int wgrid = (int)((N + WT - 1) / WT), ngrid = (int)(((long long)n * WB + 255) / 256), rgrid = (n + 255) / 256;
  const int rot_grid_n = (int)std::max<long long>(1, std::min<long long>(6 * 256, (n + GRAM_ROWS - 1) / GRAM_ROWS));
  const size_t rot_lds = (size_t)GRAM_ROWS * WB * sizeof(double);
  const int bgrid = (int)std::max<long long>(1, std::min<long long>(6 * 256, (N + GRAM_ROWS - 1) / GRAM_ROWS));
  auto orth = [&](double *src, double *tmp, double *dst, const double *ritz, unsigned long long salt) {  // dst <- orthonormal basis of span(src), in Ritz order; src and tmp are scratch
    launch_gram_n<double>(h, src, n, WB, nullptr, chunksZ, C);
    hipLaunchKernelGGL(k_chol_orth, dim3(1), dim3(64), 0, st, C, ritz, Tm, h->dwflag);
    hipLaunchKernelGGL(k_rotate_rows<double>, dim3(rot_grid_n), dim3(256), rot_lds, st, src, (long long)n, WB, (const double *)nullptr, Tm, tmp);
    hipLaunchKernelGGL(k_wide_refill, dim3(ngrid), dim3(256), 0, st, tmp, n, h->dwflag, salt);
    launch_gram_n<double>(h, tmp, n, WB, nullptr, chunksZ, C);
    hipLaunchKernelGGL(k_chol_orth, dim3(1), dim3(64), 0, st, C, (const double *)nullptr, Tm, h->dwflag);
    hipLaunchKernelGGL(k_rotate_rows<double>, dim3(rot_grid_n), dim3(256), rot_lds, st, tmp, (long long)n, WB, (const double *)nullptr, Tm, dst);
  };
  auto product_b = [&](const double *Q) {  // B = (W - mu) Q, H = B^T B, Jacobi: theta on H's diagonal, Y
    hipLaunchKernelGGL(k_wq<T>, dim3(wgrid), dim3(256), 0, st, dW, N, n, mu, Q, h->dBw);
    launch_gram_n<double>(h, h->dBw, N, WB, nullptr, chunksB, H);
    launch_jacobi_n(h, H, WB, Y, 1e-15);
  };
  // start: a fixed pseudo-random block, orthonormalised -- or, inside a depth loop (the same base re-weighted by slightly different
  // depths, 50-200 times: mvsvd_run_scaled / mvsvd_depth_step), the Ritz vectors the previous factorisation ended with: one or two
  // iterations instead of three or four.  A newly loaded matrix (mvsvd_load / mvsvd_load_base) always starts from the fixed block.
  if (!(h->wide_warm && h->wide_have_q && h->wide_q_center == (center != 0))) {
    hipLaunchKernelGGL(k_wide_init, dim3(ngrid), dim3(256), 0, st, h->dZ, n, 0x5eedull);
    orth(h->dZ, h->dQ2, h->dQ, nullptr, 1);
  }
  h->wide_have_q = false;  // (until this call has converged)
  double theta[WB], res[WB];
  const int max_iter = 2000;
  int it = 0;
  // Residuals of the leading n_rank Ritz pairs (k_ritz: the part of W^T W q_j outside the block): `worst` relative to theta_1 (what
  // the leading triplets need: <= 1e-13, or at this matrix's rounding floor -- no halving over three iterations -- below 1e-10)
  // and `rel`, each against its own pair's scale max(1e-12 theta_j, 50 eps sqrt(theta_1 theta_j)): the angle of a SMALL
  // triplet's vector is r_j / theta_j (a sigma_4 = 1e-7 sigma_1 is invisible at the 1e-13 theta_1 level), and
  // eps sigma_1 / sigma_j is what the products W q, W^T b leave of it.
  double worst = 0.0, rel = 0.0, best = 1e300, best_rel = 1e300;
  int stalled = 0, stalled_rel = 0;
  bool converged = false;
  for (; it < max_iter; ++it) {
    product_b(h->dQ);
    hipLaunchKernelGGL(k_rotate_rows<double>, dim3(bgrid), dim3(256), rot_lds, st, h->dBw, N, WB, (const double *)nullptr, Y, h->dB2);  // B Y = W (Q Y)
    hipLaunchKernelGGL(k_wtb<T>, dim3((n + WT - 1) / WT, h->zchunks), dim3(256), 0, st, dW, N, n, mu, h->dB2, h->zrows_per_chunk, h->dzpart);
    hipLaunchKernelGGL(k_sum_chunks, dim3(ngrid), dim3(256), 0, st, h->dzpart, h->zchunks, (long long)n * WB, h->dZ);
    hipLaunchKernelGGL(k_ritz, dim3(rgrid), dim3(256), 0, st, h->dQ, (const double *)h->dZ, n, Y, H, respart);
    hipLaunchKernelGGL(k_sum_chunks, dim3(1), dim3(256), 0, st, respart, rgrid, (long long)WB, dres);
    std::vector<double> hH((size_t)WB * WB);
    MVBA_HIP(hipMemcpyAsync(hH.data(), H, sizeof(double) * WB * WB, hipMemcpyDeviceToHost, st));
    MVBA_HIP(hipMemcpyAsync(res, dres, sizeof(double) * WB, hipMemcpyDeviceToHost, st));
    MVBA_HIP(hipStreamSynchronize(st));
    for (int j = 0; j < WB; ++j) {
      theta[j] = hH[(size_t)j * WB + j];
      if (!std::isfinite(theta[j]) || !std::isfinite(res[j]))  // (before anything is compared: max() and sort() swallow a NaN)
        return fail(MVBA_ERR_SINGULAR, "SVD did not converge (non-finite values in the measurement matrix)");
    }
    int order[WB];
    std::iota(order, order + WB, 0);
    std::sort(order, order + WB, [&](int a, int b) { return theta[a] > theta[b]; });
    const double tmax = std::max(theta[order[0]], 0.0);
    worst = rel = 0.0;
    for (int i = 0; i < n_rank; ++i) {
      const double rj = std::sqrt(std::max(res[order[i]], 0.0)), tj = theta[order[i]];
      worst = std::max(worst, rj);
      if (tj > 1e-28 * tmax) rel = std::max(rel, rj / std::max(1e-12 * tj, 1.1e-14 * std::sqrt(tmax * tj)));  // (sigma_j below 1e-14 sigma_1 is numerically zero: the absolute measure covers it)
    }
    worst = tmax > 0.0 ? worst / tmax : 0.0;
    if (!(worst == worst) || !(rel == rel)) return fail(MVBA_ERR_SINGULAR, "SVD did not converge (non-finite values in the measurement matrix)");
    if (worst < 0.5 * best) { best = worst; stalled = 0; } else ++stalled;
    if (rel < 0.5 * best_rel) { best_rel = rel; stalled_rel = 0; } else ++stalled_rel;
    const bool done_abs = worst <= 1e-13 || (stalled >= 3 && worst <= 1e-10), done_rel = rel <= 1.0 || stalled_rel >= 3;
    if (done_abs && done_rel) { converged = true; ++it; break; }
    orth(h->dZ, h->dQ2, h->dQ, H, 2 + (unsigned long long)it);
  }
  h->wide_iters = it;
  h->wide_have_q = converged;
  h->wide_q_center = center != 0;
  if (!converged)
    return fail(MVBA_ERR_SINGULAR, "SVD did not converge: block power iteration, residual " + std::to_string(worst) + " of sigma_1^2 after " + std::to_string(it) +
                                       " iterations (singular values " + std::to_string(n_rank) + " .. " + std::to_string(WB) + " of this matrix are too close)");
  hipEventRecord(h->ev[3], st);
  // final pass: Q holds the Ritz vectors; B = (W - mu) Q has a nearly diagonal, graded Gram matrix -> sigma and the last rotation
  product_b(h->dQ);
  hipLaunchKernelGGL(k_ritz, dim3(rgrid), dim3(256), 0, st, h->dQ, (const double *)nullptr, n, Y, H, (double *)nullptr);
  hipLaunchKernelGGL(k_rotate_rows<double>, dim3(bgrid), dim3(256), rot_lds, st, h->dBw, N, WB, (const double *)nullptr, Y, h->dB2);
  std::vector<double> hH((size_t)WB * WB), hQ((size_t)n * WB), hmu(n, 0.0);
  MVBA_HIP(hipMemcpyAsync(hH.data(), H, sizeof(double) * WB * WB, hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipMemcpyAsync(hQ.data(), h->dQ, sizeof(double) * (size_t)n * WB, hipMemcpyDeviceToHost, st));
  if (center) MVBA_HIP(hipMemcpyAsync(hmu.data(), h->dmu, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipStreamSynchronize(st));
  int order[WB];
  std::iota(order, order + WB, 0);
  std::sort(order, order + WB, [&](int a, int b) { return hH[(size_t)a * WB + a] > hH[(size_t)b * WB + b]; });
  for (int i = 0; i < n; ++i) sigma[i] = i < WB ? (T)std::sqrt(std::max(0.0, hH[(size_t)order[i] * WB + order[i]])) : (T)NAN;
  int hcol[WB] = {};
  double hsgn[WB] = {};
  for (int i = 0; i < n_rank; ++i) {
    const int col = order[i];
    int big = 0;
    for (int c = 1; c < n; ++c)
      if (std::fabs(hQ[(size_t)c * WB + col]) > std::fabs(hQ[(size_t)big * WB + col])) big = c;
    const double sg = hQ[(size_t)big * WB + col] < 0.0 ? -1.0 : 1.0;  // largest component positive
    for (int c = 0; c < n; ++c) M[(size_t)c * n_rank + i] = (T)(sg * hQ[(size_t)c * WB + col]);
    hcol[i] = col;
    hsgn[i] = sg;
  }
  if (means)
    for (int c = 0; c < n; ++c) means[c] = (T)hmu[c];
  if (n_rank > h->rank_cap) {
    if (h->dS) MVBA_HIP(hipFree(h->dS));
    h->dS = nullptr;
    MVBA_HIP(hipMalloc(&h->dS, sizeof(T) * (size_t)h->max_rows * n_rank));
    h->rank_cap = n_rank;
  }
  hipEventRecord(h->ev[4], st);
  std::vector<double> Mg((size_t)n * 4, 0.0);  // the depth loops read the leading basis vectors as [n][4] from dMr, like after run()
  for (int c = 0; c < n; ++c)
    for (int i = 0; i < std::min(4, n_rank); ++i) Mg[(size_t)c * 4 + i] = hsgn[i] * hQ[(size_t)c * WB + hcol[i]];
  MVBA_HIP(hipMemcpyAsync(h->dMr, Mg.data(), sizeof(double) * (size_t)n * 4, hipMemcpyHostToDevice, st));
  MVBA_HIP(hipMemcpyAsync(dcol, hcol, sizeof(int) * WB, hipMemcpyHostToDevice, st));
  MVBA_HIP(hipMemcpyAsync(dsgn, hsgn, sizeof(double) * WB, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_take_cols<T>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, h->dB2, N, n_rank, dcol, dsgn, (T *)h->dS);
  hipEventRecord(h->ev[5], st);
  if (S) MVBA_HIP(hipMemcpyAsync(S, h->dS, sizeof(T) * (size_t)N * n_rank, hipMemcpyDeviceToHost, st));  // (null: S stays on the device)
  MVBA_HIP(hipStreamSynchronize(st));  // (hcol / hsgn are read by the copies above)
  MVBA_HIP(hipGetLastError());
  if (timings) {
    float ms;
    timings[0] = h->h2d_ms;
    hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); timings[1] = ms;  // means
    hipEventElapsedTime(&ms, h->ev[2], h->ev[3]); timings[2] = ms;  // the iteration (in the eigen-solver's slot)
    hipEventElapsedTime(&ms, h->ev[4], h->ev[5]); timings[3] = ms;  // S out of B
    timings[4] = it;                                                // iterations (in the sweeps' slot)
    hipEventElapsedTime(&ms, h->ev[3], h->ev[4]); timings[5] = ms;  // final pass
  }
  return MVBA_OK;
}

constexpr int GS_BLOCKS = 512;

// dW <- the resident base re-weighted by the depths in dz and normalised (see mvsvd_run_scaled)
int scale_base_into_w(mvsvd_handle *h, int group, int norm) {
  const int ng = h->n / group;
  const size_t gs_stride = (size_t)std::max(256, h->n);  // (room for the finest grouping: one group per column)
  if (!h->dgs) MVBA_HIP(hipMalloc((void **)&h->dgs, sizeof(double) * (size_t)(GS_BLOCKS + 1) * gs_stride));
  double *cs = h->dgs + (size_t)GS_BLOCKS * gs_stride;
  const int sgrid = (int)std::max<long long>(1, std::min<long long>(4096, (h->base_rows + 255) / 256));
  const int gblocks = (int)std::max<long long>(1, std::min<long long>(GS_BLOCKS, h->base_rows / 64 + 1));
  if (h->dtype == 0) {
    if (norm == 2) {
      hipLaunchKernelGGL(k_group_sumsq<float>, dim3(gblocks), dim3(256), 0, h->st, (const float *)h->dX, (const float *)h->dz, h->base_rows, h->n, group, h->dgs);
      hipLaunchKernelGGL(k_group_scale, dim3((ng + 3) / 4), dim3(256), 0, h->st, h->dgs, gblocks, ng, cs);
    }
    if (tile_fits<float>(h->n + ng))
      hipLaunchKernelGGL(k_scale_rows_tiled<float>, dim3(sgrid), dim3(256), sizeof(float) * 4 * 64 * (size_t)((h->n + ng) | 1), h->st, (const float *)h->dX, (const float *)h->dz, h->base_rows, h->n, group, norm, cs, (float *)h->dW);
    else
      hipLaunchKernelGGL(k_scale_rows_wide<float>, dim3(sgrid), dim3(256), 0, h->st, (const float *)h->dX, (const float *)h->dz, h->base_rows, h->n, group, norm, cs, (float *)h->dW);
  } else {
    if (norm == 2) {
      hipLaunchKernelGGL(k_group_sumsq<double>, dim3(gblocks), dim3(256), 0, h->st, (const double *)h->dX, (const double *)h->dz, h->base_rows, h->n, group, h->dgs);
      hipLaunchKernelGGL(k_group_scale, dim3((ng + 3) / 4), dim3(256), 0, h->st, h->dgs, gblocks, ng, cs);
    }
    if (tile_fits<double>(h->n + ng))
      hipLaunchKernelGGL(k_scale_rows_tiled<double>, dim3(sgrid), dim3(256), sizeof(double) * 4 * 64 * (size_t)((h->n + ng) | 1), h->st, (const double *)h->dX, (const double *)h->dz, h->base_rows, h->n, group, norm, cs, (double *)h->dW);
    else
      hipLaunchKernelGGL(k_scale_rows_wide<double>, dim3(sgrid), dim3(256), 0, h->st, (const double *)h->dX, (const double *)h->dz, h->base_rows, h->n, group, norm, cs, (double *)h->dW);
  }
  MVBA_HIP(hipGetLastError());
  h->n_rows = h->base_rows;  // dW now holds the re-weighted base (a mvsvd_load in between may have changed n_rows)
  h->loaded = true;
  h->wide_warm = true;  // (the block iteration may start from the previous factorisation's vectors: the same base, other depths)
  return MVBA_OK;
}

constexpr int DEPTH_BLOCKS = 2048;  // blocks of the per-point passes / of the dual Gram pass
constexpr int DEPTH_MAX_IMAGES = 768;  // k_dual_apply_wide keeps U4 and w (24 doubles per image) in LDS: 147,456 B of DEPTH_LDS_MAX
constexpr size_t DEPTH_LDS_MAX = 148 * 1024;  // dynamic LDS the depth kernels may ask for (+ 2 KiB static in block_sum_to)
// rows k_dual_gram stages per pass: [rows][3 m | 1] normalised observations + [rows][4] right singular vectors in LDS
inline int dual_gram_rows(int m) { return (int)std::max<size_t>(1, std::min<size_t>(DG_ROWS_MAX, DEPTH_LDS_MAX / (sizeof(double) * (size_t)(((3 * m) | 1) + 4)))); }

template <typename T>
int depth_step(mvsvd_handle *h, int method, double f0, double *E, double *timings) {
  const int n = h->n, m = n / 3;
  const long long rows = h->base_rows;
  int rc = scale_base_into_w(h, 3, method);
  if (rc) return rc;
  std::vector<T> M((size_t)n * 4), sigma(n);
  rc = run<T>(h, 4, 0, M.data(), sigma.data(), (T *)nullptr, (T *)nullptr, timings);  // dMr = M, dS = S stay on the device
  if (rc) return rc;
  // ddep: [DEPTH_BLOCKS] error partials | [1] error | G12 [m][144] | V12 [m][144] | colsum [m][12] | w12 [m][12] | dual partials
  const int dsplit = dual_gram_split(m);  // thread groups of k_dual_gram, each with its own partial
  const int dual_blocks = std::max(64, std::min(DEPTH_BLOCKS, 95000 / m / dsplit));  // (blocks x groups partials of 14 m x 6 doubles: <= ~64 MB)
  const long long rpb = (rows + dual_blocks - 1) / dual_blocks;
  const int gblocks = (int)((rows + rpb - 1) / rpb);
  const size_t need = (size_t)DEPTH_BLOCKS + 8 + (size_t)m * (144 + 144 + 12 + 12) + (size_t)dual_blocks * dsplit * 14 * m * 6;
  if (!h->ddep) MVBA_HIP(hipMalloc((void **)&h->ddep, sizeof(double) * need));  // (m is the handle's: `need` never changes)
  if (!h->ddflag) MVBA_HIP(hipMalloc((void **)&h->ddflag, sizeof(int) * (size_t)(m + 1)));
  double *Epart = h->ddep, *Eout = Epart + DEPTH_BLOCKS, *G12 = Eout + 8, *V12 = G12 + (size_t)m * 144, *colsum = V12 + (size_t)m * 144,
         *w12 = colsum + (size_t)m * 12, *gpart = w12 + (size_t)m * 12;
  const int pgrid = (int)std::max<long long>(1, std::min<long long>(DEPTH_BLOCKS, (rows + 255) / 256));
  const bool tiled = tile_fits<T>(n + m) && 24 * m * sizeof(double) <= 24 * 1024;  // the rows go through LDS tiles (coalesced) while they fit
  const size_t tile_bytes = sizeof(T) * 4 * 64 * (size_t)((n + m) | 1) + 16;
  hipStream_t st = h->st;
  hipEventRecord(h->ev[6], st);
  if (method == 1) {
    if (tiled)
      hipLaunchKernelGGL((k_depth_primary<T, true>), dim3(pgrid), dim3(256), sizeof(double) * 12 * m + tile_bytes, st, (const T *)h->dX, h->dMr,
                         (const T *)h->dS, rows, m, (T *)h->dz, Epart);
    else
      hipLaunchKernelGGL(k_depth_primary_wide<T>, dim3(pgrid), dim3(256), sizeof(double) * (12 * (size_t)m + 4 * 64 * 14), st, (const T *)h->dX, h->dMr,
                         (const T *)h->dS, rows, m, (T *)h->dz, Epart);
  } else {
    double is[4];
    for (int i = 0; i < 4; ++i) {
      if (!((double)sigma[i] > 0.0)) return fail(MVBA_ERR_SINGULAR, "measurement matrix has rank < 4");
      is[i] = 1.0 / (double)sigma[i];
    }
    MVBA_HIP(hipMemsetAsync(h->ddflag, 0, sizeof(int), st));
    const int dg_rows = dual_gram_rows(m);  // (128 up to 47 images, 24 at 256: the launch used to FAIL from 48 images on)
    hipLaunchKernelGGL(k_dual_gram<T>, dim3(gblocks), dim3(256), sizeof(double) * ((size_t)dg_rows * ((n | 1) + 4)), st, (const T *)h->dX, (const T *)h->dS,
                       is[0], is[1], is[2], is[3], rows, m, rpb, dg_rows, gpart);
    hipLaunchKernelGGL(k_dual_reduce, dim3((14 * m * 6 + 3) / 4), dim3(256), 0, st, gpart, gblocks * dsplit, m, G12, colsum);
    hipLaunchKernelGGL(k_jacobi_small, dim3(m), dim3(JHB * JHB + JW * 6), 0, st, G12, V12, 12, 60, 1e-15, h->ddflag + 1);
    hipLaunchKernelGGL(k_dual_vec, dim3((m + 63) / 64), dim3(64), 0, st, G12, V12, colsum, m, w12, h->ddflag);
    if (tiled)
      hipLaunchKernelGGL((k_dual_apply<T, true>), dim3(pgrid), dim3(256), sizeof(double) * 24 * m + tile_bytes, st, (const T *)h->dX, (const T *)h->dS,
                         is[0], is[1], is[2], is[3], h->dMr, w12, rows, m, (T *)h->dz, Epart);
    else
      hipLaunchKernelGGL(k_dual_apply_wide<T>, dim3(pgrid), dim3(256), sizeof(double) * 24 * m, st, (const T *)h->dX, (const T *)h->dS,
                         is[0], is[1], is[2], is[3], h->dMr, w12, rows, m, (T *)h->dz, Epart);
  }
  hipLaunchKernelGGL(k_depth_error, dim3(1), dim3(64), 0, st, Epart, pgrid, (double)rows * (double)m, f0, Eout);
  hipEventRecord(h->ev[7], st);
  MVBA_HIP(hipGetLastError());
  int fl = 0;
  MVBA_HIP(hipMemcpyAsync(E, Eout, sizeof(double), hipMemcpyDeviceToHost, st));
  if (method == 2) MVBA_HIP(hipMemcpyAsync(&fl, h->ddflag, sizeof(int), hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipStreamSynchronize(st));
  if (fl) return fail(MVBA_ERR_SINGULAR, "depth iteration: an image's 12 x 12 companion matrix has no positive eigenvalue");
  if (timings) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->ev[6], h->ev[7]);
    timings[0] = ms;  // (no upload in a depth step: slot 0 carries the depth-update kernels instead)
  }
  return MVBA_OK;
}
// The iteration without the re-weighted matrix (see k_gram_xz): fp64, n = 3 m <= 32 columns, n even.
bool fused_depth_ok(const mvsvd_handle *h) {
  return h->dtype == 1 && h->n <= 32 && h->n % 2 == 0 && h->n / 3 <= FZ_MAXM && h->base_rows >= 256 && !getenv("MVSVD_DEPTH_UNFUSED");
}

template <int M, int NORM>
void launch_gram_xz(mvsvd_handle *h, bool rot, int chunks, const double *cs) {
  constexpr int n = 3 * M, m = M, MODE = n <= 16 ? 1 : (n <= 24 ? 2 : 3);
  constexpr int NP = MODE == 1 ? 1 : (MODE == 2 ? 2 : 3);
  const size_t lds = std::max<size_t>(sizeof(double) * (size_t)GRAM_ROWS * ((rot ? 2 : 1) * n + 2 * m), (size_t)NP * 8192);
  static bool attr_set = false;
  if (!attr_set) {  // (the refinement form of 10 images stages 80 KiB)
    hipFuncSetAttribute((const void *)k_gram_xz<M, NORM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEPTH_LDS_MAX);
    hipFuncSetAttribute((const void *)k_gram_xz<M, NORM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEPTH_LDS_MAX);
    attr_set = true;
  }
  if (rot)
    hipLaunchKernelGGL((k_gram_xz<M, NORM, true>), dim3(1, chunks), dim3(256), lds, h->st, (const double *)h->dX, (const double *)h->dz, h->base_rows, n, m, cs,
                       (const double *)h->dV1, h->dpart);
  else
    hipLaunchKernelGGL((k_gram_xz<M, NORM, false>), dim3(1, chunks), dim3(256), lds, h->st, (const double *)h->dX, (const double *)h->dz, h->base_rows, n, m, cs,
                       (const double *)nullptr, h->dpart);
  const int n_tiles = (n + GT - 1) / GT, slices = std::min(chunks, GRAM_SLICES);
  hipLaunchKernelGGL(k_gram_reduce, dim3(NP, slices), dim3(256), 0, h->st, h->dpart, chunks, NP, h->dpart2);
  hipLaunchKernelGGL(k_gram_finish, dim3((n + 127) / 128, n), dim3(128), 0, h->st, h->dpart2, slices, NP, n_tiles, MODE == 2 ? 1 : 0, h->dG, n);
}
template <int NORM>
void gram_xz_m(mvsvd_handle *h, bool rot, int chunks, const double *cs) {
  switch (h->n / 3) {
    case 2: launch_gram_xz<2, NORM>(h, rot, chunks, cs); break;
    case 4: launch_gram_xz<4, NORM>(h, rot, chunks, cs); break;
    case 6: launch_gram_xz<6, NORM>(h, rot, chunks, cs); break;
    case 8: launch_gram_xz<8, NORM>(h, rot, chunks, cs); break;
    default: launch_gram_xz<10, NORM>(h, rot, chunks, cs); break;
  }
}
void gram_xz(mvsvd_handle *h, int norm, bool rot, int chunks, const double *cs) {
  if (norm == 1) gram_xz_m<1>(h, rot, chunks, cs);
  else gram_xz_m<2>(h, rot, chunks, cs);
}

int depth_step_fused(mvsvd_handle *h, int method, double f0, double *E, double *timings) {
  const int n = h->n, m = n / 3;
  const long long rows = h->base_rows;
  const size_t nn = (size_t)n * n;
  hipStream_t st = h->st;
  const double *dX = (const double *)h->dX;
  double *dz = (double *)h->dz;
  const size_t gs_stride = (size_t)std::max(256, h->n);  // (room for the finest grouping: one group per column)
  if (!h->dgs) MVBA_HIP(hipMalloc((void **)&h->dgs, sizeof(double) * (size_t)(GS_BLOCKS + 1) * gs_stride));
  double *cs = h->dgs + (size_t)GS_BLOCKS * gs_stride;
  const size_t dual_part = (size_t)512 * 4 * m * 256;  // k_dual_gram_mfma: a 16 x 16 tile per wave and image, at most 512 workgroups
  const size_t need = (size_t)DEPTH_BLOCKS + 8 + (size_t)m * (144 + 144 + 12 + 12) + dual_part + (size_t)FZ_MAXM * DEPTH_BLOCKS;
  if (!h->ddep) MVBA_HIP(hipMalloc((void **)&h->ddep, sizeof(double) * need));  // (the fused path is chosen per handle: `need` never changes)
  if (!h->ddflag) MVBA_HIP(hipMalloc((void **)&h->ddflag, sizeof(int) * (size_t)(m + 1)));
  double *Epart = h->ddep, *Eout = Epart + DEPTH_BLOCKS, *G12 = Eout + 8, *V12 = G12 + (size_t)m * 144, *colsum = V12 + (size_t)m * 144,
         *w12 = colsum + (size_t)m * 12, *gpart = w12 + (size_t)m * 12, *gsum = gpart + dual_part;
  if (method == 2 && !h->cs_valid) {  // the per-image scales of the depths the loop holds (first dual step, or after primary steps)
    const int gb = (int)std::max<long long>(1, std::min<long long>(GS_BLOCKS, rows / 64 + 1));
    hipLaunchKernelGGL(k_group_sumsq<double>, dim3(gb), dim3(256), 0, st, dX, (const double *)dz, rows, n, 3, h->dgs);
    hipLaunchKernelGGL(k_group_scale, dim3((m + 3) / 4), dim3(256), 0, st, h->dgs, gb, m, cs);
    h->cs_valid = true;
  }
  const int chunks = chunks_for(rows, n);
  hipEventRecord(h->ev[1], st);
  gram_xz(h, method, false, chunks, cs);
  hipEventRecord(h->ev[2], st);
  launch_jacobi(h, h->dV1, 1e-15);
  hipEventRecord(h->ev[3], st);
  gram_xz(h, method, true, chunks, cs);  // B = W V1 never leaves the LDS
  launch_jacobi(h, h->dMr, 1e-15);
  hipLaunchKernelGGL(k_rotate<double>, dim3((n + 63) / 64, (n + 63) / 64), dim3(256), 0, st, h->dV1, (long long)n, n, (const double *)nullptr, h->dMr, h->dV);
  hipEventRecord(h->ev[4], st);
  // eigenvalues -> host, sort, the rank-4 basis with the same deterministic sign as mvsvd_run
  std::vector<double> hG(nn), hV(nn), Mg((size_t)n * 4);
  MVBA_HIP(hipMemcpyAsync(hG.data(), h->dG, sizeof(double) * nn, hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipMemcpyAsync(hV.data(), h->dV, sizeof(double) * nn, hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipStreamSynchronize(st));
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return hG[(size_t)a * n + a] > hG[(size_t)b * n + b]; });
  double is[4];
  for (int i = 0; i < 4; ++i) {
    const int col = order[i];
    const double sg2 = hG[(size_t)col * n + col];
    if (method == 2 && !(sg2 > 0.0)) return fail(MVBA_ERR_SINGULAR, "measurement matrix has rank < 4");
    is[i] = 1.0 / std::sqrt(std::max(sg2, 1e-300));
    int big = 0;
    for (int c = 1; c < n; ++c)
      if (std::fabs(hV[(size_t)c * n + col]) > std::fabs(hV[(size_t)big * n + col])) big = c;
    const double sg = hV[(size_t)big * n + col] < 0.0 ? -1.0 : 1.0;
    for (int c = 0; c < n; ++c) Mg[(size_t)c * 4 + i] = sg * hV[(size_t)c * n + col];
  }
  MVBA_HIP(hipMemcpyAsync(h->dMr, Mg.data(), sizeof(double) * (size_t)n * 4, hipMemcpyHostToDevice, st));
  const int pgrid = (int)std::max<long long>(1, std::min<long long>(DEPTH_BLOCKS, (rows + 255) / 256));
  const size_t tile_bytes = sizeof(double) * 4 * 64 * (size_t)((n + m) | 1) + 16;
  hipEventRecord(h->ev[6], st);
  if (method == 1) {
    hipLaunchKernelGGL(k_primary_xz, dim3(pgrid), dim3(256), sizeof(double) * 12 * m + tile_bytes, st, dX, h->dMr, rows, m, dz, Epart);
    h->cs_valid = false;
  } else {
    MVBA_HIP(hipMemsetAsync(h->ddflag, 0, sizeof(int), st));
    const int mblocks = (int)std::max<long long>(1, std::min<long long>(512, (rows + GRAM_ROWS - 1) / GRAM_ROWS));  // two workgroups per CU
    const size_t mlds = sizeof(double) * (12 * (size_t)m + 4 * ((size_t)ROWS_PER_STEP * (n + m + 4) + 4 * ((size_t)ROWS_PER_STEP * (m | 1) + 8)));
    auto dual_gram = m == 2 ? k_dual_gram_mfma<2> : (m == 4 ? k_dual_gram_mfma<4> : (m == 6 ? k_dual_gram_mfma<6> : (m == 8 ? k_dual_gram_mfma<8> : k_dual_gram_mfma<10>)));
    hipLaunchKernelGGL(dual_gram, dim3(mblocks), dim3(256), mlds, st, dX, (const double *)dz, cs, h->dMr, is[0], is[1], is[2], is[3], rows, m, gpart);
    hipLaunchKernelGGL(k_dual_reduce_mfma, dim3((m * 90 + 3) / 4), dim3(256), 0, st, gpart, mblocks * 4, m, G12, colsum);
    hipLaunchKernelGGL(k_jacobi_small, dim3(m), dim3(JHB * JHB + JW * 6), 0, st, G12, V12, 12, 60, 1e-15, h->ddflag + 1);
    hipLaunchKernelGGL(k_dual_vec, dim3((m + 63) / 64), dim3(64), 0, st, G12, V12, colsum, m, w12, h->ddflag);
    hipLaunchKernelGGL(k_dual_apply_xz, dim3(pgrid), dim3(256), sizeof(double) * 24 * m + tile_bytes, st, dX, cs, is[0], is[1], is[2], is[3], h->dMr, w12, rows, m,
                       dz, Epart, gsum);
    hipLaunchKernelGGL(k_group_scale_t, dim3((m + 3) / 4), dim3(256), 0, st, gsum, pgrid, m, cs);  // the next iteration's scales
  }
  hipLaunchKernelGGL(k_depth_error, dim3(1), dim3(64), 0, st, Epart, pgrid, (double)rows * (double)m, f0, Eout);
  hipEventRecord(h->ev[7], st);
  MVBA_HIP(hipGetLastError());
  int fl = 0;
  MVBA_HIP(hipMemcpyAsync(E, Eout, sizeof(double), hipMemcpyDeviceToHost, st));
  if (method == 2) MVBA_HIP(hipMemcpyAsync(&fl, h->ddflag, sizeof(int), hipMemcpyDeviceToHost, st));
  MVBA_HIP(hipStreamSynchronize(st));
  h->loaded = false;  // (no re-weighted matrix was written: dW holds nothing that belongs to these depths)
  if (fl) return fail(MVBA_ERR_SINGULAR, "depth iteration: an image's 12 x 12 companion matrix has no positive eigenvalue");
  if (timings) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->ev[6], h->ev[7]); timings[0] = ms;  // depth update
    hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); timings[1] = ms;  // first Gram pass
    hipEventElapsedTime(&ms, h->ev[2], h->ev[3]); timings[2] = ms;  // Jacobi (first pass)
    timings[3] = 0.0;                                                // (no projection pass)
    int sw = 0;
    hipMemcpy(&sw, h->dsw, sizeof(int), hipMemcpyDeviceToHost);
    timings[4] = sw;
    hipEventElapsedTime(&ms, h->ev[3], h->ev[4]); timings[5] = ms;  // refinement pass (rotate + Gram fused, Jacobi, V1 V2)
  }
  return MVBA_OK;
}

}  // namespace


extern "C" {

int mvsvd_create(int64_t max_rows, int32_t n_cols, int32_t dtype, int32_t device, mvsvd_handle **out) {
  if (!out) return fail(MVBA_ERR_BADARG, "null argument");
  if (max_rows < 1 || n_cols < 1 || n_cols > MVSVD_MAX_COLS)
    return fail(MVBA_ERR_BADARG, "need max_rows >= 1 and 1 <= n_cols <= " + std::to_string(MVSVD_MAX_COLS) + " (three rows of W per image at the engine's camera limit)");
  if (dtype != 0 && dtype != 1) return fail(MVBA_ERR_BADARG, "dtype must be 0 (float32) or 1 (float64)");
  if (device >= 0) MVBA_HIP(hipSetDevice(device));
  mvsvd_handle *h = new mvsvd_handle();
  MVBA_HIP(hipGetDevice(&h->device));
  h->dtype = dtype; h->n = n_cols; h->max_rows = max_rows;
  // (MVSVD_WIDE_MIN: experiments -- the column count above which the block iteration takes over, never below its own width)
  const int wide_min = getenv("MVSVD_WIDE_MIN") ? std::max(WB, atoi(getenv("MVSVD_WIDE_MIN"))) : WIDE_MIN;
  const bool wide = n_cols > wide_min, dense = n_cols <= JACOBI_MAX;  // beyond JACOBI_MAX no n x n matrix at all: see run_wide
  h->wide = wide;
  const size_t el = dtype ? 8 : 4, nn = dense ? (size_t)n_cols * n_cols : (size_t)WB * WB;
  const int gram_n = dense ? n_cols : WB;  // the Gram kernels run on the workspace's matrix and on the wide path's 32-column blocks
  const int n_tiles = (gram_n + GT - 1) / GT, n_pairs = n_tiles * (n_tiles + 1) / 2;
  h->chunks = chunks_for(max_rows, gram_n);
  // partial tiles of the Gram kernels: the workspace's own matrix and / or the wide path's blocks (N x 32 and n x 32: three tile pairs)
  size_t part_tiles = (size_t)h->chunks * n_pairs, part2_tiles = (size_t)GRAM_SLICES * n_pairs;
  if (wide) {
    part_tiles = std::max(part_tiles, (size_t)3 * std::max(chunks_for(max_rows, WB), chunks_for(n_cols, WB)));
    part2_tiles = std::max(part2_tiles, (size_t)3 * GRAM_SLICES);
  }
#define SVD_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { mvsvd_destroy(h); return fail(MVBA_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
  SVD_TRY(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  for (auto &e : h->ev) SVD_TRY(hipEventCreate(&e));
  SVD_TRY(hipMalloc(&h->dW, el * (size_t)max_rows * n_cols));
  SVD_TRY(hipMalloc((void **)&h->dG, sizeof(double) * nn));
  SVD_TRY(hipMalloc((void **)&h->dV, sizeof(double) * nn));
  SVD_TRY(hipMalloc((void **)&h->dV1, sizeof(double) * nn));
  SVD_TRY(hipMalloc((void **)&h->dMr, sizeof(double) * std::max(nn, (size_t)4 * n_cols)));
  SVD_TRY(hipMalloc((void **)&h->dsum, sizeof(double) * n_cols * COLSUM_SLICES));
  SVD_TRY(hipMalloc((void **)&h->dmu, sizeof(double) * n_cols));
  SVD_TRY(hipMalloc((void **)&h->dsw, sizeof(int)));
  SVD_TRY(hipMalloc((void **)&h->dpart, sizeof(double) * part_tiles * 256));
  SVD_TRY(hipMalloc((void **)&h->dpart2, sizeof(double) * part2_tiles * 256));
  if (wide) {
    const size_t nb = (size_t)n_cols * WB;
    // row chunks of Z = W^T B: enough workgroups for the chip beside the n / 64 column blocks, a multiple of 64 rows each
    const long long col_blocks = (n_cols + WT - 1) / WT, want = std::max<long long>(1, 2048 / col_blocks);
    h->zrows_per_chunk = std::max<long long>(WT, ((max_rows + want - 1) / want + WT - 1) / WT * WT);
    h->zchunks = (int)((max_rows + h->zrows_per_chunk - 1) / h->zrows_per_chunk);
    SVD_TRY(hipMalloc((void **)&h->dQ, sizeof(double) * nb));
    SVD_TRY(hipMalloc((void **)&h->dZ, sizeof(double) * nb));
    SVD_TRY(hipMalloc((void **)&h->dQ2, sizeof(double) * nb));
    SVD_TRY(hipMalloc((void **)&h->dBw, sizeof(double) * (size_t)max_rows * WB));
    SVD_TRY(hipMalloc((void **)&h->dB2, sizeof(double) * (size_t)max_rows * WB));
    SVD_TRY(hipMalloc((void **)&h->dzpart, sizeof(double) * (size_t)h->zchunks * nb));
    SVD_TRY(hipMalloc((void **)&h->dsmall, sizeof(double) * ((size_t)5 * WB * WB + 2 * WB + (size_t)WB * ((n_cols + 255) / 256))));
    SVD_TRY(hipMalloc((void **)&h->dwflag, sizeof(int) * 2 * WB));
  }
  SVD_TRY(hipFuncSetAttribute((const void *)k_jacobi<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  SVD_TRY(hipFuncSetAttribute((const void *)k_project<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  SVD_TRY(hipFuncSetAttribute((const void *)k_rotate_rows<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SVD_TRY(hipFuncSetAttribute((const void *)k_rotate_rows<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SVD_TRY(hipFuncSetAttribute((const void *)k_project<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  for (const void *f : {(const void *)k_scale_rows_tiled<float>, (const void *)k_scale_rows_tiled<double>, (const void *)k_depth_primary<float, true>,
                        (const void *)k_depth_primary<double, true>, (const void *)k_dual_apply<float, true>, (const void *)k_dual_apply<double, true>,
                        (const void *)k_dual_gram<float>, (const void *)k_dual_gram<double>, (const void *)k_primary_xz, (const void *)k_dual_gram_mfma<2>, (const void *)k_dual_gram_mfma<4>, (const void *)k_dual_gram_mfma<6>, (const void *)k_dual_gram_mfma<8>,
                        (const void *)k_dual_gram_mfma<10>,
                        (const void *)k_dual_apply_xz, (const void *)k_depth_primary_wide<float>, (const void *)k_depth_primary_wide<double>,
                        (const void *)k_dual_apply_wide<float>, (const void *)k_dual_apply_wide<double>})
    SVD_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DEPTH_LDS_MAX));
#undef SVD_TRY
  *out = h;
  return MVBA_OK;
}

void mvsvd_destroy(mvsvd_handle *h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->st) hipStreamSynchronize(h->st);
  for (void *p : {h->dW, h->dS, (void *)h->dG, (void *)h->dV, (void *)h->dV1, (void *)h->dsum, (void *)h->dMr, (void *)h->dmu,
                  (void *)h->dsw, (void *)h->dpart, (void *)h->dpart2, (void *)h->dB, h->dX, h->dz, h->dstage, (void *)h->dgs, (void *)h->ddep, (void *)h->ddflag,
                  (void *)h->dQ, (void *)h->dZ, (void *)h->dQ2, (void *)h->dBw, (void *)h->dB2, (void *)h->dzpart, (void *)h->dsmall, (void *)h->dwflag})
    if (p) hipFree(p);
  for (auto &e : h->ev)
    if (e) hipEventDestroy(e);
  if (h->st) hipStreamDestroy(h->st);
  delete h;
}

int mvsvd_load(mvsvd_handle *h, const void *Wt, int64_t n_rows) {
  if (!h || !Wt) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_rows < 1 || n_rows > h->max_rows) return fail(MVBA_ERR_BADARG, "n_rows outside the workspace (1 .. max_rows)");
  MVBA_HIP(hipSetDevice(h->device));
  hipEventRecord(h->ev[0], h->st);
  MVBA_HIP(hipMemcpyAsync(h->dW, Wt, (h->dtype ? 8 : 4) * (size_t)n_rows * h->n, hipMemcpyHostToDevice, h->st));
  hipEventRecord(h->ev[1], h->st);
  MVBA_HIP(hipStreamSynchronize(h->st));
  float ms = 0.f;
  hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
  h->h2d_ms = ms;
  h->n_rows = n_rows;
  h->loaded = true;
  h->wide_warm = h->wide_have_q = false;  // a new matrix: the block iteration starts from its fixed block
  return MVBA_OK;
}

int mvsvd_load_images(mvsvd_handle *h, const void *const *xy, int32_t n_images, int64_t n_rows, int32_t src_dtype) {
  if (!h || !xy) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_rows < 1 || n_rows > h->max_rows) return fail(MVBA_ERR_BADARG, "n_rows outside the workspace (1 .. max_rows)");
  if (n_images < 1 || 2 * (long long)n_images != h->n) return fail(MVBA_ERR_BADARG, "the workspace must have 2 columns per image");
  if (src_dtype != 0 && src_dtype != 1) return fail(MVBA_ERR_BADARG, "src_dtype must be 0 (float32) or 1 (float64)");
  for (int k = 0; k < n_images; ++k)
    if (!xy[k]) return fail(MVBA_ERR_BADARG, "null image array");
  MVBA_HIP(hipSetDevice(h->device));
  if (!h->dstage) MVBA_HIP(hipMalloc(&h->dstage, 16 * (size_t)h->max_rows));
  const size_t bytes = (src_dtype ? 16 : 8) * (size_t)n_rows;
  const unsigned grid = (unsigned)((n_rows + 255) / 256);
  hipEventRecord(h->ev[0], h->st);
  for (int k = 0; k < n_images; ++k) {  // image after image in stream order through the one staging buffer
    MVBA_HIP(hipMemcpyAsync(h->dstage, xy[k], bytes, hipMemcpyHostToDevice, h->st));
    if (src_dtype == 0 && h->dtype == 0)
      hipLaunchKernelGGL((k_image_cols<float, float>), dim3(grid), dim3(256), 0, h->st, (const float *)h->dstage, (long long)n_rows, h->n, 2 * k, (float *)h->dW);
    else if (src_dtype == 0)
      hipLaunchKernelGGL((k_image_cols<float, double>), dim3(grid), dim3(256), 0, h->st, (const float *)h->dstage, (long long)n_rows, h->n, 2 * k, (double *)h->dW);
    else if (h->dtype == 0)
      hipLaunchKernelGGL((k_image_cols<double, float>), dim3(grid), dim3(256), 0, h->st, (const double *)h->dstage, (long long)n_rows, h->n, 2 * k, (float *)h->dW);
    else
      hipLaunchKernelGGL((k_image_cols<double, double>), dim3(grid), dim3(256), 0, h->st, (const double *)h->dstage, (long long)n_rows, h->n, 2 * k, (double *)h->dW);
  }
  hipEventRecord(h->ev[1], h->st);
  MVBA_HIP(hipGetLastError());
  MVBA_HIP(hipStreamSynchronize(h->st));
  float ms = 0.f;
  hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
  h->h2d_ms = ms;
  h->n_rows = n_rows;
  h->loaded = true;
  h->wide_warm = h->wide_have_q = false;
  return MVBA_OK;
}

int mvsvd_run(mvsvd_handle *h, int32_t n_rank, int32_t center, void *M, void *sigma, void *S, void *means, double *timings_ms) {
  if (!h || !M || !sigma || !S) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->loaded) return fail(MVBA_ERR_STATE, "mvsvd_run before mvsvd_load");
  if (n_rank < 1 || n_rank > h->n) return fail(MVBA_ERR_BADARG, "need 1 <= n_rank <= n_cols");
  MVBA_HIP(hipSetDevice(h->device));
  if (h->dtype == 0) return run<float>(h, n_rank, center, (float *)M, (float *)sigma, (float *)S, (float *)means, timings_ms);
  return run<double>(h, n_rank, center, (double *)M, (double *)sigma, (double *)S, (double *)means, timings_ms);
}

int mvsvd_load_base(mvsvd_handle *h, const void *X, int64_t n_rows) {
  if (!h || !X) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_rows < 1 || n_rows > h->max_rows) return fail(MVBA_ERR_BADARG, "n_rows outside the workspace (1 .. max_rows)");
  MVBA_HIP(hipSetDevice(h->device));
  const size_t el = h->dtype ? 8 : 4;
  if (!h->dX) MVBA_HIP(hipMalloc(&h->dX, el * (size_t)h->max_rows * h->n));
  MVBA_HIP(hipMemcpyAsync(h->dX, X, el * (size_t)n_rows * h->n, hipMemcpyHostToDevice, h->st));
  MVBA_HIP(hipStreamSynchronize(h->st));
  h->base_rows = n_rows;
  h->base_loaded = true;
  h->wide_warm = h->wide_have_q = false;
  h->loaded = false;  // dW holds nothing derived from this base yet
  return MVBA_OK;
}

int mvsvd_load_base_images(mvsvd_handle *h, const double *const *xy, int32_t n_images, int64_t n_rows, double f0) {
  if (!h || !xy) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_rows < 1 || n_rows > h->max_rows) return fail(MVBA_ERR_BADARG, "n_rows outside the workspace (1 .. max_rows)");
  if (n_images < 1 || 3 * (long long)n_images != h->n) return fail(MVBA_ERR_BADARG, "the workspace must have 3 columns per image");
  if (!(f0 != 0.0) || !std::isfinite(f0)) return fail(MVBA_ERR_BADARG, "f0 must be finite and non-zero");
  for (int k = 0; k < n_images; ++k)
    if (!xy[k]) return fail(MVBA_ERR_BADARG, "null image array");
  MVBA_HIP(hipSetDevice(h->device));
  const size_t el = h->dtype ? 8 : 4;
  if (!h->dX) MVBA_HIP(hipMalloc(&h->dX, el * (size_t)h->max_rows * h->n));
  // staged through dW (16 bytes a row <= el * n: there are at least 6 columns), image after image in stream order
  if (el * (size_t)h->n < 16) return fail(MVBA_ERR_BADARG, "the workspace is too narrow to stage an image");
  const unsigned grid = (unsigned)((n_rows + 255) / 256);
  for (int k = 0; k < n_images; ++k) {
    MVBA_HIP(hipMemcpyAsync(h->dW, xy[k], 16 * (size_t)n_rows, hipMemcpyHostToDevice, h->st));
    if (h->dtype)
      hipLaunchKernelGGL(k_base_image<double>, dim3(grid), dim3(256), 0, h->st, (const double2 *)h->dW, (long long)n_rows, h->n, 3 * k, f0, (double *)h->dX);
    else
      hipLaunchKernelGGL(k_base_image<float>, dim3(grid), dim3(256), 0, h->st, (const double2 *)h->dW, (long long)n_rows, h->n, 3 * k, f0, (float *)h->dX);
  }
  MVBA_HIP(hipGetLastError());
  MVBA_HIP(hipStreamSynchronize(h->st));
  h->base_rows = n_rows;
  h->base_loaded = true;
  h->wide_warm = h->wide_have_q = false;
  h->loaded = false;  // dW was the staging buffer
  return MVBA_OK;
}

int mvsvd_run_scaled(mvsvd_handle *h, const void *z, int32_t group, int32_t norm, int32_t n_rank, void *M, void *sigma, void *S,
                     double *timings_ms) {
  if (!h || !M || !sigma || !S) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->base_loaded) return fail(MVBA_ERR_STATE, "mvsvd_run_scaled before mvsvd_load_base");
  // z == NULL: the depths a device depth loop left in the workspace (same grouping) -- nothing is uploaded at all
  if (!z && (h->depth_group != group || !h->dz)) return fail(MVBA_ERR_STATE, "mvsvd_run_scaled without depths: no device depth loop of this grouping has run");
  if (group < 1 || h->n % group) return fail(MVBA_ERR_BADARG, "group must divide n_cols");
  if (norm < 0 || norm > 2) return fail(MVBA_ERR_BADARG, "norm must be 0 (none), 1 (unit rows) or 2 (column groups by their squared norm)");
  if (n_rank < 1 || n_rank > h->n) return fail(MVBA_ERR_BADARG, "need 1 <= n_rank <= n_cols");
  const int ng = h->n / group;
  MVBA_HIP(hipSetDevice(h->device));
  const size_t el = h->dtype ? 8 : 4;
  if (!h->dz) MVBA_HIP(hipMalloc(&h->dz, el * (size_t)h->max_rows * h->n));  // (room for any grouping: a later call may ask for a finer one)
  hipEventRecord(h->ev[0], h->st);
  if (z) MVBA_HIP(hipMemcpyAsync(h->dz, z, el * (size_t)h->base_rows * ng, hipMemcpyHostToDevice, h->st));  // the only upload of the call
  hipEventRecord(h->ev[1], h->st);
  if (z) h->depth_group = 0;  // (the caller's depths replace whatever a depth loop held)
  int rc = scale_base_into_w(h, group, norm);
  if (rc) return rc;
  MVBA_HIP(hipStreamSynchronize(h->st));
  float ms = 0.f;
  hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
  h->h2d_ms = ms;
  if (h->dtype == 0) return run<float>(h, n_rank, 0, (float *)M, (float *)sigma, (float *)S, (float *)nullptr, timings_ms);
  return run<double>(h, n_rank, 0, (double *)M, (double *)sigma, (double *)S, (double *)nullptr, timings_ms);
}

int mvsvd_depth_begin(mvsvd_handle *h, int32_t group) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  if (!h->base_loaded) return fail(MVBA_ERR_STATE, "mvsvd_depth_begin before mvsvd_load_base");
  if (group != 3 || h->n % 3) return fail(MVBA_ERR_BADARG, "the depth iteration works on homogeneous image coordinates: group = 3, n_cols = 3 m");
  if (h->n / 3 > DEPTH_MAX_IMAGES) return fail(MVBA_ERR_BADARG, "at most " + std::to_string(DEPTH_MAX_IMAGES) + " images in the device depth loop (two 12-double tables per image in a workgroup's LDS)");
  MVBA_HIP(hipSetDevice(h->device));
  const size_t el = h->dtype ? 8 : 4;
  const int ng = h->n / 3;
  if (!h->dz) MVBA_HIP(hipMalloc(&h->dz, el * (size_t)h->max_rows * h->n));  // (shared with mvsvd_run_scaled, whose groups may be finer)
  const long long cnt = h->base_rows * ng;
  const int grid = (int)std::max<long long>(1, std::min<long long>(4096, (cnt + 255) / 256));
  if (h->dtype == 0) hipLaunchKernelGGL(k_fill<float>, dim3(grid), dim3(256), 0, h->st, (float *)h->dz, cnt, 1.0f);
  else hipLaunchKernelGGL(k_fill<double>, dim3(grid), dim3(256), 0, h->st, (double *)h->dz, cnt, 1.0);
  MVBA_HIP(hipGetLastError());
  h->depth_group = 3;
  h->cs_valid = false;
  return MVBA_OK;
}

int mvsvd_depth_step(mvsvd_handle *h, int32_t method, double f0, double *E, double *timings_ms) {
  if (!h || !E) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->base_loaded || h->depth_group != 3) return fail(MVBA_ERR_STATE, "mvsvd_depth_step before mvsvd_depth_begin");
  if (method != 1 && method != 2) return fail(MVBA_ERR_BADARG, "method must be 1 (primary) or 2 (dual)");
  if (h->n < 6) return fail(MVBA_ERR_BADARG, "the rank-4 depth iteration needs at least 2 images (3 m >= 4 columns, as the reference's)");
  MVBA_HIP(hipSetDevice(h->device));
  if (fused_depth_ok(h)) return depth_step_fused(h, method, f0, E, timings_ms);
  return h->dtype == 0 ? depth_step<float>(h, method, f0, E, timings_ms) : depth_step<double>(h, method, f0, E, timings_ms);
}

int mvsvd_depth_read(mvsvd_handle *h, void *z) {
  if (!h || !z) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->base_loaded || h->depth_group != 3) return fail(MVBA_ERR_STATE, "mvsvd_depth_read before mvsvd_depth_begin");
  MVBA_HIP(hipSetDevice(h->device));
  MVBA_HIP(hipMemcpyAsync(z, h->dz, (h->dtype ? 8 : 4) * (size_t)h->base_rows * (h->n / 3), hipMemcpyDeviceToHost, h->st));
  MVBA_HIP(hipStreamSynchronize(h->st));
  return MVBA_OK;
}

int mvsvd_factorize(const void *Wt, int64_t n_rows, int32_t n_cols, int32_t dtype, int32_t n_rank, int32_t center, void *M,
                    void *sigma, void *S, void *means, double *timings_ms, int32_t device) {
  if (!Wt || !M || !sigma || !S) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_rows < 1 || n_cols < 1 || n_rank < 1 || n_rank > n_cols) return fail(MVBA_ERR_BADARG, "need n_rows >= 1, 1 <= n_rank <= n_cols");
  mvsvd_handle *h = nullptr;
  int rc = mvsvd_create(n_rows, n_cols, dtype, device, &h);
  if (rc) return rc;
  rc = mvsvd_load(h, Wt, n_rows);
  if (!rc) rc = mvsvd_run(h, n_rank, center, M, sigma, S, means, timings_ms);
  mvsvd_destroy(h);
  return rc;
}

}  // extern "C"
