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

constexpr int GT = 32;  // Gram output tile (GT x GT per block), 256 threads = 16x16 with 2x2 each
constexpr int GR = 32;  // rows staged per step

template <typename T>
__global__ __launch_bounds__(256) void k_gram(const T *__restrict__ Wt, long long n_rows, int n, int n_tiles,
                                              double *__restrict__ G) {
  // blockIdx.x enumerates upper tile pairs (ti <= tj); blockIdx.y strides over row chunks
  int ti = 0, rem = blockIdx.x;
  while (rem >= n_tiles - ti) { rem -= n_tiles - ti; ++ti; }
  const int tj = ti + rem;
  __shared__ double As[GR][GT + 1], Bs[GR][GT + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[2][2] = {{0, 0}, {0, 0}};
  const long long step = (long long)gridDim.y * GR;
  for (long long r0 = (long long)blockIdx.y * GR; r0 < n_rows; r0 += step) {
    for (int q = threadIdx.x; q < GR * GT; q += 256) {
      const int rr = q / GT, cc = q % GT;
      const long long row = r0 + rr;
      const int ca = ti * GT + cc, cb = tj * GT + cc;
      As[rr][cc] = (row < n_rows && ca < n) ? (double)Wt[row * n + ca] : 0.0;
      Bs[rr][cc] = (row < n_rows && cb < n) ? (double)Wt[row * n + cb] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int rr = 0; rr < GR; ++rr) {
      const double a0 = As[rr][ty], a1 = As[rr][ty + 16], b0 = Bs[rr][tx], b1 = Bs[rr][tx + 16];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1;
      acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gi = ti * GT + ty + 16 * i, gj = tj * GT + tx + 16 * j;
      if (gi < n && gj < n) atomicAdd(&G[(size_t)gi * n + gj], acc[i][j]);
    }
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
__global__ __launch_bounds__(1024) void k_jacobi(double *__restrict__ A, double *__restrict__ V, int n, int max_sweeps,
                                                 double tol, int *__restrict__ sweeps_done) {
  extern __shared__ double sm[];  // cs[np/2][2], pairs as ints after
  const int np = (n + 1) & ~1, half = np / 2;
  double *rc = sm, *rs = sm + half;
  int *pp = reinterpret_cast<int *>(sm + 2 * half), *qq = pp + half;
  __shared__ double s_off;
  const int tid = threadIdx.x, nt = blockDim.x;
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
}

// S[i][row] = sum_c Mr[c][i] (Wt[row][c] - mu[c])   (thread per row; M in LDS)
template <typename T>
__global__ __launch_bounds__(256) void k_project(const T *__restrict__ Wt, long long n_rows, int n, int r,
                                                 const double *__restrict__ Mr, const double *__restrict__ mu,
                                                 T *__restrict__ S) {
  extern __shared__ double sm[];  // Mr [n][r], mu [n]
  double *sM = sm, *smu = sm + (size_t)n * r;
  for (int q = threadIdx.x; q < n * r; q += blockDim.x) sM[q] = Mr[q];
  for (int q = threadIdx.x; q < n; q += blockDim.x) smu[q] = mu ? mu[q] : 0.0;
  __syncthreads();
  for (long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows;
       row += (long long)gridDim.x * blockDim.x) {
    double acc[4] = {0, 0, 0, 0};
    const T *w = Wt + row * n;
    for (int c = 0; c < n; ++c) {
      const double x = (double)w[c] - smu[c];
      for (int i = 0; i < 4; ++i)
        if (i < r) acc[i] += x * sM[c * r + i];
    }
    for (int i = 0; i < r; ++i) S[(size_t)i * n_rows + row] = (T)acc[i];
  }
}

template <typename T>
int factorize(const T *Wt, long long n_rows, int n, int n_rank, int center, T *M, T *sigma, T *S, T *means,
              double *timings) {
  hipStream_t st = nullptr;
  MVBA_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  T *dW = nullptr, *dS = nullptr;
  double *dG = nullptr, *dV = nullptr, *dsum = nullptr, *dMr = nullptr, *dmu = nullptr;
  int *dsw = nullptr;
  hipEvent_t ev[6];
  for (auto &e : ev) hipEventCreate(&e);
  int rc = MVBA_OK;
  auto cleanup = [&]() {
    for (void *p : {(void *)dW, (void *)dS, (void *)dG, (void *)dV, (void *)dsum, (void *)dMr, (void *)dmu, (void *)dsw})
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
  hipEventRecord(ev[0], st);
  SVD_HIP(hipMemcpyAsync(dW, Wt, sizeof(T) * (size_t)n_rows * n, hipMemcpyHostToDevice, st));
  hipEventRecord(ev[1], st);
  SVD_HIP(hipMemsetAsync(dG, 0, sizeof(double) * nn, st));
  SVD_HIP(hipMemsetAsync(dsum, 0, sizeof(double) * n, st));
  const int n_tiles = (n + GT - 1) / GT, n_pairs = n_tiles * (n_tiles + 1) / 2;
  const int chunks = (int)std::max<long long>(1, std::min<long long>((n_rows + GR - 1) / GR, std::max(1, 4096 / n_pairs)));
  hipLaunchKernelGGL(k_gram<T>, dim3(n_pairs, chunks), dim3(256), 0, st, dW, n_rows, n, n_tiles, dG);
  if (center) {
    const int cy = (int)std::max<long long>(1, std::min<long long>(256, n_rows / 4096));
    hipLaunchKernelGGL(k_colsum<T>, dim3(n, cy), dim3(256), 0, st, dW, n_rows, n, dsum);
  }
  hipLaunchKernelGGL(k_gram_finish, dim3((n + 127) / 128, n), dim3(128), 0, st, dG, dsum, n, n_rows, center);
  hipEventRecord(ev[2], st);
  const int np = (n + 1) & ~1;
  const size_t jl = sizeof(double) * np + sizeof(int) * np;
  hipLaunchKernelGGL(k_jacobi, dim3(1), dim3(1024), jl, st, dG, dV, n, 60, 1e-15, dsw);
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
  hipLaunchKernelGGL(k_project<T>, dim3(pgrid), dim3(256), sizeof(double) * ((size_t)n * n_rank + n), st, dW, n_rows, n,
                     n_rank, dMr, center ? dmu : nullptr, dS);
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
  MVBA_HIP(hipFuncSetAttribute((const void *)k_project<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  MVBA_HIP(hipFuncSetAttribute((const void *)k_project<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  if (dtype == 0)
    return factorize<float>((const float *)Wt, n_rows, n_cols, n_rank, center, (float *)M, (float *)sigma, (float *)S,
                            (float *)means, timings_ms);
  return factorize<double>((const double *)Wt, n_rows, n_cols, n_rank, center, (double *)M, (double *)sigma,
                           (double *)S, (double *)means, timings_ms);
}
