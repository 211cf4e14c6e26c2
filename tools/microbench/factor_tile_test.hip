// Unit check of factor_tile (csrc/mvba.hip) on one wave: random SPD 32x32 tile -> L^-T, compared with a host Cholesky.
// per-phase stamps of the last call (panel p: 1+4p after a, 2+4p after b, 3+4p after c, 4+4p after d)
__device__ long long g_ft[20];
#define FT_STAMP(i) do { if (lane == 0) g_ft[i] = clock64(); } while (0)
#include "../../3d-reconstruction-from-multi-view-exp_amd/csrc/mvba.hip"
#include <cstdio>
#include <random>

__global__ __launch_bounds__(64) void k_test(const double *A, double *Zout, int *ok) {
  __shared__ double tile[NB][TS];
  __shared__ double Zt[NB][TS];
  __shared__ double Xb[64 * 9];
  const int lane = threadIdx.x;
  for (int e = lane; e < NB * NB; e += 64) {
    const int i = e / NB, j = e % NB;
    tile[i][j] = (j <= i) ? A[i * NB + j] : 0.0;
  }
  __syncthreads();
  bool good = true;
  long long best = 1LL << 60;
  for (int rep = 0; rep < 4; ++rep) {  // (factor_tile leaves `tile` untouched: it works in registers)
    __syncthreads();
    const long long t0 = clock64();
    good = factor_tile(tile, Zt, Xb, lane, false, nullptr, NB);
    const long long t1 = clock64();
    best = min(best, t1 - t0);
  }
  __syncthreads();
  for (int e = lane; e < NB * NB; e += 64) Zout[e] = Zt[e / NB][e % NB];
  if (lane == 0) { *ok = good; ok[1] = (int)best; }
}

int main() {
  std::mt19937 rng(1);
  std::normal_distribution<double> nd;
  const int n = NB;
  std::vector<double> G(n * n), A(n * n, 0.0), L(n * n, 0.0), Zi(n * n, 0.0);
  for (auto &x : G) x = nd(rng);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = (i == j) ? n : 0.0;
      for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k];
      A[i * n + j] = s;
    }
  for (int j = 0; j < n; ++j) {  // host Cholesky
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = std::sqrt(d);
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / L[j * n + j];
    }
  }
  for (int c = 0; c < n; ++c) {  // Zi = L^-1 (column c), then compare Zt = Zi^T
    for (int i = 0; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * Zi[k * n + c];
      Zi[i * n + c] = s / L[i * n + i];
    }
  }
  double *dA, *dZ; int *dok;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dZ, n * n * 8); hipMalloc(&dok, 8);
  hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dZ, dok);
  std::vector<double> Z(n * n); int ok = 0, cyc = 0;
  hipMemcpy(Z.data(), dZ, n * n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
  hipMemcpy(&cyc, dok + 1, 4, hipMemcpyDeviceToHost);
  printf("factor_tile: %d shader cycles (best of 4)\n", cyc);
  long long ft[20];
  hipMemcpyFromSymbol(ft, HIP_SYMBOL(g_ft), sizeof(ft));
  for (int p = 0; p < 4; ++p)
    printf("  panel %d: a (to registers) %lld, b (8 pivots) %lld, c (back to LDS) %lld, d (MFMA update) %lld\n", p, ft[1 + 4 * p] - ft[4 * p],
           ft[2 + 4 * p] - ft[1 + 4 * p], ft[3 + 4 * p] - ft[2 + 4 * p], ft[4 + 4 * p] - ft[3 + 4 * p]);
  double err = 0; int wi = 0, wj = 0;
  for (int k = 0; k < n; ++k)
    for (int c = 0; c < n; ++c) {
      const double e = std::fabs(Z[k * n + c] - Zi[c * n + k]);
      if (e > err) { err = e; wi = k; wj = c; }
    }
  printf("ok=%d max |Zt - L^-T| = %.3e at (%d,%d)\n", ok, err, wi, wj);
  for (int k = 0; k < 4; ++k) { for (int c = 0; c < 10; ++c) printf("%9.5f ", Z[k * n + c]); printf("\n"); }
  for (int bi = 0; bi < 4; ++bi) { for (int bj = 0; bj < 4; ++bj) { double e = 0; for (int k = 8 * bi; k < 8 * bi + 8; ++k) for (int c = 8 * bj; c < 8 * bj + 8; ++c) e = std::max(e, std::fabs(Z[k * n + c] - Zi[c * n + k])); printf("%9.2e ", e); } printf("  <- block errors (rows of Zt x column panels)\n"); }
  printf("wrong entries (row of Zt: columns):\n"); for (int k = 0; k < n; ++k) { int any = 0; for (int c = 0; c < n; ++c) if (std::fabs(Z[k * n + c] - Zi[c * n + k]) > 1e-10) { if (!any) printf("  row %2d:", k); any = 1; printf(" %d", c); } if (any) printf("\n"); }
  printf("expected:\n");
  for (int k = 0; k < 4; ++k) { for (int c = 0; c < 10; ++c) printf("%9.5f ", Zi[c * n + k]); printf("\n"); }
  return 0;
}
