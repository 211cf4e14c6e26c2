// Relative error of the raw v_rsq_f64 / v_rcp_f64 results (no Newton step) and after one step, over 1M inputs spread
// across 1e-12 .. 1e12:  hipcc --offload-arch=gfx950 -O3 rsq_accuracy.hip -o rsq_accuracy && ./rsq_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  o[4 * i] = y;
  y = y * (1.5 - 0.5 * v * y * y);
  o[4 * i + 1] = y;
  double r = __builtin_amdgcn_rcp(v);
  o[4 * i + 2] = r;
  r = r * (2.0 - v * r);
  o[4 * i + 3] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(4 * (size_t)n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = std::pow(10.0, 24.0 * u - 12.0);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 4 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double e[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
    const long double ref[4] = {rs, rs, rc, rc};
    for (int q = 0; q < 4; ++q) e[q] = std::fmax(e[q], (double)fabsl((o[4 * (size_t)i + q] - ref[q]) / ref[q]));
  }
  printf("max relative error: v_rsq_f64 raw %.3e, + one Newton step %.3e;  v_rcp_f64 raw %.3e, + one Newton step %.3e\n", e[0], e[1], e[2], e[3]);
  return 0;
}
