// How fast is a single wave on an otherwise idle MI355X?  (The Cholesky panel kernels are one-wave
// serial chains.)  Measures, with the shader clock (s_memtime) and the 100 MHz wall clock:
//   1. a dependent v_fma_f64 chain,  2. independent f64 MFMA 16x16x4 issue,  3. dependent MFMA chain,
//   4. the v_readlane x2 + v_fma_f64 pattern of the tile factorisation.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ void k(int mode, int iters, unsigned long long *out, double *sink) {
  const int lane = threadIdx.x;
  double x = 1.0 + 1e-9 * lane, y = 1.0 - 1e-12 * lane;
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  unsigned long long w0 = wall_clock64(), t0 = __builtin_readcyclecounter();
  if (mode == 0) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) x = __builtin_fma(x, y, 1e-9);
    }
  } else if (mode == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
      }
    }
  } else if (mode == 2) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) x = __builtin_fma(-y, readlane_d(y, u), x);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (lane == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
  sink[lane] = x + a0[0] + a1[1] + a2[2] + a3[3];
}

int main() {
  unsigned long long *out; double *sink;
  hipMalloc(&out, 16); hipMalloc(&sink, 64 * 8);
  const char *names[] = {"dependent v_fma_f64", "independent f64 MFMA 16x16x4 (4 accumulators)", "dependent f64 MFMA chain",
                         "2 x v_readlane + v_fma_f64 (dependent)"};
  const int iters = 4096;
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, iters, out, sink);
      hipDeviceSynchronize();
      unsigned long long h[2];
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      const double ops = 16.0 * iters, ns = h[1] * 10.0;  // wall clock: 100 MHz
      if (rep) printf("%-48s %7.2f shader ticks/op  %7.2f ns/op\n", names[mode], h[0] / ops, ns / ops);
    }
  return 0;
}
