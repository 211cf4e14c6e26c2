// Which LDS strip layout makes the 9 ds_add_f64 of a Schur pass cheapest?  63 lanes = 7 slots
// (random cameras out of 100, ascending) x 9 columns j; 9 atomics (i = 0..8) per pass.
//   layout 0: [i][9*cam + j]            (row stride W = 900)
//   layout 1: [(9*cam + j)*9 + i]       (current kernel)
//   layout 2: [(9*cam + j)*P + i], P = 10, 11, 12, 16
//   layout 3: [cam][i][j] with camera block stride Q (81 .. 96)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

__global__ void k(int iters, int mode, int P, const int* __restrict__ cams, unsigned long long* out, double* sink, int lds_doubles) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < lds_doubles; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / 9, j = lane % 9;
  double v = 1.0 + lane;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int cam = cams[((wave * 131 + it) % 512) * 8 + (slot < 7 ? slot : 0)];
    if (slot < 7) {
      int base, stride;
      if (mode == 0) { base = 9 * cam + j; stride = 900; }
      else if (mode == 1 || mode == 2) { base = (9 * cam + j) * P; stride = 1; }
      else { base = cam * P + j; stride = 9; }
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicAdd(&lds[base + i * stride], v);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[5];
}

int main() {
  const int iters = 4000;
  std::mt19937 rng(1);
  std::vector<int> cams(512 * 8);
  for (int s = 0; s < 512; ++s) {
    std::vector<int> c(100); for (int i = 0; i < 100; ++i) c[i] = i;
    std::shuffle(c.begin(), c.end(), rng); std::sort(c.begin(), c.begin() + 7);
    for (int i = 0; i < 8; ++i) cams[s * 8 + i] = c[i % 7];
  }
  int* dc; unsigned long long* dout; double* dsink;
  hipMalloc(&dc, cams.size() * 4); hipMalloc(&dout, 8 * 65536); hipMalloc(&dsink, 8 * 4096);
  hipMemcpy(dc, cams.data(), cams.size() * 4, hipMemcpyHostToDevice);
  struct Cfg { int mode, P; const char* name; };
  Cfg cfgs[] = {{0, 0, "[i][col] W=900"}, {1, 9, "[col][i] P=9"}, {2, 10, "[col][i] P=10"}, {2, 11, "[col][i] P=11"},
                {2, 12, "[col][i] P=12"}, {2, 16, "[col][i] P=16"}, {3, 81, "[cam][i][j] Q=81"}, {3, 82, "[cam][i][j] Q=82"},
                {3, 84, "[cam][i][j] Q=84"}, {3, 88, "[cam][i][j] Q=88"}, {3, 96, "[cam][i][j] Q=96"}};
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (auto& c : cfgs) {
    int lds_d = c.mode == 0 ? 8100 : (c.mode == 3 ? 100 * c.P + 128 : 900 * c.P + 16);
    if (lds_d * 8 > 150 * 1024) { printf("%-22s skipped (LDS)\n", c.name); continue; }
    const int threads = 1024, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_d * 8, 0, iters, c.mode, c.P, dc, dout, dsink, lds_d);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : h) avg += x; avg /= h.size();
    printf("%-22s %7.2f cycles per wave-instr per CU (16 waves)\n", c.name, avg / (iters * 9.0) / 16);
  }
  return 0;
}
