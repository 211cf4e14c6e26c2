// Microbenchmark: LDS accumulate throughput on gfx950 for the Schur strip kernel's access shape.
//   mode 0: ds_add_f64 (no return), 63 lanes, 7 groups of 9 consecutive doubles, 9 rows per "pass"
//   mode 1: ds_read_b64 + v_add_f64 + ds_write_b64 (non-atomic RMW), same addresses
//   mode 2: ds_add_f32 on the same shape (for reference)
//   mode 3: ds_add_f64, all 64 lanes fully contiguous
// Reports cycles per wave-instruction at several waves-per-CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(int iters, int W, const int* __restrict__ cols, unsigned long long* out, double* sink) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 9 * W; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int col = cols[(wave * 64 + lane) % 4096];
  if (MODE == 3) col = lane + 64 * wave;
  const bool act = MODE == 3 ? true : lane < 63;
  double v = 1.0 + lane;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (act) {
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        if (MODE == 0 || MODE == 3) atomicAdd(&lds[i * W + col], v);
        else if (MODE == 1) lds[i * W + col] += v;
        else atomicAdd(reinterpret_cast<float*>(lds) + i * W + col, (float)v);
      }
    }
    col = (col + 9 * 7) % (W - 64);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = lds[5];
}

int main() {
  const int W = 900, iters = 2000;
  std::vector<int> cols(4096);
  for (int w = 0; w < 64; ++w)
    for (int l = 0; l < 64; ++l) {  // 7 random cameras per wave, 9 consecutive columns each
      int s = l / 9, j = l % 9;
      int cam = (w * 37 + s * 13 + (w * s) % 7) % 90;
      cols[w * 64 + l] = 9 * cam + j;
    }
  int* dcols; unsigned long long* dout; double* dsink;
  hipMalloc(&dcols, 4096 * 4); hipMalloc(&dout, 8 * 65536); hipMalloc(&dsink, 8 * 4096);
  hipMemcpy(dcols, cols.data(), 4096 * 4, hipMemcpyHostToDevice);
  const size_t lds = 9 * W * sizeof(double);
  auto run = [&](auto kern, const char* name) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int threads : {64, 256, 512, 1024}) {
      const int blocks = 256;  // one per CU
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, iters, W, dcols, dout, dsink);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, iters, W, dcols, dout, dsink);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks * threads / 64);
      hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto x : h) avg += x; avg /= h.size();
      // readcyclecounter on gfx9 = s_memtime (constant 100MHz?) -> also report per-CU throughput
      printf("%-28s waves/CU %2d: %8.1f ticks per wave-instr per wave, %8.2f ticks per wave-instr per CU\n", name,
             threads / 64, avg / (iters * 9.0), avg / (iters * 9.0) / (threads / 64));
    }
  };
  run(k<0>, "ds_add_f64 strip-shape");
  run(k<3>, "ds_add_f64 contiguous");
  run(k<1>, "read+add+write f64 (non-atomic)");
  run(k<2>, "ds_add_f32 strip-shape");
  return 0;
}
