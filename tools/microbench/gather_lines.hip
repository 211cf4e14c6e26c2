// Microbenchmark: gathering whole 128-byte lines (one BA observation record each) by index on gfx950.
// Decides how a pair-major Schur kernel should fetch its operands.
//   mode P  per-lane: every lane reads NS x 16 B out of ITS OWN line (64 distinct lines per instruction)
//   mode C  cooperative: 8 lanes share a line (lane & 7 = 16-B slot), 8 lines per instruction, to registers
//   mode D  cooperative LDS-DMA: same addressing as C, global_load_lds_dwordx4 into a wave-private LDS tile
// Tables: 2 MB (L2-resident), 96 MB (Infinity Cache), 2 GB (HBM).  Indices uniform random.
// Reports useful GB/s (lines x 128 B; P counts NS x 16 B per line) and lines per microsecond per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NS = 7;

__global__ __launch_bounds__(256) void k_perlane(const double2 *__restrict__ tab, const int *__restrict__ idx, long long n,
                                                 double *sink) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i + stride < n; i += 2 * stride) {
    const double2 *p = tab + (size_t)idx[i] * 8, *q = tab + (size_t)idx[i + stride] * 8;
    double2 v[2 * NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { v[s] = p[s]; v[NS + s] = q[s]; }
#pragma unroll
    for (int s = 0; s < 2 * NS; ++s) acc += v[s].x + v[s].y;
  }
  if (acc == 1.2345) sink[0] = acc;
}

// 8 lanes per line; each wave-instruction covers 8 lines; UN instructions in flight per lane
template <int UN>
__global__ __launch_bounds__(256) void k_coop(const double2 *__restrict__ tab, const int *__restrict__ idx, long long n,
                                              double *sink) {
  const int lane = threadIdx.x & 63, g = lane >> 3, s = lane & 7;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  for (long long base = wave * (8 * UN); base + 8 * UN <= n; base += nwaves * (8 * UN)) {
    double2 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = tab[(size_t)idx[base + 8 * u + g] * 8 + s];
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 1.2345) sink[0] = acc;
}

template <int UN>
__global__ __launch_bounds__(256) void k_dma(const double2 *__restrict__ tab, const int *__restrict__ idx, long long n,
                                             double *sink) {
  extern __shared__ double2 lds[];  // [waves][UN][64]
  const int lane = threadIdx.x & 63, g = lane >> 3, s = lane & 7, w = threadIdx.x >> 6;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  double2 *mine = lds + (size_t)w * UN * 64;
  double acc = 0.0;
  for (long long base = wave * (8 * UN); base + 8 * UN <= n; base += nwaves * (8 * UN)) {
    int ii[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) ii[u] = idx[base + 8 * u + g];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      __builtin_amdgcn_global_load_lds(tab + (size_t)ii[u] * 8 + s, mine + u * 64, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += mine[lane].x;  // one read back per tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 1.2345) sink[0] = acc;
}

// mode R: LDS-DMA with LPR lanes per row (LPR x 16 bytes out of one line per request): 8 = whole lines, 7 = the Schur
// kernel's 112-byte records, 4 = a 64-byte half line (what compact records would ask for), 3 = a 48-byte point row
template <int LPR, int UN, int DUP = 1>  // DUP: every index is used by DUP adjacent rows of an instruction (do equal lines coalesce?)
__global__ __launch_bounds__(256) void k_dma_rows(const double2 *__restrict__ tab, const int *__restrict__ idx, long long n,
                                                  double *sink) {
  extern __shared__ double2 lds[];  // [waves][UN][64]
  constexpr int RPI = 64 / LPR;     // rows per instruction
  const int lane = threadIdx.x & 63, g = lane / LPR, s = lane - LPR * g, w = threadIdx.x >> 6;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  double2 *mine = lds + (size_t)w * UN * 64;
  double acc = 0.0;
  for (long long base = wave * (RPI * UN); base + RPI * UN <= n; base += nwaves * (RPI * UN)) {
    int ii[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) ii[u] = idx[base + RPI * u + min(g, RPI - 1) / DUP * DUP];
    if (lane < LPR * RPI) {
#pragma unroll
      for (int u = 0; u < UN; ++u)
        __builtin_amdgcn_global_load_lds(tab + (size_t)ii[u] * 8 + s, mine + u * 64, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += mine[lane].x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 1.2345) sink[0] = acc;
}

int main(int argc, char **argv) {
  const long long n_idx = 1LL << 24;  // 16.8M lines gathered per launch = 2.1 GB of lines
  double *sink; CK(hipMalloc(&sink, 64));
  int *d_idx; CK(hipMalloc(&d_idx, n_idx * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t tab_bytes[] = {2ull << 20, 96ull << 20, 2048ull << 20};
  const char *tab_name[] = {"2MB(L2)", "96MB(MALL)", "2GB(HBM)"};
  for (int t = 0; t < 3; ++t) {
    const size_t nlines = tab_bytes[t] / 128;
    double2 *tab; CK(hipMalloc(&tab, tab_bytes[t]));
    CK(hipMemset(tab, 0, tab_bytes[t]));
    std::vector<int> idx(n_idx);
    std::mt19937_64 rng(1234 + t);
    for (auto &v : idx) v = (int)(rng() % nlines);
    CK(hipMemcpy(d_idx, idx.data(), n_idx * 4, hipMemcpyHostToDevice));
    auto timeit = [&](const char *name, auto launch, double bytes_per_line) {
      launch(); launch();
      CK(hipDeviceSynchronize());
      hipEventRecord(e0);
      const int reps = 3;
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1);
      CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("%-11s %-26s %7.3f ms  %7.1f GB/s useful  %6.2f lines/us/CU\n", tab_name[t], name, ms,
             n_idx * bytes_per_line / (ms * 1e-3) / 1e9, n_idx / (ms * 1e3) / 256.0);
      fflush(stdout);
    };
    for (int bpc : {4, 8}) {  // 256-thread blocks per CU
      const int grid = 256 * bpc;
      char nm[64];
      snprintf(nm, sizeof nm, "P per-lane 7x16B  %2d w/CU", bpc * 4);
      timeit(nm, [&] { hipLaunchKernelGGL(k_perlane, dim3(grid), dim3(256), 0, 0, tab, d_idx, n_idx, sink); }, 16.0 * NS);
      snprintf(nm, sizeof nm, "C coop regs UN=4  %2d w/CU", bpc * 4);
      timeit(nm, [&] { hipLaunchKernelGGL(k_coop<4>, dim3(grid), dim3(256), 0, 0, tab, d_idx, n_idx, sink); }, 128.0);
      snprintf(nm, sizeof nm, "C coop regs UN=8  %2d w/CU", bpc * 4);
      timeit(nm, [&] { hipLaunchKernelGGL(k_coop<8>, dim3(grid), dim3(256), 0, 0, tab, d_idx, n_idx, sink); }, 128.0);
      snprintf(nm, sizeof nm, "D lds-dma UN=4    %2d w/CU", bpc * 4);
      timeit(nm, [&] { hipLaunchKernelGGL(k_dma<4>, dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 128.0);
      if (bpc == 4 && t == 0) {  // request size against request rate, from L2
        timeit("R lds-dma 8 lanes/row 128B", [&] { hipLaunchKernelGGL((k_dma_rows<8, 4>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 128.0);
        timeit("R lds-dma 7 lanes/row 112B", [&] { hipLaunchKernelGGL((k_dma_rows<7, 4>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 112.0);
        timeit("R 7 lanes/row, rows in equal PAIRS", [&] { hipLaunchKernelGGL((k_dma_rows<7, 4, 2>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 112.0);
        timeit("R 7 lanes/row, rows in equal TRIPLES", [&] { hipLaunchKernelGGL((k_dma_rows<7, 4, 3>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 112.0);
        timeit("R lds-dma 4 lanes/row  64B", [&] { hipLaunchKernelGGL((k_dma_rows<4, 4>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 64.0);
        timeit("R lds-dma 3 lanes/row  48B", [&] { hipLaunchKernelGGL((k_dma_rows<3, 4>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 48.0);
        timeit("R lds-dma 2 lanes/row  32B", [&] { hipLaunchKernelGGL((k_dma_rows<2, 4>), dim3(grid), dim3(256), 4 * 4 * 1024, 0, tab, d_idx, n_idx, sink); }, 32.0);
      }
      if (bpc == 4) {
        snprintf(nm, sizeof nm, "D lds-dma UN=8    %2d w/CU", bpc * 4);
        timeit(nm, [&] { hipLaunchKernelGGL(k_dma<8>, dim3(grid), dim3(256), 4 * 8 * 1024, 0, tab, d_idx, n_idx, sink); }, 128.0);
      }
    }
    CK(hipFree(tab));
  }
  return 0;
}
