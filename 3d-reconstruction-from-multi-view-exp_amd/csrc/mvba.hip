// libmvba.so -- bundle-adjustment engine for MI355X (gfx950), hand-written HIP.
//
// Replaces the numerical body of the reference's BundleAdjuster.optimize
// (lib/bundle_adjustment.py:103-162) with a sparse, observation-list pipeline:
//   K1  k_resid_jac      residual + 2x3 / 2x9 Jacobian rows per observation   (ref :291-427)
//   K2  (fused into K1)  E_a = 2 sum JxT Jx, dP_a = 2 sum JxT e               (ref :429-469, :519-556)
//   K3a k_point_inv      damped 3x3 inverse, v_a = E^-1 dP_a                  (ref :120-128)
//   K3  k_schur_slots (up to ~100 cameras) / k_schur_pairs + k_schur_reduce (k_schur_strip behind MVBA_SCHUR=strip)
//                        A = G^ - sum F^T E^-1 F,  b = sum F^T E^-1 dP - dF   (ref :132-143, :471-517, :618-664)
//   C1  ncclAllReduce    [A|b] across point shards                            
//   K4  k_compact, k_chol_super / k_chol_trail64 / k_chol_trail32 / k_chol_backsolve_all (+ the k_lu_* rescue)
//                        dense solve of the gauge-reduced system              (ref :146)
//   K5+K6 k_backsub, k_cost  dX_a, trial state, trial cost                    (ref :152-162, :260-281, :666-677)
// HBM layout: observations sorted by point (CSR).  The linearisation of ONE
// observation is ONE 128-byte line ("record", 8 x double2 = (row0,row1) pairs):
//   slot 0-2  J_X columns          slot 3    dJ/df
//   slot 4-6  dJ/domega columns    slot 7    residual e
// J_C's translation columns are exactly -J_X and its (u,v) columns are the
// constants (1/f0,0),(0,1/f0) (ref :350-376), so the 2x9 block is implied by the
// record.  A point's observations are consecutive lines; the Schur kernel gathers whole lines
// (k-side record, l-side record, point block) per (point, camera pair) item.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "mvba_common.h"

namespace mvba {
thread_local std::string g_err;
}
using namespace mvba;

// ------------------------------------------------------------------ RCCL binding
// libmvba.so is NOT linked against librccl: a process may already hold one (PyTorch maps its own
// bundled librccl.so the moment torch.distributed creates a group), and two copies -- or headers of
// one release bound to the code of another by accident of load order -- is the version skew a
// collective library does not forgive.  The policy is explicit instead: use the librccl the
// process has ALREADY loaded if there is one (RTLD_NOLOAD), else load ROCm's; bind the seven entry
// points by name (all part of the NCCL 2.x C API: plain pointers, enums whose values have not
// changed since 2.0, the 128-byte id), and refuse a library whose major version differs from the
// headers this file was compiled against.  mvba_get_info reports both versions.
namespace {
struct Rccl {
  void *lib = nullptr;
  int version = 0;
  std::string origin;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.lib) return MVBA_OK;
  const char *names[] = {"librccl.so.1", "librccl.so"};
  void *lib = nullptr;
  std::string origin;
  for (const char *n : names)
    if (!lib && (lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) origin = std::string(n) + " (already mapped by the process)";
  if (const char *ev = getenv("MVBA_RCCL_LIBRARY"))
    if (!lib && (lib = dlopen(ev, RTLD_NOW | RTLD_GLOBAL))) origin = ev;
  for (const char *n : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1"})
    if (!lib && (lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) origin = n;
  if (!lib) return fail(MVBA_ERR_RCCL, std::string("librccl not found: ") + dlerror());
  Rccl r;
  r.lib = lib;
  r.origin = origin;
#define BIND(field, sym)                                                                       \
  if (!(r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, sym))))                       \
    return fail(MVBA_ERR_RCCL, std::string("librccl (") + origin + ") lacks " + sym)
  BIND(GetVersion, "ncclGetVersion");
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(AllReduce, "ncclAllReduce");
  BIND(AllGather, "ncclAllGather");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  if (r.GetVersion(&r.version) != ncclSuccess) return fail(MVBA_ERR_RCCL, "ncclGetVersion failed");
  // version code: major * 10000 + minor * 100 + patch since 2.9 (major * 1000 + ... before)
  const int major = r.version >= 10000 ? r.version / 10000 : r.version / 1000;
  if (major != NCCL_MAJOR)
    return fail(MVBA_ERR_RCCL, "librccl (" + origin + ") is NCCL " + std::to_string(r.version) + ", this library was built against " +
                                   std::to_string(NCCL_VERSION_CODE));
  g_rccl = r;
  return MVBA_OK;
}
}  // namespace

// ------------------------------------------------------------------ device helpers
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the sum; fixed tree -> deterministic
}

// Deterministic block sum (fixed tree); result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *s_red /*[16]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) s_red[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) t += s_red[i];
  }
  return t;
}

__device__ __forceinline__ void load_cams_to_lds(const double *__restrict__ cam15, int m, double f0,
                                                 double *s_cam) {
  for (int k = threadIdx.x; k < m; k += blockDim.x) expand_cam(cam15 + (size_t)k * CAM_IN, f0, s_cam + k * CAM_LDS);
}

// Reduced system storage: the upper block triangle of A packed strip by strip -- strip k is a
// row-major 9 x 9(m-k) block (cameras l >= k) at offset 81 (k m - k (k-1) / 2) -- followed by b
// (9m).  81 m (m+1) / 2 + 9m doubles: half of the dense 9m x 9m, and what the all-reduce ships.
__host__ __device__ __forceinline__ size_t strip_offset(int k, int m) {
  return 81 * ((size_t)k * m - (size_t)k * (k - 1) / 2);
}

__device__ __forceinline__ int keep_index(int i, int gauge_axis) {
  // i-th kept parameter -> global parameter index; removed = {3..8, 12+axis} (ref :62-72)
  if (i < 3) return i;
  return (i + 6 < 12 + gauge_axis) ? i + 6 : i + 7;
}

// Wave-uniform operands are read through the constant address space so that the compiler issues scalar (SMEM)
// loads into SGPRs instead of 64 identical vector loads.
#define MVBA_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const T MVBA_CONST_AS *as_const(const T *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const T MVBA_CONST_AS *)p;
#pragma clang diagnostic pop
}

// ------------------------------------------------------------------ K1
// One thread per observation, grid-stride over 256-observation tiles; camera table
// staged once per block.  Each wave transposes its 64 records through LDS so that
// every global store instruction writes 1 KiB contiguous (8 whole lines).
// K2 is fused in: the per-point blocks E_a = 2 sum Jx^T Jx (6 unique) and dP_a = 2 sum Jx^T e
// (ref :429-469, :519-556) are formed by a wave-level segmented sum over the (point-sorted)
// observations -- PL[a][9] = Exx,Exy,Exz,Eyy,Eyz,Ezz,dP0..2 -- so the records are not re-read.
// Algorithmic traffic: 24 B in + 128 B out per observation + (24 in + 72 out) B per point.
constexpr int REC = 8;  // double2 slots per observation record (128 B)
constexpr int LDS_CAMERAS = 646;  // cameras whose tables (K1: 18 doubles + the waves' staging tiles; K5: 28 doubles) fit one workgroup's LDS

// GCAM: beyond LDS_CAMERAS cameras the expanded camera table does not fit a workgroup's LDS next to its staging tiles; the
// kernels then read the rows of a table in device memory (k_cam_tables; 144 bytes per camera, L2-resident) through the same
// 16-byte loads.  One template parameter per kernel: the LDS form keeps its ds_read_b128, nothing is decided per access.
template <bool GCAM>
__global__ __launch_bounds__(1024, 4) void k_resid_jac(long long nobs, int m, const double *__restrict__ cam15,
                                                   const double *__restrict__ X,
                                                   const int *__restrict__ obs_pt,
                                                   const int *__restrict__ cam_idx,
                                                   const double2 *__restrict__ xy, double f0,
                                                   const int *__restrict__ tile_start, int n_tiles,
                                                   double2 *__restrict__ rec, double *__restrict__ PL,
                                                   const int *__restrict__ tile_slot, double *__restrict__ PLsplit,
                                                   const double *__restrict__ gcam) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *s_cam = smem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  // per-wave 8 KiB staging tile, 16-byte aligned behind the camera table
  double2 *stage = reinterpret_cast<double2 *>(smem + (GCAM ? 0 : ((m * CAM_LDS + 1) & ~1))) + wave * (64 * REC);
  // the same 8 KiB is reused for the per-point sums: contrib[9][CS] doubles (odd stride: the nine
  // components of one observation sit in nine different banks), then seg_start[65], pt[64]
  constexpr int CS = 65;
  double *contrib = reinterpret_cast<double *>(stage);
  int *seg_start = reinterpret_cast<int *>(contrib + 9 * CS), *seg_pt = seg_start + 66;
  if (!GCAM) {
    load_cams_to_lds(cam15, m, f0, s_cam);
    __syncthreads();
  }
  // Wave tiles are POINT-ALIGNED (built once on the host): whole points packed greedily into at
  // most 64 observations, so every per-point sum below is complete inside one wave -> plain
  // stores, no atomics, bitwise-reproducible E_a / dP_a.  Only a point with more than 64
  // observations is split over tiles (tile_start < 0 marks such a tile: it holds ONE piece of ONE point, whose sums
  // go to slot tile_slot[tile] of a side buffer; k_sum_split adds a point's pieces in tile order afterwards).
  for (int tile = blockIdx.x * nwave + wave; tile < n_tiles; tile += gridDim.x * nwave) {
    const int ts0 = tile_start[tile], ts1 = tile_start[tile + 1];
    const bool split = ts0 < 0;
    const int slot = split ? as_const(tile_slot)[tile] : 0;  // (wave-uniform: a scalar load)
    const long long wbase = split ? ~ts0 : ts0;  // first observation of this wave's tile
    const int n = (int)((ts1 < 0 ? ~ts1 : ts1) - wbase);
    const long long o = wbase + lane;
    const bool live = lane < n;
    int a = -1;
    double c9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      a = obs_pt[o];
      const int k = cam_idx[o];
      const double2 z = xy[o];
      const double *Xa = X + 3 * (size_t)a;
      ObsJ J;
      obs_math(Xa[0], Xa[1], Xa[2], GCAM ? gcam + (size_t)k * CAM_LDS : s_cam + k * CAM_LDS, z.x, z.y, f0, J);
      // swizzled slot position (s ^ (lane & 7)): conflict-free ds_write_b128
      double2 *row = stage + lane * REC;
      const int sw = lane & 7;
      row[0 ^ sw] = make_double2(J.jx[0][0], J.jx[1][0]);
      row[1 ^ sw] = make_double2(J.jx[0][1], J.jx[1][1]);
      row[2 ^ sw] = make_double2(J.jx[0][2], J.jx[1][2]);
      row[3 ^ sw] = make_double2(J.jc[0][0], J.jc[1][0]);
      row[4 ^ sw] = make_double2(J.jc[0][6], J.jc[1][6]);
      row[5 ^ sw] = make_double2(J.jc[0][7], J.jc[1][7]);
      row[6 ^ sw] = make_double2(J.jc[0][8], J.jc[1][8]);
      row[7 ^ sw] = make_double2(J.e0, J.e1);
      // K2 fused: this observation's share of E_a = 2 sum Jx^T Jx and dP_a = 2 sum Jx^T e
      c9[0] = 2.0 * (J.jx[0][0] * J.jx[0][0] + J.jx[1][0] * J.jx[1][0]);
      c9[1] = 2.0 * (J.jx[0][0] * J.jx[0][1] + J.jx[1][0] * J.jx[1][1]);
      c9[2] = 2.0 * (J.jx[0][0] * J.jx[0][2] + J.jx[1][0] * J.jx[1][2]);
      c9[3] = 2.0 * (J.jx[0][1] * J.jx[0][1] + J.jx[1][1] * J.jx[1][1]);
      c9[4] = 2.0 * (J.jx[0][1] * J.jx[0][2] + J.jx[1][1] * J.jx[1][2]);
      c9[5] = 2.0 * (J.jx[0][2] * J.jx[0][2] + J.jx[1][2] * J.jx[1][2]);
      c9[6] = 2.0 * (J.jx[0][0] * J.e0 + J.jx[1][0] * J.e1);
      c9[7] = 2.0 * (J.jx[0][1] * J.e0 + J.jx[1][1] * J.e1);
      c9[8] = 2.0 * (J.jx[0][2] * J.e0 + J.jx[1][2] * J.e1);
    }
    wave_sync();  // wave-synchronous hand-over through LDS (in-order DS queue)
#pragma unroll
    for (int q = 0; q < REC; ++q) {
      const int ol = q * 8 + (lane >> 3), pos = lane & 7;  // local observation, stored position
      const double2 v = stage[ol * REC + pos];
      if (ol < n) rec[(wbase + ol) * REC + (pos ^ (ol & 7))] = v;
    }
    wave_sync();
    // ---- per-point sums (wave-level segmented reduction; observations are sorted by point)
    const int a_prev = __shfl_up(a, 1, 64);
    const bool head = live && (lane == 0 || a != a_prev);
    const unsigned long long heads = __ballot(head);
    const int nseg = __popcll(heads);
#pragma unroll
    for (int q = 0; q < 9; ++q) contrib[q * CS + lane] = c9[q];
    if (head) {
      const int rank = __popcll(heads & ((1ull << lane) - 1ull));
      seg_start[rank] = lane;
      seg_pt[rank] = a;
    }
    if (lane == 0) seg_start[nseg] = n;
    wave_sync();
    const int sl = lane / 9, comp = lane - 9 * sl;
    for (int s0 = 0; s0 < nseg; s0 += 7) {
      const int sg = s0 + sl;
      if (sl < 7 && sg < nseg) {
        const int i0 = seg_start[sg], i1 = seg_start[sg + 1];
        double acc = 0.0;
        for (int i = i0; i < i1; ++i) acc += contrib[comp * CS + i];
        // (split: a piece of a > 64-observation point -> its slot of the side buffer; wave-uniform choice)
        if (split) PLsplit[9 * (size_t)slot + comp] = acc;
        else PL[9 * (size_t)seg_pt[sg] + comp] = acc;
      }
    }
    wave_sync();
  }
}

// The per-point blocks of points with more than 64 observations: their pieces in tile order (fixed order: no atomics).
__global__ void k_sum_split(int n_split, const int4 *__restrict__ splits /* point, first slot, pieces */,
                            const double *__restrict__ PLsplit, double *__restrict__ PL) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, sp = t / 9, comp = t - 9 * sp;
  if (sp >= n_split) return;
  const int4 d = splits[sp];
  double v = 0.0;
  for (int q = 0; q < d.z; ++q) v += PLsplit[9 * (size_t)(d.y + q) + comp];
  PL[9 * (size_t)d.x + comp] = v;
}

// ------------------------------------------------------------------ K3a
// PB[a][PBS] = inverse of E_a with diagonal*(1+c) (6 unique), v_a = E^-1 dP_a (3), pad: one
// 128-byte line per point, so a gather of the inverse touches one 64-byte sector.
constexpr int PBS = 16;
constexpr int PACE_STRIDE = 32;  // ints between two pacing counters: one 128-byte line each (they are hammered by ~300 waves)
// Also clears the packed [A|b] the Schur kernel is about to accumulate into (one launch less on
// the path; the grid is sized for whichever of the two jobs is larger).
__global__ __launch_bounds__(256) void k_point_inv(long long npts, double c, const double *__restrict__ PL,
                                                   double *__restrict__ PB, int *__restrict__ flag,
                                                   double *__restrict__ Ab, long long nAb, int *__restrict__ prog, long long nprog, int want_r) {
  // a block's 256 points are 18 KiB of PL and 32 KiB of PB, both contiguous: moved with coalesced
  // accesses through LDS (a thread reading its own 72-byte row / writing its own 128-byte line touches
  // 64 different lines per instruction)
  // (one buffer for both directions -- a thread takes its nine inputs into registers before anybody writes a result: 32 KiB per
  // block = four blocks per CU instead of three at 50 KiB)
  __shared__ double2 s_out[256 * 8];
  double *s_in = reinterpret_cast<double *>(s_out);
  const long long a0 = (long long)blockIdx.x * 256, a = a0 + threadIdx.x;
  for (long long i = a; i < nAb; i += (long long)gridDim.x * blockDim.x) Ab[i] = 0.0;
  for (long long i = a; i < nprog; i += (long long)gridDim.x * blockDim.x) prog[i] = 0;  // pacing counters of k_schur_slots
  if (a0 >= npts) return;  // (uniform)
  const int np = (int)min<long long>(256, npts - a0);
  {  // nine independent loads per thread, then their stores (one load and one store per trip waits for every load in turn)
    double t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = PL[9 * a0 + min((int)threadIdx.x + 256 * i, 9 * np - 1)];
#pragma unroll
    for (int i = 0; i < 9; ++i)
      if ((int)threadIdx.x + 256 * i < 9 * np) s_in[threadIdx.x + 256 * i] = t[i];
  }
  __syncthreads();
  double in[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) in[i] = s_in[9 * min((int)threadIdx.x, np - 1) + i];
  __syncthreads();
  if (threadIdx.x < np) {
    const double s = 1.0 + c;
    const double xx = in[0] * s, xy = in[1], xz = in[2], yy = in[3] * s, yz = in[4], zz = in[5] * s;
    const double c00 = yy * zz - yz * yz, c01 = xz * yz - xy * zz, c02 = xy * yz - xz * yy;
    const double det = xx * c00 + xy * c01 + xz * c02;
    if (!(det != 0.0) || !isfinite(det)) atomicOr(flag, 1);  // singular 3x3 (ref :128 raises LinAlgError)
    const double id = 1.0 / det;
    const double i00 = c00 * id, i01 = c01 * id, i02 = c02 * id;
    const double i11 = (xx * zz - xz * xz) * id, i12 = (xy * xz - xx * yz) * id, i22 = (xx * yy - xy * xy) * id;
    const double g0 = in[6], g1 = in[7], g2 = in[8];
    // slot s of point p sits at s_out[8 p + (s ^ (p & 7))]: conflict-free 16-byte LDS stores, undone on the way out
    double2 *out = s_out + 8 * threadIdx.x;
    const int sw = threadIdx.x & 7;
    out[0 ^ sw] = make_double2(i00, i01);
    out[1 ^ sw] = make_double2(i02, i11);
    out[2 ^ sw] = make_double2(i12, i22);
    out[3 ^ sw] = make_double2(i00 * g0 + i01 * g1 + i02 * g2, i01 * g0 + i11 * g1 + i12 * g2);
    out[4 ^ sw] = make_double2(i02 * g0 + i12 * g1 + i22 * g2, 1.0);  // tenth double: 1 = a point (row N, the padding row, stays 0)
    if (want_r) {
      // the dense-visibility Schur form reads E^-1 = R S R^T from the row's last three slots: R lower triangular, S = diag(+-1) --
      // L D L^T with |D|^1/2 folded into the columns; S = I for the positive definite inverse of every LM step, but the damped
      // blocks of a NEGATIVE damping factor (the LU-path test; np.linalg.solve takes any system) are indefinite.  The signs ride in
      // the tenth double: 1 + (d0 < 0) + 2 (d1 < 0) + 4 (d2 < 0)  (nobody else reads it in this mode)
      const double d0 = i00, q0 = d0 != 0.0 ? 1.0 / d0 : 0.0, l10 = i01 * q0, l20 = i02 * q0;
      const double d1 = i11 - l10 * i01, q1 = d1 != 0.0 ? 1.0 / d1 : 0.0, l21 = (i12 - l20 * i01) * q1;
      const double d2 = i22 - l20 * i02 - l21 * l21 * d1;
      const double a0 = sqrt(fabs(d0)), a1 = sqrt(fabs(d1)), a2 = sqrt(fabs(d2));
      out[4 ^ sw].y = 1.0 + (d0 < 0.0 ? 1.0 : 0.0) + (d1 < 0.0 ? 2.0 : 0.0) + (d2 < 0.0 ? 4.0 : 0.0);
      out[5 ^ sw] = make_double2(a0, l10 * a0);
      out[6 ^ sw] = make_double2(l20 * a0, a1);
      out[7 ^ sw] = make_double2(l21 * a1, a2);
    } else {
      out[5 ^ sw] = out[6 ^ sw] = out[7 ^ sw] = make_double2(0.0, 0.0);
    }
  }
  __syncthreads();
  double2 *dst = reinterpret_cast<double2 *>(PB + PBS * a0);
  for (int e = threadIdx.x; e < 8 * np; e += 256) {
    const int p = e >> 3, sl = e & 7;
    dst[e] = s_out[8 * p + (sl ^ (p & 7))];
  }
}

// ------------------------------------------------------------------ K3
// Block (k, chunk, seg): accumulates the block-row strip A[9k..9k+8][9 l_lo .. 9 l_hi)
// (l >= k: upper block triangle only) in LDS over the points of `chunk` seen by
// camera k, then flushes it once.  Each wave walks camera-major records
// (obs, point, remaining-observations-in-row); the k-side operands are wave-uniform
// (scalar loads), the l-side operands are per lane: lane = 9*slot + j handles
// column j of the 9x9 block of the slot-th remaining observation of the point.
//   -(F_ak^T E^-1 F_al)[i][j] = -Jc_k[:,i] . ( 2 Jx_k E^-1 ( 2 Jx_l^T Jc_l[:,j] ) )
// The diagonal item (l == k) also adds G^_k (ref :618-664 with :123-125 damping)
// and the right-hand side 2 Jc_k^T (Jx_k v_a - e_ak)   (ref :138-143, :471-517).
// Wave-uniform operands are read through the constant address space so the
// compiler issues scalar (SMEM) loads into SGPRs instead of 64 identical vector loads.

// LDS strip layout: strip[(9 (l - l_lo) + j) * 9 + i]  (i fastest) so that the nine
// accumulations of a lane are ONE address + immediate offsets; sb (rhs) follows.
template <bool BIG, bool SEG>  // BIG: record byte offsets need 64 bits (n_obs * 128 >= 4 GiB); SEG: strips cut into 2..4 column segments
__global__ __launch_bounds__(768, 6) void k_schur_strip(
    int m, int nchunks, int lseg, int nsp, const long long *__restrict__ chunk_ptr, const int4 *__restrict__ csc,
    const int *__restrict__ cam_idx, const double2 *__restrict__ rec, const double *__restrict__ PB, double c,
    double f0, double *__restrict__ Afull, double *__restrict__ bfull) {
  extern __shared__ double strip[];
  const int k = blockIdx.x, chunk = blockIdx.y, seg = blockIdx.z;
  const int l_lo = k + seg * lseg;
  if (l_lo >= m) return;
  const int l_hi = min(m, l_lo + lseg);
  const int W = 9 * (l_hi - l_lo);
  double *sb = strip + 9 * W;
  for (int i = threadIdx.x; i < 9 * W + 9; i += blockDim.x) strip[i] = 0.0;
  // Workgroup barrier WITHOUT a compiler-level memory fence: a fencing __syncthreads() here
  // makes LLVM treat every later load as clobbered and turns the wave-uniform k-side loads
  // back into 64-lane vector loads (+70 VGPRs).  The LDS writes above are drained first.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // LDS-only: the zero-fill may not sink below the barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);
  const int lane = threadIdx.x & 63;
  const int slot = lane / 9, j = lane - 9 * slot;
  // J_C column j of an observation = alpha * record_slot[sel] + beta  (see file header);
  // the factor 4 = 2 (F = 2 Jx^T Jc) * 2 (t = 2 Jx_k h) is folded in here.
  const unsigned sel16 = 16u * ((j == 0) ? 3 : (j >= 6 ? j - 2 : (j >= 3 ? j - 3 : 0)));
  const double cu = 1.0 / f0;
  const double alpha = (j == 1 || j == 2) ? 0.0 : ((j >= 3 && j < 6) ? -4.0 : 4.0);
  const double beta_x = (j == 1) ? 4.0 * cu : 0.0, beta_y = (j == 2) ? 4.0 * cu : 0.0;
  const long long beg = chunk_ptr[(size_t)k * (nchunks + 1) + chunk];
  const long long end = chunk_ptr[(size_t)k * (nchunks + 1) + chunk + 1];
  // scalar views (plain int / double elements: HIP vector types do not cross address spaces)
  const auto *csc_c = as_const(reinterpret_cast<const int *>(csc));
  const auto *rec_c = as_const(reinterpret_cast<const double *>(rec));
  const auto *PBc = as_const(PB);
  const char *recb = reinterpret_cast<const char *>(rec);
  auto load_rec = [&](long long i) { return make_int4(csc_c[4 * i], csc_c[4 * i + 1], csc_c[4 * i + 2], SEG ? csc_c[4 * i + 3] : 0); };
  // SEG (strips cut into nsp = 2..4 column segments; more segments fall back to scanning): the remaining
  // observations of an entry are sorted by camera, so this block's items are the index range
  // [first, last) of them; csc.w packs the segment boundaries (10 bits each, built on the host).
  auto seg_first = [&](const int4 &r) { return (SEG && seg > 0) ? ((r.w >> (10 * (seg - 1))) & 1023) : 0; };
  auto seg_last = [&](const int4 &r) { return (SEG && seg + 1 < nsp) ? ((r.w >> (10 * seg)) & 1023) : r.z; };
  auto line = [&](int ol) -> const char * {
    if (BIG) return recb + ((size_t)ol << 7);
    return recb + ((unsigned)ol << 7);  // scalar base + 32-bit lane offset
  };

  // ---- software-pipelined walk over (entry, pass) pairs --------------------------------
  // Scalar loads (SMEM) and LDS atomics share lgkmcnt and SMEM returns out of order, so a
  // wait for the k-side operands is always lgkmcnt(0) and would also wait for every LDS
  // atomic still queued behind 31 other waves.  Order per iteration therefore:
  //   compute(pass n) -> issue loads for pass n+1 (vector l-side, scalar k-side if the
  //   entry changes) -> s_waitcnt lgkmcnt(0) -> issue the 9 atomics of pass n.
  // The atomics then drain under the compute of pass n+1.
  long long idx = beg + wave;
  if (idx < end) {
    int4 cur = load_rec(idx);
    int4 nxt = (idx + nw < end) ? load_rec(idx + nw) : cur;
    int base = seg_first(cur), last = seg_last(cur);
    // k-side: ONE record line + the point block, wave-uniform scalar loads
    const double MVBA_CONST_AS *qk = rec_c + (size_t)cur.x * (2 * REC);
    const double MVBA_CONST_AS *pb = PBc + PBS * (size_t)cur.y;
    double kx00 = qk[0], kx10 = qk[1], kx01 = qk[2], kx11 = qk[3], kx02 = qk[4], kx12 = qk[5];
    double kf0 = qk[6], kf1 = qk[7];
    double kw00 = qk[8], kw10 = qk[9], kw01 = qk[10], kw11 = qk[11], kw02 = qk[12], kw12 = qk[13];
    double i00 = pb[0], i01 = pb[1], i02 = pb[2], i11 = pb[3], i12 = pb[4], i22 = pb[5];
    const double *PBg = PB;
    // l-side of the first pass
    bool act = base + slot < last;
    int ol = cur.x + (act ? base + slot : 0);
    int l = cam_idx[ol];
    const char *ql = line(ol);
    double2 x0 = *reinterpret_cast<const double2 *>(ql), x1 = *reinterpret_cast<const double2 *>(ql + 16),
            x2 = *reinterpret_cast<const double2 *>(ql + 32), cs = *reinterpret_cast<const double2 *>(ql + sel16);
    // the diagonal item always lands on the same addresses for a given lane (slot 0, l == k):
    // its rhs and Marquardt-damping terms accumulate in registers and are flushed once per wave
    double bacc = 0.0, dacc = 0.0;
    while (true) {
      // ---------------- compute pass n
      const bool valid = act && slot < 7 && l >= l_lo && l < l_hi;
      double *dst = strip + (9 * (l - l_lo) + j) * 9;
      const bool diag = valid && (base + slot == 0);
      double val[9];
      // a pass with no lane inside this block's column segment (strips cut into segments, m > 236)
      // costs only its loads: wave-uniform skip of the arithmetic
      if (__ballot(valid) != 0ull) {
        const double cjx = alpha * cs.x + beta_x, cjy = alpha * cs.y + beta_y;  // 4 * Jc_l[:, j]
        const double g0 = x0.x * cjx + x0.y * cjy;  // 2 * F_al[:, j]
        const double g1 = x1.x * cjx + x1.y * cjy;
        const double g2 = x2.x * cjx + x2.y * cjy;
        const double h0 = i00 * g0 + i01 * g1 + i02 * g2;  // 2 * E^-1 F_al[:, j]
        const double h1 = i01 * g0 + i11 * g1 + i12 * g2;
        const double h2 = i02 * g0 + i12 * g1 + i22 * g2;
        double t0 = kx00 * h0 + kx01 * h1 + kx02 * h2;  // 2 Jx_k E^-1 F_al[:, j]
        double t1 = kx10 * h0 + kx11 * h1 + kx12 * h2;
        if (diag) {
          // diagonal item (l == k, Jc_l == Jc_k): G_k - S_kk = -Jc_k^T (t - 2 Jc_k[:, j]); the
          // Marquardt factor (1+c) on G's diagonal is one extra accumulation on element (j, j);
          // right-hand side 2 Jc_k[:, j] . (Jx_k E^-1 dP - e).  Rare operands come per lane.
          const double d0 = 0.5 * cjx, d1 = 0.5 * cjy;  // 2 * Jc_k[:, j]
          t0 -= d0;
          t1 -= d1;
          dacc += c * 0.125 * (cjx * cjx + cjy * cjy);  // c * 2 |Jc_k[:, j]|^2
          const double *pv = PBg + PBS * (size_t)cur.y + 6;
          const double2 ke = *reinterpret_cast<const double2 *>(line(cur.x) + 112);
          const double w0 = kx00 * pv[0] + kx01 * pv[1] + kx02 * pv[2] - ke.x;
          const double w1 = kx10 * pv[0] + kx11 * pv[1] + kx12 * pv[2] - ke.y;
          bacc += d0 * w0 + d1 * w1;
        }
        // -(Jc_k[:, i] . t) for i = f, u, v, t(3), omega(3)   (t columns of Jc are -Jx)
        val[0] = -(kf0 * t0 + kf1 * t1);
        val[1] = -(cu * t0);
        val[2] = -(cu * t1);
        val[3] = kx00 * t0 + kx10 * t1;
        val[4] = kx01 * t0 + kx11 * t1;
        val[5] = kx02 * t0 + kx12 * t1;
        val[6] = -(kw00 * t0 + kw10 * t1);
        val[7] = -(kw01 * t0 + kw11 * t1);
        val[8] = -(kw02 * t0 + kw12 * t1);
      }
      // ---------------- advance to pass n+1 and issue its loads
      bool done = false;
      if (base + 7 < last) {
        base += 7;
      } else {
        idx += nw;
        if (idx >= end) {
          done = true;
        } else {
          cur = nxt;
          base = seg_first(cur);
          last = seg_last(cur);
          if (idx + nw < end) nxt = load_rec(idx + nw);
          qk = rec_c + (size_t)cur.x * (2 * REC);
          pb = PBc + PBS * (size_t)cur.y;
          kx00 = qk[0]; kx10 = qk[1]; kx01 = qk[2]; kx11 = qk[3]; kx02 = qk[4]; kx12 = qk[5];
          kf0 = qk[6]; kf1 = qk[7];
          kw00 = qk[8]; kw10 = qk[9]; kw01 = qk[10]; kw11 = qk[11]; kw02 = qk[12]; kw12 = qk[13];
          i00 = pb[0]; i01 = pb[1]; i02 = pb[2]; i11 = pb[3]; i12 = pb[4]; i22 = pb[5];
        }
      }
      if (!done) {
        const int it = base + slot;
        act = it < last;
        ol = cur.x + (act ? it : 0);
        l = cam_idx[ol];
        ql = line(ol);
        x0 = *reinterpret_cast<const double2 *>(ql); x1 = *reinterpret_cast<const double2 *>(ql + 16);
        x2 = *reinterpret_cast<const double2 *>(ql + 32); cs = *reinterpret_cast<const double2 *>(ql + sel16);
      }
      // the scalar loads above must have landed BEFORE the atomics enter the LDS queue
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- the 9 accumulations of pass n (asynchronous from here on)
      if (valid) {
#pragma unroll
        for (int i = 0; i < 9; ++i) unsafeAtomicAdd(dst + i, val[i]);  // ds_add_f64 (this kernel IS its LDS atomics)
      }
      if (done) break;
    }
    if (seg == 0 && lane < 9) {
      unsafeAtomicAdd(strip + lane * 9 + lane, dacc);  // (1+c) damping of G_k's diagonal (ref :123-125)
      unsafeAtomicAdd(&sb[lane], bacc);
    }
  }
  __syncthreads();
  double *Ak = Afull + strip_offset(k, m) + 9 * (size_t)(l_lo - k);  // packed strip k, first column of this segment
  const int Wk = 9 * (m - k);
  for (int q = threadIdx.x; q < 9 * W; q += blockDim.x) {
    const int row = q / W, col = q - row * W;  // coalesced over the columns of A
    const double val = strip[col * 9 + row];
    if (val != 0.0) unsafeAtomicAdd(&Ak[(size_t)row * Wk + col], val);
  }
  if (seg == 0 && threadIdx.x < 9) unsafeAtomicAdd(&bfull[9 * k + threadIdx.x], sb[threadIdx.x]);
}

// ------------------------------------------------------------------ K3 (pair-major form)
// The same sum, organised by OUTPUT block instead of by camera strip.  Every (point, camera k,
// camera l >= k) triple is an "item" (built once on the host, sorted by (k, l), then by point), a
// "unit" is a contiguous run of one pair's items, and ONE WAVE owns a unit: it keeps its share of the
// 9x9 block in registers for the whole run, so there is no scatter and no atomic at all -- the
// strip kernel above spends 75 % of its cycles in the LDS atomic pipe.  Per item the block is the
// rank-2 product  -4 Jc_k^T t Jc_l  with  t = Jx_k E^-1 Jx_l^T (2x2).
//   lanes      lane = 3 item + cg: 21 items per step, column group cg owns columns 3cg..3cg+2 of
//              the block (f,u,v | t | omega) = 27 accumulators; t is formed redundantly by the 3 lanes
//   operands   whole 128-byte lines gathered by LDS-DMA (global_load_lds_dwordx4, 8 lanes -> 7 of the
//              8 slots of a record): the k-side record, the l-side record and the point block of the
//              21 items land in wave-private LDS as 112-byte rows (28-dword stride: the 16 lanes of a
//              ds_read_b128 group hit 16 different bank quads).  Per-lane loads out of 64 different
//              lines are bound by the L1 tag rate (1 line per clock: tools/microbench/gather_lines.hip,
//              297 vs 720 lines/us/CU from L2).
//   diagonal   a (k,k) pair adds G_k (t - I/2 instead of t), the Marquardt term c * diag(G_k) and
//              the right-hand side 2 Jc_k^T (Jx_k E^-1 dP - e); its l-side DMA fetches slots 1..7 of
//              the SAME record (the residual is slot 7) and the point row is 5 slots (E^-1 dP behind
//              the inverse).  Diagonal pairs hold ~10x the items of an off-diagonal pair and are dealt
//              round-robin into sub-lists, so that all units of a strip sweep the points at one pace.
//   placement  the units of strip k run on XCD k % 8 (work queues per XCD, the wave reads its
//              XCC_ID and pulls from that queue first, then steals): the ~5.5 units that need the
//              k-side record of one observation run at the same time on the same L2.
//   output     partial[unit][104] (81 block, 9 damping, 9 rhs), summed per pair in unit order by
//              k_schur_reduce into the packed strips: bitwise reproducible, no zero-fill of A.
#if defined(MVBA_FS)  // (experimental build: the slot form with one lane per item, 64 lists per wave -- csrc/mvba_fs.h; the unit form is not usable in it)
constexpr int PSTEP = 64;
#else
constexpr int PSTEP = 21;                       // items per wave step
#endif
constexpr int PROW = 7 * 16;                    // staged bytes per record (slots 0..6, or 1..7)
constexpr int PWAVE_LDS = PSTEP * (2 * PROW + 5 * 16);  // k rows, l rows, point rows
constexpr int UNIT_STRIDE = 104;                // doubles per unit partial
constexpr int SLOT_BUF = PSTEP * (2 * PROW + 3 * 16);  // slot form: one packed staging buffer (k rows, l rows, 48-byte point rows)
#if defined(MVBA_FS)
constexpr int SLOT_IDX = 3 * PSTEP;             // k[64] | l[64] | a[64]: one 768-byte DMA row
constexpr int SLOT_IDX_RING = 2;                // ... staged two steps deep in LDS
#else
constexpr int SLOT_IDX = 64;                    // slot form: ints per step in the index (k[21] | l[21] | a[21] | pad): ONE 256-byte DMA row
constexpr int SLOT_IDX_RING = 3;                // ... staged three steps deep in LDS
#endif
#if defined(MVBA_HREC_TIMING)
#ifndef MVBA_HREC_NBUF
#define MVBA_HREC_NBUF 3
#endif
constexpr int PAIRS_LDS = 2 * (PSTEP * (80 + 144));
constexpr int SLOT_LDS = MVBA_HREC_NBUF * (PSTEP * (80 + 144)) + MVBA_HREC_NBUF * SLOT_IDX * 4;  // (timing build of the h-in-the-record variant)
#else
constexpr int PAIRS_LDS = 2 * PWAVE_LDS;  // the unit form: two staging buffers per wave
constexpr int SLOT_LDS = 3 * SLOT_BUF + SLOT_IDX_RING * SLOT_IDX * 4;  // three staging buffers + the index ring per wave: 17,904 B, nine waves per CU
// (LDS is handed out in 512-byte granules: 9 x 17,920 = 161,280 of the 163,840 bytes.  16 bytes are all a wave could still have --
// with 48 more, a CU holds eight waves, the range's 284 are no longer all resident and the launch spends 6 ms in pacing time-outs:
// profiles/r05_pace_poll.txt)
#if defined(MVBA_FS)
static_assert(3 * ((SLOT_LDS + 511) / 512 * 512) <= 160 * 1024, "three waves per CU");
#else
static_assert(9 * ((SLOT_LDS + 511) / 512 * 512) <= 160 * 1024, "nine waves per CU");
#endif
#endif

__device__ __forceinline__ void lds_dma16(const void *gsrc, void *lds_wave_uniform) {
  __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_uniform, 16, 0, 0);
}

// One unit (n items from `beg`) on one wave; DIAG: the unit belongs to a (k,k) pair.
// Two LDS buffers per wave: the DMA of step n+1 is issued before step n is computed, with indices
// that were loaded one step earlier still (vmcnt counts in issue order, so one wait per step covers
// both).  12 waves x 6 KB in flight per CU is what a random-line gather needs to run at the HBM
// rate (Little: 6.4 TB/s x ~2.5 us / 256 CUs).
// Loads and DMAs are unconditional (rows past the end of the unit are clamped to its last item and
// land in LDS rows nobody reads): a load under a data-dependent branch makes hipcc wait vmcnt(0)
// per load, and so does a spilled register reloaded between two DMAs -- either drains the DMAs in
// flight.  Only lane 63 of a record DMA is masked off (9 rows x 7 slots = 63 lanes; its 16 bytes
// would land on the next chunk's first slot).
// Pacing of the slot-resident form (k_schur_slots): `seg_end[j]` = the step at which this wave has left segment j of
// its point range, `prog[j]` = how many waves of the range have left segment j, `need` = how many there are.
struct SlotPace {
  const int *seg_end;
  int *prog;
  int need, nseg, lag;
  long long *trace;  // diagnostic builds (-DMVBA_SLOT_TRACE): 16 words per wave (8 general + the loop's phase sums in shader cycles)
};
// SLOTS (the slot-resident form below, k_schur_slots): the 21 item rows of a step belong to 21 DIFFERENT lists, each
// 3-lane slot keeps its own block for the whole run and writes it to its own partial (`out` is then the array of
// partials and `slot_unit` the 21 unit ids of this wave); padding rows point at the all-zero record / point row.
template <bool DIAG, bool BIG, bool SLOTS = false>
__device__ __forceinline__ void schur_pairs_unit(char *wbuf, const int lane, const long long beg, const int n,
                                                 const int *__restrict__ it_k, const int *__restrict__ it_l,
                                                 const int *__restrict__ it_a, const double2 *__restrict__ rec,
                                                 const double *__restrict__ PB, const double c, const double cu,
                                                 double *__restrict__ out, const int *__restrict__ slot_unit = nullptr,
                                                 const SlotPace pace = SlotPace{nullptr, nullptr, 0, 0, 2, nullptr}) {
  constexpr int NPS = DIAG ? 5 : 3;                      // staged 16-byte slots of a point row
  const int it = lane / 3, cg = lane - 3 * it;           // compute: item of the step, column group
  const int drow = lane / 7, dslot = lane - 7 * drow;    // record DMA: 9 rows x 7 slots per instruction
  const int prow = lane / NPS, pslot = lane - NPS * prow;
  const int prow2 = (64 + lane) / 5, pslot2 = (64 + lane) - 5 * prow2;  // DIAG: second point-row DMA
  // column q of this lane = al12 * (row slot sel_q) + unit vector; sign and 1/f0 are applied at the end
  const int sel0 = cg == 0 ? 3 : (cg == 1 ? 0 : 4);
  const int sel1 = cg == 0 ? 0 : sel0 + 1, sel2 = cg == 0 ? 0 : sel0 + 2;
  const double al12 = cg == 0 ? 0.0 : 1.0, bx1 = cg == 0 ? 1.0 : 0.0;
  double acc[9][3];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[i][q] = 0.0;
  double dg[3] = {0.0, 0.0, 0.0}, rb[3] = {0.0, 0.0, 0.0};

  // staging buffer of one step: k rows | l rows (DIAG: the residual slots) | point rows.  The unit form keeps round
  // 2's layout (two buffers of PWAVE_LDS); the slot form packs them (three buffers in a 9-waves-per-CU budget)
  constexpr int LB_OFF = PSTEP * PROW;
  constexpr int PB_OFF = LB_OFF + ((SLOTS && DIAG) ? PSTEP * 16 : PSTEP * PROW);
  constexpr int BUFSZ = SLOTS ? PB_OFF + PSTEP * 16 * NPS : PWAVE_LDS;
  static_assert(!SLOTS || BUFSZ <= SLOT_BUF, "slot form: staging buffer larger than the launch provides");
  int ixk[1][3], ixl[1][3], ixa[1][2];  // the next step's indices (the slot form has two pinned register sets of its own)
  auto load_idx = [&](int s0, int (&xk)[3], int (&xl)[3], int (&xa)[2]) {  // indices of the step starting at item s0 (to registers)
    // uniform base + unsigned 32-bit row: the saddr form again (written with int rows the clamps and the
    // address sums were done in 64 bits per lane: ~40 vector instructions per step for seven loads)
    // (the bases go through readfirstlane and the rows through an empty asm so that the compiler can neither
    // split the uniform sum into per-lane 64-bit additions nor widen the clamp to 64 bits)
    const unsigned last = (unsigned)(min(PSTEP, n - s0) - 1);
    typedef const int __attribute__((address_space(1))) *gint_p;  // (a plain pointer rebuilt from integers would be a FLAT one)
    auto uni = [](const int *p) {
      const unsigned long long v = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return (gint_p)(((unsigned long long)hi << 32) | lo);
    };
    typedef const char __attribute__((address_space(1))) *gchar_p;
    auto row_of = [&](int r) {  // BYTE offset of the clamped row
      unsigned off = min((unsigned)r, last) << 2;
      asm("" : "+v"(off));
      return off;
    };
    auto at = [](gint_p base, unsigned off) { return *(gint_p)((gchar_p)base + off); };
    const gint_p pk = uni(it_k + (beg + s0)), pl = DIAG ? pk : uni(it_l + (beg + s0)), pa = uni(it_a + (beg + s0));
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned row = row_of(9 * q + drow);
      xk[q] = at(pk, row);
      if (!DIAG) xl[q] = at(pl, row);
    }
    if (DIAG) xl[0] = at(pk, row_of(lane));  // the record whose residual slot this lane fetches
    xa[0] = at(pa, row_of(prow));
    if (DIAG) xa[1] = at(pa, row_of(prow2));
  };
  // Addresses as "uniform base + 32-bit byte offset" (the saddr form of the memory instructions: one
  // shift-add per address instead of a sign extension, a 64-bit shift and a 64-bit add); BIG: the
  // records or the point blocks span 4 GiB or more and the offsets need 64 bits.
  auto rec_at = [&](int obs, int slot) -> const void * {
    if (BIG) return rec + (size_t)obs * REC + slot;
    return reinterpret_cast<const char *>(rec) + (((unsigned)obs << 7) + ((unsigned)slot << 4));
  };
  auto pb_at = [&](int a, int slot) -> const void * {
    if (BIG) return PB + (size_t)a * PBS + 2 * slot;
    return reinterpret_cast<const char *>(PB) + (((unsigned)a << 7) + ((unsigned)slot << 4));
  };
  // (OFFDIAG: 7 DMA instructions, DIAG: 6 -- and as many index loads per step: the slot form's counted wait relies on it)
  auto issue = [&](char *buf, const int (&xk_)[3], const int (&xl_)[3], const int (&xa_)[2]) {
    char *kb = buf, *lb = buf + LB_OFF, *pb_ = buf + PB_OFF;
#if defined(MVBA_KO_DMA)  // (timing-only knock-outs of the UNIT form, as for the slot form: no gathers / every row -> row 0 / the l side only / the point rows only)
    (void)kb; (void)lb; (void)pb_; (void)xk_; (void)xl_; (void)xa_;
    return;
#endif
#if defined(MVBA_KO_GATHER)
    const int xk[3] = {0, 0, 0}, xl[3] = {0, 0, 0}, xa[2] = {0, 0};
#elif defined(MVBA_KO_LSIDE)
    const int xk[3] = {xk_[0], xk_[1], xk_[2]}, xl[3] = {0, 0, 0}, xa[2] = {xa_[0], xa_[1]};
#elif defined(MVBA_KO_PROW)
    const int xk[3] = {xk_[0], xk_[1], xk_[2]}, xl[3] = {xl_[0], xl_[1], xl_[2]}, xa[2] = {0, 0};
#else
    const int (&xk)[3] = xk_, (&xl)[3] = xl_, (&xa)[2] = xa_;
#endif
    if (lane < 63) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        lds_dma16(rec_at(xk[q], dslot), kb + q * (9 * PROW));
        if (!DIAG) lds_dma16(rec_at(xl[q], dslot), lb + q * (9 * PROW));
      }
#ifndef MVBA_TIMING_NO_PB  // (timing-only build, -DMVBA_TIMING_NO_PB: what a record that also carried E^-1 would save -- wrong numbers)
      if (!DIAG) lds_dma16(pb_at(xa[0], pslot), pb_);
#endif
    }
    if (lane < 7 * (PSTEP - 18)) {  // third chunk: rows 18..20 only (a buffer holds 21 rows)
      lds_dma16(rec_at(xk[2], dslot), kb + 2 * (9 * PROW));
      if (!DIAG) lds_dma16(rec_at(xl[2], dslot), lb + 2 * (9 * PROW));
    }
    if (DIAG) {  // l-side == k-side: only the residual (slot 7) is fetched, 16 bytes per item
      if (lane < PSTEP) lds_dma16(rec_at(xl[0], 7), lb);
      lds_dma16(pb_at(xa[0], pslot), pb_);
      if (lane < 5 * PSTEP - 64) lds_dma16(pb_at(xa[1], pslot2), pb_ + 1024);
    }
  };
  // the arithmetic of one step on a landed buffer
  auto compute = [&](const char *buf, const int ns) {
    const char *kbuf = buf, *lbuf = buf + LB_OFF, *pbuf = buf + PB_OFF;
    if (it < ns) {
      const double2 *kr = reinterpret_cast<const double2 *>(kbuf + it * PROW);
      const double2 *lr = DIAG ? kr : reinterpret_cast<const double2 *>(lbuf + it * PROW);
      const double *pb = reinterpret_cast<const double *>(pbuf + it * (16 * NPS));
      const double2 kx0 = kr[0], kx1 = kr[1], kx2 = kr[2], kf = kr[3], kw0 = kr[4], kw1 = kr[5], kw2 = kr[6];
      const double2 lx0 = lr[0], lx1 = lr[1], lx2 = lr[2];
#ifdef MVBA_TIMING_NO_PB
      const double i00 = DIAG ? pb[0] : 1.0, i01 = DIAG ? pb[1] : 0.0, i02 = DIAG ? pb[2] : 0.0, i11 = DIAG ? pb[3] : 1.0, i12 = DIAG ? pb[4] : 0.0,
                   i22 = DIAG ? pb[5] : 1.0;
#else
      const double i00 = pb[0], i01 = pb[1], i02 = pb[2], i11 = pb[3], i12 = pb[4], i22 = pb[5];
#endif
      // h = E^-1 Jx_l^T (3x2), t = Jx_k h (2x2)
      const double h0x = i00 * lx0.x + i01 * lx1.x + i02 * lx2.x, h0y = i00 * lx0.y + i01 * lx1.y + i02 * lx2.y;
      const double h1x = i01 * lx0.x + i11 * lx1.x + i12 * lx2.x, h1y = i01 * lx0.y + i11 * lx1.y + i12 * lx2.y;
      const double h2x = i02 * lx0.x + i12 * lx1.x + i22 * lx2.x, h2y = i02 * lx0.y + i12 * lx1.y + i22 * lx2.y;
      double t00 = kx0.x * h0x + kx1.x * h1x + kx2.x * h2x, t01 = kx0.x * h0y + kx1.x * h1y + kx2.x * h2y;
      double t10 = kx0.y * h0x + kx1.y * h1x + kx2.y * h2x, t11 = kx0.y * h0y + kx1.y * h1y + kx2.y * h2y;
      double w0 = 0.0, w1 = 0.0;
      // SLOTS: the point row's tenth double is 1 for a point and 0 for the padding row, whose G_k term must vanish too
      const double wgt = (DIAG && SLOTS) ? pb[9] : 1.0;
      if (DIAG) {
        t00 -= SLOTS ? 0.5 * wgt : 0.5;
        t11 -= SLOTS ? 0.5 * wgt : 0.5;
        const double2 e = reinterpret_cast<const double2 *>(lbuf)[it];
        w0 = kx0.x * pb[6] + kx1.x * pb[7] + kx2.x * pb[8] - e.x;
        w1 = kx0.y * pb[6] + kx1.y * pb[7] + kx2.y * pb[8] - e.y;
        if (SLOTS) {  // a padding row gathers a real record (its range's first) next to the all-zero point row
          w0 *= wgt;
          w1 *= wgt;
        }
      }
      const double2 s0v = lr[sel0], s1v = lr[sel1], s2v = lr[sel2];
#if defined(MVBA_KO_VALU)  // (timing-only knock-out: every LDS read stays, the arithmetic shrinks to a handful of additions)
      acc[0][0] += (t00 + t01) + (t10 + t11) + (kf.x + kf.y) + (kw0.x + kw0.y) + (kw1.x + kw1.y) + (kw2.x + kw2.y) + (s0v.x + s0v.y) + (s1v.x + s1v.y) + (s2v.x + s2v.y) + w0 + w1;
      if (false)
#endif
      {
      const double sx[3] = {s0v.x, al12 * s1v.x + bx1, al12 * s2v.x};
      const double sy[3] = {s0v.y, al12 * s1v.y, al12 * s2v.y + bx1};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double v0 = t00 * sx[q] + t01 * sy[q], v1 = t10 * sx[q] + t11 * sy[q];
        // two chained FMAs into the accumulator per row (written out: `acc += a*b + c*d` compiles to
        // mul + fma + add, a third more fp64 instructions in a kernel whose VALU is 70 % busy)
        acc[0][q] = fma(kf.y, v1, fma(kf.x, v0, acc[0][q]));
        acc[1][q] += v0;
        acc[2][q] += v1;
        acc[3][q] = fma(kx0.y, v1, fma(kx0.x, v0, acc[3][q]));
        acc[4][q] = fma(kx1.y, v1, fma(kx1.x, v0, acc[4][q]));
        acc[5][q] = fma(kx2.y, v1, fma(kx2.x, v0, acc[5][q]));
        acc[6][q] = fma(kw0.y, v1, fma(kw0.x, v0, acc[6][q]));
        acc[7][q] = fma(kw1.y, v1, fma(kw1.x, v0, acc[7][q]));
        acc[8][q] = fma(kw2.y, v1, fma(kw2.x, v0, acc[8][q]));
        if (DIAG) {
          if (SLOTS) dg[q] = fma(wgt, sx[q] * sx[q] + sy[q] * sy[q], dg[q]);
          else dg[q] += sx[q] * sx[q] + sy[q] * sy[q];
          rb[q] += sx[q] * w0 + sy[q] * w1;
        }
      }
      }
    }
    // the LDS reads above are complete (their values were consumed) before this buffer is refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // SLOTS pacing: the waves of a point range keep within `lag` segments of each other, so that what they gather at
  // any moment fits their XCD's L2.  A performance hint only: a wave that waits too long (its siblings are not
  // resident: somebody else holds the CUs) stops pacing and runs on.
  bool pacing = SLOTS && pace.prog != nullptr;
  int seg = 0, seg_stop = pacing ? as_const(pace.seg_end)[0] * PSTEP : 0x7fffffff;
#ifdef MVBA_SLOT_TRACE
  const long long tr_t0 = __builtin_amdgcn_s_memrealtime();
  long long tr_wait = 0, tr_blocked = 0, tr_polls = 0;
#endif
  auto pace_at = [&](const int s0) {
    while (s0 == seg_stop) {  // (wave-uniform) this wave has left segment `seg`
      if (lane == 0) __hip_atomic_fetch_add(pace.prog + PACE_STRIDE * seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++seg;
      seg_stop = seg < pace.nseg ? as_const(pace.seg_end)[seg] * PSTEP : 0x7fffffff;
#ifdef MVBA_SLOT_TRACE
      const long long tr_a = __builtin_amdgcn_s_memrealtime();
#endif
      if (seg >= pace.lag && pacing) {  // nobody enters segment j before everybody has left segment j - lag
        // (a waiting wave polls every ~3 us: hundreds of waves polling one word at full speed saturate the
        // fabric's atomic path and slow the arrivals they are waiting for -- 15 ms per launch, measured)
        int tries = 0;
        while (__hip_atomic_fetch_add(pace.prog + PACE_STRIDE * (seg - pace.lag), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pace.need) {
          if (++tries > 1024) { pacing = false; break; }  // ~2 ms: the siblings are not running
          __builtin_amdgcn_s_sleep(127);  // (shorter sleeps, 48 / 16 / 4: 1.69 / 1.70 / 1.71 ms against 1.67)
        }
        // a wave that had to wait is ahead of the pack, one that did not is among those the pack waits for: the
        // SIMD's issue arbitration (priority, then age) should favour the latter (1.71 -> 1.62 ms together with
        // dispatching the diagonal waves first; three priority levels or none: no better / 1.73)
        if (tries > 0) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(2);
#ifdef MVBA_SLOT_TRACE
        tr_polls += tries + 1;
        tr_blocked += tries > 0;
#endif
      }
#ifdef MVBA_SLOT_TRACE
      tr_wait += __builtin_amdgcn_s_memrealtime() - tr_a;
#endif
    }
  };
  if (!SLOTS) {
    load_idx(0, ixk[0], ixl[0], ixa[0]);
    issue(wbuf, ixk[0], ixl[0], ixa[0]);
    if (PSTEP < n) load_idx(PSTEP, ixk[0], ixl[0], ixa[0]);
    for (int s0 = 0, par = 0; s0 < n; s0 += PSTEP, par ^= 1) {
      const int ns = min(PSTEP, n - s0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's rows have landed, the next step's indices too
      if (s0 + PSTEP < n) {
        issue(wbuf + (par ^ 1) * BUFSZ, ixk[0], ixl[0], ixa[0]);
        if (s0 + 2 * PSTEP < n) load_idx(s0 + 2 * PSTEP, ixk[0], ixl[0], ixa[0]);
      }
      compute(wbuf + par * BUFSZ, ns);
    }
  } else {
    // Three buffers, the gathers of TWO steps in flight: a wave of this form is one serial chain of ~1400 steps
    // for the whole launch and 9 of them share a CU, so what bounds it is steps-in-flight x latency, not
    // throughput (per-wave stamps: 1.18 us per step with one step in flight, every wave alike).
    // The step's INDEX -- 21 k-side, 21 l-side record indices and 21 points, one 256-byte row of the step-major index --
    // travels through LDS too: ONE LDS-DMA instruction per step (round 3 used seven 4-byte loads per step into two
    // register sets PINNED to v152..v167, because a value an asm load "returns" is not in its register yet and the
    // allocator once copied one out above its wait; that whole construction is gone: the indices are read from LDS
    // by ordinary ds_read into ordinary registers, after the counted wait that covers their DMA).
    // Order of the vector-memory operations: iteration s issues the gathers of step s + 2, then the index DMA of
    // step s + 4 (into the ring slot whose indices, those of step s + 1, were read an iteration ago); vmcnt retires
    // in issue order, so "all but the last iteration's operations" (`SLOT_OPS` of them: every gather and DMA is
    // unconditional, steps past the end are clamped to the last one) = step s has landed and the indices of step
    // s + 2 are in LDS, while step s + 1 and the indices of s + 3 stay in flight.
    // hipcc cannot be left to count these waits (it drains the queue -- vmcnt(0) -- before LDS reads that might alias
    // a DMA in flight), so every vector-memory operation of this loop is inline assembly the compiler knows nothing
    // about and the counted waits below are the only ones (csrc/check_isa.py checks that on the generated code).
    constexpr int SLOT_OPS = DIAG ? 7 : 8;
    const int nst = n / PSTEP, last_st = nst - 1;           // steps of this wave (n is a multiple of PSTEP here)
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)wbuf;
    const unsigned ldsx0 = lds0 + 3 * SLOT_BUF;             // the index ring
    const int *xring = reinterpret_cast<const int *>(wbuf + 3 * SLOT_BUF);
    const int *xbase = it_k + beg * SLOT_IDX;               // (`beg` = first STEP of this wave; wave-uniform)
    // (32-bit byte offsets only: with 64-bit per-lane addresses this loop needs more than the 168 registers of
    // three waves per SIMD, and a spill's scratch access would be a vector-memory operation the counted waits do
    // not know about -- the records are addressed from their range's first one, mvba_create sees to that)
    static_assert(!SLOTS || !BIG, "the slot form has no 64-bit-offset build");
    // (M0 is written in the SAME statement that uses it: it is compiler-reserved, an "m0" clobber only draws a warning, and
    // the compiler's own M0 users -- none in this kernel: check_isa.py fails the build if one appears -- set it themselves)
    auto dma = [&](int row, unsigned slot16, const void *base, unsigned lds) {  // 16 bytes per lane: base[row * 128 + slot16] -> LDS
#if defined(MVBA_KO_GATHER)  // (timing-only knock-out: every gather fetches row 0 of its array -- one line, always in L2 -- instead of its row)
      row = 0;
#endif
#if defined(MVBA_KO_DMA)     // (timing-only knock-out: no record / point-row gather at all; the counted waits are adjusted below)
      (void)row; (void)slot16; (void)base; (void)lds;
#else
      // (the byte offset row * 128 + slot16 is formed BETWEEN the write of M0 and the gather that reads it: the one wait state the
      // hardware asks for there was an s_nop in rounds 3-4 -- eight issue slots per step for nothing)
      unsigned o;
      asm volatile("s_mov_b32 m0, %4\n\tv_lshl_add_u32 %0, %1, 7, %2\n\tglobal_load_lds_dwordx4 %0, %3" : "=&v"(o) : "v"(row), "v"(slot16), "s"(base), "s"(lds) : "memory");
#endif
    };
    const unsigned lane16 = (unsigned)min(lane, 15) << 4;
    auto dma_idx = [&](int st, unsigned ring_off) {  // the 256-byte index row of step st -> ring slot st % 3 = byte offset ring_off (lanes 0..15, 16 bytes each)
#if defined(MVBA_KO_IDX)  // (timing-only knock-out: always the wave's FIRST index row -- in L2 after the first touch -- instead of a new line from HBM)
      const int *src = xbase + (size_t)min(st & 1, last_st) * SLOT_IDX;
#else
      const int *src = xbase + (size_t)min(st, last_st) * SLOT_IDX;  // wave-uniform
#endif
      const unsigned dst = ldsx0 + ring_off;
      if (lane < 16) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src), "s"(dst) : "memory");
    };
    const unsigned ds16 = (unsigned)dslot << 4, ps16 = (unsigned)pslot << 4, ps16b = (unsigned)pslot2 << 4;
    const int xr0 = min(drow, PSTEP - 1), xr1 = min(9 + drow, PSTEP - 1), xr2 = min(18 + drow, PSTEP - 1);  // this lane's rows of a step
    const int xa0 = 2 * PSTEP + min(prow, PSTEP - 1), xa1 = 2 * PSTEP + min(prow2, PSTEP - 1), xl = min(lane, PSTEP - 1);
    // (which of the three buffers / ring slots a step uses -- step % 3 -- is carried as three byte offsets that rotate once per
    // iteration: the loop computed the remainder three times a step before, a dozen scalar instructions)
    auto issue_step = [&](unsigned ring_off, unsigned buf_off) {  // gathers of a step from its indices in the ring (landed: the caller's wait saw to it)
      const int *x = reinterpret_cast<const int *>(reinterpret_cast<const char *>(xring) + ring_off);
      const unsigned buf = lds0 + buf_off;
      const unsigned kb = buf, lb = buf + LB_OFF, pb_ = buf + PB_OFF;
      const int k0 = x[xr0], k1 = x[xr1], k2 = x[xr2];
      if (!DIAG) {
        const int l0 = x[PSTEP + xr0], l1 = x[PSTEP + xr1], l2 = x[PSTEP + xr2], a0 = x[xa0];
        if (lane < 63) {
          dma(k0, ds16, rec, kb);
          dma(l0, ds16, rec, lb);
          dma(k1, ds16, rec, kb + 9 * PROW);
          dma(l1, ds16, rec, lb + 9 * PROW);
          dma(a0, ps16, PB, pb_);
        }
        if (lane < 7 * (PSTEP - 18)) {
          dma(k2, ds16, rec, kb + 18 * PROW);
          dma(l2, ds16, rec, lb + 18 * PROW);
        }
      } else {
        const int kl = x[xl], a0 = x[xa0], a1 = x[xa1];
        if (lane < 63) {
          dma(k0, ds16, rec, kb);
          dma(k1, ds16, rec, kb + 9 * PROW);
        }
        if (lane < 7 * (PSTEP - 18)) dma(k2, ds16, rec, kb + 18 * PROW);
        if (lane < PSTEP) dma(kl, 7u << 4, rec, lb);  // l-side == k-side: only the residual (slot 7) is fetched
        dma(a0, ps16, PB, pb_);
        if (lane < 5 * PSTEP - 64) dma(a1, ps16b, PB, pb_ + 1024);
      }
    };
    constexpr unsigned RING_B = SLOT_IDX * 4;
    dma_idx(0, 0);
    dma_idx(1, RING_B);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue_step(0, 0);
    dma_idx(2, 2 * RING_B);
    issue_step(last_st >= 1 ? RING_B : 0, BUFSZ);  // (a one-step wave: the clamped step 0 once more, into the buffer nobody reads)
    dma_idx(3, 0);
    unsigned b0 = 0, b1 = BUFSZ, b2 = 2 * BUFSZ, r0 = 0, r1 = RING_B, r2 = 2 * RING_B;  // offsets of step st, st + 1, st + 2
#ifdef MVBA_SLOT_TRACE  // where a step's cycles go: one s_memtime stamp (with its own lgkmcnt(0): ~40 cycles) between the phases
#define TR_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, tA, tB, tC, tD, tE, tF, tL0, tL1;
    TR_STAMP(tL0);
#else
#define TR_STAMP(v)
#endif
    for (int st = 0; st < nst; ++st) {
      TR_STAMP(tA);
      // step st has landed and the indices of step st + 2 are in the ring: everything but the last iteration's operations is done
#if defined(MVBA_KO_DMA)
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#else
      if (DIAG) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
      TR_STAMP(tB);
      pace_at(st * PSTEP);
      TR_STAMP(tC);
      issue_step(r2, b2);  // step st + 2 (past the end: the ring slot holds the indices of the clamped last step, its rows land in a buffer nobody reads)
      TR_STAMP(tD);
      dma_idx(st + 4, r1);  // (st + 4) % 3 = (st + 1) % 3: the slot whose indices were read an iteration ago
      TR_STAMP(tE);
      compute(wbuf + b0, PSTEP);
      { const unsigned tb = b0, tr = r0; b0 = b1; b1 = b2; b2 = tb; r0 = r1; r1 = r2; r2 = tr; }
#ifdef MVBA_SLOT_TRACE
      TR_STAMP(tF);
      ph0 += tB - tA; ph1 += tC - tB; ph2 += tD - tC; ph3 += tE - tD; ph4 += tF - tE;
#endif
    }
#ifdef MVBA_SLOT_TRACE
    TR_STAMP(tL1);
    if (pace.trace && lane == 0) {
      pace.trace[8] = (long long)ph0; pace.trace[9] = (long long)ph1; pace.trace[10] = (long long)ph2; pace.trace[11] = (long long)ph3;
      pace.trace[12] = (long long)ph4; pace.trace[13] = (long long)(tL1 - tL0);
    }
#endif
    static_assert(SLOT_OPS == (DIAG ? 7 : 8), "counted wait");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped gathers still in flight land in this wave's LDS
  }
  // J_C row i = rs_i * (record columns): f | u,v (1/f0) | t (-Jx) | omega; the same factors per column
  const double cs0 = cg == 1 ? -1.0 : 1.0, cs12 = cg == 0 ? cu : cs0;
  if (SLOTS) {  // every slot holds a finished block of its own: no sum over lanes
    if (pace.prog != nullptr && lane == 0)
      for (; seg < pace.nseg; ++seg) __hip_atomic_fetch_add(pace.prog + PACE_STRIDE * seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef MVBA_SLOT_TRACE
    if (pace.trace && lane == 0) {
      int hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
      pace.trace[0] = tr_t0; pace.trace[1] = __builtin_amdgcn_s_memrealtime(); pace.trace[2] = tr_wait; pace.trace[3] = tr_blocked;
      pace.trace[4] = tr_polls; pace.trace[5] = hwid; pace.trace[6] = xcc; pace.trace[7] = n / PSTEP;
    }
#endif
    const int u = it < PSTEP ? slot_unit[it] : -1;
    if (u >= 0) {
      double *o = out + (size_t)u * UNIT_STRIDE;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double rs = (i == 1 || i == 2) ? cu : ((i >= 3 && i < 6) ? -1.0 : 1.0);
#pragma unroll
        for (int q = 0; q < 3; ++q) o[9 * i + 3 * cg + q] = -4.0 * rs * (q == 0 ? cs0 : cs12) * acc[i][q];
      }
      if (DIAG) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const double cs = q == 0 ? cs0 : cs12;
          o[81 + 3 * cg + q] = 2.0 * c * cs * cs * dg[q];
          o[90 + 3 * cg + q] = 2.0 * cs * rb[q];
        }
      }
    }
    return;
  }
  // ---- sum over the 21 item lanes of each column group (fixed tree), lanes 0..2 write the partial
  auto tree = [&](double v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      const double o = __shfl_down(v, 3 * off, 64);
      if (it < off && it + off < PSTEP) v += o;
    }
    return v;
  };
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double rs = (i == 1 || i == 2) ? cu : ((i >= 3 && i < 6) ? -1.0 : 1.0);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double v = tree(acc[i][q]);
      if (lane < 3) out[9 * i + 3 * cg + q] = -4.0 * rs * (q == 0 ? cs0 : cs12) * v;
    }
  }
  if (DIAG) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double d = tree(dg[q]), r = tree(rb[q]);
      const double cs = q == 0 ? cs0 : cs12;
      if (lane < 3) {
        out[81 + 3 * cg + q] = 2.0 * c * cs * cs * d;  // c * diag(G_k)   (ref :123-125)
        out[90 + 3 * cg + q] = 2.0 * cs * r;           // 2 Jc_k^T (Jx_k E^-1 dP - e)
      }
    }
  }
}

#if defined(MVBA_HREC_TIMING)  // timing-only variant of both kernels (round 5, measured and not built: see the header)
#include "mvba_hrec_timing.h"
#endif
#if defined(MVBA_FS)
#include "mvba_fs.h"
#endif


// One wave per block: a wave works alone, and in a wider block its LDS and wave slots stay taken until
// the block's slowest wave has finished (4 waves per block: 1.945 ms, 2: 1.92, 1: 1.89 at config 3).
template <bool BIG>
__device__ __forceinline__ void schur_pairs_wave(const int4 *__restrict__ units, const int *__restrict__ q_ptr,
                                                       const int *__restrict__ q_units, int *__restrict__ head,
                                                       const int *__restrict__ it_k, const int *__restrict__ it_l,
                                                       const int *__restrict__ it_a, const double2 *__restrict__ rec,
                                                       const double *__restrict__ PB, double c, double f0,
                                                       double *__restrict__ partial) {
  extern __shared__ char smem_pairs[];
  const int lane = threadIdx.x;
  char *wbuf = smem_pairs;
  // ---- take one unit.  Static (default, head == nullptr): block b takes entry b / 8 of queue b % 8 -- no
  // atomic on the critical path; it relies on the round-robin block -> XCD placement for locality
  // only, never for correctness (1.90 -> 1.87 ms).  Dynamic (MVBA_PAIR_STATIC=0): own XCD's queue
  // first (XCC_ID), then the others; every queue entry is taken exactly once, the grid has as many
  // waves as there are units and a wave takes at most one.
  int pos = -1;
  if (!head) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    if (q < q_ptr[x + 1] - q_ptr[x]) pos = q_ptr[x] + q;
  } else if (lane == 0) {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    for (int t = 0; t < 8 && pos < 0; ++t) {
      const int x = (xcc + t) & 7, len = q_ptr[x + 1] - q_ptr[x];
      if (len <= 0) continue;
      const int q = atomicAdd(&head[x], 1);
      if (q < len) pos = q_ptr[x] + q;
    }
  }
  pos = __builtin_amdgcn_readfirstlane(pos);
  if (pos < 0) return;
  // unit id (where its partial goes) and descriptor, both in queue order: two independent SCALAR loads
  // (pos is wave-uniform; read through the constant address space so that beg and n live in SGPRs and
  // the per-step index bases are scalar arithmetic)
  const int u = as_const(q_units)[pos];
  const int MVBA_CONST_AS *udp = as_const(reinterpret_cast<const int *>(units)) + 4 * (size_t)pos;
  const int ud_x = udp[0], ud_y = udp[1], ud_z = udp[2], ud_w = udp[3];
  const long long beg = ((long long)ud_y << 32) | (unsigned)ud_x;
  const int n = ud_z, cam_k = (int)((unsigned)ud_w >> 16), cam_l = ud_w & 0xffff;
  double *out = partial + (size_t)u * UNIT_STRIDE;
#if defined(MVBA_HREC_TIMING)  // timing-only (wrong numbers): slot 7 of the records stands in for the residual array
  if (cam_k == cam_l) schur_pairs_unit_hrec<true, BIG>(wbuf, lane, beg, n, it_k, it_l, it_a, rec, rec + 7, PB, c, 1.0 / f0, out);
  else schur_pairs_unit_hrec<false, BIG>(wbuf, lane, beg, n, it_k, it_l, it_a, rec, rec + 7, PB, c, 1.0 / f0, out);
#else
  if (cam_k == cam_l) schur_pairs_unit<true, BIG>(wbuf, lane, beg, n, it_k, it_l, it_a, rec, PB, c, 1.0 / f0, out);
  else schur_pairs_unit<false, BIG>(wbuf, lane, beg, n, it_k, it_l, it_a, rec, PB, c, 1.0 / f0, out);
#endif
}

#define MVBA_PAIRS_ARGS                                                                                                   \
  const int4 *__restrict__ units, const int *__restrict__ q_ptr, const int *__restrict__ q_units, int *__restrict__ head, \
      const int *__restrict__ it_k, const int *__restrict__ it_l, const int *__restrict__ it_a,                          \
      const double2 *__restrict__ rec, const double *__restrict__ PB, double c, double f0, double *__restrict__ partial
// (two plain kernels rather than one template, so that profiles show one stable name per variant)
__global__ __launch_bounds__(64, 3) void k_schur_pairs(MVBA_PAIRS_ARGS) {
  schur_pairs_wave<false>(units, q_ptr, q_units, head, it_k, it_l, it_a, rec, PB, c, f0, partial);
}
__global__ __launch_bounds__(64, 3) void k_schur_pairs_big(MVBA_PAIRS_ARGS) {  // 64-bit record / point-block offsets
  schur_pairs_wave<true>(units, q_ptr, q_units, head, it_k, it_l, it_a, rec, PB, c, f0, partial);
}

// ------------------------------------------------------------------ K3 (slot-resident form)
// The pair-major kernel above reads every record ~5 times from beyond its L2 (89 M line misses for 10 M records at
// config 3): a unit sweeps 1/16 of the points for ONE pair, so what the ~300 units in flight on an XCD touch at
// any moment is spread over 80 MB of records.  Here the roles of the 21 item rows of a step are transposed:
//   slot     a 3-lane slot (item row `it`) owns ONE list -- the items of one (pair, sub-list) inside one point
//            range -- for the whole launch and keeps that pair's 9x9 block in its 27 registers per lane: still no
//            scatter, no atomic, no cross-lane sum at all (the fixed-order tree is gone), one partial per list
//   range    the points are cut into 8 ranges of equal item count, range r = the blocks b with b % 8 == r = XCD r
//            under the round-robin block placement (locality only, never correctness): a record is needed by ONE
//            XCD, and ALL lists of the range (m (m+1) / 2 pairs + the diagonal pairs' sub-lists: 5.9 k slots =
//            284 waves at m = 100) are resident on that XCD at once and sweep the range's points together
//   steps    built once on the host (mvba_create): the 21 lists of a wave are merged into steps with a bounded skew --
//            a slot whose next item lies more than `skew` observations ahead of the wave's slowest slot idles
//            for a step (its row points at the all-zero record / point row: the arithmetic runs and adds
//            exact zeros) -- so that what a wave touches at any moment fits its XCD's L2 next to its siblings'
// Same staging (LDS-DMA of whole lines), same arithmetic and the same partial -> k_schur_reduce path as above.
__device__ __forceinline__ void schur_slots_wave(const int4 *__restrict__ wdesc, const int *__restrict__ wunits,
                                                 const int *__restrict__ it_k, const int *__restrict__ it_l,
                                                 const int *__restrict__ it_a, const double2 *__restrict__ rec,
                                                 const double *__restrict__ PB, double c, double f0,
                                                 double *__restrict__ partial, int *__restrict__ head, int nR, int wpr,
                                                 const int *__restrict__ seg_end, int *__restrict__ prog, int nseg, int lag,
                                                 long long *__restrict__ trace, const long long *__restrict__ range_o0) {
  extern __shared__ char smem_pairs[];
  // which wave of which range: static (head == nullptr) block b IS wave b / nR of range b % nR; dynamic: the wave
  // reads the XCD it runs on and takes the next wave of a range of that XCD (r % 8 == XCC_ID), then of the others
  int bid = blockIdx.x;
  if (head) {
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    bid = -1;
    if (threadIdx.x == 0) {
      for (int t = 0; t < nR && bid < 0; ++t) {
        const int r = ((xcc & 7) + t) % nR;
        const int q = atomicAdd(&head[r], 1);
        if (q < wpr) bid = nR * q + r;
      }
    }
    bid = __builtin_amdgcn_readfirstlane(bid);
    if (bid < 0) return;
  }
  const int MVBA_CONST_AS *dp = as_const(reinterpret_cast<const int *>(wdesc)) + 4 * (size_t)bid;
  const int d_x = dp[0], d_y = dp[1], nsteps = dp[2], flags = dp[3];
  if (nsteps <= 0) return;  // (wave-uniform) a wave whose lists are all empty in this range: no units, nothing to write
  const long long beg = ((long long)d_y << 32) | (unsigned)d_x;
  const int *su = wunits + (size_t)bid * PSTEP;
  // flags: bit 0 diagonal wave | bits 8..19 live waves of this (round, range) | bits 20..30 round
  const int r = bid % nR, live = (flags >> 8) & 0xfff, round = (flags >> 20) & 0x7ff;
  // the record indices of the step rows are RELATIVE to the range's first observation: the 32-bit byte offsets of the
  // gathers then span a range's records (4 GiB = 33.5 M observations per range), not the scene's
  const double2 *rec_r = rec + (size_t)as_const(range_o0)[r] * REC;
  SlotPace pace{nullptr, nullptr, 0, 0, 2, trace ? trace + 16 * (size_t)bid : nullptr};
  if (prog) pace = SlotPace{seg_end + (size_t)bid * nseg, prog + ((size_t)round * nR + r) * nseg * PACE_STRIDE, live, nseg, lag, pace.trace};
#if defined(MVBA_HREC_TIMING)  // timing-only: today's records read with the new access pattern and arithmetic -- wrong numbers
  const double2 *res_r = rec_r + 7;  // (the residual "array": slot 7 of the records, 128-byte stride -> dma16 takes row << 4 ... see below)
  if (flags & 1)
    schur_slots_hrec<true>(smem_pairs, (int)threadIdx.x, beg, nsteps * PSTEP, it_k, rec_r, res_r, PB, c, 1.0 / f0, partial, su, pace, 0);
  else
    schur_slots_hrec<false>(smem_pairs, (int)threadIdx.x, beg, nsteps * PSTEP, it_k, rec_r, res_r, PB, c, 1.0 / f0, partial, su, pace, 0);
#elif defined(MVBA_FS)
  if (flags & 1) schur_slots_fs<true>(smem_pairs, (int)threadIdx.x, beg, nsteps, it_k, rec_r, PB, c, 1.0 / f0, partial, su, pace);
  else schur_slots_fs<false>(smem_pairs, (int)threadIdx.x, beg, nsteps, it_k, rec_r, PB, c, 1.0 / f0, partial, su, pace);
#else
  if (flags & 1)
    schur_pairs_unit<true, false, true>(smem_pairs, (int)threadIdx.x, beg, nsteps * PSTEP, it_k, it_l, it_a, rec_r, PB, c, 1.0 / f0, partial, su, pace);
  else
    schur_pairs_unit<false, false, true>(smem_pairs, (int)threadIdx.x, beg, nsteps * PSTEP, it_k, it_l, it_a, rec_r, PB, c, 1.0 / f0, partial, su, pace);
#endif
}
#define MVBA_SLOTS_ARGS                                                                                                  \
  const int4 *__restrict__ wdesc, const int *__restrict__ wunits, const int *__restrict__ it_k, const int *__restrict__ it_l, \
      const int *__restrict__ it_a, const double2 *__restrict__ rec, const double *__restrict__ PB, double c, double f0,  \
      double *__restrict__ partial, int *__restrict__ head, int nR, int wpr, const int *__restrict__ seg_end,            \
      int *__restrict__ prog, int nseg, int lag, long long *__restrict__ trace, const long long *__restrict__ range_o0
#ifndef MVBA_SLOT_WAVES_PER_SIMD
#if defined(MVBA_FS)
#define MVBA_SLOT_WAVES_PER_SIMD 1
#else
#define MVBA_SLOT_WAVES_PER_SIMD 3
#endif
#endif
__global__ __launch_bounds__(64, MVBA_SLOT_WAVES_PER_SIMD) void k_schur_slots(MVBA_SLOTS_ARGS) {
  schur_slots_wave(wdesc, wunits, it_k, it_l, it_a, rec, PB, c, f0, partial, head, nR, wpr, seg_end, prog, nseg, lag, trace, range_o0);
}

// One thread per element of a pair's block: the pair's unit partials in unit order -> packed strips.
// One block per PAIR (blockIdx.x = packed pair index); the partials are loaded eight at a time (one
// memory latency per eight units instead of one per unit: a diagonal pair has ~130 of them) and
// added in unit order, so the result does not depend on the schedule.
__global__ __launch_bounds__(128) void k_schur_reduce(int m, const int *__restrict__ unit_ptr,
                                                      const double *__restrict__ partial, double *__restrict__ Afull,
                                                      double *__restrict__ bfull, int *__restrict__ head) {
  if (blockIdx.x == 0 && threadIdx.x < 64) head[threadIdx.x] = 0;  // work queues for the next launch
  // pair index -> (k, l): pairs of strip k start at k m - k (k - 1) / 2
  const long long p = blockIdx.x;
  int k = (int)((2.0 * m + 1.0 - sqrt((2.0 * m + 1.0) * (2.0 * m + 1.0) - 8.0 * (double)p)) * 0.5);
  while ((long long)k * m - (long long)k * (k - 1) / 2 > p) --k;
  while ((long long)(k + 1) * m - (long long)(k + 1) * k / 2 <= p) ++k;
  const int l = k + (int)(p - ((long long)k * m - (long long)k * (k - 1) / 2));
  const int u0 = unit_ptr[p], u1 = unit_ptr[p + 1];
  const int e = threadIdx.x;
  if (e >= (k == l ? 99 : 81)) return;
  auto ordered_sum = [&](int off) {
    double v = 0.0;
    int uu = u0;
    for (; uu + 8 <= u1; uu += 8) {
      double t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = partial[(size_t)(uu + q) * UNIT_STRIDE + off];
#pragma unroll
      for (int q = 0; q < 8; ++q) v += t[q];
    }
    for (; uu < u1; ++uu) v += partial[(size_t)uu * UNIT_STRIDE + off];
    return v;
  };
  const double v = ordered_sum(e);
  double *Ak = Afull + strip_offset(k, m);
  const int Wk = 9 * (m - k);
  if (e < 81) {
    const int i = e / 9, j = e - 9 * i;
    const double d = (k == l && i == j) ? ordered_sum(81 + i) : 0.0;  // Marquardt damping of G_k's diagonal
    Ak[(size_t)i * Wk + 9 * (l - k) + j] = v + d;
  } else if (e >= 90) {
    bfull[9 * k + (e - 90)] = v;
  }
}

// ------------------------------------------------------------------ K3 (dense visibility: every point seen by every camera)
// The reference's own scenes (euclidiean_reconstruction.py, affine_reconstruction.py, BASELINE config 2) and the pipeline test at
// 1 M points x 12 images have FULL visibility and a dozen or two cameras.  The pair-major forms above then walk m (m + 1) / 2 items
// per point through gathers and an index of their own (78 items and 0.9 GB of index per million points at 12 cameras: 4.9 ms per
// solve, 63 ps per item against 30 at config 3) although a point's m records are ONE contiguous range and every pair is present.
// With E_a^-1 = R_a R_a^T (3 x 3 Cholesky) and the scaled camera Jacobian J~_ak (2 x 9: f | u, v (1 / f0) | t (-J_X) | omega)
//   G_a = [R_a^T J_Xak^T J~_ak]_k   (3 x 9m),        A = blockdiag(2 H_k + c diag(2 H_k)) - 4 sum_a G_a^T G_a,
//   H_k = sum_a J~_ak^T J~_ak,                        b_k = 2 sum_a J~_ak^T (J_Xak E_a^-1 dP_a - e_ak)
// -- one symmetric rank-3N update of a 9m x 9m matrix: a GEMM, on the f64 matrix cores, over records streamed ONCE.
// A workgroup takes a chunk of points (dense_ch: 4 or 8) at a time: their records and point rows go into LDS with contiguous 16-byte loads; a thread per
// observation forms J_X R and the right-hand-side vector w; the rows of G are written to LDS once (each is read by up to T tile
// pairs); wave w then owns the 16 x 16 tile pairs w, w + 4, ... of the upper triangle (four consecutive rows of G -- of whichever
// points -- are one v_mfma_f64_16x16x4: no padding of K) and the per-camera tiles [J~ | w]^T [J~ | w] (two points per MFMA) of the
// cameras w, w + 4, ...  Partial tiles per workgroup, summed in workgroup order by k_schur_dense_finish: no atomics, bitwise
// reproducible.  No index at all: mvba_create skips the pair-major index for such scenes.
typedef double mvba_d4 __attribute__((ext_vector_type(4)));
#ifndef MVBA_DENSE_KO
#define MVBA_DENSE_KO 0  // (timing-only builds: bit 0 no main MFMAs, 1 no rows of G, 2 no per-observation phase, 3 no per-camera MFMAs)
#endif
// Points per chunk (= producer waves) and workgroups per CU.  Up to 7 tiles (12 cameras) the kernel needs at most 126 registers: four
// waves fit a SIMD, so TWO 8-wave workgroups of 4-point chunks (~53 KB of LDS each) share a CU and fill each other's barrier waits
// (1.335 -> 1.260 ms at 1 M x 12, 1.48 -> 1.32 at 2 M x 6; three workgroups: worse again).  8 tiles take 154 registers -- three waves
// per SIMD -- and stay with one 12-wave workgroup of 8-point chunks (two of the small ones cannot both be resident: 1.65 -> 1.82 ms
// at 14 cameras); beyond that eight consumer + four producer waves.  (tools/dense_ch.sh, profiles/r05_dense_form.txt)
constexpr int dense_ch(int T) { return T == 8 ? 8 : 4; }
constexpr int dense_wgs(int T) { return T <= 7 ? 2 : 1; }
constexpr int DENSE_MAX_TILES = 12;  // 9 m <= 192: m <= 21 cameras (78 tile pairs: 20 accumulators of 4 doubles per lane)
__device__ __forceinline__ int dense_tile_elem(int row, int col) { return ((row >> 2) << 6) | ((row & 3) << 4) | col; }  // C/D layout: col = l & 15, row = (l >> 4) + 4 reg

constexpr int dense_pair_ti(int p, int T) { int a = 0; while (p >= T - a) { p -= T - a; ++a; } return a; }
constexpr int dense_pair_tj(int p, int T) { int a = 0; while (p >= T - a) { p -= T - a; ++a; } return a + p; }
#ifndef MVBA_DENSE_NC_SMALL
#define MVBA_DENSE_NC_SMALL 4
#endif
constexpr int dense_consumers(int T) { return T <= 8 ? MVBA_DENSE_NC_SMALL : 8; }  // consumer waves: at most ~10 tile pairs (40 accumulator doubles) each
constexpr bool dense_tile_used(int u, int T, int wave) {  // does consumer wave `wave` own a pair with tile u?
  const int P = T * (T + 1) / 2, NC = dense_consumers(T);
  for (int p = wave; p < P; p += NC)
    if (dense_pair_ti(p, T) == u || dense_pair_tj(p, T) == u) return true;
  // (its spare slots repeat the last pair)
  return (P + NC - 1) / NC * NC - NC + wave >= P && (dense_pair_ti(P - 1, T) == u || dense_pair_tj(P - 1, T) == u);
}
// the main products of one chunk for consumer wave WAVE: its tile pairs p = NC q + WAVE are known at compile time, so the operand of a
// tile is read from LDS ONCE per four rows of G and used from its register by every pair that needs it (two reads per MFMA, with
// the tiles indexed at run time, kept the LDS half busy under the matrix cores).  (The pair -> tile arithmetic goes through class
// templates: called as constexpr FUNCTIONS inside the unrolled loops it was evaluated at run time, with the registers indexed
// through s_set_gpr_idx.)
template <int T, int WAVE, int Q>
struct DensePair {
  static constexpr int P = T * (T + 1) / 2, NC = dense_consumers(T);
  static constexpr int p = NC * Q + WAVE < P ? NC * Q + WAVE : P - 1;  // (a spare slot repeats the last pair: computed, never stored)
  static constexpr int ti = dense_pair_ti(p, T), tj = dense_pair_tj(p, T);
};
template <int T, int WAVE, int U>
struct DenseUsed { static constexpr bool v = dense_tile_used(U, T, WAVE); };
template <int T, int WAVE, int... U>
__device__ __forceinline__ void dense_load_tiles(const double *row, double (&t)[T], std::integer_sequence<int, U...>) {
  ((t[U] = DenseUsed<T, WAVE, U>::v ? row[16 * U] : 0.0), ...);
}
template <int T, int WAVE, int... Q>
__device__ __forceinline__ void dense_mfma_pairs(const double (&ts)[T], const double (&t)[T], mvba_d4 *acc, std::integer_sequence<int, Q...>) {
  ((acc[Q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ts[DensePair<T, WAVE, Q>::ti], t[DensePair<T, WAVE, Q>::tj], acc[Q], 0, 0, 0)), ...);
}
template <int T, int WAVE, int CH>
__device__ __forceinline__ void dense_main_mfma(const double *sG, const double *sSgn, int li, int lk, mvba_d4 *acc) {
  constexpr int P = T * (T + 1) / 2, NC = dense_consumers(T), NPW = (P + NC - 1) / NC, W = 16 * T;
#pragma unroll
  for (int g = 0; g < 3 * CH / 4; ++g) {  // four rows of G per MFMA: sum_r s_r g_r^T g_r (s = +-1: E^-1 = R S R^T), the sign on the left operand
    double t[T], ts[T];
    dense_load_tiles<T, WAVE>(sG + (size_t)(4 * g + lk) * W + li, t, std::make_integer_sequence<int, T>{});
    const double sg = sSgn[4 * g + lk];
#pragma unroll
    for (int u = 0; u < T; ++u) ts[u] = t[u] * sg;
    dense_mfma_pairs<T, WAVE>(ts, t, acc, std::make_integer_sequence<int, NPW>{});
  }
}

// Two roles in a workgroup, one barrier per chunk: waves 0-3 multiply chunk i (main pairs and camera tiles out of the buffers of
// parity i & 1) while the producer waves -- one per point of a chunk -- build chunk i + 1 into the other buffers, each taking its
// point from its records (fetched into registers a chunk earlier) to the rows of G on its own, so the producers need no barrier
// among themselves.  (With every wave doing every phase in turn -- four barriers per chunk -- the workgroups of a CU ran in
// lockstep and the phases never overlapped: 1.55 ms at 1 M x 12 for 0.81 ms of MFMA phase; with four producer waves of two points
// each the producers were the longer role: 1.90 ms.)
#ifdef MVBA_DENSE_TRACE  // (timing-only build, tools/dense_trace.sh: where a role's time goes -- work or the chunk barrier)
__device__ long long g_dense_trace[1024 * 16 * 4];  // per workgroup and wave: role work, barrier wait, total, chunks (shader clock)
#define DT_NOW() ((long long)__builtin_readcyclecounter())
#define DT_BARRIER() do { const long long t0_ = DT_NOW(); __syncthreads(); dt_wait += DT_NOW() - t0_; } while (0)
#else
#define DT_BARRIER() __syncthreads()
#endif
template <int T, bool TABLE>  // TABLE: the records of a point through obs_of (missing observations), otherwise one contiguous range
__global__ __launch_bounds__(64 * (dense_consumers(T) + dense_ch(T))) void k_schur_dense(const double2 *__restrict__ rec, const double *__restrict__ PB, const int *__restrict__ obs_of,
                                                     long long N, int m, double cu, double *__restrict__ part) {
  constexpr int NC = dense_consumers(T);           // consumer waves (4 + 8 producers up to 8 tiles, 8 + 4 beyond)
  constexpr int P = T * (T + 1) / 2, NPW = (P + NC - 1) / NC, W = 16 * T, NCW = ((16 * T) / 9 + NC - 1) / NC;
  constexpr int CH = dense_ch(T);       // points per chunk = producer waves (the double-buffered rows must fit the LDS beside each other)
  constexpr int MMAX = (16 * T) / 9;               // cameras at most
  constexpr int NTHR = 64 * (NC + CH);
  extern __shared__ double2 dsm[];
  double *sG = reinterpret_cast<double *>(dsm);                     // [2][3 CH][W]
  double *sB = sG + (size_t)2 * 3 * CH * W;                         // [2][CH m][2][16]: rows x, y of [J~ (9) | w | 0 ...]
  double2 *sScr = reinterpret_cast<double2 *>(sB + (size_t)2 * CH * m * 32);  // per producer wave: records [m][8], point row [8]
  double *sFlag = reinterpret_cast<double *>(sScr + (size_t)CH * (m * REC + 8));  // per producer wave: [m] 1 for an observation, 0 for a missing one
  double *sSgn = sFlag + (size_t)CH * m;                            // [2][3 CH]: the signs of the rows of G
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;  // (wave through readfirstlane -- scalar address arithmetic for the producers -- made every shape slower: 1.36 -> 1.55 ms at 1 M x 12, 2.95 -> 12.6 at 20 cameras)
  const long long n_chunks = (N + CH - 1) / CH;
  // what no phase ever writes stays zero: the columns of G beyond 9 m, the columns 10..15 of the camera rows
  for (int e = threadIdx.x; e < 2 * 3 * CH * W; e += NTHR) sG[e] = 0.0;
  for (int e = threadIdx.x; e < 2 * CH * m * 32; e += NTHR) sB[e] = 0.0;
  __syncthreads();
  if (wave >= NC) {
    // ---------------- producer: point pw of every chunk of this workgroup.  (Without the staging -- a lane per (camera, column)
    // fetching its four record slots itself -- the per-lane loads cost more than the staging saves: 1.82 against 1.40 ms.)
    const int pw = wave - NC;
#ifndef MVBA_DENSE_PRIO
#define MVBA_DENSE_PRIO 1
#endif
    // The producers are the longer role (tools/dense_trace.py: 5,350 cycles of work per chunk against the consumers' 4,480 + 1,400 at
    // the barrier, 1 M x 12) and, as the later-dispatched waves of their SIMD, the losers of its issue arbitration (older first at
    // equal priority): they ask for the higher priority once, here.
    if (MVBA_DENSE_PRIO) __builtin_amdgcn_s_setprio(MVBA_DENSE_PRIO);
    double2 *sR = sScr + (size_t)pw * (m * REC + 8), *sP = sR + (size_t)m * REC;
    constexpr int NPRE = (MMAX * REC + 63) / 64, NIT = (MMAX * 10 + 63) / 64;
    // TWO chunks' records in flight per wave (sets A and B, used in turn): with one, a CU had 8 waves x 1.5 KB outstanding against
    // ~2 us of loaded memory latency -- 1.5 TB/s over the chip, which is where the kernel sat at every camera count (~1.0 ms per
    // million points x 12 cameras whatever the matrix cores had to do)
    double2 preA[NPRE], prepbA, preB[NPRE], prepbB;
    bool lvA, lvB;
    int oidA[NPRE], oidB[NPRE], oid_next[NPRE];    // (obs_of != nullptr) the observations of a set's point / of the point fetched next
    double *sF = sFlag + (size_t)pw * m;
    auto fetch_ids = [&](long long ch) {           // the table row of chunk ch's point (one entry per record: eight lanes share it)
      const long long a = ch * CH + pw;
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int e = lane + 64 * u;
        const int got = obs_of[(size_t)min(a, N - 1) * m + min(e >> 3, m - 1)], msk = (a < N && e < m * REC) ? -1 : 0;
        oid_next[u] = (got & msk) | ~msk;          // (-1 outside; the load itself is unconditional, see fetch)
      }
    };
    auto fetch = [&](long long ch, double2 (&pre)[NPRE], int (&oid)[NPRE], double2 &prepb, bool &lv) {  // this wave's records and point row of chunk ch (zeros past the last point)
      // The loads are UNCONDITIONAL on clamped addresses and what must be zero (a missing observation, a point past the end) is
      // zeroed on the bit pattern in build(): behind a lane-dependent branch the compiler cannot count the loads in flight and
      // waits for all of them (vmcnt(0)) -- the other set's too, which is the one meant to stay in flight
      const long long a = ch * CH + pw;
      const bool live = lv = a < N;
      if (TABLE) {                                 // through the table read a chunk earlier
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
          oid[u] = oid_next[u];
          pre[u] = rec[(size_t)max(oid[u], 0) * REC + ((lane + 64 * u) & 7)];
        }
      } else {
        const double2 *src = rec + (size_t)min(a, N - 1) * m * REC;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
          pre[u] = src[min(lane + 64 * u, m * REC - 1)];
          oid[u] = live ? 0 : -1;
        }
      }
      prepb = reinterpret_cast<const double2 *>(PB + (size_t)min(a, N - 1) * PBS)[lane & 7];
    };
    auto keep = [](bool c, double2 v) -> double2 {  // c ? v : 0 without a branch the load could sink under
      const long long msk = c ? -1LL : 0LL;
      return double2{__longlong_as_double(__double_as_longlong(v.x) & msk), __longlong_as_double(__double_as_longlong(v.y) & msk)};
    };
    auto build = [&](int buf, const double2 (&pre)[NPRE], const int (&oid)[NPRE], const double2 &prepb, bool lv) {  // registers -> the rows of G and the camera rows of this wave's point in buffer `buf`
#pragma unroll
      for (int u = 0; u < NPRE; ++u) {
        const int e = lane + 64 * u;
        if (e < m * REC) sR[e] = keep(oid[u] >= 0, pre[u]);
        if (e < m * REC && (e & 7) == 0) sF[e >> 3] = oid[u] >= 0 ? 1.0 : 0.0;
      }
      sP[lane & 7] = keep(lv, prepb);             // (every lane, eight copies of each slot: a store under `lane < 8` drew a vmcnt(0))
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // point row: E^-1 (6) | E^-1 dP (3) | sign code | R lower triangular (r00 r10 r20 r11 r21 r22) of E^-1 = R S R^T (k_point_inv), the same for every lane
      const double *pb = reinterpret_cast<const double *>(sP);
      const double d0 = pb[6], d1 = pb[7], d2 = pb[8];
      const double r00 = pb[10], r10 = pb[11], r20 = pb[12], r11 = pb[13], r21 = pb[14], r22 = pb[15];
      double *gB = sB + ((size_t)buf * CH + pw) * m * 32;
      double *gG = sG + ((size_t)buf * 3 * CH + (size_t)3 * pw) * W;
      if (lane < 3) sSgn[(size_t)buf * 3 * CH + 3 * pw + lane] = (((int)pb[9] - 1) >> lane) & 1 ? -1.0 : 1.0;
      // a lane per (camera k, column j <= 9): column j of J~ (f | u, v (1 / f0) | t (-J_X) | omega) into the two camera rows and
      // G[r][9 k + j] = (J_X R)[:, r] . J~[:, j] (E^-1 = R R^T); j = 9: w = J_X E^-1 dP - e, the tenth column of the camera rows
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int e = lane + 64 * u, k = e / 10, j = e - 10 * k;
        if (e < 10 * m && !(MVBA_DENSE_KO & 2)) {
          const double2 *r = sR + (size_t)k * REC;
          const double2 x0 = r[0], x1 = r[1], x2 = r[2];
          const double2 cv = r[j == 0 ? 3 : (j < 6 ? (j < 3 ? 0 : j - 3) : (j < 9 ? j - 2 : 7))];
          if (j == 9) {
            gB[(size_t)k * 32 + 9] = x0.x * d0 + x1.x * d1 + x2.x * d2 - cv.x;
            gB[(size_t)k * 32 + 25] = x0.y * d0 + x1.y * d1 + x2.y * d2 - cv.y;
          } else {
            const double sg = (j >= 3 && j < 6) ? -1.0 : 1.0, live = sF[k] * cu;  // (the constant columns 1 / f0 of an observation that exists)
            const double jx = j == 1 ? live : (j == 2 ? 0.0 : sg * cv.x), jy = j == 2 ? live : (j == 1 ? 0.0 : sg * cv.y);
            gB[(size_t)k * 32 + j] = jx;
            gB[(size_t)k * 32 + 16 + j] = jy;
            double *g = gG + 9 * k + j;
            g[0] = (x0.x * r00 + x1.x * r10 + x2.x * r20) * jx + (x0.y * r00 + x1.y * r10 + x2.y * r20) * jy;
            g[W] = (x1.x * r11 + x2.x * r21) * jx + (x1.y * r11 + x2.y * r21) * jy;
            g[2 * W] = (x2.x * r22) * jx + (x2.y * r22) * jy;
          }
        }
      }
    };
    long long ch = blockIdx.x;
    const long long gs = gridDim.x;
    // (a fetch past the last chunk loads nothing: zero records nobody reads; fetch_ids likewise -1)
    if (TABLE) fetch_ids(ch);
    fetch(ch, preA, oidA, prepbA, lvA);
    if (TABLE) fetch_ids(ch + gs);
    fetch(ch + gs, preB, oidB, prepbB, lvB);
    if (TABLE) fetch_ids(ch + 2 * gs);
    build(0, preA, oidA, prepbA, lvA);
    fetch(ch + 2 * gs, preA, oidA, prepbA, lvA);
    if (TABLE) fetch_ids(ch + 3 * gs);
#ifdef MVBA_DENSE_TRACE
    long long dt_wait = 0, dt_n = 0;
    const long long dt_t0 = DT_NOW();
#endif
    DT_BARRIER();
    // One barrier per chunk, as the consumers; the sets alternate: B holds chunk ch + 1, A chunk ch + 2.  Nothing in the body is
    // conditional (behind `if (ch + gs < n_chunks)` the compiler lost count of the loads in flight and waited vmcnt(0) for both
    // sets): past the last chunk a build writes a chunk of zeros into the buffer nobody reads any more.
    for (int b = 0; ch < n_chunks;) {
      build(b ^ 1, preB, oidB, prepbB, lvB);
      fetch(ch + 3 * gs, preB, oidB, prepbB, lvB);
      if (TABLE) fetch_ids(ch + 4 * gs);
      DT_BARRIER();
      ch += gs, b ^= 1;
#ifdef MVBA_DENSE_TRACE
      ++dt_n;
#endif
      if (ch >= n_chunks) break;
      build(b ^ 1, preA, oidA, prepbA, lvA);
      fetch(ch + 3 * gs, preA, oidA, prepbA, lvA);
      if (TABLE) fetch_ids(ch + 4 * gs);
      DT_BARRIER();
      ch += gs, b ^= 1;
#ifdef MVBA_DENSE_TRACE
      ++dt_n;
#endif
    }
#ifdef MVBA_DENSE_TRACE
    if (lane == 0 && blockIdx.x < 1024) {
      long long *tr = g_dense_trace + ((size_t)blockIdx.x * 16 + wave) * 4;
      const long long tot = DT_NOW() - dt_t0;
      tr[0] = tot - dt_wait; tr[1] = dt_wait; tr[2] = tot; tr[3] = dt_n;
    }
#endif
    return;
  }
  // ---------------- consumer: tile pairs NC q + wave of the upper triangle (row-major), cameras NC q + wave
  mvba_d4 acc[NPW], cacc[NCW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) acc[q] = mvba_d4{0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < NCW; ++q) cacc[q] = mvba_d4{0, 0, 0, 0};
#ifdef MVBA_DENSE_TRACE
  long long dt_wait = 0, dt_n = 0;
  const long long dt_t0 = DT_NOW();
#endif
  DT_BARRIER();  // chunk 0 is built
  int b = 0;
  for (long long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x, b ^= 1) {
    const double *bG = sG + (size_t)b * 3 * CH * W, *bB = sB + (size_t)b * CH * m * 32, *bS = sSgn + (size_t)b * 3 * CH;
#if !(MVBA_DENSE_KO & 1)
    switch (wave) {  // (uniform)
      case 0: dense_main_mfma<T, 0, CH>(bG, bS, li, lk, acc); break;
      case 1: dense_main_mfma<T, 1, CH>(bG, bS, li, lk, acc); break;
      case 2: dense_main_mfma<T, 2, CH>(bG, bS, li, lk, acc); break;
      case 3: dense_main_mfma<T, 3, CH>(bG, bS, li, lk, acc); break;
      case 4: dense_main_mfma<T, 4 % NC, CH>(bG, bS, li, lk, acc); break;  // (cases 4..7 exist with eight consumers only)
      case 5: dense_main_mfma<T, 5 % NC, CH>(bG, bS, li, lk, acc); break;
      case 6: dense_main_mfma<T, 6 % NC, CH>(bG, bS, li, lk, acc); break;
      default: dense_main_mfma<T, 7 % NC, CH>(bG, bS, li, lk, acc); break;
    }
#endif
#if !(MVBA_DENSE_KO & 8)
#pragma unroll
    for (int g = 0; g < CH / 2; ++g) {  // the per-camera tiles: rows (point 2 g, x), (2 g, y), (2 g + 1, x), (2 g + 1, y)
      const int pa = 2 * g + (lk >> 1), d = lk & 1;
#pragma unroll
      for (int q = 0; q < NCW; ++q) {
        const int k = min(NC * q + wave, m - 1);
        const double v = bB[(size_t)(pa * m + k) * 32 + 16 * d + li];
        cacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, cacc[q], 0, 0, 0);
      }
    }
#endif
    DT_BARRIER();  // this chunk's buffers may be rebuilt, the next chunk's are complete
#ifdef MVBA_DENSE_TRACE
    ++dt_n;
#endif
  }
#ifdef MVBA_DENSE_TRACE
  if (lane == 0 && blockIdx.x < 1024) {
    long long *tr = g_dense_trace + ((size_t)blockIdx.x * 16 + wave) * 4;
    const long long tot = DT_NOW() - dt_t0;
    tr[0] = tot - dt_wait; tr[1] = dt_wait; tr[2] = tot; tr[3] = dt_n;
  }
#endif
  double *out = part + (size_t)blockIdx.x * (P + m) * 256;
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    const int p = NC * q + wave;
    if (p < P)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)p * 256 + r * 64 + lane] = acc[q][r];
  }
#pragma unroll
  for (int q = 0; q < NCW; ++q) {
    const int k = NC * q + wave;
    if (k < m)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(P + k) * 256 + r * 64 + lane] = cacc[q][r];
  }
}

// One wave per element of the packed strips [A | b]: the workgroups' partial tiles summed in workgroup order (lane l takes the
// workgroups l, l + 64, ... in order, then a fixed tree).
__global__ __launch_bounds__(256) void k_schur_dense_finish(int m, int T, int blocks, const double *__restrict__ part, double c,
                                                            double *__restrict__ Afull, double *__restrict__ bfull) {
  const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, P = T * (T + 1) / 2;
  const long long nA = (long long)strip_offset(m, m);
  if (e >= nA + 9 * m) return;
  const size_t bstride = (size_t)(P + m) * 256;
  auto total = [&](int tile, int elem) {
    double t = 0.0;
    for (int b = lane; b < blocks; b += 64) t += part[(size_t)b * bstride + (size_t)tile * 256 + elem];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    return t;  // (valid in lane 0)
  };
  if (e < nA) {
    int k = 0;
    while (k + 1 < m && (long long)strip_offset(k + 1, m) <= e) ++k;
    const int Wk = 9 * (m - k), rem = (int)(e - (long long)strip_offset(k, m)), i = rem / Wk, cr = rem - i * Wk, l = k + cr / 9, j = cr - 9 * (l - k);
    const int gi = 9 * k + i, gj = 9 * l + j, lo = min(gi, gj), hi = max(gi, gj), ti = lo >> 4, tj = hi >> 4;
    double v = -4.0 * total(ti * T - ti * (ti - 1) / 2 + (tj - ti), dense_tile_elem(lo & 15, hi & 15));
    if (k == l) {
      const double hk = total(P + k, dense_tile_elem(i, j));
      v += 2.0 * hk * (i == j ? 1.0 + c : 1.0);  // G_k and its Marquardt term c diag(G_k)   (ref :123-125)
    }
    if (lane == 0) Afull[e] = v;
  } else {
    const int q = (int)(e - nA), k = q / 9, j = q - 9 * k;
    const double v = 2.0 * total(P + k, dense_tile_elem(j, 9));  // 2 Jc_k^T (Jx_k E^-1 dP - e)
    if (lane == 0) bfull[q] = v;
  }
}

// ------------------------------------------------------------------ K4: gauge strip + Cholesky
// The reduced system is SPD (Gauss-Newton Schur complement with Marquardt damping), so
// the reference's np.linalg.solve (LU, ref :146) is replaced by a blocked Cholesky
// (parity-safe).  Storage: M = (D+1) x ld row-major, lower triangle of A in
// rows 0..D-1 and the right-hand side b as ROW D, so that factorising carries the forward
// substitution along (row D ends up holding y = L^-1 b).  Two-level blocking: columns are
// processed in super-blocks of SBW = 128 (four 32-column panels).
//   k_chol_super  ONE launch per super-block, workgroup per 64 rows below it; every workgroup
//                 factors the 128x128 diagonal block in LDS (wave 0: tile factorisations in
//                 registers, identity rows alongside give L^-T) and solves its own rows with
//                 f64 MFMA (left-looking inside the super-block)
//   k_chol_trail64 / k_chol_trail32  once per super-block: C -= P P^T with K = 128 on v_mfma_f64_16x16x4_f64 for
//                 everything right of the super-block (64 x 64 tiles through LDS while there are >= 200 of them)
// then k_chol_backsolve_all does L^T x = y, last super-block first, from the L^-T tiles, in one persistent launch whose
// workgroups hand y to each other behind progress words.
constexpr int NB = 32;
constexpr int SBW = 4 * NB;

// One workgroup per 32 x 32 tile of the lower triangle (+ one per 256 columns of the right-hand-side row).  M[i][j], j <= i, is the
// packed upper element (row gj, column gi): read along gi -- the packed rows are contiguous -- and written along j, through a
// transposing LDS tile.  (Round 1-3 read it along gj, one line per element: 65 us at D = 4493.)
__global__ __launch_bounds__(256) void k_compact(int D, int ld, int m, int gauge_axis, int nt, const double *__restrict__ Afull,
                                                 const double *__restrict__ bfull, double *__restrict__ M, unsigned *__restrict__ bar, int nsync) {
  __shared__ double tile[NB][NB + 1];
  const int ntri = nt * (nt + 1) / 2, t = blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (t >= ntri) {  // the right-hand side (row D) and the back-substitution's progress words
    const int j = (t - ntri) * 256 + threadIdx.x;
    if (j < nsync) bar[j] = 0u;
    if (j < D) M[(size_t)D * ld + j] = bfull[keep_index(j, gauge_axis)];
    return;
  }
  int I = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while (I * (I + 1) / 2 > t) --I;
  while ((I + 1) * (I + 2) / 2 <= t) ++I;
  const int J = t - I * (I + 1) / 2;
  const int i_in = NB * I + tx;  // this thread reads column gi(i_in) of the packed rows gj(j), j = 32 J + ty + 8 q
  const int gi = keep_index(min(i_in, D - 1), gauge_axis);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = NB * J + ty + 8 * q, gj = keep_index(min(j, D - 1), gauge_axis);
    const int r = min(gi, gj), c = max(gi, gj), k = r / 9;  // (on a diagonal tile the upper elements read their mirror image; never stored)
    tile[ty + 8 * q][tx] = Afull[strip_offset(k, m) + (size_t)(r - 9 * k) * (9 * (m - k)) + (c - 9 * k)];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = NB * I + ty + 8 * q, j = NB * J + tx;
    if (i < D && j <= i) M[(size_t)i * ld + j] = tile[tx][ty + 8 * q];
  }
}

__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

typedef double mvba_d4 __attribute__((ext_vector_type(4)));

// MFMA operands straight from row-major rows: the product sums over k, so any permutation of k
// that A and B share is allowed.  Lane (idx = l & 15, kq = l >> 4) loads the 4 CONSECUTIVE
// doubles X[idx][16 g + 4 kq .. + 3] (one 32-byte load; a row's 16-column group is one full
// 128-byte line across kq) and feeds element u to MFMA step 4 g + u (k_chol_trail32).

// One launch per 128-column super-block [jS, jE).  Workgroup = 5 waves: wave 0 runs the serial
// chain (the four 32x32 tile factorisations), waves 1..4 ("workers", 16 rows each) own 64 rows
// below the super-block and do all the MFMA work.  EVERY workgroup factors the whole 128x128
// diagonal block in LDS (redundant, but there is no inter-workgroup dependency inside the
// super-block, so its four panels need one dispatch instead of four).  Per 32-column panel q:
//   F  wave 0: lanes 0..31 keep row r of tile (q,q) in registers, lanes 32..63 row r of the
//      IDENTITY: the column operations of the factorisation (entries of L broadcast with
//      v_readlane: no LDS, every index static) turn the identity rows into L^-T (Zt, double-buffered)
//      workers, meanwhile: the updates of panel q-1 that the chain does not wait for
//      (T[r][c] -= X[r][q-1] X[c][q-1]^T for every tile but (q,q)), their own rows of panel q-1
//      (X = P L^-T -> global) and the left-looking update of their own rows for panel q
//      (P_q -= sum_{q'<q} X_q' L[q][q']^T, A operands kept in registers)
//   T  workers: in-block tiles (r,q), r > q: X = T L_qq^-T (in place)
//   U  workers: tile (q+1,q+1) -= X[q+1][q] X[q+1][q]^T  -- the only update on the critical path
// Workgroup 0 also writes the in-block X tiles (Lblk: this super-block's 128x128 row-major block)
// and the L^-T tiles (Ztiles) for the back-substitution to their OWN buffers: the diagonal block
// of M is never written, because other workgroups may still be loading it.
// MFMA layouts: A/B lane l holds X[idx = l & 15][k = 16 g + 4 (l >> 4) + u] at step (g, u)
// (the k-permutation above); C/D: col = l & 15, row = (l >> 4) + 4 reg.
constexpr int TS = NB + 1;                               // padded LDS tile row stride
constexpr int SUPER_THREADS = 384;  // waves 0..5: chain, workers 0..2, an idle wave (keeps the chain alone on its SIMD), worker 3
constexpr int SUPER_LDS = (12 * NB * TS + 64 * TS + 64 * 9) * 8;  // 10 tiles + 2 Zt + Pt + panel buffer, bytes
constexpr int BACKSOLVE_LDS = (11 * NB * TS + 5 * SBW) * 8;  // k_chol_backsolve_all: 10 tiles + a scratch tile + y + partial sums + x, bytes
__device__ __forceinline__ int tix(int r, int c) { return r * (r + 1) / 2 + c; }

// Tile factorisation on one wave (see F above); returns false if a pivot is not positive.
// The augmented 64x32 matrix B = [tile; I] (column operations turn it into [L; L^-T]) lives in
// MFMA C/D layout (acc[row tile][column tile]).  Per panel of 8 columns:
//   a. the panel's columns go through LDS (Xb) into a row-per-lane register block bp[8]
//   b. 8 elimination steps restricted to the panel: 28 v_readlane broadcasts instead of ~200
//   c. bp back to Xb (and the L^-T rows to Zt)
//   d. every later column at once: acc -= B[:, panel] L[cols, panel]^T as f64 MFMAs whose A and
//      B operands are both read from Xb (B operand of column tile ct = A operand of row tile ct)
// 112 broadcast-FMAs + 32 MFMAs instead of 496 broadcast-FMAs: ~21k -> ~10k cycles per tile.
constexpr int XBS = 9;  // padded row stride of the 64 x 8 panel buffer
#ifndef FT_STAMP
#define FT_STAMP(i)
#endif
__device__ __forceinline__ bool factor_tile(const double (*tile)[TS], double (*Zt)[TS], double *Xb, int lane, bool store,
                                            double *__restrict__ Ztile, int nvalid) {
  const int li = lane & 15, lk = lane >> 4;
  mvba_d4 acc[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * (rt & 1) + lk + 4 * q, col = 16 * ct + li;
        acc[rt][ct][q] = (rt < 2) ? tile[row][col] : ((row == col) ? 1.0 : 0.0);
      }
  bool bad = false;
  FT_STAMP(0);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    // a. panel columns 8p .. 8p+7: C/D layout -> one row per lane
    if ((li >> 3) == (p & 1)) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) Xb[(16 * rt + lk + 4 * q) * XBS + (li & 7)] = acc[rt][p >> 1][q];
    }
    wave_sync();
    double bp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bp[j] = Xb[lane * XBS + j];
    FT_STAMP(1 + 4 * p);
    // b. elimination inside the panel
#ifndef MVBA_FACTOR_SINGLE  // (define it for the one-pivot-per-step elimination: tools/microbench/factor_tile_test.hip times both)
    // Two pivots per step: with a = B[k][k], b = B[k+1][k], c = B[k+1][k+1] the two reciprocal roots
    // 1/l11 = rsq(a) and 1/l22 = rsq(a c - b^2) * l11 do not depend on each other, so the serial chain
    // (rsq + two Newton steps + broadcast) is walked 16 times per tile instead of 32.  a c - b^2
    // loses the same digits as the usual c - b^2 / a (both are det / a up to the factor a).
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const int r0 = 8 * p + k, r1 = r0 + 1;
      const double a = readlane_d(bp[k], r0), b = readlane_d(bp[k], r1), c = readlane_d(bp[k + 1], r1);
      const double det = fma(a, c, -b * b);
      bad |= !(a > 0.0) | !(det > 0.0);
      double ra = __builtin_amdgcn_rsq(a), rd = __builtin_amdgcn_rsq(det);
      ra = ra * (1.5 - 0.5 * a * ra * ra);
      rd = rd * (1.5 - 0.5 * det * rd * rd);
      ra = ra * (1.5 - 0.5 * a * ra * ra);
      rd = rd * (1.5 - 0.5 * det * rd * rd);
      const double l21 = b * ra, inv22 = rd * (a * ra);
      bp[k] = bp[k] * ra;
      bp[k + 1] = (bp[k + 1] - bp[k] * l21) * inv22;
#pragma unroll
      for (int j = k + 2; j < 8; ++j)
        bp[j] -= bp[k] * readlane_d(bp[k], 8 * p + j) + bp[k + 1] * readlane_d(bp[k + 1], 8 * p + j);
    }
#else
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double piv = readlane_d(bp[k], 8 * p + k);
      bad |= !(piv > 0.0);
      // 1/sqrt(piv): v_rsq_f64 seed + two Newton steps (full double precision) instead of the
      // ~40-instruction sqrt and divide expansions, which sit on the serial path 32 times per tile
      double y = __builtin_amdgcn_rsq(piv);
      y = y * (1.5 - 0.5 * piv * y * y);
      y = y * (1.5 - 0.5 * piv * y * y);
      bp[k] = (lane == 8 * p + k) ? piv * y : bp[k] * y;
      // entries above the diagonal (column > row) hold values that are never read: no predicate needed
#pragma unroll
      for (int j = k + 1; j < 8; ++j) bp[j] -= bp[k] * readlane_d(bp[k], 8 * p + j);
    }
#endif
    FT_STAMP(2 + 4 * p);
    // c. finished columns back to Xb; rows 32..63 are rows of L^-T
#pragma unroll
    for (int j = 0; j < 8; ++j) Xb[lane * XBS + j] = bp[j];
    if (lane >= 32) {
      const int r = lane - 32;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 8 * p + j;
        Zt[r][c] = bp[j];
        if (store && r < nvalid && c >= r && c < nvalid) Ztile[r * NB + c] = bp[j];
      }
    }
    // d. all later columns: acc[rt][ct] -= B[16 rt .., panel] L[16 ct .., panel]^T
    wave_sync();
    FT_STAMP(3 + 4 * p);
    if (p < 3) {
      double xa[4][2];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t) xa[rt][t] = Xb[(16 * rt + li) * XBS + 4 * t + lk];
#pragma unroll
      for (int ct = (p == 0) ? 0 : 1; ct < 2; ++ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[rt][t], xa[ct][t], acc[rt][ct], 0, 0, 0);
      wave_sync();  // the next panel overwrites Xb
    }
    FT_STAMP(4 + 4 * p);
  }
  return !bad;
}

__device__ __forceinline__ void chol_super_body(double *lds, double *M, int ld, int D, int jS, double *__restrict__ Ztiles,
                                                double *__restrict__ Lblk, int *__restrict__ flag, int bid) {
  double (*T)[NB][TS] = reinterpret_cast<double (*)[NB][TS]>(lds);
  double (*Zt)[NB][TS] = reinterpret_cast<double (*)[NB][TS]>(lds + 10 * NB * TS);
  double (*Pt)[TS] = reinterpret_cast<double (*)[TS]>(lds + 12 * NB * TS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ww = (wave == 0 || wave == 4) ? -1 : (wave == 5 ? 3 : wave - 1);  // worker index (chain and idle waves: -1)
  const int li = lane & 15, lk = lane >> 4;
  const int nbS = min(SBW, D - jS), jE = jS + nbS, nq = (nbS + NB - 1) / NB;
  const int R0 = jE + bid * 64 + 16 * ww;  // this worker's first row below the super-block
  const bool wg0 = bid == 0;
  const mvba_d4 zero4 = {0.0, 0.0, 0.0, 0.0};
  // Loads, in the order they are needed (one CU draws ~25 GB/s from beyond its L2, so the 144 KiB a
  // workgroup needs take ~8 us: the chain must not wait for all of it):
  //   1. diagonal tile (0,0), by every thread -> LDS -> barrier: wave 0 starts factoring it
  //   2. the workers' own rows of all four panels (C/D layout; consumed panel by panel) and the other
  //      nine tiles of the diagonal block, by waves 1..5: in LDS before barrier B1 of panel 0
  // (lower block triangle; identity padding beyond nbS).  The loads are UNCONDITIONAL on clamped
  // addresses: behind a lane-dependent branch the compiler cannot count how many loads are in flight
  // and waits for all of them (vmcnt(0)) at the first use.
  // (the selection is done on the bit pattern: written as "cond ? loaded : other" the compiler sinks
  // the load back under the branch)
  auto pick = [](bool c, double loaded, double other) -> double {
    const long long m = c ? -1LL : 0LL;
    return __longlong_as_double((__double_as_longlong(loaded) & m) | (__double_as_longlong(other) & ~m));
  };
  // tile_raw requests an element, tile_fix (at the point of use, so that no wait is scheduled
  // earlier) replaces what lies outside the lower triangle / the matrix by the identity padding
  auto tile_raw = [&](int t, int idx) -> double {
    const int r = (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0, c = t - r * (r + 1) / 2;
    const int i = idx >> 5, j = idx & 31, gi = NB * r + i, gj = NB * c + j, gic = min(gi, nbS - 1);
    return M[(size_t)(jS + gic) * ld + jS + min(gj, gic)];
  };
  auto tile_fix = [&](int t, int idx, double raw) -> double {
    const int r = (t >= 6) ? 3 : (t >= 3) ? 2 : (t >= 1) ? 1 : 0, c = t - r * (r + 1) / 2;
    const int i = idx >> 5, j = idx & 31, gi = NB * r + i, gj = NB * c + j;
    return pick(idx < NB * NB && gi < nbS && gj <= gi, raw, (gi == gj) ? 1.0 : 0.0);
  };
  constexpr int P0 = (NB * NB + SUPER_THREADS - 1) / SUPER_THREADS;
  constexpr int RT = SUPER_THREADS - 64, PR = (NB * NB + RT - 1) / RT;  // the rest: every wave but the chain
  double v0[P0];
#pragma unroll
  for (int ps = 0; ps < P0; ++ps) v0[ps] = tile_raw(0, tid + ps * SUPER_THREADS);
  auto store_tile0 = [&]() {
#pragma unroll
    for (int ps = 0; ps < P0; ++ps) {
      const int idx = tid + ps * SUPER_THREADS;
      if (idx < NB * NB) T[0][idx >> 5][idx & 31] = tile_fix(0, idx, v0[ps]);
    }
    __syncthreads();  // tile (0,0) loaded
  };
  if (wave == 0) {
    // ---- the chain: nothing else to load (its own code path, so that its wait counts only its loads)
    store_tile0();
    for (int q = 0; q < nq; ++q) {
      // ---- F
      const bool ok = factor_tile(T[tix(q, q)], Zt[q & 1], lds + 12 * NB * TS + 64 * TS, lane, wg0, Ztiles + (size_t)q * NB * NB, nbS - NB * q);
      if (!ok && wg0 && lane == 0) atomicOr(flag, 2);  // not positive definite
      __syncthreads();  // B1: Zt[q & 1] ready; tiles (r,q), r > q, final
      __syncthreads();  // B2: X tiles of panel q complete
      __syncthreads();  // B3: tile (q+1,q+1) final
    }
    return;  // (of this inlined body: the caller's code after it still runs)
  }
  // ---- workers (and the idle wave): tile (0,0) first, so that the chain starts after ONE short round
  // trip; their own loads follow in one batch and have the ~7 us of the first tile factorisation to land
  store_tile0();
  double P[4][4][2];
  double vr[9][PR];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {  // (the idle wave loads a clamped dummy row group like a worker)
      const double *src = M + (size_t)min(max(R0 + lk + 4 * qq, 0), D) * ld + jS;
      P[q][qq][0] = src[min(NB * q + li, nbS - 1)];
      P[q][qq][1] = src[min(NB * q + 16 + li, nbS - 1)];
    }
#pragma unroll
  for (int t = 1; t < 10; ++t)
#pragma unroll
    for (int ps = 0; ps < PR; ++ps) vr[t - 1][ps] = tile_raw(t, tid - 64 + ps * RT);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int row = R0 + lk + 4 * qq;
      P[q][qq][0] = pick(ww >= 0 && row <= D && NB * q + li < nbS, P[q][qq][0], 0.0);
      P[q][qq][1] = pick(ww >= 0 && row <= D && NB * q + 16 + li < nbS, P[q][qq][1], 0.0);
    }
#pragma unroll
  for (int t = 1; t < 10; ++t)
#pragma unroll
    for (int ps = 0; ps < PR; ++ps) {
      const int idx = tid - 64 + ps * RT;
      if (idx < NB * NB) T[t][idx >> 5][idx & 31] = tile_fix(t, idx, vr[t - 1][ps]);
    }
  mvba_d4 XA[3][2];  // own rows' X of the earlier panels, A layout
#pragma unroll
  for (int q = 0; q < 3; ++q) XA[q][0] = XA[q][1] = zero4;

  // own rows of panel q: X = P Zt (Zt[k][c] = 0 for k > c: the left column tile needs only g = 0),
  // store, and keep X in A layout (through this worker's slice of Pt) for the later panels
  auto own_trsm = [&](int q, mvba_d4 &xa0, mvba_d4 &xa1) {
    const double (*Z)[TS] = Zt[q & 1];
    mvba_d4 x0 = zero4, x1 = zero4;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = 16 * g + 4 * lk + u;
        const double av = Pt[16 * ww + li][k];
        if (g == 0) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Z[k][li], x0, 0, 0, 0);
        x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Z[k][16 + li], x1, 0, 0, 0);
      }
    wave_sync();  // every lane has read its operands out of this slice of Pt
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int row = R0 + lk + 4 * qq;
      if (row <= D) {
        double *dst = M + (size_t)row * ld + jS + NB * q;
        if (NB * q + li < nbS) dst[li] = x0[qq];
        if (NB * q + 16 + li < nbS) dst[16 + li] = x1[qq];
      }
      Pt[16 * ww + lk + 4 * qq][li] = x0[qq];
      Pt[16 * ww + lk + 4 * qq][16 + li] = x1[qq];
    }
    wave_sync();  // C/D layout written by some lanes, A layout read by others
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xa0[u] = Pt[16 * ww + li][4 * lk + u];
      xa1[u] = Pt[16 * ww + li][16 + 4 * lk + u];
    }
  };
  // 16x16 sub-tile (ih, jh) of T[r][c] -= X[r][q] X[c][q]^T
  auto update_sub = [&](int r, int c, int q, int ih, int jh) {
    mvba_d4 acc = zero4;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = 16 * g + 4 * lk + u;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[tix(r, q)][16 * ih + li][k], T[tix(c, q)][16 * jh + li][k], acc, 0, 0, 0);
      }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) T[tix(r, c)][16 * ih + lk + 4 * qq][16 * jh + li] -= acc[qq];
  };

  // Role split by whole waves (every role executes the same number of workgroup barriers per
  // panel: F | B1 | T | B2 | U | B3).  Separate code paths keep the chain's registers (a tile row)
  // and the workers' registers (own rows, A operands) out of each other's live ranges.
  if (ww < 0) {  // the idle wave (the chain returned above)
    for (int q = 0; q < nq; ++q) {
      __syncthreads();
      __syncthreads();
      __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= nq) break;  // uniform
    // ---- F (while wave 0 factors tile (q,q))
    if (q > 0) {
      int sidx = 0;  // panel q-1's updates of every tile but (q,q)
      for (int r = q; r < nq; ++r)
        for (int c = q; c <= r; ++c) {
          if (r == q && c == q) continue;
          for (int sub = 0; sub < 4; ++sub) {
            const int ih = sub >> 1, jh = sub & 1;
            if (r == c && jh > ih) continue;  // upper sub-tile of a diagonal tile: never read
            if ((sidx++ & 3) == ww) update_sub(r, c, q - 1, ih, jh);
          }
        }
      own_trsm(q - 1, XA[q - 1][0], XA[q - 1][1]);
    }
    {  // own rows: left-looking update of panel q, result to this worker's slice of Pt
      mvba_d4 acc0 = zero4, acc1 = zero4;
#pragma unroll
      for (int qp = 0; qp < q; ++qp)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = 16 * g + 4 * lk + u;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(XA[qp][g][u], T[tix(q, qp)][li][k], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(XA[qp][g][u], T[tix(q, qp)][16 + li][k], acc1, 0, 0, 0);
          }
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        Pt[16 * ww + lk + 4 * qq][li] = P[q][qq][0] - acc0[qq];
        Pt[16 * ww + lk + 4 * qq][16 + li] = P[q][qq][1] - acc1[qq];
      }
    }
    __syncthreads();  // B1
    // ---- T: in-block tiles (r,q): X = T Zt in place, one 16-row unit per worker at a time
    {
      const double (*Z)[TS] = Zt[q & 1];
      for (int e = ww; e < 2 * (nq - 1 - q); e += 4) {
        const int r = q + 1 + (e >> 1), h = e & 1;
        double (*tile)[TS] = T[tix(r, q)];
        double av[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int u = 0; u < 4; ++u) av[g][u] = tile[16 * h + li][16 * g + 4 * lk + u];
        mvba_d4 x0 = zero4, x1 = zero4;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = 16 * g + 4 * lk + u;
            if (g == 0) x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][u], Z[k][li], x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g][u], Z[k][16 + li], x1, 0, 0, 0);
          }
        wave_sync();  // in place: every lane has read its operands before any lane overwrites the unit
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int i = 16 * h + lk + 4 * qq;
          tile[i][li] = x0[qq];
          tile[i][16 + li] = x1[qq];
          if (wg0 && NB * r + i < nbS) {  // columns of panel q are all < nbS here (q < r)
            double *dst = Lblk + (size_t)(NB * r + i) * SBW + NB * q;
            dst[li] = x0[qq];
            dst[16 + li] = x1[qq];
          }
        }
      }
    }
    __syncthreads();  // B2
    // ---- U: the next diagonal tile, three 16x16 sub-tiles on three workers
    if (q + 1 < nq && ww < 3) update_sub(q + 1, q + 1, q, ww == 0 ? 0 : 1, ww == 2 ? 1 : 0);
    __syncthreads();  // B3
  }
  {  // own rows of the last panel
    mvba_d4 d0, d1;
    own_trsm(nq - 1, d0, d1);
  }
}

__global__ __launch_bounds__(SUPER_THREADS) void k_chol_super(double *M, int ld, int D, int jS, double *__restrict__ Ztiles,
                                                              double *__restrict__ Lblk, int *__restrict__ flag) {
  extern __shared__ double lds[];
  chol_super_body(lds, M, ld, D, jS, Ztiles, Lblk, flag, blockIdx.x);
}

// Trailing update with f64 MFMA for the finished super-block [jS, jE): C -= P P^T on rows/cols
// >= jE (jE - jS == SBW).  One workgroup per 32 x 32 tile of the lower triangle, the K = 128 columns
// split over its four waves (32 each) and the four partial tiles summed through LDS.  Round 1 gave a
// workgroup a 64 x 64 tile with the whole K per wave: 160 KiB of operands per workgroup instead of
// 72 KiB, and one CU draws only ~25 GB/s from beyond its L2 -- 20 us per update at D = 893 where
// this takes 6, and still 0.6 ms slower per solve at D = 4493 (four times fewer workgroups to
// spread the loads over).
__global__ __launch_bounds__(256) void k_chol_trail32(double *M, int ld, int D, int jS, int jE) {
  __shared__ double part[4][NB][NB + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  // lower-triangle tile index -> (by, bx <= by)
  const int t = blockIdx.x;
  int by = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while (by * (by + 1) / 2 > t) --by;
  while ((by + 1) * (by + 2) / 2 <= t) ++by;
  const int bx = t - by * (by + 1) / 2;
  const int r0 = jE + NB * by, c0 = jE + NB * bx;
  // C (this thread's four elements), requested first
  double cv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = threadIdx.x + 256 * q, rr = r0 + (e >> 5), cc = c0 + (e & 31);
    cv[q] = M[(size_t)min(rr, D) * ld + min(cc, D - 1)];
  }
  mvba_d4 av[2][2], bv[2][2];  // [row group of 16][k group of 16]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      av[i][g] = *reinterpret_cast<const mvba_d4 *>(M + (size_t)min(r0 + 16 * i + li, D) * ld + jS + 32 * wave + 16 * g + 4 * lk);
      bv[i][g] = *reinterpret_cast<const mvba_d4 *>(M + (size_t)min(c0 + 16 * i + li, D) * ld + jS + 32 * wave + 16 * g + 4 * lk);
    }
  mvba_d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = mvba_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i][g][u], bv[j][g][u], acc[i][j], 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[wave][16 * i + lk + 4 * q][16 * j + li] = acc[i][j][q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = threadIdx.x + 256 * q, i = e >> 5, j = e & 31, rr = r0 + i, cc = c0 + j;
    const double sum = (part[0][i][j] + part[1][i][j]) + (part[2][i][j] + part[3][i][j]);
    if (rr <= D && cc < D && cc <= rr) M[(size_t)rr * ld + cc] = cv[q] - sum;
  }
}

// The same update for LARGE trailing matrices (round 4): k_chol_trail32 reads 64 KiB of operands per 32 x 32 tile straight
// into registers -- every tile of a tile row re-reads that row's 32 x 128 panel from L2 -- and its waves live ~8 us for
// 0.4 us of MFMA (profiles/r04_m_pmc_summary_config4_shard.txt: 82 % L2 hits, 53 % of the wave cycles waiting): 24-29 TFLOP/s,
// 1.1-1.2 of the 2.8 ms of a D = 4493 solve.  Here a workgroup owns 64 x 64 of C (four waves, 32 x 32 each), the K = 128
// columns stream through LDS in chunks of 16 (A and B chunk 8 KiB each, double-buffered, the next chunk's global loads in
// flight during the MFMAs of this one): 32 KiB of operands per 32 x 32 of C instead of 64, one batch of C loads that lands
// during the loop, four workgroups per CU in different phases.  First update at D = 4493: 82.8 -> 60.1 us (40 TFLOP/s);
// a 128 x 128 tile (64 x 64 per wave, 234 VGPRs, two workgroups per CU) has twice the reuse and is SLOWER, 91 us: a lone
// workgroup needs 27 us (8 us for its 64-value-per-lane read-modify-write of C alone) and two per CU run in lockstep.
// Diagonal workgroups compute the full square and store the lower triangle (their upper-right wave only keeps the barriers
// company).  Below ~200 workgroups k_chol_trail32 wins: a workgroup here lives >= 8 us, there 2-5
// (profiles/r04_trail_block_trace.txt).
constexpr int TB = 64, TB_KC = 16, TB_LD = TB_KC + 2;                  // tile, chunk of K, padded LDS row (144 bytes: 16-byte aligned)
constexpr int TRAIL64_LDS = 2 * 2 * TB * TB_LD * (int)sizeof(double);  // 2 buffers x (A, B): 36,864 bytes
__global__ __launch_bounds__(256, 4) void k_chol_trail64(double *M, int ld, int D, int jS, int jE) {
  extern __shared__ double lds[];
  double (*As)[TB][TB_LD] = reinterpret_cast<double (*)[TB][TB_LD]>(lds);
  double (*Bs)[TB][TB_LD] = reinterpret_cast<double (*)[TB][TB_LD]>(lds + 2 * TB * TB_LD);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4, wy = wave >> 1, wx = wave & 1;
  const int t = blockIdx.x;
  int by = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while (by * (by + 1) / 2 > t) --by;
  while ((by + 1) * (by + 2) / 2 <= t) ++by;
  const int bx = t - by * (by + 1) / 2;
  const int r0 = jE + TB * by, c0 = jE + TB * bx;
  const bool idle = (bx == by) && wy == 0 && wx == 1;  // entirely above the diagonal
  // this thread's share of a chunk: 4 consecutive doubles of one row of A and of B (four threads per 128-byte line)
  const int prow = tid >> 2, pcol = 4 * (tid & 3);
  const double *asrc = M + (size_t)min(r0 + prow, D) * ld + jS + pcol;
  const double *bsrc = M + (size_t)min(c0 + prow, D) * ld + jS + pcol;
  mvba_d4 pa = *reinterpret_cast<const mvba_d4 *>(asrc), pb = *reinterpret_cast<const mvba_d4 *>(bsrc);
  // C (C/D layout: col = li, row = lk + 4 q): requested now, needed after the loop
  double cv[2][2][4];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        cv[rg][cg][q] = M[(size_t)min(r0 + 16 * (2 * wy + rg) + lk + 4 * q, D) * ld + min(c0 + 16 * (2 * wx + cg) + li, D - 1)];
  mvba_d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = mvba_d4{0.0, 0.0, 0.0, 0.0};
  auto stage = [&](int b) {
    *reinterpret_cast<mvba_d4 *>(&As[b][prow][pcol]) = pa;
    *reinterpret_cast<mvba_d4 *>(&Bs[b][prow][pcol]) = pb;
  };
  stage(0);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < SBW / TB_KC; ++g) {
    const int b = g & 1;
    if (g + 1 < SBW / TB_KC) {
      pa = *reinterpret_cast<const mvba_d4 *>(asrc + TB_KC * (g + 1));
      pb = *reinterpret_cast<const mvba_d4 *>(bsrc + TB_KC * (g + 1));
    }
    if (!idle) {
      // lane (li, lk) holds X[16 rg + li][4 lk + u] for MFMA step u: the k-permutation both operands share (see above)
      mvba_d4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = *reinterpret_cast<const mvba_d4 *>(&As[b][16 * (2 * wy + i) + li][4 * lk]);
        bv[i] = *reinterpret_cast<const mvba_d4 *>(&Bs[b][16 * (2 * wx + i) + li][4 * lk]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
          for (int cg = 0; cg < 2; ++cg) acc[rg][cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rg][u], bv[cg][u], acc[rg][cg], 0, 0, 0);
    }
    if (g + 1 < SBW / TB_KC) stage(b ^ 1);  // (read by compute(g - 1): every wave is past it since the barrier below)
    __syncthreads();
  }
  if (idle) return;
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = r0 + 16 * (2 * wy + rg) + lk + 4 * q, cc = c0 + 16 * (2 * wx + cg) + li;
        if (rr <= D && cc < D && cc <= rr) M[(size_t)rr * ld + cc] = cv[rg][cg][q] - acc[rg][cg][q];
      }
}

// L^T x = y (y = row D of M, overwritten by x), one launch per 128-column super-block, last one
// first.  Launch for super-block [jS, jE), given the finished x of the super-block above it
// [jE, jE2):
//   workgroups >= 1   y[c] -= sum_r L[r][c] x[r]  (r in [jE, jE2)) for the columns c < jS, one column
//                     per thread, rows read coalesced: the bulk of the memory traffic, chip-wide
//   workgroup 0       the same for its own columns [jS, jE), then the four tiles from the top (in-block
//                     L from Lblk, see k_chol_super):
//                     x_t = (L^-T tile) y_t is a 32x32 mat-vec (tiles from k_chol_super: no serial
//                     substitution) and y -= L[tile rows][cols left of it in the super-block]^T x_t;
//                     scatters x into the full 9m vector (zeros at the gauge slots).
__device__ __forceinline__ void chol_backsolve_body(double *M, int ld, int D, int m, int gauge_axis,
                                                    const double *Ztiles, const double *Lblk, double *dxi_full, int jS, int jE,
                                                    int jE2, int bid, int tid) {
  __shared__ double xp[SBW];  // x of the super-block above
  __shared__ double ys[SBW];  // y, then x, of this super-block
  __shared__ double part[2][SBW];
  __shared__ double T[NB][NB + 1];
  const bool act = tid < 256;  // the body is written for 256 threads; further threads only keep the barriers company
  double *y = M + (size_t)D * ld;
  const int np = jE2 - jE;  // 0 for the first launch (top super-block)
  __syncthreads();          // a previous call's readers of xp / ys are done
  if (tid < SBW) xp[tid] = (tid < np) ? y[jE + tid] : 0.0;
  __syncthreads();
  if (bid > 0) {
    const int c = (bid - 1) * 256 + tid;
    if (act && c < jS) {
      const double *col = M + (size_t)jE * ld + c;
      double s0 = 0.0, s1 = 0.0;
      int r = 0;
      for (; r + 16 <= np; r += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = col[(size_t)(r + u) * ld];
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          s0 += v[u] * xp[r + u];
          s1 += v[u + 1] * xp[r + u + 1];
        }
      }
      for (; r < np; ++r) s0 += col[(size_t)r * ld] * xp[r];
      y[c] -= s0 + s1;
    }
    return;
  }
  const int ns = jE - jS;
  {
    const int c = tid & (SBW - 1), h = tid >> 7;  // two threads per column, alternate rows
    double s = 0.0;
    if (act && c < ns) {
      const double *col = M + (size_t)jE * ld + jS + c;
#pragma unroll 8
      for (int r = h; r < np; r += 2) s += col[(size_t)r * ld] * xp[r];
    }
    if (act) part[h][c] = s;
    __syncthreads();
    if (tid < SBW) ys[tid] = (tid < ns) ? y[jS + tid] - part[0][tid] - part[1][tid] : 0.0;
    if (np == 0 && act)  // first launch on the stream: clear the gauge slots before any x is scattered
      for (int i = tid; i < 9 * m; i += 256) dxi_full[i] = 0.0;
    __syncthreads();
  }
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = (ns + NB - 1) / NB - 1; t >= 0; --t) {
    const int jb = jS + t * NB, nb = min(NB, jE - jb);
    if (act)
      for (int q = tid; q < NB * NB; q += 256) {
        const int r = q / NB, c = q % NB;
        T[r][c] = (r < nb && c >= r && c < nb) ? Ztiles[(size_t)(jb / NB) * NB * NB + q] : 0.0;
      }
    // operands of this tile's update inside the super-block, one column per thread
    double lcol[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) lcol[r] = (tid < t * NB && r < nb) ? Lblk[(size_t)(t * NB + r) * SBW + tid] : 0.0;
    __syncthreads();
    if (wave == 0) {
      const int r = lane & 31, h = lane >> 5;  // two lanes per row, 16 columns each
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int q = 0; q < NB / 2; q += 2) {
        s0 += T[r][16 * h + q] * ys[t * NB + 16 * h + q];
        s1 += T[r][16 * h + q + 1] * ys[t * NB + 16 * h + q + 1];
      }
      double xr = s0 + s1;
      xr += __shfl_xor(xr, 32, 64);
      if (lane < nb) ys[t * NB + lane] = xr;  // all reads of ys above precede this write (one wave, in order)
    }
    __syncthreads();
    if (tid < t * NB) {
      double sacc = 0.0;
#pragma unroll
      for (int r = 0; r < NB; ++r) sacc += lcol[r] * ys[t * NB + r];  // lcol = 0 beyond nb
      ys[tid] -= sacc;
    }
    __syncthreads();
  }
  if (tid < ns) {
    y[jS + tid] = ys[tid];
    dxi_full[keep_index(jS + tid, gauge_axis)] = ys[tid];
  }
}

__global__ __launch_bounds__(256) void k_chol_backsolve(double *M, int ld, int D, int m, int gauge_axis,
                                                        const double *__restrict__ Ztiles, const double *__restrict__ Lblk,
                                                        double *__restrict__ dxi_full, int jS, int jE, int jE2) {
  chol_backsolve_body(M, ld, D, m, gauge_axis, Ztiles, Lblk, dxi_full, jS, jE, jE2, blockIdx.x, threadIdx.x);
}

// Device-wide barrier of a persistent grid (every workgroup co-resident: at most one per CU).  Every
// workgroup reaches every barrier (the counts depend on D only), and a barrier gives up after `max_polls`
// polls (2^22 by default; flag bit 8 -> the host redoes the solve with one launch per super-block) instead of
// spinning for ever, so the grid always drains.
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned target, int *flag, unsigned max_polls) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this wave's global writes are visible device-wide
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > max_polls) {
        atomicOr(flag, 8);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // and the other workgroups' writes are visible to this wave
}

#ifdef MVBA_BS_TRACE  // timing-only build (tools/bs_trace.py): 100 MHz stamps of every chain workgroup of the last launch
__device__ long long g_bs_trace[8 * 256];
#define BS_STAMP(s, i) do { if (threadIdx.x == 0) g_bs_trace[8 * (s) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BS_STAMP(s, i)
#endif
// Point-to-point version of the same (round 4): thread 0 of a workgroup waits until a progress word has reached `target`
// (words only grow).  No device-wide fences: an agent-scope release / acquire pair is a write-back / invalidate of the whole
// L2 of the XCD (1.7 + 1.5 us of an 11.6 us chain step with 140 workgroups doing the same, tools/bs_trace.py), and the only
// data that travels between workgroups here is the vector y.  So every access to y inside this kernel is an agent-scope
// atomic (sc1: stores write through to memory, loads do not hit a stale line), a producer's waves wait for their stores
// (s_waitcnt vmcnt(0)), meet at a barrier, and then thread 0 stores the word; the consumer polls it,
// passes a barrier and loads.  Gives up like grid_barrier does (and at once when somebody else already has), so the grid drains.
template <int SLEEP>
__device__ __forceinline__ void flow_wait(const unsigned *word, unsigned target, int *flag, unsigned max_polls) {
  unsigned spins = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(SLEEP);
    ++spins;
    if (spins > max_polls || ((spins & 127u) == 0u && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 8))) {
      atomicOr(flag, 8);
      break;
    }
  }
}
__device__ __forceinline__ void flow_post(unsigned *word, unsigned value) {
  // this wave's sc1 stores have COMPLETED before anybody stores the word: a workgroup-scope release fence is not enough (in
  // this execution mode the compiler lowers it to nothing -- waves of a workgroup share their L1 -- and the word would chase
  // the data through different L2 channels), hence the explicit wait
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double flow_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flow_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- L^T x = y for all super-blocks in ONE persistent launch (last block first), S - 1 device-wide
// barriers instead of S launches.  Step s (x_{s+1} known):
//   workgroup s ("chain" of block s)
//       y_s -= (rows of the block above)^T x_{s+1}; then the block's four tiles from the bottom:
//       x_t = Z_t y_t (Z = L_tt^-T from the factorisation), y_{t' < t} -= L[t][t']^T x_t.
//       Everything static it needs (Z tiles and in-block L tiles -> LDS, the 128 x 128 panel below the
//       block -> registers: 208 KiB, ~10 us for one CU) is loaded when the kernel starts, by all S chain
//       workgroups at once; at its step a chain workgroup reads only x_{s+1} and y_s (2 KiB).
//   workgroups >= S ("bulk")
//       y[c] -= (rows of block s+1)^T x_{s+1} for the columns left of block s, 32 columns x 8 row
//       chunks per workgroup: one batch of 16 loads per thread, LDS reduction.
// One barrier per step: x_s must reach the next chain and the bulk workgroups, their updates the chain.
// FLOW (round 4, the default): no device-wide barriers.  sync[0] = number of x blocks published (chain s posts S - s);
// sync[1 + g] = number of x blocks applied to the 32-column group g, whose owner among the bulk workgroups applies them
// last block first.  Chain s waits for x_{s+1} and for x_{s+2} on its own four groups -- applied a whole chain step
// earlier, so it practically never waits for the bulk -- and a bulk workgroup for the x it is about to apply.  With nobody
// else in a barrier the bulk can be one workgroup per group (140 at D = 4493 instead of 16: 11 us of a 14 us step were
// their share of the matrix at ~25 GB/s each).
template <bool FLOW>
__global__ __launch_bounds__(SUPER_THREADS) void k_chol_backsolve_all(double *M, int ld, int D, int m, int gauge_axis,
                                                                      const double *Ztiles, const double *Lblk_all, double *dxi_full,
                                                                      int *flag, unsigned *bar, unsigned max_polls) {
  extern __shared__ double lds[];
  const int G = gridDim.x, bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *y = M + (size_t)D * ld;
  const int S = (D + SBW - 1) / SBW;
  double (*Zs)[NB][TS] = reinterpret_cast<double (*)[NB][TS]>(lds);                // [4]
  double (*Ls)[NB][TS] = reinterpret_cast<double (*)[NB][TS]>(lds + 4 * NB * TS);  // [6]: in-block tile (r, c < r) at r (r - 1) / 2 + c
  double (*Tt)[TS] = reinterpret_cast<double (*)[TS]>(lds + 10 * NB * TS);        // scratch tile of the block inverse
  double *ys = lds + 11 * NB * TS, *part = ys + SBW, *xs = part + 3 * SBW;         // part[3][SBW]; bulk: red[8][32], then x
  unsigned epoch = 0;
  if (bid < S) {
    // ================= chain of block s = bid
    const int s = bid, jS = s * SBW, jE = min(jS + SBW, D), jE2 = min(jE + SBW, D), ns = jE - jS, np = jE2 - jE;
    BS_STAMP(s, 0);
    // ---- static operands, one batch.  Every load is "uniform base + one per-thread offset", so the
    // addresses live in SGPRs and the batch fits the register file.
    const double *Lblk = Lblk_all + (size_t)s * SBW * SBW;
    constexpr int NT = SUPER_THREADS, NPASS = (NB * NB + NT - 1) / NT, NPN = (SBW + 2) / 3;
    double zl[10][NPASS], pn[NPN];
#pragma unroll
    for (int tile = 0; tile < 10; ++tile) {
      // tiles 0..3: Z (upper triangle of L_tt^-T); 4..9: in-block L tile (tr, tc < tr)
      const int e = tile - 4, tr = (e >= 3) ? 3 : (e >= 1) ? 2 : 1, tc = e - tr * (tr - 1) / 2;
      const double *base = (tile < 4) ? Ztiles + (size_t)(jS / NB + tile) * NB * NB : Lblk + (size_t)(NB * tr) * SBW + NB * tc;
      const int nb = ns - NB * (tile < 4 ? tile : tr);  // valid rows (and columns, for Z) of the tile
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int w = tid + NT * ps, r = w >> 5, c = w & 31;
        const bool ok = w < NB * NB && r < nb && (tile >= 4 || (c >= r && c < nb));
        zl[tile][ps] = ok ? base[tile < 4 ? w : r * SBW + c] : 0.0;
      }
    }
    const int pc = tid & (SBW - 1), ph = tid >> 7;  // three threads per column, every third row
    const int poff = ph * ld + pc;
#pragma unroll
    for (int i = 0; i < NPN; ++i) {
      const double *rowbase = M + (size_t)(jE + 3 * i) * ld + jS;  // uniform
      pn[i] = (pc < ns && ph + 3 * i < np) ? rowbase[poff] : 0.0;
    }
#pragma unroll
    for (int tile = 0; tile < 10; ++tile)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int w = tid + NT * ps;
        if (w < NB * NB) Zs[tile][w >> 5][w & 31] = zl[tile][ps];  // Ls follows Zs: tile 4 + e lands in Ls[e]
      }
    // ---- FLOW, blocks with three chain steps of waiting ahead of them: the block's whole L_ss^-T instead of a walk over its
    // four tiles at the step (2.8 of 5.7 us, tools/bs_trace.py).  From x_r = Z_r (y_r - sum_{c > r} L_cr^T x_c):
    //   x_r = sum_{c >= r} W_rc y_c,  W_rr = Z_r,  W_rc = -Z_r sum_{r < k <= c} L_kr^T W_kc,
    // six off-diagonal tiles, each two MFMA passes (the sum into the scratch tile, then -Z_r times it) on four waves, one
    // 16 x 16 quarter each; W_rc lands in the slot of L_cr, which nobody needs any more in this order.
    const bool inverse = FLOW && (S - 1 - s) >= 3;
    if (inverse) {
      __syncthreads();  // the tiles are in LDS
      const int li = lane & 15, lk = lane >> 4, qi = (wave >> 1) & 1, qj = wave & 1;  // waves 0..3: quarter (qi, qj)
      auto lslot = [](int tr, int tc) { return tr * (tr - 1) / 2 + tc; };
#pragma unroll 1
      for (int pair = 0; pair < 6; ++pair) {
        const int r = (pair == 0) ? 2 : (pair == 1 || pair == 3) ? 1 : 0, c = (pair < 3) ? 3 : (pair < 5) ? 2 : 1;
        if (wave < 4) {
          mvba_d4 acc = {0.0, 0.0, 0.0, 0.0};
          for (int k = r + 1; k <= c; ++k) {
            const double (*Lk)[TS] = Ls[lslot(k, r)];                         // L_kr: A = L_kr^T, A[i][kk] = L_kr[kk][i]
            const double (*Wk)[TS] = (k == c) ? Zs[c] : Ls[lslot(c, k)];       // W_kc (already in L_ck's slot), W_cc = Z_c
#pragma unroll
            for (int g = 0; g < 8; ++g) {
              const int kk = 4 * g + lk;
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lk[kk][16 * qi + li], Wk[kk][16 * qj + li], acc, 0, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) Tt[16 * qi + lk + 4 * q][16 * qj + li] = acc[q];
        }
        __syncthreads();
        mvba_d4 w = {0.0, 0.0, 0.0, 0.0};
        if (wave < 4) {
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const int kk = 4 * g + lk;
            w = __builtin_amdgcn_mfma_f64_16x16x4f64(-Zs[r][16 * qi + li][kk], Tt[kk][16 * qj + li], w, 0, 0, 0);
          }
        }
        __syncthreads();  // every wave has read L_cr's slot (only pair (r, c) itself uses it from here on) and the scratch tile
        if (wave < 4) {
          double (*Wo)[TS] = Ls[lslot(c, r)];
#pragma unroll
          for (int q = 0; q < 4; ++q) Wo[16 * qi + lk + 4 * q][16 * qj + li] = w[q];
        }
        __syncthreads();
      }
      for (int e = tid; e < NB * TS; e += NT) (&Tt[0][0])[e] = 0.0;  // the scratch tile now stands in for the tiles left of the diagonal
    }
    // this thread's row of W, tile by tile (x = W y below: row tid & 127)
    const double *wrow[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int r = (tid & (SBW - 1)) >> 5, ii = tid & 31;
      wrow[ct] = (ct < r) ? &Tt[ii][0] : (ct == r) ? &Zs[r][ii][0] : &Ls[ct * (ct - 1) / 2 + r][ii][0];
    }
    // ---- wait for the step
    BS_STAMP(s, 1);
    if (FLOW) {
      if (s < S - 1) {
        if (tid == 0) {
          if (s + 2 <= S - 1)  // (posted a whole chain step ago: checked first, while x_{s+1} is still on its way)
            for (int q = 0; q < 4; ++q) flow_wait<1>(bar + 1 + 4 * s + q, (unsigned)(S - (s + 2)), flag, max_polls);
          flow_wait<0>(bar, (unsigned)(S - 1 - s), flag, max_polls);
        }
        BS_STAMP(s, 2);
        __syncthreads();
        BS_STAMP(s, 3);
      }
    } else
      for (int t = S - 1; t > s; --t) grid_barrier(bar, ++epoch * G, flag, max_polls);
    // x_{s+1} (published by chain s+1 before the barrier) and y_s (complete but for the panel's share)
    const double yv = (tid < ns) ? (FLOW ? flow_load(y + jS + tid) : y[jS + tid]) : 0.0;
    if (FLOW) {  // one sc1 load per element of x_{s+1}, the rest reads LDS (43 sc1 loads per thread: 2.2 us of a step)
      if (tid < SBW) xs[tid] = (tid < np) ? flow_load(y + jE + tid) : 0.0;
      __syncthreads();
    }
    double sa = 0.0, sb = 0.0;
    if (np > 0) {
      double xv[NPN];
#pragma unroll
      for (int i = 0; i < NPN; ++i) xv[i] = (ph + 3 * i < np) ? (FLOW ? xs[ph + 3 * i] : y[jE + ph + 3 * i]) : 0.0;
#pragma unroll
      for (int i = 0; i + 1 < NPN; i += 2) {
        sa += pn[i] * xv[i];
        sb += pn[i + 1] * xv[i + 1];
      }
      if (NPN & 1) sa += pn[NPN - 1] * xv[NPN - 1];
    }
    part[ph * SBW + pc] = sa + sb;
    __syncthreads();
    if (tid < SBW) ys[tid] = (tid < ns) ? yv - part[tid] - part[SBW + tid] - part[2 * SBW + tid] : 0.0;
    __syncthreads();
    BS_STAMP(s, 4);
    if (inverse) {  // x = W y: row i = tid & 127, the columns c with c % 3 == tid / 128
      // (tile bases per thread, the column offsets compile-time constants -- one code path per third: 86 LDS reads with
      // immediate offsets and 43 FMAs; with the addresses computed per element this took as long as the walk it replaces)
      auto row_dot = [&](auto PH) {
        constexpr int ph0 = decltype(PH)::value;
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NPN; ++k) {
          const int c = ph0 + 3 * k;
          if (c < SBW) sacc[k & 3] += wrow[c >> 5][c & 31] * ys[c];
        }
        return (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
      };
      const double dot = ph == 0 ? row_dot(std::integral_constant<int, 0>{}) : ph == 1 ? row_dot(std::integral_constant<int, 1>{}) : row_dot(std::integral_constant<int, 2>{});
      part[ph * SBW + (tid & (SBW - 1))] = dot;
      __syncthreads();
      if (tid < SBW) ys[tid] = part[tid] + part[SBW + tid] + part[2 * SBW + tid];
      __syncthreads();
    } else
    // ---- the tile chain
    for (int t = (ns + NB - 1) / NB - 1; t >= 0; --t) {
      if (wave == 0) {
        const int r = lane & 31, h = lane >> 5;  // two lanes per row, 16 columns each
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < NB / 2; q += 2) {
          s0 += Zs[t][r][16 * h + q] * ys[t * NB + 16 * h + q];
          s1 += Zs[t][r][16 * h + q + 1] * ys[t * NB + 16 * h + q + 1];
        }
        double xr = s0 + s1;
        xr += __shfl_xor(xr, 32, 64);
        if (lane < NB) ys[t * NB + lane] = xr;  // rows beyond the block's end: Zs row = 0 -> 0
      }
      __syncthreads();
      if (tid < t * NB) {
        const double (*Lt)[TS] = Ls[t * (t - 1) / 2 + (tid >> 5)];
        double sacc = 0.0;
#pragma unroll
        for (int r = 0; r < NB; ++r) sacc += Lt[r][tid & 31] * ys[t * NB + r];
        ys[tid] -= sacc;
      }
      __syncthreads();
    }
    if (tid < ns) {
      if (FLOW) flow_store(y + jS + tid, ys[tid]);
      else y[jS + tid] = ys[tid];
      dxi_full[keep_index(jS + tid, gauge_axis)] = ys[tid];
    }
    if (s == 0 && tid < 7) dxi_full[tid < 6 ? 3 + tid : 12 + gauge_axis] = 0.0;  // the removed (gauge) parameters
    BS_STAMP(s, 5);
    if (FLOW) {
      if (s > 0) flow_post(bar, (unsigned)(S - s));
      BS_STAMP(s, 6);
    } else
      for (int t = s; t > 0; --t) grid_barrier(bar, ++epoch * G, flag, max_polls);
    return;
  }
  // ================= bulk: at step s, columns left of block s, rows of block s+1
  const bool act = tid < 256;
  double *red = part;
  const int cj = tid & 31, ch = tid >> 5, nbulk = G - S;
  for (int s = S - 1; s >= 0; --s) {
    const int jS = s * SBW, jE = min(jS + SBW, D), jE2 = min(jE + SBW, D), np = jE2 - jE, ngrp = (jS + 31) / 32;
    if (FLOW) {  // x_{s+1} -> the groups this workgroup owns, the rightmost (the next chain's) first
      if (np <= 0 || bid - S >= ngrp) continue;  // (uniform; owned groups only become fewer as s falls)
      if (tid == 0) flow_wait<8>(bar, (unsigned)(S - 1 - s), flag, max_polls);  // (the bulk has a chain step of slack: polls at leisure)
      __syncthreads();
    }
    if (np > 0)
      for (int g0 = bid - S; g0 < ngrp; g0 += nbulk) {
        const int g = FLOW ? (bid - S) + ((ngrp - 1 - (bid - S)) / nbulk) * nbulk - (g0 - (bid - S)) : g0;
        const int c = 32 * g + cj;
        double lv[16], xv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int r = 16 * ch + u;
          const bool ok = act && c < jS && r < np;
          lv[u] = ok ? M[(size_t)(jE + r) * ld + c] : 0.0;
          xv[u] = ok ? (FLOW ? flow_load(y + jE + r) : y[jE + r]) : 0.0;
        }
        const double yc = (tid < 32 && c < jS) ? (FLOW ? flow_load(y + c) : y[c]) : 0.0;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          s0 += lv[u] * xv[u];
          s1 += lv[u + 1] * xv[u + 1];
        }
        if (act) red[ch * 32 + cj] = s0 + s1;
        __syncthreads();
        if (tid < 32 && c < jS) {
          double tot = 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k) tot += red[k * 32 + cj];
          if (FLOW) flow_store(y + c, yc - tot);
          else y[c] = yc - tot;
        }
        if (FLOW) flow_post(bar + 1 + g, (unsigned)(S - 1 - s));
        __syncthreads();
      }
    if (!FLOW && s > 0) grid_barrier(bar, ++epoch * G, flag, max_polls);
  }
}

// ---- fallback: LU with partial pivoting (what np.linalg.solve / LAPACK gesv does, ref :146) ----
// Only reached when the Cholesky meets a non-positive pivot (reduced system not positive definite,
// e.g. a negative damping factor): rare, but a library path, so it is blocked and runs on the whole
// chip (round 2's single-workgroup kernel took 0.11 s at D = 893 and 13 s at D = 4493).
// Right-looking, panels of LU_NB columns, on the AUGMENTED matrix [F | rhs] (ld = D + 1: the row
// swaps, the triangular solve and the trailing update carry the forward substitution along):
//   k_lu_panel  one workgroup: partial pivoting inside the panel (rows swapped in the panel only)
//   k_lu_swap   the panel's row swaps on every other column
//   k_lu_trsm   U12 = L11^-1 A12 (a thread per column)
//   k_lu_gemm   A22 -= L21 U12 (64 x 64 tiles, K = LU_NB through LDS)
// then k_lu_backsub solves U x = y row by row.
constexpr int LU_NB = 32;

__global__ void k_compact_full(int D, int m, int gauge_axis, const double *__restrict__ Apk,
                               const double *__restrict__ bfull, double *__restrict__ F) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;  // j in [0, D]: column D is the right-hand side
  if (j > D) return;
  const int gi = keep_index(i, gauge_axis);
  if (j == D) { F[(size_t)i * (D + 1) + D] = bfull[gi]; return; }
  const int gj = keep_index(j, gauge_axis);
  const int r = min(gi, gj), c = max(gi, gj);
  const int k = r / 9;
  F[(size_t)i * (D + 1) + j] = Apk[strip_offset(k, m) + (size_t)(r - 9 * k) * (9 * (m - k)) + (c - 9 * k)];
}

__global__ __launch_bounds__(1024) void k_lu_panel(double *__restrict__ F, int D, int j0, int nb, int *__restrict__ ipiv,
                                                   int *__restrict__ flag) {
  __shared__ double s_val[1024];
  __shared__ int s_idx[1024];
  __shared__ int s_piv;
  const int tid = threadIdx.x, nt = blockDim.x, ld = D + 1;
  for (int c = 0; c < nb; ++c) {
    const int col = j0 + c;
    double best = -1.0;
    int bi = col;
    for (int i = col + tid; i < D; i += nt) {
      const double v = fabs(F[(size_t)i * ld + col]);
      if (v > best) { best = v; bi = i; }
    }
    s_val[tid] = best; s_idx[tid] = bi;
    __syncthreads();
    for (int off = nt >> 1; off > 0; off >>= 1) {
      if (tid < off && (s_val[tid + off] > s_val[tid] || (s_val[tid + off] == s_val[tid] && s_idx[tid + off] < s_idx[tid]))) {
        s_val[tid] = s_val[tid + off]; s_idx[tid] = s_idx[tid + off];
      }
      __syncthreads();
    }
    if (tid == 0) {
      s_piv = s_idx[0];
      ipiv[col] = s_idx[0];
      if (!(s_val[0] > 0.0)) atomicOr(flag, 4);  // exactly singular: LAPACK's info > 0 -> LinAlgError
    }
    __syncthreads();
    const int p = s_piv;
    if (p != col && tid < nb) {  // swap inside the panel (k_lu_swap does the other columns)
      const double a = F[(size_t)col * ld + j0 + tid];
      F[(size_t)col * ld + j0 + tid] = F[(size_t)p * ld + j0 + tid];
      F[(size_t)p * ld + j0 + tid] = a;
    }
    __syncthreads();
    const double ipv = 1.0 / F[(size_t)col * ld + col];
    const int rem = nb - c - 1;  // panel columns right of this one
    // multipliers and the rank-1 update of the rest of the panel, one row per (rem + 1) consecutive threads' work
    for (long long q = tid; q < (long long)(D - col - 1) * (rem + 1); q += nt) {
      const int i = col + 1 + (int)(q / (rem + 1)), cc = (int)(q % (rem + 1));
      const double mult = F[(size_t)i * ld + col] * ipv;
      if (cc == 0) ;  // (the multiplier itself is stored after the update below: other threads of the row still read the old value)
      else F[(size_t)i * ld + col + cc] -= mult * F[(size_t)col * ld + col + cc];
    }
    __syncthreads();
    for (int i = col + 1 + tid; i < D; i += nt) F[(size_t)i * ld + col] *= ipv;
    __syncthreads();
  }
}

__global__ void k_lu_swap(double *__restrict__ F, int D, int j0, int nb, const int *__restrict__ ipiv) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;  // every column of [F | rhs] outside the panel
  if (j >= j0) j += nb;
  if (j > D) return;
  const int ld = D + 1;
  for (int c = 0; c < nb; ++c) {
    const int r = j0 + c, p = ipiv[r];
    if (p != r) {
      const double a = F[(size_t)r * ld + j];
      F[(size_t)r * ld + j] = F[(size_t)p * ld + j];
      F[(size_t)p * ld + j] = a;
    }
  }
}

__global__ __launch_bounds__(256) void k_lu_trsm(double *__restrict__ F, int D, int j0, int nb) {
  __shared__ double L[LU_NB][LU_NB + 1];
  const int ld = D + 1;
  for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) L[e / nb][e % nb] = F[(size_t)(j0 + e / nb) * ld + j0 + e % nb];
  __syncthreads();
  const int j = j0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (j > D) return;
  double u[LU_NB];
  for (int c = 0; c < nb; ++c) {
    double v = F[(size_t)(j0 + c) * ld + j];
    for (int cc = 0; cc < c; ++cc) v -= L[c][cc] * u[cc];
    u[c] = v;
    F[(size_t)(j0 + c) * ld + j] = v;
  }
}

__global__ __launch_bounds__(256) void k_lu_gemm(double *__restrict__ F, int D, int j0, int nb) {
  __shared__ double sL[64][LU_NB + 1], sU[LU_NB][65];
  const int ld = D + 1, i0 = j0 + nb + blockIdx.y * 64, c0 = j0 + nb + blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * nb; e += 256) {
    const int r = e / nb, k = e % nb;
    sL[r][k] = (i0 + r < D) ? F[(size_t)(i0 + r) * ld + j0 + k] : 0.0;
  }
  for (int e = threadIdx.x; e < nb * 64; e += 256) {
    const int k = e >> 6, c = e & 63;
    sU[k][c] = (c0 + c <= D) ? F[(size_t)(j0 + k) * ld + c0 + c] : 0.0;
  }
  __syncthreads();
  const int tr = threadIdx.x >> 4, tc = threadIdx.x & 15;
  double acc[4][4] = {};
  for (int k = 0; k < nb; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = sL[tr + 16 * u][k];
#pragma unroll
    for (int v = 0; v < 4; ++v) b[v] = sU[k][tc + 16 * v];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + tr + 16 * u, c = c0 + tc + 16 * v;
      if (i < D && c <= D) F[(size_t)i * ld + c] -= acc[u][v];
    }
}

// U x = y (y = column D of the factored augmented matrix), row by row from the bottom; then the gauge slots
__global__ __launch_bounds__(1024) void k_lu_backsub(double *__restrict__ F, int D, int m, int gauge_axis,
                                                     double *__restrict__ dxi_full) {
  __shared__ double s_red[16];
  const int tid = threadIdx.x, nt = blockDim.x, ld = D + 1;
  for (int k = D - 1; k >= 0; --k) {
    double part = 0.0;
    for (int j = k + 1 + tid; j < D; j += nt) part += F[(size_t)k * ld + j] * F[(size_t)j * ld + D];
    const double t = block_sum(part, s_red);
    if (tid == 0) F[(size_t)k * ld + D] = (F[(size_t)k * ld + D] - t) / F[(size_t)k * ld + k];
    __syncthreads();
  }
  for (int i = tid; i < 9 * m; i += nt) dxi_full[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < D; i += nt) dxi_full[keep_index(i, gauge_axis)] = F[(size_t)i * ld + D];
}

// ------------------------------------------------------------------ K6a: trial cameras (ref :263-281, utils.py:10-29)
__global__ void k_update_cams(int m, const double *__restrict__ cam15, const double *__restrict__ dxi,
                              double *__restrict__ out15) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const double *in = cam15 + (size_t)k * CAM_IN, *d = dxi + 9 * (size_t)k;
  double *o = out15 + (size_t)k * CAM_IN;
#pragma unroll
  for (int i = 0; i < 6; ++i) o[i] = in[i] + d[i];
  const double w0 = d[6], w1 = d[7], w2 = d[8];
  if (w0 == 0.0 && w1 == 0.0 && w2 == 0.0) {  // exact-zero shortcut (utils.py:14-15)
#pragma unroll
    for (int i = 0; i < 9; ++i) o[6 + i] = in[6 + i];
    return;
  }
  const double th = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
  const double n0 = w0 / th, n1 = w1 / th, n2 = w2 / th;
  const double cs = cos(th), sn = sin(th), oc = 1.0 - cs;
  const double Q[9] = {oc * n0 * n0 + cs,      oc * n0 * n1 - sn * n2, oc * n0 * n2 + sn * n1,
                       oc * n1 * n0 + sn * n2, oc * n1 * n1 + cs,      oc * n1 * n2 - sn * n0,
                       oc * n2 * n0 - sn * n1, oc * n2 * n1 + sn * n0, oc * n2 * n2 + cs};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      o[6 + 3 * r + cc] = Q[3 * r] * in[6 + cc] + Q[3 * r + 1] * in[9 + cc] + Q[3 * r + 2] * in[12 + cc];
}

// ------------------------------------------------------------------ K5
// dX_a = -E_a^-1 (sum_o F_ao dxi_k + dP_a), X' = X + dX  (ref :152, :260-261).  Eight lanes per
// point, one observation per lane.  y_o = 2 Jx^T (Jc dxi_k) is RECOMPUTED from the committed point
// and the committed camera (LDS table) instead of re-reading the 128-byte records: the kernel then
// moves 24 B/point + 4 B/observation instead of 128 B/observation (1.7 GB -> 0.25 GB at config 3),
// and it is evaluated as a directional derivative (obs_backsub: ~75 fp64 operations, the Jacobian rows
// are never formed) -- the same linear map the Schur kernel assembled, implied columns included.
// The trial cost is k_cost on the trial state (K6).
// BT threads per block: 256, or 1024 once the camera tables (m x 28 doubles) leave room for one block per CU only
// (beyond ~230 cameras: four waves per CU then; config 4's shard 0.47 -> see profiles/r04_m_*)
template <int G, int BT = 256, bool GCAM = false>
__global__ __launch_bounds__(BT) void k_backsub(long long npts, int m, const long long *__restrict__ pt_ptr,
                                                 const int *__restrict__ cam_idx, const double *__restrict__ PB,
                                                 const double *__restrict__ dxi, const double *__restrict__ X,
                                                 const double *__restrict__ cam15, double f0, double *__restrict__ Xt,
                                                 double *__restrict__ dX, const double *__restrict__ gcam, const double *__restrict__ gdxi) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *s_dxi = smem, *s_cam = smem + DXI_LDS * m;
  if (!GCAM) {
  for (int i0 = threadIdx.x; i0 < 9 * m; i0 += 4 * blockDim.x) {  // (loads in batches of four: see k_point_inv)
    double t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = dxi[min(i0 + u * (int)blockDim.x, 9 * m - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      if (i < 9 * m) s_dxi[(i / 9) * DXI_LDS + i % 9] = t[u];
    }
  }
  load_cams_to_lds(cam15, m, f0, s_cam);
  __syncthreads();
  }
  // G lanes per point (template).  Measured with 2 / 4 / 8 lanes: config 3 (10 observations per point)
  // 0.198 / 0.207 / 0.235 ms, config-4 shard (25 per point) 0.783 / 0.816 / 0.873 ms; one lane: 0.208
  const int s = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  const long long a_first = (long long)blockIdx.x * (BT / G) + grp, a_step = (long long)gridDim.x * (BT / G);
  long long nx0 = 0, nx1 = 0;  // observation range of the NEXT point of this group, requested one iteration ahead
  if (a_first < npts) { nx0 = pt_ptr[a_first]; nx1 = pt_ptr[a_first + 1]; }
  for (long long a = a_first; a < npts; a += a_step) {
    const long long o0 = nx0, o1 = nx1;
    if (a + a_step < npts) { nx0 = pt_ptr[a + a_step]; nx1 = pt_ptr[a + a_step + 1]; }
    const double *pb = PB + PBS * a;
    const double pb0 = pb[0], pb1 = pb[1], pb2 = pb[2], pb3 = pb[3], pb4 = pb[4], pb5 = pb[5], pb6 = pb[6], pb7 = pb[7],
                 pb8 = pb[8];
    const double Xa0 = X[3 * a], Xa1 = X[3 * a + 1], Xa2 = X[3 * a + 2];
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    // the camera ids of this lane's first PF observations in ONE batch (unconditional loads on clamped
    // indices: one memory latency per point instead of one per observation), the rare rest one by one
    constexpr int PF = 4;
    int kk[PF];
    const long long olast = max(o1 - 1, o0);
#pragma unroll
    for (int u = 0; u < PF; ++u) kk[u] = cam_idx[min(o0 + s + G * u, olast)];
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (o0 + s + G * u < o1) {
        double t0, t1, t2;
        obs_backsub(Xa0, Xa1, Xa2, GCAM ? gcam + (size_t)kk[u] * CAM_LDS : s_cam + kk[u] * CAM_LDS, GCAM ? gdxi + (size_t)DXI_LDS * kk[u] : s_dxi + DXI_LDS * kk[u], f0, t0, t1, t2);
        y0 += t0;
        y1 += t1;
        y2 += t2;
      }
    for (long long o = o0 + s + G * PF; o < o1; o += G) {
      const int k = cam_idx[o];
      double t0, t1, t2;
      obs_backsub(Xa0, Xa1, Xa2, GCAM ? gcam + (size_t)k * CAM_LDS : s_cam + k * CAM_LDS, GCAM ? gdxi + (size_t)DXI_LDS * k : s_dxi + DXI_LDS * k, f0, t0, t1, t2);
      y0 += t0;
      y1 += t1;
      y2 += t2;
    }
#pragma unroll
    for (int msk = 1; msk < G; msk <<= 1) {  // fixed butterfly: every lane ends with the group sum
      y0 += __shfl_xor(y0, msk, G);
      y1 += __shfl_xor(y1, msk, G);
      y2 += __shfl_xor(y2, msk, G);
    }
    if (s == 0) {
      const double d0 = -(pb0 * y0 + pb1 * y1 + pb2 * y2) - pb6;
      const double d1 = -(pb1 * y0 + pb3 * y1 + pb4 * y2) - pb7;
      const double d2 = -(pb2 * y0 + pb4 * y1 + pb5 * y2) - pb8;
      dX[3 * a] = d0; dX[3 * a + 1] = d1; dX[3 * a + 2] = d2;
      Xt[3 * a] = Xa0 + d0; Xt[3 * a + 1] = Xa1 + d1; Xt[3 * a + 2] = Xa2 + d2;
    }
  }
}

// Beyond LDS_CAMERAS: the expanded camera rows (and, for the back-substitution, the padded update rows) in device memory
__global__ void k_cam_tables(int m, const double *__restrict__ cam15, const double *__restrict__ dxi, double f0, double *__restrict__ cam18,
                             double *__restrict__ dxi10) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  expand_cam(cam15 + (size_t)k * CAM_IN, f0, cam18 + (size_t)k * CAM_LDS);
  if (dxi) {
    for (int i = 0; i < 9; ++i) dxi10[(size_t)k * DXI_LDS + i] = dxi[9 * (size_t)k + i];
    dxi10[(size_t)k * DXI_LDS + 9] = 0.0;
  }
}

// residual-only pass at a given state (initial cost, ref :85-87)
template <bool GCAM>
__global__ __launch_bounds__(512) void k_cost(long long nobs, int m, const double *__restrict__ cam15,
                                              const double *__restrict__ X, const int *__restrict__ obs_pt,
                                              const int *__restrict__ cam_idx, const double2 *__restrict__ xy,
                                              double f0, double *__restrict__ partials, const double *__restrict__ gcam) {
  extern __shared__ __attribute__((aligned(16))) double s_cam[];
  __shared__ double s_red[16];
  if (!GCAM) {
    load_cams_to_lds(cam15, m, f0, s_cam);
    __syncthreads();
  }
  double cost = 0.0;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // two dependent memory latencies per observation (its indices, then its point): the next
  // observation's indices are requested before this one is evaluated (clamped, unconditional loads)
  long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long olast = nobs - 1;
  int a_n = obs_pt[min(o, olast)], k_n = cam_idx[min(o, olast)];
  double2 z_n = xy[min(o, olast)];
  for (; o < nobs; o += stride) {
    const int a = a_n, k = k_n;
    const double2 z = z_n;
    const double X0 = X[3 * (size_t)a], X1 = X[3 * (size_t)a + 1], X2 = X[3 * (size_t)a + 2];
    const long long on = min(o + stride, olast);
    a_n = obs_pt[on];
    k_n = cam_idx[on];
    z_n = xy[on];
    cost += obs_cost(X0, X1, X2, GCAM ? gcam + (size_t)k * CAM_LDS : s_cam + k * CAM_LDS, z.x, z.y, f0);
  }
  const double t = block_sum(cost, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// out[0] = cost; the status flags ride along in the next 8 bytes so that the host needs ONE
// 16-byte device-to-host copy per trial step.
// `mail` (may be null): the same two words once more in pinned HOST memory, then -- behind a system-scope fence -- the
// sequence number the host is spinning on: the cost reaches the LM loop without a copy kernel and without waking a
// thread that sleeps in hipStreamSynchronize (~35 us between the last kernel of a step and the first of the next).
__global__ __launch_bounds__(1024) void k_sum_partials(const double *__restrict__ partials, int n,
                                                       double *__restrict__ out, const int *__restrict__ flag,
                                                       double *mail, unsigned long long seq) {
  __shared__ double s_red[16];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += partials[i];
  const double t = block_sum(v, s_red);
  if (threadIdx.x == 0) {
    const int fl = *flag;
    out[0] = t;
    reinterpret_cast<int *>(out + 1)[0] = fl;
    if (mail) {
      mail[0] = t;
      reinterpret_cast<int *>(mail + 1)[0] = fl;
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(mail + 2), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------ projection (scene side of BA)
// x_o = K_k [R_k^T | -R_k^T t_k] [X_a; 1], inhomogeneous (ref lib/camera.py:13-14 get_camera_matrix,
// :30-34 project_points, :74-81 calc_projected_points) for an observation list.  The 3x4 camera
// matrices are formed once per block in LDS in the reference's order of operations (K times the
// stacked [R^T, -R^T t]); one thread per observation, coalesced reads of the index arrays and a
// coalesced 16-byte store.  obs_pt == nullptr: dense grid, observation o = point * m + camera.
__global__ __launch_bounds__(256) void k_project_obs(long long nobs, int m, const double *__restrict__ X,
                                                     const double *__restrict__ K, const double *__restrict__ R,
                                                     const double *__restrict__ t, const int *__restrict__ obs_pt,
                                                     const int *__restrict__ cam_idx, double2 *__restrict__ xy) {
  extern __shared__ double sP[];  // [m][12]
  for (int k = threadIdx.x; k < m; k += blockDim.x) {
    const double *Kk = K + 9 * (size_t)k, *Rk = R + 9 * (size_t)k, *tk = t + 3 * (size_t)k;
    double Rt[3][4];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Rt[i][j] = Rk[3 * j + i];
      Rt[i][3] = -(Rt[i][0] * tk[0] + Rt[i][1] * tk[1] + Rt[i][2] * tk[2]);
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) sP[12 * k + 4 * i + j] = Kk[3 * i] * Rt[0][j] + Kk[3 * i + 1] * Rt[1][j] + Kk[3 * i + 2] * Rt[2][j];
  }
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < nobs; o += stride) {
    const long long a = obs_pt ? obs_pt[o] : o / m;
    const int k = cam_idx ? cam_idx[o] : (int)(o - a * m);
    const double *Xa = X + 3 * a, *P = sP + 12 * k;
    const double p0 = Xa[0] * P[0] + Xa[1] * P[1] + Xa[2] * P[2] + P[3];
    const double p1 = Xa[0] * P[4] + Xa[1] * P[5] + Xa[2] * P[6] + P[7];
    const double p2 = Xa[0] * P[8] + Xa[1] * P[9] + Xa[2] * P[10] + P[11];
    xy[o] = make_double2(p0 / p2, p1 / p2);
  }
}

// ------------------------------------------------------------------ way back to the input frame (ref :242-258)
// X <- s X R0^T + t0, t <- s t R0^T + t0, R <- R0 R on the committed state, in place: what the
// reference's optimize() does on the host before it returns (:198-200), without moving 48 MB of
// points across PCIe and through NumPy temporaries (8.5 ms at config 3, a quarter of a 10-iteration
// optimize()).
__global__ __launch_bounds__(256) void k_similarity(long long npts, int m, double *__restrict__ X, double *__restrict__ cam15,
                                                    const double *__restrict__ T13) {
  const double *R0 = T13, *t0 = T13 + 9;
  const double s = T13[12];
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a < npts) {
    double *x = X + 3 * a;
    const double x0 = s * x[0], x1 = s * x[1], x2 = s * x[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = x0 * R0[3 * i] + x1 * R0[3 * i + 1] + x2 * R0[3 * i + 2] + t0[i];
  }
  if (a < m) {
    double *c = cam15 + (size_t)a * CAM_IN;
    const double p0 = s * c[3], p1 = s * c[4], p2 = s * c[5];
    double Rn[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rn[3 * i + j] = R0[3 * i] * c[6 + j] + R0[3 * i + 1] * c[9 + j] + R0[3 * i + 2] * c[12 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) c[3 + i] = p0 * R0[3 * i] + p1 * R0[3 * i + 1] + p2 * R0[3 * i + 2] + t0[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[6 + i] = Rn[i];
  }
}

// ------------------------------------------------------------------ Schur index on the device (mvba_create)
// The slot form's index -- every (point, camera pair) item, counting-sorted by pair with ascending points inside a
// pair, dealt into sub-lists, then merged wave by wave into step-major rows -- built by kernels instead of 16 host
// threads (0.7 s of a 0.83 s mvba_create at config 3).  A STABLE counting sort: wave w owns a contiguous chunk of
// points and walks them in order, its lanes taking the items of one (point, first camera) row at a time -- distinct
// pairs, so the wave's private histogram in LDS needs no atomics and an item's rank inside its pair is
// [items of earlier waves] + [items of this wave so far]: exactly the position the host's sequential pass gives it.
constexpr int IDX_WAVES = 4;  // waves per block of the counting / filling kernels (one P-int histogram each in LDS)

__device__ __forceinline__ int idx_pair_id(int k, int l, int m) { return k * m - k * (k - 1) / 2 + (l - k); }

// hist[w][q] = items of pair q among the points of wave w's chunk.
// GLOBAL: the pair histogram does not fit LDS (more than ~138 cameras: 125 k pairs at 500) -- the wave counts in its
// own row of `hist` in device memory instead (zeroed by the caller).  Same walk, same ranks; a row visit is then a
// dependent round trip to L2 (agent-scope accesses: a later visit of the same pair, possibly from another lane of
// this wave, must see the count), ~1-2 us per (point, first camera) row instead of ~0.1.
template <bool GLOBAL>
__global__ __launch_bounds__(64 * IDX_WAVES) void k_idx_count(long long N, int m, int P, const long long *__restrict__ pt_ptr,
                                                              const int *__restrict__ cam_idx, int chunk,
                                                              int *__restrict__ hist, const int *__restrict__ order) {
  extern __shared__ int s_hist_all[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long w = (long long)blockIdx.x * IDX_WAVES + wv;
  int *s_hist = GLOBAL ? hist + (size_t)w * P : s_hist_all + (size_t)wv * P;
  if (!GLOBAL) {
    for (int q = lane; q < P; q += 64) s_hist[q] = 0;
    wave_sync();
  }
  const long long a0 = w * chunk, a1 = min(N, a0 + chunk);
  for (long long ai = a0; ai < a1; ++ai) {
    const long long a = order ? order[ai] : ai;  // (the sweep order of the points: see `point_order` in mvba_create)
    const long long o0 = pt_ptr[a];
    const int d = (int)(pt_ptr[a + 1] - o0);
    for (int i = 0; i < d; ++i) {
      const int k = cam_idx[o0 + i];
      for (int j = i + lane; j < d; j += 64) {  // distinct pairs per instruction
        int *c = s_hist + idx_pair_id(k, cam_idx[o0 + j], m);
        if (GLOBAL) __hip_atomic_store(c, __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *c += 1;
      }
      wave_sync();
    }
  }
  if (!GLOBAL)
    for (int q = lane; q < P; q += 64) hist[(size_t)w * P + q] = s_hist[q];
}

// exclusive prefix over the waves, per pair (in place); cnt[q] = total
__global__ void k_idx_scan(int P, int n_waves, int *__restrict__ hist, long long *__restrict__ cnt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P) return;
  long long run = 0;
  for (int w = 0; w < n_waves; ++w) {
    const int v = hist[(size_t)w * P + q];
    hist[(size_t)w * P + q] = (int)run;
    run += v;
  }
  cnt[q] = run;
}

// the same walk again: item r of pair q goes to sub-list r % S[q], position r / S[q] (the host's dealing)
// (GLOBAL: the running ranks live in the wave's row of `hist` itself, which the scan has turned into start ranks)
template <bool GLOBAL>
__global__ __launch_bounds__(64 * IDX_WAVES) void k_idx_fill(long long N, int m, int P, const long long *__restrict__ pt_ptr,
                                                             const int *__restrict__ cam_idx, int chunk,
                                                             int *__restrict__ hist, const int *__restrict__ S,
                                                             const int *__restrict__ vp_ptr, const long long *__restrict__ vp_off,
                                                             int *__restrict__ it_k, int *__restrict__ it_l, int *__restrict__ it_a,
                                                             const int *__restrict__ order) {
  extern __shared__ int s_hist_all[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long w = (long long)blockIdx.x * IDX_WAVES + wv;
  int *s_rank = GLOBAL ? hist + (size_t)w * P : s_hist_all + (size_t)wv * P;
  if (!GLOBAL) {
    for (int q = lane; q < P; q += 64) s_rank[q] = hist[(size_t)w * P + q];  // items of earlier waves
    wave_sync();
  }
  const long long a0 = w * chunk, a1 = min(N, a0 + chunk);
  for (long long ai = a0; ai < a1; ++ai) {
    const long long a = order ? order[ai] : ai;
    const long long o0 = pt_ptr[a];
    const int d = (int)(pt_ptr[a + 1] - o0);
    for (int i = 0; i < d; ++i) {
      const int k = cam_idx[o0 + i];
      for (int j = i + lane; j < d; j += 64) {
        const int q = idx_pair_id(k, cam_idx[o0 + j], m);
        int r;
        if (GLOBAL) {
          r = __hip_atomic_load(s_rank + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(s_rank + q, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          r = s_rank[q];
          s_rank[q] = r + 1;
        }
        const int sq = S[q];
        const long long pos = vp_off[vp_ptr[q] + (r % sq)] + r / sq;
        it_k[pos] = (int)(o0 + i);
        it_l[pos] = (int)(o0 + j);
        it_a[pos] = (int)a;
      }
      wave_sync();
    }
  }
}

// lo[v][r] = first item of list v whose point is >= range_lo[r]  (r = 0 .. nR: the last column is the list's end)
__global__ void k_idx_bounds(int VP, int nR, const long long *__restrict__ vp_off, const long long *__restrict__ range_lo,
                             const int *__restrict__ it_a, long long *__restrict__ lo, const int *__restrict__ rank) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)VP * (nR + 1)) return;
  const int v = (int)(t / (nR + 1)), r = (int)(t - (long long)v * (nR + 1));
  long long b = vp_off[v], e = vp_off[v + 1];
  const long long key = range_lo[r];
  while (b < e) {  // lower_bound
    const long long mid = (b + e) >> 1;
    if ((rank ? rank[it_a[mid]] : it_a[mid]) < key) b = mid + 1;  // (a list is ascending in sweep order; ranges are contiguous in it)
    else e = mid;
  }
  lo[t] = b;
}

// The bounded-skew merge of one wave's 21 lists (see k_schur_slots), lane = slot.  FILL = false: count the steps.
// FILL = true: write the step-major rows (record indices RELATIVE to the range's first observation), the pacing table
// (steps taken when the slowest slot leaves a segment) and the padding rows (`pad_obs` = 0: the range's first record,
// any finite one will do, times `pad_pt` = N: the all-zero point row).
template <bool FILL>
__global__ __launch_bounds__(64) void k_idx_merge(long long n_waves, int nR, int nSeg, long long skew, long long segG,
                                                  const long long *__restrict__ sl_beg, const int *__restrict__ sl_len,
                                                  const long long *__restrict__ range_o0, const int *__restrict__ it_k,
                                                  const int *__restrict__ it_l, const int *__restrict__ it_a,
                                                  const long long *__restrict__ w_beg, int pad_obs, int pad_pt,
                                                  int *__restrict__ w_steps, int *__restrict__ st_k, int *__restrict__ st_l,
                                                  int *__restrict__ st_a, int *__restrict__ seg_end, const long long *__restrict__ pkey) {
  const long long b = blockIdx.x;
  const int lane = threadIdx.x;
  const bool slot = lane < PSTEP;
  long long cur = slot ? sl_beg[b * PSTEP + lane] : 0;
  const long long end = slot ? cur + sl_len[b * PSTEP + lane] : 0;
  const long long o_lo = range_o0[b % nR];
  const long long base = FILL ? w_beg[b] : 0;
  constexpr long long NONE = 1LL << 40;
  int steps = 0, sg = 0;
  // an item's key = where its point sits in the sweep, in observations (pkey[a] = observations of the points swept
  // before a; with the natural order that is a's first observation); the k-side record index is what goes into the row
  // two keys ahead in registers: the next step's key never waits for a load issued in this step
  auto key_at = [&](long long c) -> long long { return pkey[it_a[c]]; };
  long long k0 = cur < end ? key_at(cur) : NONE, k1 = cur + 1 < end ? key_at(cur + 1) : NONE;
  while (true) {
    long long lo = k0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lo = min(lo, __shfl_xor(lo, off, 64));
    if (FILL && lo < NONE && lane == 0)
      while (sg < nSeg && lo >= o_lo + (sg + 1) * segG) seg_end[b * nSeg + sg++] = steps;
    if (lo >= NONE) break;
    const bool take = k0 < NONE && k0 <= lo + skew;
    if (FILL && slot) {
      const long long o = (base + steps) * PSTEP + lane;
      st_k[o] = take ? (int)(it_k[cur] - o_lo) : pad_obs;  // record indices relative to the range's first observation
      st_l[o] = take ? (int)(it_l[cur] - o_lo) : pad_obs;
      st_a[o] = take ? it_a[cur] : pad_pt;
    }
    if (take) {
      ++cur;
      k0 = k1;
      k1 = cur + 1 < end ? key_at(cur + 1) : NONE;
    }
    ++steps;
  }
  if (FILL) {
    if (lane == 0)
      while (sg < nSeg) seg_end[b * nSeg + sg++] = steps;
  } else if (lane == 0) {
    w_steps[b] = steps;
  }
}

// the slot kernel's index: one 256-byte row per step, k[21] | l[21] | a[21] | pad -- ONE LDS-DMA instruction per step
__global__ void k_idx_interleave(long long n_steps, const int *__restrict__ st_k, const int *__restrict__ st_l, const int *__restrict__ st_a,
                                 int *__restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_steps * SLOT_IDX) return;
  const long long st = t / SLOT_IDX;
  const int e = (int)(t - st * SLOT_IDX), which = e / PSTEP, sl = e - which * PSTEP;
  const int *src = which == 0 ? st_k : (which == 1 ? st_l : st_a);
  out[t] = which < 3 ? src[st * PSTEP + sl] : 0;
}

}  // namespace

// ------------------------------------------------------------------ host side
enum { SCHUR_STRIP = 0, SCHUR_PAIRS = 1, SCHUR_SLOTS = 2, SCHUR_DENSE = 3 };
struct mvba_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  long long N = 0, nobs = 0;
  int m = 0, gauge_axis = 0, D = 0, ld = 0, nsp = 0;
  double f0 = 1.0;
  // topology
  long long *d_pt_ptr = nullptr;
  int *d_cam = nullptr, *d_obs_pt = nullptr;
  double2 *d_xy = nullptr;
  int4 *d_csc = nullptr;
  int *d_tiles = nullptr;  // K1 point-aligned wave tiles
  int *d_tile_slot = nullptr;      // points with more than 64 observations: slot of every piece tile in d_PLsplit,
  int4 *d_splits = nullptr;        // (point, first slot, pieces) per such point
  double *d_PLsplit = nullptr;
  int n_splits = 0;
  int n_tiles = 0;
  bool any_split = false;
  long long *d_chunk_ptr = nullptr;
  int nchunks = 1, lseg = 0, nseg = 1, schur_threads = 768, k1_threads = 512;
  // pair-major Schur index (k_schur_pairs): items sorted by (k, l, point), units, per-XCD work queues
  bool use_pairs = true;              // pair-major index present (schur_mode != SCHUR_STRIP)
  int schur_mode = 2;                 // SCHUR_STRIP / SCHUR_PAIRS / SCHUR_SLOTS (see k_schur_slots)
  long long slot_skew = 12288;        // bounded skew of the slot form's step merge, in observations
  long long slot_window = 1LL << 40;  // ... and the window in which all waves of a range take equally many steps (off)
  long long n_items = 0, n_items_offdiag = 0, n_slot_items = 0;
  int n_units = 0, rccl_version = 0, q_max = 0, n_waves = 0, slot_nR = 8, slot_nseg = 0, slot_rounds = 1, slot_groups = 1;
  long long *d_range_o0 = nullptr;    // slot form: first observation of every point range (the record base of its waves)
  long long slot_seg = 8192;          // pacing segment of the slot form, in observations
  bool slot_pace = true;
  int slot_lag = 4;                   // a wave enters segment j only when all waves of its range have left segment j - lag
  int *d_seg_end = nullptr, *d_prog = nullptr;
  bool check_solve = false;           // MVBA_CHECK_SOLVE=1: every accepted dense solve is checked on the host against the packed system it solved
  double check_solve_tol = 1e-8;      // (MVBA_CHECK_SOLVE=<t> with 0 < t < 1: that tolerance -- the tests ask for an impossible one to see the check fire)
  bool gcam = false;                  // more than LDS_CAMERAS cameras: the kernels read the camera tables from device memory (d_cam18, d_dxi10)
  double *d_cam18 = nullptr, *d_dxi10 = nullptr;
  bool index_on_device = false;       // the Schur index was built by the k_idx_* kernels (nothing to upload)
  long long *d_trace = nullptr;       // -DMVBA_SLOT_TRACE builds with MVBA_SLOT_TRACE=<file>: per-wave timings of the last launch
  int4 *d_wdesc = nullptr;
  int *d_it_x = nullptr;              // slot form: the step-major index, 64 ints per step (k[21] | l[21] | a[21] | pad)
  int *d_wunits = nullptr;
  // experiment knobs, read from the environment ONCE in mvba_create (tools/README.md lists them)
  bool pair_static = true, force_big = false;
  int backsub_lanes = 0;
  int *d_it_k = nullptr, *d_it_l = nullptr, *d_it_a = nullptr, *d_unit_ptr = nullptr, *d_q_ptr = nullptr, *d_q_units = nullptr,
      *d_q_head = nullptr;
  int4 *d_units = nullptr;
  double *d_partial = nullptr;
  double *d_dense_part = nullptr;     // SCHUR_DENSE: partial tiles per workgroup
  int *d_dense_obs = nullptr;         // ... and, with missing observations, the observation of every (point, camera) or -1
  int dense_blocks = 0, dense_tiles = 0;
  bool dense_attr_set = false;
  // state: [cur] committed, [1-cur] trial
  double *d_X[2] = {nullptr, nullptr}, *d_cam15[2] = {nullptr, nullptr};
  int cur = 0;
  bool have_params = false, linearized = false, have_trial = false;
  // linearisation
  double2 *d_rec = nullptr;  // [n_obs][8] double2: one 128-B line per observation
  double *d_PL = nullptr, *d_PB = nullptr;
  // reduced system: [A (9m x 9m) | b (9m)] contiguous for the all-reduce
  double *d_Ab = nullptr, *d_Ared = nullptr, *d_Ztiles = nullptr, *d_Lblk = nullptr, *d_dxi = nullptr, *d_dX = nullptr, *d_lu = nullptr;
  int *d_ipiv = nullptr;
  // cost
  double *d_partials = nullptr, *d_cost = nullptr, *h_cost = nullptr;
  double *d_mail = nullptr;               // device address of h_cost (pinned, mapped): the cost kernel's mailbox
  unsigned long long cost_seq = 0;
  bool mail_pending = false;
  int n_partials = 0, cost_grid = 0;
  int *d_flag = nullptr, *h_flag = nullptr;
  unsigned *d_bar = nullptr;
  int n_cu = 1;
  bool chol_onepass = true;  // L^T x = y as one persistent launch (MVBA_CHOL=launches: one launch per super-block)
  bool chol_flow = true;     // ... synchronised point to point (MVBA_CHOL=barriers: round 2's device-wide barriers)
  int trail64_min = 200;     // trailing updates of at least this many 64 x 64 workgroups run k_chol_trail64 (MVBA_TRAIL64_MIN)
  unsigned barrier_polls = 1u << 22;  // what a device-wide barrier of that launch polls before it gives up (MVBA_CHOL_BARRIER_POLLS)
  // comm
  ncclComm_t comm = nullptr;
  mvba_host_allreduce_fn host_ar = nullptr;  // host-staged transport (mvba_comm_init_host) instead of RCCL
  void *host_ar_user = nullptr;
  std::vector<double> host_buf;
  int rank = 0, nranks = 1;
  double *d_allcost = nullptr, *h_allcost = nullptr, *d_sim = nullptr;
  // debug log (mvba_snapshot): committed states kept in device memory, SNAP_SLAB entries per allocation
  std::vector<double *> snap_slabs;
  long long n_snap = 0;
  // profiling
  int profiling = 0;  // 0 off, 1 every phase, 2 the Schur and residual-Jacobian kernels only (mvba_set_profiling)
  mvba_stats stats{};
  struct Ev { int kid; hipEvent_t a, b; };
  std::vector<Ev> pending;
  std::vector<hipEvent_t> pool;
};

namespace {

const char *kKernelNames[MVBA_K_COUNT] = {"resid_jac", "point_blocks", "point_inv", "schur",
                                          "allreduce", "solve",        "backsub_cost", "cost"};

struct Timed {
  mvba_handle *h;
  int kid;
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  Timed(mvba_handle *h_, int kid_) : h(h_), kid(kid_) {
    // level 2: the two kernels a roofline is quoted for, nothing else (every timed phase is two marker packets on the
    // stream and ~10 us of a 2.5 ms step)
    on = h->profiling == 1 || (h->profiling == 2 && (kid == MVBA_K_SCHUR || kid == MVBA_K_RESID_JAC));
    if (!on) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!h->pool.empty()) { e = h->pool.back(); h->pool.pop_back(); }
      else hipEventCreate(&e);
      return e;
    };
    a = get(); b = get();
    hipEventRecord(a, h->stream);
  }
  ~Timed() {
    if (!on) return;
    hipEventRecord(b, h->stream);
    h->pending.push_back({kid, a, b});
  }
};

void drain_events(mvba_handle *h) {  // call after a stream sync
  for (auto &p : h->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      h->stats.ms[p.kid] += ms;
      h->stats.launches[p.kid] += 1;
    }
    h->pool.push_back(p.a);
    h->pool.push_back(p.b);
  }
  h->pending.clear();
}

template <typename T>
int dmalloc(T **p, size_t n) {
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  const hipError_t e = hipMalloc((void **)p, bytes);
  if (e != hipSuccess) {  // say how much was asked for and how much there is: "out of memory" alone does not tell a scene from a knob
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    return fail(MVBA_ERR_HIP, std::string("hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e) + " (" + std::to_string(fr >> 20) +
                                  " MiB free of " + std::to_string(tot >> 20) + ")");
  }
  return MVBA_OK;
}

int sync_and_drain(mvba_handle *h) {
  MVBA_HIP(hipStreamSynchronize(h->stream));
  drain_events(h);
  return MVBA_OK;
}

// Sum of the per-rank costs in rank order and the OR of the per-rank status flags: both identical on
// every rank, so that all ranks take the same accept/reject, LU-rescue and error decisions (a
// rank that branched alone would leave the others waiting in the next collective).  (C1b: one
// all-gather of 16 bytes per rank per trial, on top of C1.)
inline void cpu_relax() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  __builtin_ia32_pause();
#endif
}

// Single rank, not every phase timed: the cost kernel mails its result to pinned host memory (see k_sum_partials).
// Returns the device address of the mailbox and advances the sequence number, or null for the copy + sync path.
double *cost_mail(mvba_handle *h) {
  if (h->comm || h->host_ar || h->profiling == 1 || !h->d_mail) return nullptr;
  ++h->cost_seq;
  h->mail_pending = true;
  return h->d_mail;
}

int global_cost(mvba_handle *h, double *E) {
  if (h->mail_pending) {
    h->mail_pending = false;
    volatile unsigned long long *seq = reinterpret_cast<volatile unsigned long long *>(h->h_cost + 2);
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (unsigned spins = 0;; ++spins) {
      if (*seq == h->cost_seq) { seen = true; break; }
      // (everything earlier on the stream is complete once the number has arrived; if it does not arrive -- a fault, a
      // hung kernel -- the ordinary synchronisation below reports what happened)
      if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
      cpu_relax();
    }
    if (seen) {
      std::atomic_thread_fence(std::memory_order_acquire);
      drain_events(h);  // (level 2: the timers of K1 and K3 ended before the cost kernel started)
      *E = h->h_cost[0];
      return MVBA_OK;
    }
    int rc = sync_and_drain(h);
    if (rc) return rc;
    *E = h->h_cost[0];
    return MVBA_OK;
  }
  if (h->comm) {
    Timed t(h, MVBA_K_ALLREDUCE);
    ncclResult_t r = g_rccl.AllGather(h->d_cost, h->d_allcost, 2, ncclDouble, h->comm, h->stream);
    if (r != ncclSuccess) return fail(MVBA_ERR_RCCL, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
    MVBA_HIP(hipMemcpyAsync(h->h_allcost, h->d_allcost, 2 * sizeof(double) * h->nranks, hipMemcpyDeviceToHost, h->stream));
  } else {
    MVBA_HIP(hipMemcpyAsync(h->h_cost, h->d_cost, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));  // cost + flags
  }
  int rc = sync_and_drain(h);
  if (rc) return rc;
  if (h->host_ar) {  // all-gather as a sum of one-hot slices: {cost, flags as a count} per rank
    std::vector<double> v(2 * (size_t)h->nranks, 0.0);
    v[2 * h->rank] = h->h_cost[0];
    v[2 * h->rank + 1] = (double)*h->h_flag;  // small non-negative integer: exact in a double
    if (h->host_ar(h->host_ar_user, v.data(), (int64_t)v.size())) return fail(MVBA_ERR_RCCL, "host all-reduce callback failed");
    double s = 0.0;
    int fl = 0;
    for (int i = 0; i < h->nranks; ++i) { s += v[2 * i]; fl |= (int)v[2 * i + 1]; }
    *E = s;
    *h->h_flag = fl;
  } else if (h->comm) {
    double s = 0.0;
    int fl = 0;
    for (int i = 0; i < h->nranks; ++i) {
      s += h->h_allcost[2 * i];
      int f;
      memcpy(&f, &h->h_allcost[2 * i + 1], sizeof(int));
      fl |= f;
    }
    *E = s;
    *h->h_flag = fl;
  } else {
    *E = *h->h_cost;
  }
  return MVBA_OK;
}

// the camera tables in device memory for the kernels that cannot hold them in LDS (no-op up to LDS_CAMERAS cameras)
void cam_tables(mvba_handle *h, const double *cam15, const double *dxi) {
  if (h->gcam) hipLaunchKernelGGL(k_cam_tables, dim3((h->m + 255) / 256), dim3(256), 0, h->stream, h->m, cam15, dxi, h->f0, h->d_cam18, h->d_dxi10);
}
void launch_cost_kernel(mvba_handle *h, const double *cam15, const double *X) {
  const size_t lds = h->gcam ? 0 : (size_t)h->m * CAM_LDS * sizeof(double);
  cam_tables(h, cam15, nullptr);
  // (512 threads once the camera table leaves room for two blocks per CU only: config 4's 500 cameras)
  hipLaunchKernelGGL(h->gcam ? k_cost<true> : k_cost<false>, dim3(h->cost_grid), dim3(lds > 40 * 1024 ? 512 : 256), lds, h->stream, h->nobs, h->m, cam15, X,
                     h->d_obs_pt, h->d_cam, h->d_xy, h->f0, h->d_partials, h->d_cam18);
}
int launch_cost(mvba_handle *h, const double *cam15, const double *X) {
  Timed t(h, MVBA_K_COST);
  launch_cost_kernel(h, cam15, X);
  double *mail = cost_mail(h);  // (advances cost_seq: sequenced before the launch reads it)
  const unsigned long long seq = h->cost_seq;
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, h->stream, h->d_partials, h->cost_grid, h->d_cost, h->d_flag, mail, seq);
  MVBA_HIP(hipGetLastError());
  return MVBA_OK;
}

// mvba_create, xy_layout 1: the observations of a fully visible scene arrive as image planes [m][N] and leave in observation
// order [N][m] (a wave reads 1 KiB of one plane and writes 64 records m * 16 bytes apart; once per engine)
__global__ __launch_bounds__(256) void k_xy_from_planes(const double2 *__restrict__ planes, long long N, int m, double2 *__restrict__ xy) {
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  if (a < N) xy[a * m + k] = planes[(long long)k * N + a];
}

}  // namespace

extern "C" {
#ifdef MVBA_DENSE_TRACE
int mvba_dense_trace_read(long long *out, int n) {  // (timing-only build)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_trace), sizeof(long long) * n, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 3;
}
#endif
#ifdef MVBA_BS_TRACE
int mvba_debug_bs_trace(long long *out /* [8 * 256] */) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bs_trace), sizeof(long long) * 8 * 256) == hipSuccess ? 0 : 1;
}
#endif


const char *mvba_version(void) { return "mvba 0.2 (gfx950)"; }
const char *mvba_last_error(void) { return g_err.c_str(); }
const char *mvba_kernel_name(int32_t k) { return (k >= 0 && k < MVBA_K_COUNT) ? kKernelNames[k] : ""; }

int mvba_device_count(int32_t *count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(MVBA_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return MVBA_OK;
}

int mvba_create(const mvba_problem *p, mvba_handle **out) {
  if (!p || !out) return fail(MVBA_ERR_BADARG, "null argument");
  // MVBA_CREATE_TRACE=1: wall time of this function's stages to stderr (tools/time_pipeline.py: the engine's construction is a
  // third of the reference's pipeline at 1 M points x 12 images)
  struct Trace {
    bool on = getenv("MVBA_CREATE_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    std::string s;
    void mark(const char *label) {
      if (!on) return;
      const auto n = std::chrono::steady_clock::now();
      char buf[96];
      snprintf(buf, sizeof buf, " %s %.1f ms;", label, std::chrono::duration<double, std::milli>(n - t).count());
      s += buf;
      t = n;
    }
    ~Trace() { if (on) fprintf(stderr, "mvba_create:%s\n", s.c_str()); }
  } trace;
  if (p->n_points < 0 || p->n_images < 2 || p->n_obs < 0 || !p->pt_ptr || (p->n_obs && (!p->cam_idx || !p->xy)))
    return fail(MVBA_ERR_BADARG, "bad problem sizes or null arrays (need n_images >= 2)");
  if (p->gauge_axis != 0 && p->gauge_axis != 1) return fail(MVBA_ERR_BADARG, "gauge_axis must be 0 or 1");
  if (p->xy_layout != 0 && p->xy_layout != 1) return fail(MVBA_ERR_BADARG, "xy_layout must be 0 (observation order) or 1 (image planes)");
  if (p->xy_layout == 1 && p->n_obs != p->n_points * (int64_t)p->n_images)
    return fail(MVBA_ERR_BADARG, "xy as image planes needs every point observed in every image (n_obs = n_points * n_images)");
  // the kernels keep the whole camera table in LDS (K1: 18 doubles per camera + 8 x 8 KiB of wave
  // tiles; back-substitution: 28 per camera): 160 KiB per workgroup caps the camera count.  (The documented limit is
  // round 1's, from 19 doubles per camera; 18 would admit 682.)
  static_assert(LDS_CAMERAS * CAM_LDS + 8 * 64 * 2 * 8 + 2 <= 160 * 1024 / 8 && LDS_CAMERAS * (CAM_LDS + DXI_LDS) <= 160 * 1024 / 8,
                "the camera tables of LDS_CAMERAS cameras fit one workgroup's LDS");
  // (beyond LDS_CAMERAS the same kernels read the tables from device memory -- round 5; what caps the count now is the dense
  // reduced system: D = 9 m - 7 = 36,857 at 4096 cameras is 10.9 GB of matrix, and the unit descriptors keep camera ids in 16 bits)
  constexpr int MAX_CAMERAS = 4096;
  if (p->n_images > MAX_CAMERAS)
    return fail(MVBA_ERR_BADARG, "n_images = " + std::to_string(p->n_images) + " exceeds the " + std::to_string(MAX_CAMERAS) +
                                     " cameras this build solves a dense reduced system for");
  if (p->pt_ptr[0] != 0 || p->pt_ptr[p->n_points] != p->n_obs) return fail(MVBA_ERR_BADARG, "pt_ptr does not span n_obs");
  if (p->n_obs >= (1LL << 31) || p->n_points >= (1LL << 31))
    return fail(MVBA_ERR_BADARG, "n_obs and n_points per handle must be < 2^31");
  const long long N = p->n_points, nobs = p->n_obs;
  const int m = p->n_images;
  // MVBA_CREATE_TIMING=1: phase times of this call on stderr (a diagnostic, read once here)
  const bool timing = getenv("MVBA_CREATE_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "mvba_create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // validate + build point-of-observation and the camera-major index
  std::vector<int> obs_pt(nobs);
  std::vector<long long> csc_ptr(m + 1, 0);
  for (long long a = 0; a < N; ++a) {
    const long long o0 = p->pt_ptr[a], o1 = p->pt_ptr[a + 1];
    if (o1 < o0 || o1 > nobs) return fail(MVBA_ERR_BADARG, "pt_ptr not monotone / out of range");
    for (long long o = o0; o < o1; ++o) {
      const int k = p->cam_idx[o];
      if (k < 0 || k >= m) return fail(MVBA_ERR_BADARG, "cam_idx out of range");
      if (o > o0 && p->cam_idx[o - 1] >= k) return fail(MVBA_ERR_BADARG, "cam_idx must ascend within a point");
      obs_pt[o] = (int)a;
      csc_ptr[k + 1]++;
    }
  }
  trace.mark("validation pass");
  for (int k = 0; k < m; ++k) csc_ptr[k + 1] += csc_ptr[k];
  // the camera-major index belongs to the strip kernel alone (round 1's K3: MVBA_SCHUR=strip, or more cameras than a
  // pair id holds): 160 MB at config 3 that the other forms never read
  const bool want_strip = (getenv("MVBA_SCHUR") && !strcmp(getenv("MVBA_SCHUR"), "strip")) || m > 65535;
  std::vector<int4> csc(want_strip ? nobs : 0);
  if (want_strip) {
    std::vector<long long> fill(csc_ptr.begin(), csc_ptr.end() - 1);
    for (long long a = 0; a < N; ++a)
      for (long long o = p->pt_ptr[a]; o < p->pt_ptr[a + 1]; ++o)
        csc[fill[p->cam_idx[o]]++] = make_int4((int)o, (int)a, (int)(p->pt_ptr[a + 1] - o), 0);
  }

  lap("validate, obs_pt, csc");
  // K1 wave tiles: whole points packed greedily into <= 64 observations; a point with more than
  // 64 observations is cut into pieces whose tiles are flagged by a complemented (negative) start
  std::vector<int> tiles, tile_slot;
  std::vector<int4> splits;
  int n_split_slots = 0;
  bool any_split = false;
  {
    long long cur0 = 0, fill = 0;
    auto flush = [&](bool split_flag) { tiles.push_back(split_flag ? ~(int)cur0 : (int)cur0); };
    for (long long a = 0; a < N; ++a) {
      const long long d = p->pt_ptr[a + 1] - p->pt_ptr[a];
      if (d > 64) {
        if (fill) { flush(false); cur0 += fill; fill = 0; }
        any_split = true;
        splits.push_back(make_int4((int)a, n_split_slots, (int)((d + 63) / 64), 0));
        for (long long q = 0; q < d; q += 64) {
          tile_slot.resize(tiles.size() + 1, -1);
          tile_slot[tiles.size()] = n_split_slots++;
          flush(true);
          cur0 += std::min<long long>(64, d - q);
        }
        continue;
      }
      if (fill + d > 64) { flush(false); cur0 += fill; fill = 0; }
      fill += d;
    }
    if (fill) { flush(false); cur0 += fill; }
    tiles.push_back((int)nobs);  // terminator (never negative: only its magnitude is used)
  }

  lap("K1 tiles");
  mvba_handle *h = new mvba_handle();
  if (p->device >= 0) {
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) { delete h; return fail(MVBA_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); }
  }
  {
    hipError_t e = hipGetDevice(&h->device);
    if (e != hipSuccess) { delete h; return fail(MVBA_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e)); }
  }
  h->N = N; h->nobs = nobs; h->m = m; h->gauge_axis = p->gauge_axis; h->f0 = p->f0; h->D = 9 * m - 7; h->ld = (h->D + 3) & ~3;
  h->gcam = m > LDS_CAMERAS;
  trace.mark("K1 tiles, device");
#define TRY(x) do { int rc_ = (x); if (rc_) { mvba_destroy(h); return rc_; } } while (0)
#define TRYH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { mvba_destroy(h); return fail(MVBA_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
  // the topology goes up first: the Schur index is built from it on the device
  TRYH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  TRY(dmalloc(&h->d_pt_ptr, N + 1));
  TRY(dmalloc(&h->d_cam, nobs));
  TRYH(hipMemcpy(h->d_pt_ptr, p->pt_ptr, sizeof(long long) * (N + 1), hipMemcpyHostToDevice));
  if (nobs) TRYH(hipMemcpy(h->d_cam, p->cam_idx, sizeof(int) * nobs, hipMemcpyHostToDevice));

  trace.mark("topology upload");
  // Schur launch geometry (measured sweep at config 3, profiles/): ~800 camera-list entries per
  // block is the optimum (tail balance vs strip flush); small problems still get >= 2048 blocks
  // as long as a block keeps >= 128 entries.
  const long long avg_len = std::max<long long>(1, nobs / m);
  h->nchunks = (int)std::max<long long>(1, std::max<long long>(avg_len / 800,
                                                               std::min<long long>((2048 + m - 1) / m, avg_len / 128)));
  // tuning overrides (experiments only)
  if (const char *ev = getenv("MVBA_SCHUR_THREADS")) h->schur_threads = std::max(64, std::min(768, atoi(ev) / 64 * 64));
  {  // K1: the waves of a block share one camera table in LDS and bring 8 KiB of staging each.  Up to ~100 cameras two blocks
    // of 8 waves fill a CU (16 waves: the register limit); beyond that ONE block fits and its size decides the occupancy --
    // the smallest block that reaches the most waves per CU (200 cameras: 16 waves, 0.96 -> 0.70 ms at 1 M points x 10 %;
    // 300: 14, 0.41 -> 0.32-0.37; 500: 11, config 4's shard 1.53-1.58 -> 1.32-1.36; tools/sweep_k1.sh)
    const size_t table = h->gcam ? 0 : (size_t)((m * CAM_LDS + 1) & ~1) * sizeof(double), per_wave = 64 * 2 * REC * sizeof(double);
    int best_w = 8, best_tot = 0;
    for (int w = 8; w <= 16; ++w) {
      const size_t per = table + w * per_wave;
      if (per > 160 * 1024) break;
      const int tot = std::min<int>(16, (int)(160 * 1024 / per) * w);
      if (tot > best_tot) { best_tot = tot; best_w = w; }
    }
    h->k1_threads = 64 * best_w;
  }
  if (const char *ev = getenv("MVBA_K1_THREADS")) h->k1_threads = std::max(64, std::min(1024, atoi(ev) / 64 * 64));
  if (const char *ev = getenv("MVBA_SCHUR_CHUNKS")) h->nchunks = std::max(1, atoi(ev));
  const size_t lds_cap = 150 * 1024;
  h->lseg = (int)std::min<size_t>(m, (lds_cap / 8 - 9) / 81);
  if (const char *ev = getenv("MVBA_SCHUR_LSEG")) h->lseg = std::max(1, std::min(h->lseg, atoi(ev)));
  h->nseg = (m + h->lseg - 1) / h->lseg;
  h->nsp = (h->nseg >= 2 && h->nseg <= 4 && m < 1024) ? h->nseg : 0;
  if (h->nsp && want_strip) {  // segment boundaries inside each entry's remaining (camera-sorted) observations
    for (long long e = 0; e < nobs; ++e) {
      const int *cb = p->cam_idx + csc[e].x, *ce = cb + csc[e].z;
      int w = 0;
      for (int sgi = 1; sgi < h->nsp; ++sgi)
        w |= (int)(std::lower_bound(cb, ce, *cb + sgi * h->lseg) - cb) << (10 * (sgi - 1));
      csc[e].w = w;
    }
  }
  std::vector<long long> chunk_ptr(want_strip ? (size_t)m * (h->nchunks + 1) : 0);
  for (int k = 0; k < m && want_strip; ++k) {
    const int4 *b = csc.data() + csc_ptr[k], *e = csc.data() + csc_ptr[k + 1];
    for (int c = 0; c <= h->nchunks; ++c) {
      const long long a_lo = (long long)((__int128)N * c / h->nchunks);
      const int4 *it = std::lower_bound(b, e, a_lo, [](const int4 &r, long long v) { return r.y < v; });
      chunk_ptr[(size_t)k * (h->nchunks + 1) + c] = it - csc.data();
    }
  }

  lap("strip chunk index");
  // ---- pair-major Schur index (see k_schur_pairs).  Items (obs of k, obs of l, point) for every
  // pair k <= l of a point's cameras, counting-sorted by pair, ascending point inside a pair.
  if (const char *ev = getenv("MVBA_SCHUR"))
    h->schur_mode = !strcmp(ev, "strip") ? SCHUR_STRIP : (!strcmp(ev, "pairs") ? SCHUR_PAIRS : SCHUR_SLOTS);
  if (const char *ev = getenv("MVBA_SLOT_SKEW")) h->slot_skew = std::max(0LL, atoll(ev));
  if (const char *ev = getenv("MVBA_SLOT_WINDOW")) h->slot_window = std::max(1LL, atoll(ev));
  if (const char *ev = getenv("MVBA_SLOT_LAG")) h->slot_lag = std::max(1, atoi(ev));
  if (const char *ev = getenv("MVBA_SLOT_SEG")) { h->slot_seg = std::max(1LL, atoll(ev)); h->slot_pace = atoll(ev) > 0; }
  if (const char *ev = getenv("MVBA_PAIR_STATIC")) h->pair_static = atoi(ev) != 0;
  if (const char *ev = getenv("MVBA_BACKSUB_LANES")) h->backsub_lanes = atoi(ev);
  h->force_big = getenv("MVBA_FORCE_BIG") != nullptr;
  if (const char *ev = getenv("MVBA_CHECK_SOLVE")) {
    const double v = atof(ev);
    h->check_solve = v != 0.0;
    if (v > 0.0 && v < 1.0) h->check_solve_tol = v;
  }
  if (m > 65535) h->schur_mode = SCHUR_STRIP;
  std::vector<int> dense_obs;  // SCHUR_DENSE with missing observations: [N][m] observation of (point, camera) or -1
  {  // up to 21 cameras and most (point, camera) pairs observed: the dense form (no pair index).  Full visibility in camera order
     // (the reference's own scenes): a point's records are read as one contiguous range; otherwise through a table, a missing
     // observation standing as a zero record (the matrix cores multiply the zeros: worth it from ~60 % visibility on)
    bool few = m >= 1 && 9 * m <= 16 * DENSE_MAX_TILES && N > 0, full = few && nobs == N * (long long)m;
    for (long long a = 0; a < N && full; ++a) {
      if (p->pt_ptr[a + 1] - p->pt_ptr[a] != m) { full = false; break; }
      const int *ci = p->cam_idx + p->pt_ptr[a];
      for (int k = 0; k < m; ++k)
        if (ci[k] != k) { full = false; break; }
    }
    const char *ev = getenv("MVBA_SCHUR");
    const bool forced = ev && !strcmp(ev, "dense");
    bool masked = few && !full && (forced || (!ev && (double)nobs >= 0.6 * (double)N * m)) && (long long)N * m < (1LL << 31);
    if (masked) {
      dense_obs.assign((size_t)N * m, -1);
      for (long long a = 0; a < N && masked; ++a)
        for (long long o = p->pt_ptr[a]; o < p->pt_ptr[a + 1]; ++o) {
          int &slot = dense_obs[(size_t)a * m + p->cam_idx[o]];
          if (slot >= 0) { masked = false; break; }  // (a camera twice in one point: the pair-major forms take such scenes)
          slot = (int)o;
        }
      if (!masked) dense_obs.clear();
    }
    if ((full && (!ev || forced)) || masked) h->schur_mode = SCHUR_DENSE;
  }
  trace.mark("form of K3");
  h->use_pairs = h->schur_mode != SCHUR_STRIP && h->schur_mode != SCHUR_DENSE;
  std::vector<int> it_k, it_l, it_a, unit_ptr, q_ptr(9, 0), q_units, st_k, st_l, st_a, wunits, seg_end;
  std::vector<int4> units, wdesc;
  if (h->use_pairs) {
    const long long P = (long long)m * (m + 1) / 2;
    auto pair_id = [m](int k, int l) { return (long long)k * m - (long long)k * (k - 1) / 2 + (l - k); };
    std::vector<long long> cnt(P, 0);
    // Both passes over the points run on host threads that OWN strips (camera k belongs to thread
    // k % n_thr): every thread scans the whole observation list but touches only its own pairs, so
    // there is nothing to lock and the order inside a pair's list stays ascending by point.
    const int n_thr = (int)std::max(1u, std::min({std::thread::hardware_concurrency(), 16u, (unsigned)m}));
    auto on_threads = [&](auto body) {
      std::vector<std::thread> th;
      for (int t = 1; t < n_thr; ++t) th.emplace_back(body, t);
      body(0);
      for (auto &x : th) x.join();
    };
    // The index is built on the DEVICE (k_idx_*): a stable counting sort by pair, every wave walking its own chunk of
    // points with a private pair histogram -- in LDS when P ints x 4 waves per block fit (up to ~138 cameras), else in
    // the wave's own row of a device buffer (at most 2 GiB of rows: 4096 waves at 500 cameras).  MVBA_INDEX=host keeps
    // the host threads, MVBA_INDEX=global forces the device-memory histogram (the tests that the builds are identical).
    const char *idx_env = getenv("MVBA_INDEX");
    const bool hist_lds = (size_t)P * sizeof(int) * IDX_WAVES <= 150 * 1024 && !(idx_env && !strcmp(idx_env, "global"));
    bool dev_build = N > 0 && nobs > 0 && !(idx_env && !strcmp(idx_env, "host"));
    const long long max_idx_waves = hist_lds ? 4096 : std::max<long long>(IDX_WAVES, std::min<long long>(4096, (2LL << 30) / (4 * P)));
    const int idx_chunk = (int)std::max<long long>(32, (N + max_idx_waves - 1) / max_idx_waves);  // points per wave
    const long long idx_waves = dev_build ? ((N + idx_chunk - 1) / idx_chunk + IDX_WAVES - 1) / IDX_WAVES * IDX_WAVES : 0;
    const size_t idx_lds = hist_lds ? P * sizeof(int) * IDX_WAVES : 0;
    int *d_hist = nullptr;
    long long *d_cnt = nullptr;
    if (dev_build) {
      TRY(dmalloc(&d_hist, (size_t)idx_waves * P));
      TRY(dmalloc(&d_cnt, (size_t)P));
      if (hist_lds) {
        TRYH(hipFuncSetAttribute((const void *)k_idx_count<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)idx_lds));
        TRYH(hipFuncSetAttribute((const void *)k_idx_fill<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)idx_lds));
      } else {
        TRYH(hipMemsetAsync(d_hist, 0, sizeof(int) * (size_t)idx_waves * P, h->stream));
      }
      hipLaunchKernelGGL(hist_lds ? k_idx_count<false> : k_idx_count<true>, dim3((unsigned)(idx_waves / IDX_WAVES)), dim3(64 * IDX_WAVES), idx_lds,
                         h->stream, N, m, (int)P, h->d_pt_ptr, h->d_cam, idx_chunk, d_hist, (const int *)nullptr);
      hipLaunchKernelGGL(k_idx_scan, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, h->stream, (int)P, (int)idx_waves, d_hist, d_cnt);
      TRYH(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(long long) * P, hipMemcpyDeviceToHost, h->stream));
      TRYH(hipStreamSynchronize(h->stream));
    } else
    on_threads([&](int tid) {
      for (long long a = 0; a < N; ++a) {
        const int *cb = p->cam_idx + p->pt_ptr[a];
        const int d = (int)(p->pt_ptr[a + 1] - p->pt_ptr[a]);
        for (int i = 0; i < d; ++i) {
          if (cb[i] % n_thr != tid) continue;
          long long *row = cnt.data() + pair_id(cb[i], cb[i]) - cb[i];  // row[l] = cnt[pair(k, l)]
          for (int j = i; j < d; ++j) row[cb[j]]++;
        }
      }
    });
    lap("pair counts");
    long long T = 0, Tdiag = 0;
    for (int k = 0; k < m; ++k) Tdiag += cnt[pair_id(k, k)];
    for (long long q = 0; q < P; ++q) T += cnt[q];
    if (T >= (1LL << 40)) { mvba_destroy(h); return fail(MVBA_ERR_BADARG, "too many (point, camera pair) items"); }
    // a pair much larger than the typical off-diagonal one (the diagonal pairs: every observation of
    // the camera) is dealt round-robin into S sub-lists that sweep the points at the common pace
    const long long target = std::max<long long>(1, (T - Tdiag) / std::max<long long>(1, P - m));
    // items per unit (a wave's run): shorter units keep the sibling units of a strip closer together in time (more L2
    // hits on the k side) but cost a serial prologue and a 27-value tree each.  With one-wave blocks and static
    // assignment the best length is ~600 (config 3: 1.99 / 1.86 / 1.80 / 1.81 / 1.89 / 2.03 ms at 320 / 448 / 576 /
    // 640 / 768 / 1024; with four-wave blocks and atomic queues it was 768)
    // Round 4, re-swept where the unit form actually runs (beyond one round of the slot form): the sparser the pairs, the longer
    // the stretch of points a unit of U items spans (U / p^2) and the further its siblings drift apart -- config 4's shard (5 %):
    // 14.40 / 13.89 / 13.99 / 14.49 / 15.6 ms at 600 / 400 / 300 / 250 / 200 (L2 misses 743 M -> 553 M at 300, the per-unit
    // prologue and tree eat the rest); 1 M x 200 x 10 %: 6.21 / 6.61 / 7.12 at 600 / 400 / 300; 300 k x 300 x 5 %: 1.37 / 1.23 / 1.23
    // (profiles/r04_sweep_pairs_unit.txt).  So: 600 at one item per pair per 100 points, 400 at one per 400.
    // (Two gathers in flight -- this form on the slot kernel's ring loop -- make it SLOWER, 15.5 ms: the wider window of
    // points misses L2 more often, profiles/r04_sweep_pairs_ring.txt.)
    const double pair_rate = N > 0 && P > m ? (double)(T - Tdiag) / ((double)(P - m) * (double)N) : 0.01;
    long long unit_items = (long long)std::max(300.0, std::min(600.0, 200.0 + 4000.0 * std::sqrt(pair_rate)));
    if (const char *ev = getenv("MVBA_PAIR_UNIT")) unit_items = std::max(21, atoi(ev));
    std::vector<int> S(P), vp_ptr(P + 1, 0);
    // (MVBA_SLOT_DIAG_SCALE: the sub-lists of a pair much larger than the target -- the diagonal pairs -- are cut shorter by
    // this factor: a diagonal step costs more instructions than an off-diagonal one, and the pace of a range is its slowest wave's)
    double big_scale = 1.0;
    if (const char *ev = getenv("MVBA_SLOT_DIAG_SCALE")) big_scale = std::max(0.25, std::min(4.0, atof(ev)));
    for (long long q = 0; q < P; ++q) {
      S[q] = (int)std::max<long long>(1, std::min<long long>(256, (cnt[q] + target / 2) / target));
      if (S[q] >= 2 && big_scale != 1.0) S[q] = (int)std::max(1.0, std::min(256.0, std::floor((double)cnt[q] / (double)target * big_scale + 0.5)));
      vp_ptr[q + 1] = vp_ptr[q] + S[q];
    }
    const int VP = vp_ptr[P];
    // ---- which form of the kernel: the slot-resident one (k_schur_slots) needs all lists that sweep a point range
    // TOGETHER resident on one XCD at once -- 9 waves per CU (LDS) x n_cu / 8 CUs x 21 slots = 6048 lists.  Up to ~100
    // cameras at 10 % visibility (4950 pairs + ~1000 sub-lists of the diagonal pairs) that is every list: one ROUND.
    // Beyond that the cameras are cut into `ng` groups and a round holds the pairs of one group pair (g1 <= g2): its
    // waves sweep the range's points together and touch only the records of the two groups' cameras, so the footprint
    // per round still fits the L2; the rounds of a range follow each other inside ONE launch (the blocks of round
    // r + 1 start as the waves of round r finish).  A record is then read once per round its camera's group is in
    // (ng + 1 times) instead of once per unit that needs it (the unit form: ~16 times from beyond the L2 at m = 500).
    int n_cu_dev = 256;
    hipDeviceGetAttribute(&n_cu_dev, hipDeviceAttributeMultiprocessorCount, h->device);
#if defined(MVBA_HREC_TIMING)
    const int xcd_waves = std::max(1, n_cu_dev / 8) * std::min(160 * 1024 / SLOT_LDS, 4 * MVBA_SLOT_WAVES_PER_SIMD);
#else
    const int xcd_waves = std::max(1, n_cu_dev / 8) * (160 * 1024 / SLOT_LDS);  // 9 waves of 17,136 B of LDS per CU
#endif
    int ng = 1, G = m, max_round_waves = 0;
    auto round_waves = [&](int G_, int g1, int g2) {  // waves (of 21 lists) of round (g1, g2): diagonal + off-diagonal
      long long ld = 0, lo = 0;
      for (int k = g1 * G_; k < std::min(m, (g1 + 1) * G_); ++k) {
        if (g1 == g2) ld += S[pair_id(k, k)];
        for (int l = std::max(k + 1, g2 * G_); l < std::min(m, (g2 + 1) * G_); ++l) lo += S[pair_id(k, l)];
      }
      return (int)((ld + PSTEP - 1) / PSTEP + (lo + PSTEP - 1) / PSTEP);
    };
    if (const char *ev = getenv("MVBA_SLOT_GROUPS")) ng = std::max(1, std::min(m, atoi(ev)));
    for (;; ++ng) {
      G = (m + ng - 1) / ng;
      max_round_waves = 0;
      for (int g1 = 0; g1 * G < m; ++g1)
        for (int g2 = g1; g2 * G < m; ++g2) max_round_waves = std::max(max_round_waves, round_waves(G, g1, g2));
      if (max_round_waves <= xcd_waves || G == 1) break;
    }
    ng = (m + G - 1) / G;  // (groups that hold a camera)
    // (below ~4 M items the launch is all prologue and pacing: the unit form's many short waves win -- config 2,
    // 10k points x 20 cameras: 0.095 against 0.124 ms; equal at 5.5 M items; MVBA_SCHUR=slots keeps the slot form)
    const bool slots_forced = getenv("MVBA_SCHUR") && !strcmp(getenv("MVBA_SCHUR"), "slots");
    // (the gathers use 32-bit byte offsets: point rows from the array's start, records from their RANGE's first
    // observation -- checked below, once the ranges are known)
    // More than one round pays only while a list keeps enough items per L2 window for its wave's 21 lists to march in
    // step: at config 4 (500 cameras, 5 %: 390 items per list and range, ~10 per L2 window) the rounds cut the fabric
    // traffic 2.3x (742 M -> 328 M lines per launch) and still lose to the unit form, 18.1 against 14.3 ms -- 19-40 % padding
    // rows and the pacing waits of 250 waves on lists that sparse (profiles/r04_sweep_c4_rounds.txt).  MVBA_SCHUR=slots forces them.
    long long min_round_list = 1LL << 40;  // items per list and range, thinnest round
    if (ng > 1) {
      long long lists_total = 0;
      for (long long q = 0; q < P; ++q) lists_total += S[q];
      min_round_list = T / std::max<long long>(1, lists_total) / 8;
    }
    // ... and with denser lists as well (1 M points x 200 cameras x 10 %, 1250 items per list and range, three rounds: 8.9
    // against 6.3 ms; 2 M x 150 x 10 %: 8.6 against 7.4; 1 M x 300 x 5 %: 5.3 against 4.3 -- profiles/r04_sweep_rounds_crossover.txt):
    // more than one round runs only on request (MVBA_SCHUR=slots, or MVBA_SLOT_ROUND_MIN=<items per list and range>).
    long long round_min_items = 1LL << 40;
    if (const char *ev = getenv("MVBA_SLOT_ROUND_MIN")) round_min_items = std::max(0LL, atoll(ev));
    if (h->schur_mode == SCHUR_SLOTS && (max_round_waves > xcd_waves || (N + 1) * 128LL >= (1LL << 32) || h->force_big || ng > 2047 ||
                                         ((T < 4000000 || (ng > 1 && min_round_list < round_min_items)) && !slots_forced)))
      h->schur_mode = SCHUR_PAIRS;
    // point ranges.  Unit form: long runs for big problems, but small ones still get ~4096 units of >= 128 items.
    // Slot form: 8 ranges (one per XCD) -- 8 j while j ranges' worth of waves fit an XCD and a list keeps >= 64 items.
    int nR = 1;
    std::vector<long long> range_lo;
    auto make_ranges = [&]() {
      if (h->schur_mode == SCHUR_SLOTS) {
        const long long j = std::max<long long>(1, std::min<long long>(xcd_waves / std::max(1, max_round_waves), target / (8 * 64)));
        nR = (int)(8 * std::min<long long>(j, 8));
        range_lo.assign(nR + 1, 0);
        // equal ITEM counts: the ranges run side by side, one per XCD
        std::vector<long long> pre(N + 1, 0);
        for (long long a = 0; a < N; ++a) {
          const long long d = p->pt_ptr[a + 1] - p->pt_ptr[a];
          pre[a + 1] = pre[a] + d * (d + 1) / 2;
        }
        for (int r = 0; r <= nR; ++r)
          range_lo[r] = std::lower_bound(pre.begin(), pre.end(), (long long)((__int128)pre[N] * r / nR)) - pre.begin();
        range_lo[0] = 0; range_lo[nR] = N;
      } else {
        const long long nR_big = (target + unit_items / 2) / unit_items, nR_fill = std::min<long long>((4096 + VP - 1) / VP, target / 128);
        nR = (int)std::max<long long>(1, std::min<long long>(64, std::max(nR_big, nR_fill)));
        range_lo.assign(nR + 1, 0);
        for (int r = 0; r <= nR; ++r) range_lo[r] = (long long)((__int128)N * r / nR);
      }
    };
    make_ranges();
    // (few cameras with dense visibility: a dozen cameras are 78 lists = 5 waves per range, 320 waves on the whole chip even with 64
    // ranges -- 1 M points x 12 cameras, all visible: 9.3 ms against 4.9 for the unit form; at 20 cameras, 704 waves, the slot form is
    // ahead again, 9.8 against 10.8: profiles/r05_sweep_few_cameras.txt)
    if (h->schur_mode == SCHUR_SLOTS && !slots_forced && (long long)nR * max_round_waves < 512) {
      h->schur_mode = SCHUR_PAIRS;
      make_ranges();
    }
    if (h->schur_mode == SCHUR_SLOTS) {
      long long widest = 0;
      for (int r = 0; r < nR; ++r) widest = std::max<long long>(widest, p->pt_ptr[range_lo[r + 1]] - p->pt_ptr[range_lo[r]]);
      if ((widest + 1) * 128LL >= (1LL << 32)) {  // a range's records span 4 GiB: the unit form's 64-bit-offset build
        h->schur_mode = SCHUR_PAIRS;
        make_ranges();
      }
    }
    const bool slots = h->schur_mode == SCHUR_SLOTS;
    // ---- sweep order of the points inside a range (slot form).  The 21 lists of a wave march through the range in step
    // and a slot whose next item lies beyond the skew window idles: with the points in their natural (random) order a
    // list is a Poisson process -- 12.4 % padding rows at config 3, and the waves' steps per pacing segment scatter as
    // widely, which is what they wait for at the crossings.  Any order is as good for the kernel (a record is a line of
    // its own, the sums are per slot), so the index is built over a LOW-DISCREPANCY order: inside blocks of 8192 points
    // the next point is the best of `cand` random candidates by the summed deficit of its pairs (expected minus actual
    // count so far) -- the variance / mean of a pair's count per 2000-point window falls from 0.9 to ~0.25.  Deterministic
    // (fixed seeds), host threads by block.  order[i] = point at sweep position i, rank = its inverse, pkey[a] =
    // observations of the points swept before a, counted from the scene's start like a record index.
    std::vector<int> order, rank;
    std::vector<long long> pkey(N, 0);
    {
      const char *po = getenv("MVBA_POINT_ORDER");
      int cand = 32;
      if (const char *ev = getenv("MVBA_POINT_ORDER_CAND")) cand = std::max(1, atoi(ev));
      // MEASURED at config 3 (profiles/r04_sweep_point_order.txt): padding rows 12.4 % -> 10.4 % (8 / 32 / 64 candidates alike),
      // k_schur_slots 1.691 -> 1.677 ms, mvba_create 0.05 -> 0.17 s: the lists' unequal LENGTHS and the pacing, not their
      // local irregularity, are what is left -- so the natural order stays the default and MVBA_POINT_ORDER=greedy asks for this one.
      const bool reorder = slots && N > 0 && po && !strcmp(po, "greedy");
      if (reorder) {
        order.resize(N); rank.resize(N);
        constexpr long long OB = 8192;
        std::vector<std::pair<long long, long long>> blocks;
        for (int r = 0; r < nR; ++r)
          for (long long b0 = range_lo[r]; b0 < range_lo[r + 1]; b0 += OB) blocks.push_back({b0, std::min(range_lo[r + 1], b0 + OB)});
        std::atomic<size_t> next{0};
        const int nt = (int)std::max(1u, std::min({std::thread::hardware_concurrency(), 32u, (unsigned)blocks.size()}));
        auto work = [&]() {
          std::vector<int> count(P);
          std::vector<double> R;
          for (;;) {
            const size_t bi = next.fetch_add(1);
            if (bi >= blocks.size()) break;
            const long long b0 = blocks[bi].first, b1 = blocks[bi].second, nb = b1 - b0;
            std::fill(count.begin(), count.end(), 0);
            R.assign(nb, 0.0);
            int *ord = order.data() + b0;
            for (long long i = 0; i < nb; ++i) {
              ord[i] = (int)(b0 + i);
              const int *cb = p->cam_idx + p->pt_ptr[b0 + i];
              const int d = (int)(p->pt_ptr[b0 + i + 1] - p->pt_ptr[b0 + i]);
              double rs = 0.0;
              for (int x = 0; x < d; ++x)
                for (int y = x; y < d; ++y) rs += (double)cnt[pair_id(cb[x], cb[y])];
              R[i] = rs / (double)N;  // expected arrivals of this point's pairs per point swept
            }
            unsigned long long rng = 0x9E3779B97F4A7C15ull * (bi + 1);
            for (long long t = 0; t < nb; ++t) {
              long long best = t;
              double best_s = -1e300;
              for (int c = 0; c < cand; ++c) {
                rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
                const long long j = t + (long long)(rng % (unsigned long long)(nb - t));
                const long long a = ord[j];
                const int *cb = p->cam_idx + p->pt_ptr[a];
                const int d = (int)(p->pt_ptr[a + 1] - p->pt_ptr[a]);
                long long have = 0;
                for (int x = 0; x < d; ++x) {
                  const int *row = count.data() + pair_id(cb[x], cb[x]) - cb[x];
                  for (int y = x; y < d; ++y) have += row[cb[y]];
                }
                const double sc = (double)t * R[a - b0] - (double)have;
                if (sc > best_s) { best_s = sc; best = j; }
              }
              std::swap(ord[t], ord[best]);
              const long long a = ord[t];
              const int *cb = p->cam_idx + p->pt_ptr[a];
              const int d = (int)(p->pt_ptr[a + 1] - p->pt_ptr[a]);
              for (int x = 0; x < d; ++x) {
                int *row = count.data() + pair_id(cb[x], cb[x]) - cb[x];
                for (int y = x; y < d; ++y) row[cb[y]]++;
              }
            }
          }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
        for (long long i = 0; i < N; ++i) rank[order[i]] = (int)i;
      }
      long long run = 0;
      for (long long i = 0; i < N; ++i) {
        const long long a = order.empty() ? i : order[i];
        pkey[a] = run;
        run += p->pt_ptr[a + 1] - p->pt_ptr[a];
      }
    }
    lap("point order");
    int *d_order = nullptr, *d_rank = nullptr;
    long long *d_pkey = nullptr;
    if (dev_build && slots) {
      TRY(dmalloc(&d_pkey, (size_t)N));
      TRYH(hipMemcpyAsync(d_pkey, pkey.data(), sizeof(long long) * N, hipMemcpyHostToDevice, h->stream));
      if (!order.empty()) {
        TRY(dmalloc(&d_order, (size_t)N)); TRY(dmalloc(&d_rank, (size_t)N));
        TRYH(hipMemcpyAsync(d_order, order.data(), sizeof(int) * N, hipMemcpyHostToDevice, h->stream));
        TRYH(hipMemcpyAsync(d_rank, rank.data(), sizeof(int) * N, hipMemcpyHostToDevice, h->stream));
        // the per-wave start ranks again, in sweep order (the totals are the same)
        if (!hist_lds) TRYH(hipMemsetAsync(d_hist, 0, sizeof(int) * (size_t)idx_waves * P, h->stream));
        hipLaunchKernelGGL(hist_lds ? k_idx_count<false> : k_idx_count<true>, dim3((unsigned)(idx_waves / IDX_WAVES)), dim3(64 * IDX_WAVES), idx_lds,
                           h->stream, N, m, (int)P, h->d_pt_ptr, h->d_cam, idx_chunk, d_hist, d_order);
        hipLaunchKernelGGL(k_idx_scan, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, h->stream, (int)P, (int)idx_waves, d_hist, d_cnt);
      }
    }
    std::vector<long long> vp_off(VP + 1, 0);
    for (long long q = 0; q < P; ++q)
      for (int sI = 0; sI < S[q]; ++sI) vp_off[vp_ptr[q] + sI + 1] = (cnt[q] - sI + S[q] - 1) / S[q];
    for (int v = 0; v < VP; ++v) vp_off[v + 1] += vp_off[v];
    // the window-equalised merge (MVBA_SLOT_WINDOW, an experiment) exists on the host only
    const bool dev_items = dev_build && (!slots || h->slot_window >= (1LL << 39));
    int *d_pk = nullptr, *d_pl = nullptr, *d_pa = nullptr, *d_S = nullptr, *d_vp_ptr = nullptr;
    long long *d_vp_off = nullptr;
    auto free_dev_tmp = [&]() {
      for (void *q : {(void *)d_hist, (void *)d_cnt, (void *)d_pk, (void *)d_pl, (void *)d_pa, (void *)d_S, (void *)d_vp_ptr, (void *)d_vp_off,
                      (void *)d_order, (void *)d_rank, (void *)d_pkey})
        if (q) hipFree(q);
      d_hist = nullptr; d_cnt = nullptr; d_pk = d_pl = d_pa = d_S = d_vp_ptr = d_order = d_rank = nullptr; d_vp_off = d_pkey = nullptr;
    };
    if (dev_items) {
      TRY(dmalloc(&d_pk, (size_t)T)); TRY(dmalloc(&d_pl, (size_t)T)); TRY(dmalloc(&d_pa, (size_t)T));
      TRY(dmalloc(&d_S, (size_t)P)); TRY(dmalloc(&d_vp_ptr, (size_t)P + 1)); TRY(dmalloc(&d_vp_off, (size_t)VP + 1));
      TRYH(hipMemcpyAsync(d_S, S.data(), sizeof(int) * P, hipMemcpyHostToDevice, h->stream));
      TRYH(hipMemcpyAsync(d_vp_ptr, vp_ptr.data(), sizeof(int) * (P + 1), hipMemcpyHostToDevice, h->stream));
      TRYH(hipMemcpyAsync(d_vp_off, vp_off.data(), sizeof(long long) * (VP + 1), hipMemcpyHostToDevice, h->stream));
      hipLaunchKernelGGL(hist_lds ? k_idx_fill<false> : k_idx_fill<true>, dim3((unsigned)(idx_waves / IDX_WAVES)), dim3(64 * IDX_WAVES), idx_lds,
                         h->stream, N, m, (int)P, h->d_pt_ptr, h->d_cam, idx_chunk, d_hist, d_S, d_vp_ptr, d_vp_off, d_pk, d_pl, d_pa,
                         (const int *)d_order);
      TRYH(hipGetLastError());
    } else {
      if (dev_build) { hipFree(d_hist); hipFree(d_cnt); d_hist = nullptr; d_cnt = nullptr; }
    it_k.resize(T); it_l.resize(T); it_a.resize(T);
    {
      std::vector<long long> run(P, 0);
      on_threads([&](int tid) {
        for (long long ai = 0; ai < N; ++ai) {
          const long long a = order.empty() ? ai : order[ai];
          const long long o0 = p->pt_ptr[a];
          const int *cb = p->cam_idx + o0;
          const int d = (int)(p->pt_ptr[a + 1] - o0);
          for (int i = 0; i < d; ++i) {
            if (cb[i] % n_thr != tid) continue;
            const long long rowp = pair_id(cb[i], cb[i]) - cb[i];
            for (int j = i; j < d; ++j) {
              const long long q = rowp + cb[j], r = run[q]++;
              const int sI = (int)(r % S[q]);
              const long long pos = vp_off[vp_ptr[q] + sI] + r / S[q];
              it_k[pos] = (int)(o0 + i); it_l[pos] = (int)(o0 + j); it_a[pos] = (int)a;
            }
          }
        }
      });
    }
    }
    lap("items sorted by pair");
    // units: (pair, sub-list, point range), numbered pair-major (k_schur_reduce sums them in this order)
    unit_ptr.assign(P + 1, 0);
    std::vector<int> uid((size_t)VP * nR, -1);
    std::vector<long long> lo_tab;  // device build: lower bounds of every list at every range boundary
    if (dev_items) {
      long long *d_rl = nullptr, *d_lo = nullptr;
      TRY(dmalloc(&d_rl, (size_t)nR + 1)); TRY(dmalloc(&d_lo, (size_t)VP * (nR + 1)));
      TRYH(hipMemcpyAsync(d_rl, range_lo.data(), sizeof(long long) * (nR + 1), hipMemcpyHostToDevice, h->stream));
      const long long nt = (long long)VP * (nR + 1);
      hipLaunchKernelGGL(k_idx_bounds, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->stream, VP, nR, d_vp_off, d_rl, d_pa, d_lo, (const int *)d_rank);
      lo_tab.resize(nt);
      TRYH(hipMemcpyAsync(lo_tab.data(), d_lo, sizeof(long long) * nt, hipMemcpyDeviceToHost, h->stream));
      TRYH(hipStreamSynchronize(h->stream));
      hipFree(d_rl); hipFree(d_lo);
    }
    for (int k = 0; k < m; ++k)
      for (int l = k; l < m; ++l) {
        const long long q = pair_id(k, l);
        unit_ptr[q] = (int)units.size();
        for (int sI = 0; sI < S[q]; ++sI) {
          const int v = vp_ptr[q] + sI;
          const int *b = it_a.data() + (dev_items ? 0 : vp_off[v]), *e = it_a.data() + (dev_items ? 0 : vp_off[v + 1]);
          for (int r = 0; r < nR; ++r) {
            auto before = [&](int a, long long key) { return (rank.empty() ? (long long)a : (long long)rank[a]) < key; };
            const long long lo = dev_items ? lo_tab[(size_t)v * (nR + 1) + r] : std::lower_bound(b, e, range_lo[r], before) - it_a.data();
            const long long hi = dev_items ? lo_tab[(size_t)v * (nR + 1) + r + 1] : std::lower_bound(b, e, range_lo[r + 1], before) - it_a.data();
            if (hi <= lo) continue;
            uid[(size_t)v * nR + r] = (int)units.size();
            units.push_back(make_int4((int)(lo & 0xffffffffLL), (int)(lo >> 32), (int)(hi - lo), (k << 16) | l));
          }
        }
      }
    unit_ptr[P] = (int)units.size();
    lap("units");
    if (slots) {
      // ---- waves of 21 lists, every wave once per range, round by round (one round up to ~100 cameras); inside a round
      // the diagonal pairs' sub-lists come FIRST: a CU's SIMDs arbitrate by age, the blocks dispatched last share a
      // SIMD three ways as its youngest wave and fall behind -- and a diagonal step is the dearer one
      std::vector<int> wl;                    // [wave of a range][21] list ids v = vp_ptr[pair] + sub-list, -1: none
      std::vector<int> w_round, w_isdiag;     // per wave of a range
      int n_rounds = 0;
      for (int g1 = 0; g1 * G < m; ++g1)
        for (int g2 = g1; g2 * G < m; ++g2, ++n_rounds) {
          std::vector<int> ld, lo;
          for (int k = g1 * G; k < std::min(m, (g1 + 1) * G); ++k) {
            if (g1 == g2)
              for (int sI = 0; sI < S[pair_id(k, k)]; ++sI) ld.push_back(vp_ptr[pair_id(k, k)] + sI);
            for (int l = std::max(k + 1, g2 * G); l < std::min(m, (g2 + 1) * G); ++l)
              for (int sI = 0; sI < S[pair_id(k, l)]; ++sI) lo.push_back(vp_ptr[pair_id(k, l)] + sI);
          }
          for (const std::vector<int> *src : {&ld, &lo})
            for (size_t first = 0; first < src->size(); first += PSTEP) {
              for (int sl = 0; sl < PSTEP; ++sl) wl.push_back(first + sl < src->size() ? (*src)[first + sl] : -1);
              w_round.push_back(n_rounds);
              w_isdiag.push_back(src == &ld);
            }
        }
      const int wpr = (int)w_round.size();    // waves per range
      const long long n_waves = (long long)wpr * nR;
      wdesc.assign(n_waves, make_int4(0, 0, 0, 0));
      wunits.assign((size_t)n_waves * PSTEP, -1);
      std::vector<long long> w_steps(n_waves, 0), w_beg(n_waves + 1, 0);
      // block b = nR w + r: wave w of range r runs on XCD r % 8
      auto wave_lists = [&](long long b, int *vs) {  // the 21 list ids of block b (-1: none); returns the range
        const int *src = wl.data() + (size_t)(b / nR) * PSTEP;
        for (int sl = 0; sl < PSTEP; ++sl) vs[sl] = src[sl];
        return (int)(b % nR);
      };
      auto round_range = [&](long long b) { return (size_t)w_round[b / nR] * nR + (size_t)(b % nR); };
      // a round touches the records of 2 of ng camera groups only: its skew and pacing segments, counted in
      // observations of the range, stretch accordingly (the footprint in the L2 is what they bound)
      const long long stretch = std::max(1, ng / 2);
      // Bounded-skew merge of a wave's lists into steps (see k_schur_slots), window by window: the observations of a
      // range are cut into windows of `slot_window`, and every wave of the range is padded to the same number of
      // steps per window (the slowest wave's), so that all waves of an XCD reach a window boundary at the same step
      // index and cannot drift apart by more than their rate difference inside one window.
      const long long skew = h->slot_skew * stretch, window = std::max<long long>(1, h->slot_window);
      int nWin = 1;
      for (int r = 0; r < nR; ++r)
        nWin = std::max<long long>(nWin, (p->pt_ptr[range_lo[r + 1]] - p->pt_ptr[range_lo[r]] + window - 1) / window);
      const bool equalize = h->slot_window < (1LL << 39);
      std::vector<int> win_steps((size_t)n_waves * nWin, 0), win_max((size_t)n_rounds * nR * nWin, 0);
      // pacing segments: seg_end[b][j] = steps wave b has taken when its slowest slot leaves segment j of the range
      // (no pacing, MVBA_SLOT_SEG=0: one segment -- the table has a row per wave and segment)
      const long long segG = h->slot_pace ? std::max<long long>(1, h->slot_seg * stretch) : (1LL << 40);
      int nSeg = 1;
      for (int r = 0; r < nR; ++r)
        nSeg = std::max<long long>(nSeg, (p->pt_ptr[range_lo[r + 1]] - p->pt_ptr[range_lo[r]] + segG - 1) / segG);
      seg_end.assign((size_t)n_waves * nSeg, 0);
      h->slot_nseg = nSeg;
      auto merge = [&](long long b, long long base, bool fill) {
        int vs[PSTEP];
        const int r = wave_lists(b, vs);
        long long cur[PSTEP], end[PSTEP];
        for (int sl = 0; sl < PSTEP; ++sl) {
          const int id = vs[sl] >= 0 ? uid[(size_t)vs[sl] * nR + r] : -1;
          if (id < 0) { cur[sl] = end[sl] = 0; continue; }
          cur[sl] = ((long long)units[id].y << 32) | (unsigned)units[id].x;
          end[sl] = cur[sl] + units[id].z;
          if (fill) wunits[(size_t)b * PSTEP + sl] = id;
        }
        long long steps = 0;
        const long long o_lo = p->pt_ptr[range_lo[r]];
        int sg = 0;
        for (int j = 0; j < nWin; ++j) {
          const long long limit = j + 1 < nWin ? o_lo + (j + 1) * window : (1LL << 62);
          long long ws = 0;
          while (true) {
            long long lo = -1;
            auto key_of = [&](int sl) { return pkey[it_a[cur[sl]]]; };  // where the item's point sits in the sweep, in observations
            for (int sl = 0; sl < PSTEP; ++sl)
              if (cur[sl] < end[sl] && key_of(sl) < limit && (lo < 0 || key_of(sl) < lo)) lo = key_of(sl);
            if (fill && lo >= 0)
              while (sg < nSeg && lo >= o_lo + (sg + 1) * segG) seg_end[(size_t)b * nSeg + sg++] = (int)(steps + ws);
            if (lo < 0) break;
            for (int sl = 0; sl < PSTEP; ++sl) {
              const bool take = cur[sl] < end[sl] && key_of(sl) < limit && key_of(sl) <= lo + skew;
              if (fill) {
                const long long o = (base + steps + ws) * PSTEP + sl;
                if (take) { st_k[o] = (int)(it_k[cur[sl]] - o_lo); st_l[o] = (int)(it_l[cur[sl]] - o_lo); st_a[o] = it_a[cur[sl]]; }
                else { st_k[o] = st_l[o] = 0; st_a[o] = (int)N; }  // the range's first record (any finite one) x the all-zero point row
              }
              if (take) ++cur[sl];
            }
            ++ws;
          }
          if (!fill) { win_steps[(size_t)b * nWin + j] = (int)ws; steps += ws; continue; }
          const long long target_ws = equalize ? win_max[round_range(b) * nWin + j] : ws;
          for (; ws < target_ws; ++ws)
            for (int sl = 0; sl < PSTEP; ++sl) {
              const long long o = (base + steps + ws) * PSTEP + sl;
              st_k[o] = st_l[o] = 0; st_a[o] = (int)N;
            }
          steps += target_ws;
        }
        if (fill)
          while (sg < nSeg) seg_end[(size_t)b * nSeg + sg++] = (int)steps;
        return steps;
      };
      // device merge: the slots' list spans (from the units) go up, the step counts come back
      long long *d_slbeg = nullptr, *d_ro0 = nullptr, *d_wbeg = nullptr;
      int *d_sllen = nullptr, *d_wsteps = nullptr;
      {  // first observation of every range: the merge kernels and k_schur_slots (its record base) read it
        std::vector<long long> ro0(nR);
        for (int r = 0; r < nR; ++r) ro0[r] = p->pt_ptr[range_lo[r]];
        TRY(dmalloc(&h->d_range_o0, (size_t)nR));
        TRYH(hipMemcpy(h->d_range_o0, ro0.data(), sizeof(long long) * nR, hipMemcpyHostToDevice));
        d_ro0 = h->d_range_o0;
      }
      if (dev_items) {
        std::vector<long long> sl_beg((size_t)n_waves * PSTEP, 0);
        std::vector<int> sl_len((size_t)n_waves * PSTEP, 0);
        for (long long b = 0; b < n_waves; ++b) {
          int vs[PSTEP];
          const int r = wave_lists(b, vs);
          for (int sl = 0; sl < PSTEP; ++sl) {
            const int id = vs[sl] >= 0 ? uid[(size_t)vs[sl] * nR + r] : -1;
            wunits[(size_t)b * PSTEP + sl] = id;
            if (id < 0) continue;
            sl_beg[(size_t)b * PSTEP + sl] = ((long long)units[id].y << 32) | (unsigned)units[id].x;
            sl_len[(size_t)b * PSTEP + sl] = units[id].z;
          }
        }
        TRY(dmalloc(&d_slbeg, sl_beg.size())); TRY(dmalloc(&d_sllen, sl_len.size()));
        TRY(dmalloc(&d_wsteps, (size_t)n_waves)); TRY(dmalloc(&d_wbeg, (size_t)n_waves + 1));
        TRYH(hipMemcpyAsync(d_slbeg, sl_beg.data(), sizeof(long long) * sl_beg.size(), hipMemcpyHostToDevice, h->stream));
        TRYH(hipMemcpyAsync(d_sllen, sl_len.data(), sizeof(int) * sl_len.size(), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_idx_merge<false>, dim3((unsigned)n_waves), dim3(64), 0, h->stream, n_waves, nR, nSeg, skew, segG, d_slbeg, d_sllen,
                           d_ro0, d_pk, d_pl, d_pa, d_wbeg, 0, (int)N, d_wsteps, (int *)nullptr, (int *)nullptr, (int *)nullptr,
                           (int *)nullptr, (const long long *)d_pkey);
        std::vector<int> ws32(n_waves);
        TRYH(hipMemcpyAsync(ws32.data(), d_wsteps, sizeof(int) * n_waves, hipMemcpyDeviceToHost, h->stream));
        TRYH(hipStreamSynchronize(h->stream));  // (sl_beg / sl_len / ro0 live until here)
        for (long long b = 0; b < n_waves; ++b) win_steps[(size_t)b * nWin] = ws32[b];
      } else
      on_threads([&](int tid) {
        for (long long b = tid; b < n_waves; b += n_thr) merge(b, 0, false);
      });
      lap("slot merge (count)");
      for (long long b = 0; b < n_waves; ++b)
        for (int j = 0; j < nWin; ++j) {
          int &mx = win_max[round_range(b) * nWin + j];
          mx = std::max(mx, win_steps[(size_t)b * nWin + j]);
        }
      for (long long b = 0; b < n_waves; ++b) {
        long long t = 0;
        bool any = false;
        for (int j = 0; j < nWin; ++j) {
          t += equalize ? win_max[round_range(b) * nWin + j] : win_steps[(size_t)b * nWin + j];
          any |= win_steps[(size_t)b * nWin + j] > 0;
        }
        w_steps[b] = any ? t : 0;  // a wave without any item does not run at all
      }
      for (long long b = 0; b < n_waves; ++b) w_beg[b + 1] = w_beg[b] + w_steps[b];
      const long long total_steps = w_beg[n_waves];
      if (total_steps * PSTEP >= (1LL << 40)) { mvba_destroy(h); return fail(MVBA_ERR_BADARG, "too many (point, camera pair) items"); }
      {  // the step-major index -- its size follows the opt-in knobs (skew, window, groups: padding rows) -- against the memory that is
        // there, BEFORE anything of it is allocated: three 4-byte arrays of step rows, then the interleaved 256-byte rows beside them
        const size_t need = (size_t)total_steps * PSTEP * 12 + (size_t)total_steps * SLOT_IDX * 4 + seg_end.size() * 4;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && need > fr) {
          mvba_destroy(h);
          return fail(MVBA_ERR_BADARG, "the slot-form Schur index needs " + std::to_string(need >> 20) + " MiB (" + std::to_string(total_steps * PSTEP) + " step rows for " +
                                           std::to_string(T) + " items: " + std::to_string(n_rounds) + " rounds x " + std::to_string(nR) + " ranges, skew " + std::to_string(skew) +
                                           ", window " + std::to_string(h->slot_window) + (equalize ? " (equalised)" : "") + "), " + std::to_string(fr >> 20) +
                                           " MiB of device memory are free: relax MVBA_SLOT_SKEW / MVBA_SLOT_WINDOW / MVBA_SLOT_GROUPS or use MVBA_SCHUR=pairs");
        }
      }
      if (dev_items) {  // the step-major arrays are written where the kernel will read them
        const size_t rows = (size_t)total_steps * PSTEP;
        TRY(dmalloc(&h->d_it_k, rows)); TRY(dmalloc(&h->d_it_l, rows)); TRY(dmalloc(&h->d_it_a, rows));
        TRY(dmalloc(&h->d_seg_end, seg_end.size()));
        TRYH(hipMemcpyAsync(d_wbeg, w_beg.data(), sizeof(long long) * (n_waves + 1), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_idx_merge<true>, dim3((unsigned)n_waves), dim3(64), 0, h->stream, n_waves, nR, nSeg, skew, segG, d_slbeg, d_sllen,
                           d_ro0, d_pk, d_pl, d_pa, d_wbeg, 0, (int)N, d_wsteps, h->d_it_k, h->d_it_l, h->d_it_a, h->d_seg_end, (const long long *)d_pkey);
        TRYH(hipGetLastError());
        TRYH(hipStreamSynchronize(h->stream));
        for (void *q : {(void *)d_slbeg, (void *)d_sllen, (void *)d_wsteps, (void *)d_wbeg}) hipFree(q);
        free_dev_tmp();
        h->index_on_device = true;
      } else {
      st_k.resize(total_steps * PSTEP); st_l.resize(total_steps * PSTEP); st_a.resize(total_steps * PSTEP);
      on_threads([&](int tid) {
        for (long long b = tid; b < n_waves; b += n_thr)
          if (w_steps[b]) merge(b, w_beg[b], true);
      });
      }
      lap("slot merge (fill)");
      std::vector<int> live((size_t)n_rounds * nR, 0);  // waves of a (round, range) that run at all: what a pacing counter has to reach
      for (long long b = 0; b < n_waves; ++b) live[round_range(b)] += w_steps[b] > 0;
      for (long long b = 0; b < n_waves; ++b) {
        const long long beg = w_beg[b];  // first step of the wave in the step-major index
        // flags: bit 0 diagonal wave | bits 8..19 live waves of its (round, range) | bits 20..30 round
        wdesc[b] = make_int4((int)(beg & 0xffffffffLL), (int)(beg >> 32), (int)w_steps[b],
                             (w_isdiag[b / nR] ? 1 : 0) | (live[round_range(b)] << 8) | (w_round[b / nR] << 20));
      }
      h->slot_rounds = n_rounds;
      h->slot_groups = ng;
      h->n_waves = (int)n_waves;
      h->slot_nR = nR;
#ifdef MVBA_SLOT_TRACE
      if (const char *ev = getenv("MVBA_SLOT_DUMP"))  // diagnostic build: the pacing table (steps at each segment boundary)
        if (FILE *f = fopen(ev, "wb")) {
          const int hdr[4] = {(int)n_waves, nSeg, nR, wpr};  // (pacing is per (round, range): MVBA_SLOT_GROUPS=1 scenes for the replay tool)
          fwrite(hdr, sizeof(int), 4, f);
          fwrite(seg_end.data(), sizeof(int), seg_end.size(), f);
          fclose(f);
        }
#endif
      h->n_slot_items = total_steps * PSTEP;
      if (!dev_items) { it_k.swap(st_k); it_l.swap(st_l); it_a.swap(st_a); }  // what is uploaded below: the step-major arrays
      std::vector<int>().swap(st_k); std::vector<int>().swap(st_l); std::vector<int>().swap(st_a);
    } else {
    // work queues: strip k on XCD k % 8, inside a queue by (k, range, l, sub-list)
    // range-major: every XCD sweeps the point ranges in the same order, so the l-side records of
    // a range (needed once per strip, ~4.5 times in all) are re-read from the Infinity Cache while
    // the whole chip is on that range: 2.28 -> 2.04 ms at config 3 (MVBA_PAIR_ORDER=kr: strip-major)
    const bool range_major = !(getenv("MVBA_PAIR_ORDER") && !strcmp(getenv("MVBA_PAIR_ORDER"), "kr"));
    for (int x = 0; x < 8; ++x) {
      auto push_group = [&](int k, int r) {
        for (int l = k; l < m; ++l) {
          const long long q = pair_id(k, l);
          for (int sI = 0; sI < S[q]; ++sI) {
            const int id = uid[(size_t)(vp_ptr[q] + sI) * nR + r];
            if (id >= 0) q_units.push_back(id);
          }
        }
      };
      if (range_major) {
        for (int r = 0; r < nR; ++r)
          for (int k = x; k < m; k += 8) push_group(k, r);
      } else {
        for (int k = x; k < m; k += 8)
          for (int r = 0; r < nR; ++r) push_group(k, r);
      }
      q_ptr[x + 1] = (int)q_units.size();
    }
    if (dev_items) {  // the pair-major arrays stay where the fill kernel wrote them
      h->d_it_k = d_pk; h->d_it_l = d_pl; h->d_it_a = d_pa;
      d_pk = d_pl = d_pa = nullptr;
      free_dev_tmp();
      h->index_on_device = true;
    }
    }
    h->n_items = T;
    h->n_items_offdiag = T - Tdiag;
    h->n_units = (int)units.size();
  }
  lap("queues / wave descriptors");
  h->cost_grid = (int)std::max<long long>(1, std::min<long long>(2048, (nobs + 255) / 256));
  h->n_partials = std::max(h->cost_grid, 4096);  // k_cost uses cost_grid blocks

  trace.mark("K3 index");
  TRY(dmalloc(&h->d_obs_pt, nobs));
  TRY(dmalloc(&h->d_xy, nobs));
  TRY(dmalloc(&h->d_csc, csc.size()));
  h->n_tiles = (int)tiles.size() - 1;
  h->any_split = any_split;
  if (any_split) {
    tile_slot.resize(tiles.size(), -1);
    h->n_splits = (int)splits.size();
    TRY(dmalloc(&h->d_tile_slot, tile_slot.size()));
    TRY(dmalloc(&h->d_splits, splits.size()));
    TRY(dmalloc(&h->d_PLsplit, 9 * (size_t)n_split_slots));
    TRYH(hipMemcpy(h->d_tile_slot, tile_slot.data(), sizeof(int) * tile_slot.size(), hipMemcpyHostToDevice));
    TRYH(hipMemcpy(h->d_splits, splits.data(), sizeof(int4) * splits.size(), hipMemcpyHostToDevice));
  }
  TRY(dmalloc(&h->d_tiles, tiles.size()));
  TRY(dmalloc(&h->d_chunk_ptr, chunk_ptr.size()));
  for (int i = 0; i < 2; ++i) { TRY(dmalloc(&h->d_X[i], 3 * N)); TRY(dmalloc(&h->d_cam15[i], (size_t)CAM_IN * m)); }
  TRY(dmalloc(&h->d_rec, (size_t)REC * (nobs + 1)));  // + the all-zero record and point row the slot form's padding points at
  TRY(dmalloc(&h->d_PL, 9 * N));
  TRY(dmalloc(&h->d_PB, (size_t)PBS * (N + 1)));
  TRYH(hipMemset(h->d_rec + (size_t)REC * nobs, 0, sizeof(double2) * REC));
  TRYH(hipMemset(h->d_PB + (size_t)PBS * N, 0, sizeof(double) * PBS));
  const size_t n9 = 9 * (size_t)m;
  TRY(dmalloc(&h->d_Ab, strip_offset(m, m) + n9));
  TRY(dmalloc(&h->d_Ared, (size_t)(h->D + 1) * h->ld));
  TRY(dmalloc(&h->d_Lblk, (size_t)((h->D + SBW - 1) / SBW) * SBW * SBW));
  TRY(dmalloc(&h->d_Ztiles, (size_t)((h->D + NB - 1) / NB) * NB * NB));
  TRY(dmalloc(&h->d_dxi, n9));
  TRY(dmalloc(&h->d_dX, 3 * N));
  TRY(dmalloc(&h->d_partials, h->n_partials));
  TRY(dmalloc(&h->d_cost, 2));
  TRY(dmalloc(&h->d_flag, 1));
  TRY(dmalloc(&h->d_bar, 1 + 4 * (size_t)((9 * m + SBW - 1) / SBW)));  // barrier counter / progress words of the back-substitution
  TRYH(hipHostMalloc((void **)&h->h_cost, 4 * sizeof(double), hipHostMallocMapped));
  memset(h->h_cost, 0, 4 * sizeof(double));
  if (hipHostGetDevicePointer((void **)&h->d_mail, h->h_cost, 0) != hipSuccess) h->d_mail = nullptr;  // (no mapping: copy + sync as before)
  h->h_flag = reinterpret_cast<int *>(h->h_cost + 1);  // cost and flags come back in one copy
  trace.mark("allocations");
  if (nobs) {
    TRYH(hipMemcpy(h->d_obs_pt, obs_pt.data(), sizeof(int) * nobs, hipMemcpyHostToDevice));
    if (p->xy_layout == 1) {  // image planes [m][N][2], as a caller's stack of per-image arrays lies in memory: into observation order here
      double2 *planes = nullptr;  // (the host's strided gather of the same bytes: 0.1 s at 1 M points x 12 images)
      TRYH(hipMalloc((void **)&planes, sizeof(double2) * nobs));
      hipError_t e = hipMemcpy(planes, p->xy, sizeof(double2) * nobs, hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_xy_from_planes, dim3((unsigned)((N + 255) / 256), m), dim3(256), 0, 0, planes, N, m, h->d_xy);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
      }
      hipFree(planes);
      TRYH(e);
    } else {
      TRYH(hipMemcpy(h->d_xy, p->xy, sizeof(double2) * nobs, hipMemcpyHostToDevice));
    }
    if (want_strip) TRYH(hipMemcpy(h->d_csc, csc.data(), sizeof(int4) * nobs, hipMemcpyHostToDevice));
  }
  TRYH(hipMemcpy(h->d_tiles, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice));
  // points without observations are never written by K1: their blocks stay zero (-> singular, ref :128)
  TRYH(hipMemset(h->d_PL, 0, sizeof(double) * 9 * std::max<long long>(N, 1)));
  if (want_strip) TRYH(hipMemcpy(h->d_chunk_ptr, chunk_ptr.data(), sizeof(long long) * chunk_ptr.size(), hipMemcpyHostToDevice));
  TRYH(hipMemset(h->d_flag, 0, sizeof(int)));
  if (h->schur_mode == SCHUR_DENSE) {  // partial tiles of k_schur_dense: (tile pairs + one per camera) x 256 doubles per workgroup
    const int T = (9 * m + 15) / 16;
    int n_cu_dense = 256;
    hipDeviceGetAttribute(&n_cu_dense, hipDeviceAttributeMultiprocessorCount, h->device);
    h->dense_tiles = T * (T + 1) / 2 + m;
    h->dense_blocks = (int)std::max<long long>(1, std::min<long long>((N + dense_ch(T) - 1) / dense_ch(T), (long long)n_cu_dense * dense_wgs(T)));  // dense_wgs workgroups per CU (LDS and registers: see dense_ch)
    TRY(dmalloc(&h->d_dense_part, (size_t)h->dense_blocks * h->dense_tiles * 256));
    if (!dense_obs.empty()) {
      TRY(dmalloc(&h->d_dense_obs, dense_obs.size()));
      TRYH(hipMemcpy(h->d_dense_obs, dense_obs.data(), sizeof(int) * dense_obs.size(), hipMemcpyHostToDevice));
    }
  }
  if (h->use_pairs) {
    const size_t P1 = (size_t)m * (m + 1) / 2 + 1;
    if (!h->index_on_device) { TRY(dmalloc(&h->d_it_k, it_k.size())); TRY(dmalloc(&h->d_it_l, it_l.size())); TRY(dmalloc(&h->d_it_a, it_a.size())); }
    TRY(dmalloc(&h->d_units, units.size())); TRY(dmalloc(&h->d_unit_ptr, P1));
    TRY(dmalloc(&h->d_q_ptr, 9)); TRY(dmalloc(&h->d_q_units, q_units.size())); TRY(dmalloc(&h->d_q_head, 64));
    TRY(dmalloc(&h->d_wdesc, wdesc.size())); TRY(dmalloc(&h->d_wunits, wunits.size()));
#ifdef MVBA_SLOT_TRACE
    if (getenv("MVBA_SLOT_TRACE")) { TRY(dmalloc(&h->d_trace, 16 * std::max<size_t>(1, wdesc.size()))); TRYH(hipMemset(h->d_trace, 0, 128 * std::max<size_t>(1, wdesc.size()))); }
#endif
    if (!h->index_on_device) TRY(dmalloc(&h->d_seg_end, seg_end.size()));
    TRY(dmalloc(&h->d_prog, (size_t)h->slot_rounds * h->slot_nR * std::max(1, h->slot_nseg) * PACE_STRIDE));
    if (!seg_end.empty() && !h->index_on_device) TRYH(hipMemcpy(h->d_seg_end, seg_end.data(), sizeof(int) * seg_end.size(), hipMemcpyHostToDevice));
    if (!wdesc.empty()) {
      TRYH(hipMemcpy(h->d_wdesc, wdesc.data(), sizeof(int4) * wdesc.size(), hipMemcpyHostToDevice));
      TRYH(hipMemcpy(h->d_wunits, wunits.data(), sizeof(int) * wunits.size(), hipMemcpyHostToDevice));
    }
    TRY(dmalloc(&h->d_partial, (size_t)UNIT_STRIDE * units.size()));
    if (!it_k.empty()) {
      TRYH(hipMemcpy(h->d_it_k, it_k.data(), sizeof(int) * it_k.size(), hipMemcpyHostToDevice));
      TRYH(hipMemcpy(h->d_it_l, it_l.data(), sizeof(int) * it_l.size(), hipMemcpyHostToDevice));
      TRYH(hipMemcpy(h->d_it_a, it_a.data(), sizeof(int) * it_a.size(), hipMemcpyHostToDevice));
    }
    if (!it_k.empty() || h->index_on_device) {
      if (h->schur_mode == SCHUR_PAIRS) {
        std::vector<int4> qdesc(units.size());  // descriptors in queue order (the kernel indexes both arrays by queue position)
        for (size_t i = 0; i < q_units.size(); ++i) qdesc[i] = units[q_units[i]];
        TRYH(hipMemcpy(h->d_units, qdesc.data(), sizeof(int4) * qdesc.size(), hipMemcpyHostToDevice));
        int mx = 0;
        for (int x = 0; x < 8; ++x) mx = std::max(mx, q_ptr[x + 1] - q_ptr[x]);
        h->q_max = mx;
      }
      if (!q_units.empty()) TRYH(hipMemcpy(h->d_q_units, q_units.data(), sizeof(int) * q_units.size(), hipMemcpyHostToDevice));
    }
    if (h->schur_mode == SCHUR_SLOTS && h->n_slot_items) {  // the three step-major arrays -> one 256-byte row per step; they go
      const long long n_steps = h->n_slot_items / PSTEP;
      TRY(dmalloc(&h->d_it_x, (size_t)n_steps * SLOT_IDX));
      hipLaunchKernelGGL(k_idx_interleave, dim3((unsigned)((n_steps * SLOT_IDX + 255) / 256)), dim3(256), 0, h->stream, n_steps, h->d_it_k, h->d_it_l,
                         h->d_it_a, h->d_it_x);
      TRYH(hipGetLastError());
      TRYH(hipStreamSynchronize(h->stream));
      hipFree(h->d_it_k); hipFree(h->d_it_l); hipFree(h->d_it_a);
      h->d_it_k = h->d_it_l = h->d_it_a = nullptr;
    }
    TRYH(hipMemcpy(h->d_unit_ptr, unit_ptr.data(), sizeof(int) * P1, hipMemcpyHostToDevice));
    TRYH(hipMemcpy(h->d_q_ptr, q_ptr.data(), sizeof(int) * 9, hipMemcpyHostToDevice));
    TRYH(hipMemset(h->d_q_head, 0, sizeof(int) * 64));
    TRYH(hipMemset(h->d_partial, 0, sizeof(double) * UNIT_STRIDE * std::max<size_t>(units.size(), 1)));
  }
  trace.mark("uploads");
  // opt in to large dynamic LDS
  const int strip_lds = (int)((81 * (size_t)h->lseg + 9) * sizeof(double));
  TRYH(hipFuncSetAttribute((const void *)k_schur_strip<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, strip_lds));
  TRYH(hipFuncSetAttribute((const void *)k_schur_strip<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, strip_lds));
  TRYH(hipFuncSetAttribute((const void *)k_schur_strip<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, strip_lds));
  TRYH(hipFuncSetAttribute((const void *)k_schur_strip<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, strip_lds));
  const int cam_lds = h->gcam ? 0 : (int)((size_t)m * (CAM_LDS + DXI_LDS) * sizeof(double));
  if (h->gcam) { TRY(dmalloc(&h->d_cam18, (size_t)m * CAM_LDS)); TRY(dmalloc(&h->d_dxi10, (size_t)m * DXI_LDS)); }
  for (const void *f : {(const void *)k_backsub<2>, (const void *)k_backsub<4>, (const void *)k_backsub<8>, (const void *)k_backsub<2, 512>,
                        (const void *)k_backsub<4, 512>, (const void *)k_backsub<8, 512>, (const void *)k_backsub<2, 1024>,
                        (const void *)k_backsub<4, 1024>, (const void *)k_backsub<8, 1024>})
    TRYH(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cam_lds));
  TRYH(hipFuncSetAttribute((const void *)k_chol_super, hipFuncAttributeMaxDynamicSharedMemorySize, SUPER_LDS));
  TRYH(hipFuncSetAttribute((const void *)k_chol_backsolve_all<false>, hipFuncAttributeMaxDynamicSharedMemorySize, BACKSOLVE_LDS));
  TRYH(hipFuncSetAttribute((const void *)k_chol_backsolve_all<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BACKSOLVE_LDS));
  {
    // the persistent back-substitution needs its whole grid resident: at most one workgroup per CU
    int per_cu = 0;
    TRYH(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    TRYH(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_chol_backsolve_all<true>, SUPER_THREADS, BACKSOLVE_LDS));
    h->chol_onepass = per_cu >= 1 && !(getenv("MVBA_CHOL") && !strcmp(getenv("MVBA_CHOL"), "launches"));
    h->chol_flow = !(getenv("MVBA_CHOL") && !strcmp(getenv("MVBA_CHOL"), "barriers"));
    if (const char *ev = getenv("MVBA_TRAIL64_MIN")) h->trail64_min = std::max(0, atoi(ev));
    if (const char *ev = getenv("MVBA_CHOL_BARRIER_POLLS")) h->barrier_polls = (unsigned)std::max(0LL, atoll(ev));
  }
  TRYH(hipFuncSetAttribute(h->gcam ? (const void *)k_resid_jac<true> : (const void *)k_resid_jac<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)((size_t)((h->gcam ? 0 : ((m * CAM_LDS + 1) & ~1)) + (h->k1_threads / 64) * 64 * 2 * REC) * sizeof(double))));
  TRYH(hipFuncSetAttribute((const void *)k_cost<false>, hipFuncAttributeMaxDynamicSharedMemorySize, cam_lds));
#undef TRY
#undef TRYH
  lap("device allocations + uploads");
  trace.mark("attributes");
  *out = h;
  return MVBA_OK;
}

void mvba_destroy(mvba_handle *h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
#ifdef MVBA_SLOT_TRACE
  if (h->d_trace && getenv("MVBA_SLOT_TRACE")) {
    std::vector<long long> tr(16 * (size_t)h->n_waves);
    hipMemcpy(tr.data(), h->d_trace, sizeof(long long) * tr.size(), hipMemcpyDeviceToHost);
    if (FILE *f = fopen(getenv("MVBA_SLOT_TRACE"), "w")) {
      fprintf(f, "# block t0 t1 wait blocked polls hwid xcc nsteps (100 MHz ticks) | shader cycles summed over the steps: vmcnt-wait pace gather-issue index-dma compute loop-total - -; nR=%d lag=%d nseg=%d\n", h->slot_nR, h->slot_lag, h->slot_nseg);
      for (int b = 0; b < h->n_waves; ++b) {
        fprintf(f, "%d", b);
        for (int q = 0; q < 16; ++q) fprintf(f, " %lld", tr[16 * (size_t)b + q]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
  if (h->comm) g_rccl.CommDestroy(h->comm);
  void *ptrs[] = {h->d_pt_ptr, h->d_cam, h->d_obs_pt, h->d_xy, h->d_csc, h->d_tiles, h->d_tile_slot, h->d_splits, h->d_PLsplit, h->d_chunk_ptr, h->d_X[0], h->d_X[1],
                  h->d_cam15[0], h->d_cam15[1], h->d_rec, h->d_PL, h->d_PB, h->d_Ab, h->d_Ared, h->d_Ztiles, h->d_Lblk, h->d_lu,
                  h->d_dxi, h->d_dX, h->d_partials, h->d_cost, h->d_flag, h->d_allcost, h->d_it_k, h->d_it_l, h->d_it_a,
                  h->d_units, h->d_unit_ptr, h->d_q_ptr, h->d_q_units, h->d_q_head, h->d_partial, h->d_dense_part, h->d_dense_obs, h->d_sim, h->d_bar, h->d_wdesc,
                  h->d_wunits, h->d_seg_end, h->d_prog, h->d_trace, h->d_ipiv, h->d_range_o0, h->d_it_x, h->d_cam18, h->d_dxi10};
  for (void *q : ptrs) if (q) hipFree(q);
  for (double *q : h->snap_slabs) hipFree(q);
  if (h->h_cost) hipHostFree(h->h_cost);
  if (h->h_allcost) hipHostFree(h->h_allcost);
  for (auto &p : h->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  for (auto e : h->pool) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int mvba_set_params(mvba_handle *h, const double *X, const double *f, const double *u, const double *t, const double *R) {
  if (!h || !X || !f || !u || !t || !R) return fail(MVBA_ERR_BADARG, "null argument");
  MVBA_HIP(hipSetDevice(h->device));
  std::vector<double> cam((size_t)h->m * CAM_IN);
  for (int k = 0; k < h->m; ++k) {
    double *c = cam.data() + (size_t)k * CAM_IN;
    c[0] = f[k]; c[1] = u[2 * k]; c[2] = u[2 * k + 1];
    for (int i = 0; i < 3; ++i) c[3 + i] = t[3 * k + i];
    for (int i = 0; i < 9; ++i) c[6 + i] = R[9 * k + i];
  }
  MVBA_HIP(hipMemcpyAsync(h->d_X[h->cur], X, sizeof(double) * 3 * h->N, hipMemcpyHostToDevice, h->stream));
  MVBA_HIP(hipMemcpyAsync(h->d_cam15[h->cur], cam.data(), sizeof(double) * cam.size(), hipMemcpyHostToDevice, h->stream));
  MVBA_HIP(hipStreamSynchronize(h->stream));
  h->have_params = true; h->linearized = false; h->have_trial = false;
  return MVBA_OK;
}

int mvba_get_params(mvba_handle *h, double *X, double *f, double *u, double *t, double *R) {
  if (!h || !X || !f || !u || !t || !R) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->have_params) return fail(MVBA_ERR_STATE, "no parameters set");
  MVBA_HIP(hipSetDevice(h->device));
  std::vector<double> cam((size_t)h->m * CAM_IN);
  MVBA_HIP(hipMemcpyAsync(X, h->d_X[h->cur], sizeof(double) * 3 * h->N, hipMemcpyDeviceToHost, h->stream));
  MVBA_HIP(hipMemcpyAsync(cam.data(), h->d_cam15[h->cur], sizeof(double) * cam.size(), hipMemcpyDeviceToHost, h->stream));
  MVBA_HIP(hipStreamSynchronize(h->stream));
  for (int k = 0; k < h->m; ++k) {
    const double *c = cam.data() + (size_t)k * CAM_IN;
    f[k] = c[0]; u[2 * k] = c[1]; u[2 * k + 1] = c[2];
    for (int i = 0; i < 3; ++i) t[3 * k + i] = c[3 + i];
    for (int i = 0; i < 9; ++i) R[9 * k + i] = c[6 + i];
  }
  return MVBA_OK;
}

int mvba_apply_similarity(mvba_handle *h, const double *R0, const double *t0, double scale) {
  if (!h || !R0 || !t0) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->have_params) return fail(MVBA_ERR_STATE, "no parameters set");
  MVBA_HIP(hipSetDevice(h->device));
  double T[13];
  memcpy(T, R0, 9 * sizeof(double));
  memcpy(T + 9, t0, 3 * sizeof(double));
  T[12] = scale;
  if (!h->d_sim) {
    int rc = dmalloc(&h->d_sim, 13);
    if (rc) return rc;
  }
  MVBA_HIP(hipMemcpyAsync(h->d_sim, T, sizeof(T), hipMemcpyHostToDevice, h->stream));
  const long long n = std::max<long long>(h->N, h->m);
  hipLaunchKernelGGL(k_similarity, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->N, h->m, h->d_X[h->cur],
                     h->d_cam15[h->cur], h->d_sim);
  MVBA_HIP(hipGetLastError());
  MVBA_HIP(hipStreamSynchronize(h->stream));  // T lives on this frame's stack
  h->linearized = false; h->have_trial = false;
  return MVBA_OK;
}

int mvba_cost(mvba_handle *h, double *E) {
  if (!h || !E) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->have_params) return fail(MVBA_ERR_STATE, "no parameters set");
  MVBA_HIP(hipSetDevice(h->device));
  int rc = launch_cost(h, h->d_cam15[h->cur], h->d_X[h->cur]);
  if (rc) return rc;
  return global_cost(h, E);
}

int mvba_linearize(mvba_handle *h) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  if (!h->have_params) return fail(MVBA_ERR_STATE, "no parameters set");
  MVBA_HIP(hipSetDevice(h->device));
  if (h->nobs) {
    Timed t(h, MVBA_K_RESID_JAC);  // K1 with K2 (per-point blocks) fused in
    const int kt = h->k1_threads;  // 8 waves share one camera table: 2 blocks = 16 waves per CU
    const size_t lds = (size_t)((h->gcam ? 0 : ((h->m * CAM_LDS + 1) & ~1)) + (kt / 64) * 64 * 2 * REC) * sizeof(double);
    cam_tables(h, h->d_cam15[h->cur], nullptr);
    const int wpb = kt / 64;
    const int grid = std::max(1, std::min(2048 * 256 / kt, (h->n_tiles + wpb - 1) / wpb));
    hipLaunchKernelGGL(h->gcam ? k_resid_jac<true> : k_resid_jac<false>, dim3(grid), dim3(kt), lds, h->stream, h->nobs, h->m, h->d_cam15[h->cur],
                       h->d_X[h->cur], h->d_obs_pt, h->d_cam, h->d_xy, h->f0, h->d_tiles, h->n_tiles, h->d_rec, h->d_PL,
                       h->d_tile_slot, h->d_PLsplit, h->d_cam18);
    if (h->n_splits)
      hipLaunchKernelGGL(k_sum_split, dim3((9 * h->n_splits + 255) / 256), dim3(256), 0, h->stream, h->n_splits, h->d_splits,
                         h->d_PLsplit, h->d_PL);
  }
  MVBA_HIP(hipGetLastError());
  h->linearized = true; h->have_trial = false;
  h->stats.n_linearize++;
  return MVBA_OK;
}

int mvba_try_step(mvba_handle *h, double c, double *E_trial) {
  if (!h || !E_trial) return fail(MVBA_ERR_BADARG, "null argument");
  if (!h->linearized) return fail(MVBA_ERR_STATE, "try_step before linearize");
  MVBA_HIP(hipSetDevice(h->device));
  const int m = h->m, D = h->D;
  const size_t n9 = 9 * (size_t)m;
  const size_t nA = strip_offset(m, m);  // packed upper block triangle
  double *d_A = h->d_Ab, *d_b = h->d_Ab + nA;
  {
    Timed t(h, MVBA_K_POINT_INV);
    const long long nAb = (long long)(nA + n9);
    const unsigned grid = (unsigned)std::max<long long>((h->N + 255) / 256, std::min<long long>((nAb + 1023) / 1024, 4096));
    hipLaunchKernelGGL(k_point_inv, dim3(std::max(grid, 1u)), dim3(256), 0, h->stream, h->N, c, h->d_PL, h->d_PB, h->d_flag,
                       h->d_Ab, nAb, h->d_prog, h->slot_pace ? (long long)h->slot_rounds * h->slot_nR * h->slot_nseg * PACE_STRIDE : 0LL,
                       h->schur_mode == SCHUR_DENSE ? 1 : 0);
  }
  if (h->use_pairs) {
    Timed t(h, MVBA_K_SCHUR);
    // 64-bit offsets only when the records or the point blocks (+ the padding row) span 4 GiB (MVBA_FORCE_BIG: at test sizes too)
    const bool big = (std::max<long long>(h->nobs, h->N) + 1) * 128LL >= (1LL << 32) || h->force_big;
    if (h->schur_mode == SCHUR_SLOTS) {
      if (h->n_waves)
        hipLaunchKernelGGL(k_schur_slots, dim3(h->n_waves), dim3(64), SLOT_LDS, h->stream, h->d_wdesc,
                           h->d_wunits, h->d_it_x, (const int *)nullptr, (const int *)nullptr, h->d_rec, h->d_PB, c, h->f0, h->d_partial,
                           h->pair_static ? nullptr : h->d_q_head, h->slot_nR, h->n_waves / std::max(1, h->slot_nR), h->d_seg_end,
                           h->slot_pace ? h->d_prog : nullptr, h->slot_nseg, h->slot_lag, h->d_trace, h->d_range_o0);
    } else if (h->n_units) {
      const bool stat = h->pair_static;
      hipLaunchKernelGGL(big ? k_schur_pairs_big : k_schur_pairs, dim3(stat ? 8 * h->q_max : h->n_units), dim3(64),
                         PAIRS_LDS, h->stream, h->d_units, h->d_q_ptr, h->d_q_units, stat ? nullptr : h->d_q_head, h->d_it_k,
                         h->d_it_l, h->d_it_a, h->d_rec, h->d_PB, c, h->f0, h->d_partial);
    }
    hipLaunchKernelGGL(k_schur_reduce, dim3((unsigned)((long long)m * (m + 1) / 2)), dim3(128), 0, h->stream, m, h->d_unit_ptr,
                       h->d_partial, d_A, d_b, h->d_q_head);
  } else if (h->schur_mode == SCHUR_DENSE) {
    Timed t(h, MVBA_K_SCHUR);
    const int T = (9 * m + 15) / 16;
    const int CH = dense_ch(T);
    const size_t lds = sizeof(double) * ((size_t)2 * 3 * CH * 16 * T + (size_t)2 * CH * m * 32) + sizeof(double2) * CH * ((size_t)m * REC + 8) + sizeof(double) * (CH * (size_t)m + 2 * 3 * CH);
    auto launch = [&](auto kern) {
      if (!h->dense_attr_set) {  // (once per engine; the limit of the INSTANTIATION -- its largest camera count -- so that engines with other m share it)
        const int mm = 16 * T / 9;
        const size_t lds_max = sizeof(double) * ((size_t)2 * 3 * CH * 16 * T + (size_t)2 * CH * mm * 32 + (size_t)CH * mm + 2 * 3 * CH) + sizeof(double2) * CH * ((size_t)mm * REC + 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        h->dense_attr_set = true;
      }
      hipLaunchKernelGGL(kern, dim3(h->dense_blocks), dim3(64 * (dense_consumers(T) + CH)), lds, h->stream, (const double2 *)h->d_rec, (const double *)h->d_PB, (const int *)h->d_dense_obs,
                         (long long)h->N, m, 1.0 / h->f0, h->d_dense_part);
    };
    const bool table = h->d_dense_obs != nullptr;
#define MVBA_DENSE_CASE(t) case t: if (table) launch(k_schur_dense<t, true>); else launch(k_schur_dense<t, false>); break;
    switch (T) {
      MVBA_DENSE_CASE(1) MVBA_DENSE_CASE(2) MVBA_DENSE_CASE(3) MVBA_DENSE_CASE(4) MVBA_DENSE_CASE(5) MVBA_DENSE_CASE(6)
      MVBA_DENSE_CASE(7) MVBA_DENSE_CASE(8) MVBA_DENSE_CASE(9) MVBA_DENSE_CASE(10) MVBA_DENSE_CASE(11)
      default: if (table) launch(k_schur_dense<12, true>); else launch(k_schur_dense<12, false>); break;
    }
#undef MVBA_DENSE_CASE
    const long long n_el = (long long)nA + 9 * m;
    hipLaunchKernelGGL(k_schur_dense_finish, dim3((unsigned)((n_el + 3) / 4)), dim3(256), 0, h->stream, m, T, h->dense_blocks,
                       (const double *)h->d_dense_part, c, d_A, d_b);
  } else if (h->nobs) {
    Timed t(h, MVBA_K_SCHUR);
    const size_t lds = (81 * (size_t)h->lseg + 9) * sizeof(double);
    const bool big = h->nobs * 128LL >= (1LL << 32) || h->force_big;  // (MVBA_FORCE_BIG: exercise the 64-bit-offset kernels at test sizes)
    auto kern = h->nsp ? (big ? k_schur_strip<true, true> : k_schur_strip<false, true>)
                       : (big ? k_schur_strip<true, false> : k_schur_strip<false, false>);
    hipLaunchKernelGGL(kern, dim3(m, h->nchunks, h->nseg), dim3(h->schur_threads), lds, h->stream, m, h->nchunks, h->lseg, h->nsp,
                       h->d_chunk_ptr, h->d_csc, h->d_cam, h->d_rec, h->d_PB, c, h->f0, d_A, d_b);
  }
  MVBA_HIP(hipGetLastError());
  if (h->comm) {
    Timed t(h, MVBA_K_ALLREDUCE);
    ncclResult_t r = g_rccl.AllReduce(h->d_Ab, h->d_Ab, nA + n9, ncclDouble, ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) return fail(MVBA_ERR_RCCL, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  } else if (h->host_ar) {  // host-staged transport: D2H, caller's sum, H2D (no RCCL; see mvba_comm_init_host)
    h->host_buf.resize(nA + n9);
    MVBA_HIP(hipMemcpyAsync(h->host_buf.data(), h->d_Ab, sizeof(double) * (nA + n9), hipMemcpyDeviceToHost, h->stream));
    MVBA_HIP(hipStreamSynchronize(h->stream));
    const auto t0 = std::chrono::steady_clock::now();  // the wait above belongs to the Schur kernel, not to the exchange
    if (h->host_ar(h->host_ar_user, h->host_buf.data(), (int64_t)(nA + n9))) return fail(MVBA_ERR_RCCL, "host all-reduce callback failed");
    MVBA_HIP(hipMemcpyAsync(h->d_Ab, h->host_buf.data(), sizeof(double) * (nA + n9), hipMemcpyHostToDevice, h->stream));
    if (h->profiling) {  // host wall time of the exchange (callback + staging copy issue); the device path uses events
      h->stats.ms[MVBA_K_ALLREDUCE] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      h->stats.launches[MVBA_K_ALLREDUCE] += 1;
    }
  }
  auto launch_solve = [&](bool onepass) {  // K4: gauge strip, blocked Cholesky, back-substitution
    Timed t(h, MVBA_K_SOLVE);
    const int ld = h->ld;
    const int ntc = (D + NB - 1) / NB;
    hipLaunchKernelGGL(k_compact, dim3(ntc * (ntc + 1) / 2 + (D + 255) / 256), dim3(256), 0, h->stream, D, ld, m, h->gauge_axis, ntc, d_A, d_b,
                       h->d_Ared, h->d_bar, 1 + 4 * ((D + SBW - 1) / SBW));
    for (int jS = 0; jS < D; jS += SBW) {
      const int jE = std::min(jS + SBW, D);
      hipLaunchKernelGGL(k_chol_super, dim3((D + 1 - jE + 63) / 64), dim3(SUPER_THREADS), SUPER_LDS, h->stream, h->d_Ared, ld, D, jS,
                         h->d_Ztiles + (size_t)(jS / NB) * NB * NB, h->d_Lblk + (size_t)(jS / SBW) * SBW * SBW, h->d_flag);
      if (jE < D) {
        const int nt32 = (D + 1 - jE + NB - 1) / NB, nt64 = (D + 1 - jE + TB - 1) / TB;
        if (jE - jS == SBW && nt64 * (nt64 + 1) / 2 >= h->trail64_min)
          hipLaunchKernelGGL(k_chol_trail64, dim3(nt64 * (nt64 + 1) / 2), dim3(256), TRAIL64_LDS, h->stream, h->d_Ared, ld, D, jS, jE);
        else
          hipLaunchKernelGGL(k_chol_trail32, dim3(nt32 * (nt32 + 1) / 2), dim3(256), 0, h->stream, h->d_Ared, ld, D, jS, jE);
      }
    }
    const int S = (D + SBW - 1) / SBW;
    if (onepass && (S == 1 || S < h->n_cu)) {  // one persistent pass for L^T x = y (see k_chol_backsolve_all)
      // few bulk workgroups (each then takes several column groups per step): a barrier gets dearer with
      // every workgroup -- its release/acquire writes back and invalidates that XCD's L2 for everybody on
      // it.  D = 4493: 16 bulk workgroups 2.75 ms per solve, 64: 2.93, 220: 3.24 (tools/ab_solve.py).
      // (point to point -- the default since round 4 -- nobody pays for anybody else: one bulk workgroup per column group)
      const bool flow = h->chol_flow;
      const int ngrp = ((S - 1) * SBW + 31) / 32, nbulk = S > 1 ? std::max(1, std::min(std::min(h->n_cu - S, flow ? ngrp : 16), ngrp)) : 0;
      hipLaunchKernelGGL(flow ? k_chol_backsolve_all<true> : k_chol_backsolve_all<false>, dim3(S + nbulk), dim3(SUPER_THREADS),
                         BACKSOLVE_LDS, h->stream, h->d_Ared, ld, D, m, h->gauge_axis, h->d_Ztiles, h->d_Lblk, h->d_dxi, h->d_flag,
                         h->d_bar, h->barrier_polls);
    } else
    for (int jS = ((D - 1) / SBW) * SBW; jS >= 0; jS -= SBW) {
      const int jE = std::min(jS + SBW, D), jE2 = std::min(jE + SBW, D);
      const int nwg = (jE == D) ? 1 : 1 + (jS + 255) / 256;
      hipLaunchKernelGGL(k_chol_backsolve, dim3(nwg), dim3(256), 0, h->stream, h->d_Ared, ld, D, m, h->gauge_axis, h->d_Ztiles,
                         h->d_Lblk + (size_t)(jS / SBW) * SBW * SBW, h->d_dxi, jS, jE, jE2);
    }
  };
  launch_solve(h->chol_onepass);
  MVBA_HIP(hipGetLastError());
  const int trial = 1 - h->cur;
  auto launch_tail = [&]() {  // K6a + K5/K6: trial cameras, back-substitution, trial cost
    Timed t(h, MVBA_K_BACKSUB_COST);
    hipLaunchKernelGGL(k_update_cams, dim3((m + 63) / 64), dim3(64), 0, h->stream, m, h->d_cam15[h->cur], h->d_dxi,
                       h->d_cam15[trial]);
    if (h->N) {
      const size_t lds = h->gcam ? 0 : (size_t)m * (CAM_LDS + DXI_LDS) * sizeof(double);
      cam_tables(h, h->d_cam15[h->cur], h->d_dxi);
      const int lanes_env = h->backsub_lanes;  // (MVBA_BACKSUB_LANES at create; 0 = by mean degree)
      const double deg = (double)h->nobs / (double)h->N;
      const int G = lanes_env ? lanes_env : (deg <= 40.0 ? 2 : (deg <= 100.0 ? 4 : 8));
      // (a camera table above half the LDS leaves one block per CU: 1024 threads then, so that the CU still holds 16 waves)
      // (16 waves per CU is what the registers allow: 256-thread blocks reach it while four of them fit -- tables up to 40 KiB,
      // ~180 cameras --, 512-thread blocks while two fit, one 1024-thread block beyond)
      const bool wide = lds > 80 * 1024, mid = !wide && lds > 40 * 1024;
      const int bt = wide ? 1024 : (mid ? 512 : 256);
      const int nblk = (int)std::min<long long>(4096 * 256 / bt, (h->N * G + bt - 1) / bt);
      auto kern = h->gcam ? (G == 2 ? k_backsub<2, 256, true> : (G == 4 ? k_backsub<4, 256, true> : k_backsub<8, 256, true>))
                  : wide ? (G == 2 ? k_backsub<2, 1024> : (G == 4 ? k_backsub<4, 1024> : k_backsub<8, 1024>))
                  : mid ? (G == 2 ? k_backsub<2, 512> : (G == 4 ? k_backsub<4, 512> : k_backsub<8, 512>))
                        : (G == 2 ? k_backsub<2> : (G == 4 ? k_backsub<4> : k_backsub<8>));
      hipLaunchKernelGGL(kern, dim3(nblk), dim3(bt), lds, h->stream, h->N, m, h->d_pt_ptr, h->d_cam, h->d_PB, h->d_dxi,
                         h->d_X[h->cur], h->d_cam15[h->cur], h->f0, h->d_X[trial], h->d_dX, h->d_cam18, h->d_dxi10);
    }
    // K6: trial cost = the residual-only pass at the trial state (fixed grid, fixed tree: deterministic)
    launch_cost_kernel(h, h->d_cam15[trial], h->d_X[trial]);
    double *mail = cost_mail(h);  // (advances cost_seq: sequenced before the launch reads it)
    const unsigned long long seq = h->cost_seq;
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, h->stream, h->d_partials, h->cost_grid, h->d_cost, h->d_flag, mail, seq);
  };
  launch_tail();
  MVBA_HIP(hipGetLastError());
  h->stats.n_try_step++;
  int rc = global_cost(h, E_trial);
  if (rc) return rc;
  if ((*h->h_flag & 8) && !(*h->h_flag & 1)) {
    // A device-wide barrier of the persistent back-substitution gave up: its grid was not co-resident (another
    // process holds CUs -- e.g. ranks sharing a GPU).  Nothing is lost but time: the packed [A|b] is intact, so the
    // solve is redone with one launch per super-block (no barrier), and this handle stays on that path.
    MVBA_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
    h->chol_onepass = false;
    h->stats.n_barrier_fallback++;
    launch_solve(false);
    launch_tail();
    MVBA_HIP(hipGetLastError());
    rc = global_cost(h, E_trial);
    if (rc) return rc;
  }
  if ((*h->h_flag & 2) && !(*h->h_flag & (1 | 8))) {
    // The Cholesky met a non-positive pivot: the reduced system is not positive definite (e.g. a
    // negative damping factor).  The reference's np.linalg.solve is LU with partial pivoting and
    // does not care, so redo the solve that way (slow path, rare) and the tail of the step.
    if (!h->d_lu) {
      int rc_ = dmalloc(&h->d_lu, (size_t)D * (D + 1));
      if (rc_) return rc_;
      rc_ = dmalloc(&h->d_ipiv, (size_t)D);
      if (rc_) return rc_;
    }
    MVBA_HIP(hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream));
    {
      Timed t(h, MVBA_K_SOLVE);
      hipLaunchKernelGGL(k_compact_full, dim3((D + 1 + 255) / 256, D), dim3(256), 0, h->stream, D, m, h->gauge_axis, d_A, d_b, h->d_lu);
      for (int j0 = 0; j0 < D; j0 += LU_NB) {
        const int nb = std::min(LU_NB, D - j0), right = D + 1 - (j0 + nb);  // columns right of the panel incl. the rhs
        hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(1024), 0, h->stream, h->d_lu, D, j0, nb, h->d_ipiv, h->d_flag);
        hipLaunchKernelGGL(k_lu_swap, dim3((D + 1 - nb + 255) / 256), dim3(256), 0, h->stream, h->d_lu, D, j0, nb, h->d_ipiv);
        if (right > 0) hipLaunchKernelGGL(k_lu_trsm, dim3((right + 255) / 256), dim3(256), 0, h->stream, h->d_lu, D, j0, nb);
        const int below = D - (j0 + nb);
        if (below > 0)
          hipLaunchKernelGGL(k_lu_gemm, dim3((right + 63) / 64, (below + 63) / 64), dim3(256), 0, h->stream, h->d_lu, D, j0, nb);
      }
      hipLaunchKernelGGL(k_lu_backsub, dim3(1), dim3(1024), 0, h->stream, h->d_lu, D, m, h->gauge_axis, h->d_dxi);
    }
    h->stats.n_lu_fallback++;
    launch_tail();
    MVBA_HIP(hipGetLastError());
    rc = global_cost(h, E_trial);
    if (rc) return rc;
  }
  if (*h->h_flag) {
    const int fl = *h->h_flag;
    hipMemsetAsync(h->d_flag, 0, sizeof(int), h->stream);
    if (fl & 8) return fail(MVBA_ERR_HIP, "k_chol_backsolve_all: a device-wide barrier timed out twice (is another process holding the CUs?)");
    return fail(MVBA_ERR_SINGULAR, (fl & 1) ? "Singular matrix" : "Singular matrix (reduced camera system)");
  }
  if (h->check_solve) {
    // Debug mode (MVBA_CHECK_SOLVE=1, read in mvba_create): the residual of the reduced system, b - A dxi over the kept parameters,
    // from the packed [A|b] the solve started from (it is intact: k_compact copied it) and the dxi it produced -- on the host, in
    // plain loops.  The persistent back-substitution orders its hand-overs with sc1 accesses and explicit waits, not with the
    // memory model's fences: a stale read there would be a silently wrong camera step, which this catches.
    std::vector<double> Ab(nA + n9), x(n9);
    MVBA_HIP(hipMemcpyAsync(Ab.data(), h->d_Ab, sizeof(double) * (nA + n9), hipMemcpyDeviceToHost, h->stream));
    MVBA_HIP(hipMemcpyAsync(x.data(), h->d_dxi, sizeof(double) * n9, hipMemcpyDeviceToHost, h->stream));
    MVBA_HIP(hipStreamSynchronize(h->stream));
    auto kept = [&](size_t g) { return !((g >= 3 && g <= 8) || g == (size_t)(12 + h->gauge_axis)); };
    std::vector<double> r(n9, 0.0), an(n9, 0.0);  // r = A x, an = |A| |x|  (rows of the full symmetric matrix from the packed upper strips)
    for (int k = 0; k < m; ++k) {
      const double *Ak = Ab.data() + strip_offset(k, m);
      const int Wk = 9 * (m - k);
      for (int i = 0; i < 9; ++i)
        for (int c = 0; c < Wk; ++c) {
          const size_t gi = 9 * (size_t)k + i, gj = 9 * (size_t)k + c;
          const double a = Ak[(size_t)i * Wk + c];
          r[gi] += a * x[gj]; an[gi] += std::fabs(a * x[gj]);
          if (c >= 9) { r[gj] += a * x[gi]; an[gj] += std::fabs(a * x[gi]); }  // the mirror image below the diagonal blocks
        }
    }
    double worst = 0.0;
    for (size_t g = 0; g < n9; ++g)
      if (kept(g)) {
        const double bg = Ab[nA + g], scale = an[g] + std::fabs(bg);
        worst = std::max(worst, scale > 0.0 ? std::fabs(bg - r[g]) / scale : 0.0);
      }
    if (!(worst <= h->check_solve_tol))
      return fail(MVBA_ERR_STATE, "MVBA_CHECK_SOLVE: the dense solve left a relative residual of " + std::to_string(worst) +
                                      " on the reduced camera system (a stale hand-over in the back-substitution?)");
  }
  h->have_trial = true;
  return MVBA_OK;
}

int mvba_commit(mvba_handle *h) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  if (!h->have_trial) return fail(MVBA_ERR_STATE, "commit without a trial step");
  h->cur = 1 - h->cur;
  h->have_trial = false; h->linearized = false;
  h->stats.n_commit++;
  return MVBA_OK;
}

// ---- debug log: per-iteration copies of the committed state, device-resident (ref :89-98, :175-183, :204-206)
namespace {
constexpr int SNAP_SLAB = 8;  // log entries per device allocation
size_t snap_stride(const mvba_handle *h) { return (3 * (size_t)h->N + (size_t)CAM_IN * h->m + 1) & ~(size_t)1; }
}  // namespace

int mvba_snapshot(mvba_handle *h) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  if (!h->have_params) return fail(MVBA_ERR_STATE, "no parameters set");
  MVBA_HIP(hipSetDevice(h->device));
  const size_t stride = snap_stride(h);
  const size_t slab = (size_t)(h->n_snap / SNAP_SLAB);
  if (slab >= h->snap_slabs.size()) {
    double *p = nullptr;
    int rc = dmalloc(&p, stride * SNAP_SLAB);
    if (rc) return rc;
    h->snap_slabs.push_back(p);
  }
  double *dst = h->snap_slabs[slab] + stride * (size_t)(h->n_snap % SNAP_SLAB);
  // same stream as the kernels: ordered after everything that wrote the committed state and before whatever
  // overwrites it; the host does not wait
  MVBA_HIP(hipMemcpyAsync(dst, h->d_X[h->cur], sizeof(double) * 3 * h->N, hipMemcpyDeviceToDevice, h->stream));
  MVBA_HIP(hipMemcpyAsync(dst + 3 * h->N, h->d_cam15[h->cur], sizeof(double) * CAM_IN * h->m, hipMemcpyDeviceToDevice, h->stream));
  h->n_snap++;
  return MVBA_OK;
}

int mvba_snapshot_count(mvba_handle *h, int64_t *n) {
  if (!h || !n) return fail(MVBA_ERR_BADARG, "null argument");
  *n = h->n_snap;
  return MVBA_OK;
}

int mvba_snapshot_read(mvba_handle *h, int64_t i, double *X, double *f, double *u, double *t, double *R) {
  if (!h || !X || !f || !u || !t || !R) return fail(MVBA_ERR_BADARG, "null argument");
  if (i < 0 || i >= h->n_snap) return fail(MVBA_ERR_BADARG, "no such log entry");
  MVBA_HIP(hipSetDevice(h->device));
  const double *src = h->snap_slabs[(size_t)(i / SNAP_SLAB)] + snap_stride(h) * (size_t)(i % SNAP_SLAB);
  std::vector<double> cam((size_t)h->m * CAM_IN);
  MVBA_HIP(hipMemcpyAsync(X, src, sizeof(double) * 3 * h->N, hipMemcpyDeviceToHost, h->stream));
  MVBA_HIP(hipMemcpyAsync(cam.data(), src + 3 * h->N, sizeof(double) * cam.size(), hipMemcpyDeviceToHost, h->stream));
  MVBA_HIP(hipStreamSynchronize(h->stream));
  for (int k = 0; k < h->m; ++k) {
    const double *c = cam.data() + (size_t)k * CAM_IN;
    f[k] = c[0]; u[2 * k] = c[1]; u[2 * k + 1] = c[2];
    for (int q = 0; q < 3; ++q) t[3 * k + q] = c[3 + q];
    for (int q = 0; q < 9; ++q) R[9 * k + q] = c[6 + q];
  }
  return MVBA_OK;
}

int mvba_snapshot_restore(mvba_handle *h, int64_t i) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  if (i < 0 || i >= h->n_snap) return fail(MVBA_ERR_BADARG, "no such log entry");
  MVBA_HIP(hipSetDevice(h->device));
  const double *src = h->snap_slabs[(size_t)(i / SNAP_SLAB)] + snap_stride(h) * (size_t)(i % SNAP_SLAB);
  // what mvba_set_params does, from device memory: the committed state is replaced, linearisation and trial are void
  MVBA_HIP(hipMemcpyAsync(h->d_X[h->cur], src, sizeof(double) * 3 * h->N, hipMemcpyDeviceToDevice, h->stream));
  MVBA_HIP(hipMemcpyAsync(h->d_cam15[h->cur], src + 3 * h->N, sizeof(double) * CAM_IN * h->m, hipMemcpyDeviceToDevice, h->stream));
  h->have_params = true; h->linearized = false; h->have_trial = false;
  return MVBA_OK;
}

int mvba_snapshot_clear(mvba_handle *h) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  h->n_snap = 0;  // (the slabs stay for the next run; mvba_destroy frees them)
  return MVBA_OK;
}

int mvba_set_profiling(mvba_handle *h, int32_t enabled) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  h->profiling = enabled == 2 ? 2 : (enabled != 0);
  return MVBA_OK;
}

int mvba_get_stats(mvba_handle *h, mvba_stats *out) {
  if (!h || !out) return fail(MVBA_ERR_BADARG, "null argument");
  MVBA_HIP(hipSetDevice(h->device));
  int rc = sync_and_drain(h);
  if (rc) return rc;
  *out = h->stats;
  return MVBA_OK;
}

int mvba_get_info(mvba_handle *h, int64_t *out8) {
  if (!h || !out8) return fail(MVBA_ERR_BADARG, "null argument");
  out8[0] = h->n_items;
  out8[1] = h->n_items_offdiag;
  out8[2] = h->n_units;
  out8[3] = h->schur_mode | ((long long)(h->schur_mode == SCHUR_SLOTS ? h->slot_rounds : 0) << 8) |
            ((long long)(h->schur_mode == SCHUR_SLOTS ? h->slot_groups : 0) << 32);
  out8[4] = h->rccl_version;
  out8[5] = NCCL_VERSION_CODE;
  out8[6] = h->nranks;
  out8[7] = h->schur_mode == SCHUR_SLOTS ? h->n_slot_items : 0;  // step-major rows incl. padding
  return MVBA_OK;
}

int mvba_reset_stats(mvba_handle *h) {
  if (!h) return fail(MVBA_ERR_BADARG, "null handle");
  MVBA_HIP(hipSetDevice(h->device));
  int rc = sync_and_drain(h);
  if (rc) return rc;
  h->stats = mvba_stats{};
  return MVBA_OK;
}

int mvba_comm_unique_id(void *id128) {
  if (!id128) return fail(MVBA_ERR_BADARG, "null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  if (int rc = rccl_load()) return rc;
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(MVBA_ERR_RCCL, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
  memcpy(id128, &id, 128);
  return MVBA_OK;
}

int mvba_comm_init(mvba_handle *h, const void *id128, int32_t rank, int32_t n_ranks) {
  if (!h || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MVBA_ERR_BADARG, "bad comm arguments");
  MVBA_HIP(hipSetDevice(h->device));
  if (int rc = rccl_load()) return rc;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  ncclResult_t r = g_rccl.CommInitRank(&h->comm, n_ranks, id, rank);
  if (r != ncclSuccess) return fail(MVBA_ERR_RCCL, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  h->rank = rank; h->nranks = n_ranks;
  h->rccl_version = g_rccl.version;
  int rc = dmalloc(&h->d_allcost, 2 * (size_t)n_ranks);
  if (rc) return rc;
  MVBA_HIP(hipHostMalloc((void **)&h->h_allcost, 2 * sizeof(double) * n_ranks));
  return MVBA_OK;
}

int mvba_comm_init_host(mvba_handle *h, int32_t rank, int32_t n_ranks, mvba_host_allreduce_fn fn, void *user) {
  if (!h || !fn || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(MVBA_ERR_BADARG, "bad comm arguments");
  if (h->comm) return fail(MVBA_ERR_STATE, "handle already has an RCCL communicator");
  h->host_ar = fn; h->host_ar_user = user;
  h->rank = rank; h->nranks = n_ranks;
  return MVBA_OK;
}

int mvba_debug_read(mvba_handle *h, int32_t which, double *out, int64_t capacity, int64_t *n) {
  if (!h || !n) return fail(MVBA_ERR_BADARG, "null argument");
  MVBA_HIP(hipSetDevice(h->device));
  const size_t n9 = 9 * (size_t)h->m;
  long long cnt = 0;
  switch (which) {
    case MVBA_BUF_RESIDUAL: cnt = 2 * h->nobs; break;
    case MVBA_BUF_JX: cnt = 6 * h->nobs; break;
    case MVBA_BUF_JC: cnt = 18 * h->nobs; break;
    case MVBA_BUF_E: cnt = 6 * h->N; break;
    case MVBA_BUF_DP: cnt = 3 * h->N; break;
    case MVBA_BUF_A_FULL: cnt = n9 * n9; break;
    case MVBA_BUF_B_FULL: cnt = n9; break;
    case MVBA_BUF_DXI: cnt = n9; break;
    case MVBA_BUF_DX: case MVBA_BUF_TRIAL_X: cnt = 3 * h->N; break;
    case MVBA_BUF_TRIAL_CAM: cnt = (long long)CAM_IN * h->m; break;
    case MVBA_BUF_INDEX_K: case MVBA_BUF_INDEX_L: case MVBA_BUF_INDEX_A:
      cnt = h->schur_mode == SCHUR_SLOTS ? h->n_slot_items : (h->schur_mode == SCHUR_PAIRS ? h->n_items : 0);
      break;
    case MVBA_BUF_INDEX_SEG: cnt = h->schur_mode == SCHUR_SLOTS ? (long long)h->n_waves * h->slot_nseg : 0; break;
    default: return fail(MVBA_ERR_BADARG, "unknown buffer id");
  }
  *n = cnt;
  if (!out) return MVBA_OK;
  if (capacity < cnt) return fail(MVBA_ERR_BADARG, "output buffer too small");
  MVBA_HIP(hipStreamSynchronize(h->stream));
  auto d2h = [&](const void *src, size_t bytes) { return hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost); };
  if (which == MVBA_BUF_RESIDUAL || which == MVBA_BUF_JX || which == MVBA_BUF_JC) {
    std::vector<double> r((size_t)2 * REC * h->nobs);  // expand the records on the host
    MVBA_HIP(hipMemcpy(r.data(), h->d_rec, sizeof(double) * r.size(), hipMemcpyDeviceToHost));
    const double cu = 1.0 / h->f0;
    for (long long o = 0; o < h->nobs; ++o) {
      const double *q = r.data() + (size_t)o * 2 * REC;  // slot s -> (q[2s], q[2s+1]) = (row0, row1)
      if (which == MVBA_BUF_RESIDUAL) {
        out[2 * o] = q[14]; out[2 * o + 1] = q[15];
      } else if (which == MVBA_BUF_JX) {
        for (int rr = 0; rr < 2; ++rr) for (int i = 0; i < 3; ++i) out[6 * o + 3 * rr + i] = q[2 * i + rr];
      } else {
        double *jc = out + 18 * o;
        for (int rr = 0; rr < 2; ++rr) {
          jc[9 * rr + 0] = q[6 + rr];
          jc[9 * rr + 1] = rr == 0 ? cu : 0.0;
          jc[9 * rr + 2] = rr == 1 ? cu : 0.0;
          for (int i = 0; i < 3; ++i) jc[9 * rr + 3 + i] = -q[2 * i + rr];
          for (int i = 0; i < 3; ++i) jc[9 * rr + 6 + i] = q[8 + 2 * i + rr];
        }
      }
    }
  } else if (which == MVBA_BUF_E || which == MVBA_BUF_DP) {
    std::vector<double> pl(9 * (size_t)h->N);
    MVBA_HIP(hipMemcpy(pl.data(), h->d_PL, sizeof(double) * pl.size(), hipMemcpyDeviceToHost));
    for (long long a = 0; a < h->N; ++a) {
      if (which == MVBA_BUF_E) for (int i = 0; i < 6; ++i) out[6 * a + i] = pl[9 * a + i];
      else for (int i = 0; i < 3; ++i) out[3 * a + i] = pl[9 * a + 6 + i];
    }
  } else if (which == MVBA_BUF_A_FULL) {  // expand the packed strips to the dense symmetric matrix
    const int m = h->m;
    std::vector<double> pk(strip_offset(m, m));
    MVBA_HIP(hipMemcpy(pk.data(), h->d_Ab, sizeof(double) * pk.size(), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < n9; ++r)
      for (size_t cc = 0; cc < n9; ++cc) {
        const size_t lo = std::min(r, cc), hi = std::max(r, cc);
        const int k = (int)(lo / 9), l = (int)(hi / 9);
        // inside a diagonal block both halves are stored: keep the entry as computed
        const size_t rr = (k == l) ? r : lo, c2 = (k == l) ? cc : hi;
        out[r * n9 + cc] = pk[strip_offset(k, m) + (rr - 9 * k) * (size_t)(9 * (m - k)) + (c2 - 9 * k)];
      }
  } else if (which == MVBA_BUF_B_FULL) {
    MVBA_HIP(d2h(h->d_Ab + strip_offset(h->m, h->m), sizeof(double) * cnt));
  } else if (which == MVBA_BUF_DXI) {
    MVBA_HIP(d2h(h->d_dxi, sizeof(double) * cnt));
  } else if (which == MVBA_BUF_DX) {
    MVBA_HIP(d2h(h->d_dX, sizeof(double) * cnt));
  } else if (which == MVBA_BUF_TRIAL_X) {
    MVBA_HIP(d2h(h->d_X[1 - h->cur], sizeof(double) * cnt));
  } else if (which == MVBA_BUF_TRIAL_CAM) {
    MVBA_HIP(d2h(h->d_cam15[1 - h->cur], sizeof(double) * cnt));
  } else if (which >= MVBA_BUF_INDEX_K && which <= MVBA_BUF_INDEX_SEG) {  // the Schur index as the kernel reads it (ints, widened)
    if (h->schur_mode == SCHUR_SLOTS && which != MVBA_BUF_INDEX_SEG) {  // one 64-int row per step: k[21] | l[21] | a[21] | pad
      const long long n_steps = cnt / PSTEP;
      std::vector<int> tmp((size_t)n_steps * SLOT_IDX);
      if (cnt) MVBA_HIP(hipMemcpy(tmp.data(), h->d_it_x, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost));
      const int off = which == MVBA_BUF_INDEX_K ? 0 : (which == MVBA_BUF_INDEX_L ? PSTEP : 2 * PSTEP);
      for (long long st = 0; st < n_steps; ++st)
        for (int sl = 0; sl < PSTEP; ++sl) out[st * PSTEP + sl] = (double)tmp[(size_t)st * SLOT_IDX + off + sl];
      return MVBA_OK;
    }
    const int *src = which == MVBA_BUF_INDEX_K ? h->d_it_k : (which == MVBA_BUF_INDEX_L ? h->d_it_l : (which == MVBA_BUF_INDEX_A ? h->d_it_a : h->d_seg_end));
    std::vector<int> tmp((size_t)cnt);
    if (cnt) MVBA_HIP(hipMemcpy(tmp.data(), src, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost));
    for (long long i = 0; i < cnt; ++i) out[i] = (double)tmp[i];
  }
  return MVBA_OK;
}

int mvba_project(const double *X, int64_t n_points, const double *K, const double *R, const double *t, int32_t n_images,
                 const int64_t *pt_ptr, const int32_t *cam_idx, int64_t n_obs, double *xy, int32_t device) {
  if (!X || !K || !R || !t || !xy) return fail(MVBA_ERR_BADARG, "null argument");
  if (n_points < 0 || n_images < 1 || n_obs < 0 || (pt_ptr && !cam_idx)) return fail(MVBA_ERR_BADARG, "bad sizes");
  if (!pt_ptr && n_obs != n_points * (int64_t)n_images) return fail(MVBA_ERR_BADARG, "dense grid needs n_obs = n_points * n_images");
  if (n_points >= (1LL << 31)) return fail(MVBA_ERR_BADARG, "n_points must be < 2^31");
  if ((size_t)n_images * 12 * sizeof(double) > 160 * 1024 - 256) return fail(MVBA_ERR_BADARG, "too many cameras for the LDS camera table (max 1704)");
  if (n_obs == 0) return MVBA_OK;
  if (device >= 0) MVBA_HIP(hipSetDevice(device));
  std::vector<int> obs_pt;
  if (pt_ptr) {
    if (pt_ptr[0] != 0 || pt_ptr[n_points] != n_obs) return fail(MVBA_ERR_BADARG, "pt_ptr does not span n_obs");
    obs_pt.resize(n_obs);
    for (int64_t a = 0; a < n_points; ++a)
      for (int64_t o = pt_ptr[a]; o < pt_ptr[a + 1]; ++o) {
        if (cam_idx[o] < 0 || cam_idx[o] >= n_images) return fail(MVBA_ERR_BADARG, "cam_idx out of range");
        obs_pt[o] = (int)a;
      }
  }
  double *dX = nullptr, *dK = nullptr, *dR = nullptr, *dt = nullptr;
  double2 *dxy = nullptr;
  int *dpt = nullptr, *dcam = nullptr;
  auto cleanup = [&]() { for (void *q : {(void *)dX, (void *)dK, (void *)dR, (void *)dt, (void *)dxy, (void *)dpt, (void *)dcam}) if (q) hipFree(q); };
#define PRJ(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(MVBA_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
  PRJ(hipMalloc((void **)&dX, sizeof(double) * 3 * std::max<int64_t>(n_points, 1)));
  PRJ(hipMalloc((void **)&dK, sizeof(double) * 9 * n_images));
  PRJ(hipMalloc((void **)&dR, sizeof(double) * 9 * n_images));
  PRJ(hipMalloc((void **)&dt, sizeof(double) * 3 * n_images));
  PRJ(hipMalloc((void **)&dxy, sizeof(double2) * n_obs));
  PRJ(hipMemcpy(dX, X, sizeof(double) * 3 * n_points, hipMemcpyHostToDevice));
  PRJ(hipMemcpy(dK, K, sizeof(double) * 9 * n_images, hipMemcpyHostToDevice));
  PRJ(hipMemcpy(dR, R, sizeof(double) * 9 * n_images, hipMemcpyHostToDevice));
  PRJ(hipMemcpy(dt, t, sizeof(double) * 3 * n_images, hipMemcpyHostToDevice));
  if (pt_ptr) {
    PRJ(hipMalloc((void **)&dpt, sizeof(int) * n_obs));
    PRJ(hipMalloc((void **)&dcam, sizeof(int) * n_obs));
    PRJ(hipMemcpy(dpt, obs_pt.data(), sizeof(int) * n_obs, hipMemcpyHostToDevice));
    PRJ(hipMemcpy(dcam, cam_idx, sizeof(int) * n_obs, hipMemcpyHostToDevice));
  }
  const int lds = (int)(sizeof(double) * 12 * n_images);
  PRJ(hipFuncSetAttribute((const void *)k_project_obs, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n_obs + 255) / 256));
  hipLaunchKernelGGL(k_project_obs, dim3(grid), dim3(256), lds, 0, (long long)n_obs, n_images, dX, dK, dR, dt, dpt, dcam, dxy);
  PRJ(hipGetLastError());
  PRJ(hipMemcpy(xy, dxy, sizeof(double2) * n_obs, hipMemcpyDeviceToHost));
#undef PRJ
  cleanup();
  return MVBA_OK;
}

int mvba_host_obs_math(const double *X3, const double *cam15, const double *xy2, double f0, double *out26) {
  if (!X3 || !cam15 || !xy2 || !out26) return fail(MVBA_ERR_BADARG, "null argument");
  alignas(16) double c[CAM_LDS];
  expand_cam(cam15, f0, c);
  ObsJ J;
  obs_math(X3[0], X3[1], X3[2], c, xy2[0], xy2[1], f0, J);
  out26[0] = J.e0; out26[1] = J.e1;
  for (int r = 0; r < 2; ++r) for (int i = 0; i < 3; ++i) out26[2 + 3 * r + i] = J.jx[r][i];
  for (int r = 0; r < 2; ++r) for (int i = 0; i < 9; ++i) out26[8 + 9 * r + i] = J.jc[r][i];
  return MVBA_OK;
}

}  // extern "C"
