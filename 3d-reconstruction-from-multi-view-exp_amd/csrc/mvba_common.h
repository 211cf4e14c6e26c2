// Shared helpers for libmvba.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/mvba.h"

namespace mvba {

extern thread_local std::string g_err;
inline int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

#define MVBA_HIP(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return ::mvba::fail(MVBA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

// Per-camera record as the kernels keep it in LDS.  18 doubles = 9 quad-words, rows 16-byte aligned: a lane takes its
// camera's row in nine 16-byte reads (ds_read_b128), and nine being odd the rows of 16 different cameras start in 16
// different bank quads.  (Rounds 1-3: 19 doubles read 8 bytes at a time -- 27 reads per observation in the
// back-substitution, every one of them a 32-into-32 birthday problem over the cameras of the wave's lanes: LDS 78 % busy,
// 43 % of that in bank conflicts, profiles/r04_n_pmc_summary_config3.txt.)
constexpr int CAM_IN = 15;   // f,u,v,t[3],R[9]  (HBM layout, [m][15])
constexpr int CAM_LDS = 18;  // + 1/f, u/f0, v/f0
constexpr int DXI_LDS = 10;  // a camera's 9 update components, padded to five quad-words (odd again)

struct alignas(16) mvba_quad { double x, y; };
// n doubles (n even) of a 16-byte aligned row into registers, 16 bytes at a time
template <int N>
__host__ __device__ __forceinline__ void load_row(const double *row, double (&v)[N]) {
  static_assert(N % 2 == 0, "rows are whole quad-words");
  const mvba_quad *q = reinterpret_cast<const mvba_quad *>(row);
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const mvba_quad t = q[i];
    v[2 * i] = t.x;
    v[2 * i + 1] = t.y;
  }
}

struct ObsJ {
  double e0, e1;
  double jx[2][3];
  double jc[2][9];
};

// Residual and Jacobian rows of one observation.  Follows the reference's
// operation order where it matters for rounding:
//   p,q,r                    lib/bundle_adjustment.py:302-305
//   dp/df = (p - u/f0*r)/f   :336-337      dp/du = r/f0   :350-355
//   dp/dt = -(f r1 + u r3)   :368-376      d/domega = (-d/dt) x (X - t)   :391-396
//   rows (r d(p|q) - (p|q) dr) / r^2       :450-467, :492-509
// c = pointer to a CAM_LDS record (f,u,v,t[3],R[9] row-major with columns = axes,
// 1/f, u/f0, v/f0).
__host__ __device__ __forceinline__ void obs_math(double X0, double X1, double X2, const double *c,
                                                  double x, double y, double f0, ObsJ &J) {
  double cc[CAM_LDS];
  load_row(c, cc);
  c = cc;
  const double f = c[0], u = c[1], v = c[2];
  const double d0 = X0 - c[3], d1 = X1 - c[4], d2 = X2 - c[5];
  const double *R = c + 6;
  const double c1 = R[0] * d0 + R[3] * d1 + R[6] * d2;
  const double c2 = R[1] * d0 + R[4] * d1 + R[7] * d2;
  const double c3 = R[2] * d0 + R[5] * d1 + R[8] * d2;
  const double p = f * c1 + u * c3, q = f * c2 + v * c3, r = f0 * c3;
  J.e0 = p / r - x / f0;
  J.e1 = q / r - y / f0;
  const double inv_r2 = 1.0 / (r * r);
  double ap[3], aq[3], ar[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ap[i] = f * R[3 * i + 0] + u * R[3 * i + 2];
    aq[i] = f * R[3 * i + 1] + v * R[3 * i + 2];
    ar[i] = f0 * R[3 * i + 2];
    J.jx[0][i] = (r * ap[i] - p * ar[i]) * inv_r2;
    J.jx[1][i] = (r * aq[i] - q * ar[i]) * inv_r2;
    J.jc[0][3 + i] = -J.jx[0][i];
    J.jc[1][3 + i] = -J.jx[1][i];
  }
  const double dpdf = (p - c[16] * r) * c[15];
  const double dqdf = (q - c[17] * r) * c[15];
  J.jc[0][0] = (r * dpdf) * inv_r2;
  J.jc[1][0] = (r * dqdf) * inv_r2;
  const double rf0 = r / f0;
  J.jc[0][1] = (r * rf0) * inv_r2;
  J.jc[1][1] = 0.0;
  J.jc[0][2] = 0.0;
  J.jc[1][2] = (r * rf0) * inv_r2;
  const double d[3] = {d0, d1, d2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    const double wp = ap[i1] * d[i2] - ap[i2] * d[i1];
    const double wq = aq[i1] * d[i2] - aq[i2] * d[i1];
    const double wr = ar[i1] * d[i2] - ar[i2] * d[i1];
    J.jc[0][6 + i] = (r * wp - p * wr) * inv_r2;
    J.jc[1][6 + i] = (r * wq - q * wr) * inv_r2;
  }
}

// y_o = 2 Jx^T (Jc dxi_k) of one observation WITHOUT forming the Jacobian rows (K5): the same
// linear map as obs_math's jx / jc (with the implied columns (u,v) -> 1/f0, t -> -J_X), evaluated
// as a directional derivative.  With d = X - t, e = d x domega, g = e - dt, h = R^T g:
//   delta p = dpdf df + f h1 + u h3,  delta q = dqdf df + f h2 + v h3,  delta r = f0 h3
//   s = ((r delta p - p delta r) / r^2 + du / f0, (r delta q - q delta r) / r^2 + dv / f0)
//   y = (2 / r^2) R [r f s0, r f s1, r (u s0 + v s1) - f0 (p s0 + q s1)]^T
// (the row-vector identities (a x d) . w = a . (d x w) and a_p . g = f h1 + u h3 fold the 18
// entries of Jc into h).  ~75 fp64 operations and one division instead of ~250 and three.
__host__ __device__ __forceinline__ void obs_backsub(double X0, double X1, double X2, const double *c, const double *dk,
                                                     double f0, double &y0, double &y1, double &y2) {
  double cc[CAM_LDS], dd[DXI_LDS];  // dk: a DXI_LDS row (16-byte aligned, the tenth double is padding)
  load_row(c, cc);
  load_row(dk, dd);
  c = cc;
  dk = dd;
  const double f = c[0], u = c[1], v = c[2];
  const double d0 = X0 - c[3], d1 = X1 - c[4], d2 = X2 - c[5];
  const double *R = c + 6;
  const double c1 = R[0] * d0 + R[3] * d1 + R[6] * d2;
  const double c2 = R[1] * d0 + R[4] * d1 + R[7] * d2;
  const double c3 = R[2] * d0 + R[5] * d1 + R[8] * d2;
  const double p = f * c1 + u * c3, q = f * c2 + v * c3, r = f0 * c3;
  const double g0 = d1 * dk[8] - d2 * dk[7] - dk[3];
  const double g1 = d2 * dk[6] - d0 * dk[8] - dk[4];
  const double g2 = d0 * dk[7] - d1 * dk[6] - dk[5];
  const double h1 = R[0] * g0 + R[3] * g1 + R[6] * g2;
  const double h2 = R[1] * g0 + R[4] * g1 + R[7] * g2;
  const double h3 = R[2] * g0 + R[5] * g1 + R[8] * g2;
  const double dpdf = (p - c[16] * r) * c[15], dqdf = (q - c[17] * r) * c[15];
  const double dp = dpdf * dk[0] + f * h1 + u * h3, dq = dqdf * dk[0] + f * h2 + v * h3, dr = f0 * h3;
  const double inv_r2 = 1.0 / (r * r), cu = 1.0 / f0;
  const double s0 = (r * dp - p * dr) * inv_r2 + cu * dk[1];
  const double s1 = (r * dq - q * dr) * inv_r2 + cu * dk[2];
  const double w0 = r * f * s0, w1 = r * f * s1, w2 = r * (u * s0 + v * s1) - f0 * (p * s0 + q * s1);
  const double sc = 2.0 * inv_r2;
  y0 = sc * (R[0] * w0 + R[1] * w1 + R[2] * w2);
  y1 = sc * (R[3] * w0 + R[4] * w1 + R[5] * w2);
  y2 = sc * (R[6] * w0 + R[7] * w1 + R[8] * w2);
}

// Residual only (trial cost, ref :666-677).
__host__ __device__ __forceinline__ double obs_cost(double X0, double X1, double X2, const double *c,
                                                    double x, double y, double f0) {
  double cc[CAM_LDS];
  load_row(c, cc);
  c = cc;
  const double d0 = X0 - c[3], d1 = X1 - c[4], d2 = X2 - c[5];
  const double *R = c + 6;
  const double c1 = R[0] * d0 + R[3] * d1 + R[6] * d2;
  const double c2 = R[1] * d0 + R[4] * d1 + R[7] * d2;
  const double c3 = R[2] * d0 + R[5] * d1 + R[8] * d2;
  const double p = c[0] * c1 + c[1] * c3, q = c[0] * c2 + c[2] * c3, r = f0 * c3;
  const double e0 = p / r - x / f0, e1 = q / r - y / f0;
  return e0 * e0 + e1 * e1;
}

__host__ __device__ __forceinline__ void expand_cam(const double *in15, double f0, double *out18) {
#pragma unroll
  for (int i = 0; i < CAM_IN; ++i) out18[i] = in15[i];
  out18[15] = 1.0 / in15[0];
  out18[16] = in15[1] / f0;
  out18[17] = in15[2] / f0;
}


// Lanes of ONE wave exchanging data through LDS: the hardware executes a wave's LDS instructions
// in order, but the compiler only reasons per thread (it may, e.g., sink loads into the arms of a
// divergent branch, where the other lanes' stores have not happened yet).  This is the
// wave-level equivalent of __syncthreads(): a convergent barrier plus release/acquire fences.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#else
__device__ inline void wave_sync() {}
#endif

}  // namespace mvba
