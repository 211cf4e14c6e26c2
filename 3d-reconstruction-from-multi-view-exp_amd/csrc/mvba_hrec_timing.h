// Timing-only builds (-DMVBA_HREC_TIMING; tools/build_hrec_timing.sh): the "h in the record" variant of the two pair-major Schur
// kernels, included by mvba.hip in the middle of its K3 section (it uses that file's constants and helpers).  Today's records are
// read with the NEW access pattern and arithmetic: right traffic, right instruction mix, WRONG numbers -- never a product library.
// What it measured (round 5, profiles/r05_hrec_timing.txt): two row gathers per off-diagonal item instead of three and 15 % more
// vector instructions per step, 150 / 154 VGPRs and no scratch at three waves per SIMD -- and the same 1.61 ms for the slot form at
// config 3 (1.52-1.54 against 1.49 at 94 cameras, also with three gathers in flight), 15.06 against 13.96 ms for the unit form on
// config 4's shard, before the per-trial scatter of h into the records that the real thing would add.  Not built.
#pragma once
// ------------------------------------------------------------------ K3, "h in the record" variant of the slot form
// (round 5; the verdict's item 1: TWO row gathers per off-diagonal item instead of three, on whatever register budget it needs)
// Record layout this loop reads (8 x double2, one 128-byte line per observation):
//   slot 0-2  J_X columns      slot 3  dJ/df      slot 4  (sigma, -)      slot 5-7  h = E^-1 J_X^T rows (h_i = (h[i][0], h[i][1]))
// with the residual in an array of its own.  J_omega is not stored: d = X - t is orthogonal to both rows j0, j1 of J_X
// (j . d = (r (a . d) - p (a_r . d)) / r^2 = 0), so d = sigma (j0 x j1) and
//   J_omega row 0 = j0 x d = sigma (b j0 - a j1),  row 1 = j1 x d = sigma (c j0 - b j1),   a = j0.j0, b = j0.j1, c = j1.j1.
// Off-diagonal item: k row = slots 0..4 (5 lanes, 12 rows per DMA instruction), l row = the whole record (9 lanes: slot 7 twice,
// so that the LDS stride is 9 quad-words -- odd), no point row: t = J_Xk h_l.  The omega ROWS of the block come out of the same two
// FMAs as the translation rows with (v0, v1) replaced by (g0, g1) = sigma_k [[b, c], [-a, -b]] (v0, v1); the omega COLUMNS are the
// translation columns with t replaced by t N_l, N_l = sigma_l [[b, -a], [c, -b]] -- chosen per lane by (alpha, beta) = (1, 0) / (0, 1).
// Diagonal item: k row = the whole record, residual (16 bytes from its own array), point row slots 3..4 (E^-1 dP, weight).
#ifndef MVBA_HREC_NBUF
#define MVBA_HREC_NBUF 3
#endif
constexpr int HK_L = 5, HL_L = 9;                            // 16-byte lanes per staged k row / l row
constexpr int HK_ROW = 16 * HK_L, HL_ROW = 16 * HL_L;        // 80 / 144 bytes
constexpr int HREC_BUF = PSTEP * (HK_ROW + HL_ROW);          // 4,704 B per staging buffer
constexpr int HREC_NBUF = MVBA_HREC_NBUF;                    // staging buffers = gathers in flight + 1
static_assert(SLOT_LDS == HREC_NBUF * HREC_BUF + HREC_NBUF * SLOT_IDX * 4, "launch size");
constexpr int HREC_OPS = 6;                                  // LDS-DMA operations per step, both kinds of wave
template <bool DIAG>
__device__ __forceinline__ void hrec_step(const char *buf, const int it, const int cg, double (&acc)[9][3], double (&dg)[3], double (&rb)[3]) {
  constexpr int KROW = DIAG ? HL_ROW : HK_ROW;
  constexpr int LB_OFF = PSTEP * KROW, PB_OFF = LB_OFF + PSTEP * 16;
  const int sel0 = cg == 0 ? 3 : 0;
  const double al12 = cg == 0 ? 0.0 : 1.0, bx1 = cg == 0 ? 1.0 : 0.0;
  const double alpha = cg == 2 ? 0.0 : 1.0, beta = cg == 2 ? 1.0 : 0.0;
  {
    const double2 *kr = reinterpret_cast<const double2 *>(buf + it * KROW);
    const double2 *lr = DIAG ? kr : reinterpret_cast<const double2 *>(buf + LB_OFF + it * HL_ROW);
    const double2 kx0 = kr[0], kx1 = kr[1], kx2 = kr[2], kf = kr[3];
    const double ksig = kr[4].x;
    const double2 lh0 = lr[5], lh1 = lr[6], lh2 = lr[7];
    double t00 = kx0.x * lh0.x + kx1.x * lh1.x + kx2.x * lh2.x, t01 = kx0.x * lh0.y + kx1.x * lh1.y + kx2.x * lh2.y;
    double t10 = kx0.y * lh0.x + kx1.y * lh1.x + kx2.y * lh2.x, t11 = kx0.y * lh0.y + kx1.y * lh1.y + kx2.y * lh2.y;
    // k side: sigma_k (a, b, c)
    const double ka = ksig * (kx0.x * kx0.x + kx1.x * kx1.x + kx2.x * kx2.x);
    const double kb = ksig * (kx0.x * kx0.y + kx1.x * kx1.y + kx2.x * kx2.y);
    const double kc = ksig * (kx0.y * kx0.y + kx1.y * kx1.y + kx2.y * kx2.y);
    const double2 lx0 = lr[0], lx1 = lr[1], lx2 = lr[2], s0v = lr[sel0];
    // l side: P = alpha I + beta N_l
    const double ls = beta * (DIAG ? ksig : lr[4].x);
    const double sa = ls * (lx0.x * lx0.x + lx1.x * lx1.x + lx2.x * lx2.x);
    const double sb = ls * (lx0.x * lx0.y + lx1.x * lx1.y + lx2.x * lx2.y);
    const double sc = ls * (lx0.y * lx0.y + lx1.y * lx1.y + lx2.y * lx2.y);
    const double p00 = alpha + sb, p11 = alpha - sb;
    double sx[3] = {s0v.x, al12 * lx1.x + bx1, al12 * lx2.x};
    double sy[3] = {s0v.y, al12 * lx1.y, al12 * lx2.y + bx1};
    double w0 = 0.0, w1 = 0.0, wgt = 1.0;
    if (DIAG) {
      const double2 e = reinterpret_cast<const double2 *>(buf + LB_OFF)[it];
      const double *pb = reinterpret_cast<const double *>(buf + PB_OFF + it * 32);  // E^-1 dP (3), weight
      wgt = pb[3];
      t00 -= 0.5 * wgt;
      t11 -= 0.5 * wgt;
      w0 = (kx0.x * pb[0] + kx1.x * pb[1] + kx2.x * pb[2] - e.x) * wgt;
      w1 = (kx0.y * pb[0] + kx1.y * pb[1] + kx2.y * pb[2] - e.y) * wgt;
#pragma unroll
      for (int q = 0; q < 3; ++q) {  // the actual columns of J_C (needed for the damping term and the right-hand side)
        const double ex = p00 * sx[q] - sa * sy[q], ey = sc * sx[q] + p11 * sy[q];
        sx[q] = ex;
        sy[q] = ey;
      }
    } else {
      // t <- t P
      const double e00 = t00 * p00 + t01 * sc, e01 = t01 * p11 - t00 * sa;
      const double e10 = t10 * p00 + t11 * sc, e11 = t11 * p11 - t10 * sa;
      t00 = e00; t01 = e01; t10 = e10; t11 = e11;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double v0 = t00 * sx[q] + t01 * sy[q], v1 = t10 * sx[q] + t11 * sy[q];
      const double g0 = kb * v0 + kc * v1, g1 = ka * v0 + kb * v1;
      acc[0][q] = fma(kf.y, v1, fma(kf.x, v0, acc[0][q]));
      acc[1][q] += v0;
      acc[2][q] += v1;
      acc[3][q] = fma(kx0.y, v1, fma(kx0.x, v0, acc[3][q]));
      acc[4][q] = fma(kx1.y, v1, fma(kx1.x, v0, acc[4][q]));
      acc[5][q] = fma(kx2.y, v1, fma(kx2.x, v0, acc[5][q]));
      acc[6][q] = fma(-kx0.y, g1, fma(kx0.x, g0, acc[6][q]));
      acc[7][q] = fma(-kx1.y, g1, fma(kx1.x, g0, acc[7][q]));
      acc[8][q] = fma(-kx2.y, g1, fma(kx2.x, g0, acc[8][q]));
      if (DIAG) {
        dg[q] = fma(wgt, sx[q] * sx[q] + sy[q] * sy[q], dg[q]);
        rb[q] += sx[q] * w0 + sy[q] * w1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <bool DIAG>
__device__ __forceinline__ void schur_slots_hrec(char *wbuf, const int lane, const long long beg, const int n,
                                                 const int *__restrict__ it_x, const double2 *__restrict__ rec,
                                                 const double2 *__restrict__ res, const double *__restrict__ PB, const double c,
                                                 const double cu, double *__restrict__ out, const int *__restrict__ slot_unit,
                                                 const SlotPace pace, const int n_pad_pt) {
  const int it = lane / 3, cg = lane - 3 * it;
  const int r5 = lane / 5, s5 = lane - 5 * r5;
  const int r9 = lane / 9, s9 = min(lane - 9 * r9, 7);
  double acc[9][3];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[i][q] = 0.0;
  double dg[3] = {0.0, 0.0, 0.0}, rb[3] = {0.0, 0.0, 0.0};
  constexpr int KROW = DIAG ? HL_ROW : HK_ROW;
  constexpr int LB_OFF = PSTEP * KROW;                     // l rows (off-diagonal) / residuals (diagonal)
  constexpr int PB_OFF = LB_OFF + PSTEP * 16;              // diagonal: point rows (32 bytes each) behind the residuals
  static_assert(LB_OFF + (DIAG ? PSTEP * 48 : PSTEP * HL_ROW) <= HREC_BUF, "staging buffer");
  auto compute = [&](const char *buf) { hrec_step<DIAG>(buf, it, cg, acc, dg, rb); };
  bool pacing = pace.prog != nullptr;
  int seg = 0, seg_stop = pacing ? as_const(pace.seg_end)[0] * PSTEP : 0x7fffffff;
  auto pace_at = [&](const int s0) {
    while (s0 == seg_stop) {
      if (lane == 0) __hip_atomic_fetch_add(pace.prog + PACE_STRIDE * seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++seg;
      seg_stop = seg < pace.nseg ? as_const(pace.seg_end)[seg] * PSTEP : 0x7fffffff;
      if (seg >= pace.lag && pacing) {
        int tries = 0;
        while (__hip_atomic_fetch_add(pace.prog + PACE_STRIDE * (seg - pace.lag), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pace.need) {
          if (++tries > 1024) { pacing = false; break; }
          __builtin_amdgcn_s_sleep(127);
        }
        if (tries > 0) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(2);
      }
    }
  };
  constexpr int G = HREC_NBUF - 1;                          // gathers in flight
  const int nst = n / PSTEP, last_st = nst - 1;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)wbuf;
  const unsigned ldsx0 = lds0 + HREC_NBUF * HREC_BUF;
  const int *xring = reinterpret_cast<const int *>(wbuf + HREC_NBUF * HREC_BUF);
  const int *xbase = it_x + beg * SLOT_IDX;
  auto dma = [&](int row, unsigned slot16, const void *base, unsigned lds) {
#if defined(MVBA_KO_GATHER)
    row = 0;
#endif
    const unsigned o = ((unsigned)row << 7) + slot16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(o), "s"(base), "s"(lds) : "memory");
  };
  auto dma16 = [&](int row, const void *base, unsigned lds) {  // 16-byte rows (the residual array)
#if defined(MVBA_KO_GATHER)
    row = 0;
#endif
    const unsigned o = (unsigned)row << 4;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(o), "s"(base), "s"(lds) : "memory");
  };
  const unsigned lane16 = (unsigned)min(lane, 15) << 4;
  auto dma_idx = [&](int st) {
    const int *src = xbase + (size_t)min(st, last_st) * SLOT_IDX;
    const unsigned dst = ldsx0 + (unsigned)(st % HREC_NBUF) * (SLOT_IDX * 4);
    if (lane < 16) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src), "s"(dst) : "memory");
  };
  const unsigned s5_16 = (unsigned)s5 << 4, s9_16 = (unsigned)s9 << 4;
  const int k5a = min(r5, PSTEP - 1), k5b = min(12 + r5, PSTEP - 1);
  const int q9a = min(r9, PSTEP - 1), q9b = min(7 + r9, PSTEP - 1), q9c = min(14 + r9, PSTEP - 1);
  const int xres = min(lane, PSTEP - 1), xpt = 2 * PSTEP + min(lane >> 1, PSTEP - 1);
  const unsigned pslot16 = (unsigned)(3 + (lane & 1)) << 4;
  auto issue_step = [&](int st) {
    const int *x = xring + (st % HREC_NBUF) * SLOT_IDX;
    const unsigned buf = lds0 + (unsigned)(st % HREC_NBUF) * HREC_BUF;
    if (!DIAG) {
      const int ka_ = x[k5a], kb_ = x[k5b], la_ = x[PSTEP + q9a], lb_ = x[PSTEP + q9b], lc_ = x[PSTEP + q9c];
      if (lane < 60) dma(ka_, s5_16, rec, buf);
      if (lane < 45) dma(kb_, s5_16, rec, buf + 12 * HK_ROW);
      if (lane < 63) {
        dma(la_, s9_16, rec, buf + LB_OFF);
        dma(lb_, s9_16, rec, buf + LB_OFF + 7 * HL_ROW);
        dma(lc_, s9_16, rec, buf + LB_OFF + 14 * HL_ROW);
      }
    } else {
      const int ka_ = x[q9a], kb_ = x[q9b], kc_ = x[q9c], kr_ = x[xres], a_ = x[xpt];
      if (lane < 63) {
        dma(ka_, s9_16, rec, buf);
        dma(kb_, s9_16, rec, buf + 7 * HL_ROW);
        dma(kc_, s9_16, rec, buf + 14 * HL_ROW);
      }
      if (lane < PSTEP) dma16(kr_, res, buf + LB_OFF);
      if (lane < 2 * PSTEP) dma(a_, pslot16, PB, buf + PB_OFF);
    }
  };
#pragma unroll
  for (int i = 0; i < G; ++i) dma_idx(i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < G; ++i) {
    issue_step(min(i, last_st));
    dma_idx(G + i);
  }
  for (int st = 0; st < nst; ++st) {
    if (G == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    pace_at(st * PSTEP);
    issue_step(st + G);
    dma_idx(st + 2 * G);
    compute(wbuf + (st % HREC_NBUF) * HREC_BUF);
  }
  static_assert(HREC_OPS == 6 && (G == 2 || G == 3), "counted wait");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const double cs0 = cg == 1 ? -1.0 : 1.0, cs12 = cg == 0 ? cu : cs0;
  if (pace.prog != nullptr && lane == 0)
    for (; seg < pace.nseg; ++seg) __hip_atomic_fetch_add(pace.prog + PACE_STRIDE * seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
}
// The unit form (one pair per wave, two staging buffers, compiler-counted waits) on the same rows and arithmetic.
template <bool DIAG, bool BIG>
__device__ __forceinline__ void schur_pairs_unit_hrec(char *wbuf, const int lane, const long long beg, const int n,
                                                      const int *__restrict__ it_k, const int *__restrict__ it_l,
                                                      const int *__restrict__ it_a, const double2 *__restrict__ rec,
                                                      const double2 *__restrict__ res, const double *__restrict__ PB,
                                                      const double c, const double cu, double *__restrict__ out) {
  const int it = lane / 3, cg = lane - 3 * it;
  const int r5 = lane / 5, s5 = lane - 5 * r5;
  const int r9 = lane / 9, s9 = min(lane - 9 * r9, 7);
  double acc[9][3];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[i][q] = 0.0;
  double dg[3] = {0.0, 0.0, 0.0}, rb[3] = {0.0, 0.0, 0.0};
  constexpr int KROW = DIAG ? HL_ROW : HK_ROW;
  constexpr int LB_OFF = PSTEP * KROW, PB_OFF = LB_OFF + PSTEP * 16;
  int xk[3], xl[3], xa[1];  // DIAG: xk = the three 9-lane chunks, xl[0] = residual row, xa = point row; else xk[0..1] 5-lane chunks, xl 9-lane chunks
  auto load_idx = [&](int s0) {
    const unsigned last = (unsigned)(min(PSTEP, n - s0) - 1);
    typedef const int __attribute__((address_space(1))) *gint_p;
    typedef const char __attribute__((address_space(1))) *gchar_p;
    auto uni = [](const int *p) {
      const unsigned long long v = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return (gint_p)(((unsigned long long)hi << 32) | lo);
    };
    auto row_of = [&](int r) {
      unsigned off = min((unsigned)r, last) << 2;
      asm("" : "+v"(off));
      return off;
    };
    auto at = [](gint_p base, unsigned off) { return *(gint_p)((gchar_p)base + off); };
    const gint_p pk = uni(it_k + (beg + s0)), pl = DIAG ? pk : uni(it_l + (beg + s0)), pa = uni(it_a + (beg + s0));
    if (DIAG) {
      xk[0] = at(pk, row_of(r9)); xk[1] = at(pk, row_of(7 + r9)); xk[2] = at(pk, row_of(14 + r9));
      xl[0] = at(pk, row_of(lane));
      xa[0] = at(pa, row_of(lane >> 1));
    } else {
      xk[0] = at(pk, row_of(r5)); xk[1] = at(pk, row_of(12 + r5));
      xl[0] = at(pl, row_of(r9)); xl[1] = at(pl, row_of(7 + r9)); xl[2] = at(pl, row_of(14 + r9));
    }
  };
  auto rec_at = [&](int obs, int slot) -> const void * {
    if (BIG) return rec + (size_t)obs * REC + slot;
    return reinterpret_cast<const char *>(rec) + (((unsigned)obs << 7) + ((unsigned)slot << 4));
  };
  auto issue = [&](char *buf) {
    if (DIAG) {
      if (lane < 63) {
        lds_dma16(rec_at(xk[0], s9), buf);
        lds_dma16(rec_at(xk[1], s9), buf + 7 * HL_ROW);
        lds_dma16(rec_at(xk[2], s9), buf + 14 * HL_ROW);
      }
      if (lane < PSTEP) lds_dma16(res + (BIG ? (size_t)xl[0] : (size_t)(unsigned)xl[0]), buf + LB_OFF);
      if (lane < 2 * PSTEP) lds_dma16(PB + (size_t)(unsigned)xa[0] * PBS + 2 * (3 + (lane & 1)), buf + PB_OFF);
    } else {
      if (lane < 60) lds_dma16(rec_at(xk[0], s5), buf);
      if (lane < 45) lds_dma16(rec_at(xk[1], s5), buf + 12 * HK_ROW);
      if (lane < 63) {
        lds_dma16(rec_at(xl[0], s9), buf + LB_OFF);
        lds_dma16(rec_at(xl[1], s9), buf + LB_OFF + 7 * HL_ROW);
        lds_dma16(rec_at(xl[2], s9), buf + LB_OFF + 14 * HL_ROW);
      }
    }
  };
  load_idx(0);
  issue(wbuf);
  if (PSTEP < n) load_idx(PSTEP);
  for (int s0 = 0, par = 0; s0 < n; s0 += PSTEP, par ^= 1) {
    const int ns = min(PSTEP, n - s0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (s0 + PSTEP < n) {
      issue(wbuf + (par ^ 1) * HREC_BUF);
      if (s0 + 2 * PSTEP < n) load_idx(s0 + 2 * PSTEP);
    }
    if (it < ns) hrec_step<DIAG>(wbuf + par * HREC_BUF, it, cg, acc, dg, rb);
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const double cs0 = cg == 1 ? -1.0 : 1.0, cs12 = cg == 0 ? cu : cs0;
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
        out[81 + 3 * cg + q] = 2.0 * c * cs * cs * d;
        out[90 + 3 * cg + q] = 2.0 * cs * r;
      }
    }
  }
}
