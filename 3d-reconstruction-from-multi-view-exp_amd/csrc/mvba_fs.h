// K3, slot form with ONE LANE PER ITEM (-DMVBA_FS: 64 lists per wave; included by mvba.hip in the middle of its K3 section, it uses
// that file's constants and helpers).  A lane owns one list -- the items of one (pair, sub-list) inside one point range -- and
// keeps that pair's WHOLE 9 x 9 block in its registers (81 fp64 accumulators; the three-lanes-per-item form of k_schur_slots
// keeps 27 per lane and forms t = J_Xk E^-1 J_Xl^T three times per item).  Per item: ~205 instead of 282 vector instructions,
// 272 instead of ~780 bytes of LDS reads (a lane reads its own rows once), a third of the index rows, pacing checks and scalar
// bookkeeping; the row gathers (3 per item) are the same.  What it costs: a step stages 64 x 272 B = 17,408 B, three buffers (two
// gathers in flight) + a two-deep index ring are 53,760 B per wave -- THREE waves per CU, 96 per XCD for the 93 a point range needs at
// 100 cameras -- so every latency is hidden by the wave's own software pipeline or not at all.
//
// Staging buffer of a step (one per lane = slot = item):  k rows [64][112 B] | l rows [64][112 B] (DIAG: the residual slots,
// [64][16 B]) | point rows [64][48 B] (DIAG: [64][80 B]: E^-1, E^-1 dP, weight).  Every gather instruction is a full wave:
// 64 rows x 7 slots = 7 instructions per record side, 3 (5) for the point rows, 1 for the residuals; lane-slot e = 64 j + lane of
// instruction j is row e / 7 (e / 3, e / 5), slot e % 7, and lands at byte 16 e of its region -- the row-major layout.
// Index row of a step: k[64] | l[64] | a[64] (768 B, one 48-lane LDS-DMA), two deep: iteration s reads the indices of step s + 2
// from ring slot s % 2 into registers, waits for those reads, sends the index DMA of step s + 4 into the SAME slot, then the gathers
// of step s + 2, then computes step s.  vmcnt retires in order, so "all but the last iteration's G + 1 operations" at the top of
// iteration s = the gathers of step s have landed and so have the indices of step s + 2 -- an index row has two iterations to
// arrive.  (With the row of step s + 3 sent into the OTHER slot it had one, and a fresh 768-byte row from HBM takes longer than a
// step's arithmetic: every step waited for it -- 1.04 ms for the kernel with no gathers at all.)
#pragma once

constexpr int FS_N = 64;                      // slots (= lanes = items) per step
constexpr int FS_IDX = 3 * FS_N;              // ints per index row
static_assert(PSTEP == FS_N && SLOT_IDX == FS_IDX, "the index builder must be compiled for 64 slots per wave");

template <bool DIAG>
__device__ __forceinline__ void schur_slots_fs(char *wbuf, const int lane, const long long beg, const int nst,
                                               const int *__restrict__ it_x, const double2 *__restrict__ rec,
                                               const double *__restrict__ PB, const double c, const double cu,
                                               double *__restrict__ out, const int *__restrict__ slot_unit, const SlotPace pace) {
  constexpr int NPS = DIAG ? 5 : 3;                        // staged 16-byte slots of a point row
  constexpr int LB_OFF = FS_N * PROW;
  constexpr int PB_OFF = LB_OFF + (DIAG ? FS_N * 16 : FS_N * PROW);
  constexpr int BUFSZ = PB_OFF + FS_N * 16 * NPS;
  constexpr int NBUF = 3;
  constexpr int G = DIAG ? 7 + 1 + 5 : 7 + 7 + 3;          // gathers per step
  static_assert(NBUF * BUFSZ + 2 * FS_IDX * 4 <= SLOT_LDS, "staging larger than the launch provides");
  double acc[9][9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = 0.0;
  double dg[9], rb[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) dg[j] = rb[j] = 0.0;

  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)wbuf;
  const unsigned ldsx0 = lds0 + NBUF * BUFSZ;              // the index ring (two rows)
  const int *xring = reinterpret_cast<const int *>(wbuf + NBUF * BUFSZ);
  const int *xbase = it_x + beg * FS_IDX;                  // (`beg` = first STEP of this wave; wave-uniform)
  const int last_st = nst - 1;
  // this lane's (row, slot) of gather instruction j: records 7 slots per row, point rows NPS
  auto rec_row = [&](int j) { return (64 * j + lane) / 7; };
  auto rec_s16 = [&](int j) { return (unsigned)((64 * j + lane) % 7) << 4; };
  auto pb_row = [&](int j) { return (64 * j + lane) / NPS; };
  auto pb_s16 = [&](int j) { return (unsigned)((64 * j + lane) % NPS) << 4; };
  auto dma = [&](int row, unsigned slot16, const void *base, unsigned lds) {  // 16 bytes per lane: base[row * 128 + slot16] -> LDS
#if defined(MVBA_KO_GATHER)  // (timing-only knock-outs, as in k_schur_slots: every gather fetches row 0 of its array / no gather at all)
    row = 0;
#endif
#if defined(MVBA_KO_DMA)
    (void)row; (void)slot16; (void)base; (void)lds;
#else
    unsigned o;
    asm volatile("s_mov_b32 m0, %4\n\tv_lshl_add_u32 %0, %1, 7, %2\n\tglobal_load_lds_dwordx4 %0, %3" : "=&v"(o) : "v"(row), "v"(slot16), "s"(base), "s"(lds) : "memory");
#endif
  };
  const unsigned lane16 = (unsigned)min(lane, 47) << 4;
  auto dma_idx = [&](int st, unsigned ring_off) {  // the 768-byte index row of step st -> ring slot (lanes 0..47, 16 bytes each)
    const int *src = xbase + (size_t)min(st, last_st) * FS_IDX;  // wave-uniform
    const unsigned dst = ldsx0 + ring_off;
    if (lane < 48) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(src), "s"(dst) : "memory");
  };
  constexpr unsigned RING_B = FS_IDX * 4;
  // the gathers of step `st_next` from its indices in ring slot `ring_off` (landed: the caller's wait saw to it); once the indices are
  // in registers the slot is refilled with the row of step `st_idx`
  auto issue_step = [&](unsigned ring_off, unsigned buf_off, int st_idx) {
    const int *x = reinterpret_cast<const int *>(reinterpret_cast<const char *>(xring) + ring_off);
    const unsigned buf = lds0 + buf_off;
    int kx[7], ax[NPS], lx[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) kx[j] = x[rec_row(j)];
#pragma unroll
    for (int j = 0; j < NPS; ++j) ax[j] = x[2 * FS_N + pb_row(j)];
#pragma unroll
    for (int j = 0; j < 7; ++j) lx[j] = DIAG ? (j == 0 ? x[lane] : 0) : x[FS_N + rec_row(j)];
#pragma unroll
    for (int j = 0; j < 7; ++j) asm volatile("" : "+v"(kx[j]), "+v"(lx[j]));  // (the reads are issued here, not sunk below the DMA that overwrites their source)
#pragma unroll
    for (int j = 0; j < NPS; ++j) asm volatile("" : "+v"(ax[j]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_idx(st_idx, ring_off);
    if (!DIAG) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        dma(kx[j], rec_s16(j), rec, buf + 1024 * j);
        dma(lx[j], rec_s16(j), rec, buf + LB_OFF + 1024 * j);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) dma(kx[j], rec_s16(j), rec, buf + 1024 * j);
      dma(lx[0], 7u << 4, rec, buf + LB_OFF);  // l-side == k-side: only the residual (slot 7) is fetched
    }
#pragma unroll
    for (int j = 0; j < NPS; ++j) dma(ax[j], pb_s16(j), PB, buf + PB_OFF + 1024 * j);
  };

  // pacing (see schur_pairs_unit): the waves of a point range keep within `lag` segments of each other
  bool pacing = pace.prog != nullptr;
  int seg = 0, seg_stop = pacing ? as_const(pace.seg_end)[0] : 0x7fffffff;  // (in steps)
  auto pace_at = [&](const int st) {
    while (st == seg_stop) {  // (wave-uniform) this wave has left segment `seg`
      if (lane == 0) __hip_atomic_fetch_add(pace.prog + PACE_STRIDE * seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++seg;
      seg_stop = seg < pace.nseg ? as_const(pace.seg_end)[seg] : 0x7fffffff;
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

  // the arithmetic of one step on a landed buffer: this lane's item
  auto compute = [&](const char *buf) {
    const double2 *kr = reinterpret_cast<const double2 *>(buf + lane * PROW);
    const double2 *lr = DIAG ? kr : reinterpret_cast<const double2 *>(buf + LB_OFF + lane * PROW);
    const double *pb = reinterpret_cast<const double *>(buf + PB_OFF + lane * (16 * NPS));
    const double2 kx0 = kr[0], kx1 = kr[1], kx2 = kr[2];
    const double2 lx0 = lr[0], lx1 = lr[1], lx2 = lr[2];
    const double i00 = pb[0], i01 = pb[1], i02 = pb[2], i11 = pb[3], i12 = pb[4], i22 = pb[5];
    // h = E^-1 Jx_l^T (3x2), t = Jx_k h (2x2)
    const double h0x = i00 * lx0.x + i01 * lx1.x + i02 * lx2.x, h0y = i00 * lx0.y + i01 * lx1.y + i02 * lx2.y;
    const double h1x = i01 * lx0.x + i11 * lx1.x + i12 * lx2.x, h1y = i01 * lx0.y + i11 * lx1.y + i12 * lx2.y;
    const double h2x = i02 * lx0.x + i12 * lx1.x + i22 * lx2.x, h2y = i02 * lx0.y + i12 * lx1.y + i22 * lx2.y;
    double t00 = kx0.x * h0x + kx1.x * h1x + kx2.x * h2x, t01 = kx0.x * h0y + kx1.x * h1y + kx2.x * h2y;
    double t10 = kx0.y * h0x + kx1.y * h1x + kx2.y * h2x, t11 = kx0.y * h0y + kx1.y * h1y + kx2.y * h2y;
    double w0 = 0.0, w1 = 0.0, wgt = 1.0;
    if (DIAG) {  // the point row's tenth double is 1 for a point and 0 for the padding row, whose G_k term must vanish too
      wgt = pb[9];
      t00 -= 0.5 * wgt;
      t11 -= 0.5 * wgt;
      const double2 e = reinterpret_cast<const double2 *>(buf + LB_OFF)[lane];
      w0 = (kx0.x * pb[6] + kx1.x * pb[7] + kx2.x * pb[8] - e.x) * wgt;
      w1 = (kx0.y * pb[6] + kx1.y * pb[7] + kx2.y * pb[8] - e.y) * wgt;
    }
    // the nine columns of J_Cl (f | u, v | t | omega; signs and 1 / f0 are applied at the end): v_j = t (sx_j, sy_j)
    const double2 lf = lr[3], lw0 = lr[4], lw1 = lr[5], lw2 = lr[6];
    const double sx[9] = {lf.x, 1.0, 0.0, lx0.x, lx1.x, lx2.x, lw0.x, lw1.x, lw2.x};
    const double sy[9] = {lf.y, 0.0, 1.0, lx0.y, lx1.y, lx2.y, lw0.y, lw1.y, lw2.y};
    double v0[9], v1[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      if (j == 1) { v0[j] = t00; v1[j] = t10; }
      else if (j == 2) { v0[j] = t01; v1[j] = t11; }
      else { v0[j] = t00 * sx[j] + t01 * sy[j]; v1[j] = t10 * sx[j] + t11 * sy[j]; }
      if (DIAG) {
        dg[j] = fma(wgt, sx[j] * sx[j] + sy[j] * sy[j], dg[j]);
        rb[j] += sx[j] * w0 + sy[j] * w1;
      }
    }
    const double2 kf = kr[3];
#if defined(MVBA_KO_VALU)  // (timing-only: every LDS read stays, the arithmetic shrinks to a handful of additions)
    {
      const double2 kw0 = kr[4], kw1 = kr[5], kw2 = kr[6];
      double a = (kf.x + kf.y) + (kw0.x + kw0.y) + (kw1.x + kw1.y) + (kw2.x + kw2.y) + w0 + w1;
#pragma unroll
      for (int j = 0; j < 9; ++j) a += v0[j] + v1[j];
      acc[0][0] += a;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      return;
    }
#endif
    // rows of J_Ck: f | u, v | t | omega; two chained FMAs into the accumulator per entry
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      acc[0][j] = fma(kf.y, v1[j], fma(kf.x, v0[j], acc[0][j]));
      acc[1][j] += v0[j];
      acc[2][j] += v1[j];
      acc[3][j] = fma(kx0.y, v1[j], fma(kx0.x, v0[j], acc[3][j]));
      acc[4][j] = fma(kx1.y, v1[j], fma(kx1.x, v0[j], acc[4][j]));
      acc[5][j] = fma(kx2.y, v1[j], fma(kx2.x, v0[j], acc[5][j]));
    }
    const double2 kw0 = kr[4], kw1 = kr[5], kw2 = kr[6];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      acc[6][j] = fma(kw0.y, v1[j], fma(kw0.x, v0[j], acc[6][j]));
      acc[7][j] = fma(kw1.y, v1[j], fma(kw1.x, v0[j], acc[7][j]));
      acc[8][j] = fma(kw2.y, v1[j], fma(kw2.x, v0[j], acc[8][j]));
    }
    // the LDS reads above are complete (their values were consumed) before this buffer is refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // prologue = the iterations s = -2, -1 without a step to compute
  dma_idx(0, 0);
  dma_idx(1, RING_B);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  issue_step(0, 0, 2);
  issue_step(RING_B, BUFSZ, 3);
  unsigned b0 = 0, b1 = BUFSZ, b2 = 2 * BUFSZ, r0 = 0, r1 = RING_B;  // buffers of step st, st + 1, st + 2; ring slots of step st + 2, st + 3
  for (int st = 0; st < nst; ++st) {
    // step st has landed and the indices of step st + 2 are in the ring: everything but the last iteration's operations is done
#if defined(MVBA_KO_DMA)
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#else
    if (DIAG) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
#endif
    pace_at(st);
    issue_step(r0, b2, st + 4);  // step st + 2 (past the end: the clamped last step once more, into a buffer nobody reads)
    compute(wbuf + b0);
    { const unsigned tb = b0, tr = r0; b0 = b1; b1 = b2; b2 = tb; r0 = r1; r1 = tr; }
  }
  static_assert(G == (DIAG ? 13 : 17), "counted wait");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped gathers still in flight land in this wave's LDS
  if (pace.prog != nullptr && lane == 0)
    for (; seg < pace.nseg; ++seg) __hip_atomic_fetch_add(pace.prog + PACE_STRIDE * seg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // J_C row i = rs_i * (record columns): f | u,v (1/f0) | t (-Jx) | omega; the same factors per column
  const int u = slot_unit[lane];
  if (u >= 0) {
    double *o = out + (size_t)u * UNIT_STRIDE;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double rs = (i == 1 || i == 2) ? cu : ((i >= 3 && i < 6) ? -1.0 : 1.0);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const double cs = (j == 1 || j == 2) ? cu : ((j >= 3 && j < 6) ? -1.0 : 1.0);
        o[9 * i + j] = -4.0 * rs * cs * acc[i][j];
      }
    }
    if (DIAG) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const double cs = (j == 1 || j == 2) ? cu : ((j >= 3 && j < 6) ? -1.0 : 1.0);
        o[81 + j] = 2.0 * c * cs * cs * dg[j];  // c * diag(G_k)   (ref :123-125)
        o[90 + j] = 2.0 * cs * rb[j];           // 2 Jc_k^T (Jx_k E^-1 dP - e)
      }
    }
  }
}
